// A host program over the C ABI alone: a -p 1 stream under --tiles <cols>x<rows> --wpp.  What the encoder's frame loop does with its tile
// states (encoder_state_encode's walk over the TILE children, src/encoderstate.c:1221), in one call per group of pictures:
// uvghip_tiles_plan_create once, per group upload + uvghip_tiles_plan_run + uvghip_tiles_plan_nals.
// Build:  make -C examples      Usage:  tiles <width> <height> <bitdepth 8|10> <qp> <pictures> <cols> <rows> <in.yuv> <out.nals>
//   out.nals: every picture's slice NAL (the entry points of all tiles' substreams) + hash SEI -- behind the encoder's parameter sets (its PPS
//   carries the grid) this is the .266 the encoder writes with the same --tiles (tests/test_gpu_example.py against tests/golden/ref_tiles_*).
// Prints the grid and per picture the bytes written and the CRC-32 of the output picture.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/uvg266_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define UVG_OK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s failed: %d (%s)\n", #x, rc_, uvghip_last_error()); return 1; } } while (0)

static uint32_t crc32(const uint8_t *p, size_t n, uint32_t crc = 0)
{
  static uint32_t tab[256];
  if (!tab[1]) for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1; tab[i] = c; }
  crc = ~crc;
  for (size_t i = 0; i < n; ++i) crc = tab[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
  return ~crc;
}

int main(int argc, char **argv)
{
  if (argc < 10) { fprintf(stderr, "usage: %s width height bitdepth qp pictures cols rows in.yuv out.nals\n", argv[0]); return 2; }
  const int W = atoi(argv[1]), H = atoi(argv[2]), depth = atoi(argv[3]), qp = atoi(argv[4]), n = atoi(argv[5]), cols = atoi(argv[6]), rows = atoi(argv[7]);
  const size_t b = depth == 8 ? 1 : 2, ysz = (size_t)W * H * b, csz = ysz / 4, psz = ysz + 2 * csz;
  const int wc = (W + 63) / 64, hc = (H + 63) / 64, ctus = wc * hc;
  setenv("GPU_MAX_HW_QUEUES", "8", 0);          // the tile sizes' launches run on their own streams; before the first HIP call
  UVG_OK(uvghip_init(0));

  std::vector<uvghip_rect_t> grid((size_t)cols * rows);
  std::vector<int32_t> first((size_t)cols * rows);
  UVG_OK(uvghip_tile_grid(W, H, cols, rows, grid.data(), first.data()));
  for (size_t t = 0; t < grid.size(); ++t) printf("tile %zu: %dx%d at (%d, %d), first CTU %d\n", t, grid[t].w, grid[t].h, grid[t].x, grid[t].y, first[t]);

  uvghip_ctu_params_t P;
  memset(&P, 0, sizeof P);
  P.pic_w = W; P.pic_h = H; P.qp = qp; P.qp_c = qp; P.depth_min = 1; P.depth_max = 4; P.wpp = 1; P.combine_intra_cus = 1; P.rough_levels = 2;
  P.lambda = 0.57 * pow(2.0, (qp - 12) / 3.0); P.lambda_sqrt = sqrt(P.lambda); P.c_lambda = P.lambda; P.chroma_weight_u = P.chroma_weight_v = 1.0;
  P.c_lambda_tu = P.lambda;

  std::vector<std::vector<uint8_t>> src(n, std::vector<uint8_t>(psz));
  FILE *f = fopen(argv[8], "rb");
  if (!f) { perror(argv[8]); return 1; }
  for (int i = 0; i < n; ++i) if (fread(src[i].data(), 1, psz, f) != psz) { fprintf(stderr, "short read\n"); return 1; }
  fclose(f);

  // the WHOLE pictures on the device; the plan cuts its tiles out of them (origin + the picture's strides)
  std::vector<uvghip_loop_picture_t> pics(n);
  std::vector<uint8_t *> dsrc(n), dout(n);
  for (int i = 0; i < n; ++i) {
    uint8_t *s, *r, *o;
    void *cu, *coeff, *models;
    HIP_OK(hipMalloc(&s, psz)); HIP_OK(hipMalloc(&r, psz)); HIP_OK(hipMalloc(&o, psz));
    HIP_OK(hipMalloc(&cu, (size_t)hc * 16 * wc * 16 * sizeof(uvghip_scu_t))); HIP_OK(hipMemset(cu, 0, (size_t)hc * 16 * wc * 16 * sizeof(uvghip_scu_t)));
    HIP_OK(hipMalloc(&coeff, (size_t)ctus * 6144 * 2)); HIP_OK(hipMalloc(&models, (size_t)ctus * 3 * UVGHIP_CTU_MODELS * 4));
    HIP_OK(hipMemset(r, 0, psz));
    dsrc[i] = s; dout[i] = o;
    uvghip_loop_picture_t &q = pics[i];
    memset(&q, 0, sizeof q);
    q.search.src_y = s; q.search.src_u = s + ysz; q.search.src_v = s + ysz + csz; q.search.src_stride = W; q.search.src_stride_c = W / 2;
    q.search.rec_y = r; q.search.rec_u = r + ysz; q.search.rec_v = r + ysz + csz; q.search.rec_stride = W; q.search.rec_stride_c = W / 2;
    q.search.cu = (uvghip_scu_t *)cu; q.search.cu_stride = wc * 16; q.search.coeff = (int16_t *)coeff; q.search.models = (uint32_t *)models;
    q.out_y = o; q.out_u = o + ysz; q.out_v = o + ysz + csz; q.out_stride = W; q.out_stride_c = W / 2;
  }
  const size_t ws_bytes = uvghip_tiles_workspace_bytes(depth, n, W, H, cols, rows);
  if (!ws_bytes) { fprintf(stderr, "the grid does not fit the picture\n"); return 1; }
  void *ws;
  HIP_OK(hipMalloc(&ws, ws_bytes));
  uvghip_tiles_plan_t *plan;
  UVG_OK(uvghip_tiles_plan_create(depth, &P, pics.data(), n, cols, rows, 3, ws, &plan));
  int n_tiles, n_classes, n_sub;
  UVG_OK(uvghip_tiles_plan_layout(plan, &n_tiles, &n_classes, &n_sub));
  printf("%d tiles of %d sizes, %d substreams per picture\n", n_tiles, n_classes, n_sub);
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  for (int i = 0; i < n; ++i) HIP_OK(hipMemcpyAsync(dsrc[i], src[i].data(), psz, hipMemcpyHostToDevice, st));
  UVG_OK(uvghip_tiles_plan_run(plan, st));
  std::vector<uint8_t> nals((size_t)n * (psz * 2 + 4096));
  std::vector<size_t> lens(n);
  UVG_OK(uvghip_tiles_plan_nals(plan, 0, n, 0, nals.data(), nals.size(), lens.data(), st));
  FILE *nf = fopen(argv[9], "wb");
  if (!nf) { perror(argv[9]); return 1; }
  std::vector<uint8_t> out(psz);
  size_t at = 0;
  for (int i = 0; i < n; ++i) {
    if (fwrite(nals.data() + at, 1, lens[i], nf) != lens[i]) { fprintf(stderr, "short write\n"); return 1; }
    at += lens[i];
    HIP_OK(hipMemcpy(out.data(), dout[i], psz, hipMemcpyDeviceToHost));
    printf("picture %d: %zu bytes of NAL units, output picture crc %08x\n", i, lens[i], crc32(out.data(), psz));
  }
  fclose(nf);
  uvghip_tiles_plan_destroy(plan);
  return 0;
}
