// A host program over the C ABI alone (no Python, no torch): what an encoder's frame loop does with the library for a group of
// all-intra pictures -- allocate device planes, upload the sources, uvghip_loop_plan_create once, uvghip_loop_plan_run per group,
// download the filtered pictures.  Build:  make -C examples   (hipcc, links ../uvg266_amd/libuvg266hip.so)
// Usage:  closed_loop <width> <height> <bitdepth 8|10> <qp> <pictures> [in.yuv] [repeats] [out.nals]
//   out.nals: every picture's slice NAL + hash SEI (uvghip_picture_checksum, uvghip_write_picture_nals) one after the other -- behind the
//   encoder's parameter sets this is the encoder's .266 of the same pictures (tests/test_gpu_example.py)
//   without in.yuv a deterministic synthetic source is used (a moving gradient with texture; NOT layout.synthetic_yuv420).
// Prints per picture the CRC-32 of the source and of the output picture (Y, U, V) and the decided SAO types' histogram, then the
// rate of `repeats` further runs of the same plan.  tests/test_gpu_example.py runs it on a yuv file the tests wrote and compares the
// CRCs with the Python-driven path.
#include <hip/hip_runtime.h>
#include <chrono>
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
  if (argc < 6) { fprintf(stderr, "usage: %s width height bitdepth qp pictures [in.yuv] [repeats]\n", argv[0]); return 2; }
  const int W = atoi(argv[1]), H = atoi(argv[2]), depth = atoi(argv[3]), qp = atoi(argv[4]), n = atoi(argv[5]);
  const char *yuv = argc > 6 && strcmp(argv[6], "-") ? argv[6] : nullptr;
  const int repeats = argc > 7 ? atoi(argv[7]) : 0;
  const char *nals_path = argc > 8 ? argv[8] : nullptr;
  const size_t b = depth == 8 ? 1 : 2, ysz = (size_t)W * H * b, csz = ysz / 4, psz = ysz + 2 * csz;
  const int wc = (W + 63) / 64, hc = (H + 63) / 64, ctus = wc * hc;
  UVG_OK(uvghip_init(0));

  // the search's parameters as the encoder derives them for an intra picture (rate_control.c: lambda = 0.57 * 2^((qp - 12) / 3))
  uvghip_ctu_params_t P;
  memset(&P, 0, sizeof P);
  P.pic_w = W; P.pic_h = H; P.qp = qp; P.qp_c = qp; P.depth_min = 1; P.depth_max = 4; P.wpp = 1; P.combine_intra_cus = 1; P.rough_levels = 2;
  P.lambda = 0.57 * pow(2.0, (qp - 12) / 3.0); P.lambda_sqrt = sqrt(P.lambda); P.c_lambda = P.lambda; P.chroma_weight_u = P.chroma_weight_v = 1.0;
  P.c_lambda_tu = P.lambda;

  // host pictures
  std::vector<std::vector<uint8_t>> src(n, std::vector<uint8_t>(psz));
  FILE *f = yuv ? fopen(yuv, "rb") : nullptr;
  if (yuv && !f) { perror(yuv); return 1; }
  for (int i = 0; i < n; ++i) {
    if (f) { if (fread(src[i].data(), 1, psz, f) != psz) { fprintf(stderr, "short read\n"); return 1; } }
    else {
      const int maxv = (1 << depth) - 1;
      for (int p = 0; p < 3; ++p) {
        const int w = p ? W / 2 : W, h = p ? H / 2 : H;
        uint8_t *d = src[i].data() + (p == 0 ? 0 : ysz + (p - 1) * csz);
        for (int y = 0; y < h; ++y)
          for (int x = 0; x < w; ++x) {
            const int v = (int)(maxv * (0.5 + 0.25 * sin(0.05 * (x + 3 * i) + 0.3 * p) + 0.2 * sin(0.11 * y + 0.02 * x * (p + 1)))) + ((x * 7 + y * 13 + i) & 7);
            const int c = v < 0 ? 0 : v > maxv ? maxv : v;
            if (b == 1) d[(size_t)y * w + x] = (uint8_t)c; else ((uint16_t *)d)[(size_t)y * w + x] = (uint16_t)c;
          }
      }
    }
  }
  if (f) fclose(f);

  // device memory: per picture source, reconstruction, output planes; side information, levels, models
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
  void *ws;
  HIP_OK(hipMalloc(&ws, uvghip_loop_workspace_bytes(depth, n, W, H)));
  uvghip_loop_plan_t *plan;
  UVG_OK(uvghip_loop_plan_create(depth, &P, pics.data(), n, 3, ws, &plan));
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));

  for (int i = 0; i < n; ++i) HIP_OK(hipMemcpyAsync(dsrc[i], src[i].data(), psz, hipMemcpyHostToDevice, st));
  UVG_OK(uvghip_loop_plan_run(plan, st));
  std::vector<uint8_t> out(psz);
  const uint8_t *d_rows; const int32_t *d_row_bytes; int row_cap, n_rows;
  UVG_OK(uvghip_loop_plan_slice_data(plan, &d_rows, &d_row_bytes, &row_cap, &n_rows));
  std::vector<int32_t> row_bytes((size_t)n * n_rows);
  HIP_OK(hipMemcpyAsync(row_bytes.data(), d_row_bytes, row_bytes.size() * 4, hipMemcpyDeviceToHost, st));
  const int32_t *d_info;
  UVG_OK(uvghip_loop_plan_results(plan, &d_info, nullptr));
  std::vector<int32_t> info((size_t)n * ctus * 34);
  HIP_OK(hipMemcpyAsync(info.data(), d_info, info.size() * 4, hipMemcpyDeviceToHost, st));
  FILE *nf = nals_path ? fopen(nals_path, "wb") : nullptr;
  if (nals_path && !nf) { fprintf(stderr, "cannot write %s\n", nals_path); return 1; }
  uint32_t *d_sums;
  HIP_OK(hipMalloc(&d_sums, 3 * sizeof(uint32_t)));
  for (int i = 0; i < n; ++i) {
    HIP_OK(hipMemcpyAsync(out.data(), dout[i], psz, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    int types[3] = {0, 0, 0};
    for (int k = 0; k < ctus; ++k) types[info[((size_t)i * ctus + k) * 34] % 3]++;
    // the slice data: the rows' substreams one after the other (what follows the slice header in the .266)
    uint32_t scrc = 0; size_t sbytes = 0;
    std::vector<uint8_t> row;
    for (int r = 0; r < n_rows; ++r) {
      const int nb = row_bytes[(size_t)i * n_rows + r];
      row.resize(nb);
      HIP_OK(hipMemcpy(row.data(), d_rows + ((size_t)i * n_rows + r) * row_cap, nb, hipMemcpyDeviceToHost));
      scrc = crc32(row.data(), nb, scrc); sbytes += nb;
    }
    printf("picture %d src %08x out %08x sao luma none/band/edge %d/%d/%d slice data %zu bytes crc %08x\n", i, crc32(src[i].data(), psz),
           crc32(out.data(), psz), types[0], types[1], types[2], sbytes, scrc);
    if (nf) {
      // the picture's NAL units: the checksum of the final picture on the device, header + rows + SEI on the host
      uint32_t sums[3];
      UVG_OK(uvghip_picture_checksum(depth, dout[i], W, dout[i] + ysz, dout[i] + ysz + csz, W / 2, W, H, d_sums, st));
      HIP_OK(hipMemcpyAsync(sums, d_sums, sizeof sums, hipMemcpyDeviceToHost, st));
      std::vector<uint8_t> rows((size_t)n_rows * row_cap);
      HIP_OK(hipMemcpyAsync(rows.data(), d_rows + (size_t)i * n_rows * row_cap, rows.size(), hipMemcpyDeviceToHost, st));
      HIP_OK(hipStreamSynchronize(st));
      std::vector<uint8_t> nals(sbytes + 64 + 4 * (size_t)n_rows);
      size_t len = 0;
      UVG_OK(uvghip_write_picture_nals(i, 1, rows.data(), (size_t)row_cap, &row_bytes[(size_t)i * n_rows], n_rows, sums, nals.data(), nals.size(), &len));
      if (fwrite(nals.data(), 1, len, nf) != len) { fprintf(stderr, "short write\n"); return 1; }
    }
  }
  if (nf) fclose(nf);
  if (repeats > 0) {
    HIP_OK(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < repeats; ++r) UVG_OK(uvghip_loop_plan_run(plan, st));
    HIP_OK(hipStreamSynchronize(st));
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%d x %d pictures %dx%d %d-bit in %.3f s = %.2f pictures/s\n", repeats, n, W, H, depth, s, repeats * n / s);
  }
  uvghip_loop_plan_destroy(plan);
  return 0;
}
