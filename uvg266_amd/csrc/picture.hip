// "picture" strategy group on gfx950: SAD / SATD / SSD / residual.
//
// Reference behaviour reproduced (bit-exact, integer):
//   reg_sad / hor_sad / ver_sad / cor_sad via uvg_image_calc_sad
//       src/strategies/generic/picture-generic.c:99,1266,1308, src/image.c:280-473
//   satd_4x4..64x64, satd_any_size(_quad), *_dual
//       src/strategies/generic/picture-generic.c:118-479, src/strategies/strategies-picture.h:54-109
//   pixels_calc_ssd :1115, generate_residual :1360
//
// Layout: planes live in HBM as the encoder keeps them (row-major, arbitrary
// stride).  One launch handles n blocks of ONE size (the caller buckets PUs by
// size, as the search does).  A block is owned by a power-of-two group of
// lanes inside one wavefront (4x4: 4-8 lanes ... >=16x16 SATD / >=512 px SAD:
// the whole 64-lane wave), partial costs are combined with wave-level
// butterflies, one 4-byte store per block.
#include "uvghip_common.h"
#include "percall.h"
#include "satd_dev.h"

// ------------------------------------------------------------------ SAD ----

template <typename PX> __device__ __forceinline__ int sad_words(uint32_t a, uint32_t b, int acc);
template <> __device__ __forceinline__ int sad_words<uint8_t>(uint32_t a, uint32_t b, int acc)
{
  return (int)__builtin_amdgcn_sad_u8(a, b, (uint32_t)acc);   // v_sad_u8: 4 packed bytes
}
template <> __device__ __forceinline__ int sad_words<uint16_t>(uint32_t a, uint32_t b, int acc)
{
  return (int)__builtin_amdgcn_sad_u16(a, b, (uint32_t)acc);  // v_sad_u16: 2 packed halves
}

// UNIT pixels of a row, packed in dwords; edge-replicated when CLAMP.
template <typename PX, int UNIT>
__device__ __forceinline__ void load_packed(const PX *plane, int stride, int W, int H, int x, int y, bool clamp,
                                            uint32_t (&w)[UNIT * sizeof(PX) / 4])
{
  constexpr int NW = UNIT * sizeof(PX) / 4;
  constexpr int PPW = 4 / sizeof(PX);   // pixels per dword
  const bool inside = !clamp || (x >= 0 && x + UNIT <= W && y >= 0 && y < H);
  if (inside) {
    const PX *p = plane + (size_t)y * stride + x;
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = reinterpret_cast<const u32_unaligned *>(p)[i];
  } else {
    const PX *row = plane + (size_t)clampi(y, 0, H - 1) * stride;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      uint32_t acc = 0;
#pragma unroll
      for (int k = 0; k < PPW; ++k)
        acc |= (uint32_t)row[clampi(x + i * PPW + k, 0, W - 1)] << (k * 8 * sizeof(PX));
      w[i] = acc;
    }
  }
}

template <typename PX, int UNIT>
__global__ void __launch_bounds__(256)
sad_batch_kernel(const PX *__restrict__ cur, int cs, const PX *__restrict__ ref, int rs, int W, int H,
                 int bw, int bh, const uvghip_blk_t *__restrict__ blks, int n, uint32_t *__restrict__ out,
                 int lpb, int shift, int clamp)
{
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int bpw = 64 / lpb;
  const int blk = wave * bpw + lane / lpb;
  const int l = lane & (lpb - 1);
  const bool active = blk < n;
  int acc = 0;
  if (active) {
    const uvghip_blk_t b = blks[blk];
    if constexpr (UNIT == 1) {
      for (int u = l; u < bw * bh; u += lpb) {
        const int uy = u / bw, ux = u - uy * bw;
        const int a = cur[(size_t)(b.cur_y + uy) * cs + b.cur_x + ux];
        const int ry = clamp ? clampi(b.ref_y + uy, 0, H - 1) : b.ref_y + uy;
        const int rx = clamp ? clampi(b.ref_x + ux, 0, W - 1) : b.ref_x + ux;
        const int r = ref[(size_t)ry * rs + rx];
        acc += abs(a - r);
      }
    } else {
      constexpr int NW = UNIT * sizeof(PX) / 4;
      const int ux_n = bw / UNIT, units = ux_n * bh;
      for (int u = l; u < units; u += lpb) {
        const int uy = u / ux_n, ux = (u - uy * ux_n) * UNIT;
        uint32_t wa[NW], wb[NW];
        load_packed<PX, UNIT>(cur, cs, 0, 0, b.cur_x + ux, b.cur_y + uy, false, wa);
        load_packed<PX, UNIT>(ref, rs, W, H, b.ref_x + ux, b.ref_y + uy, clamp != 0, wb);
#pragma unroll
        for (int i = 0; i < NW; ++i) acc = sad_words<PX>(wa[i], wb[i], acc);
      }
    }
  }
  acc = group_sum(acc, lpb);
  if (active && l == 0) out[blk] = (uint32_t)acc >> shift;
}

static int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

template <typename PX>
static int launch_sad(const void *cur, int cs, const void *ref, int rs, int W, int H, int bw, int bh,
                      const uvghip_blk_t *blks, int n, uint32_t *out, int clamp, hipStream_t st)
{
  if (n <= 0) return 0;
  const int unit = (bw % 8 == 0) ? 8 : (bw % 4 == 0 ? 4 : 1);
  const int units = (bw / unit) * bh;
  int lpb = pow2ceil(units); if (lpb > 64) lpb = 64;
  const int bpw = 64 / lpb, waves = (n + bpw - 1) / bpw, grid = (waves + 3) / 4;
  const int shift = px_traits<PX>::depth - 8;
  const PX *c = (const PX *)cur, *r = (const PX *)ref;
  if (unit == 8) sad_batch_kernel<PX, 8><<<grid, 256, 0, st>>>(c, cs, r, rs, W, H, bw, bh, blks, n, out, lpb, shift, clamp);
  else if (unit == 4) sad_batch_kernel<PX, 4><<<grid, 256, 0, st>>>(c, cs, r, rs, W, H, bw, bh, blks, n, out, lpb, shift, clamp);
  else sad_batch_kernel<PX, 1><<<grid, 256, 0, st>>>(c, cs, r, rs, W, H, bw, bh, blks, n, out, lpb, shift, clamp);
  UVGHIP_CHECK_LAUNCH();
}

extern "C" int uvghip_sad_batch(int bitdepth, const void *cur, int cur_stride, const void *ref, int ref_stride,
                                int ref_w, int ref_h, int bw, int bh, const uvghip_blk_t *blks, int n,
                                uint32_t *out, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (bw < 1 || bh < 1 || bw > 128 || bh > 128) return uvghip_set_error(hipErrorInvalidValue, __func__);
  return bitdepth == 8 ? launch_sad<uint8_t>(cur, cur_stride, ref, ref_stride, ref_w, ref_h, bw, bh, blks, n, out, 1, uvghip_stream(stream))
                       : launch_sad<uint16_t>(cur, cur_stride, ref, ref_stride, ref_w, ref_h, bw, bh, blks, n, out, 1, uvghip_stream(stream));
}

// ----------------------------------------------------------------- SATD ----

template <typename PX>
struct plane_diff_loader {
  const PX *cur; int cs; const PX *ref; int rs; int W, H; int cx, cy, rx, ry; bool clamp;
  __device__ __forceinline__ void operator()(int x, int y, int (&d)[8]) const
  {
    int a[8], b[8];
    load_row<PX, 8>(cur, cs, cx + x, cy + y, a);
    if (clamp) load_row_clamped<PX, 8>(ref, rs, W, H, rx + x, ry + y, b);
    else load_row<PX, 8>(ref, rs, rx + x, ry + y, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = a[i] - b[i];
  }
  __device__ __forceinline__ void operator()(int x, int y, int (&d)[4]) const
  {
    int a[4], b[4];
    load_row<PX, 4>(cur, cs, cx + x, cy + y, a);
    if (clamp) load_row_clamped<PX, 4>(ref, rs, W, H, rx + x, ry + y, b);
    else load_row<PX, 4>(ref, rs, rx + x, ry + y, b);
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = a[i] - b[i];
  }
};

template <typename PX>
__global__ void __launch_bounds__(256)
satd_batch_kernel(const PX *__restrict__ cur, int cs, const PX *__restrict__ ref, int rs, int W, int H,
                  int bw, int bh, const uvghip_blk_t *__restrict__ blks, int n, uint32_t *__restrict__ out,
                  int lpb, int shift, int clamp)
{
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int bpw = 64 / lpb;
  const int blk = wave * bpw + lane / lpb;
  const int l = lane & (lpb - 1);
  const bool active = blk < n;
  const satd_tiling t = make_tiling(bw, bh);
  uvghip_blk_t b = {0, 0, 0, 0};
  if (active) b = blks[blk];
  plane_diff_loader<PX> ld{cur, cs, ref, rs, W, H, b.cur_x, b.cur_y, b.ref_x, b.ref_y, clamp != 0};
  const int total = satd_block<PX>(t, l, lpb, active, ld);
  if (active && l == 0) out[blk] = (uint32_t)total >> shift;
}

static int satd_lpb(int bw, int bh)
{
  const satd_tiling t = make_tiling(bw, bh);
  const int tasks = t.n8 + (t.n4c + t.n4r + 1) / 2;
  int lpb = 8 * pow2ceil(tasks < 1 ? 1 : tasks);
  return lpb > 64 ? 64 : lpb;
}

template <typename PX>
static int launch_satd(const void *cur, int cs, const void *ref, int rs, int W, int H, int bw, int bh,
                       const uvghip_blk_t *blks, int n, uint32_t *out, int clamp, int shift, hipStream_t st)
{
  if (n <= 0) return 0;
  const int lpb = satd_lpb(bw, bh);
  const int bpw = 64 / lpb, waves = (n + bpw - 1) / bpw, grid = (waves + 3) / 4;
  satd_batch_kernel<PX><<<grid, 256, 0, st>>>((const PX *)cur, cs, (const PX *)ref, rs, W, H, bw, bh, blks, n, out,
                                              lpb, shift, clamp);
  UVGHIP_CHECK_LAUNCH();
}

extern "C" int uvghip_satd_batch(int bitdepth, const void *cur, int cur_stride, const void *ref, int ref_stride,
                                 int ref_w, int ref_h, int bw, int bh, const uvghip_blk_t *blks, int n,
                                 uint32_t *out, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (bw < 4 || bh < 4 || (bw & 3) || (bh & 3) || bw > 128 || bh > 128)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  return bitdepth == 8 ? launch_satd<uint8_t>(cur, cur_stride, ref, ref_stride, ref_w, ref_h, bw, bh, blks, n, out, 1, 0, uvghip_stream(stream))
                       : launch_satd<uint16_t>(cur, cur_stride, ref, ref_stride, ref_w, ref_h, bw, bh, blks, n, out, 1, 2, uvghip_stream(stream));
}

// ------------------------------------------------------------------ SSD ----

template <typename PX>
__global__ void __launch_bounds__(256)
ssd_batch_kernel(const PX *__restrict__ a, int as, const PX *__restrict__ b, int bs, int bw, int bh,
                 const uvghip_blk_t *__restrict__ blks, int n, uint32_t *__restrict__ out, int lpb, int shift)
{
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int bpw = 64 / lpb;
  const int blk = wave * bpw + lane / lpb;
  const int l = lane & (lpb - 1);
  const bool active = blk < n;
  int acc = 0;
  if (active) {
    const uvghip_blk_t d = blks[blk];
    for (int u = l; u < bw * bh; u += lpb) {
      const int uy = u / bw, ux = u - uy * bw;
      const int diff = (int)a[(size_t)(d.cur_y + uy) * as + d.cur_x + ux] - (int)b[(size_t)(d.ref_y + uy) * bs + d.ref_x + ux];
      acc += diff * diff;
    }
  }
  acc = group_sum(acc, lpb);
  if (active && l == 0) out[blk] = (uint32_t)(acc >> shift);   // int accumulator, arithmetic shift (picture-generic.c:1119-1130)
}

extern "C" int uvghip_ssd_batch(int bitdepth, const void *a, int a_stride, const void *b, int b_stride,
                                int bw, int bh, const uvghip_blk_t *blks, int n, uint32_t *out, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (n <= 0) return 0;
  int lpb = pow2ceil(bw * bh); if (lpb > 64) lpb = 64;
  const int bpw = 64 / lpb, waves = (n + bpw - 1) / bpw, grid = (waves + 3) / 4;
  hipStream_t st = uvghip_stream(stream);
  if (bitdepth == 8) ssd_batch_kernel<uint8_t><<<grid, 256, 0, st>>>((const uint8_t *)a, a_stride, (const uint8_t *)b, b_stride, bw, bh, blks, n, out, lpb, 0);
  else ssd_batch_kernel<uint16_t><<<grid, 256, 0, st>>>((const uint16_t *)a, a_stride, (const uint16_t *)b, b_stride, bw, bh, blks, n, out, lpb, 4);
  UVGHIP_CHECK_LAUNCH();
}

// ------------------------------------------------------------- residual ----

template <typename PX>
__global__ void __launch_bounds__(256)
residual_plane_kernel(const PX *__restrict__ a, int as, const PX *__restrict__ b, int bs,
                      int16_t *__restrict__ res, int rstride, int w, int h)
{
  const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int y = blockIdx.y;
  if (x >= w || y >= h) return;
  const PX *pa = a + (size_t)y * as + x, *pb = b + (size_t)y * bs + x;
  int16_t *pr = res + (size_t)y * rstride + x;
  if (x + 4 <= w) {
    int va[4], vb[4];
    load4(pa, va); load4(pb, vb);
#pragma unroll
    for (int i = 0; i < 4; ++i) pr[i] = (int16_t)(va[i] - vb[i]);
  } else {
    for (int i = 0; x + i < w; ++i) pr[i] = (int16_t)((int)pa[i] - (int)pb[i]);
  }
}

extern "C" int uvghip_residual_plane(int bitdepth, const void *a, int a_stride, const void *b, int b_stride,
                                     int16_t *res, int res_stride, int w, int h, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (w <= 0 || h <= 0) return 0;
  dim3 grid(((w + 3) / 4 + 255) / 256, h);
  hipStream_t st = uvghip_stream(stream);
  if (bitdepth == 8) residual_plane_kernel<uint8_t><<<grid, 256, 0, st>>>((const uint8_t *)a, a_stride, (const uint8_t *)b, b_stride, res, res_stride, w, h);
  else residual_plane_kernel<uint16_t><<<grid, 256, 0, st>>>((const uint16_t *)a, a_stride, (const uint16_t *)b, b_stride, res, res_stride, w, h);
  UVGHIP_CHECK_LAUNCH();
}

// -------------------------------------------------- full-search SAD surface ----
//
// One workgroup (256 threads) per bw x bh block.  The (bw+2r) x (bh+2r)
// search window is staged once in LDS with edge replication (so border
// candidates equal uvg_image_calc_sad's hor/ver/cor_sad result), the current
// block is staged next to it, and each thread then walks candidates
// c = tid, tid+256, ...  HBM traffic per block: window + block, read once.
template <typename PX>
__global__ void __launch_bounds__(256)
sad_surface_kernel(const PX *__restrict__ cur, int cs, const PX *__restrict__ ref, int rs, int W, int H,
                   int bw, int bh, int range, int blocks_x, uint32_t *__restrict__ out, int shift)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  PX *win = reinterpret_cast<PX *>(smem_raw);
  const int ww = bw + 2 * range, wh = bh + 2 * range;
  const int wpitch = ww + 1;   // odd pitch: candidate rows of neighbouring threads hit different banks
  PX *blk = win + wpitch * wh;
  const int bidx = blockIdx.x;
  const int by = (bidx / blocks_x) * bh, bx = (bidx % blocks_x) * bw;
  for (int i = threadIdx.x; i < ww * wh; i += blockDim.x) {
    const int y = i / ww, x = i - y * ww;
    win[y * wpitch + x] = ref[(size_t)clampi(by - range + y, 0, H - 1) * rs + clampi(bx - range + x, 0, W - 1)];
  }
  for (int i = threadIdx.x; i < bw * bh; i += blockDim.x) {
    const int y = i / bw, x = i - y * bw;
    blk[i] = cur[(size_t)(by + y) * cs + bx + x];
  }
  __syncthreads();
  const int side = 2 * range + 1, ncand = side * side;
  for (int c = threadIdx.x; c < ncand; c += blockDim.x) {
    const int dy = c / side, dx = c - dy * side;
    int acc = 0;
    for (int y = 0; y < bh; ++y) {
      const PX *wr = win + (y + dy) * wpitch + dx;
      const PX *br = blk + y * bw;
      for (int x = 0; x < bw; ++x) acc += abs((int)br[x] - (int)wr[x]);
    }
    out[(size_t)bidx * ncand + c] = (uint32_t)acc >> shift;
  }
}

// 8-bit fast path: v_qsad_pk_u16_u8.  One QSAD takes 8 consecutive reference bytes and 4 current bytes and returns the
// four SADs at byte offsets 0..3 -- four neighbouring candidates of a 4-sample segment per instruction -- accumulated
// in four packed u16.  A lane owns (dy, group of four dx): per block row it reads bw/4 + 1 aligned reference dwords of
// window row y + dy and bw/4 current dwords (the same address for every lane: a broadcast), and issues bw/4 QSADs.
// 64 QSADs can add at most 64 * 4 * 255 = 65280 to a u16 lane, so the packed sums are folded into 32-bit totals every
// 64 instructions.  All LDS reads are aligned dwords; candidates that differ by a sub-dword shift never cost an
// unaligned access.
template <int BW>
__global__ void __launch_bounds__(128)
sad_surface_qsad_kernel(const uint8_t *__restrict__ cur, int cs, const uint8_t *__restrict__ ref, int rs, int W, int H,
                        int bh, int range, int blocks_x, uint32_t *__restrict__ out)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int SEG = BW / 4;
  const int side = 2 * range + 1, groups = (side + 3) >> 2;
  const int ww = BW + 2 * range, wh = bh + 2 * range;
  const int pitch = ((ww + 3) >> 2) + 2;                 // dwords per window row (+ the dword a QSAD window may run into)
  uint32_t *win = reinterpret_cast<uint32_t *>(smem_raw);
  uint32_t *blk = win + (size_t)pitch * wh;              // bh rows of SEG dwords
  const int bidx = blockIdx.x;
  const int by = (bidx / blocks_x) * bh, bx = (bidx % blocks_x) * BW;
  // stage the window with edge replication (= uvg_image_calc_sad's border handling), one dword per thread and step
  for (int i = threadIdx.x; i < pitch * wh; i += blockDim.x) {
    const int y = i / pitch, xd = i - y * pitch;
    const uint8_t *row = ref + (size_t)clampi(by - range + y, 0, H - 1) * rs;
    const int x = bx - range + 4 * xd;
    uint32_t v;
    if (x >= 0 && x + 3 < W) v = *reinterpret_cast<const u32_unaligned *>(row + x);
    else v = (uint32_t)row[clampi(x, 0, W - 1)] | ((uint32_t)row[clampi(x + 1, 0, W - 1)] << 8) |
             ((uint32_t)row[clampi(x + 2, 0, W - 1)] << 16) | ((uint32_t)row[clampi(x + 3, 0, W - 1)] << 24);
    win[i] = v;
  }
  for (int i = threadIdx.x; i < SEG * bh; i += blockDim.x) {
    const int y = i / SEG, s = i - y * SEG;
    blk[i] = *reinterpret_cast<const u32_unaligned *>(cur + (size_t)(by + y) * cs + bx + 4 * s);
  }
  __syncthreads();
  const int ntasks = side * groups;
  for (int t = threadIdx.x; t < ntasks; t += blockDim.x) {
    const int dy = t / groups, g = t - dy * groups;
    uint32_t tot[4] = {0, 0, 0, 0};
    unsigned long long acc = 0;
    int since = 0;
    for (int y = 0; y < bh; ++y) {
      const uint32_t *wr = win + (size_t)(y + dy) * pitch + g;
      uint32_t d[SEG + 1];
#pragma unroll
      for (int k = 0; k <= SEG; ++k) d[k] = wr[k];
#pragma unroll
      for (int sgm = 0; sgm < SEG; ++sgm)
        acc = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)d[sgm + 1] << 32) | d[sgm], blk[y * SEG + sgm], acc);
      since += SEG;
      if (since + SEG > 64 || y == bh - 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) tot[k] += (uint32_t)(acc >> (16 * k)) & 0xffffu;
        acc = 0; since = 0;
      }
    }
    uint32_t *o = out + ((size_t)bidx * side + dy) * side + 4 * g;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (4 * g + k < side) o[k] = tot[k];
  }
}

// 10-bit fast path: v_sad_u16 on packed sample pairs.  The window lives in LDS as pair rows (dword k of a row = samples k and
// k + 1, as in the intra search), so the pair a candidate needs at any horizontal offset is an aligned dword.  A lane owns
// (dy, four neighbouring dx): per block row and chunk of CH columns it reads the run of pair dwords 4g .. 4g + CH + 3 with
// aligned b128 reads (candidate dx = 4g + k uses dwords k, k + 2, ...), the chunk's CH/2 current pairs (same address in all
// lanes: a broadcast) and issues 4 * CH/2 v_sad_u16, each accumulating two absolute differences into a 32-bit sum.
template <int BW>
__global__ void __launch_bounds__(128)
sad_surface_pk16_kernel(const uint16_t *__restrict__ cur, int cs, const uint16_t *__restrict__ ref, int rs, int W, int H,
                        int bh, int range, int blocks_x, uint32_t *__restrict__ out, int shift)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int CH = BW >= 16 ? 16 : 8;                  // columns per chunk
  constexpr int NQ = (CH + 4 + 3) / 4;                   // b128 reads covering pair dwords 0 .. CH + 2 of a group
  const int side = 2 * range + 1, groups = (side + 3) >> 2;
  const int ww = BW + 2 * range, wh = bh + 2 * range;
  const int pitch = ((ww + 3) & ~3) + 8;                 // dwords per pair row, a multiple of 4; the last group may read past ww
  uint32_t *win = reinterpret_cast<uint32_t *>(smem_raw);
  uint32_t *blk = win + (size_t)pitch * wh;              // bh rows of BW / 2 pair dwords
  const int bidx = blockIdx.x;
  const int by = (bidx / blocks_x) * bh, bx = (bidx % blocks_x) * BW;
  for (int i = threadIdx.x; i < pitch * wh; i += blockDim.x) {
    const int y = i / pitch, k = i - y * pitch;
    const uint16_t *row = ref + (size_t)clampi(by - range + y, 0, H - 1) * rs;
    const int x = bx - range + k;
    win[i] = (uint32_t)row[clampi(x, 0, W - 1)] | ((uint32_t)row[clampi(x + 1, 0, W - 1)] << 16);
  }
  for (int i = threadIdx.x; i < (BW / 2) * bh; i += blockDim.x) {
    const int y = i / (BW / 2), j = i - y * (BW / 2);
    blk[i] = *reinterpret_cast<const u32_unaligned *>(cur + (size_t)(by + y) * cs + bx + 2 * j);
  }
  __syncthreads();
  const int ntasks = side * groups;
  for (int t = threadIdx.x; t < ntasks; t += blockDim.x) {
    const int dy = t / groups, g = t - dy * groups;
    uint32_t tot[4] = {0, 0, 0, 0};
    for (int y = 0; y < bh; ++y) {
      const uint32_t *wr = win + (size_t)(y + dy) * pitch + 4 * g;
      const uint32_t *br = blk + y * (BW / 2);
#pragma unroll
      for (int c0 = 0; c0 < BW; c0 += CH) {
        uint32_t d[4 * NQ], cpk[CH / 2];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const uint4 v = *reinterpret_cast<const uint4 *>(wr + c0 + 4 * q);
          d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int q = 0; q < CH / 8; ++q) {
          const uint4 v = *reinterpret_cast<const uint4 *>(br + c0 / 2 + 4 * q);
          cpk[4 * q] = v.x; cpk[4 * q + 1] = v.y; cpk[4 * q + 2] = v.z; cpk[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int j = 0; j < CH / 2; ++j) tot[k] = __builtin_amdgcn_sad_u16(d[k + 2 * j], cpk[j], tot[k]);
      }
    }
    uint32_t *o = out + ((size_t)bidx * side + dy) * side + 4 * g;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (4 * g + k < side) o[k] = tot[k] >> shift;
  }
}

extern "C" int uvghip_sad_surface(int bitdepth, const void *cur, int cur_stride, const void *ref, int ref_stride,
                                  int w, int h, int bw, int bh, int range, uint32_t *out, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (bw < 4 || bh < 4 || bw > 64 || bh > 64 || range < 0 || range > 64 || w < bw || h < bh)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  const int blocks_x = w / bw, blocks_y = h / bh;
  const size_t es = bitdepth == 8 ? 1 : 2;
  const size_t lds = ((size_t)(bw + 2 * range + 1) * (bh + 2 * range) + (size_t)bw * bh) * es;
  if (lds > 160 * 1024) return uvghip_set_error(hipErrorInvalidValue, "uvghip_sad_surface: window exceeds LDS");
  hipStream_t st = uvghip_stream(stream);
  if (bitdepth == 8 && (bw == 8 || bw == 16 || bw == 32 || bw == 64)) {
    const size_t pitch = (size_t)((bw + 2 * range + 3) / 4 + 2);
    const size_t ql = (pitch * (bh + 2 * range) + (size_t)(bw / 4) * bh) * 4;
    if (ql <= 64 * 1024) {
      const int nb = blocks_x * blocks_y;
#define QS(BWV) sad_surface_qsad_kernel<BWV><<<nb, 128, ql, st>>>((const uint8_t *)cur, cur_stride, (const uint8_t *)ref, ref_stride, w, h, bh, range, blocks_x, out)
      if (bw == 8) QS(8); else if (bw == 16) QS(16); else if (bw == 32) QS(32); else QS(64);
#undef QS
      UVGHIP_CHECK_LAUNCH();
    }
  }
  if (bitdepth != 8 && (bw == 8 || bw == 16 || bw == 32 || bw == 64)) {
    const size_t pitch = (size_t)(((bw + 2 * range + 3) & ~3) + 8);
    const size_t pl = (pitch * (bh + 2 * range) + (size_t)(bw / 2) * bh) * 4;
    if (pl <= 64 * 1024) {
      const int nb = blocks_x * blocks_y;
#define PK(BWV) sad_surface_pk16_kernel<BWV><<<nb, 128, pl, st>>>((const uint16_t *)cur, cur_stride, (const uint16_t *)ref, ref_stride, w, h, bh, range, blocks_x, out, bitdepth - 8)
      if (bw == 8) PK(8); else if (bw == 16) PK(16); else if (bw == 32) PK(32); else PK(64);
#undef PK
      UVGHIP_CHECK_LAUNCH();
    }
  }
  if (bitdepth == 8) {
    if (lds > 64 * 1024) UVGHIP_TRY(hipFuncSetAttribute((const void *)sad_surface_kernel<uint8_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    sad_surface_kernel<uint8_t><<<blocks_x * blocks_y, 256, lds, st>>>((const uint8_t *)cur, cur_stride, (const uint8_t *)ref, ref_stride, w, h, bw, bh, range, blocks_x, out, 0);
  } else {
    if (lds > 64 * 1024) UVGHIP_TRY(hipFuncSetAttribute((const void *)sad_surface_kernel<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    sad_surface_kernel<uint16_t><<<blocks_x * blocks_y, 256, lds, st>>>((const uint16_t *)cur, cur_stride, (const uint16_t *)ref, ref_stride, w, h, bw, bh, range, blocks_x, out, 2);
  }
  UVGHIP_CHECK_LAUNCH();
}

// ---- IBC hash and variance (rows a8: crc32c_4x4/8x8, pixel_var) -------------------------------
// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) over the bytes of a 4x4 / 8x8 block, row by row, initial
// value and final xor 0xFFFFFFFF (picture-generic.c:1371-1443; 10-bit blocks hash both bytes of every sample,
// low byte first).  The table is computed, entry by entry, in LDS.
__device__ __forceinline__ uint32_t crc32c_table_entry(uint32_t i)
{
  uint32_t c = i;
#pragma unroll
  for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
  return c;
}
template <typename PX>
__global__ void __launch_bounds__(256)
crc32c_kernel(const PX *__restrict__ plane, int stride, int size, const uvghip_tu_t *__restrict__ blks, int n, uint32_t *__restrict__ out)
{
  __shared__ uint32_t sT[256];
  sT[threadIdx.x] = crc32c_table_entry(threadIdx.x);
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const PX *b = plane + (size_t)blks[i].y * stride + blks[i].x;
  uint32_t crc = 0xFFFFFFFFu;
  for (int y = 0; y < size; ++y)
    for (int x = 0; x < size; ++x) {
      const uint32_t v = b[(size_t)y * stride + x];
      crc = (crc >> 8) ^ sT[(crc ^ v) & 0xFF];
      if constexpr (sizeof(PX) == 2) crc = (crc >> 8) ^ sT[(crc ^ (v >> 8)) & 0xFF];
    }
  out[i] = crc ^ 0xFFFFFFFFu;
}

extern "C" int uvghip_crc32c_batch(int bitdepth, const void *plane, int stride, int size, const uvghip_tu_t *blks, int n,
                                   uint32_t *out, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if ((bitdepth != 8 && bitdepth != 10) || (size != 4 && size != 8)) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n <= 0) return 0;
  hipStream_t st = uvghip_stream(stream);
  if (bitdepth == 8) crc32c_kernel<uint8_t><<<(n + 255) / 256, 256, 0, st>>>((const uint8_t *)plane, stride, size, blks, n, out);
  else crc32c_kernel<uint16_t><<<(n + 255) / 256, 256, 0, st>>>((const uint16_t *)plane, stride, size, blks, n, out);
  UVGHIP_CHECK_LAUNCH();
}

// Variance of n arrays of `len` samples (picture-generic.c:1334-1357): mean = sum / len and the mean of squared
// deviations, in double.  The reference accumulates the len squared deviations in index order; here each lane of a
// wave accumulates a strided subset and the 64 partial sums are added in a fixed tree, so the result can differ
// from the reference in the last bits (relative 1e-13 for realistic len): a floating-point row, compared with a
// tolerance, not bit-exact.
template <typename PX>
__global__ void __launch_bounds__(256)
pixel_var_kernel(const PX *__restrict__ arr, uint32_t len, int n, double *__restrict__ out)
{
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= n) return;
  const PX *a = arr + (size_t)wave * len;
  unsigned long long s = 0;
  for (uint32_t i = lane; i < len; i += 64) s += a[i];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += ((unsigned long long)__shfl_xor((uint32_t)(s >> 32), off, 64) << 32) | __shfl_xor((uint32_t)s, off, 64);
  const double mean = (double)s / (double)len;
  double v = 0;
  for (uint32_t i = lane; i < len; i += 64) { const double t = (double)a[i] - mean; v += t * t; }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  if (lane == 0) out[wave] = v / len;
}

extern "C" int uvghip_pixel_var_batch(int bitdepth, const void *arr, uint32_t len, int n, double *out, void *stream)
{
  UVGHIP_REQUIRE_READY();
  if ((bitdepth != 8 && bitdepth != 10) || len == 0) return uvghip_set_error(hipErrorInvalidValue, __func__);
  if (n <= 0) return 0;
  hipStream_t st = uvghip_stream(stream);
  const int grid = (n + 3) / 4;
  if (bitdepth == 8) pixel_var_kernel<uint8_t><<<grid, 256, 0, st>>>((const uint8_t *)arr, len, n, out);
  else pixel_var_kernel<uint16_t><<<grid, 256, 0, st>>>((const uint16_t *)arr, len, n, out);
  UVGHIP_CHECK_LAUNCH();
}

// =================================================== drop-in strategy layer ====
// Host-pointer functions with the reference typedefs
// (src/strategies/strategies-picture.h:112-158).  The reference fixes
// uvg_pixel at compile time; one set per depth is instantiated here and the
// registrar picks by `bitdepth`.

namespace {

template <typename PX>
unsigned percall_block_cost(int kind, const PX *a, const PX *b, int w, int h, unsigned sa, unsigned sb)
{
  percall_ctx *c = percall_get((size_t)2 * w * h * sizeof(PX) + 1024);
  const size_t oa = c->stage_block(a, sa, w, h, sizeof(PX));
  const size_t ob = c->stage_block(b, sb, w, h, sizeof(PX));
  const size_t od = c->take(sizeof(uvghip_blk_t)), oo = c->take(sizeof(uint32_t));
  *c->hp<uvghip_blk_t>(od) = uvghip_blk_t{0, 0, 0, 0};
  c->upload(0, c->used);
  int rc;
  if (kind == 0)      rc = launch_sad<PX>(c->dp<PX>(oa), w, c->dp<PX>(ob), w, w, h, w, h, c->dp<uvghip_blk_t>(od), 1, c->dp<uint32_t>(oo), 0, c->stream);
  else if (kind == 1) rc = launch_satd<PX>(c->dp<PX>(oa), w, c->dp<PX>(ob), w, w, h, w, h, c->dp<uvghip_blk_t>(od), 1, c->dp<uint32_t>(oo), 0, 0, c->stream);
  else                rc = uvghip_ssd_batch(px_traits<PX>::depth, c->dp<PX>(oa), w, c->dp<PX>(ob), w, w, h, c->dp<uvghip_blk_t>(od), 1, c->dp<uint32_t>(oo), c->stream);
  c->must(rc, "picture launch");
  c->download(oo, sizeof(uint32_t));
  c->sync();
  return *c->hp<uint32_t>(oo);
}

// Raw (unshifted) SAD of a host w x h block against a host reference region of rw x rh samples whose sample
// (rx, ry) lies under the block's top-left; positions outside the region are edge-replicated by the kernel's
// coordinate clamp.  reg_sad itself does not shift by depth-8 (picture-generic.c:99), hence shift 0.
template <typename PX>
unsigned percall_sad(const PX *pic, unsigned pic_stride, int w, int h, const PX *ref, unsigned ref_stride, int rw, int rh,
                     int rx, int ry)
{
  percall_ctx *c = percall_get(((size_t)w * h + (size_t)rw * rh) * sizeof(PX) + 1024);
  const size_t oa = c->stage_block(pic, pic_stride, w, h, sizeof(PX));
  const size_t ob = c->stage_block(ref, ref_stride, rw, rh, sizeof(PX));
  const size_t od = c->take(sizeof(uvghip_blk_t)), oo = c->take(sizeof(uint32_t));
  *c->hp<uvghip_blk_t>(od) = uvghip_blk_t{0, 0, rx, ry};
  c->upload(0, c->used);
  const int unit = (w % 8 == 0) ? 8 : (w % 4 == 0 ? 4 : 1);
  int lpb = pow2ceil((w / unit) * h); if (lpb > 64) lpb = 64;
  const PX *da = c->dp<PX>(oa), *db = c->dp<PX>(ob);
  const uvghip_blk_t *dd = c->dp<uvghip_blk_t>(od); uint32_t *dout = c->dp<uint32_t>(oo);
  if (unit == 8) sad_batch_kernel<PX, 8><<<1, 64, 0, c->stream>>>(da, w, db, rw, rw, rh, w, h, dd, 1, dout, lpb, 0, 1);
  else if (unit == 4) sad_batch_kernel<PX, 4><<<1, 64, 0, c->stream>>>(da, w, db, rw, rw, rh, w, h, dd, 1, dout, lpb, 0, 1);
  else sad_batch_kernel<PX, 1><<<1, 64, 0, c->stream>>>(da, w, db, rw, rw, rh, w, h, dd, 1, dout, lpb, 0, 1);
  if (hipGetLastError() != hipSuccess) c->fail("sad launch");
  c->download(oo, sizeof(uint32_t));
  c->sync();
  return *c->hp<uint32_t>(oo);
}

// reg_sad_func
template <typename PX>
unsigned reg_sad_hip(const PX *data1, const PX *data2, const int width, const int height,
                     const unsigned stride1, const unsigned stride2)
{
  if (width <= 0 || height <= 0) return 0;
  return percall_sad<PX>(data1, stride1, width, height, data2, stride2, width, height, 0, 0);
}

// ver_sad_func (picture-generic.c:1266): every block row against the one reference row -> a 1-row region.
template <typename PX>
uint32_t ver_sad_hip(const PX *pic_data, const PX *ref_data, int32_t block_width, int32_t block_height, uint32_t pic_stride)
{
  if (block_width <= 0 || block_height <= 0) return 0;
  return percall_sad<PX>(pic_data, pic_stride, block_width, block_height, ref_data, (unsigned)block_width, block_width, 1, 0, 0);
}

// hor_sad_func (picture-generic.c:1308): `left` columns replicate reference column `left`, else `right` columns
// replicate column width-right-1 (left wins when both are set, as in the reference's if/else-if).
template <typename PX>
uint32_t hor_sad_hip(const PX *pic_data, const PX *ref_data, int32_t width, int32_t height, uint32_t pic_stride,
                     uint32_t ref_stride, uint32_t left, uint32_t right)
{
  if (width <= 0 || height <= 0) return 0;
  if (left) return percall_sad<PX>(pic_data, pic_stride, width, height, ref_data + left, ref_stride, width - (int)left, height, -(int)left, 0);
  if (right) return percall_sad<PX>(pic_data, pic_stride, width, height, ref_data, ref_stride, width - (int)right, height, 0, 0);
  return percall_sad<PX>(pic_data, pic_stride, width, height, ref_data, ref_stride, width, height, 0, 0);
}

// get_optimized_sad_func (strategies-picture.h:128, optimized_sad_func_ptr_t.h:13): width-specialised SAD or NULL.
template <typename PX, int W>
uint32_t opt_sad_hip(const PX *const a, const PX *const b, const int32_t height, const uint32_t stride1, const uint32_t stride2)
{
  return height > 0 ? percall_sad<PX>(a, stride1, W, height, b, stride2, W, height, 0, 0) : 0;
}
template <typename PX>
void *get_optimized_sad_hip(int32_t width)
{
  switch (width) {
    case 4: return (void *)&opt_sad_hip<PX, 4>;    case 8: return (void *)&opt_sad_hip<PX, 8>;
    case 12: return (void *)&opt_sad_hip<PX, 12>;  case 16: return (void *)&opt_sad_hip<PX, 16>;
    case 24: return (void *)&opt_sad_hip<PX, 24>;  case 32: return (void *)&opt_sad_hip<PX, 32>;
    case 64: return (void *)&opt_sad_hip<PX, 64>;
    default: return nullptr;   // the caller then uses reg_sad (image.c:259-263)
  }
}

// crc32c_4x4_func / crc32c_8x8_func (strategies-picture.h:157-158): (const uvg_pixel *buf, uint32_t pic_stride)
template <typename PX, int N> uint32_t crc32c_hip(const PX *buf, uint32_t pic_stride)
{
  percall_ctx *c = percall_get((size_t)N * N * sizeof(PX) + 1024);
  const size_t oa = c->stage_block(buf, pic_stride, N, N, sizeof(PX));
  const size_t od = c->take(sizeof(uvghip_tu_t)), oo = c->take(sizeof(uint32_t));
  *c->hp<uvghip_tu_t>(od) = uvghip_tu_t{0, 0};
  c->upload(0, c->used);
  c->must(uvghip_crc32c_batch(px_traits<PX>::depth, c->dp<PX>(oa), N, N, c->dp<uvghip_tu_t>(od), 1, c->dp<uint32_t>(oo), c->stream), "crc32c");
  c->download(oo, sizeof(uint32_t));
  c->sync();
  return *c->hp<uint32_t>(oo);
}
// pixel_var_func (strategies-picture.h:150): (const uvg_pixel *buf, const uint32_t len)
template <typename PX> double pixel_var_hip(const PX *buf, const uint32_t len)
{
  percall_ctx *c = percall_get((size_t)len * sizeof(PX) + 1024);
  const size_t oa = c->take((size_t)len * sizeof(PX)), oo = c->take(sizeof(double));
  memcpy(c->hp<PX>(oa), buf, (size_t)len * sizeof(PX));
  c->upload(0, c->used);
  c->must(uvghip_pixel_var_batch(px_traits<PX>::depth, c->dp<PX>(oa), len, 1, c->dp<double>(oo), c->stream), "pixel_var");
  c->download(oo, sizeof(double));
  c->sync();
  return *c->hp<double>(oo);
}

// cost_pixel_nxn_func: contiguous NxN
template <typename PX, int N> unsigned sad_nxn_hip(const PX *b1, const PX *b2)
{
  return percall_block_cost<PX>(0, b1, b2, N, N, N, N);      // shifted by depth-8 (picture-generic.c:1063)
}
template <typename PX, int N> unsigned satd_nxn_hip(const PX *b1, const PX *b2)
{
  const unsigned raw = percall_block_cost<PX>(1, b1, b2, N, N, N, N);
  return N == 4 ? raw : raw >> (px_traits<PX>::depth - 8);    // satd_4x4 is unshifted (picture-generic.c:170)
}
// cost_pixel_any_size_func
template <typename PX>
unsigned satd_any_size_hip(int width, int height, const PX *b1, int s1, const PX *b2, int s2)
{
  return percall_block_cost<PX>(1, b1, b2, width, height, (unsigned)s1, (unsigned)s2) >> (px_traits<PX>::depth - 8);
}
// cost_pixel_any_size_multi_func, 4 predictors (picture-generic.c:412-479, caller search_inter.c:1172).  The
// reference's loop nest is not satd_any_size x 4: its 8x8 part starts at row (h - 4) % 8 taken *after* h -= 4 and
// its first-row strip restarts at column 0, so for h % 8 == 4 rows 0..h-5 are covered again and the last four rows
// only by nothing.  Bit-exactness means reproducing that: the cost is the sum of up to three rectangular regions
// (first column 4 x h, first row w' x 4, 8x8 body w' x h' at column w % 8, row 0), each of which the batch kernel's
// own any_size tiling covers with exactly the reference's tiles.  `num_modes` and `valid` are ignored as upstream.
__global__ void quad_finish_kernel(const uint32_t *__restrict__ parts, int shift, uint32_t *__restrict__ out)
{
  const int k = threadIdx.x;
  if (k < 4) out[k] = (parts[k] + parts[4 + k] + parts[8 + k]) >> shift;
}
template <typename PX>
void satd_any_size_quad_hip(int width, int height, const PX **preds, const int stride, const PX *orig, const int orig_stride,
                            unsigned /*num_modes*/, unsigned *costs_out, int8_t * /*valid*/)
{
  percall_ctx *c = percall_get((size_t)5 * width * height * sizeof(PX) + 4096);
  if (width < 4 || height < 4 || (width & 3) || (height & 3) || width > 64 || height > 64) c->fail("satd_any_size_quad: unsupported size");
  const size_t oo = c->stage_block(orig, (size_t)orig_stride, width, height, sizeof(PX));
  const size_t op = c->take((size_t)4 * width * height * sizeof(PX));
  for (int k = 0; k < 4; ++k)
    for (int y = 0; y < height; ++y)
      memcpy(c->hp<PX>(op) + ((size_t)k * height + y) * width, preds[k] + (size_t)y * stride, (size_t)width * sizeof(PX));
  const size_t ob = c->take(12 * sizeof(uvghip_blk_t)), oparts = c->take(12 * sizeof(uint32_t));
  memset(c->hp<uint32_t>(oparts), 0, 12 * sizeof(uint32_t));
  const int wmod = width % 8, w2 = wmod ? width - 4 : width, h2 = (height % 8) ? height - 4 : height;
  uvghip_blk_t *hb = c->hp<uvghip_blk_t>(ob);
  for (int k = 0; k < 4; ++k) {
    hb[k] = uvghip_blk_t{0, 0, 0, k * height};            // first column, first row: both planes from column 0
    hb[4 + k] = hb[k];
    hb[8 + k] = uvghip_blk_t{wmod, 0, wmod, k * height};  // 8x8 body
  }
  c->upload(0, c->used);
  const size_t ores = c->take(4 * sizeof(uint32_t));
  const PX *dorig = c->dp<PX>(oo), *dpred = c->dp<PX>(op);
  const uvghip_blk_t *db = c->dp<uvghip_blk_t>(ob);
  uint32_t *dparts = c->dp<uint32_t>(oparts);
  int rc = 0;
  if (wmod) rc |= launch_satd<PX>(dorig, width, dpred, width, width, 4 * height, 4, height, db, 4, dparts, 0, 0, c->stream);
  if ((height % 8) && w2 > 0) rc |= launch_satd<PX>(dorig, width, dpred, width, width, 4 * height, w2, 4, db + 4, 4, dparts + 4, 0, 0, c->stream);
  if (w2 > 0 && h2 > 0) rc |= launch_satd<PX>(dorig, width, dpred, width, width, 4 * height, w2, h2, db + 8, 4, dparts + 8, 0, 0, c->stream);
  c->must(rc, "satd quad launch");
  quad_finish_kernel<<<1, 64, 0, c->stream>>>(dparts, px_traits<PX>::depth - 8, c->dp<uint32_t>(ores));
  if (hipGetLastError() != hipSuccess) c->fail("satd quad finish");
  c->download(ores, 4 * sizeof(uint32_t));
  c->sync();
  for (int k = 0; k < 4; ++k) costs_out[k] = c->hp<uint32_t>(ores)[k];
}
// cost_pixel_nxn_multi_func (pred_buffer = PX (*)[32*32]); orig is the first Hadamard operand
template <typename PX, int N>
void sad_nxn_dual_hip(PX (*preds)[32 * 32], const PX *orig, unsigned num_modes, unsigned *costs_out)
{
  costs_out[0] = percall_block_cost<PX>(0, preds[0], orig, N, N, N, N);
  costs_out[1] = percall_block_cost<PX>(0, preds[1], orig, N, N, N, N);
}
template <typename PX, int N>
void satd_nxn_dual_hip(PX (*preds)[32 * 32], const PX *orig, unsigned num_modes, unsigned *costs_out)
{
  for (int k = 0; k < 2; ++k) {
    const unsigned raw = percall_block_cost<PX>(1, orig, preds[k], N, N, N, N);
    costs_out[k] = N == 4 ? raw : raw >> (px_traits<PX>::depth - 8);
  }
}
// pixels_calc_ssd_func
template <typename PX>
unsigned pixels_calc_ssd_hip(const PX *ref, const PX *rec, int ref_stride, int rec_stride, int width, int height)
{
  return percall_block_cost<PX>(2, ref, rec, width, height, (unsigned)ref_stride, (unsigned)rec_stride);
}
// generate_residual_func
template <typename PX>
void generate_residual_hip(const PX *ref_in, const PX *pred_in, int16_t *residual, int width, int height,
                           int ref_stride, int pred_stride)
{
  percall_ctx *c = percall_get((size_t)width * height * (2 * sizeof(PX) + 2) + 1024);
  const size_t oa = c->stage_block(ref_in, ref_stride, width, height, sizeof(PX));
  const size_t ob = c->stage_block(pred_in, pred_stride, width, height, sizeof(PX));
  const size_t oo = c->take((size_t)width * height * 2);
  c->upload(0, oo);
  c->must(uvghip_residual_plane(px_traits<PX>::depth, c->dp<PX>(oa), width, c->dp<PX>(ob), width,
                                c->dp<int16_t>(oo), width, width, height, c->stream), "residual");
  c->download(oo, (size_t)width * height * 2);
  c->sync();
  memcpy(residual, c->hp<int16_t>(oo), (size_t)width * height * 2);
}

template <typename PX>
int register_picture(void *opaque)
{
  int ok = 1;
#define REG(type, fn) ok &= uvghip_do_register(opaque, type, (void *)(fn))
  REG("reg_sad", (&reg_sad_hip<PX>));
  REG("ver_sad", (&ver_sad_hip<PX>));  REG("hor_sad", (&hor_sad_hip<PX>));
  REG("get_optimized_sad", (&get_optimized_sad_hip<PX>));
  REG("sad_4x4", (&sad_nxn_hip<PX, 4>));     REG("sad_8x8", (&sad_nxn_hip<PX, 8>));
  REG("sad_16x16", (&sad_nxn_hip<PX, 16>));  REG("sad_32x32", (&sad_nxn_hip<PX, 32>));
  REG("sad_64x64", (&sad_nxn_hip<PX, 64>));
  REG("satd_4x4", (&satd_nxn_hip<PX, 4>));   REG("satd_8x8", (&satd_nxn_hip<PX, 8>));
  REG("satd_16x16", (&satd_nxn_hip<PX, 16>)); REG("satd_32x32", (&satd_nxn_hip<PX, 32>));
  REG("satd_64x64", (&satd_nxn_hip<PX, 64>));
  REG("sad_4x4_dual", (&sad_nxn_dual_hip<PX, 4>));   REG("sad_8x8_dual", (&sad_nxn_dual_hip<PX, 8>));
  REG("sad_16x16_dual", (&sad_nxn_dual_hip<PX, 16>)); REG("sad_32x32_dual", (&sad_nxn_dual_hip<PX, 32>));
  // 64x64: pred_buffer rows are 32*32 samples, so preds[1] = preds[0] + 1024 overlaps -- same reads as upstream
  REG("sad_64x64_dual", (&sad_nxn_dual_hip<PX, 64>));  REG("satd_64x64_dual", (&satd_nxn_dual_hip<PX, 64>));
  REG("satd_4x4_dual", (&satd_nxn_dual_hip<PX, 4>)); REG("satd_8x8_dual", (&satd_nxn_dual_hip<PX, 8>));
  REG("satd_16x16_dual", (&satd_nxn_dual_hip<PX, 16>)); REG("satd_32x32_dual", (&satd_nxn_dual_hip<PX, 32>));
  REG("satd_any_size", (&satd_any_size_hip<PX>));
  REG("satd_any_size_quad", (&satd_any_size_quad_hip<PX>));
  REG("pixels_calc_ssd", (&pixels_calc_ssd_hip<PX>));
  REG("crc32c_4x4", (&crc32c_hip<PX, 4>));
  REG("crc32c_8x8", (&crc32c_hip<PX, 8>));
  REG("pixel_var", (&pixel_var_hip<PX>));
  REG("generate_residual", (&generate_residual_hip<PX>));
#undef REG
  return ok;
}

}  // namespace

// Not registered (left to generic/avx2 by priority): satd_any_size_vtm
// (double sqrt), bipred_average (takes lcu_t).
extern "C" int uvg_strategy_register_picture_hip(void *opaque, uint8_t bitdepth)
{
  if (!uvghip_ready() && uvghip_init(0) != 0) return 0;
  return bitdepth == 8 ? register_picture<uint8_t>(opaque) : register_picture<uint16_t>(opaque);
}
