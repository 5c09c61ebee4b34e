// Device-side building blocks shared by the transform kernel (dct.hip) and the
// fused TU round-trip kernel (quant.hip).  See dct.hip for the design notes.
#pragma once
#include "uvghip_common.h"
#include "vvc_tables.h"

typedef short v2s __attribute__((ext_vector_type(2)));

enum { TR_DCT2 = 0, TR_DCT8 = 1, TR_DST7 = 2 };   // src/uvg266.h:235-237

struct tr_pass {
  int R, C, K;          // lines, outputs per line, taps
  int rmax, cmax, kmax; // lines processed / outputs kept / taps summed (rest -> 0)
  int shift;
};
struct tr_params {
  int w, h;
  int type_hor, type_ver;
  tr_pass f1, f2;       // forward: horizontal then vertical
  tr_pass i1, i2;       // inverse: vertical then horizontal
};

// Which 1-D kernels honour skip_line2 in the reference (dct-generic.c):
//   forward: DST7/DCT8 with n >= 8 zero rows >= cutoff (:1651,:1768,:2014,:2139,:2264,:2335);
//            every DCT2 kernel and the 4-point DST7/DCT8 ignore it.
//   inverse: only the 8-point DST7/DCT8 stop their sums at cutoff (:2280,:2351).
static inline bool tr_fwd_cut(int type, int n) { return type != TR_DCT2 && n >= 8; }
static inline bool tr_inv_cut(int type, int n) { return type != TR_DCT2 && n == 8; }
static inline int tr_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline bool tr_valid_dim(int v) { return v == 4 || v == 8 || v == 16 || v == 32; }

// shifts: dct-generic.c:724-725,735-736 / :2606-2607,2658-2662
static inline tr_params tr_make_params(int bitdepth, int th, int tv, int w, int h, int sw, int sh)
{
  tr_params P;
  P.w = w; P.h = h; P.type_hor = th; P.type_ver = tv;
  P.f1 = tr_pass{h, w, w, h, tr_fwd_cut(th, w) ? w - sw : w, w, tr_ilog2(w) - 1 + bitdepth - 8};
  P.f2 = tr_pass{w, h, h, w - sw, tr_fwd_cut(tv, h) ? h - sh : h, h, tr_ilog2(h) + 6};
  P.i1 = tr_pass{w, h, h, w - sw, h, tr_inv_cut(tv, h) ? h - sh : h, 7};
  P.i2 = tr_pass{h, w, w, h, w, tr_inv_cut(th, w) ? w - sw : w, 20 - bitdepth};
  return P;
}

__device__ __forceinline__ const int16_t *tr_matrix_dev(int type, int n)
{
  if (type == TR_DCT2) return n == 4 ? VVC_DCT2_4 : n == 8 ? VVC_DCT2_8 : n == 16 ? VVC_DCT2_16 : VVC_DCT2_32;
  if (type == TR_DCT8) return n == 4 ? VVC_DCT8_4 : n == 8 ? VVC_DCT8_8 : n == 16 ? VVC_DCT8_16 : VVC_DCT8_32;
  return n == 4 ? VVC_DST7_4 : n == 8 ? VVC_DST7_8 : n == 16 ? VVC_DST7_16 : VVC_DST7_32;
}

// LDS pitches: K+2 int16 per row keeps rows 4-byte aligned and walks the
// banks (row stride = K/2+1 dwords, odd) so lanes reading different rows at
// the same k do not collide.
__device__ __forceinline__ int tr_pitch(int k) { return k + 2; }

// LDS image of an n x n kernel matrix: B[c][k] = T[c][k] (forward) or T[k][c] (inverse).
__device__ __forceinline__ void tr_stage_matrix(int16_t *dst, int type, int n, bool transposed)
{
  const int16_t *T = tr_matrix_dev(type, n);
  const int p = tr_pitch(n);
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
    const int a = e / n, b = e - a * n;
    if (transposed) dst[b * p + a] = T[e]; else dst[a * p + b] = T[e];
  }
}

// One 1-D pass over all blocks held by the workgroup:
//   acc(r,c) = sum_{k<kmax} A[b][r][k] * B[c][k];  v = (acc + rnd) >> shift
// forward truncates v to int16 (dct-generic.c:411), inverse clips (:438).
// r_fast: consecutive lanes take consecutive r (B row broadcast) else consecutive c.
// epi(b, r, c, v) stores the result.
template <bool INVERSE, typename Epi>
__device__ __forceinline__ void tr_run_pass(const tr_pass &p, const int16_t *A, int a_blk, const int16_t *B,
                                            bool r_fast, int nblk_here, Epi epi)
{
  const int pa = tr_pitch(p.K);
  const int per_blk = p.R * p.C, total = per_blk * nblk_here;
  const int add = p.shift > 0 ? 1 << (p.shift - 1) : 0;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int b = e / per_blk, rem = e - b * per_blk;
    int r, c;
    if (r_fast) { c = rem / p.R; r = rem - c * p.R; } else { r = rem / p.C; c = rem - r * p.C; }
    int v = 0;
    if (r < p.rmax && c < p.cmax) {
      const int *a2 = reinterpret_cast<const int *>(A + b * a_blk + r * pa);
      const int *b2 = reinterpret_cast<const int *>(B + c * pa);
      int acc = 0;
#pragma unroll 4
      for (int k = 0; k < p.kmax / 2; ++k)
        acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a2[k]), __builtin_bit_cast(v2s, b2[k]), acc, false);
      v = (acc + add) >> p.shift;
      v = INVERSE ? clampi(v, -32768, 32767) : (int)(int16_t)v;
    }
    epi(b, r, c, v);
  }
}

// LDS needed (int16 elements) by a line buffer holding 1024 coefficients in any 4..32 shape.
#define TR_LINEBUF_ELEMS (1024 + 2 * 256 + 64)
#define TR_MATRIX_ELEMS (32 * 34)
