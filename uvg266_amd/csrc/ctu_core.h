// Closed-loop intra search of one CTU by one workgroup (uvg_search_lcu / search_cu, src/search.c:1299-2479, for the all-intra
// --preset medium configuration: quad-tree depths pu-depth-intra min..max, rd = 0, RDOQ, no MTS / LFNST / ISP / MRL / MIP / CCLM /
// JCCR / transform skip / dual tree, WPP), followed by the real coder's model adaptation (uvg_encode_coding_tree) so that the next
// CTU starts from the models the reference would hand it.
//
// NOT a translation of the reference's work tree.  The reference recurses with one lcu_t copy per depth and copies winners up;
// here ONE set of "decided" planes (D) lives in LDS with a one-sample border holding the neighbouring CTUs' samples, so every
// block's intra references -- inside or across the CTU edge -- are fetched the same way.  A CU is reconstructed straight into
// D; if its split will be tried the result is parked in a per-depth candidate buffer, the children overwrite D, and the parked
// candidate is put back only if the split loses (work_tree_copy_up inverted).  The walk is an explicit depth-first loop, not
// recursion.  Dead work of the reference is not done: the chroma blocks are reconstructed once per CU (the reference does it
// twice with the same inputs, search.c:1496 and :1572) and once per 8x8 area of 4x4 CUs (the reference redoes it for each of
// the four, only the last survives).
//
// Execution model: regions that touch many samples run on all lanes (PAR_FOR: predictions + SATD of the rough search, residual,
// transforms, dequantisation, reconstruction, SSD, copies); regions that are a recurrence through the CABAC models or through
// floating-point sums whose order the reference fixes (RDOQ walk, coefficient bit cost, mode bits, the RD comparisons) run on
// lane 0 (SERIAL).  Every hand-over between regions is a workgroup barrier.  The same source compiles for the host with one
// "lane" and no barriers -- the CPU tests run that emulation against the CPU restatement of the reference (tests/emul/); the product only ever runs the
// device build.
//
// All RD arithmetic is IEEE double in the reference's operation order; compile with -ffp-contract=off.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include "intra_pred_dev.h"
#if defined(__HIPCC__)
#include "satd_dev.h"
#endif
#include "vvc_tables.h"
#include "vvc_ctx_init.h"
#include "vvc_rdoq_tables.h"
#include "../../include/uvg266_hip.h"
#if defined(CTU_PB)
#include "inter_cand_dev.h"
#endif

#if defined(__HIPCC__)
#define CTU_NOINLINE __attribute__((noinline))
#define CTU_INLINE1 __attribute__((always_inline))          // one call site: no call, no callee-saved registers through the stack
#define CTU_DEV static __device__
// Everything the search computes is WAVE-local: a depth of the quad tree is worked by one wave (see search_ctu), so "all lanes"
// means the 64 lanes of that wave and a hand-over between regions is a wave-level fence, not a workgroup barrier.
#define CTU_TID ((int)(threadIdx.x & 63))
#define CTU_NT 64
// A wave's ROLE -- 0 walks the CTU, 1..3 evaluate the depths 3..1 -- is its index in the workgroup relative to S->rot, which the launcher
// picks so that the walkers of the workgroups sharing a CU spread over the SIMDs (four walkers on one SIMD cost 9 %,
// profiles/r04_simd_placement.txt).  The role must NOT be read off the hardware (HW_REG_HW_ID): with several queues busy the
// scheduler saves waves and restores them elsewhere, and a role that changes under a wave is a wrong CTU.  Where a wave sits
// when its workgroup starts is only a hint for S->rot.  `S` is in scope wherever the role is asked for.
#define CTU_WAVE (__builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 6) - S->rot) & 3))
#define CTU_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
// the few workgroup-wide regions (CTU load / store)
#define BLK_TID ((int)threadIdx.x)
#define BLK_NT ((int)blockDim.x)
#define BLK_SYNC() __syncthreads()
#else
#define CTU_NOINLINE
#define CTU_INLINE1
#define CTU_DEV static inline
#define CTU_TID 0
#define CTU_NT 1
#define CTU_WAVE (ctu::g_emul_wave)
#define CTU_SYNC() ((void)0)
#define BLK_TID 0
#define BLK_NT 1
#define BLK_SYNC() ((void)0)
#endif
// a load that must see what another wave of this workgroup stored to global memory earlier (served by L2, not by this CU's L1)
#if defined(__HIPCC__)
#define CTU_GLOAD(p) __builtin_nontemporal_load(p)
#define CTU_LDS __attribute__((address_space(3)))
#define CTU_GLB __attribute__((address_space(1)))
#else
#define CTU_GLOAD(p) (*(p))
#define CTU_LDS
#define CTU_GLB
#endif
// optional phase timers (lane 0 of the wave, s_memtime ticks) -- compiled in with -DCTU_PROFILE, results in scratch::prof[wave]
#if defined(__HIPCC__) && defined(CTU_PROFILE)
#define CTU_T0() const unsigned long long ctu_t0__ = __builtin_amdgcn_s_memtime()
#define CTU_T1(W, slot) do { if (CTU_TID == 0) (W)->prof[CTU_WAVE][slot] += __builtin_amdgcn_s_memtime() - ctu_t0__; } while (0)
#else
#define CTU_T0() ((void)0)
#define CTU_T1(W, slot) ((void)0)
#endif
#if defined(__HIPCC__) && defined(CTU_PROFILE)
#define RQ_T(slot) do { const unsigned long long t2 = __builtin_amdgcn_s_memtime(); if (CTU_TID == 0) W->prof[CTU_WAVE][slot] += t2 - tq; tq = t2; } while (0)
#else
#define RQ_T(slot) ((void)0)
#endif
#if defined(__HIPCC__) && defined(CTU_PROFILE)
#define LF_T(slot) do { const unsigned long long t2 = __builtin_amdgcn_s_memtime(); if (CTU_TID == 0 && CTU_WAVE == 0) J.W->prof_lf[slot] += t2 - tq; tq = t2; } while (0)
#define LF_T0() unsigned long long tq = __builtin_amdgcn_s_memtime()
#else
#define LF_T(slot) ((void)0)
#define LF_T0() ((void)0)
#endif
#define LDSP(T, p) ((CTU_LDS T *)(p))          // a pointer known to point into the workgroup's LDS image (device: ds_* instead of flat_*)
// ... and one that points into the LDS image OR, in a slim build (lds_cfg below), into the workgroup's global scratch: a generic pointer
// there (flat_*: the LDS aperture costs a lone wave the same 52 cycles as ds_read, profiles/r05_lat_probe.txt)
#define MGP(PX, T, p) ((typename ctu::mg_ptr<PX, T>::type)(p))
#define PAR_FOR(i, n) for (int i = CTU_TID; i < (n); i += CTU_NT)
#define BLK_FOR(i, n) for (int i = BLK_TID; i < (n); i += BLK_NT)
#define SERIAL if (CTU_TID == 0)
#define LANE0 if (CTU_TID == 0)
#define WFOR(i, n) PAR_FOR(i, n)
#define WSYNC() CTU_SYNC()

// the 4x4 leaves of an I picture's CTU go through the register-resident formulation of ctu_leaf4.h (device builds; -DCTU_LEAF_OLD keeps
// the general CU evaluation for them: the A/B build of tools/dev)
#if defined(__HIPCC__) && !defined(CTU_LEAF_OLD)
#define CTU_LEAF4 1
#endif
// ... what the I-picture kernel builds on top of it (the 8x8 CU's chroma blocks and the 64x64 candidate's chroma on other waves, the
// coder pass by 8x8 areas, the slim 10-bit image); the P / B kernel (ctu_pb.h) takes the 4x4 CU itself and the 4x4 bit count
#if defined(CTU_LEAF4) && !defined(CTU_PB)
#define CTU_LEAF4X 1
#endif

namespace ctu {

#if !defined(__HIPCC__)
static int g_emul_wave = 0;      // host emulation: which wave's scratch the code running now uses
static int g_emul_lazy = 0;      // host emulation: pretend a CU's own cost is never known before all its children are done
static int g_emul_depthwave = 0; // ... of its three-wave build: 32x32 / 16x16 CUs handed to the depth wave, the walk goes on into their children
static int g_emul_leafwave = 0;  // host emulation of the P / B kernel's two-wave build: the four 4x4 CUs of an 8x8 area go to the leaf wave (ctu_pb.h)
#endif

enum { LCU = 64, LCU_C = 32, PY = 68, PC = 36, NMODELS = 257 };
// P / B pictures (ctu_pb.h, compiled with -DCTU_PB): the 18 models of the inter syntax sit behind the 257 in every LDS model set
#if defined(CTU_PB)
enum { NMX = NMODELS + 18 };
#else
enum { NMX = NMODELS };
#endif
enum { M_SIGGRP = 0, M_SIG = 4, M_PAR = 28, M_GT1 = 70, M_GT2 = 112, M_LASTX = 154, M_LASTY = 194, M_CBF_LUMA = 234, M_CBF_CB = 238,
       M_CBF_CR = 240, M_SPLIT = 244, M_MPM = 253, M_PLANAR = 254, M_CHROMA_PRED = 256 };
enum { CU_NOTSET = 0, CU_INTRA = 1, CU_INTER = 2 };
#define CTU_MAX_DOUBLE 1.7976931348623157e308

// what the search reads from encoder_state_t / encoder_control_t (= uvghip_ctu_params_t, include/uvg266_hip.h)
struct params {
  int32_t pic_w, pic_h, qp, qp_c, depth_min, depth_max, wpp, combine_intra_cus, rough_levels, rd;       // rd: cfg.rdo, 0 or 1 (uvghip_ctu_params_t.rd)
  double lambda, lambda_sqrt, c_lambda, cw_u, cw_v;
  double c_lambda_tu;      // uvg_calculate_chroma_lambda (rate_control.c:1216-1233), evaluated by the host: lambda / 2^((qp - qp_c) / 3)
};

static_assert(sizeof(params) == sizeof(uvghip_ctu_params_t) && offsetof(params, rd) == offsetof(uvghip_ctu_params_t, rd) && offsetof(params, lambda) == offsetof(uvghip_ctu_params_t, lambda) &&
              offsetof(params, c_lambda_tu) == offsetof(uvghip_ctu_params_t, c_lambda_tu), "ctu::params mirrors uvghip_ctu_params_t field by field");

// one 4x4 unit of the CTU's side information while the search runs (the slice of cu_info_t this path reads back)
struct cu4 {
  uint8_t type, log2, cbf, luma_edges, chroma_edges, log2_c;
  int8_t mode, mode_chroma;
};

struct level_state {        // search_cu's locals, per depth
  double cost, split_cost, split_bits;
  int x, y;                 // picture coordinates
  int child;                // next child to visit
  int type, mode, cbf;      // the parked no-split candidate
  int has_chroma;           // carries the chroma of its area
  int pending;              // its unsplit evaluation was handed to the depth's wave
  int evalp, known, can;    // P / B depth wave (ctu_pb.h): evaluation posted; its result has been taken; can_inter | can_intra << 1 of the request
  uint32_t split_tree, mode_type_tree;
#if defined(CTU_PB)
  int32_t mot[8];           // the parked candidate's motion (icand::unit: type, mv[2][2], ref[2], dir)
  uint8_t fl[8];            // uvghip_inter4_t: skipped, merged, merge_idx, root_cbf, mv_cand0, mv_cand1, mv_ref0, mv_ref1
  int32_t cbf4[4];          // a 64x64 inter CU: the flags of its four transform units
#endif
};

// The LDS diet of the 10-bit I-picture kernel: with 2-byte samples the image is 8.9 KB over a quarter of the CU's 160 KB, which costs
// the fourth workgroup per CU.  A slim build keeps out of LDS what only the 32x32 depth (the least loaded wave) and the rare 64x64
// candidate touch: the depth-1 candidate's samples, the depth-1 scratch's levels, the 16x16 / 32x32 coefficient scans -- they live in
// the workgroup's global scratch (L1 / L2 resident) behind generic pointers.  8-bit, P / B and host builds are not slim.
// (slim_scan alone: only the two large scans move -- the P / B kernel, whose image also carries the inter state)
#if defined(CTU_PB) && defined(__HIPCC__) && !defined(CTU_NO_SLIM)
template <typename PX> struct lds_cfg { enum { slim = 0, slim_scan = 1 }; };
#else
template <typename PX> struct lds_cfg { enum { slim = 0, slim_scan = 0 }; };
#endif
#if defined(CTU_LEAF4X) && !defined(CTU_NO_SLIM)
template <> struct lds_cfg<uint16_t> { enum { slim = 1, slim_scan = 1 }; };
#endif
template <typename PX, typename T, bool SLIM = (lds_cfg<PX>::slim != 0)> struct mg_ptr { typedef CTU_LDS T *type; };
template <typename PX, typename T> struct mg_ptr<PX, T, true> { typedef T *type; };

template <typename PX> struct px_info;
template <> struct px_info<uint8_t> { enum { depth = 8, maxv = 255 }; };
template <> struct px_info<uint16_t> { enum { depth = 10, maxv = 1023 }; };

// LDS image of a workgroup
// scratch of ONE wave (= one depth of the quad tree): pointers into the workgroup's arena, sized for that depth's blocks, and the
// small fixed-size pieces inline
struct wctx {
  uint16_t *top, *left, *ftop, *fleft;              // reference rows: 4 n + 8 entries each
  int16_t *t0, *t1;                                 // transform scratch, n * n each, adjacent (together also the n * n words of coeff_bits)
  int16_t *lv0, *lv1, *lv2;                         // levels of the transform blocks being evaluated (y, u, v)
  double *rq_cc, *rq_cs;                            // RDOQ per-position costs in LDS (nullptr: the depth uses the workgroup's global scratch)
  uint32_t *part;                                   // rough search: (satd, sad) per (listed mode, tile)
  uint32_t *cur;                                    // the models this wave's bit counting works on
  union {                                           // a wave is in one phase at a time
    double rq_stage[2 * 16];                        // RDOQ: costs of the coefficient group in flight (also the coefficient bit count's 64 group totals)
    struct { double rs_cand[3 + 24], rs_best_cost[2][3]; };      // rough search: costs of the survivors and of the listed modes; the survivors, double-buffered
  };
  double u_d0, u_d1;
  int32_t rq_i[16];
  int32_t rs_list[24], rs_best_mode[2][3];
  uint32_t rs_chk[3];
  int32_t red[8];
  int32_t u_avail_left, u_avail_top, u_mode, u_flag, u_n_modes;
  uint8_t cg_flag[64];
  int8_t mpm[6];
  int16_t refn;                                     // entries of a reference row
#if defined(CTU_PB)
  int32_t rq_root;                                  // RDOQ prices a luma block's "no coefficients" with the root cbf (a block of an inter CU, rdo.c:1774)
#endif
};
#if defined(CTU_PB)
#define CTU_RQ_ROOT(V) ((V)->rq_root)
#else
#define CTU_RQ_ROOT(V) 0
#endif

constexpr int arena_bytes(int n, bool slim = false)     // one depth's share of the arena (n = its luma block size)
{
  if (slim && n == 32) return (4 * (4 * n + 8) * 2 + 2 * n * n * 2 + 15) & ~15;      // (the levels live in scratch::lv32)
  // reference rows, two transform buffers, levels (y, u, v), [4x4 only: the rough search's partial costs -- larger blocks keep
  // them in the transform buffers, idle during the rough search], [<= 8x8: RDOQ's two per-position cost arrays]
#if defined(CTU_LEAF4)
  // (the 4x4 leaf keeps the rough search's costs and RDOQ's per-position costs in registers: reference rows, transform buffers, levels only)
  if (n == 4) return (4 * (4 * n + 8) * 2 + 2 * n * n * 2 + (n * n + 2 * 16) * 2 + 15) & ~15;
#endif
  return (4 * (4 * n + 8) * 2 + 2 * n * n * 2 + (n * n + 2 * ((n / 2) * (n / 2) < 16 ? 16 : (n / 2) * (n / 2))) * 2 +
          (n >= 8 ? 0 : 2 * 18 * 4) + (n <= 8 ? 8 + 2 * n * n * 8 : 0) + 15) & ~15;
}
#if defined(CTU_PB)
// one wave walks a P / B CTU, a depth at a time: the depths' scratch areas share ONE region sized for the largest (ctu_pb.h); the 4x4
// depth has its own behind it -- in the two-wave build the leaf wave works there while the walk evaluates the 8x8 CU above
enum { ARENA_BYTES = arena_bytes(32) + arena_bytes(4) };
#else
enum { ARENA_BYTES = arena_bytes(4) + arena_bytes(8) + arena_bytes(16) + arena_bytes(32) };
#endif
enum { ARENA_BYTES_SLIM = arena_bytes(4) + arena_bytes(8) + arena_bytes(16) + arena_bytes(32, true) };

struct scratch;
#if defined(CTU_PB)
// ---- P / B pictures: what the inter search keeps beside the intra state (ctu_pb.h) ----
struct pb_cand {            // one entry of the reference's unit_stats_map_t: a candidate motion with its cost
  icand::unit m;
  uint8_t merged, skipped, merge_idx, cand0, cand1, pad[3];
  double cost, bits;
};
struct pb_state {
  // motion of every 4x4 unit of the CTU and of the row / column before it ([289]: the CTU above right, unused with WPP) and the
  // uvghip_inter4_t of the same units: in the workgroup's global scratch (scratch::pb_mot / pb_fl) -- 11.6 KB of LDS were the fourth
  // workgroup per CU, and the tables are read a few entries per candidate list
  icand::unit *mot;
  uint8_t (*fl)[8];
  const int32_t *hm;                   // the history table the evaluation in progress reads: the one at the entry of its node (hmvp_entry[L])
  int32_t hmvp[41];                    // the row's history table as the search sees it: [0] entries, then 5 units, most recent first
  int32_t hmvp_entry[4][41];           // ... at the entry of the node of each depth 0..3 (search_cu's hmvp_lut)
  int32_t hmvp_coder[41];              // ... as the real coder leaves it (what the next CTU of the row starts from)
  uint32_t work0[NMX];                 // the models the 64x64 candidate works on (work[L - 1] of depth 0)
  uint32_t cnt_models[NMX];            // a counting-only bit cost adapts the last-position / group-flag models on this copy
  icand::frame_ctx f;                  // the call context of the candidate derivations
  icand::amvp_ws ws;
  icand::merge_cand mc[6];
  int32_t n_mc;
  pb_cand cur;                         // search_pu_inter's cur_pu
  pb_cand merge[6], amvp[3][8];
  int8_t merge_keys[8], amvp_keys[3][8];
  int32_t merge_size, amvp_size[3];
  int32_t mv_cand[2][2];               // info->mv_cand
  int32_t amvp_cache[2][8][4];         // the predictors of (list, reference index) for the CU being evaluated: derived once
  uint32_t amvp_have[2];
  int32_t amvp_key[3];                 // the CU (x, y, size) the cache belongs to
  int32_t colc[2][8];                  // the two positions of the collocated picture a CU's temporal candidate can come from, fetched once per CU
  int32_t colc_idx[2];                 // ... their indices on the 8x8 grid (-1: none)
  int32_t out4[4];
  int32_t ref_idx2[2];
  int32_t i0, i1, i2, i3;              // small hand-overs from lane 0 to the wave
  int32_t early_tag;                   // 1 + the merge index whose luma levels the early-skip test left in the CU's coefficient target (0: none)
  double d0, d1;
};
#endif
// LDS image of a workgroup
template <typename PX> struct lds {
#if defined(CTU_PROFILE)
  scratch *prof_w;
#endif
  PX Dy[65 * PY], Du[33 * PC], Dv[33 * PC];         // decided planes, index (y + 1) * pitch + x + 1
  PX cand_px[lds_cfg<PX>::slim ? 480 : 2016];       // a depth's CU while its split is being tried (depths 1..3; slim: 2..3, depth 1 in scratch::cand_px32)
  cu4 cu[17 * 17];                                  // index (y4 + 1) * 17 + x4 + 1
  scratch *scr;                                     // the workgroup's global scratch (the split / mode-type trees per 4x4 live there)
  uint32_t cur[NMX];                                // state->search_cabac models of the walk: state0 | state1 << 16
  uint32_t work[3][NMX];                            // [L - 1], L = 1..3: the models the depth-L candidate starts from (written by the walk when it
                                                    // posts the evaluation) and, adapted in place, leaves behind; [2] doubles as scratch for the 64x64 candidate
  uint32_t coder[NMX];                              // state->cabac models
  uint8_t rdoq_state[244];                          // CTX_STATE of the coder's models at the CTU's start (what uvg_rdoq prices with)
  uint16_t scan[lds_cfg<PX>::slim_scan ? 64 + 16 : 1024 + 256 + 64 + 16];      // coefficient scans of the four square shapes (slim: 8x8 and 4x4, the others in scratch::scan_g)
  uint8_t inv4[16];                                 // scan index of the 4x4 group's raster position y * 4 + x
  uint16_t deps4[16];                               // per scan index of a 4x4 group: the scan indices (bits) its context template reads inside the group
  int32_t last_bits[92];                            // RDOQ: bit cost of the last-position prefix per (luma/chroma, log2 size, x/y, group index): last_bits_off
  level_state lvl[5];
  wctx wv[4];                                       // [0] depth 4 (4x4), [1] depth 3, [2] depth 2, [3] depth 1 (32x32)
  int32_t vsel[4];                                  // which wv[] a wave is using (a wave may borrow a larger one while its owner idles)
  int32_t rot;                                      // index of the wave with role 0 (CTU_WAVE)
  int32_t req[4], done[4];                          // depth pipeline: evaluation requests / completions per depth
  int32_t hreq, hdone, help[6];                     // the chroma helper (help_post): requests / completions; area x, y, mode -> has_coeffs, SSD
#if defined(CTU_LEAF4)
  // ctu_leaf4.h: the cubic interpolation filter (4 x int8 per phase), intraPredAngle | invAngle << 8 per |mode distance|, per raster
  // position of a 4x4 block (positions later in scan order | scan index << 16 | raster of the next scan index << 20), the 8x8 area
  // whose source samples are parked in lf_src ([0, 64) luma, [64, 80) Cb, [80, 96) Cr)
  uint32_t lf_cubic[32], lf_disp[17], lf_rq[16];
  double lf_escale[2];                              // uvg_rdoq's error scale of a 4x4 block: luma, chroma
  int32_t lf_qbits[2], lf_q[2];                     // ... q_bits, quantiser scale
  int32_t lf_tag;
  PX lf_src[96];
#endif
#if !defined(CTU_PB)
  // the 64x64 candidate (post64 / finish64): posted once the first 32x32 area is decided, its luma blocks taken by depth 1's wave and
  // its chroma blocks by depth 2's between their evaluations; the modes (luma, chroma) of the request; per 32x32 transform unit the
  // flags, the SSDs and the bit counts (bits: chroma, bits_y: luma)
  int32_t req64, done64[2], m64[2];
  int32_t creq, cdone[2];                           // the coder's pass by model (run_ctu): asked for; the flags / the chroma coefficients are done
  struct { int32_t cu, cv, ssd_u, ssd_v, cy, ssd_y; double bits, bits_y; } h64[4];
#endif
  alignas(16) unsigned char arena[lds_cfg<PX>::slim ? (int)ARENA_BYTES_SLIM : (int)ARENA_BYTES];
#if defined(CTU_PB)
  pb_state pb;
  // the leaf wave of the two-wave build (ctu_pb.h post_leaves): on / off, the costs of the four 4x4 CUs of the area it was handed
  int32_t leaf_wave;
  double leaf_cost[4];
  double leaf_limit;                   // the leaf wave stops when the costs of its 4x4 CUs exceed it (the walk lowers it when the 8x8 CU's cost is in);
                                       // (in front of pbx: the two-wave launch asks for the image up to there, pb_lds_bytes)
  // the three-wave build (ctu_pb.h): the wave that evaluates the 32x32 / 16x16 CUs ahead of the walk has its own search state (of `pb`
  // it reads the shared tables only: mot, fl, hmvp_entry) and the 8x8 depth its own scratch; the one- and two-wave launches ask for
  // the image up to here (pb_lds_bytes)
  int32_t depth_wave, skip_eval[4];
  pb_state pbx;
  alignas(16) unsigned char arena8[arena_bytes(8)];
  // ... and the 8x8 CU's batched merge analysis (ctu_pb.h merge_batch8) its own scratch: [candidate][list][15][8] intermediates, the
  // predictions [candidate][64], the SATDs (the one-wave build borrows the 32x32 depth's transform buffers for them)
  alignas(16) int16_t b8_tmp[6 * 2 * 120];
  PX b8_pred[6 * 64];
  int32_t b8_satd[8];
  // the four-wave build: 32x32 CUs on a wave of their own (search state and, for the 16x16 depth, scratch)
  pb_state pbx2;
  alignas(16) unsigned char arena16[arena_bytes(16)];
  // ... and the 64x64 CU evaluated there as well, beside the walk: its samples and levels (64x64 + 2 x 32x32) while the split is tried,
  // the walk's models as they were after the CU's split flag (what a pruned CTU goes on with)
  alignas(16) PX cand64_px[6144];
  int16_t cand64_co[6144];
  uint32_t cur64[NMX];
#endif
};
#if defined(CTU_PB)
template <typename PX> constexpr size_t pb_lds_bytes(int waves) { return waves >= 4 ? sizeof(lds<PX>) : (waves == 3 ? offsetof(lds<PX>, pbx2) : offsetof(lds<PX>, pbx)); }
// the search state of the wave that asks (roles 2, 3: the depth waves)
template <typename PX> CTU_DEV pb_state &pbq(lds<PX> *S) { const int r = CTU_WAVE; return r == 2 ? S->pbx : (r == 3 ? S->pbx2 : S->pb); }
#endif
template <typename PX> CTU_DEV wctx *wv_of(lds<PX> *S) { return &S->wv[S->vsel[CTU_WAVE]]; }

// per-workgroup scratch in global memory
struct scratch {
  double cost_coeff[1024], cost_sig[1024], cost_coeff0[1024];
  uint16_t save_px[6144];          // the 64x64 candidate's samples and levels beside the split's (I pictures: post64; P / B one-wave build: the
  int16_t save_co[6144];           // whole D of a CTU while the candidate is tried in place)
  cu4 save_cu[256];
  int16_t cand_co[2016];           // levels of the candidate CUs of depths 1..3 (cand_px_off)
  // a slim build (lds_cfg): the depth-1 candidate's samples, the levels of the depth-1 scratch (y, u, v), the scans of 32x32 and 16x16
  uint16_t cand_px32[1536];
  int16_t lv32[1536];
  uint16_t scan_g[1024 + 256];
  uint32_t save_tree[512];
  uint16_t tree[256], mtt[256];    // split_tree / mode_type_tree per 4x4 of the decided CTU (3 / 2 bits per depth, depths 0..4)
#if defined(CTU_PB)
  int32_t save_mot[256][8];        // the 64x64 candidate of a P / B picture while its split is tried: motion, flags
  uint8_t save_fl[256][8];
  int32_t pb_mot[17 * 17 + 1][8];  // pb_state::mot / fl (icand::unit is eight int32)
  uint8_t pb_fl[17 * 17][8];
#endif
#if defined(CTU_PB)
  unsigned long long prof_pb[24];     // CTU_PROFILE, ctu_pb.h: cycles of the phases of the P / B walk (lane 0 of the wave); 16.. : waits of the walk, the other waves' busy time
#endif
  unsigned long long prof_lf[16];     // CTU_PROFILE, ctu_leaf4.h: cycles of the 4x4 leaf's steps (the walk's wave)
  unsigned long long prof[4][32];     // CTU_PROFILE: 0 rough search, 1 refs + prediction, 2 residual + transforms + reconstruction, 3 RDOQ, 4 SSD,
                                   // 5 RD cost (bits), 6 park / unpark / model copies, 7 64x64 candidate, 8 coder pass, 9 load, 10 store, 11 total
};

// everything one workgroup needs to know about its CTU
template <typename PX> struct job {
  params P;
  const PX *src_y, *src_u, *src_v;       // source planes
  int src_stride, src_stride_c;
  PX *rec_y, *rec_u, *rec_v;             // reconstruction before the in-loop filters (also the neighbours' samples)
  int rec_stride, rec_stride_c;
  uvghip_scu_t *cu_tab;                  // the picture's side information, one entry per 4x4
  int cu_stride;
  int16_t *coeff;                        // this CTU's lcu_coeff_t: y[64*64], u[32*32], v[32*32]
  uint32_t *models_out;                  // this CTU's three model sets [3][NMODELS]: at its start, after its search, after the coder
  const uint32_t *models_in;             // the coder's models this CTU starts from (NULL: initialise for an I slice at P.qp)
  scratch *W;
  int x, y;                              // CTU origin
#if defined(CTU_PB)
  const struct pb_job *pb;               // the picture's inter state (ctu_pb.h)
  uint32_t *pbm_out;                     // this CTU's three sets of the 18 inter-syntax models
  const uint32_t *pbm_in;                // ... of the CTU it starts from (NULL with models_in)
  int slice_type, init_qp;               // what the slice's models are initialised with (0 B, 1 P; state->frame->QP)
#endif
};

// the source samples of a block of `color` at CTU-local (bx, by) (in that plane's samples), read from the picture; -> pointer, pitch
// (blocks never reach outside the picture: a CU is only coded when it lies inside)
template <typename PX> CTU_DEV CTU_GLB const PX *src_block(const job<PX> &J, int color, int bx, int by, int *pitch)
{
  const PX *p = color == 0 ? J.src_y : (color == 1 ? J.src_u : J.src_v);
  const int st = color == 0 ? J.src_stride : J.src_stride_c, sh = color != 0;
  *pitch = st;
  return (CTU_GLB const PX *)(p + (size_t)((J.y >> sh) + by) * st + (J.x >> sh) + bx);
}


// ------------------------------------------------------------------------------------------------------------ models ------
// Tables every serial walk reads per bin live in LDS (a lone lane pays the full latency of every load: out of device memory the
// entropy table alone cost most of the search's time): the 512 bit costs, the models' window bytes, and the bit costs of the
// models uvg_rdoq prices with (fixed for the CTU).  Function-scope LDS, filled by load_ctu.
#if defined(__HIPCC__)
#define CTU_SHARED __shared__
#else
#define CTU_SHARED static
#endif
CTU_DEV uint32_t *tab_ebits() { CTU_SHARED uint32_t t[512]; return t; }
CTU_DEV uint8_t *tab_rate() { CTU_SHARED uint8_t t[(NMX + 7) & ~7]; return t; }
CTU_DEV uint32_t *tab_rdoq_bits() { CTU_SHARED uint32_t t[2 * 244]; return t; }
#define kRate (tab_rate())         // the window byte of every model (rate0 << 4 | rate1)

template <typename MP> CTU_DEV int m_state(MP m, int c) { return (int)(((m[c] & 0xffffu) + (m[c] >> 16)) >> 8); }
template <typename MP> CTU_DEV void m_update(MP m, int c, int bin)            // CTX_UPDATE, cabac.h:182-193
{
  const int r0 = kRate[c] >> 4, r1 = kRate[c] & 15;
  uint32_t s0 = m[c] & 0xffffu, s1 = m[c] >> 16;
  s0 -= (s0 >> r0) & 0x7fe0u;
  s1 -= (s1 >> r1) & 0x7ffeu;
  if (bin) { s0 += (0x7fffu >> r0) & 0x7fe0u; s1 += (0x7fffu >> r1) & 0x7ffeu; }
  m[c] = (s0 & 0xffffu) | (s1 << 16);
}
// uvg_f_entropy_bits (rdo.c:143, a float table) = uvg_entropy_bits / 2^15: every entry is an integer below 2^24 over 2^15, so the
// float and this double quotient are the same number
template <typename MP> CTU_DEV double m_fbits(MP m, int c, int bin) { return (double)tab_ebits()[(m_state(m, c) << 1) ^ bin] / 32768.0; }
// CABAC_FBITS_UPDATE with only_count = 1
template <typename MP> CTU_DEV void m_code(MP m, int update, int c, int bin, double &bits)
{
  bits += m_fbits(m, c, bin);
  if (update) m_update(m, c, bin);
}

CTU_DEV void models_init_one(uint32_t *m, int i, int qp, int slice)     // uvg_ctx_init, context.c:471-492
{
  const int v = k_ctx_init[slice][i];
  if (v == 255) { m[i] = 0; return; }
  const int slope = (v >> 3) - 4, offset = ((v & 7) * 18) + 1;
  int s = ((slope * (qp - 16)) >> 1) + offset;
  s = s < 1 ? 1 : (s > 127 ? 127 : s);
  const uint32_t p1 = (uint32_t)s << 8;
  m[i] = (p1 & 0x7fe0u) | ((p1 & 0x7ffeu) << 16);
}

// ------------------------------------------------------------------------------------------------------------- scans ------
CTU_DEV int scan_base(int log2n) { return log2n == 5 ? 0 : log2n == 4 ? 1024 : log2n == 3 ? 1280 : 1344; }
// the scan of the 1 << log2n square: the LDS table, or (slim build, 16x16 and 32x32) the workgroup's global copy
template <typename PX> CTU_DEV uint16_t *scan_of(lds<PX> *S, int log2n)
{
  if (lds_cfg<PX>::slim_scan) return log2n >= 4 ? S->scr->scan_g + (log2n == 5 ? 0 : 1024) : S->scan + (log2n == 3 ? 0 : 64);
  return S->scan + scan_base(log2n);
}
// last_bits: [luma 4, 8, 16, 32; chroma 4, 8, 16][x, y][group index 0 .. group_idx(n - 1)]
CTU_DEV int last_bits_off(int t, int log2n, int xy)
{
  const int cnt = 2 * log2n;                                       // 4, 6, 8, 10 entries
  const int base = log2n == 2 ? 0 : (log2n == 3 ? 8 : (log2n == 4 ? 20 : 36));
  return (t ? 56 : 0) + base + (xy ? cnt : 0);
}
// H.266 6.5.2 up-right diagonal scan of 4x4 groups, groups in diagonal order (tables.c g_scan_order, SCAN_DIAG)
CTU_DEV void diag_order(int n, uint8_t *out)      // out[i] = y * n + x of the i-th position of an n x n diagonal scan
{
  int i = 0;
  for (int d = 0; d < 2 * n - 1; ++d)
    for (int x = 0; x <= d; ++x) {
      const int y = d - x;
      if (x < n && y < n) out[i++] = (uint8_t)(y * n + x);
    }
}
template <typename PX> CTU_DEV void build_scans(lds<PX> *S)
{
  SERIAL {
    uint8_t in[16], cg[64];
    diag_order(4, in);
    for (int l2 = 2; l2 <= 5; ++l2) {
      const int n = 1 << l2, cgw = n >> 2;
      diag_order(cgw, cg);
      uint16_t *sc = scan_of(S, l2);
      if (l2 == 2) {
        for (int k = 0; k < 16; ++k) S->inv4[in[k]] = (uint8_t)k;
        for (int k = 0; k < 16; ++k) {
          const int x = in[k] & 3, y = in[k] >> 2;
          unsigned m = 0;
          if (x + 1 < 4) m |= 1u << S->inv4[y * 4 + x + 1];
          if (x + 2 < 4) m |= 1u << S->inv4[y * 4 + x + 2];
          if (y + 1 < 4) m |= 1u << S->inv4[(y + 1) * 4 + x];
          if (y + 2 < 4) m |= 1u << S->inv4[(y + 2) * 4 + x];
          if (x + 1 < 4 && y + 1 < 4) m |= 1u << S->inv4[(y + 1) * 4 + x + 1];
          S->deps4[k] = (uint16_t)m;
        }
      }
      for (int g = 0; g < cgw * cgw; ++g) {
        const int gx = cg[g] % cgw, gy = cg[g] / cgw;
        for (int k = 0; k < 16; ++k) sc[g * 16 + k] = (uint16_t)((gy * 4 + in[k] / 4) * n + gx * 4 + in[k] % 4);
      }
    }
  }
  CTU_SYNC();
}

// ------------------------------------------------------------------------------------------- reference construction ------
CTU_DEV int16_t *lv_of(wctx *V, int color) { return color == 0 ? V->lv0 : (color == 1 ? V->lv1 : V->lv2); }
template <typename PX> CTU_DEV PX *plane(lds<PX> *S, int color) { return color == 0 ? S->Dy : (color == 1 ? S->Du : S->Dv); }
CTU_DEV int pitch_of(int color) { return color == 0 ? PY : PC; }
template <typename PX> CTU_DEV cu4 *cu_at(lds<PX> *S, int lx, int ly) { return &S->cu[((ly >> 2) + 1) * 17 + (lx >> 2) + 1]; }   // lx, ly >= -4

// uvg_count_available_edge_cus (cu.c:516-537) for a square CU at CTU-local (lx, ly)
template <typename PX> CTU_DEV int count_edge_cus(lds<PX> *S, int x, int y, int lx, int ly, int n, int left)
{
  if ((left && x == 0) || (!left && y == 0)) return 0;
  if (left && lx == 0) return (LCU - ly) / 4;
  if (!left && ly == 0) return n / 2;
  int amount = n & ~3;
  if (left) {
    if (ly == 0 && lx == 32 && cu_at(S, lx, ly)->log2 == 6) return 8;
    while (ly + amount < LCU && cu_at(S, lx - 4, ly + amount)->type != CU_NOTSET) amount += 4;
  } else {
    while (lx + amount < LCU && cu_at(S, lx + amount, ly - 4)->type != CU_NOTSET) amount += 4;
  }
  return (amount / 4) > (n / 4) ? amount / 4 : n / 4;
}

// uvg_intra_build_reference (intra.c:1344; _inner :1065-1341 when the block touches neither picture edge, _any :756-1063 else) for the
// w x w block of `color` whose luma position is (x, y) / CTU-local (lx, ly) with luma size n; + the smoothed rows (:190-225)
// cu_n: the size of the CU the block belongs to.  The smoothing filter covers 2 * (the CU's size) entries, not 2 * (the block's)
// (intra.c:715-725): in a 32x32 transform block of a 64x64 CU entry 2N is smoothed too, against the padding behind it.
template <typename PX> CTU_NOINLINE CTU_DEV void build_refs(lds<PX> *S, const params &P, int color, int x, int y, int lx, int ly, int n, int cu_n)
{
  wctx *const V = wv_of(S);
  const int c = color != 0;
  const int w = n >> c;
  const int px_x = lx >> c, px_y = ly >> c;
  const int pit = pitch_of(color);
  const PX *D = plane(S, color) + (px_y + 1) * pit + px_x + 1;     // the block's top-left sample
  SERIAL {
    int al = count_edge_cus(S, x, y, lx, ly, n, 1) * (c ? 2 : 4);
    if (al > 2 * w) al = 2 * w;
    if (al > ((P.pic_h - y) >> c)) al = (P.pic_h - y) >> c;
    int at = count_edge_cus(S, x, y, lx, ly, n, 0) * (c ? 2 : 4);
    if (at > 2 * w) at = 2 * w;
    if (at > ((P.pic_w - x) >> c)) at = (P.pic_w - x) >> c;
    if (x > 0 && y > 0 && P.wpp && px_y == 0 && at > (LCU >> c) - px_x) at = (LCU >> c) - px_x;
    V->u_avail_left = al; V->u_avail_top = at;
  }
  CTU_SYNC();
  const int al = V->u_avail_left, at = V->u_avail_top;
  const int dc = 1 << (px_info<PX>::depth - 1);
  CTU_LDS uint16_t *const r_top = LDSP(uint16_t, V->top), *const r_left = LDSP(uint16_t, V->left);
  CTU_LDS uint16_t *const r_ftop = LDSP(uint16_t, V->ftop), *const r_fleft = LDSP(uint16_t, V->fleft);
  PAR_FOR(i, V->refn - 1) {
    int lv, tv;
    if (x > 0) lv = D[(i < al ? i : al - 1) * pit - 1];
    else lv = y > 0 ? D[-pit] : dc;
    if (y > 0) tv = D[-pit + (i < at ? i : at - 1)];
    else tv = x > 0 ? D[-1] : dc;
    r_left[i + 1] = (uint16_t)lv;
    r_top[i + 1] = (uint16_t)tv;
  }
  SERIAL {
    int corner;
    if (x > 0 && y > 0) corner = D[-pit - 1];
    else corner = x > 0 ? D[-1] : (y > 0 ? D[-pit] : dc);       // "copy reference clockwise": left[1]
    r_left[0] = r_top[0] = (uint16_t)corner;
  }
  CTU_SYNC();
  const int flim = 2 * (cu_n >> c) < V->refn - 1 ? 2 * (cu_n >> c) : V->refn - 1;
  PAR_FOR(i, V->refn) {
    int fl, ft;
    if (i == 0) fl = ft = (r_left[1] + 2 * r_left[0] + r_top[1] + 2) >> 2;
    else {
      fl = i < flim ? (r_left[i - 1] + 2 * r_left[i] + r_left[i + 1] + 2) >> 2 : r_left[i];
      ft = i < flim ? (r_top[i - 1] + 2 * r_top[i] + r_top[i + 1] + 2) >> 2 : r_top[i];
    }
    r_fleft[i] = (uint16_t)fl;
    r_ftop[i] = (uint16_t)ft;
  }
  CTU_SYNC();
}

// prediction of the w x w block of `color` from the reference rows into dst (pitch dp)
template <typename PX> CTU_NOINLINE CTU_DEV void predict_block(lds<PX> *S, int mode, int color, int w, PX *dst_, int dp)
{
  wctx *const V = wv_of(S);
  const mode_info M = make_mode_info(mode, w, w, color != 0);
  const ref_rows_t<CTU_LDS const uint16_t *> R = {LDSP(const uint16_t, V->top), LDSP(const uint16_t, V->left), LDSP(const uint16_t, V->ftop), LDSP(const uint16_t, V->fleft)};
  const int dc = mode == 1 ? dc_value(R.top, R.left, w, w) : 0;
  typename mg_ptr<PX, PX>::type const dst = MGP(PX, PX, dst_);
  const int segs = w >> 2;
  PAR_FOR(t, w * segs) {
    const int yd = t / segs, xd0 = (t - yd * segs) * 4;
    int out[4];
    predict_row<4>(M, R, dc, color != 0, w, w, yd, xd0, (int)px_info<PX>::maxv, out);
    for (int i = 0; i < 4; ++i) {
      const int bx = M.vertical ? xd0 + i : yd, by = M.vertical ? yd : xd0 + i;
      dst[by * dp + bx] = (PX)out[i];
    }
  }
  CTU_SYNC();
}

// --------------------------------------------------------------------------------------------------- rough search ------
CTU_DEV int iabs_(int v) { return v < 0 ? -v : v; }
// 8x8 / 4x4 Hadamard SATD of a difference tile (picture-generic.c:118-200, 256-348); the magnitude multiset is transpose-invariant
CTU_DEV unsigned satd8_tile(int (&d)[64])
{
  for (int pass = 0; pass < 2; ++pass) {
    const int s = pass ? 8 : 1, t = pass ? 1 : 8;      // pass 0: along rows, pass 1: along columns
    for (int l = 0; l < 8; ++l) {
      int *v = &d[l * t];
      for (int half = 4; half >= 1; half >>= 1)
        for (int base = 0; base < 8; base += 2 * half)
          for (int i = 0; i < half; ++i) {
            const int p = v[(base + i) * s], q = v[(base + i + half) * s];
            v[(base + i) * s] = p + q;
            v[(base + i + half) * s] = p - q;
          }
    }
  }
  unsigned sad = 0;
  for (int i = 0; i < 64; ++i) sad += (unsigned)iabs_(d[i]);
  sad -= (unsigned)iabs_(d[0]);
  sad += (unsigned)iabs_(d[0]) >> 2;
  return (sad + 2) >> 2;
}
CTU_DEV unsigned satd4_tile(int (&d)[16])
{
  for (int pass = 0; pass < 2; ++pass) {
    const int s = pass ? 4 : 1, t = pass ? 1 : 4;
    for (int l = 0; l < 4; ++l) {
      int *v = &d[l * t];
      const int a = v[0] + v[2 * s], b = v[0] - v[2 * s], c2 = v[s] + v[3 * s], e = v[s] - v[3 * s];
      v[0] = a + c2; v[s] = a - c2; v[2 * s] = b + e; v[3 * s] = b - e;
    }
  }
  unsigned satd = 0;
  for (int i = 0; i < 16; ++i) satd += (unsigned)iabs_(d[i]);
  satd -= (unsigned)iabs_(d[0]);
  satd += (unsigned)iabs_(d[0]) >> 2;
  return (satd + 1) >> 1;
}

// costs of the listed modes for the n x n luma block at (lx, ly); get_cost_dual (search_intra.c:133-192).
// Device: every (mode, tile, row) is one lane -- a lane predicts one row of the tile in registers (in the mode's work domain:
// transposed for the horizontal modes; SAD and the Hadamard magnitude multiset are transpose-invariant), the T lanes of a tile
// finish the Hadamard with DPP exchanges (satd_dev.h).  Host emulation: the tile as a whole.
template <typename PX> CTU_NOINLINE CTU_DEV void rough_costs(lds<PX> *S, const job<PX> &J, int lx, int ly, int n, const int32_t *modes, int n_modes)
{
  wctx *const V = wv_of(S);
  int sps;
  CTU_GLB const PX *Sy = src_block(J, 0, lx, ly, &sps);
  const int T = n >= 8 ? 8 : 4, tiles_x = n / T, tiles = tiles_x * tiles_x;
  const ref_rows_t<CTU_LDS const uint16_t *> R = {LDSP(const uint16_t, V->top), LDSP(const uint16_t, V->left), LDSP(const uint16_t, V->ftop), LDSP(const uint16_t, V->fleft)};
  const int dcv = dc_value(R.top, R.left, n, n);
  CTU_LDS uint32_t *const part = LDSP(uint32_t, V->part);
#if defined(__HIPCC__)
  // Two segments: the angular modes of the list, then planar / DC (the first round's list starts with them).  A pass whose lanes
  // are all of one kind skips the other kind's code (no lane enters it); mixed passes would run both for everybody -- for a 4x4
  // block 16 angular modes x 4 rows fill one pass exactly and planar + DC become a short second one.
  const int n_flat = (n_modes > 0 && modes[0] < 2) + (n_modes > 1 && modes[1] < 2);
  for (int seg = 0; seg < 2; ++seg) {
  const int m_first = seg ? 0 : n_flat, total = (seg ? n_flat : n_modes - n_flat) * tiles * T;
  for (int base = 0; base < total; base += CTU_NT) {
    const int idx = base + CTU_TID;
    const bool on = idx < total;
    const int task0 = on ? idx / T : 0, r = idx & (T - 1);
    const int task = task0 + m_first * tiles;
    const int mi = task / tiles, tile = task - mi * tiles;
    const mode_info M = make_mode_info(modes[mi], n, n, 0);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    int sad = 0, satd;
    if (T == 8) {
      int d[8], out[8];
      predict_row<8>(M, R, dcv, 0, n, n, ty * 8 + r, tx * 8, (int)px_info<PX>::maxv, out);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int wx = tx * 8 + i, wy = ty * 8 + r;
        const int bx = M.vertical ? wx : wy, by = M.vertical ? wy : wx;
        d[i] = on ? (int)Sy[by * sps + bx] - out[i] : 0;
        sad += iabs_(d[i]);
      }
      satd = satd8_cost(d, r);
      sad = dpp_group_sum<8>(sad);
    } else {
      int d[4], out[4];
      predict_row<4>(M, R, dcv, 0, n, n, r, 0, (int)px_info<PX>::maxv, out);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int bx = M.vertical ? i : r, by = M.vertical ? r : i;
        d[i] = on ? (int)Sy[by * sps + bx] - out[i] : 0;
        sad += iabs_(d[i]);
      }
      satd = satd4_cost(d, r);
      sad = dpp_group_sum<4>(sad);
    }
    if (on && r == 0) { part[2 * task] = (uint32_t)satd; part[2 * task + 1] = (uint32_t)sad; }
  }
  }
  CTU_SYNC();
#else
  PAR_FOR(task, n_modes * tiles) {
    const int mi = task / tiles, tile = task - mi * tiles;
    const int mode = modes[mi];
    const mode_info M = make_mode_info(mode, n, n, 0);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    unsigned sad = 0, satd;
    if (T == 8) {
      int d[64];
      for (int r = 0; r < 8; ++r) {
        int out[8];
        predict_row<8>(M, R, dcv, 0, n, n, ty * 8 + r, tx * 8, (int)px_info<PX>::maxv, out);
        for (int i = 0; i < 8; ++i) {
          const int wx = tx * 8 + i, wy = ty * 8 + r;
          const int bx = M.vertical ? wx : wy, by = M.vertical ? wy : wx;
          const int df = (int)Sy[by * sps + bx] - out[i];
          d[r * 8 + i] = df;
          sad += (unsigned)iabs_(df);
        }
      }
      satd = satd8_tile(d);
    } else {
      int d[16];
      for (int r = 0; r < 4; ++r) {
        int out[4];
        predict_row<4>(M, R, dcv, 0, n, n, r, 0, (int)px_info<PX>::maxv, out);
        for (int i = 0; i < 4; ++i) {
          const int bx = M.vertical ? i : r, by = M.vertical ? r : i;
          const int df = (int)Sy[by * sps + bx] - out[i];
          d[r * 4 + i] = df;
          sad += (unsigned)iabs_(df);
        }
      }
      satd = satd4_tile(d);
    }
    part[2 * task] = satd;
    part[2 * task + 1] = sad;
  }
  CTU_SYNC();
#endif
}

// count_bits (search_intra.c:949-984)
CTU_DEV double count_bits(const int8_t *preds, double planar, double not_planar, double mpm_bit, double not_mpm_bit, int mode)
{
  int i = 0, smaller = 0;
  for (; i < 6; i++) {
    if (preds[i] == mode) break;
    if (mode > preds[i]) smaller += 1;
  }
  if (i == 0) return planar + mpm_bit;
  if (i < 6) return not_planar + mpm_bit + (i < 4 ? i : 4);
  return not_mpm_bit + 5 + (mode - smaller > 2);
}

// uvg_intra_get_dir_luma_predictor (intra.c:88-188), MIP off
CTU_DEV void dir_luma_predictor(int y, int8_t *preds, const cu4 *left_pu, const cu4 *above_pu)
{
  int left_dir = 0, above_dir = 0;
  if (left_pu && left_pu->type == CU_INTRA) left_dir = left_pu->mode;
  if (above_pu && above_pu->type == CU_INTRA && y % LCU != 0) above_dir = above_pu->mode;
  const int offset = 61, mod = 64;
  preds[0] = 0; preds[1] = 1; preds[2] = 50; preds[3] = 18; preds[4] = 46; preds[5] = 54;
  if (left_dir == above_dir) {
    if (left_dir > 1) {
      preds[0] = 0; preds[1] = (int8_t)left_dir;
      preds[2] = (int8_t)(((left_dir + offset) % mod) + 2); preds[3] = (int8_t)(((left_dir - 1) % mod) + 2);
      preds[4] = (int8_t)(((left_dir + offset - 1) % mod) + 2); preds[5] = (int8_t)((left_dir % mod) + 2);
    }
  } else if (left_dir > 1 && above_dir > 1) {
    preds[0] = 0; preds[1] = (int8_t)left_dir; preds[2] = (int8_t)above_dir;
    const int mx = preds[1] > preds[2] ? 1 : 2, mn = preds[1] > preds[2] ? 2 : 1;
    const int diff = preds[mx] - preds[mn];
    if (diff == 1) {
      preds[3] = (int8_t)(((preds[mn] + offset) % mod) + 2); preds[4] = (int8_t)(((preds[mx] - 1) % mod) + 2);
      preds[5] = (int8_t)(((preds[mn] + offset - 1) % mod) + 2);
    } else if (diff >= 62) {
      preds[3] = (int8_t)(((preds[mn] - 1) % mod) + 2); preds[4] = (int8_t)(((preds[mx] + offset) % mod) + 2);
      preds[5] = (int8_t)((preds[mn] % mod) + 2);
    } else if (diff == 2) {
      preds[3] = (int8_t)(((preds[mn] - 1) % mod) + 2); preds[4] = (int8_t)(((preds[mn] + offset) % mod) + 2);
      preds[5] = (int8_t)(((preds[mx] - 1) % mod) + 2);
    } else {
      preds[3] = (int8_t)(((preds[mn] + offset) % mod) + 2); preds[4] = (int8_t)(((preds[mn] - 1) % mod) + 2);
      preds[5] = (int8_t)(((preds[mx] + offset) % mod) + 2);
    }
  } else if (left_dir + above_dir >= 2) {
    preds[0] = 0;
    preds[1] = (int8_t)(left_dir < above_dir ? above_dir : left_dir);
    preds[2] = (int8_t)(((preds[1] + offset) % mod) + 2); preds[3] = (int8_t)(((preds[1] - 1) % mod) + 2);
    preds[4] = (int8_t)(((preds[1] + offset - 1) % mod) + 2); preds[5] = (int8_t)((preds[1] % mod) + 2);
  }
}
// the two neighbours uvg_search_cu_intra (search_intra.c:1792-1803) and uvg_encode_intra_luma_coding_unit (:1114-1150) look at
template <typename PX> CTU_DEV void mpm_neighbours(lds<PX> *S, int x, int y, int lx, int ly, int n, const cu4 **left, const cu4 **above)
{
  *left = x > 0 ? cu_at(S, lx - 1, ly + n - 1) : nullptr;
  *above = (ly > 0 && y > 0) ? cu_at(S, lx + n - 1, ly - 1) : nullptr;
}

// search_intra_rough (search_intra.c:986-1229), three survivors; the winner goes to V->u_mode
template <typename PX> CTU_INLINE1 CTU_DEV void search_intra_rough(lds<PX> *S, const job<PX> &J, int x, int y, int lx, int ly, int n)
{
  wctx *const V = wv_of(S);
  const params &P = J.P;
  const int T = n >= 8 ? 8 : 4, tiles = (n / T) * (n / T);
#if defined(__HIPCC__) && defined(CTU_PROFILE)
  scratch *const W = J.W;
  unsigned long long tq = __builtin_amdgcn_s_memtime();
#endif
  SERIAL {
    const cu4 *l, *a;
    mpm_neighbours(S, x, y, lx, ly, n, &l, &a);
    dir_luma_predictor(y, V->mpm, l, a);
    const int offset = 1 << P.rough_levels;
    int k = 0;
    uint32_t h0 = 3, h1 = 0, h2 = 0;            // modes costed so far
    V->rs_list[k++] = 0; V->rs_list[k++] = 1;
    for (int mode = 2 + offset / 2; mode <= 66; mode += 2 * offset)
      for (int i = 0; i < 2; ++i) {
        const int m = mode + i * offset;
        if (m > 66) continue;
        V->rs_list[k++] = m;
        const uint32_t bit = 1u << (m & 31);
        if (m < 32) h0 |= bit; else if (m < 64) h1 |= bit; else h2 |= bit;
      }
    V->rs_chk[0] = h0; V->rs_chk[1] = h1; V->rs_chk[2] = h2;
    V->u_n_modes = k;
  }
  CTU_SYNC();
  RQ_T(24);
  // (rs_list holds at most 18 entries with rough_levels >= 2; the host refuses smaller values)
  // The reference inserts every costed mode into a three-entry list with a strict "<" (search_intra.c:1071-1143), i.e. the list is
  // the three smallest of everything costed so far under the order (cost, insertion sequence); DC is inserted ahead of planar when
  // the two tie (:1089-1106).  Here a lane per candidate counts the candidates ahead of it -- no loop-carried list.
  int offset = 1 << P.rough_levels;
  for (int round = 0;; ++round) {
    rough_costs(S, J, lx, ly, n, V->rs_list, V->u_n_modes);
    RQ_T(25);
    const int nm = V->u_n_modes;
    CTU_LDS const uint32_t *const mdl = LDSP(const uint32_t, V->cur), *const part = LDSP(const uint32_t, V->part);
    PAR_FOR(mi, nm) {                           // a lane per mode: tile sums, bit cost (count_bits reads the four flag costs itself)
      const double mpm_bit = m_fbits(mdl, M_MPM, 1), not_mpm_bit = m_fbits(mdl, M_MPM, 0);
      const double planar = m_fbits(mdl, M_PLANAR + 1, 0), not_planar = m_fbits(mdl, M_PLANAR + 1, 1);
      const int mode = V->rs_list[mi];
      unsigned satd = 0, sad = 0;
      for (int t = 0; t < tiles; ++t) { satd += part[2 * (mi * tiles + t)]; sad += part[2 * (mi * tiles + t) + 1]; }
      if (n >= 8) satd >>= (px_info<PX>::depth - 8);       // satd_NxN shifts, the 4x4 function does not (picture-generic.c:170)
      sad >>= (px_info<PX>::depth - 8);
      double c = (double)(satd < sad * 2 ? satd : sad * 2);
      c += count_bits(V->mpm, planar, not_planar, mpm_bit, not_mpm_bit, mode) * P.lambda_sqrt;
      V->rs_cand[3 + mi] = c;
    }
    CTU_SYNC();
    RQ_T(26);
    // candidates: the survivors so far (sequence 0..2, rounds > 0), then the listed modes (sequence 3 + index; round 0: DC 0, planar 1)
    const int first = round ? 0 : 3;
    const int cur = round & 1;
    PAR_FOR(ci0, nm + 3 - first) {
      const int ci = ci0 + first;
      const double c = V->rs_cand[ci];
      const int seq = (round == 0 && ci < 5) ? 4 - ci : ci;
      int rank = 0, differs = 0;
      const double c_first = V->rs_cand[first];
      for (int cj = first; cj < nm + 3; ++cj) {
        const double o = V->rs_cand[cj];
        const int oseq = (round == 0 && cj < 5) ? 4 - cj : cj;
        rank += (o < c || (o == c && oseq < seq)) ? 1 : 0;
        differs |= o != c_first;
      }
      if (rank < 3) { V->rs_best_mode[cur][rank] = ci < 3 ? V->rs_best_mode[cur ^ 1][ci] : V->rs_list[ci - 3]; V->rs_best_cost[cur][rank] = c; }
      if (round == 0 && ci0 == 0) V->u_flag = differs;          // min_cost != max_cost (:1082-1143): only the first round moves them
    }
    CTU_SYNC();
    SERIAL {
      // next round's list (search_intra.c:1146-1215)
      const int b0 = V->rs_best_mode[cur][0], b1 = V->rs_best_mode[cur][1], b2 = V->rs_best_mode[cur][2];
      const double k0 = V->rs_best_cost[cur][0], k1 = V->rs_best_cost[cur][1], k2 = V->rs_best_cost[cur][2];
      uint32_t h0 = V->rs_chk[0], h1 = V->rs_chk[1], h2 = V->rs_chk[2];
      const int go = (offset >> 1) > 0 && V->u_flag;
      int k = 0;
      if (go) {
        const int off = offset >> 1;
#define CTU_TRY(m_) do { const int m = (m_); if (m >= 2 && m <= 66) { const uint32_t bit = 1u << (m & 31); const uint32_t w = m < 32 ? h0 : (m < 64 ? h1 : h2); \
          if (!(w & bit)) { V->rs_list[k++] = m; if (m < 32) h0 |= bit; else if (m < 64) h1 |= bit; else h2 |= bit; } } } while (0)
        if (b0 >= 3 && b0 <= 65) { CTU_TRY(b0 - off); CTU_TRY(b0 + off); }
        if (b1 >= 3 && b1 <= 65) { CTU_TRY(b1 - off); CTU_TRY(b1 + off); }
        if (b2 >= 3 && b2 <= 65) { CTU_TRY(b2 - off); CTU_TRY(b2 + off); }
#undef CTU_TRY
      }
      V->rs_chk[0] = h0; V->rs_chk[1] = h1; V->rs_chk[2] = h2;
      V->rs_cand[0] = k0; V->rs_cand[1] = k1; V->rs_cand[2] = k2;
      V->u_n_modes = k;
      V->u_d0 = go;
      V->u_mode = b0;
    }
    offset >>= 1;
    CTU_SYNC();
    RQ_T(27);
    if (V->u_d0 == 0) break;
  }
}

// ---------------------------------------------------------------------------------------------------- transforms ------
CTU_DEV const int16_t *dct2_matrix(int n) { return n == 4 ? VVC_DCT2_4 : n == 8 ? VVC_DCT2_8 : n == 16 ? VVC_DCT2_16 : VVC_DCT2_32; }
// dct_NxN (dct-generic.c:396-419, 720-729): dst[j * n + i] = trunc16((sum_k T[j][k] * src[i][k] + add) >> shift), twice
CTU_NOINLINE CTU_DEV void fwd_pass(int n, const int16_t *src_, int16_t *dst_, int shift)
{
  CTU_LDS const int16_t *const src = LDSP(const int16_t, src_);
  CTU_LDS int16_t *const dst = LDSP(int16_t, dst_);
  const int16_t *T = dct2_matrix(n);
  const int add = shift > 0 ? 1 << (shift - 1) : 0;
  PAR_FOR(e, n * n) {
    const int j = e / n, i = e - j * n;
    int acc = 0;
    for (int k = 0; k < n; ++k) acc += (int)T[j * n + k] * src[i * n + k];
    dst[j * n + i] = (int16_t)((acc + add) >> shift);
  }
  CTU_SYNC();
}
// idct_NxN (:422-446, 731-740): dst[i * n + j] = clip16((sum_k src[k * n + i] * T[k][j] + add) >> shift), twice
CTU_NOINLINE CTU_DEV void inv_pass(int n, const int16_t *src_, int16_t *dst_, int shift)
{
  CTU_LDS const int16_t *const src = LDSP(const int16_t, src_);
  CTU_LDS int16_t *const dst = LDSP(int16_t, dst_);
  const int16_t *T = dct2_matrix(n);
  const int add = 1 << (shift - 1);
  PAR_FOR(e, n * n) {
    const int i = e / n, j = e - i * n;
    int acc = 0;
    for (int k = 0; k < n; ++k) acc += (int)src[k * n + i] * T[k * n + j];
    dst[i * n + j] = (int16_t)clampi((acc + add) >> shift, -32768, 32767);
  }
  CTU_SYNC();
}

// ----------------------------------------------------------------------------------------------------------- RDOQ ------
CTU_DEV int group_idx(int pos)
{
  if (pos < 4) return pos;
  int l = 0;
  while ((pos >> (l + 1)) != 0) ++l;
  return 2 * l + ((pos >> (l - 1)) & 1);
}
CTU_DEV int go_rice_par(unsigned s) { return (s >= 7) + (s >= 14) + (s >= 28); }

struct rdoq_env {
  const uint8_t *st;      // CTX_STATE per model
  int t;                  // 0 luma, 1 chroma
  double lambda, error_scale;
  int q_bits, q;
};
CTU_DEV int32_t rbits(const rdoq_env &E, int model, int bin) { return (int32_t)tab_rdoq_bits()[2 * model + bin]; }

// uvg_get_ic_rate (rdo.c:465-581), limited prefix length
CTU_DEV int32_t ic_rate(const rdoq_env &E, uint32_t abs_level, int ctx, int go_rice, uint32_t reg_bins)
{
  int32_t rate = 1 << 15;
  const uint32_t go_rice_zero = 1u << go_rice;
  const int max_log2 = 15, thr = 5;
  const int o_par = M_PAR + (E.t ? 21 : 0) + ctx, o_gt1 = M_GT1 + (E.t ? 21 : 0) + ctx, o_gt2 = M_GT2 + (E.t ? 21 : 0) + ctx;
  if (reg_bins < 4) {
    const uint32_t symbol = (abs_level == 0 ? go_rice_zero : abs_level <= go_rice_zero ? abs_level - 1 : abs_level);
    if (symbol < ((uint32_t)thr << go_rice)) {
      rate += (int32_t)(((symbol >> go_rice) + 1 + go_rice) << 15);
    } else {
      const uint32_t max_prefix = 32 - (thr + max_log2);
      uint32_t prefix = 0;
      const uint32_t suffix = (symbol >> go_rice) - thr;
      while (prefix < max_prefix && (int32_t)suffix > ((2 << prefix) - 2)) prefix++;
      const uint32_t suffix_len = prefix == max_prefix ? (uint32_t)(max_log2 - go_rice) : prefix + 1;
      rate += (int32_t)((thr + prefix + suffix_len + go_rice) << 15);
    }
    return rate;
  }
  if (abs_level >= 4) {
    const int32_t symbol = (int32_t)abs_level - 4;
    if (symbol < (thr << go_rice)) {
      rate += ((symbol >> go_rice) + 1 + go_rice) << 15;
    } else {
      const uint32_t max_prefix = 32 - (thr + max_log2);
      uint32_t prefix = 0;
      const uint32_t suffix = (uint32_t)(symbol >> go_rice) - thr;
      while (prefix < max_prefix && (int32_t)suffix > ((2 << prefix) - 2)) prefix++;
      const uint32_t suffix_len = prefix == max_prefix ? (uint32_t)(max_log2 - go_rice) : prefix + 1;
      rate += (int32_t)((thr + prefix + suffix_len + go_rice) << 15);
    }
    rate += rbits(E, o_par, (abs_level - 2) & 1);
    rate += rbits(E, o_gt1, 1);
    rate += rbits(E, o_gt2, 1);
  } else if (abs_level == 1) {
    rate += rbits(E, o_gt1, 0);
  } else if (abs_level == 2) {
    rate += rbits(E, o_par, 0); rate += rbits(E, o_gt1, 1); rate += rbits(E, o_gt2, 0);
  } else if (abs_level == 3) {
    rate += rbits(E, o_par, 1); rate += rbits(E, o_gt1, 1); rate += rbits(E, o_gt2, 0);
  } else {
    rate = 0;
  }
  return rate;
}

// uvg_get_coded_level (rdo.c:597-640)
CTU_DEV uint32_t coded_level(const rdoq_env &E, double *coded_cost, double coded_cost0, double *coded_cost_sig, int32_t level_double,
                             uint32_t max_abs_level, int ctx_sig, int ctx_set, int go_rice, uint32_t reg_bins, int last)
{
  double cur_cost_sig = 0;
  uint32_t best = 0;
  const int o_sig = M_SIG + (E.t ? 12 : 0) + ctx_sig;
  if (!last && max_abs_level < 3) {
    *coded_cost_sig = E.lambda * rbits(E, o_sig, 0);
    *coded_cost = coded_cost0 + *coded_cost_sig;
    if (max_abs_level == 0) return best;
  } else {
    *coded_cost = 1.7e+308;
  }
  if (!last) cur_cost_sig = E.lambda * rbits(E, o_sig, 1);
  const int32_t min_abs = max_abs_level > 1 ? (int32_t)max_abs_level - 1 : 1;
  for (int32_t a = (int32_t)max_abs_level; a >= min_abs; a--) {
    const double err = (double)(level_double - (a * (1 << E.q_bits)));
    double cur = err * err * E.error_scale + E.lambda * ic_rate(E, (uint32_t)a, ctx_set, go_rice, reg_bins);
    cur += cur_cost_sig;
    if (cur < *coded_cost) { best = (uint32_t)a; *coded_cost = cur; *coded_cost_sig = cur_cost_sig; }
  }
  return best;
}

// context_get_sig_ctx_idx_abs / templateAbsSum on a level array (rdo.c:1400-1438, 846-871), no MTS zero-out
template <typename LP> CTU_DEV int sig_ctx_abs(LP lv, int px, int py, int n, int color, int *diag_out, int *sum_out)
{
  const auto d = lv + px + py * n;
  int num_pos = 0, sum_abs = 0;
#define CTU_UPD(v) { const int a = iabs_((int)(v)); sum_abs += (4 + (a & 1)) < a ? (4 + (a & 1)) : a; num_pos += a ? 1 : 0; }
  if (px < n - 1) {
    CTU_UPD(d[1]);
    if (px < n - 2) CTU_UPD(d[2]);
    if (py < n - 1) CTU_UPD(d[n + 1]);
  }
  if (py < n - 1) {
    CTU_UPD(d[n]);
    if (py < n - 2) CTU_UPD(d[n << 1]);
  }
#undef CTU_UPD
  const int diag = px + py;
  int ofs = (((sum_abs + 1) >> 1) < 3 ? ((sum_abs + 1) >> 1) : 3) + (diag < 2 ? 4 : 0);
  if (color == 0) ofs += diag < 5 ? 4 : 0;
  *diag_out = diag;
  *sum_out = sum_abs - num_pos;
  return ofs;
}
template <typename LP> CTU_DEV unsigned template_abs_sum(LP lv, int base_level, int px, int py, int n)
{
  const auto p = lv + px + py * n;
  int16_t sum = 0;                        // coeff_t accumulator, as in the reference (rdo.c:849)
  if (px < n - 1) {
    sum = (int16_t)(sum + iabs_(p[1]));
    if (px < n - 2) sum = (int16_t)(sum + iabs_(p[2]));
    if (py < n - 1) sum = (int16_t)(sum + iabs_(p[n + 1]));
  }
  if (py < n - 1) {
    sum = (int16_t)(sum + iabs_(p[n]));
    if (py < n - 2) sum = (int16_t)(sum + iabs_(p[n << 1]));
  }
  int v = sum - 5 * base_level;
  v = v < 31 ? v : 31;
  return (unsigned)(v > 0 ? v : 0);
}

__device__ static const int16_t kQuantScales[6] = {26214, 23302, 20560, 18396, 16384, 14564};      // uvg_g_quant_scales[0] (scalinglist.c:91)
__device__ static const int16_t kInvQuantScales[6] = {40, 45, 51, 57, 64, 72};                     // uvg_g_inv_quant_scales[0]
__device__ static const double kPow2[16] = {1.0, 2.0, 4.0, 8.0, 16.0, 32.0, 64.0, 128.0, 256.0, 512.0, 1024.0, 2048.0, 4096.0, 8192.0, 16384.0, 32768.0};

// uvg_rdoq (rdo.c:1449-1870) of an n x n block (square: no sqrt2 scaling), intra, no LFNST / MTS / sign hiding; lane 0 only.
// coef -> levels in dst (both n * n, raster).  Returns whether any level survived.
CTU_NOINLINE CTU_DEV int rdoq_serial(const uint8_t *st, const uint16_t *scan, scratch *W, const int16_t *coef, int16_t *dst, int n, int color, int cbf_u,
                        int qp_scaled, double lambda, int bitdepth)
{
  const int l2 = ilog2_dev(n);
  rdoq_env E;
  E.st = st; E.t = color ? 1 : 0; E.lambda = lambda;
  const int transform_shift = 15 - bitdepth - l2;
  uint32_t reg_bins = (uint32_t)(n * n * 28) >> 4;
  E.q_bits = 14 + qp_scaled / 6 + transform_shift;
  const int32_t q = kQuantScales[qp_scaled % 6];
  // scale = 32768 * 2^(-2 * transform_shift) (rdo.c:1527: pow() of an integer exponent, exact), error_scale = scale / q / q
  double scale = 32768;
  scale = transform_shift >= 0 ? scale / kPow2[2 * transform_shift] : scale * kPow2[-2 * transform_shift];
  E.error_scale = scale / q / q;
  double *cost_coeff = W->cost_coeff, *cost_sig = W->cost_sig, *cost_coeff0 = W->cost_coeff0;
  double block_uncoded_cost = 0, base_cost = 0;
  const int cg_num = (n * n) >> 4, cgw = n >> 2;
  double cost_cg_sig[64];
  uint8_t cg_flag[64];
  uint8_t lv_spend[1024];                           // coefficient bit cost: regular bins a scan position spends
  for (int i = 0; i < cg_num; ++i) cg_flag[i] = 0;
  for (int i = 0; i < n * n; ++i) dst[i] = 0;
  int ctx_set = 0, temp_diag = -1, temp_sum = -1;
  int go_rice = 0;
  int cg_last_scanpos = -1, last_scanpos = -1;
  const int cap_half = 1 << (E.q_bits - 1);
  int cgs;
  for (cgs = cg_num - 1; cgs >= 0; cgs--) {
    for (int sp = 15; sp >= 0; sp--) {
      const int scanpos = cgs * 16 + sp;
      const int blkpos = scan[scanpos];
      const int64_t prod = (int64_t)iabs_((int)coef[blkpos]) * q;
      const int32_t cap = 0x7fffffff - cap_half;
      const int32_t level_double = (int32_t)(prod < cap ? prod : cap);
      const uint32_t max_abs_level = (uint32_t)(level_double + cap_half) >> E.q_bits;
      const double err = (double)level_double;
      cost_coeff0[scanpos] = err * err * E.error_scale;
      dst[blkpos] = (int16_t)max_abs_level;
      if (max_abs_level > 0) { last_scanpos = scanpos; cg_last_scanpos = cgs; break; }
      block_uncoded_cost += cost_coeff0[scanpos];
      base_cost += cost_coeff0[scanpos];
    }
    if (last_scanpos != -1) break;
  }
  if (last_scanpos == -1) return 0;
  for (; cgs >= 0; cgs--) cost_cg_sig[cgs] = 0;

  for (cgs = cg_last_scanpos; cgs >= 0; cgs--) {
    // the group's raster index = position of its first coefficient / 4
    const int first = scan[cgs * 16];
    const int cg_pos_x = (first & (n - 1)) >> 2, cg_pos_y = (first >> l2) >> 2;
    const int cg_blkpos = cg_pos_y * cgw + cg_pos_x;
    double rd_coded = 0, rd_uncoded = 0, rd_sig = 0, rd_sig0 = 0;
    int nnz_before_pos0 = 0;
    for (int sp = 15; sp >= 0; sp--) {
      const int scanpos = cgs * 16 + sp;
      if (scanpos > last_scanpos) continue;
      const int blkpos = scan[scanpos];
      const int64_t prod = (int64_t)iabs_((int)coef[blkpos]) * q;
      const int32_t cap = 0x7fffffff - cap_half;
      const int32_t level_double = (int32_t)(prod < cap ? prod : cap);
      const uint32_t max_abs_level = (uint32_t)(level_double + cap_half) >> E.q_bits;
      dst[blkpos] = (int16_t)max_abs_level;
      const double err = (double)level_double;
      cost_coeff0[scanpos] = err * err * E.error_scale;
      block_uncoded_cost += cost_coeff0[scanpos];
      {
        const int pos_y = blkpos >> l2, pos_x = blkpos - (pos_y << l2);
        int ctx_sig = 0;
        if (scanpos != last_scanpos) ctx_sig = sig_ctx_abs(dst, pos_x, pos_y, n, color, &temp_diag, &temp_sum);
        if (temp_diag != -1)
          ctx_set = ((temp_sum < 4 ? temp_sum : 4) + 1) +
                    (!temp_diag ? ((color == 0) ? 15 : 5) : (color == 0) ? (temp_diag < 3 ? 10 : (temp_diag < 10 ? 5 : 0)) : 0);
        else ctx_set = 0;
        if (reg_bins < 4) go_rice = go_rice_par(template_abs_sum(dst, 0, pos_x, pos_y, n));
        const int level = (int)coded_level(E, &cost_coeff[scanpos], cost_coeff0[scanpos], &cost_sig[scanpos], level_double, max_abs_level,
                                           scanpos == last_scanpos ? 0 : ctx_sig, ctx_set, go_rice, reg_bins, scanpos == last_scanpos);
        dst[blkpos] = (int16_t)level;
        base_cost += cost_coeff[scanpos];
        if ((scanpos % 16 == 0) && scanpos > 0) go_rice = 0;
        else if (reg_bins >= 4) {
          reg_bins -= (uint32_t)((level < 2 ? level : 3) + (scanpos != last_scanpos));
          go_rice = go_rice_par(template_abs_sum(coef, 4, pos_x, pos_y, n));       // sic: the INPUT coefficients (rdo.c:1697)
        }
      }
      rd_sig += cost_sig[scanpos];
      if (sp == 0) rd_sig0 = cost_sig[scanpos];
      if (dst[blkpos]) {
        cg_flag[cg_blkpos] = 1;
        rd_coded += cost_coeff[scanpos] - cost_sig[scanpos];
        rd_uncoded += cost_coeff0[scanpos];
        if (sp != 0) nnz_before_pos0++;
      }
    }
    if (cgs) {
      unsigned right = 0, lower = 0;
      if (cg_pos_x + 1 < cgw) right = cg_flag[cg_blkpos + 1];
      if (cg_pos_y + 1 < cgw) lower = cg_flag[cg_blkpos + cgw];
      const int o_grp = M_SIGGRP + (E.t ? 2 : 0) + ((right || lower) ? 1 : 0);
      if (cg_flag[cg_blkpos] == 0) {
        cost_cg_sig[cgs] = lambda * rbits(E, o_grp, 0);
        base_cost += cost_cg_sig[cgs] - rd_sig;
      } else if (cgs < cg_last_scanpos) {
        if (nnz_before_pos0 == 0) { base_cost -= rd_sig0; rd_sig -= rd_sig0; }
        double cost_zero_cg = base_cost;
        cost_cg_sig[cgs] = lambda * rbits(E, o_grp, 1);
        base_cost += cost_cg_sig[cgs];
        cost_zero_cg += lambda * rbits(E, o_grp, 0);
        cost_zero_cg += rd_uncoded;
        cost_zero_cg -= rd_coded;
        cost_zero_cg -= rd_sig;
        if (cost_zero_cg < base_cost) {
          cg_flag[cg_blkpos] = 0;
          base_cost = cost_zero_cg;
          cost_cg_sig[cgs] = lambda * rbits(E, o_grp, 0);
          for (int sp = 15; sp >= 0; sp--) {
            const int scanpos = cgs * 16 + sp;
            const int blkpos = scan[scanpos];
            if (dst[blkpos]) { dst[blkpos] = 0; cost_coeff[scanpos] = cost_coeff0[scanpos]; cost_sig[scanpos] = 0; }
          }
        }
      }
    } else {
      cg_flag[cg_blkpos] = 1;
    }
  }

  double best_cost;
  int best_last_idx_p1 = 0;
  {
    const int o_cbf = color == 0 ? M_CBF_LUMA : color == 1 ? M_CBF_CB : M_CBF_CR + (cbf_u ? 1 : 0);
    best_cost = block_uncoded_cost + lambda * rbits(E, o_cbf, 0);
    base_cost += lambda * rbits(E, o_cbf, 1);
  }
  // calc_last_bits (rdo.c:664-700)
  int32_t last_x_bits[32], last_y_bits[32];
  {
    const int prefix_ctx[8] = {0, 0, 0, 3, 6, 10, 15, 21};
    const int off = E.t ? 0 : prefix_ctx[l2];
    const int sh = E.t ? clampi(n >> 3, 0, 2) : ((l2 + 1) >> 2);
    int32_t bx = 0, by = 0;
    int ctx;
    for (ctx = 0; ctx < group_idx(n - 1); ctx++) {
      const int o = off + (ctx >> sh);
      const int mx = M_LASTX + (E.t ? 20 : 0) + o, my = M_LASTY + (E.t ? 20 : 0) + o;
      last_x_bits[ctx] = bx + rbits(E, mx, 0); bx += rbits(E, mx, 1);
      last_y_bits[ctx] = by + rbits(E, my, 0); by += rbits(E, my, 1);
    }
    last_x_bits[ctx] = bx; last_y_bits[ctx] = by;
  }
  int found_last = 0;
  for (cgs = cg_last_scanpos; cgs >= 0; cgs--) {
    const int first = scan[cgs * 16];
    const int cg_blkpos = ((first >> l2) >> 2) * cgw + ((first & (n - 1)) >> 2);
    base_cost -= cost_cg_sig[cgs];
    if (cg_flag[cg_blkpos]) {
      for (int sp = 15; sp >= 0; sp--) {
        const int scanpos = cgs * 16 + sp;
        if (scanpos > last_scanpos) continue;
        const int blkpos = scan[scanpos];
        if (dst[blkpos]) {
          const int pos_y = blkpos >> l2, pos_x = blkpos - (pos_y << l2);
          const int cx = group_idx(pos_x), cy = group_idx(pos_y);
          double cl = last_x_bits[cx] + last_y_bits[cy];
          if (cx > 3) cl += 32768 * ((cx - 2) >> 1);
          if (cy > 3) cl += 32768 * ((cy - 2) >> 1);
          const double cost_last = lambda * cl;
          const double total = base_cost + cost_last - cost_sig[scanpos];
          if (total < best_cost) { best_last_idx_p1 = scanpos + 1; best_cost = total; }
          if (dst[blkpos] > 1) { found_last = 1; break; }
          base_cost -= cost_coeff[scanpos];
          base_cost += cost_coeff0[scanpos];
        } else {
          base_cost -= cost_sig[scanpos];
        }
      }
      if (found_last) break;
    }
  }
  int any = 0;
  for (int scanpos = 0; scanpos < best_last_idx_p1; scanpos++) {
    const int b = scan[scanpos], level = dst[b];
    any |= level;
    dst[b] = (int16_t)((coef[b] < 0) ? -level : level);
  }
  for (int scanpos = best_last_idx_p1; scanpos <= last_scanpos; scanpos++) dst[scan[scanpos]] = 0;
  return any != 0;
}

// one position of uvg_rdoq's main walk (rdo.c:1600-1716): the level that survives, its cost and the cost of its significance flag.
// regular: regular bins remain (reg_bins >= 4), go_rice is then the value carried from the position coded before; otherwise the
// position is priced as bypass-coded and its Rice parameter comes from the levels decided around it.
struct rdoq_pos { int level; double cc, cs; };
#if defined(__HIPCC__)
#define CTU_INLINE __attribute__((always_inline))
#else
#define CTU_INLINE
#endif
template <typename DP> CTU_INLINE CTU_DEV rdoq_pos rdoq_decide_inl(const rdoq_env &E, CTU_LDS const int16_t *coef, DP dst, int n, int l2, int color, int blkpos, bool is_last,
                             bool regular, int go_rice, double c0, int *mal_out)
{
  const int cap_half = 1 << (E.q_bits - 1);
  const int64_t prod = (int64_t)iabs_((int)coef[blkpos]) * E.q;
  const int32_t cap = 0x7fffffff - cap_half;
  const int32_t level_double = (int32_t)(prod < cap ? prod : cap);
  const uint32_t max_abs_level = (uint32_t)(level_double + cap_half) >> E.q_bits;
  const int pos_y = blkpos >> l2, pos_x = blkpos - (pos_y << l2);
  int ctx_sig = 0, ctx_set = 0;
  if (!is_last) {
    int diag, tsum;
    ctx_sig = sig_ctx_abs(dst, pos_x, pos_y, n, color, &diag, &tsum);
    ctx_set = ((tsum < 4 ? tsum : 4) + 1) + (!diag ? ((color == 0) ? 15 : 5) : (color == 0) ? (diag < 3 ? 10 : (diag < 10 ? 5 : 0)) : 0);
  }
  if (!regular) go_rice = go_rice_par(template_abs_sum(dst, 0, pos_x, pos_y, n));
  rdoq_pos r;
  r.cs = 0;        // (the reference leaves cost_sig of the last position as it was: never read before it is written, see below)
  r.level = (int)coded_level(E, &r.cc, c0, &r.cs, level_double, max_abs_level, ctx_sig, ctx_set, go_rice, regular ? 4u : 0u, is_last);
  *mal_out = (int)max_abs_level;
  return r;
}
template <typename DP> CTU_NOINLINE CTU_DEV rdoq_pos rdoq_decide(const rdoq_env &E, CTU_LDS const int16_t *coef, DP dst, int n, int l2, int color, int blkpos, bool is_last,
                             bool regular, int go_rice, double c0, int *mal_out)
{
  return rdoq_decide_inl(E, coef, dst, n, l2, color, blkpos, is_last, regular, go_rice, c0, mal_out);
}

#if !defined(__HIPCC__)
// uvg_rdoq (rdo.c:1449-1870) by the first wave: same arithmetic as rdoq_serial, restructured around what is sequential in it.
//   * every position's quantisation candidates and its level-0 cost: all positions at once;
//   * a level decision reads only levels decided on later anti-diagonals (the context template looks right / down), so the <= 4
//     positions of one anti-diagonal of a 4x4 group are decided together, 7 steps per group -- as long as the regular-bin budget
//     cannot run out inside the group (an upper bound from the candidates says so) or has run out for good; the one or two groups
//     where it does run out are walked position by position;
//   * the double-precision sums the reference forms in scan order (base cost, group statistics) and the group decision that
//     compares them: lane 0, from the group's staged costs; the final cbf / last-position search: lane 0.
// Result: V->rq_i[1] = whether any level survived; levels in dst.
template <typename PX> CTU_NOINLINE CTU_DEV void rdoq_wave(lds<PX> *S, scratch *W, const int16_t *coef_, int16_t *dst_, int n, int color, int cbf_u, int qp_scaled,
                                              double lambda, int bitdepth)
{
  wctx *const V = wv_of(S);
  CTU_LDS const int16_t *const coef = LDSP(const int16_t, coef_);
  typename mg_ptr<PX, int16_t>::type const dst = MGP(PX, int16_t, dst_);
  const int l2 = ilog2_dev(n), nn = n * n, cgw = n >> 2;
  const uint16_t *scan = scan_of(S, l2);
  rdoq_env E;
  E.st = S->rdoq_state; E.t = color ? 1 : 0; E.lambda = lambda;
  const int transform_shift = 15 - bitdepth - l2;
  E.q_bits = 14 + qp_scaled / 6 + transform_shift;
  E.q = kQuantScales[qp_scaled % 6];
  double scale = 32768;
  scale = transform_shift >= 0 ? scale / kPow2[2 * transform_shift] : scale * kPow2[-2 * transform_shift];
  E.error_scale = scale / E.q / E.q;
  const bool small = V->rq_cc != nullptr;        // the per-position cost arrays are in LDS (this wave's depth has them)
  double *CC = small ? V->rq_cc : W->cost_coeff, *CS = small ? V->rq_cs : W->cost_sig, *C0 = W->cost_coeff0;
#define RQ_LD(p) (small ? *(p) : CTU_GLOAD(p))
  double *cost_cg_sig = (double *)V->t1;         // (the transform's other buffer: dead while a block is quantised; <= 64 groups)
  const int cap_half = 1 << (E.q_bits - 1);
  // ---- every position: candidate, level-0 cost; the last candidate in scan order ----
#if defined(__HIPCC__) && defined(CTU_PROFILE)
  unsigned long long tq = __builtin_amdgcn_s_memtime();
#endif
  int my_last = -1;
  WFOR(sp, nn) {
    const int blk = scan[sp];
    const int64_t prod = (int64_t)iabs_((int)coef[blk]) * E.q;
    const int32_t cap = 0x7fffffff - cap_half;
    const int32_t level_double = (int32_t)(prod < cap ? prod : cap);
    const int mal = (int)((uint32_t)(level_double + cap_half) >> E.q_bits);
    const double err = (double)level_double;
    C0[sp] = err * err * E.error_scale;
    dst[blk] = (int16_t)mal;
    if (mal > 0 && sp > my_last) my_last = sp;
    if (sp < 64) V->cg_flag[sp] = 0;
  }
#if defined(__HIPCC__)
  for (int o = 32; o >= 1; o >>= 1) { const int v = __shfl_xor(my_last, o, 64); my_last = v > my_last ? v : my_last; }
#endif
  const int last_scanpos = my_last;
  WSYNC();
  if (last_scanpos < 0) { if (CTU_TID == 0) V->rq_i[1] = 0; WSYNC(); return; }
  RQ_T(12);
  const int cg_last = last_scanpos >> 4;
  // lane 0's running sums (rdo.c:1556-1583: the positions behind the last candidate only add their level-0 cost)
  double block_uncoded_cost = 0, base_cost = 0;
  if (CTU_TID == 0) {
    for (int sp = nn - 1; sp > last_scanpos; --sp) { const double c = RQ_LD(&C0[sp]); block_uncoded_cost += c; base_cost += c; }
    for (int g = 0; g <= cg_last; ++g) cost_cg_sig[g] = 0;
    V->rq_i[4] = (int)((uint32_t)(nn * 28) >> 4);      // reg_bins
    V->rq_i[5] = 1;                                    // regular bins remain
  }
  WSYNC();
  RQ_T(13);
  for (int cgs = cg_last; cgs >= 0; --cgs) {
    const int first = scan[cgs * 16];
    const int cg_pos_x = (first & (n - 1)) >> 2, cg_pos_y = (first >> l2) >> 2;
    const int cg_blkpos = cg_pos_y * cgw + cg_pos_x;
    int reg_bins = V->rq_i[4];
    const int regular = V->rq_i[5];
    // can the regular-bin budget run out inside this group?  (a position spends at most min(candidate, 2 -> 3) + 1 bins)
    int fast = !regular;
    if (regular) {
      int bound = 0;
      WFOR(sp, 16) {
        const int scanpos = cgs * 16 + sp;
        if (scanpos <= last_scanpos) { const int mal = dst[scan[scanpos]]; bound += (mal < 2 ? mal : 3) + (scanpos != last_scanpos); }
      }
#if defined(__HIPCC__)
      for (int o = 8; o >= 1; o >>= 1) bound += __shfl_xor(bound, o, 64);
      bound = __shfl(bound, 0, 64);
#endif
      fast = reg_bins - bound >= 4;
    }
    if (fast) {
      // a position can be decided once the positions of its context template that may keep a level are decided; the others hold
      // their final 0 already.  Lanes 0..15 own the group's scan positions; the rounds follow the chains of candidates (<= 7).
#if defined(__HIPCC__)
      {
        const int sp = CTU_TID & 15, scanpos = cgs * 16 + sp;
        const bool mine = CTU_TID < 16 && scanpos <= last_scanpos;
        const int blk = scan[mine ? scanpos : cgs * 16];
        const int mal0 = mine ? (int)dst[blk] : 0;
        const unsigned nz = (unsigned)__ballot(mine && mal0 > 0) & 0xffffu;
        const unsigned deps = (unsigned)S->deps4[sp] & nz;
        int go_rice = 0;
        if (mine && regular && sp != 15 && scanpos != last_scanpos) {
          const int nb = scan[scanpos + 1];
          go_rice = go_rice_par(template_abs_sum(coef, 4, nb & (n - 1), nb >> l2, n));      // sic: the INPUT coefficients (rdo.c:1697)
        }
        const double c0 = mine ? RQ_LD(&C0[scanpos]) : 0.0;
        bool done = !mine;
        unsigned decided = 0;
        for (;;) {
          const bool ready = !done && (deps & ~decided) == 0;
          if (ready) {
            int mal;
            const rdoq_pos r = rdoq_decide(E, coef, dst, n, l2, color, blk, scanpos == last_scanpos, regular != 0, go_rice, c0, &mal);
            dst[blk] = (int16_t)r.level;
            V->rq_stage[sp] = r.cc; V->rq_stage[16 + sp] = r.cs;
            done = true;
          }
          WSYNC();
          decided = (unsigned)__ballot(done) & 0xffffu;
          if (decided == 0xffffu) break;
        }
      }
#else
      {
        unsigned nz = 0, decided = 0, all = 0;
        for (int sp = 0; sp < 16; ++sp) { const int scanpos = cgs * 16 + sp; if (scanpos <= last_scanpos) { all |= 1u << sp; if (dst[scan[scanpos]] > 0) nz |= 1u << sp; } }
        decided = ~all & 0xffffu;
        while (decided != 0xffffu) {
          const unsigned before = decided;
          int16_t newlev[16];
          for (int sp = 0; sp < 16; ++sp) {
            if ((before >> sp) & 1) continue;
            if (((unsigned)S->deps4[sp] & nz) & ~before) continue;
            const int scanpos = cgs * 16 + sp, blk = scan[scanpos];
            int go_rice = 0;
            if (regular && sp != 15 && scanpos != last_scanpos) {
              const int nb = scan[scanpos + 1];
              go_rice = go_rice_par(template_abs_sum(coef, 4, nb & (n - 1), nb >> l2, n));
            }
            int mal;
            const rdoq_pos r = rdoq_decide(E, coef, dst, n, l2, color, blk, scanpos == last_scanpos, regular != 0, go_rice, RQ_LD(&C0[scanpos]), &mal);
            newlev[sp] = (int16_t)r.level;
            V->rq_stage[sp] = r.cc; V->rq_stage[16 + sp] = r.cs;
            decided |= 1u << sp;
          }
          for (int sp = 0; sp < 16; ++sp) if (((decided & ~before) >> sp) & 1) dst[scan[cgs * 16 + sp]] = newlev[sp];    // a round's levels appear together
        }
      }
#endif
    } else if (CTU_TID == 0) {
      // the budget may run out in this group: position by position, exactly as the reference walks
      int go_rice = 0;
      for (int sp = 15; sp >= 0; --sp) {
        const int scanpos = cgs * 16 + sp;
        if (scanpos > last_scanpos) continue;
        const int blk = scan[scanpos];
        int mal;
        const rdoq_pos r = rdoq_decide(E, coef, dst, n, l2, color, blk, scanpos == last_scanpos, reg_bins >= 4, go_rice, RQ_LD(&C0[scanpos]), &mal);
        dst[blk] = (int16_t)r.level;
        V->rq_stage[sp] = r.cc; V->rq_stage[16 + sp] = r.cs;
        if ((scanpos % 16 == 0) && scanpos > 0) go_rice = 0;
        else if (reg_bins >= 4) {
          reg_bins -= (r.level < 2 ? r.level : 3) + (scanpos != last_scanpos);
          go_rice = go_rice_par(template_abs_sum(coef, 4, blk & (n - 1), blk >> l2, n));
        }
      }
      V->rq_i[4] = reg_bins;
      V->rq_i[5] = reg_bins >= 4;
    }
    WSYNC();
    RQ_T(14);
    if (CTU_TID == 0) {
      // the sums in scan order and the group's decision (rdo.c:1689-1772)
      double rd_coded = 0, rd_uncoded = 0, rd_sig = 0, rd_sig0 = 0;
      int nnz_before_pos0 = 0, flag = 0, spent = 0;
      for (int sp = 15; sp >= 0; --sp) {
        const int scanpos = cgs * 16 + sp;
        if (scanpos > last_scanpos) continue;
        const double cc = V->rq_stage[sp], cs = V->rq_stage[16 + sp], c0 = RQ_LD(&C0[scanpos]);
        const int level = dst[scan[scanpos]];
        block_uncoded_cost += c0;
        base_cost += cc;
        // (the first position of a group other than group 0 resets the Rice parameter INSTEAD of paying: rdo.c:1690-1697)
        if (!(sp == 0 && cgs > 0)) spent += (level < 2 ? level : 3) + (scanpos != last_scanpos);
        rd_sig += cs;
        if (sp == 0) rd_sig0 = cs;
        if (level) {
          flag = 1;
          rd_coded += cc - cs;
          rd_uncoded += c0;
          if (sp != 0) nnz_before_pos0++;
        }
      }
      if (fast && regular) V->rq_i[4] = reg_bins - spent;
      int zeroed = 0;
      if (cgs) {
        unsigned right = 0, lower = 0;
        if (cg_pos_x + 1 < cgw) right = V->cg_flag[cg_blkpos + 1];
        if (cg_pos_y + 1 < cgw) lower = V->cg_flag[cg_blkpos + cgw];
        const int o_grp = M_SIGGRP + (E.t ? 2 : 0) + ((right || lower) ? 1 : 0);
        if (!flag) {
          cost_cg_sig[cgs] = lambda * rbits(E, o_grp, 0);
          base_cost += cost_cg_sig[cgs] - rd_sig;
        } else if (cgs < cg_last) {
          if (nnz_before_pos0 == 0) { base_cost -= rd_sig0; rd_sig -= rd_sig0; }
          double cost_zero_cg = base_cost;
          cost_cg_sig[cgs] = lambda * rbits(E, o_grp, 1);
          base_cost += cost_cg_sig[cgs];
          cost_zero_cg += lambda * rbits(E, o_grp, 0);
          cost_zero_cg += rd_uncoded;
          cost_zero_cg -= rd_coded;
          cost_zero_cg -= rd_sig;
          if (cost_zero_cg < base_cost) {
            flag = 0;
            zeroed = 1;
            base_cost = cost_zero_cg;
            cost_cg_sig[cgs] = lambda * rbits(E, o_grp, 0);
          }
        }
      } else {
        flag = 1;
      }
      V->cg_flag[cg_blkpos] = (uint8_t)flag;
      V->rq_i[6] = zeroed;
    }
    WSYNC();
    RQ_T(15);
    {
      // the group's costs go to the per-position arrays the last-position search reads; a zeroed group's positions fall back to level 0
      const int zeroed = V->rq_i[6];
      WFOR(sp, 16) {
        const int scanpos = cgs * 16 + sp;
        if (scanpos <= last_scanpos) {
          const int blk = scan[scanpos];
          if (zeroed && dst[blk]) { dst[blk] = 0; CC[scanpos] = RQ_LD(&C0[scanpos]); CS[scanpos] = 0; }
          else { CC[scanpos] = V->rq_stage[sp]; CS[scanpos] = V->rq_stage[16 + sp]; }
        }
      }
    }
    WSYNC();
  }
  RQ_T(16);
  // ---- coded block flag and the last significant position (rdo.c:1774-1833) ----
  if (CTU_TID == 0) {
    double best_cost;
    int best_last_idx_p1 = 0;
    {
      const int o_cbf = color == 0 ? (CTU_RQ_ROOT(V) ? 243 : M_CBF_LUMA) : color == 1 ? M_CBF_CB : M_CBF_CR + (cbf_u ? 1 : 0);
      best_cost = block_uncoded_cost + lambda * rbits(E, o_cbf, 0);
      base_cost += lambda * rbits(E, o_cbf, 1);
    }
    const int32_t *last_x_bits = S->last_bits + last_bits_off(E.t, l2, 0), *last_y_bits = S->last_bits + last_bits_off(E.t, l2, 1);
    int found_last = 0;
    for (int cgs = cg_last; cgs >= 0; cgs--) {
      const int first = scan[cgs * 16];
      const int cg_blkpos = ((first >> l2) >> 2) * cgw + ((first & (n - 1)) >> 2);
      base_cost -= cost_cg_sig[cgs];
      if (V->cg_flag[cg_blkpos]) {
        for (int sp = 15; sp >= 0; sp--) {
          const int scanpos = cgs * 16 + sp;
          if (scanpos > last_scanpos) continue;
          const int blkpos = scan[scanpos];
          if (dst[blkpos]) {
            const int pos_y = blkpos >> l2, pos_x = blkpos - (pos_y << l2);
            const int cx = group_idx(pos_x), cy = group_idx(pos_y);
            double cl = last_x_bits[cx] + last_y_bits[cy];
            if (cx > 3) cl += 32768 * ((cx - 2) >> 1);
            if (cy > 3) cl += 32768 * ((cy - 2) >> 1);
            const double cost_last = lambda * cl;
            const double total = base_cost + cost_last - RQ_LD(&CS[scanpos]);
            if (total < best_cost) { best_last_idx_p1 = scanpos + 1; best_cost = total; }
            if (dst[blkpos] > 1) { found_last = 1; break; }
            base_cost -= RQ_LD(&CC[scanpos]);
            base_cost += RQ_LD(&C0[scanpos]);
          } else {
            base_cost -= RQ_LD(&CS[scanpos]);
          }
        }
        if (found_last) break;
      }
    }
    V->rq_i[0] = best_last_idx_p1;
    V->rq_i[1] = best_last_idx_p1 > 0;
  }
  WSYNC();
  RQ_T(17);
  const int best_last_idx_p1 = V->rq_i[0];
  WFOR(scanpos, last_scanpos + 1) {
    const int b = scan[scanpos];
    if (scanpos < best_last_idx_p1) { const int level = dst[b]; dst[b] = (int16_t)((coef[b] < 0) ? -level : level); }
    else dst[b] = 0;
  }
  WSYNC();
  RQ_T(18);
}
#else
// uvg_rdoq (rdo.c:1449-1870) by one wave: same arithmetic as rdoq_serial, restructured around what is sequential in it.
//   * every position's quantisation candidate: all positions at once;
//   * the positions of a 4x4 group are decided together by lanes 0..15.  A decision reads only levels of positions later in scan
//     order (the context template looks right / down), so the group's decisions are the unique solution of a set of equations over
//     a DAG: the lanes start from the candidates, re-decide against each other's current levels and stop when a round changes
//     nothing -- at that point every position holds exactly what the reference's walk leaves there (<= 8 rounds, mostly 2).
//     This needs the regular-bin budget not to run out inside the group (an upper bound from the candidates says so) or to have run
//     out for good; the one or two groups where it does run out are walked position by position by lane 0;
//   * the double-precision sums the reference forms in scan order (base cost, group statistics, the last-position search): every
//     lane forms them, in the reference's order, from the owners' registers (v_readlane) -- no memory in the chain.
// Result: V->rq_i[1] = whether any level survived; levels in dst.
CTU_DEV double rl64(double v, int lane)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
template <typename PX> CTU_INLINE1 CTU_DEV void rdoq_wave(lds<PX> *S, scratch *W, const int16_t *coef_, int16_t *dst_, int n, int color, int cbf_u, int qp_scaled,
                                              double lambda, int bitdepth)
{
  wctx *const V = wv_of(S);
  CTU_LDS const int16_t *const coef = LDSP(const int16_t, coef_);
  typename mg_ptr<PX, int16_t>::type const dst = MGP(PX, int16_t, dst_);
  const int l2 = ilog2_dev(n), nn = n * n, cgw = n >> 2;
  const uint16_t *scan = scan_of(S, l2);
  rdoq_env E;
  E.st = S->rdoq_state; E.t = color ? 1 : 0; E.lambda = lambda;
  const int transform_shift = 15 - bitdepth - l2;
  E.q_bits = 14 + qp_scaled / 6 + transform_shift;
  E.q = kQuantScales[qp_scaled % 6];
  double scale = 32768;
  scale = transform_shift >= 0 ? scale / kPow2[2 * transform_shift] : scale * kPow2[-2 * transform_shift];
  E.error_scale = scale / E.q / E.q;
  const bool small = V->rq_cc != nullptr;        // the per-position cost arrays are in LDS (this wave's depth has them)
  CTU_LDS double *const CCl = LDSP(double, V->rq_cc), *const CSl = LDSP(double, V->rq_cs);
  // blocks larger than 8x8 keep them in the workgroup's global scratch, and TWO waves can be quantising such a block at the same
  // moment (depth 1: 32x32 luma, 16x16 chroma; depth 2: 16x16 luma): each depth has its own pair of arrays
  const bool d1 = S->vsel[CTU_WAVE] == 3;
  double *const CCg = d1 ? W->cost_coeff : W->cost_coeff0, *const CSg = d1 ? W->cost_sig : W->cost_coeff0 + 512;
  double *cost_cg_sig = (double *)V->t1;         // (the transform's other buffer: dead while a block is quantised; <= 64 groups)
  const int cap_half = 1 << (E.q_bits - 1);
  const int32_t cap = 0x7fffffff - cap_half;
#if defined(CTU_PROFILE)
  unsigned long long tq = __builtin_amdgcn_s_memtime();
#endif
  // ---- every position: candidate; the last candidate in scan order ----
  int my_last = -1;
  WFOR(sp, nn) {
    const int blk = scan[sp];
    const int64_t prod = (int64_t)iabs_((int)coef[blk]) * E.q;
    const int32_t level_double = (int32_t)(prod < cap ? prod : cap);
    const int mal = (int)((uint32_t)(level_double + cap_half) >> E.q_bits);
    dst[blk] = (int16_t)mal;
    if (mal > 0 && sp > my_last) my_last = sp;
    if (sp < 64) V->cg_flag[sp] = 0;
  }
  for (int o = 32; o >= 1; o >>= 1) { const int v = __shfl_xor(my_last, o, 64); my_last = v > my_last ? v : my_last; }
  const int last_scanpos = my_last;
  WSYNC();
  if (last_scanpos < 0) { if (CTU_TID == 0) V->rq_i[1] = 0; WSYNC(); return; }
  RQ_T(12);
  const int cg_last = last_scanpos >> 4;
  const int lane = CTU_TID, sp = lane & 15;
  const bool own = lane < 16;
  // the positions behind the last candidate only add their level-0 cost (rdo.c:1556-1583), in descending scan order
  double block_uncoded_cost = 0, base_cost = 0;
  for (int top = nn - 16; top + 15 > last_scanpos; top -= 16) {
    const int scanpos = top + sp;
    const int64_t prod = (int64_t)iabs_((int)coef[scan[scanpos]]) * E.q;
    const double err = (double)(int32_t)(prod < cap ? prod : cap);
    const double c0 = err * err * E.error_scale;
#pragma nounroll
    for (int k = 15; k >= 0; --k) if (top + k > last_scanpos) { const double c = rl64(c0, k); block_uncoded_cost += c; base_cost += c; }
  }
  if (CTU_TID == 0) for (int g = 0; g <= cg_last; ++g) cost_cg_sig[g] = 0;
  int reg_bins = (int)((uint32_t)(nn * 28) >> 4);
  WSYNC();
  RQ_T(13);
  double f_cc = 0, f_cs = 0, f_c0 = 0;          // cg_last == 0 (every 4x4 block, sparse larger ones): the group's numbers stay in
  int f_lev = 0;                                 // registers for the last-position search -- no cost arrays
  for (int cgs = cg_last; cgs >= 0; --cgs) {
    const int first = scan[cgs * 16];
    const int cg_pos_x = (first & (n - 1)) >> 2, cg_pos_y = (first >> l2) >> 2;
    const int cg_blkpos = cg_pos_y * cgw + cg_pos_x;
    const int regular = reg_bins >= 4;
    const int scanpos = cgs * 16 + sp;
    const bool mine = own && scanpos <= last_scanpos;
    const int blk = scan[mine ? scanpos : cgs * 16];
    const bool is_last = scanpos == last_scanpos;
    const int64_t prod = (int64_t)iabs_((int)coef[blk]) * E.q;
    const int32_t level_double = (int32_t)(prod < cap ? prod : cap);
    const int mal0 = mine ? (int)((uint32_t)(level_double + cap_half) >> E.q_bits) : 0;
    const double c0 = mine ? (double)level_double * (double)level_double * E.error_scale : 0.0;
    // can the regular-bin budget run out inside this group?  (a position spends at most min(candidate, 2 -> 3) + 1 bins)
    int fast = !regular;
    if (regular) {
      int bound = mine ? (mal0 < 2 ? mal0 : 3) + (is_last ? 0 : 1) : 0;
      for (int o = 8; o >= 1; o >>= 1) bound += __shfl_xor(bound, o, 64);
      bound = __builtin_amdgcn_readfirstlane(bound);
      fast = reg_bins - bound >= 4;
    }
    double cc = 0, cs = 0;
    int lev = 0;
    if (fast) {
      int go_rice = 0;
      if (mine && regular && sp != 15 && !is_last) {
        const int nb = scan[scanpos + 1];
        go_rice = go_rice_par(template_abs_sum(coef, 4, nb & (n - 1), nb >> l2, n));      // sic: the INPUT coefficients (rdo.c:1697)
      }
      lev = mal0;
      for (;;) {
        bool changed = false;
        if (mine) {
          int mal;
          const rdoq_pos r = rdoq_decide_inl(E, coef, dst, n, l2, color, blk, is_last, regular != 0, go_rice, c0, &mal);
          cc = r.cc; cs = r.cs;
          changed = r.level != lev;
          lev = r.level;
        }
        WSYNC();                               // every lane has read the levels it needs
        if (changed) dst[blk] = (int16_t)lev;
        const bool any = __ballot(changed) != 0;
        WSYNC();
        if (!any) break;
      }
    } else {
      // the budget may run out in this group: position by position, exactly as the reference walks
      if (CTU_TID == 0) {
        int go_rice = 0, rb = reg_bins;
        for (int k = 15; k >= 0; --k) {
          const int spos = cgs * 16 + k;
          if (spos > last_scanpos) continue;
          const int b = scan[spos];
          const int64_t pr = (int64_t)iabs_((int)coef[b]) * E.q;
          const double e0 = (double)(int32_t)(pr < cap ? pr : cap);
          int mal;
          const rdoq_pos r = rdoq_decide(E, coef, dst, n, l2, color, b, spos == last_scanpos, rb >= 4, go_rice, e0 * e0 * E.error_scale, &mal);
          dst[b] = (int16_t)r.level;
          V->rq_stage[k] = r.cc; V->rq_stage[16 + k] = r.cs;
          if ((spos % 16 == 0) && spos > 0) go_rice = 0;
          else if (rb >= 4) {
            rb -= (r.level < 2 ? r.level : 3) + (spos != last_scanpos);
            go_rice = go_rice_par(template_abs_sum(coef, 4, b & (n - 1), b >> l2, n));
          }
        }
        V->rq_i[4] = rb;
      }
      WSYNC();
      if (mine) { cc = V->rq_stage[sp]; cs = V->rq_stage[16 + sp]; lev = dst[blk]; }
      reg_bins = V->rq_i[4];
      WSYNC();
    }
    RQ_T(14);
    // the sums in scan order and the group's decision (rdo.c:1689-1772): every lane, from the owners' registers
    double rd_coded = 0, rd_uncoded = 0, rd_sig = 0, rd_sig0 = 0;
    int nnz_before_pos0 = 0, flag = 0, spent = 0;
#pragma nounroll
    for (int k = 15; k >= 0; --k) {
      if (cgs * 16 + k > last_scanpos) continue;
      const double kc = rl64(cc, k), ks = rl64(cs, k), k0 = rl64(c0, k);
      const int level = __builtin_amdgcn_readlane(lev, k);
      block_uncoded_cost += k0;
      base_cost += kc;
      // (the first position of a group other than group 0 resets the Rice parameter INSTEAD of paying: rdo.c:1690-1697)
      if (!(k == 0 && cgs > 0)) spent += (level < 2 ? level : 3) + (cgs * 16 + k != last_scanpos);
      if (cgs) {                                // the group statistics only matter where a group can be zeroed out
        rd_sig += ks;
        if (k == 0) rd_sig0 = ks;
        if (level) {
          flag = 1;
          rd_coded += kc - ks;
          rd_uncoded += k0;
          if (k != 0) nnz_before_pos0++;
        }
      }
    }
    if (fast && regular) reg_bins -= spent;
    int zeroed = 0;
    if (cgs) {
      unsigned right = 0, lower = 0;
      if (cg_pos_x + 1 < cgw) right = V->cg_flag[cg_blkpos + 1];
      if (cg_pos_y + 1 < cgw) lower = V->cg_flag[cg_blkpos + cgw];
      const int o_grp = M_SIGGRP + (E.t ? 2 : 0) + ((right || lower) ? 1 : 0);
      double cgc = 0;
      if (!flag) {
        cgc = lambda * rbits(E, o_grp, 0);
        base_cost += cgc - rd_sig;
      } else if (cgs < cg_last) {
        if (nnz_before_pos0 == 0) { base_cost -= rd_sig0; rd_sig -= rd_sig0; }
        double cost_zero_cg = base_cost;
        cgc = lambda * rbits(E, o_grp, 1);
        base_cost += cgc;
        cost_zero_cg += lambda * rbits(E, o_grp, 0);
        cost_zero_cg += rd_uncoded;
        cost_zero_cg -= rd_coded;
        cost_zero_cg -= rd_sig;
        if (cost_zero_cg < base_cost) {
          flag = 0;
          zeroed = 1;
          base_cost = cost_zero_cg;
          cgc = lambda * rbits(E, o_grp, 0);
        }
      }
      WSYNC();                                 // (the neighbours' flags are read)
      if (CTU_TID == 0) cost_cg_sig[cgs] = cgc;
    } else {
      flag = 1;
    }
    if (CTU_TID == 0) V->cg_flag[cg_blkpos] = (uint8_t)flag;
    // the group's costs go to the per-position arrays the last-position search reads; a zeroed group's positions fall back to level 0
    if (cg_last == 0) { f_cc = cc; f_cs = cs; f_c0 = c0; f_lev = lev; }
    else if (mine) {
      const double wc = (zeroed && lev) ? c0 : cc, ws = (zeroed && lev) ? 0.0 : cs;
      if (zeroed && lev) dst[blk] = 0;
      if (small) { CCl[scanpos] = wc; CSl[scanpos] = ws; } else { CCg[scanpos] = wc; CSg[scanpos] = ws; }
    }
    WSYNC();
    RQ_T(15);
  }
  RQ_T(16);
  // ---- coded block flag and the last significant position (rdo.c:1774-1833) ----
  double best_cost;
  int best_last_idx_p1 = 0;
  {
    const int o_cbf = color == 0 ? (CTU_RQ_ROOT(V) ? 243 : M_CBF_LUMA) : color == 1 ? M_CBF_CB : M_CBF_CR + (cbf_u ? 1 : 0);
    best_cost = block_uncoded_cost + lambda * rbits(E, o_cbf, 0);
    base_cost += lambda * rbits(E, o_cbf, 1);
  }
  const int32_t *last_x_bits = S->last_bits + last_bits_off(E.t, l2, 0), *last_y_bits = S->last_bits + last_bits_off(E.t, l2, 1);
  int found_last = 0;
  for (int cgs = cg_last; cgs >= 0 && !found_last; cgs--) {
    const int first = scan[cgs * 16];
    const int cg_blkpos = ((first >> l2) >> 2) * cgw + ((first & (n - 1)) >> 2);
    base_cost -= cost_cg_sig[cgs];
    if (!V->cg_flag[cg_blkpos]) continue;
    // the owners fetch their position's numbers, then the walk runs over registers
    const int scanpos = cgs * 16 + sp;
    const bool mine = own && scanpos <= last_scanpos;
    const int blkpos = scan[mine ? scanpos : cgs * 16];
    const int lev = cg_last == 0 ? f_lev : (mine ? (int)dst[blkpos] : 0);
    double kcs = 0, kcc = 0, k0 = 0, klast = 0;
    if (mine) {
      kcs = cg_last == 0 ? f_cs : (small ? CSl[scanpos] : CTU_GLOAD(&CSg[scanpos]));
      if (lev) {
        if (cg_last == 0) { kcc = f_cc; k0 = f_c0; }
        else {
          kcc = small ? CCl[scanpos] : CTU_GLOAD(&CCg[scanpos]);
          const int64_t prod = (int64_t)iabs_((int)coef[blkpos]) * E.q;
          const double err = (double)(int32_t)(prod < cap ? prod : cap);
          k0 = err * err * E.error_scale;
        }
        const int pos_y = blkpos >> l2, pos_x = blkpos - (pos_y << l2);
        const int cx = group_idx(pos_x), cy = group_idx(pos_y);
        double cl = last_x_bits[cx] + last_y_bits[cy];
        if (cx > 3) cl += 32768 * ((cx - 2) >> 1);
        if (cy > 3) cl += 32768 * ((cy - 2) >> 1);
        klast = lambda * cl;
      }
    }
#pragma nounroll
    for (int k = 15; k >= 0; --k) {
      if (found_last || cgs * 16 + k > last_scanpos) continue;
      const int level = __builtin_amdgcn_readlane(lev, k);
      const double s_ = rl64(kcs, k);
      if (level) {
        const double total = base_cost + rl64(klast, k) - s_;
        if (total < best_cost) { best_last_idx_p1 = cgs * 16 + k + 1; best_cost = total; }
        if (level > 1) { found_last = 1; continue; }
        base_cost -= rl64(kcc, k);
        base_cost += rl64(k0, k);
      } else {
        base_cost -= s_;
      }
    }
  }
  if (CTU_TID == 0) V->rq_i[1] = best_last_idx_p1 > 0;
  RQ_T(17);
  WFOR(scanpos, last_scanpos + 1) {
    const int b = scan[scanpos];
    if (scanpos < best_last_idx_p1) { const int level = dst[b]; dst[b] = (int16_t)((coef[b] < 0) ? -level : level); }
    else dst[b] = 0;
  }
  WSYNC();
  RQ_T(18);
}
#endif

// -------------------------------------------------------------------------------------------- coefficient bit cost ------
CTU_DEV int coeff_remain_bits(uint32_t remainder, uint32_t rice, unsigned cutoff)     // uvg_cabac_write_coeff_remain, cabac.c:318-354
{
  const unsigned threshold = cutoff << rice;
  if (remainder < threshold) return (int)((remainder >> rice) + 1 + rice);
  const unsigned max_prefix = 32 - cutoff - 15;
  unsigned prefix = 0, suffix_len;
  const unsigned code_value = (remainder >> rice) - cutoff;
  if ((int32_t)code_value >= ((1 << max_prefix) - 1)) { prefix = max_prefix; suffix_len = 15; }
  else { while ((int32_t)code_value > ((2 << prefix) - 2)) prefix++; suffix_len = prefix + rice + 1; }
  return (int)(prefix + cutoff + suffix_len);
}
template <typename LP> CTU_DEV int abs_sum_tmpl(LP coeff, int px, int py, int n, int baselevel)      // uvg_abs_sum, context.c:846-877
{
  const auto d = coeff + px + py * n;
  int sum = 0;
  if (px < n - 1) {
    sum += iabs_((int)d[1]);
    if (px < n - 2) sum += iabs_((int)d[2]);
    if (py < n - 1) sum += iabs_((int)d[n + 1]);
  }
  if (py < n - 1) {
    sum += iabs_((int)d[n]);
    if (py < n - 2) sum += iabs_((int)d[n << 1]);
  }
  int v = sum - 5 * baselevel;
  v = v < 31 ? v : 31;
  return v > 0 ? v : 0;
}

// uvg_encode_coeff_nxn in count mode (encode_coding_tree-generic.c:53-323, uvg_encode_last_significant_xy :415-470) on the models m,
// which adapt bin by bin (get_coeff_cabac_cost always lets its COPY adapt; the caller decides whether to keep it); lane 0 only.
// No dependent quantisation, no sign hiding, diagonal scan, regular residual coding.  Returns 0 for an all-zero block (rdo.c:312-320).
CTU_NOINLINE CTU_DEV double coeff_bits_serial(uint32_t *m, const uint16_t *scan, const int16_t *coeff, int n, int color)
{
  const int l2 = ilog2_dev(n);
  const int t = color ? 1 : 0;
  const int cgw = n >> 2;
  uint8_t sig_cg[64];
  for (int i = 0; i < cgw * cgw; ++i) sig_cg[i] = 0;
  int scan_pos_last = -1;
  for (int i = 0; i < n * n; ++i)
    if (coeff[scan[i]]) {
      scan_pos_last = i;
      const int f = scan[i & ~15];
      sig_cg[((f >> l2) >> 2) * cgw + ((f & (n - 1)) >> 2)] = 1;
    }
  if (scan_pos_last < 0) return 0;
  const int scan_cg_last = scan_pos_last >> 4;
  double bits_out = 0;
  {
    const int pos_last = scan[scan_pos_last];
    const int last_y = pos_last >> l2, last_x = pos_last - (last_y << l2);
    const int prefix_ctx[8] = {0, 0, 0, 3, 6, 10, 15, 21};
    const int off = t ? 0 : prefix_ctx[l2];
    const int sh = t ? clampi(n >> 3, 0, 2) : ((l2 + 1) >> 2);
    const int bx = M_LASTX + 20 * t + off, by = M_LASTY + 20 * t + off;
    const int gx = group_idx(last_x), gy = group_idx(last_y), gmax = group_idx(n - 1);
    double bits = 0;
    int k = 0;
    for (; k < gx; k++) m_code(m, 1, bx + (k >> sh), 1, bits);
    if (gx < gmax) m_code(m, 1, bx + (k >> sh), 0, bits);
    k = 0;
    for (; k < gy; k++) m_code(m, 1, by + (k >> sh), 1, bits);
    if (gy < gmax) m_code(m, 1, by + (k >> sh), 0, bits);
    if (gx > 3) bits += (gx - 2) / 2;
    if (gy > 3) bits += (gy - 2) / 2;
    bits_out += bits;
  }
  double bits = 0;
  uint8_t ctx_offset[16];
  int temp_diag = -1, temp_sum = -1;
  int32_t reg_bins = (n * n * 28) >> 4;
  for (int i = scan_cg_last; i >= 0; i--) {
    const int first = scan[i * 16];
    const int cg_pos_x = (first & (n - 1)) >> 2, cg_pos_y = (first >> l2) >> 2;
    const int cg_blk = cg_pos_y * cgw + cg_pos_x;
    if (i == scan_cg_last || i == 0) {
      sig_cg[cg_blk] = 1;
    } else {
      unsigned right = 0, lower = 0;
      if (cg_pos_x + 1 < cgw) right = sig_cg[cg_blk + 1];
      if (cg_pos_y + 1 < cgw) lower = sig_cg[cg_blk + cgw];
      m_code(m, 1, M_SIGGRP + 2 * t + ((right || lower) ? 1 : 0), sig_cg[cg_blk] != 0, bits);
    }
    if (!sig_cg[cg_blk]) continue;
    const int min_sub_pos = i << 4;
    const int first_sig_pos = (i == scan_cg_last) ? scan_pos_last : (min_sub_pos + 15);
    int next_sig_pos = first_sig_pos;
    const int infer_sig_pos = (next_sig_pos != scan_pos_last) ? ((i != 0) ? min_sub_pos : -1) : next_sig_pos;
    int num_non_zero = 0;
    for (next_sig_pos = first_sig_pos; next_sig_pos >= min_sub_pos && reg_bins >= 4; next_sig_pos--) {
      const int blk = scan[next_sig_pos];
      const int py = blk >> l2, px = blk - (py << l2);
      const int sig = coeff[blk] != 0;
      if (num_non_zero || next_sig_pos != infer_sig_pos) {
        const int ctx_sig = sig_ctx_abs(coeff, px, py, n, color, &temp_diag, &temp_sum);
        m_code(m, 1, M_SIG + 12 * t + (t ? (ctx_sig < 7 ? ctx_sig : 7) : ctx_sig), sig, bits);
        reg_bins--;
      } else if (next_sig_pos != scan_pos_last) {
        (void)sig_ctx_abs(coeff, px, py, n, color, &temp_diag, &temp_sum);
      }
      if (sig) {
        uint8_t *offset = &ctx_offset[next_sig_pos - min_sub_pos];
        num_non_zero++;
        *offset = 0;
        if (temp_diag != -1) {
          *offset = (uint8_t)((temp_sum < 4 ? temp_sum : 4) + 1);
          *offset = (uint8_t)(*offset + (!temp_diag ? (color == 0 ? 15 : 5) : color == 0 ? (temp_diag < 3 ? 10 : (temp_diag < 10 ? 5 : 0)) : 0));
        }
        int rem = iabs_((int)coeff[blk]) - 1;
        const int gt1 = rem ? 1 : 0;
        m_code(m, 1, M_GT1 + 21 * t + *offset, gt1, bits);
        reg_bins--;
        if (gt1) {
          rem -= 1;
          m_code(m, 1, M_PAR + 21 * t + *offset, rem & 1, bits);
          rem >>= 1;
          reg_bins--;
          m_code(m, 1, M_GT2 + 21 * t + *offset, rem ? 1 : 0, bits);
          reg_bins--;
        }
      }
    }
    for (int sp = first_sig_pos; sp > next_sig_pos; sp--) {
      const int blk = scan[sp];
      const unsigned a = (unsigned)iabs_((int)coeff[blk]);
      if (a >= 4) {
        const int py = blk >> l2, px = blk - (py << l2);
        const int rice = go_rice_par((unsigned)abs_sum_tmpl(coeff, px, py, n, 4));
        bits += coeff_remain_bits((a - 4) >> 1, (uint32_t)rice, 5);
      }
    }
    for (int sp = next_sig_pos; sp >= min_sub_pos; sp--) {
      const int blk = scan[sp];
      const int py = blk >> l2, px = blk - (py << l2);
      const unsigned a = (unsigned)iabs_((int)coeff[blk]);
      const int rice = go_rice_par((unsigned)abs_sum_tmpl(coeff, px, py, n, 0));
      const unsigned pos0 = 1u << rice;
      const unsigned remainder = a == 0 ? pos0 : (a <= pos0 ? a - 1 : a);
      bits += coeff_remain_bits(remainder, (uint32_t)rice, 5);
      if (a) num_non_zero++;
    }
    bits += num_non_zero;
  }
  return bits_out + bits;
}


// =================================================================================================== the CTU search ======
CTU_DEV int co_off(int color) { return color == 0 ? 0 : (color == 1 ? 4096 : 5120); }
// where the depth-L candidate's samples are kept: LDS, or (slim build, depth 1) the workgroup's global scratch
template <typename PX> CTU_DEV PX *cand_px_of(lds<PX> *S, const job<PX> &J, int L, int color);
CTU_DEV int cand_px_off(int L, int color)      // L = 1..3
{
  const int base = L == 1 ? 0 : (L == 2 ? 1536 : 1920);
  const int n = 64 >> L;
  return base + (color == 0 ? 0 : n * n + (color - 1) * (n / 2) * (n / 2));
}

template <typename PX> CTU_DEV PX *cand_px_of(lds<PX> *S, const job<PX> &J, int L, int color)
{
  if (lds_cfg<PX>::slim) return L == 1 ? (PX *)J.W->cand_px32 + cand_px_off(1, color) : S->cand_px + cand_px_off(L, color) - 1536;
  return S->cand_px + cand_px_off(L, color);
}

template <typename PX> CTU_DEV int scaled_qp(const params &P, int color) { return (color == 0 ? P.qp : P.qp_c) + 6 * ((int)px_info<PX>::depth - 8); }

// predict + uvg_quantize_residual (quant-generic.c:460-612, RDOQ branch) of one transform block straight into D; its levels stay in
// lv_of(V, color) and go to the CTU's coefficient array.  (x, y) / (lx, ly): luma position, n: luma size of the area.  -> has_coeffs
// CTU_PB, flags: 1 the prediction is already in dst (a block of an inter CU), 2 no reconstruction (the early-skip test only wants
// has_coeffs), 4 RDOQ prices "no luma coefficients" with the root cbf, 8 the levels are in co already (reconstruction only)
template <typename PX> CTU_NOINLINE CTU_DEV void rdoq_wave_ool(lds<PX> *S, scratch *W, const int16_t *coef_, int16_t *dst_, int n, int color, int cbf_u, int qp_scaled,
                                                           double lambda, int bitdepth)
{
  rdoq_wave(S, W, coef_, dst_, n, color, cbf_u, qp_scaled, lambda, bitdepth);
}
template <typename PX, bool OOL = false> CTU_INLINE1 CTU_DEV int recon_tu_inl(lds<PX> *S, const job<PX> &J, int color, int x, int y, int lx, int ly, int n, int mode, int cbf_u,
                                                            PX *dst_, int dp, int16_t *co, int cp, int cu_n
#if defined(CTU_PB)
                                                            , int flags = 0
#endif
                                                            )
{
  // dst / dp: where the block is reconstructed (the decided planes, or the depth's candidate buffer); co / cp: where its levels go
  wctx *const V = wv_of(S);
  typename mg_ptr<PX, PX>::type const dst = MGP(PX, PX, dst_);
  CTU_LDS int16_t *const t0 = LDSP(int16_t, V->t0);
  typename mg_ptr<PX, int16_t>::type const lv = MGP(PX, int16_t, lv_of(V, color));
  const int c = color != 0, w = n >> c, l2 = ilog2_dev(w);
  int sps;
  CTU_GLB const PX *Sp = src_block(J, color, lx >> c, ly >> c, &sps);
  const int depth = (int)px_info<PX>::depth;
#if defined(CTU_PB)
  LANE0 V->rq_root = (flags & 4) && color == 0;
  if (!(flags & 1))
#endif
  { CTU_T0();
  build_refs(S, J.P, color, x, y, lx, ly, n, cu_n);
  predict_block(S, mode, color, w, dst_, dp);
  CTU_T1(J.W, 1); }
  const int qps = scaled_qp<PX>(J.P, color);
#if defined(CTU_PB)
  if (flags & 8) {
    // the block's levels are in co already: an earlier call quantised this very residual (same prediction, same parameters -- the
    // early-skip test of the merge candidate that became the CU, ctu_pb.h finish_inter); only the reconstruction is left to do
    int nz = 0;
    PAR_FOR(e, w * w) { const int r = e >> l2, q = e & (w - 1); const int16_t v = co[r * cp + q]; lv[e] = v; nz |= v != 0; }
#if defined(__HIPCC__)
    nz = __ballot(nz) != 0;
#endif
    LANE0 V->rq_i[1] = nz;
    CTU_SYNC();
  } else
#endif
  {
  { CTU_T0();
  PAR_FOR(e, w * w) { const int r = e >> l2, q = e & (w - 1); t0[e] = (int16_t)((int)Sp[r * sps + q] - (int)dst[r * dp + q]); }
  CTU_SYNC();
  fwd_pass(w, V->t0, V->t1, l2 - 1 + depth - 8);
  fwd_pass(w, V->t1, V->t0, l2 + 6);                     // (two buffers: the coefficients end up where the residual was)
  CTU_T1(J.W, 2); }
  CTU_T0();
  {
    // chroma blocks: state->c_lambda as uvg_quantize_lcu_residual replaces it (transform.c:1575)
    const double lambda = c ? J.P.c_lambda_tu : J.P.lambda;
    if constexpr (OOL) rdoq_wave_ool(S, J.W, V->t0, lv_of(V, color), w, color, cbf_u, qps, lambda, depth);
    else rdoq_wave(S, J.W, V->t0, lv_of(V, color), w, color, cbf_u, qps, lambda, depth);
  }
  CTU_SYNC();
  CTU_T1(J.W, 3);
  }
  const int has = V->rq_i[1];
  PAR_FOR(e, w * w) { const int r = e >> l2, q = e & (w - 1); co[r * cp + q] = lv[e]; }
#if defined(CTU_PB)
  if (has && !(flags & 2)) {
#else
  if (has) {
#endif
    const int transform_shift = 15 - depth - l2;
    const int shift = 20 - 14 - transform_shift;
    const int32_t scale = (int32_t)kInvQuantScales[qps % 6] << (qps / 6);
    const int32_t add = 1 << (shift - 1);
    PAR_FOR(e, w * w) t0[e] = (int16_t)clampi((lv[e] * scale + add) >> shift, -32768, 32767);      // uvg_dequant, quant-generic.c:618-669
    CTU_SYNC();
    inv_pass(w, V->t0, V->t1, 7);
    inv_pass(w, V->t1, V->t0, 12 - (depth - 8));
    PAR_FOR(e, w * w) {
      const int r = e >> l2, q = e & (w - 1);
      const int16_t val = (int16_t)(t0[e] + (int)dst[r * dp + q]);
      dst[r * dp + q] = (PX)clampi(val, 0, (int)px_info<PX>::maxv);
    }
  }
  CTU_SYNC();
  return has;
}
// Called per block, a noinline function saves ~45 callee-saved VGPRs to the stack and reloads them (256 B each): with RDOQ and the
// rough search, 20 MB of stack traffic per CTU.  The CU evaluation therefore has ONE call site (a loop over the colours) with the body
// inlined; the P / B walk's blocks go through this out-of-line copy (the I pictures' 64x64 candidate has its own: recon_tu64).
template <typename PX> CTU_NOINLINE CTU_DEV int recon_tu(lds<PX> *S, const job<PX> &J, int color, int x, int y, int lx, int ly, int n, int mode, int cbf_u,
                                                         PX *dst_, int dp, int16_t *co, int cp, int cu_n
#if defined(CTU_PB)
                                                         , int flags = 0
#endif
                                                         )
{
#if defined(CTU_PB)
  return recon_tu_inl(S, J, color, x, y, lx, ly, n, mode, cbf_u, dst_, dp, co, cp, cu_n, flags);
#else
  return recon_tu_inl<PX, true>(S, J, color, x, y, lx, ly, n, mode, cbf_u, dst_, dp, co, cp, cu_n);      // (RDOQ through the copy it shares with recon_tu64)
#endif
}

// uvg_pixels_calc_ssd of a w x w block of D against the source, into V->red[slot] (valid after the barrier)
template <typename PX> CTU_NOINLINE CTU_DEV void ssd_block(lds<PX> *S, const job<PX> &J, int color, int lx, int ly, int n, int slot, const PX *rec_, int rp)
{
  wctx *const V = wv_of(S);
  typename mg_ptr<PX, const PX>::type const rec = MGP(PX, const PX, rec_);
  const int c = color != 0, w = n >> c, l2 = ilog2_dev(w);
  int sps;
  CTU_GLB const PX *Sp = src_block(J, color, lx >> c, ly >> c, &sps);
  int acc = 0;
  PAR_FOR(e, w * w) { const int r = e >> l2, q = e & (w - 1); const int d = (int)Sp[r * sps + q] - (int)rec[r * rp + q]; acc += d * d; }
#if defined(__HIPCC__)
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
  LANE0 V->red[slot] = acc >> (2 * ((int)px_info<PX>::depth - 8));
#else
  V->red[slot] = acc >> (2 * ((int)px_info<PX>::depth - 8));
#endif
  CTU_SYNC();
}

// uvg_write_split_flag (encode_coding_tree.c:1240-1363) with the multi-type splits off: only split_cu_flag exists; lane 0
template <typename PX> CTU_DEV void split_flag_bits(lds<PX> *S, const params &P, uint32_t *m_, int update, int x, int y, int lx, int ly, int n, int split,
                                                    double &bits)
{
  CTU_LDS uint32_t *const m = LDSP(uint32_t, m_);
  const int inside = P.pic_w >= x + n && P.pic_h >= y + n;
  if (!inside || n <= 4) return;                      // implicit split, or nothing to split: no flag
  const cu4 *left = x > 0 ? cu_at(S, lx - 1, ly) : nullptr, *above = y > 0 ? cu_at(S, lx, ly - 1) : nullptr;
  int model = 0;
  if (left && (1 << left->log2) < n) model++;
  if (above && (1 << above->log2) < n) model++;
  m_code(m, update, M_SPLIT + model, split, bits);   // split_num = 1 -> + 3 * 0
}

// uvg_encode_intra_luma_coding_unit (encode_coding_tree.c:992-1238) in count mode; lane 0
template <typename PX> CTU_NOINLINE CTU_DEV void luma_mode_bits(lds<PX> *S, uint32_t *m_, int update, int x, int y, int lx, int ly, int n, int mode, double &bits_out)
{
  CTU_LDS uint32_t *const m = LDSP(uint32_t, m_);
  const cu4 *l, *a;
  int8_t preds[6];
  mpm_neighbours(S, x, y, lx, ly, n, &l, &a);
  dir_luma_predictor(y, preds, l, a);
  double bits = 0;
  int mpm = -1;
  for (int i = 0; i < 6; ++i) if (preds[i] == mode) { mpm = i; break; }
  m_code(m, update, M_MPM, mpm != -1, bits);
  if (mpm != -1) {
    m_code(m, update, M_PLANAR + 1, mpm > 0, bits);
    if (mpm > 0) bits += 1;
    if (mpm > 1) bits += 1;
    if (mpm > 2) bits += 1;
    if (mpm > 3) bits += 1;
  } else {
    // the reference sorts the list and steps the mode down past every smaller entry (:1197-1215): mode - #{entries < mode}
    int tmp = mode;
    for (int i = 0; i < 6; ++i) tmp -= preds[i] < mode;
    bits_out += (tmp < 3) ? 5 : 6;                    // truncated binary code of 61 symbols (cabac.c:203-229)
  }
  bits_out += bits;
}
CTU_DEV void chroma_mode_bits(uint32_t *m_, int update, int chroma_mode, int luma_dir, double &bits)     // encode_chroma_intra_cu, :902-990
{
  CTU_LDS uint32_t *const m = LDSP(uint32_t, m_);
  const int derived = chroma_mode == luma_dir;
  m_code(m, update, M_CHROMA_PRED, derived ? 0 : 1, bits);
  if (!derived) bits += 2;
}

// mark_deblocking (search.c:1075-1174) for an n x n CU; sep: a 4x4 CU of an 8x8 area, chroma: it carries the area's chroma; lane 0
template <typename PX> CTU_NOINLINE CTU_DEV void mark_deblocking(lds<PX> *S, int x, int y, int lx, int ly, int n, int sep, int chroma)
{
  if (x) {
    for (int xx = lx; xx < lx + n; xx += 32)
      for (int yy = ly; yy < ly + n; yy += 4) { cu_at(S, xx, yy)->luma_edges |= 1; if (!sep) cu_at(S, xx, yy)->chroma_edges |= 1; }
  } else if (n == 64) {
    for (int yy = ly; yy < ly + n; yy += 4) { cu_at(S, 32, yy)->luma_edges |= 1; if (!sep) cu_at(S, 32, yy)->chroma_edges |= 1; }
  }
  if (y) {
    for (int yy = ly; yy < ly + n; yy += 32)
      for (int xx = lx; xx < lx + n; xx += 4) { cu_at(S, xx, yy)->luma_edges |= 2; if (!sep) cu_at(S, xx, yy)->chroma_edges |= 2; }
  } else if (n == 64) {
    for (int xx = lx; xx < lx + n; xx += 4) { cu_at(S, xx, 32)->luma_edges |= 2; if (!sep) cu_at(S, xx, 32)->chroma_edges |= 2; }
  }
  if (sep && chroma) {
    const int cx = lx & ~7, cy = ly & ~7;
    if (x & ~7) for (int yy = cy; yy < cy + 8; yy += 4) cu_at(S, cx, yy)->chroma_edges |= 1;
    if (y & ~7) for (int xx = cx; xx < cx + 8; xx += 4) cu_at(S, xx, cy)->chroma_edges |= 2;
  }
}

// regular bins a scan position spends, from its record (level in bits 0..15, "its sig flag is coded" in bit 29)
CTU_DEV int rec_spend(uint32_t rec) { const uint32_t a = rec & 0xffffu; return (int)((rec >> 29) & 1u) + (a ? 1 + (a > 1 ? 2 : 0) : 0); }

#if defined(CTU_LEAF4)
#include "ctu_leaf4.h"
#endif

#if defined(__HIPCC__)
// coeff_bits for a 4x4 block (one coefficient group): the shape the 4x4 leaves, their 8x8 areas' chroma and most of the coder
// pass consist of.  Same bins and adaptation as the general function below, but nothing goes through memory: lane sp (0..15) holds
// position sp's record, the budget of regular bins is a lane sum, the models' sweep (one: 12 + 3 * 16 luma / 8 + 3 * 11 chroma
// models) hands the records out with v_readlane and runs branch-free.
template <typename PX> CTU_DEV double coeff_bits4(lds<PX> *S, CTU_LDS uint32_t *m, int update, CTU_LDS const int16_t *coeff, int color)
{
  const int lane = CTU_TID, sp = lane & 15, t = color ? 1 : 0;
  const uint16_t *scan = scan_of(S, 2);
  const int blk = scan[sp], py = blk >> 2, px = blk & 3;
  const int a = iabs_((int)coeff[blk]);
  const unsigned nzmask = (unsigned)__ballot(a != 0) & 0xffffu;
  if (nzmask == 0) return 0.0;
  const int last = 31 - __clz((int)nzmask);
  int diag, tsum;
  int ctx_sig = sig_ctx_abs(coeff, px, py, 4, color, &diag, &tsum);
  if (t && ctx_sig > 7) ctx_sig = 7;
  int ofs = 0;
  if (sp != last) ofs = ((tsum < 4 ? tsum : 4) + 1) + (!diag ? (color == 0 ? 15 : 5) : color == 0 ? (diag < 3 ? 10 : (diag < 10 ? 5 : 0)) : 0);
  const int r4 = go_rice_par((unsigned)abs_sum_tmpl(coeff, px, py, 4, 4)), r0 = go_rice_par((unsigned)abs_sum_tmpl(coeff, px, py, 4, 0));
  const bool live = sp <= last;
  const int sig_coded = live && sp != last;
  const int spend = live ? sig_coded + (a ? 1 + (a > 1 ? 2 : 0) : 0) : 0;
  const uint32_t rec = (uint32_t)(a > 0xffff ? 0xffff : a) | (uint32_t)ctx_sig << 16 | (uint32_t)ofs << 20 | (uint32_t)sig_coded << 29;
  // where the regular-bin budget (28 for 16 coefficients) runs out: positions <= sw are bypass-coded
  int tot = spend;
  for (int o = 8; o >= 1; o >>= 1) tot += __shfl_xor(tot, o, 64);
  int sw = -1;
  if (28 - tot < 4) {
    int rb = 28;
    for (int j = last; j >= 0; --j) {
      if (rb < 4) { sw = j; break; }
      rb -= __builtin_amdgcn_readlane(spend, j);
    }
  }
  // ---- the models, one per lane ----
  const int nk0 = t ? 8 : 12, nks = t ? 11 : 16;
  int role = -1, k = 0;
  if (lane < nk0) { role = 0; k = lane; }
  else if (lane < nk0 + 3 * nks) { role = 1 + (lane - nk0) / nks; k = (lane - nk0) % nks; }
  if (!t && role > 0 && k > 0) k += 5;          // 4x4 luma: set offsets 0, 6..20
  const int model = role < 0 ? 0 : (role == 0 ? M_SIG + 12 * t : role == 1 ? M_GT1 + 21 * t : role == 2 ? M_PAR + 21 * t : M_GT2 + 21 * t) + k;
  uint32_t st = m[model];
  const int rw = kRate[model], r0w = rw >> 4, r1w = rw & 15;
  const uint32_t add0 = (0x7fffu >> r0w) & 0x7fe0u, add1 = (0x7fffu >> r1w) & 0x7ffeu;
  const uint32_t fsh = role == 0 ? 16 : 20, fmask = role == 0 ? 15u : 31u, rsel = role < 0 ? 31u : (uint32_t)role;
  CTU_LDS const uint32_t *const ebits = LDSP(const uint32_t, tab_ebits());
  uint32_t acc = 0;
  for (int j = last; j > sw; --j) {
    const uint32_t rj = (uint32_t)__builtin_amdgcn_readlane((int)rec, j);
    const uint32_t aj = rj & 0xffffu;
    // per role: does the position code a bin with one of the role's models, and which   (sig, gt1, parity, gt2)
    const uint32_t gates = ((rj >> 29) & 1u) | (aj != 0 ? 2u : 0u) | (aj > 1 ? 12u : 0u);
    if (gates == 0) continue;
    const uint32_t bins = (aj != 0 ? 1u : 0u) | (aj > 1 ? 2u : 0u) | ((aj & 1u) << 2) | (aj >= 4 ? 8u : 0u);
    const bool hit = ((gates >> rsel) & 1u) && ((rj >> fsh) & fmask) == (uint32_t)k;
    const uint32_t bin = (bins >> rsel) & 1u;
    uint32_t s0 = st & 0xffffu, s1 = st >> 16;
    const uint32_t cost = ebits[(((s0 + s1) >> 8) << 1) ^ bin];
    s0 -= (s0 >> r0w) & 0x7fe0u;
    s1 -= (s1 >> r1w) & 0x7ffeu;
    s0 += bin ? add0 : 0u;
    s1 += bin ? add1 : 0u;
    st = hit ? ((s0 & 0xffffu) | (s1 << 16)) : st;
    acc += hit ? cost : 0u;
  }
  if (role >= 0 && update) m[model] = st;
  unsigned long long q15 = acc;
  // ---- bypass-coded parts: remainders, bypass positions, signs ----
  int ibits = 0;
  if (lane < 16 && live) {
    if (sp > sw) { if (a >= 4) ibits += coeff_remain_bits(((unsigned)a - 4) >> 1, (uint32_t)r4, 5); }
    else {
      const unsigned pos0 = 1u << r0;
      ibits += coeff_remain_bits(a == 0 ? pos0 : ((unsigned)a <= pos0 ? (unsigned)a - 1 : (unsigned)a), (uint32_t)r0, 5);
    }
    ibits += a != 0;
  }
  // ---- lane 0: the last-position prefix (its models are nobody else's; counting only: on a copy) ----
  if (lane == 0) {
    double bits = 0;
    CTU_LDS uint32_t *mk = m;
    if (!update) {
#if defined(CTU_PB)
      mk = LDSP(uint32_t, CTU_WAVE == 0 ? S->pb.cnt_models : (CTU_WAVE == 1 ? S->pb.work0 : pbq(S).cnt_models));      // (the leaf wave counts on the 64x64 candidate's set: idle once the walk is below depth 0)
#else
      mk = LDSP(uint32_t, S->work[2]);
#endif
      for (int i = M_LASTX; i < M_CBF_LUMA; ++i) mk[i] = m[i];
    }
    const int pos_last = scan[last], last_y = pos_last >> 2, last_x = pos_last & 3;
    const int bx = M_LASTX + 20 * t, by = M_LASTY + 20 * t;      // 4x4: prefix offset 0, shift 0, three prefix models per axis
    for (int q = 0; q < last_x; q++) m_code(mk, 1, bx + q, 1, bits);
    if (last_x < 3) m_code(mk, 1, bx + last_x, 0, bits);
    for (int q = 0; q < last_y; q++) m_code(mk, 1, by + q, 1, bits);
    if (last_y < 3) m_code(mk, 1, by + last_y, 0, bits);
    q15 += (unsigned long long)(bits * 32768.0);
  }
  for (int o = 32; o >= 1; o >>= 1) {
    q15 += __shfl_xor(q15, o, 64);
    ibits += __shfl_xor(ibits, o, 64);
  }
  WSYNC();
  return (double)q15 / 32768.0 + (double)ibits;
}
#endif

// Coefficient bit cost by the first wave (same bins, same model adaptation as coeff_bits_serial).  What is sequential in the
// reference's coder is only the adaptation of each context model along ITS OWN bins, and the bit count is a sum of exact
// multiples of 2^-15 (order-free).  So: every position's contexts / Rice parameters / bin values are derived by all lanes from the
// levels (they depend on nothing else), the point where the regular-bin budget runs out is found from per-group totals, and then
// each lane owns ONE model and walks the positions in coding order, adapting its model at the bins that use it -- two sweeps
// (sig / gt1 / parity, then gt2) cover the <= 75 models of a block.  Lane 0 codes the last-position prefix and the group flags.
// All 64 lanes of wave 0 call it; the returned value is the same on every lane.  Host emulation: the serial walk.
template <typename PX> CTU_NOINLINE CTU_DEV double coeff_bits(lds<PX> *S, uint32_t *m_, int update, const int16_t *coeff_, int n, int color)
{
  wctx *const V = wv_of(S);
#if !defined(__HIPCC__)
  uint32_t *const m = m_; const int16_t *const coeff = coeff_;
  uint32_t tmp[NMODELS];
  uint32_t *mm = m;
  if (!update) { for (int i = 0; i < NMODELS; ++i) tmp[i] = m[i]; mm = tmp; }
  return coeff_bits_serial(mm, scan_of(S, ilog2_dev(n)), coeff, n, color);
#else
  const int lane = CTU_TID;
  const int l2 = ilog2_dev(n), nn = n * n, cgw = n >> 2, ncg = nn >> 4, t = color ? 1 : 0;
  const uint16_t *scan = scan_of(S, l2);
  CTU_LDS uint32_t *const m = (CTU_LDS uint32_t *)m_;
  typename mg_ptr<PX, const int16_t>::type const coeff = MGP(PX, const int16_t, coeff_);
#if defined(CTU_LEAF4)
  if (n == 4) return coeff_bits4r(S, m, update, (int)coeff[lane & 15], color);
#else
  if (n == 4) return coeff_bits4(S, m, update, coeff, color);
#endif
  CTU_LDS uint32_t *recs = (CTU_LDS uint32_t *)(V->t0);       // t0 + t1: 1024 words, free while costs are counted
  CTU_LDS uint8_t *cgf = (CTU_LDS uint8_t *)V->cg_flag;                                   // per group (raster): has a level
  CTU_LDS int32_t *gtot = (CTU_LDS int32_t *)(V->rq_stage);   // per group (scan order): regular bins it would spend, bit 30: a level among k = 1..15
#if defined(CTU_PROFILE)
  scratch *const W = S->prof_w;
  unsigned long long tq = __builtin_amdgcn_s_memtime();
#endif
  // ---- last significant position, group flags ----
  int my_last = -1;
  for (int sp = lane; sp < nn; sp += 64) if (coeff[scan[sp]]) my_last = sp;
  for (int o = 32; o >= 1; o >>= 1) { const int v = __shfl_xor(my_last, o, 64); my_last = v > my_last ? v : my_last; }
  const int last = my_last;
  if (last < 0) return 0.0;
  const int cg_last = last >> 4;
  for (int g = lane; g < ncg; g += 64) {
    int any0 = coeff[scan[g * 16]] != 0, anyr = 0;
    for (int k = 1; k < 16; ++k) anyr |= coeff[scan[g * 16 + k]] != 0;
    const int f = scan[g * 16];
    cgf[((f >> l2) >> 2) * cgw + ((f & (n - 1)) >> 2)] = (uint8_t)(any0 | anyr);
    gtot[g] = anyr << 30;
  }
  WSYNC();
  RQ_T(28);
  // ---- per position: level, contexts, Rice parameters, whether its sig flag is coded, the regular bins it would spend ----
  for (int sp = lane; sp <= last; sp += 64) {
    const int blk = scan[sp], py = blk >> l2, px = blk - (py << l2), g = sp >> 4;
    const int a = iabs_((int)coeff[blk]);
    int diag, tsum;
    int ctx_sig = sig_ctx_abs(coeff, px, py, n, color, &diag, &tsum);
    if (t && ctx_sig > 7) ctx_sig = 7;
    int ofs = 0;
    if (sp != last) ofs = ((tsum < 4 ? tsum : 4) + 1) + (!diag ? (color == 0 ? 15 : 5) : color == 0 ? (diag < 3 ? 10 : (diag < 10 ? 5 : 0)) : 0);
    const int r4 = go_rice_par((unsigned)abs_sum_tmpl(coeff, px, py, n, 4)), r0 = go_rice_par((unsigned)abs_sum_tmpl(coeff, px, py, n, 0));
    const int inferred = (sp & 15) == 0 && g != 0 && g != cg_last && !((gtot[g] >> 30) & 1);
    const int sig_coded = sp != last && !inferred;
    const int spend = sig_coded + (a ? 1 + (a > 1 ? 2 : 0) : 0);
    recs[sp] = (uint32_t)(a > 0xffff ? 0xffff : a) | (uint32_t)ctx_sig << 16 | (uint32_t)ofs << 20 | (uint32_t)r4 << 25 | (uint32_t)r0 << 27 |
               (uint32_t)sig_coded << 29;
  }
  WSYNC();
  for (int g = lane; g <= cg_last; g += 64) {
    int tot = 0;
    for (int k = 0; k < 16; ++k) if (g * 16 + k <= last) tot += rec_spend(recs[g * 16 + k]);
    gtot[g] = (gtot[g] & (1 << 30)) | tot;
  }
  WSYNC();
  // ---- where the regular-bin budget runs out (lane 0; whole groups while they fit) ----
  if (lane == 0) {
    int rb = (nn * 28) >> 4, sw = -1;
    for (int g = cg_last; g >= 0 && sw < 0; --g) {
      const int f = scan[g * 16];
      const int sig_grp = cgf[((f >> l2) >> 2) * cgw + ((f & (n - 1)) >> 2)] || g == 0;
      if (!sig_grp) continue;
      const int tot = gtot[g] & 0xffff;
      if (rb - tot >= 4) { rb -= tot; continue; }
      for (int sp = (g == cg_last ? last : g * 16 + 15); sp >= g * 16; --sp) {
        if (rb < 4) { sw = sp; break; }
        rb -= rec_spend(recs[sp]);
      }
      if (sw < 0 && rb < 4) sw = g * 16 - 1;          // ran out exactly at the group's end: everything below is bypass-coded
    }
    V->rq_i[8] = sw;
  }
  WSYNC();
  const int sw = V->rq_i[8];          // scan positions <= sw are bypass-coded
  RQ_T(29);
  // ---- the models, one per lane, along the positions in coding order ----
  // A group's 16 records are fetched by lanes 0..15 at once and handed out with v_readlane: no memory in the adaptation chain.
  // 4x4 luma blocks use 16 of the 21 greater-1 / parity / greater-2 models (the diagonal classes of larger blocks never occur) and
  // chroma has 8 + 3 * 11 models: one sweep covers them; other luma blocks take two (sig / gt1 / parity, then gt2).
  unsigned long long q15 = 0;         // sum of bit costs in units of 2^-15
  unsigned long long grp_mask = 0;    // bit g: group g (scan order) is coded
  for (int g0 = 0; g0 <= cg_last; g0 += 64) {
    const int g = g0 + lane;
    int on = 0;
    if (g <= cg_last) { const int f = scan[g * 16]; on = cgf[((f >> l2) >> 2) * cgw + ((f & (n - 1)) >> 2)] || g == 0; }
    grp_mask = __ballot(on);            // (ncg <= 64)
  }
  const int compact = !t && n == 4;     // 4x4 luma: set offsets 0, 6..20 -> k 0, 1..15
  const int nk0 = t ? 8 : 12, nks = t ? 11 : (compact ? 16 : 21);
  // 8 + 3 * 11 chroma / 12 + 3 * 16 compact-luma models fit the 64 lanes; other luma blocks have 12 + 3 * 21 = 75: there the lanes 0..20
  // own a SECOND model (greater-2, set k = lane) beside their first -- one walk over the positions serves both (the two lookups of a
  // step are in flight together; two walks cost twice the steps)
  const bool dual = nk0 + 3 * nks > 64;
  {
    int role = -1, k = 0;
    if (!dual) {
      if (lane < nk0) { role = 0; k = lane; }
      else if (lane < nk0 + 3 * nks) { role = 1 + (lane - nk0) / nks; k = (lane - nk0) % nks; }
      if (compact && role > 0 && k > 0) k += 5;
    } else { if (lane < 12) { role = 0; k = lane; } else if (lane < 33) { role = 1; k = lane - 12; } else if (lane < 54) { role = 2; k = lane - 33; } }
    const bool two = dual && lane < 21;
    const int model = role < 0 ? 0 : (role == 0 ? M_SIG + 12 * t : role == 1 ? M_GT1 + 21 * t : role == 2 ? M_PAR + 21 * t : M_GT2 + 21 * t) + k;
    const int model2 = M_GT2 + 21 * t + (two ? lane : 0);
    uint32_t st = m[model], st2 = m[model2];
    const int r0 = kRate[model] >> 4, r1 = kRate[model] & 15, q0 = kRate[model2] >> 4, q1 = kRate[model2] & 15;
    const uint32_t add0 = (0x7fffu >> r0) & 0x7fe0u, add1 = (0x7fffu >> r1) & 0x7ffeu, bdd0 = (0x7fffu >> q0) & 0x7fe0u, bdd1 = (0x7fffu >> q1) & 0x7ffeu;
    // what a record must show for this lane's model to code a bin: field (sig: bits 16..19, others: 20..24) == k, and the gate
    const uint32_t fsh = role == 0 ? 16 : 20, fmask = role == 0 ? 15u : 31u, rsel = role < 0 ? 31u : (uint32_t)role;
    CTU_LDS const uint32_t *const ebits = LDSP(const uint32_t, tab_ebits());
    uint32_t acc = 0;
    for (int g = cg_last; g >= 0 && g * 16 + 15 > sw; --g) {
      if (!((grp_mask >> g) & 1)) continue;
      const uint32_t myrec = recs[g * 16 + (lane & 15)];
#pragma nounroll
      for (int j = 15; j >= 0; --j) {
        const int sp = g * 16 + j;
        if (sp > last || sp <= sw) continue;
        const uint32_t rj = (uint32_t)__builtin_amdgcn_readlane((int)myrec, j);
        const uint32_t aj = rj & 0xffffu;
        // per role (sig, gt1, parity, gt2): does the position code a bin with one of the role's models, and which -- branch-free
        const uint32_t gates = ((rj >> 29) & 1u) | (aj != 0 ? 2u : 0u) | (aj > 1 ? 12u : 0u);
        if (gates == 0) continue;
        const uint32_t bins = (aj != 0 ? 1u : 0u) | (aj > 1 ? 2u : 0u) | ((aj & 1u) << 2) | (aj >= 4 ? 8u : 0u);
        const bool hit = ((gates >> rsel) & 1u) && ((rj >> fsh) & fmask) == (uint32_t)k;
        const uint32_t bin = (bins >> rsel) & 1u;
        uint32_t s0 = st & 0xffffu, s1 = st >> 16;
        const uint32_t cost = ebits[(((s0 + s1) >> 8) << 1) ^ bin];
        s0 -= (s0 >> r0) & 0x7fe0u;
        s1 -= (s1 >> r1) & 0x7ffeu;
        s0 += bin ? add0 : 0u;
        s1 += bin ? add1 : 0u;
        st = hit ? ((s0 & 0xffffu) | (s1 << 16)) : st;
        acc += hit ? cost : 0u;
        if (dual && (gates & 8u)) {                 // (wave-uniform: the position codes a greater-2 bin)
          const bool hit2 = two && ((rj >> 20) & 31u) == (uint32_t)lane;
          const uint32_t bin2 = (bins >> 3) & 1u;
          uint32_t t0 = st2 & 0xffffu, t1 = st2 >> 16;
          const uint32_t cost2 = ebits[(((t0 + t1) >> 8) << 1) ^ bin2];
          t0 -= (t0 >> q0) & 0x7fe0u;
          t1 -= (t1 >> q1) & 0x7ffeu;
          t0 += bin2 ? bdd0 : 0u;
          t1 += bin2 ? bdd1 : 0u;
          st2 = hit2 ? ((t0 & 0xffffu) | (t1 << 16)) : st2;
          acc += hit2 ? cost2 : 0u;
        }
      }
    }
    if (role >= 0 && update) m[model] = st;
    if (two && update) m[model2] = st2;
    q15 += acc;
  }
  RQ_T(30);
  // ---- bypass-coded parts: remainders, bypass positions, signs ----
  int ibits = 0;
  for (int sp = lane; sp <= last; sp += 64) {
    const int g = sp >> 4, f = scan[g * 16];
    if (!(cgf[((f >> l2) >> 2) * cgw + ((f & (n - 1)) >> 2)] || g == 0)) continue;
    const uint32_t rec = recs[sp];
    const unsigned a = (unsigned)iabs_((int)coeff[scan[sp]]);
    if (sp > sw) { if (a >= 4) ibits += coeff_remain_bits((a - 4) >> 1, (rec >> 25) & 3, 5); }
    else {
      const unsigned rice = (rec >> 27) & 3, pos0 = 1u << rice;
      ibits += coeff_remain_bits(a == 0 ? pos0 : (a <= pos0 ? a - 1 : a), rice, 5);
    }
    ibits += a != 0;
  }
  // ---- lane 0: last-position prefix and the group flags (their models are nobody else's) ----
  if (lane == 0) {
    double bits = 0;
    CTU_LDS uint32_t *mk = m, *mg = m;        // the models of the last position / of the group flags
    if (!update) {
      // counting only: these bins still adapt their models WITHIN the block (the reference counts on a copy) -- work on a copy
      // of the few models involved
#if defined(CTU_PB)
      mk = (CTU_LDS uint32_t *)(CTU_WAVE == 0 ? S->pb.cnt_models : (CTU_WAVE == 1 ? S->pb.work0 : pbq(S).cnt_models));       // (every work[] set is some depth's here; the leaf wave: see coeff_bits)
      mg = mk;
#else
      // (the 64x64 candidate, counted beside the walk by two waves at a time while every work[] set is some depth's: the 84 models live in
      // the wave's reference rows -- a block is counted after its reconstruction -- behind the 4 group-flag models)
      mk = (CTU_LDS uint32_t *)V->top - (M_LASTX - 4);
      mg = (CTU_LDS uint32_t *)V->top;
      for (int i = 0; i < 4; ++i) mg[M_SIGGRP + i] = m[M_SIGGRP + i];
      for (int i = M_LASTX; i < M_CBF_LUMA; ++i) mk[i] = m[i];
#endif
#if defined(CTU_PB)
      for (int i = 0; i < 4; ++i) mk[M_SIGGRP + i] = m[M_SIGGRP + i];
      for (int i = M_LASTX; i < M_CBF_LUMA; ++i) mk[i] = m[i];
#endif
    }
    const int pos_last = scan[last];
    const int last_y = pos_last >> l2, last_x = pos_last - (last_y << l2);
    const int prefix_ctx[8] = {0, 0, 0, 3, 6, 10, 15, 21};
    const int off = t ? 0 : prefix_ctx[l2];
    const int sh = t ? clampi(n >> 3, 0, 2) : ((l2 + 1) >> 2);
    const int bx = M_LASTX + 20 * t + off, by = M_LASTY + 20 * t + off;
    const int gx = group_idx(last_x), gy = group_idx(last_y), gmax = group_idx(n - 1);
    int k = 0;
    for (; k < gx; k++) m_code(mk, 1, bx + (k >> sh), 1, bits);
    if (gx < gmax) m_code(mk, 1, bx + (k >> sh), 0, bits);
    k = 0;
    for (; k < gy; k++) m_code(mk, 1, by + (k >> sh), 1, bits);
    if (gy < gmax) m_code(mk, 1, by + (k >> sh), 0, bits);
    if (gx > 3) ibits += (gx - 2) / 2;
    if (gy > 3) ibits += (gy - 2) / 2;
    for (int g = cg_last - 1; g >= 1; --g) {
      const int f = scan[g * 16];
      const int cx = (f & (n - 1)) >> 2, cy = (f >> l2) >> 2, cb = cy * cgw + cx;
      unsigned right = 0, lower = 0;
      if (cx + 1 < cgw) right = cgf[cb + 1];
      if (cy + 1 < cgw) lower = cgf[cb + cgw];
      m_code(mg, 1, M_SIGGRP + 2 * t + ((right || lower) ? 1 : 0), cgf[cb] != 0, bits);
    }
    q15 += (unsigned long long)(bits * 32768.0);        // exact: a sum of table entries / 2^15
  }
  // with update == 0 lane 0's m_code calls above must not move the models either: m_code honours the flag
  for (int o = 32; o >= 1; o >>= 1) {
    q15 += __shfl_xor(q15, o, 64);
    ibits += __shfl_xor(ibits, o, 64);
  }
  WSYNC();
  RQ_T(31);
  return (double)q15 / 32768.0 + (double)ibits;
#endif
}

// the transform-tree part of the RD cost of one <= 32 block whose levels are in S->lv (cu_rd_cost_tr_split_accurate, search.c:724-986);
// red[0..2]: SSD of y, u, v.  lane 0.  update: state->search_cabac.update.
template <typename PX> CTU_NOINLINE CTU_DEV double tr_cost(lds<PX> *S, const params &P, int update, int n, int cbf, int has_chroma, int cn)
{
  wctx *const V = wv_of(S);
  // called by all lanes of the first wave; the flag bins are lane 0's, the coefficient costs the wave's
  double coeff_bits_ = 0, luma_bits = 0, chroma_bits = 0;
  const int cb_y = cbf & 1, cb_u = (cbf >> 1) & 1, cb_v = (cbf >> 2) & 1;
  LANE0 {
    CTU_LDS uint32_t *const m = LDSP(uint32_t, V->cur);
    if (has_chroma) {
      m_code(m, update, M_CBF_CB + 0, cb_u, chroma_bits);
      m_code(m, update, M_CBF_CR + cb_u, cb_v, chroma_bits);
    }
    m_code(m, update, M_CBF_LUMA + 0, cb_y, luma_bits);
  }
  WSYNC();
  const unsigned luma_ssd = (unsigned)V->red[0];
  // uvg_get_coeff_cost counts on a copy of the models that is kept only when update is set (rdo.c:322-356)
  if (cb_y) coeff_bits_ += coeff_bits(S, V->cur, update, lv_of(V, 0), n, 0);
  unsigned chroma_ssd = 0;
  if (has_chroma) {
    const unsigned ssd_u = (unsigned)((unsigned)V->red[1] * P.cw_u), ssd_v = (unsigned)((unsigned)V->red[2] * P.cw_v);
    chroma_ssd = ssd_u + ssd_v;
    chroma_bits += coeff_bits(S, V->cur, update, lv_of(V, 1), cn, 1);
    chroma_bits += coeff_bits(S, V->cur, update, lv_of(V, 2), cn, 2);
  }
  const double bits = luma_bits + coeff_bits_;
  return luma_ssd * 1.0 + chroma_ssd * 1.0 + (bits + chroma_bits) * P.lambda;
}

// cur_cu->mode_type_tree: the tree's bits plus the parent's type at the CU's own depth (search.c:1387-1388)
CTU_DEV uint32_t cu_mtt(uint32_t tree_mtt, int L) { return tree_mtt | ((tree_mtt >> ((L - 1 > 0 ? L - 1 : 0) * 2)) & 3u) << (L * 2); }

// lcu_fill_cu_info (search.c:314-353) for an n x n intra CU; lane 0
template <typename PX> CTU_NOINLINE CTU_DEV void fill_cu(lds<PX> *S, int lx, int ly, int n, int mode, int mode_chroma, int log2_c, uint32_t split_tree, uint32_t mtt)
{
  const int l2 = ilog2_dev(n);
  for (int yy = ly; yy < ly + n; yy += 4)
    for (int xx = lx; xx < lx + n; xx += 4) {
      cu4 *c = cu_at(S, xx, yy);
      c->type = CU_INTRA; c->log2 = (uint8_t)l2; c->log2_c = (uint8_t)log2_c; c->mode = (int8_t)mode; c->mode_chroma = (int8_t)mode_chroma;
#if defined(CTU_PB)
      { const int u = ((yy >> 2) + 1) * 17 + (xx >> 2) + 1; S->pb.mot[u].type = CU_INTRA; S->pb.fl[u][0] = 0; S->pb.fl[u][1] = 0; }   // (skipped / merged: what the neighbours' contexts read)
#endif
      S->scr->tree[(yy >> 2) * 16 + (xx >> 2)] = (uint16_t)split_tree;
      S->scr->mtt[(yy >> 2) * 16 + (xx >> 2)] = (uint16_t)mtt;
    }
}

#if defined(CTU_PB)
template <typename PX> CTU_DEV void pb_intra_flag_bits(lds<PX> *S, const job<PX> &J, uint32_t *m_, int update, int x, int y, int lx, int ly, int n, double &bits);
#endif
// search + reconstruction + RD cost of the n x n CU at depth L as ONE coding unit (the part of search_cu before the split loop,
// search.c:1395-1774), by the calling wave.  to_cand = 0 (a CU that cannot be split): straight into the decided planes / side
// information / coefficient array, on the walk's models.  to_cand = 1 (its split is tried as well, by another wave at the same
// time): into the depth's candidate buffers, on the depth's own copy of the entry models -- nothing another wave reads is touched.
// Cost / mode / cbf go to S->lvl[L].
#if defined(__HIPCC__)
CTU_DEV int mb_load(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
CTU_DEV void mb_store(int32_t *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
#endif

#if !defined(CTU_PB)
// ---- the chroma helper ------------------------------------------------------------------------------------------------------
// The fourth 4x4 CU of an 8x8 area carries the area's two 4x4 chroma blocks (search.c:355-400).  They depend on that CU's luma MODE
// only, not on its luma block, and the walk's chain of 4x4 CUs is what a CTU's time is made of: the Cb block goes to the wave of
// depth 3 -- idle by then, the area's own 8x8 evaluation was posted four CUs ago -- while the walk reconstructs the luma block; Cr
// follows on the walk (its cbf context wants Cb's flag).  Same arithmetic on the same inputs either way: when the wave is still
// busy the walk simply does all three blocks itself.
template <typename PX> CTU_DEV void help_run(lds<PX> *S, const job<PX> &J)          // depth 3's wave
{
#if defined(__HIPCC__)
  __builtin_amdgcn_s_setprio(3);                   // the walk waits for this block
#endif
  const int cx = S->help[0], cy = S->help[1], mode = S->help[2];
  const int lx = cx & 63, ly = cy & 63;
  PX *const ru = S->Du + ((ly >> 1) + 1) * PC + (lx >> 1) + 1;
  int16_t *const ku = J.coeff + 4096 + (ly >> 1) * LCU_C + (lx >> 1);
#if defined(CTU_LEAF4)
  const lf_block B = leaf_recon(S, J, wv_of(S), 1, mode, 0, cx, cy, lx, ly, 8, 0, ru, PC, ku, LCU_C);
  const int has = B.has;
  LANE0 S->help[4] = B.ssd;
#else
  const int has = recon_tu(S, J, 1, cx, cy, lx, ly, 8, mode, 0, ru, PC, ku, LCU_C, 8);
#endif
  {
    CTU_LDS const int16_t *const from = LDSP(const int16_t, wv_of(S)->lv1);          // the walk counts the levels' bits from ITS scratch
    CTU_LDS int16_t *const to = LDSP(int16_t, S->wv[0].lv1);
    PAR_FOR(e, 16) to[e] = from[e];
  }
  LANE0 S->help[3] = has;
  CTU_SYNC();
#if defined(__HIPCC__)
  __builtin_amdgcn_s_setprio(0);
#endif
}
// the walk: hand the Cb block of the area at (cx, cy) over if depth 3's wave has nothing to do
template <typename PX> CTU_DEV bool help_post(lds<PX> *S, const job<PX> &J, int cx, int cy, int mode)
{
#if defined(__HIPCC__)
  if (__builtin_amdgcn_readfirstlane(mb_load(&S->done[3]) == S->req[3]) == 0) return false;
  LANE0 { S->help[0] = cx; S->help[1] = cy; S->help[2] = mode; }
  CTU_SYNC();
  LANE0 mb_store(&S->hreq, S->hreq + 1);
  return true;
#else
  if (g_emul_lazy) return false;                   // (the host tests take both roads)
  S->help[0] = cx; S->help[1] = cy; S->help[2] = mode;
  const int me = g_emul_wave;
  g_emul_wave = 1;
  help_run(S, J);
  g_emul_wave = me;
  return true;
#endif
}
template <typename PX> CTU_DEV int help_wait(lds<PX> *S)
{
#if defined(__HIPCC__)
  while (mb_load(&S->hdone) != S->hreq) __builtin_amdgcn_s_sleep(1);
  CTU_SYNC();
#endif
  return S->help[3];
}
#endif

template <typename PX> CTU_NOINLINE CTU_DEV void eval_cu(lds<PX> *S, const job<PX> &J, int L, int to_cand
#if defined(CTU_PB)
                                                         , int forced_mode = -1      // P / B: the rough search already ran (ctu_pb.h); >= 0: its mode
#endif
                                                         )
{
  wctx *const V = wv_of(S);
  const params &P = J.P;
  level_state &N = S->lvl[L];
  const int n = 64 >> L, x = N.x, y = N.y, lx = x & 63, ly = y & 63;
  const int sep = n == 4;                                 // a 4x4 CU: its chroma belongs to the 8x8 area, carried by the fourth one
  const int has_chroma = N.has_chroma;
  if (to_cand) {
    LANE0 V->cur = S->work[L - 1];             // (the walk put the CU's entry models there before posting the request)
  } else {
    LANE0 {
      V->cur = S->cur;
      cu4 *c = cu_at(S, lx, ly);                           // the CU's own entry is reset (search.c:1371-1388)
      c->type = CU_NOTSET; c->cbf = 0; c->luma_edges = 0; c->chroma_edges = 0; c->mode = 0; c->mode_chroma = 0; c->log2 = (uint8_t)ilog2_dev(n);
      c->log2_c = (uint8_t)(sep ? 2 : ilog2_dev(n) - 1);
#if defined(CTU_PB)
      { const int u = ((ly >> 2) + 1) * 17 + (lx >> 2) + 1; S->pb.mot[u].type = CU_NOTSET; S->pb.fl[u][0] = 0; S->pb.fl[u][1] = 0; }
#endif
    }
  }
  CTU_SYNC();
#if defined(CTU_PB)
  if (forced_mode < 0)
#endif
  { CTU_T0();
  build_refs(S, P, 0, x, y, lx, ly, n, n);
  search_intra_rough(S, J, x, y, lx, ly, n);
  CTU_T1(J.W, 0); }
#if defined(CTU_PB)
  const int mode = forced_mode < 0 ? V->u_mode : forced_mode;
#else
  const int mode = V->u_mode;
#endif
  if (!to_cand) { SERIAL fill_cu(S, lx, ly, n, mode, mode, sep ? 2 : ilog2_dev(n) - 1, N.split_tree, cu_mtt(N.mode_type_tree, L)); CTU_SYNC(); }
#if defined(CTU_LEAF4X)
  if (n == 8 && !to_cand) leaf_load_area(S, J, lx, ly);          // (an 8x8 leaf: the walk's own wave)
#endif
  // where the three blocks are reconstructed and where their levels go
  int cn = n >> 1, cx = x, cy = y;                        // the chroma area (luma coordinates) and its block size
  if (sep) { cn = 4; cx = x & ~7; cy = y & ~7; }
  const int area = sep ? 8 : n;
  PX *ry, *ru, *rv;
  int16_t *ky, *ku, *kv;
  int rpy, rpc, kpy, kpc;
  if (to_cand) {
    ry = cand_px_of(S, J, L, 0); ru = cand_px_of(S, J, L, 1); rv = cand_px_of(S, J, L, 2);
    int16_t *const kb = J.W->cand_co;
    ky = kb + cand_px_off(L, 0); ku = kb + cand_px_off(L, 1); kv = kb + cand_px_off(L, 2);
    rpy = kpy = n; rpc = kpc = cn;
  } else {
    ry = S->Dy + (ly + 1) * PY + lx + 1;
    ru = S->Du + (((cy & 63) >> 1) + 1) * PC + ((cx & 63) >> 1) + 1; rv = S->Dv + (((cy & 63) >> 1) + 1) * PC + ((cx & 63) >> 1) + 1;
    ky = J.coeff + ly * LCU + lx;
    ku = J.coeff + 4096 + ((cy & 63) >> 1) * LCU_C + ((cx & 63) >> 1); kv = J.coeff + 5120 + ((cy & 63) >> 1) * LCU_C + ((cx & 63) >> 1);
    rpy = PY; rpc = PC; kpy = LCU; kpc = LCU_C;
  }
  int cbf = 0;
#if !defined(CTU_PB)
  const bool helped = sep && has_chroma && !to_cand && help_post(S, J, cx, cy, mode);
#if defined(__HIPCC__) && defined(CTU_PROFILE)
  if (sep && has_chroma && !to_cand) { LANE0 J.W->prof[1][helped ? 19 : 20] += 1; }
#endif
#else
  const bool helped = false;
#endif
#if defined(__HIPCC__)
#pragma nounroll
#endif
  for (int color = 0; color < (has_chroma ? 3 : 1); ++color) {
    const bool c = color != 0;
#if !defined(CTU_PB)
    if (helped && color == 1) {
#if defined(__HIPCC__) && defined(CTU_PROFILE)
      const unsigned long long tw = __builtin_amdgcn_s_memtime();
#endif
      cbf |= help_wait(S) << 1;
#if defined(__HIPCC__) && defined(CTU_PROFILE)
      LANE0 J.W->prof[1][21] += __builtin_amdgcn_s_memtime() - tw;
#endif
      continue;
    }
#endif
#if defined(CTU_LEAF4X)
    if (c && n == 8) {
      // the 4x4 chroma blocks of an 8x8 CU: the register-resident block of ctu_leaf4.h (the area's source samples are in S->lf_src)
      CTU_T0();
      const lf_block B = leaf_recon(S, J, V, color, mode, color == 2 ? (cbf >> 1) & 1 : 0, x, y, lx, ly, 8, 0, color == 1 ? ru : rv, rpc, color == 1 ? ku : kv, kpc);
      LANE0 V->red[color] = B.ssd;
      cbf |= B.has << color;
      CTU_T1(J.W, 3);
      continue;
    }
#endif
    const int has = recon_tu_inl(S, J, color, c ? cx : x, c ? cy : y, c ? cx & 63 : lx, c ? cy & 63 : ly, c ? area : n, mode, color == 2 ? (cbf >> 1) & 1 : 0,
                                 color == 0 ? ry : (color == 1 ? ru : rv), c ? rpc : rpy, color == 0 ? ky : (color == 1 ? ku : kv), c ? kpc : kpy, c ? area : n);
    cbf |= has << color;
  }
#if defined(CTU_LEAF4X)
  if (has_chroma && n != 8)
#else
  if (has_chroma)
#endif
  {
    { CTU_T0();
    ssd_block(S, J, 1, cx & 63, cy & 63, area, 1, ru, rpc);
    ssd_block(S, J, 2, cx & 63, cy & 63, area, 2, rv, rpc);
    CTU_T1(J.W, 4); }
  }
  { CTU_T0();
  ssd_block(S, J, 0, lx, ly, n, 0, ry, rpy);
  CTU_T1(J.W, 4); }
  CTU_T0();
  {
    double bits = 0;
    LANE0 {
      if (!to_cand) {
        cu4 *c = cu_at(S, lx, ly);
        c->cbf = (uint8_t)(cbf & 1);
        if (has_chroma) {
          if (!sep) c->cbf = (uint8_t)cbf;
          else {
            // the area's chroma flags sit at its first entry and are copied to all four (lcu_fill_chroma_cbfs), with the chroma mode and
            // chroma size of this, the last, CU (lcu_fill_chroma_cu_info, search.c:355-400)
            for (int k = 0; k < 4; ++k) {
              cu4 *q = cu_at(S, (cx & 63) + (k & 1) * 4, (cy & 63) + (k >> 1) * 4);
              q->cbf = (uint8_t)((q->cbf & 1) | (cbf & 6));
              q->mode_chroma = (int8_t)mode;
              q->log2_c = 2;
            }
          }
        }
      }
      // uvg_mock_encode_coding_unit with search_cabac.update = 1 (search.c:1700-1716)
      split_flag_bits(S, P, V->cur, 1, x, y, lx, ly, n, 0, bits);
#if defined(CTU_PB)
      pb_intra_flag_bits(S, J, V->cur, 1, x, y, lx, ly, n, bits);       // a P / B slice: skip flag 0, prediction mode "intra"
#endif
      luma_mode_bits(S, V->cur, 1, x, y, lx, ly, n, mode, bits);
      if (has_chroma) chroma_mode_bits(V->cur, 1, mode, mode, bits);
    }
    CTU_SYNC();
    const double trc = tr_cost(S, P, 1, n, cbf, has_chroma, cn);       // cu_rd_cost_tr_split_accurate (:1718)
    LANE0 {
      double cost = bits * P.lambda;
      cost += trc;
      if (!to_cand) mark_deblocking(S, x, y, lx, ly, n, sep, has_chroma);
      N.cost = cost; N.type = CU_INTRA; N.mode = mode; N.cbf = cbf;
    }
  }
  CTU_SYNC();
  CTU_T1(J.W, 5);
}

#if defined(CTU_LEAF4)
// eval_cu(S, J, 4, 0) for the 4x4 CU described by lvl[4] on the register-resident formulation of ctu_leaf4.h: same decisions, same
// models, same reconstruction; the walk's wave, on wv[0].
template <typename PX> CTU_NOINLINE CTU_DEV void eval_cu4(lds<PX> *S, const job<PX> &J)
{
  wctx *const V = wv_of(S);
  const params &P = J.P;
  level_state &N = S->lvl[4];
  const int x = N.x, y = N.y, lx = x & 63, ly = y & 63;
  const int has_chroma = N.has_chroma;
  LANE0 {
    V->cur = S->cur;
    cu4 *c = cu_at(S, lx, ly);                           // the CU's own entry is reset (search.c:1371-1388)
    c->type = CU_NOTSET; c->cbf = 0; c->luma_edges = 0; c->chroma_edges = 0; c->mode = 0; c->mode_chroma = 0; c->log2 = 2; c->log2_c = 2;
#if defined(CTU_PB)
    { const int u = ((ly >> 2) + 1) * 17 + (lx >> 2) + 1; S->pb.mot[u].type = CU_NOTSET; S->pb.fl[u][0] = 0; S->pb.fl[u][1] = 0; }
#endif
  }
  LF_T0();
  leaf_load_area(S, J, lx, ly);
  CTU_SYNC();
  LF_T(0);
  int mode, mpm[6];
  { CTU_T0();
  leaf_refs(S, P, V, 0, x, y, lx, ly, 4);
  LF_T(1);
  mode = leaf_rough(S, J, V, x, y, lx, ly, leaf_src(S, 0, lx, ly), mpm);
  CTU_T1(J.W, 0); }
  LANE0 {          // lcu_fill_cu_info (search.c:314-353) for the one entry of a 4x4 CU
    cu4 *c = cu_at(S, lx, ly);
    c->type = CU_INTRA; c->log2 = 2; c->log2_c = 2; c->mode = (int8_t)mode; c->mode_chroma = (int8_t)mode;
#if defined(CTU_PB)
    { const int u = ((ly >> 2) + 1) * 17 + (lx >> 2) + 1; S->pb.mot[u].type = CU_INTRA; S->pb.fl[u][0] = 0; S->pb.fl[u][1] = 0; }   // (as fill_cu: what the neighbours' contexts read)
#endif
    scratch *const W = S->scr;
    W->tree[(ly >> 2) * 16 + (lx >> 2)] = (uint16_t)N.split_tree;
    W->mtt[(ly >> 2) * 16 + (lx >> 2)] = (uint16_t)cu_mtt(N.mode_type_tree, 4);
  }
  CTU_SYNC();
  const int cx = x & ~7, cy = y & ~7, clx = cx & 63, cly = cy & 63;          // the chroma area
  PX *const ry = S->Dy + (ly + 1) * PY + lx + 1;
  PX *const ru = S->Du + ((cly >> 1) + 1) * PC + (clx >> 1) + 1, *const rv = S->Dv + ((cly >> 1) + 1) * PC + (clx >> 1) + 1;
  int16_t *const ky = J.coeff + ly * LCU + lx;
  int16_t *const ku = J.coeff + 4096 + (cly >> 1) * LCU_C + (clx >> 1), *const kv = J.coeff + 5120 + (cly >> 1) * LCU_C + (clx >> 1);
#if defined(CTU_PB)
  const bool helped = false;              // (no chroma helper: the P / B CTU has no depth waves)
#else
  const bool helped = has_chroma && help_post(S, J, cx, cy, mode);
#if defined(CTU_PROFILE)
  if (has_chroma) { LANE0 J.W->prof[1][helped ? 19 : 20] += 1; }
#endif
#endif
  int ssd_y = 0, ssd_u = 0, ssd_v = 0, cbf = 0, lev_y = 0, lev_u = 0, lev_v = 0;
  { CTU_T0();
#pragma nounroll
  for (int color = 0; color < (has_chroma ? 3 : 1); ++color) {          // ONE call site of the block function (see leaf_recon)
#if !defined(CTU_PB)
    if (helped && color == 1) {
#if defined(CTU_PROFILE)
      const unsigned long long tw = __builtin_amdgcn_s_memtime();
#endif
      cbf |= help_wait(S) << 1;
      ssd_u = S->help[4];
      lev_u = (int)LDSP(const int16_t, V->lv1)[CTU_TID & 15];         // (help_run left the levels in the walk's scratch)
#if defined(CTU_PROFILE)
      LANE0 J.W->prof[1][21] += __builtin_amdgcn_s_memtime() - tw;
#endif
      continue;
    }
#endif
    const bool c = color != 0;
    const lf_block B = leaf_recon_inl(S, J, V, color, mode, color == 2 ? (cbf >> 1) & 1 : 0, c ? cx : x, c ? cy : y, c ? clx : lx, c ? cly : ly, c ? 8 : 4, c ? 0 : 1,
                                      color == 0 ? ry : (color == 1 ? ru : rv), c ? PC : PY, color == 0 ? ky : (color == 1 ? ku : kv), c ? LCU_C : LCU);
    cbf |= B.has << color;
    if (color == 0) { ssd_y = B.ssd; lev_y = B.level; } else if (color == 1) { ssd_u = B.ssd; lev_u = B.level; } else { ssd_v = B.ssd; lev_v = B.level; }
  }
  CTU_T1(J.W, 3); }
  CTU_T0();
  LF_T(9);
  // ---- the CU's side information and RD cost (search.c:1700-1774) ----
  double bits = 0;
  LANE0 {
    cu4 *c = cu_at(S, lx, ly);
    c->cbf = (uint8_t)(cbf & 1);
    // mark_deblocking (search.c:1075-1174) for a 4x4 CU: its own luma edges; the area's chroma edges with the CU that carries the chroma
    c->luma_edges = (uint8_t)((x ? 1 : 0) | (y ? 2 : 0));
    if (has_chroma) {
      // the area's chroma flags sit at its first entry and are copied to all four (lcu_fill_chroma_cbfs), with the chroma mode and
      // chroma size of this, the last, CU (lcu_fill_chroma_cu_info, search.c:355-400)
      for (int k = 0; k < 4; ++k) {
        cu4 *q = cu_at(S, clx + (k & 1) * 4, cly + (k >> 1) * 4);
        q->cbf = (uint8_t)((q->cbf & 1) | (cbf & 6));
        q->mode_chroma = (int8_t)mode;
        q->log2_c = 2;
        if ((cx & ~7) && !(k & 1)) q->chroma_edges |= 1;
        if ((cy & ~7) && !(k >> 1)) q->chroma_edges |= 2;
      }
    }
    // uvg_mock_encode_coding_unit with search_cabac.update = 1 (search.c:1700-1716); a 4x4 CU has no split flag
    lf_luma_mode_bits(V->cur, mpm, mode, bits);
    if (has_chroma) chroma_mode_bits(V->cur, 1, mode, mode, bits);
  }
  LF_T(11);
  // cu_rd_cost_tr_split_accurate (search.c:724-986): the three coded block flags, a lane each (their models are distinct) ...
  double tbits;
  {
    CTU_LDS uint32_t *const m = LDSP(uint32_t, V->cur);
    const int lane = CTU_TID;
    const int fmodel = lane == 0 ? M_CBF_LUMA : (lane == 1 ? M_CBF_CB : M_CBF_CR + ((cbf >> 1) & 1));
    const int fbin = lane == 0 ? cbf & 1 : (lane == 1 ? (cbf >> 1) & 1 : (cbf >> 2) & 1);
    int fq = 0;
    if (lane == 0 || (lane < 3 && has_chroma)) {
      const uint32_t st = m[fmodel];
      const int rw = LDSP(const uint8_t, kRate)[fmodel], r0 = rw >> 4, r1 = rw & 15;
      uint32_t s0 = st & 0xffffu, s1 = st >> 16;
      fq = (int)LDSP(const uint32_t, tab_ebits())[(((s0 + s1) >> 8) << 1) ^ (uint32_t)fbin];
      s0 -= (s0 >> r0) & 0x7fe0u;
      s1 -= (s1 >> r1) & 0x7ffeu;
      if (fbin) { s0 += (0x7fffu >> r0) & 0x7fe0u; s1 += (0x7fffu >> r1) & 0x7ffeu; }
      m[fmodel] = (s0 & 0xffffu) | (s1 << 16);
    }
    const int fsum = __builtin_amdgcn_readlane(fq, 0) + __builtin_amdgcn_readlane(fq, 1) + __builtin_amdgcn_readlane(fq, 2);
    CTU_SYNC();
    // ... and the levels' bits (every term a multiple of 2^-15: the order of the sum is immaterial)
    tbits = (double)fsum / 32768.0;
    if (cbf & 1) tbits += coeff_bits4r(S, m, 1, lev_y, 0);
    if (has_chroma) { tbits += coeff_bits4r(S, m, 1, lev_u, 1); tbits += coeff_bits4r(S, m, 1, lev_v, 2); }
  }
  LF_T(12);
  LANE0 {
    const unsigned chroma_ssd = has_chroma ? (unsigned)((unsigned)ssd_u * P.cw_u) + (unsigned)((unsigned)ssd_v * P.cw_v) : 0u;
    const double trc = (unsigned)ssd_y * 1.0 + chroma_ssd * 1.0 + tbits * P.lambda;
    double cost = bits * P.lambda;
    cost += trc;
    N.cost = cost; N.type = CU_INTRA; N.mode = mode; N.cbf = cbf;
  }
  CTU_SYNC();
  LF_T(13);
  CTU_T1(J.W, 5);
}
#endif

// the depth's unsplit candidate becomes the decision: the split lost (work_tree_copy_up in reverse)
template <typename PX> CTU_NOINLINE CTU_DEV void unpark(lds<PX> *S, const job<PX> &J, int L)
{
  const level_state &N = S->lvl[L];
  const int n = 64 >> L, lx = N.x & 63, ly = N.y & 63;
  for (int color = 0; color < 3; ++color) {
    const int c = color != 0, w = n >> c, l2 = ilog2_dev(w), pit = pitch_of(color), spit = c ? LCU_C : LCU;
    PX *D = plane(S, color) + ((ly >> c) + 1) * pit + (lx >> c) + 1;
    int16_t *co = J.coeff + co_off(color) + (ly >> c) * spit + (lx >> c);
    const int off = cand_px_off(L, color);
    if (lds_cfg<PX>::slim && L == 1) {
      const PX *const from = (const PX *)J.W->cand_px32 + off;
      PAR_FOR(e, w * w) { const int r = e >> l2, q = e & (w - 1); D[r * pit + q] = CTU_GLOAD(&from[e]); co[r * spit + q] = CTU_GLOAD(&J.W->cand_co[off + e]); }
      continue;
    }
    const PX *const from = cand_px_of(S, J, L, color);
    PAR_FOR(e, w * w) { const int r = e >> l2, q = e & (w - 1); D[r * pit + q] = from[e]; co[r * spit + q] = CTU_GLOAD(&J.W->cand_co[off + e]); }
  }
  SERIAL {
    for (int yy = ly; yy < ly + n; yy += 4)
      for (int xx = lx; xx < lx + n; xx += 4) { cu4 *c = cu_at(S, xx, yy); c->cbf = 0; c->luma_edges = 0; c->chroma_edges = 0; }
    fill_cu(S, lx, ly, n, N.mode, N.mode, ilog2_dev(n) - 1, N.split_tree, cu_mtt(N.mode_type_tree, L));
    cu_at(S, lx, ly)->cbf = (uint8_t)N.cbf;
    mark_deblocking(S, N.x, N.y, lx, ly, n, 0, 1);
  }
  CTU_SYNC();
}

CTU_NOINLINE CTU_DEV void copy_models(uint32_t *dst_, const uint32_t *src_)
{
  CTU_LDS uint32_t *const dst = LDSP(uint32_t, dst_);
  CTU_LDS const uint32_t *const src = LDSP(const uint32_t, src_);
  PAR_FOR(i, NMX) dst[i] = src[i];
  CTU_SYNC();
}

#if !defined(CTU_PB)
// ---- the 64x64 CU tried with the mode of the first 32x32 CU (combine_intra_cus, search.c:2082-2143) ------------------------------
// The reference tries it after the four 32x32 areas are decided, in place, on the CTU's entry models without adaptation.  Nothing it
// computes depends on the split's outcome but the two modes, and those are the first area's: so the candidate is POSTED when the first
// area is decided (post64) and built beside the walk -- its four luma transform blocks by depth 1's wave, its chroma blocks by depth 2's,
// each between the evaluations that wave is asked for (worker_loop) -- into the workgroup's global scratch (save_px / save_co, laid
// out like the decided planes / the coefficient array), never into D.  A block's references come from the CTU's border (D's row and
// column -1, fixed for the CTU) and from the candidate's own earlier blocks (build_refs64).  At the end of the walk finish64 waits for
// the two chains, prices the CU in the reference's order of summation and, if it wins, brings it into D (unpark64); if the split wins
// there is nothing to undo.  (Rounds 3-5 built it on the walk's wave after the split, saving and restoring the whole CTU around it:
// 4.7 M of a CTU's 25.4 M cycles, profiles/r05_ctu_intra_phases_1080p8.txt.)

// uvg_intra_build_reference for transform block i (raster order) of the candidate: what build_refs computes when the side information says
// "one 64x64 intra CU" -- the counts of uvg_count_available_edge_cus (cu.c:516-537) are those of that state, whatever the split left
// in S->cu -- with the samples inside the CTU taken from the candidate's own planes.
template <typename PX> CTU_NOINLINE CTU_DEV void build_refs64(lds<PX> *S, const job<PX> &J, int color, int i)
{
  wctx *const V = wv_of(S);
  const params &P = J.P;
  const int c = color != 0, w = 32 >> c, pw = 64 >> c;
  const int lx = (i & 1) * 32, ly = (i >> 1) * 32, x = J.x + lx, y = J.y + ly;
  const int px_x = lx >> c, px_y = ly >> c;
  const int pit = pitch_of(color);
  const PX *const Dp = plane(S, color);
  const uint16_t *const C = J.W->save_px + co_off(color);
  // a sample at (row, column) of the plane relative to the CTU's origin
  auto at_ = [&](int r, int q) -> int { return (r < 0 || q < 0) ? (int)Dp[(r + 1) * pit + q + 1] : (int)CTU_GLOAD(&C[r * pw + q]); };
  SERIAL {
    const int lcnt = x == 0 ? 0 : (lx == 0 ? (LCU - ly) / 4 : 8);
    const int tcnt = y == 0 ? 0 : ((ly == 0 || lx == 0) ? 16 : 8);
    int al = lcnt * (c ? 2 : 4);
    if (al > 2 * w) al = 2 * w;
    if (al > ((P.pic_h - y) >> c)) al = (P.pic_h - y) >> c;
    int at = tcnt * (c ? 2 : 4);
    if (at > 2 * w) at = 2 * w;
    if (at > ((P.pic_w - x) >> c)) at = (P.pic_w - x) >> c;
    if (x > 0 && y > 0 && P.wpp && px_y == 0 && at > (LCU >> c) - px_x) at = (LCU >> c) - px_x;
    V->u_avail_left = al; V->u_avail_top = at;
  }
  CTU_SYNC();
  const int al = V->u_avail_left, at = V->u_avail_top;
  const int dc = 1 << (px_info<PX>::depth - 1);
  CTU_LDS uint16_t *const r_top = LDSP(uint16_t, V->top), *const r_left = LDSP(uint16_t, V->left);
  CTU_LDS uint16_t *const r_ftop = LDSP(uint16_t, V->ftop), *const r_fleft = LDSP(uint16_t, V->fleft);
  PAR_FOR(k, V->refn - 1) {
    int lv, tv;
    if (x > 0) lv = at_(px_y + (k < al ? k : al - 1), px_x - 1);
    else lv = y > 0 ? at_(px_y - 1, px_x) : dc;
    if (y > 0) tv = at_(px_y - 1, px_x + (k < at ? k : at - 1));
    else tv = x > 0 ? at_(px_y, px_x - 1) : dc;
    r_left[k + 1] = (uint16_t)lv;
    r_top[k + 1] = (uint16_t)tv;
  }
  SERIAL {
    int corner;
    if (x > 0 && y > 0) corner = at_(px_y - 1, px_x - 1);
    else corner = x > 0 ? at_(px_y, px_x - 1) : (y > 0 ? at_(px_y - 1, px_x) : dc);
    r_left[0] = r_top[0] = (uint16_t)corner;
  }
  CTU_SYNC();
  const int flim = 2 * (64 >> c) < V->refn - 1 ? 2 * (64 >> c) : V->refn - 1;
  PAR_FOR(k, V->refn) {
    int fl, ft;
    if (k == 0) fl = ft = (r_left[1] + 2 * r_left[0] + r_top[1] + 2) >> 2;
    else {
      fl = k < flim ? (r_left[k - 1] + 2 * r_left[k] + r_left[k + 1] + 2) >> 2 : r_left[k];
      ft = k < flim ? (r_top[k - 1] + 2 * r_top[k] + r_top[k + 1] + 2) >> 2 : r_top[k];
    }
    r_fleft[k] = (uint16_t)fl;
    r_ftop[k] = (uint16_t)ft;
  }
  CTU_SYNC();
}

// predict + quantise (RDOQ) + reconstruct transform block i of the candidate, colour `color`, on the calling wave's scratch: the
// prediction is staged in the transform's second buffer and made AGAIN after the inverse transform (it is the cheap part, and the
// image has no room for another tile); samples and levels go to the scratch planes, the levels stay in lv, the SSD lands in
// V->red[color].  -> has_coeffs
template <typename PX> CTU_NOINLINE CTU_DEV int recon_tu64(lds<PX> *S, const job<PX> &J, int color, int i, int mode, int cbf_u, int16_t *lv_)
{
  wctx *const V = wv_of(S);
  scratch *const W = J.W;
  const int c = color != 0, w = 32 >> c, l2 = c ? 4 : 5, pw = 64 >> c;
  const int lx = (i & 1) * 32, ly = (i >> 1) * 32;
  CTU_LDS int16_t *const t0 = LDSP(int16_t, V->t0);
  PX *const pred_ = (PX *)V->t1;
  typename mg_ptr<PX, PX>::type const pred = MGP(PX, PX, pred_);
  typename mg_ptr<PX, int16_t>::type const lv = MGP(PX, int16_t, lv_);
  int sps;
  CTU_GLB const PX *Sp = src_block(J, color, lx >> c, ly >> c, &sps);
  const int depth = (int)px_info<PX>::depth;
  { CTU_T0();
  build_refs64(S, J, color, i);
  predict_block(S, mode, color, w, pred_, w);
  CTU_T1(J.W, 1); }
  { CTU_T0();
  PAR_FOR(e, w * w) { const int r = e >> l2, q = e & (w - 1); t0[e] = (int16_t)((int)Sp[r * sps + q] - (int)pred[e]); }
  CTU_SYNC();
  fwd_pass(w, V->t0, V->t1, l2 - 1 + depth - 8);
  fwd_pass(w, V->t1, V->t0, l2 + 6);
  CTU_T1(J.W, 2); }
  const int qps = scaled_qp<PX>(J.P, color);
  { CTU_T0();
  const double lambda = c ? J.P.c_lambda_tu : J.P.lambda;
  rdoq_wave_ool(S, W, V->t0, lv_, w, color, cbf_u, qps, lambda, depth);
  CTU_SYNC();
  CTU_T1(J.W, 3); }
  const int has = V->rq_i[1];
  uint16_t *const px64 = W->save_px + co_off(color) + (ly >> c) * pw + (lx >> c);
  int16_t *const co64 = W->save_co + co_off(color) + (ly >> c) * pw + (lx >> c);
  PAR_FOR(e, w * w) { const int r = e >> l2, q = e & (w - 1); co64[r * pw + q] = lv[e]; }
  if (has) {
    const int transform_shift = 15 - depth - l2;
    const int shift = 20 - 14 - transform_shift;
    const int32_t scale = (int32_t)kInvQuantScales[qps % 6] << (qps / 6);
    const int32_t add = 1 << (shift - 1);
    PAR_FOR(e, w * w) t0[e] = (int16_t)clampi((lv[e] * scale + add) >> shift, -32768, 32767);
    CTU_SYNC();
    inv_pass(w, V->t0, V->t1, 7);
    inv_pass(w, V->t1, V->t0, 12 - (depth - 8));
  }
  predict_block(S, mode, color, w, pred_, w);
  int acc = 0;
  PAR_FOR(e, w * w) {
    const int r = e >> l2, q = e & (w - 1);
    int v = (int)pred[e];
    if (has) v = clampi((int16_t)(t0[e] + v), 0, (int)px_info<PX>::maxv);
    px64[r * pw + q] = (uint16_t)v;
    const int d = (int)Sp[r * sps + q] - v;
    acc += d * d;
  }
#if defined(__HIPCC__)
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
#endif
  LANE0 V->red[color] = acc >> (2 * (depth - 8));
  CTU_SYNC();
  return has;
}

// luma transform block i of the candidate (depth 1's wave, its own scratch)
template <typename PX> CTU_NOINLINE CTU_DEV void luma64_step(lds<PX> *S, const job<PX> &J, int i)
{
  wctx *const V = wv_of(S);
  const int cy = recon_tu64(S, J, 0, i, S->m64[0], 0, V->lv0);
  double by = 0;
  LANE0 m_code(LDSP(uint32_t, S->coder), 0, M_CBF_LUMA + 0, cy, by);
  WSYNC();
  if (cy) by += coeff_bits(S, S->coder, 0, V->lv0, 32, 0);
  LANE0 { S->h64[i].cy = cy; S->h64[i].ssd_y = V->red[0]; S->h64[i].bits_y = by; }
  CTU_SYNC();
}
// a chroma block of transform unit i (depth 2's wave: a 16x16 block is that depth's luma size, so the levels of Cb and Cr take turns
// in its luma level array)
template <typename PX> CTU_NOINLINE CTU_DEV void chroma64_step(lds<PX> *S, const job<PX> &J, int i, int color)
{
  wctx *const V = wv_of(S);
  const int cu = color == 2 ? S->h64[i].cu : 0;
  const int has = recon_tu64(S, J, color, i, S->m64[1], cu, V->lv0);
  const double cb = coeff_bits(S, S->coder, 0, V->lv0, 16, color);
  LANE0 {
    if (color == 1) { S->h64[i].cu = has; S->h64[i].ssd_u = V->red[1]; S->h64[i].bits = cb; }
    else {
      double fb = 0;
      m_code(LDSP(uint32_t, S->coder), 0, M_CBF_CB + 0, cu, fb);
      m_code(LDSP(uint32_t, S->coder), 0, M_CBF_CR + cu, has, fb);
      double both = S->h64[i].bits;
      both += cb;
      S->h64[i].cv = has; S->h64[i].ssd_v = V->red[2]; S->h64[i].bits = fb + both;
    }
  }
  CTU_SYNC();
}

// the walk: the first 32x32 area is decided and is one intra CU -- ask for the candidate
template <typename PX> CTU_DEV void post64(lds<PX> *S, const job<PX> &J)
{
  SERIAL { S->m64[0] = cu_at(S, 0, 0)->mode; S->m64[1] = cu_at(S, 0, 0)->mode_chroma; }
  CTU_SYNC();
#if defined(__HIPCC__)
  LANE0 mb_store(&S->req64, 1);
#else
  const int me = g_emul_wave;           // host emulation: the other waves' work happens right here
  S->req64 = 1;
  g_emul_wave = 3;
  for (int i = 0; i < 4; ++i) luma64_step(S, J, i);
  g_emul_wave = 2;
  for (int i = 0; i < 4; ++i) { chroma64_step(S, J, i, 1); chroma64_step(S, J, i, 2); }
  g_emul_wave = me;
  S->done64[0] = S->done64[1] = 1;
#endif
}

// the candidate won: into the decided planes, the coefficient array and the side information (the walk's wave)
template <typename PX> CTU_NOINLINE CTU_DEV void unpark64(lds<PX> *S, const job<PX> &J)
{
  scratch *W = J.W;
  level_state &N = S->lvl[0];
  for (int color = 0; color < 3; ++color) {
    const int w = color ? 32 : 64, l2 = color ? 5 : 6, pit = pitch_of(color);
    PX *D = plane(S, color) + pit + 1;
    PAR_FOR(e, w * w) { D[(e >> l2) * pit + (e & (w - 1))] = (PX)CTU_GLOAD(&W->save_px[co_off(color) + e]); J.coeff[co_off(color) + e] = CTU_GLOAD(&W->save_co[co_off(color) + e]); }
  }
  SERIAL {
    for (int e = 0; e < 256; ++e) { cu4 *c = cu_at(S, (e & 15) * 4, (e >> 4) * 4); c->cbf = 0; c->luma_edges = 0; c->chroma_edges = 0; }
    fill_cu(S, 0, 0, 64, S->m64[0], S->m64[1], 5, N.split_tree, cu_mtt(N.mode_type_tree, 0));
    for (int i = 0; i < 4; ++i) cu_at(S, (i & 1) * 32, (i >> 1) * 32)->cbf = (uint8_t)(S->h64[i].cy | S->h64[i].cu << 1 | S->h64[i].cv << 2);
    mark_deblocking(S, N.x, N.y, 0, 0, 64, 0, 1);
    N.type = CU_INTRA;
  }
  CTU_SYNC();
}

// the walk, after the four 32x32 areas: the candidate's cost into lvl[0].cost; -> the split wins
template <typename PX> CTU_DEV bool finish64(lds<PX> *S, const job<PX> &J)
{
  const params &P = J.P;
  level_state &N = S->lvl[0];
#if defined(__HIPCC__)
  { CTU_T0();
  while (mb_load(&S->done64[0]) == 0 || mb_load(&S->done64[1]) == 0) __builtin_amdgcn_s_sleep(2);
  CTU_T1(J.W, 12); }        // (profile slots 12 / 13 of the walk's wave: the wait for the two chains, unpark64)
#endif
  CTU_SYNC();
  LANE0 {
    // the models are the CTU's entry models (the coder's, untouched until its pass after the search) and do not adapt
    // (search_cabac.update is 0 on this path): bits only
    const int mode = S->m64[0], mode_chroma = S->m64[1];
    double bits = 0;
    split_flag_bits(S, P, S->coder, 0, N.x, N.y, 0, 0, 64, 0, bits);
    double mode_bits = 0;
    {   // calc_mode_bits (search.c:988-1003): the luma mode on a copy of the models, the chroma mode without adaptation
      for (int k = 0; k < NMODELS; ++k) S->work[2][k] = S->coder[k];          // (depth 3's entry models: nobody's after the walk)
      luma_mode_bits(S, S->work[2], 0, N.x, N.y, 0, 0, 64, mode, mode_bits);
      if (mode_chroma == mode) mode_bits += m_fbits(S->coder, M_CHROMA_PRED, 0);
      else mode_bits += 2.0 + m_fbits(S->coder, M_CHROMA_PRED, 1);
    }
    mode_bits += bits;
    const double d0 = mode_bits * P.lambda;
    double d1 = 0;
    for (int i = 0; i < 4; ++i) {
      // cu_rd_cost_tr_split_accurate of the unit (tr_cost): SSDs + (luma bits + chroma bits) * lambda
      const unsigned chroma_ssd = (unsigned)((unsigned)S->h64[i].ssd_u * P.cw_u) + (unsigned)((unsigned)S->h64[i].ssd_v * P.cw_v);
      d1 += (unsigned)S->h64[i].ssd_y * 1.0 + chroma_ssd * 1.0 + (S->h64[i].bits_y + S->h64[i].bits) * P.lambda;
    }
    double c2 = 0;
    c2 += d0;
    c2 += d1 + 0 * P.lambda;          // the sum of the four blocks + luma_bits (0) * lambda (search.c:779)
    N.cost = c2;
  }
  CTU_SYNC();
  const bool split_wins = N.split_cost < N.cost;
  CTU_SYNC();
  // post_search_cabac = the unadapted entry models; search_cabac = temp_cabac, the models after the split (:2140-2141): S->cur as it is
  if (split_wins) { SERIAL N.cost = N.split_cost; CTU_SYNC(); }
  else { CTU_T0(); unpark64(S, J); CTU_T1(J.W, 13); }
  return split_wins;
}
#endif

#define V_flag(S) (wv_of(S)->u_flag)

// ---- the depth pipeline ---------------------------------------------------------------------------------------------------
// A CU whose split is tried as well is evaluated unsplit by the wave of ITS depth while the walk goes on into its children: the
// evaluation needs only what is decided before the CU (samples, side information, the entry models) and writes only its own
// candidate buffers.  The reference evaluates the CU first and uses its cost to cut the children short (search.c:1952-1956,
// 2002-2005); evaluating children it would have skipped changes nothing: every cut decides "not split", and so does the final
// comparison whenever a cut would have applied (costs only grow child by child; the pruning test is re-applied when the cost is known).

// ask the wave of depth L (1..3) to evaluate the CU described by lvl[L] / pre[L] (walk's wave, lane 0 has written both)
template <typename PX> CTU_DEV void post_eval(lds<PX> *S, const job<PX> &J, int L)
{
#if defined(__HIPCC__)
  CTU_SYNC();
  LANE0 mb_store(&S->req[L], S->req[L] + 1);
#else
  const int me = g_emul_wave;
  g_emul_wave = 4 - L;                  // host emulation: the other wave's work happens right here
  eval_cu(S, J, L, 1);
  g_emul_wave = me;
  S->done[L] = ++S->req[L];
#endif
}
template <typename PX> CTU_DEV bool eval_ready(lds<PX> *S, int L)
{
#if defined(__HIPCC__)
  return __builtin_amdgcn_readfirstlane(mb_load(&S->done[L]) == S->req[L]) != 0;      // lane 0's view, for every lane
#else
  return !g_emul_lazy;
#endif
}
template <typename PX> CTU_DEV void wait_eval(lds<PX> *S, int L)
{
#if defined(__HIPCC__)
  while (mb_load(&S->done[L]) != S->req[L]) __builtin_amdgcn_s_sleep(2);
  CTU_SYNC();
#endif
}
enum { CODER_FLAGS = 1, CODER_LUMA = 2, CODER_CHROMA = 4, CODER_ALL = 7 };      // the coder's pass by model (below, at coder_pass)
template <typename PX> CTU_DEV void coder_pass(lds<PX> *S, const job<PX> &J, int parts);
#if defined(__HIPCC__)
// waves 1..3: evaluate depth 4 - wave whenever asked, until told to stop (req < 0)
template <typename PX> CTU_DEV void worker_loop(lds<PX> *S, const job<PX> &J)
{
  const int L = 4 - CTU_WAVE;
  int seen = 0, hseen = 0;
  // the 64x64 candidate (post64): depth 1's wave owes it four luma blocks, depth 2's eight chroma blocks, one at a time whenever the
  // wave has no evaluation to do
  const int steps64 = L == 1 ? 4 : (L == 2 ? 8 : 0);
  int n64 = 0;
  bool coded = false;
  for (;;) {
    int r, h = hseen;
    bool step = false;
    for (;;) {
      r = mb_load(&S->req[L]);
      if (r != seen) break;
      if (L == 3) { h = mb_load(&S->hreq); if (h != hseen) break; }        // depth 3's wave also takes the walk's Cb blocks (help_post)
      if (n64 < steps64 && mb_load(&S->req64)) { step = true; break; }
      if (L >= 2 && !coded && mb_load(&S->creq)) {
        // the search is over: this wave's part of the coder's pass (depth 3: the flags, depth 2: the chroma coefficients)
        { CTU_T0();
        coder_pass(S, J, L == 3 ? CODER_FLAGS : CODER_CHROMA);
        CTU_T1(J.W, 8); }
        CTU_SYNC();
        LANE0 mb_store(&S->cdone[3 - L], 1);
        coded = true;
        continue;
      }
      if (L == 3) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(4);
    }
    if (r != seen) {
      if (r < 0) break;
      seen = r;
      eval_cu(S, J, L, 1);
      CTU_SYNC();
      LANE0 mb_store(&S->done[L], r);
    } else if (step) {
      if (L == 1) luma64_step(S, J, n64); else chroma64_step(S, J, n64 >> 1, 1 + (n64 & 1));
      if (++n64 == steps64) { CTU_SYNC(); LANE0 mb_store(&S->done64[L - 1], 1); }
    } else {
      hseen = h;
      help_run(S, J);
      LANE0 mb_store(&S->hdone, h);
    }
  }
}
#endif

// search_cu (search.c:1299-2221) as a depth-first loop over the quad tree of one CTU, run by the first wave
template <typename PX> CTU_DEV void search_ctu(lds<PX> *S, const job<PX> &J)
{
  const params &P = J.P;
  SERIAL {
    level_state &R = S->lvl[0];
    R.x = J.x; R.y = J.y; R.split_tree = 0; R.mode_type_tree = 0; R.has_chroma = 1; R.child = 0;
  }
  CTU_SYNC();
  int L = 0;
  int entering = 1;
  double ret = 0;            // the cost a finished node hands to its parent (uniform: read from LDS after a fence)
  for (;;) {
    level_state &N = S->lvl[L];
    const int n = 64 >> L;
    if (entering) {
      const int x = N.x, y = N.y;
      CTU_SYNC();            // every lane holds its copy before lane 0 may reach the parent's bookkeeping and rewrite this entry
      if (x >= P.pic_w || y >= P.pic_h) { ret = 0; entering = 0; if (L == 0) break; --L; continue; }     // outside: nothing to code (search.c:1350)
      const int inside = x + n <= P.pic_w && y + n <= P.pic_h;
      // check_can_use_intra (search.c:1257-1287)
      const int min_w = 64 >> P.depth_max;
      const int can_intra = inside && ((L >= P.depth_min && L <= P.depth_max) || (x & ~(min_w - 1)) + min_w > P.pic_w || (y & ~(min_w - 1)) + min_w > P.pic_h);
      const int can_split = (!can_intra || L < P.depth_max) && n > 4;
      if (!can_split) {
        // a leaf: evaluated here, straight into the decided state
        if (can_intra) {
          CTU_T0();
          LANE0 S->vsel[CTU_WAVE] = 4 - L;          // the scratch sized for this depth (its own wave has nothing to do for a leaf)
          CTU_SYNC();
#if defined(CTU_LEAF4)
          if (L == 4) eval_cu4(S, J); else
#endif
          eval_cu(S, J, L, 0);
          LANE0 S->vsel[CTU_WAVE] = 0;
          CTU_SYNC();
          CTU_T1(J.W, 19 + L);
        }
        else { SERIAL { N.cost = CTU_MAX_DOUBLE; N.type = CU_NOTSET; } CTU_SYNC(); }
        ret = N.cost; entering = 0; if (L == 0) break; --L; continue;
      }
      SERIAL { N.type = can_intra ? CU_INTRA : CU_NOTSET; N.cost = CTU_MAX_DOUBLE; N.pending = can_intra; }
#if defined(CTU_LEAF4)
      if (n == 8) leaf_load_area(S, J, x & 63, y & 63);          // (the 8x8 CU's chroma blocks and the four 4x4 CUs below it read the source from there)
#endif
      if (can_intra) { copy_models(S->work[L - 1], S->cur); post_eval(S, J, L); }          // its own wave evaluates the CU unsplit from these models ...
      // ... while the walk tries the split: its flag first (models from the CU's entry: cur still holds them)
      SERIAL {
        double split_bits = 0;
        split_flag_bits(S, P, S->cur, 1, x, y, x & 63, y & 63, n, 1, split_bits);
        N.split_bits = split_bits;
        N.split_cost = split_bits * P.lambda;
        N.child = 0;
        level_state &C = S->lvl[L + 1];
        const int cond_infer = (N.mode_type_tree >> ((L - 1 > 0 ? L - 1 : 0) * 2) & 3) == 0 && n == 8;      // uvg_derive_mode_type_cond: MODE_TYPE_INFER
        const uint32_t mode_type = cond_infer ? 2u : (N.mode_type_tree >> ((L - 1 > 0 ? L - 1 : 0) * 2) & 3);
        C.split_tree = N.split_tree | 1u << (L * 3);
        C.mode_type_tree = N.mode_type_tree | mode_type << (L * 2);
        C.x = N.x; C.y = N.y; C.has_chroma = (n >> 1) == 4 ? 0 : 1;
      }
      CTU_SYNC();
      ++L;
      continue;
    }
    // a child of N came back with `ret`
    const bool known = !N.pending || eval_ready(S, L);      // is the CU's own cost there yet?  (only to stop early; never changes the outcome)
    if (known && N.pending) wait_eval(S, L);
    SERIAL {
      N.split_cost += ret;
      const int k = N.child;
      const int last = k == 3;
      V_flag(S) = (known && N.split_cost > N.cost) || last;       // (best_split_cost is still MAX_DOUBLE: one split type)
      N.child = k + 1;
      if (!V_flag(S)) {
        level_state &C = S->lvl[L + 1];
        const int h = n >> 1, k1 = k + 1;
        C.x = N.x + (k1 & 1) * h; C.y = N.y + (k1 >> 1) * h;
        C.has_chroma = h == 4 ? (k1 == 3) : 1;
      }
    }
    CTU_SYNC();
#if !defined(CTU_PB)
    // the first 32x32 area is decided: if it is one intra CU the 64x64 candidate will be tried with its modes (combine_intra_cus) -- ask
    // for it now, the depth waves build it beside the walk (post64).  Only when no leaf can borrow those waves' scratch (vsel).
    if (L == 0 && N.child == 1 && N.type == CU_NOTSET && P.combine_intra_cus && P.depth_max >= 3 && N.x + 64 <= P.pic_w && N.y + 64 <= P.pic_h &&
        cu_at(S, 0, 0)->type == CU_INTRA && cu_at(S, 0, 0)->log2 == 5) post64(S, J);
#endif
    if (!V_flag(S)) { ++L; entering = 1; continue; }
    // the split is complete (or was cut short): the CU's own cost is needed now
    if (N.pending) wait_eval(S, L);
    // The comparison is taken by every lane BEFORE lane 0 may overwrite a cost.
    const double factor = P.qp > 30 ? 1.1 : 1.075;
    const bool pruned = N.type != CU_NOTSET && N.split_bits * P.lambda + N.cost / factor > N.cost;      // the reference would not have tried the split (search.c:1952-1956)
    const bool split_wins = !pruned && N.split_cost < N.cost;
    const int ntype = N.type;
    CTU_SYNC();
#if !defined(CTU_PB)
    if (L == 0 && ntype == CU_NOTSET && P.combine_intra_cus && N.x + 64 <= P.pic_w && N.y + 64 <= P.pic_h &&
        cu_at(S, 0, 0)->type == CU_INTRA && cu_at(S, 0, 0)->log2 == 5) {
      CTU_T0();
      if (!S->req64) post64(S, J);                       // (leaves above depth 3 borrow the depth waves' scratch: nothing was posted beside them)
      (void)finish64(S, J);
      CTU_T1(J.W, 7);
      ret = N.cost;
      break;
    }
#endif
    if (split_wins) {
      SERIAL N.cost = N.split_cost;
      CTU_SYNC();
    } else {
      CTU_T0();
      if (L > 0) copy_models(S->cur, S->work[L - 1]);      // post_search_cabac: the models as the unsplit CU leaves them
      if (ntype != CU_NOTSET) unpark(S, J, L);
      CTU_T1(J.W, 6);
    }
    ret = N.cost;
    if (L == 0) break;
    --L;
  }
}

// ====================================================================== the real coder's model adaptation + CTU in / out ======
CTU_DEV int z_to_x(int z) { return (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4) | ((z >> 3) & 8); }
// A context model is a state machine of its own: what it becomes depends on the bins coded with IT, in coding order, and on nothing
// else; which model a bin uses depends on the syntax (levels, modes, flags, neighbours), never on another model's state.  So the pass
// over the decided CTU splits by MODEL into three passes that each walk the CUs in coding order and touch disjoint entries of
// S->coder: the flags (split, prediction modes, the three cbf), the luma coefficients (significance / greater-than / parity /
// last position / group flags of colour type 0) and the chroma coefficients (the same of colour type 1, Cb before Cr).  Three waves
// run them side by side (run_ctu); the host emulation runs them one after the other.  (enum CODER_*: above worker_loop)

#if defined(CTU_LEAF4)
// the real coder's walk over an 8x8 area of four 4x4 CUs (the shape most of a detailed CTU consists of): the area's 64 + 32 levels are
// fetched once, a lane each; every block's bins go through coeff_bits4r from registers.  Same bins in the same order as coder_pass.
template <typename PX> CTU_DEV void coder_area4(lds<PX> *S, const job<PX> &J, int lx, int ly, int parts)
{
  const params &P = J.P;
  const int lane = CTU_TID, r = lane & 15;
  int lev_all = 0, clev = 0;
  {
    const int k = lane >> 4;
    if (parts & CODER_LUMA) lev_all = CTU_GLOAD(&J.coeff[(ly + (k >> 1) * 4 + (r >> 2)) * LCU + lx + (k & 1) * 4 + (r & 3)]);
    if ((parts & CODER_CHROMA) && lane < 32) clev = CTU_GLOAD(&J.coeff[4096 + k * 1024 + ((ly >> 1) + (r >> 2)) * LCU_C + (lx >> 1) + (r & 3)]);
  }
  CTU_LDS uint32_t *const m = LDSP(uint32_t, S->coder);
  for (int k = 0; k < 4; ++k) {
    const int clx = lx + (k & 1) * 4, cly = ly + (k >> 1) * 4, x = J.x + clx, y = J.y + cly;
    const cu4 *c = cu_at(S, clx, cly);
    const int mode = __builtin_amdgcn_readfirstlane((int)c->mode), cb_y = __builtin_amdgcn_readfirstlane((int)c->cbf) & 1;
    if (parts & CODER_FLAGS) {
      int mpm[6];
      {
        const cu4 *l, *a;
        mpm_neighbours(S, x, y, clx, cly, 4, &l, &a);
        int left_dir = 0, above_dir = 0;
        if (l && l->type == CU_INTRA) left_dir = l->mode;
        if (a && a->type == CU_INTRA && y % LCU != 0) above_dir = a->mode;
        lf_mpm(__builtin_amdgcn_readfirstlane(left_dir), __builtin_amdgcn_readfirstlane(above_dir), mpm);
      }
      LANE0 {
        double dummy = 0;
        if (k == 0)          // split flags of the enclosing quad-tree nodes that begin here (a 4x4 CU has none of its own)
          for (int d = 0; (64 >> d) > 4; ++d) {
            const int sz = 64 >> d;
            if (!(lx & (sz - 1)) && !(ly & (sz - 1))) split_flag_bits(S, P, S->coder, 1, x, y, lx, ly, sz, 1, dummy);
          }
        lf_luma_mode_bits(S->coder, mpm, mode, dummy);
        m_code(m, 1, M_CBF_LUMA + 0, cb_y, dummy);
      }
      WSYNC();
    }
    if ((parts & CODER_LUMA) && cb_y) (void)coeff_bits4r<PX, false>(S, m, 1, lf_shfl(lev_all, k * 16 + r), 0);
    if (k == 3) {
      // the area's chroma after its last luma CU: mode (the co-located luma CU is this one), cbfs of the area's first entry, levels
      const cu4 *a = cu_at(S, lx, ly);
      const int acbf = __builtin_amdgcn_readfirstlane((int)a->cbf), au = (acbf >> 1) & 1, av = (acbf >> 2) & 1;
      if (parts & CODER_FLAGS) {
        LANE0 {
          double dummy = 0;
          chroma_mode_bits(S->coder, 1, c->mode_chroma, c->mode, dummy);
          m_code(m, 1, M_CBF_CB + 0, au, dummy);
          m_code(m, 1, M_CBF_CR + au, av, dummy);
        }
        WSYNC();
      }
      if (parts & CODER_CHROMA) {
        if (au) (void)coeff_bits4r<PX, false>(S, m, 1, lf_shfl(clev, r), 1);
        if (av) (void)coeff_bits4r<PX, false>(S, m, 1, lf_shfl(clev, 16 + r), 2);
      }
    }
  }
  CTU_SYNC();
}
#endif

template <typename PX> CTU_NOINLINE CTU_DEV void coder_pass(lds<PX> *S, const job<PX> &J, int parts)
{
  wctx *const V = wv_of(S);
  const params &P = J.P;
  // the chroma pass runs on depth 2's scratch: a 16x16 block is that depth's luma size, Cb and Cr take turns in its luma level array
  int16_t *const lvu = parts == CODER_CHROMA ? V->lv0 : lv_of(V, 1), *const lvv = parts == CODER_CHROMA ? V->lv0 : lv_of(V, 2);
  for (int z = 0; z < 256; ++z) {
    const int lx = z_to_x(z) * 4, ly = z_to_x(z >> 1) * 4;
    const int x = J.x + lx, y = J.y + ly;
    if (x >= P.pic_w || y >= P.pic_h) continue;
    const cu4 *c = cu_at(S, lx, ly);
    const int n = 1 << c->log2;
    if ((lx & (n - 1)) || (ly & (n - 1))) continue;
#if defined(CTU_LEAF4)
    if (n == 4) {              // (4x4 CUs come as whole 8x8 areas, the first of the four at the area's origin)
      if (!(lx & 4) && !(ly & 4)) coder_area4(S, J, lx, ly, parts);
      continue;
    }
#endif
    const int sep = n == 4, last4 = sep && (lx & 4) && (ly & 4);
    const int tus = n == 64 ? 4 : 1, tn = n == 64 ? 32 : n;
    for (int tu = 0; tu < tus; ++tu) {
      const int tlx = lx + (tu & 1) * 32, tly = ly + (tu >> 1) * 32;
      const cu4 *t = cu_at(S, tlx, tly);
      const int cb_y = t->cbf & 1, cb_u = (t->cbf >> 1) & 1, cb_v = (t->cbf >> 2) & 1;
      // the chroma blocks of this transform unit (for the last 4x4 CU of an 8x8 area: the area's)
      const int has_c = !sep || last4;
      const int cw = sep ? 4 : tn >> 1, cl2 = ilog2_dev(cw);
      const int cbx = (sep ? (tlx & ~7) : tlx) >> 1, cby = (sep ? (tly & ~7) : tly) >> 1;
      const cu4 *a = sep ? cu_at(S, lx & ~7, ly & ~7) : t;       // (4x4 CUs: the flags of the area's first entry)
      const int au = sep ? (a->cbf >> 1) & 1 : cb_u, av = sep ? (a->cbf >> 2) & 1 : cb_v;
      uint32_t *m = S->coder;
      if (parts & CODER_FLAGS) {
        LANE0 {
          double dummy = 0;
          if (tu == 0) {
            // split flags of the enclosing quad-tree nodes that begin here, then this CU's own
            for (int d = 0; (64 >> d) > n; ++d) {
              const int s = 64 >> d;
              if (!(lx & (s - 1)) && !(ly & (s - 1))) split_flag_bits(S, P, m, 1, x, y, lx, ly, s, 1, dummy);
            }
            split_flag_bits(S, P, m, 1, x, y, lx, ly, n, 0, dummy);
            luma_mode_bits(S, m, 1, x, y, lx, ly, n, c->mode, dummy);
            if (!sep) chroma_mode_bits(m, 1, c->mode_chroma, c->mode, dummy);
          }
          if (!sep) {
            m_code(m, 1, M_CBF_CB + 0, cb_u, dummy);
            m_code(m, 1, M_CBF_CR + cb_u, cb_v, dummy);
          }
          m_code(m, 1, M_CBF_LUMA + 0, cb_y, dummy);      // luma_cbf_ctx stays 0: one transform unit per CU, or a CU that is not a TU
          if (last4) {
            // the area's chroma after its last luma CU: mode (the co-located luma CU is this one), cbfs of the area's first entry
            chroma_mode_bits(m, 1, c->mode_chroma, c->mode, dummy);
            m_code(m, 1, M_CBF_CB + 0, au, dummy);
            m_code(m, 1, M_CBF_CR + au, av, dummy);
          }
        }
        WSYNC();
      }
      if ((parts & CODER_LUMA) && cb_y) {
        const int16_t *co = J.coeff + tly * LCU + tlx;
        const int l2 = ilog2_dev(tn);
        PAR_FOR(e, tn * tn) lv_of(V, 0)[e] = CTU_GLOAD(&co[(e >> l2) * LCU + (e & (tn - 1))]);
        CTU_SYNC();
        (void)coeff_bits(S, m, 1, lv_of(V, 0), tn, 0);
      }
      if ((parts & CODER_CHROMA) && has_c) {
        if (au) {
          PAR_FOR(e, cw * cw) lvu[e] = CTU_GLOAD(&J.coeff[4096 + (cby + (e >> cl2)) * LCU_C + cbx + (e & (cw - 1))]);
          CTU_SYNC();
          (void)coeff_bits(S, m, 1, lvu, cw, 1);
        }
        if (av) {
          PAR_FOR(e, cw * cw) lvv[e] = CTU_GLOAD(&J.coeff[5120 + (cby + (e >> cl2)) * LCU_C + cbx + (e & (cw - 1))]);
          CTU_SYNC();
          (void)coeff_bits(S, m, 1, lvv, cw, 2);
        }
      }
      CTU_SYNC();
    }
  }
}

template <typename PX> CTU_NOINLINE CTU_DEV void load_ctu(lds<PX> *S, const job<PX> &J)
{
  const params &P = J.P;
  const int x = J.x, y = J.y, W = P.pic_w, H = P.pic_h;
  BLK_FOR(i, 17 * 17) { cu4 z = {0, 0, 0, 0, 0, 0, 0, 0}; S->cu[i] = z; }
  BLK_FOR(i, 256) { J.W->tree[i] = 0; J.W->mtt[i] = 0; }
  if (BLK_TID == 0) S->scr = J.W;
  BLK_SYNC();
  BLK_FOR(i, 33) {
    // i = 0: the corner, 1..16 the row above, 17..32 the column to the left
    int ax, ay, ok;
    if (i == 0) { ax = x - 4; ay = y - 4; ok = x > 0 && y > 0; }
    else if (i <= 16) { ax = x + (i - 1) * 4; ay = y - 4; ok = y > 0 && ax < W; }
    else { ax = x - 4; ay = y + (i - 17) * 4; ok = x > 0 && ay < H; }
    if (ok) {
      const uvghip_scu_t *s = &J.cu_tab[(ay >> 2) * J.cu_stride + (ax >> 2)];
      cu4 *c = cu_at(S, ax - x, ay - y);
      c->type = s->type; c->log2 = s->log2_width; c->cbf = s->cbf; c->log2_c = s->log2_chroma_width;
      c->mode = (int8_t)(s->mv[0][0] & 0xff); c->mode_chroma = (int8_t)((s->mv[0][0] >> 8) & 0xff);
    }
  }
  for (int color = 0; color < 3; ++color) {
    const int c = color != 0, w = 64 >> c, pit = pitch_of(color);
    PX *D = plane(S, color);
    const PX *rec = color == 0 ? J.rec_y : (color == 1 ? J.rec_u : J.rec_v);
    const int rs = c ? J.rec_stride_c : J.rec_stride;
    const int px = x >> c, py = y >> c, pw = W >> c, ph = H >> c;
    BLK_FOR(i, 2 * w + 1) {
      // i = 0 corner, 1..w the row above, w + 1..2w the column to the left
      if (i == 0) { if (px > 0 && py > 0) D[0] = rec[(py - 1) * rs + px - 1]; }
      else if (i <= w) { const int q = px + i - 1; if (py > 0 && q < pw) D[i] = rec[(py - 1) * rs + q]; }
      else { const int r = py + i - w - 1; if (px > 0 && r < ph) D[(i - w) * pit] = rec[r * rs + px - 1]; }
    }
  }
  BLK_FOR(i, 512) tab_ebits()[i] = kEntropyBits[i];
  BLK_FOR(i, NMODELS) {
    tab_rate()[i] = k_ctx_init[3][i];
    if (J.models_in) S->coder[i] = J.models_in[i];
#if defined(CTU_PB)
    else models_init_one(S->coder, i, J.init_qp, J.slice_type);
#else
    else models_init_one(S->coder, i, P.qp, 2);
#endif
  }
#if defined(CTU_PB)
  BLK_FOR(i, NMX - NMODELS) {
    tab_rate()[NMODELS + i] = k_ctx_init_inter[3][i];
    if (J.models_in) S->coder[NMODELS + i] = J.pbm_in[i];
    else {
      const int v = k_ctx_init_inter[J.slice_type][i];
      const int slope = (v >> 3) - 4, offset = ((v & 7) * 18) + 1;
      int st = ((slope * (J.init_qp - 16)) >> 1) + offset;
      st = st < 1 ? 1 : (st > 127 ? 127 : st);
      S->coder[NMODELS + i] = (uint32_t)((st << 8) & 0x7fe0) | ((uint32_t)((st << 8) & 0x7ffe) << 16);
    }
  }
#endif
  BLK_SYNC();
#if defined(CTU_PB)
  BLK_FOR(i, NMX - NMODELS) { const uint32_t v = S->coder[NMODELS + i]; S->cur[NMODELS + i] = v; J.pbm_out[i] = v; }
#endif
  BLK_FOR(i, NMODELS) {
    const uint32_t v = S->coder[i];
    S->cur[i] = v;
    J.models_out[i] = v;
    if (i < 244) {
      const int st = m_state(S->coder, i);
      S->rdoq_state[i] = (uint8_t)st;
      tab_rdoq_bits()[2 * i] = kEntropyBits[st << 1];
      tab_rdoq_bits()[2 * i + 1] = kEntropyBits[(st << 1) ^ 1];
    }
  }
  BLK_SYNC();
  BLK_FOR(cfg, 16) {            // calc_last_bits (rdo.c:664-700) for every shape once: the models uvg_rdoq prices with are the CTU's
    const int t = cfg >> 3, l2 = 2 + ((cfg >> 1) & 3), xy = cfg & 1, n = 1 << l2;
    const int prefix_ctx[8] = {0, 0, 0, 3, 6, 10, 15, 21};
    const int off = t ? 0 : prefix_ctx[l2];
    const int sh = t ? clampi(n >> 3, 0, 2) : ((l2 + 1) >> 2);
    const int base = (xy ? M_LASTY : M_LASTX) + (t ? 20 : 0) + off;
    int32_t b = 0;
    int ctx;
    if (t && l2 == 5) continue;                 // (no 32x32 chroma block)
    int32_t *const lb = S->last_bits + last_bits_off(t, l2, xy);
    for (ctx = 0; ctx < group_idx(n - 1); ctx++) {
      const int mdl = base + (ctx >> sh);
      lb[ctx] = b + (int32_t)tab_rdoq_bits()[2 * mdl];
      b += (int32_t)tab_rdoq_bits()[2 * mdl + 1];
    }
    lb[ctx] = b;
  }
  BLK_SYNC();
}

// copy_lcu_to_cu_data (search.c:2331-2377) + the models of the three checkpoints
template <typename PX> CTU_NOINLINE CTU_DEV void store_ctu(lds<PX> *S, const job<PX> &J)
{
  const params &P = J.P;
  const int x = J.x, y = J.y, W = P.pic_w, H = P.pic_h;
  for (int color = 0; color < 3; ++color) {
    const int c = color != 0, w = 64 >> c, pit = pitch_of(color);
    const PX *D = plane(S, color) + pit + 1;
    PX *rec = color == 0 ? J.rec_y : (color == 1 ? J.rec_u : J.rec_v);
    const int rs = c ? J.rec_stride_c : J.rec_stride;
    const int px = x >> c, py = y >> c, pw = W >> c, ph = H >> c;
    BLK_FOR(e, w * w) {
      const int r = e / w, q = e - r * w;
      if (py + r < ph && px + q < pw) rec[(py + r) * rs + px + q] = D[r * pit + q];
    }
  }
  BLK_FOR(e, 256) {
    const int lx = (e & 15) * 4, ly = (e >> 4) * 4;
    if (x + lx < W && y + ly < H) {
      const cu4 *c = cu_at(S, lx, ly);
      uvghip_scu_t s;
      memset(&s, 0, sizeof s);
      s.luma_edges = c->luma_edges; s.chroma_edges = c->chroma_edges; s.type = c->type; s.cbf = c->cbf; s.qp = (int8_t)P.qp;
      s.log2_width = s.log2_height = c->log2; s.log2_chroma_width = s.log2_chroma_height = c->log2_c;
      s.mv[0][0] = (int32_t)((uint32_t)(uint8_t)c->mode | (uint32_t)(uint8_t)c->mode_chroma << 8);
      s.mv[0][1] = (int32_t)CTU_GLOAD(&J.W->tree[e]);
      s.mv[1][0] = (int32_t)CTU_GLOAD(&J.W->mtt[e]);
      J.cu_tab[((y + ly) >> 2) * J.cu_stride + ((x + lx) >> 2)] = s;
    }
  }
}

// carve the arena: wv[k] serves depth 4 - k (blocks of 4 << k)
template <typename PX> CTU_DEV void setup_waves(lds<PX> *S, scratch *W = nullptr)
{
  if (lds_cfg<PX>::slim_scan) { if (BLK_TID == 0) S->scr = W; }      // (scan_of reads it before load_ctu sets it again)
  BLK_FOR(k, 4) {
    const int n = 4 << k, nn = n * n, c2 = (n / 2) * (n / 2) < 16 ? 16 : (n / 2) * (n / 2), tiles = n >= 8 ? (n / 8) * (n / 8) : 1;
    int off = 0;
#if !defined(CTU_PB)
    for (int j = 0; j < k; ++j) off += arena_bytes(4 << j);          // (the slim depth-1 share is the last one)
#else
    if (k == 0) off = arena_bytes(32);
#endif
    unsigned char *a = S->arena + off;
#if defined(CTU_PB)
    if (k == 1 && BLK_NT > 128) a = S->arena8;             // (three waves: the walk evaluates 8x8 CUs while the depth wave is in the big region)
    if (k == 2 && BLK_NT > 192) a = S->arena16;            // (four waves: 16x16 and 32x32 CUs on a wave each)
#endif
    unsigned char *const a0 = a;
    wctx *V = &S->wv[k];
    const int rn = 4 * n + 8;
    V->refn = (int16_t)rn;
    V->top = (uint16_t *)a; V->left = V->top + rn; V->ftop = V->left + rn; V->fleft = V->ftop + rn; a += 4 * rn * 2;
    V->t0 = (int16_t *)a; V->t1 = V->t0 + nn; a += 2 * nn * 2;
    if (lds_cfg<PX>::slim && n == 32) { V->lv0 = W->lv32; V->lv1 = V->lv0 + nn; V->lv2 = V->lv1 + c2; }
    else { V->lv0 = (int16_t *)a; V->lv1 = V->lv0 + nn; V->lv2 = V->lv1 + c2; a += (nn + 2 * c2) * 2; }
    // the rough search's (satd, sad) per (mode, tile) -- 2 * 18 * tiles words -- fit the three transform buffers from 8x8 on, which
    // idle until the mode is chosen
#if defined(CTU_LEAF4)
    if (n == 4) { V->part = nullptr; V->rq_cc = V->rq_cs = nullptr; }       // (the general evaluation never runs on this scratch: eval_cu4)
    else
#endif
    {
    if (n >= 8) V->part = (uint32_t *)V->t0;
    else { V->part = (uint32_t *)a; a += 2 * 18 * tiles * 4; }
    a = a0 + ((a - a0 + 7) & ~7);
    if (n <= 8) { V->rq_cc = (double *)a; V->rq_cs = V->rq_cc + nn; }
    else V->rq_cc = V->rq_cs = nullptr;
    }
    V->cur = S->cur;
    S->vsel[k] = k;
    S->req[k] = 0; S->done[k] = 0;
    if (k == 0) { S->hreq = 0; S->hdone = 0; }
#if !defined(CTU_PB)
    if (k == 0) { S->req64 = 0; S->done64[0] = 0; S->done64[1] = 0; S->creq = 0; S->cdone[0] = 0; S->cdone[1] = 0; }
#endif
  }
  BLK_SYNC();
}

// one CTU, start to finish (all four waves)
template <typename PX> CTU_DEV void run_ctu(lds<PX> *S, const job<PX> &J)
{
#if defined(__HIPCC__) && defined(CTU_PROFILE)
  BLK_FOR(i, 4 * 32) J.W->prof[i >> 5][i & 31] = 0;
  BLK_FOR(i, 16) J.W->prof_lf[i] = 0;
  S->prof_w = J.W;
#endif
  CTU_T0();
  { CTU_T0();
  setup_waves(S, J.W);
#if defined(CTU_LEAF4)
  leaf_tables(S, J.P);
#endif
  if (CTU_WAVE == 0) build_scans(S);
  BLK_SYNC();
  load_ctu(S, J);
  // the CTU's coefficient array starts empty (lcu->coeff = calloc, encoderstate.c:752)
  BLK_FOR(e, 6144) J.coeff[e] = 0;
  BLK_SYNC();
  CTU_T1(J.W, 9); }
#if defined(__HIPCC__)
  if (CTU_WAVE == 0) {
    __builtin_amdgcn_s_setprio(3);        // the walker is the CTU's critical path: it wins the issue slot against the depth waves sharing its SIMD
#endif
    search_ctu(S, J);
    PAR_FOR(i, NMODELS) J.models_out[NMODELS + i] = S->cur[i];
    { CTU_T0();
    LANE0 S->vsel[CTU_WAVE] = 3;          // the depth-1 scratch for the coder's 32x32 blocks: the other waves are idle now
    CTU_SYNC();
#if defined(__HIPCC__)
    // the coder's pass, split by model (CODER_*): the flags on depth 3's wave, the chroma coefficients on depth 2's, the luma
    // coefficients here
    LANE0 mb_store(&S->creq, 1);
    { CTU_T0();
    coder_pass(S, J, CODER_LUMA);
    CTU_T1(J.W, 14); }        // (profile slot 14 of the walk's wave: its own part; slot 8: with the wait for the other two)
    while (mb_load(&S->cdone[0]) == 0 || mb_load(&S->cdone[1]) == 0) __builtin_amdgcn_s_sleep(1);
    CTU_SYNC();
#else
    coder_pass(S, J, CODER_LUMA);
    LANE0 S->vsel[CTU_WAVE] = 0;
    { const int me = g_emul_wave; g_emul_wave = 1; coder_pass(S, J, CODER_FLAGS); g_emul_wave = 2; coder_pass(S, J, CODER_CHROMA); g_emul_wave = me; }
#endif
    CTU_T1(J.W, 8); }
#if defined(__HIPCC__)
    LANE0 { for (int L = 1; L <= 3; ++L) mb_store(&S->req[L], -1); }
  } else {
    worker_loop(S, J);
  }
#endif
  BLK_SYNC();
  { CTU_T0();
  BLK_FOR(i, NMODELS) J.models_out[2 * NMODELS + i] = S->coder[i];
  store_ctu(S, J);
  BLK_SYNC();
  CTU_T1(J.W, 10); }
  CTU_T1(J.W, 11);
}

}  // namespace ctu
