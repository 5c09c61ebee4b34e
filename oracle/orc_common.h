/*
 * oracle/ -- CPU restatement of the uvg266 generic-C strategy kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under uvg266_amd/ (the product) may
 * include, link, dlopen or call anything in this directory.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only
 * as the checker.
 *
 * Every file is compiled twice (-DORC_BIT_DEPTH=8 and =10); exported symbols
 * are prefixed orc8_ / orc10_ because the reference fixes the pixel type at
 * compile time (uvg266.h:89-99: uvg_pixel = uint8_t or uint16_t).
 *
 * Pinning status (see DESIGN.md "Oracle"): the SAD/SATD/coeff_abs_sum
 * functions are pinned by the reference's own known-answer tests
 * (tests/satd_tests.c:122,140,159; tests/sad_tests.c:144-285,388-409;
 * tests/intra_sad_tests.c; tests/coeff_sum_tests.c), restated in
 * tests/golden/kat_*.json.  Everything else has no upstream unit vector and
 * is pinned by tests/golden/ref_*.bin (provenance: tests/golden/README.md).
 */
#ifndef ORC_COMMON_H_
#define ORC_COMMON_H_

#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#ifndef ORC_BIT_DEPTH
#error "compile with -DORC_BIT_DEPTH=8 or 10"
#endif

#if ORC_BIT_DEPTH == 8
typedef uint8_t orc_px;
#define ORC_FN(name) orc8_##name
#else
typedef uint16_t orc_px;
#define ORC_FN(name) orc10_##name
#endif

#define ORC_PX_MAX ((1 << ORC_BIT_DEPTH) - 1)
#define ORC_DSHIFT (ORC_BIT_DEPTH - 8)

#define ORC_EXPORT __attribute__((visibility("default")))

static inline int orc_clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int orc_clip16(int v) { return orc_clip3(-32768, 32767, v); }
static inline orc_px orc_clip_px(int v) { return (orc_px)orc_clip3(0, ORC_PX_MAX, v); }
static inline int orc_iabs(int v) { return v < 0 ? -v : v; }
static inline int orc_log2i(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

/* count mode of the arithmetic coder (orc_coeff_cost.c): range, renormalisation shifts and the estimate share of the regular bins */
typedef struct orc_cabac_sim {
  int on;                       /* 0 off, 1 count (range + shifts), 2 the whole arithmetic coder (low, carries, bytes) */
  uint32_t range; uint64_t shifts; double regular_fbits;
  uint32_t low, buffered_byte; int32_t bits_left, num_buffered_bytes;      /* cabac_data_t (cabac.h:56-66) */
  uint8_t *out; size_t out_len, out_cap;                                    /* payload bytes handed to the bitstream */
} orc_cabac_sim;

#endif
