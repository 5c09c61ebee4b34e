// ABI mirrors of the reference structs that cross the strategy boundary
// (SURVEY.md section 8(a) row T).  The layouts must be byte-identical to the
// host encoder's (gcc/clang x86-64 bit-field rules); sizes are asserted.
//   cu_info_t : src/cu.h:134-198   (40 bytes)
//   cu_loc_t  : src/cu.h:200-209   (10 bytes)
// Only the host-side per-call wrappers read these; device code receives
// plain unpacked parameters.
#pragma once
#include <stdint.h>
#ifndef __cplusplus
#define static_assert _Static_assert
#endif

typedef int32_t ref_mv_t;  // src/global.h:129

typedef struct ref_cu_info {
  uint8_t type : 3, skipped : 1, merged : 1, merge_idx : 3;
  uint8_t tr_skip : 3, tr_idx : 3, joint_cb_cr : 2;
  uint8_t log2_width : 3, log2_height : 3;
  uint8_t log2_chroma_width : 3, log2_chroma_height : 3;
  uint16_t cbf;
  uint8_t root_cbf;
  uint32_t split_tree : 27;
  uint32_t mode_type_tree : 18;
  uint8_t qp;
  uint8_t bdpcmMode : 1, violates_mts_coeff_constraint : 1, mts_last_scan_pos : 1,
          violates_lfnst_constrained_luma : 1, violates_lfnst_constrained_chroma : 1,
          lfnst_last_scan_pos : 1, lfnst_idx : 2;
  uint8_t cr_lfnst_idx : 2, luma_deblocking : 2, chroma_deblocking : 2;
  union {
    struct {
      int8_t mode, mode_chroma;
      uint8_t multi_ref_idx;
      int8_t mip_flag, mip_is_transposed, isp_mode;
      uint8_t isp_cbfs : 4, isp_index : 2;
    } intra;
    struct {
      ref_mv_t mv[2][2];
      uint8_t mv_ref[2];
      uint8_t mv_cand0 : 1, mv_cand1 : 1, mv_dir : 2, imv : 2;
    } inter;
  };
} ref_cu_info;
static_assert(sizeof(ref_cu_info) == 40, "cu_info_t mirror must be 40 bytes (SURVEY.md 2: [measured])");

enum { REF_CU_NOTSET = 0, REF_CU_INTRA = 1, REF_CU_INTER = 2 };  // src/cu.h cu_type_t (first values)

typedef struct ref_cu_loc {
  int16_t x, y;
  uint8_t local_x, local_y, width, height, chroma_width, chroma_height;
} ref_cu_loc;
static_assert(sizeof(ref_cu_loc) == 10, "cu_loc_t mirror");

// sao_info_t : src/sao.h:55-63 (enums are ints)
typedef struct ref_sao_info {
  int type, eo_class;
  int ddistortion, merge_left_flag, merge_up_flag;
  int band_position[2];
  int offsets[10];
} ref_sao_info;
static_assert(sizeof(ref_sao_info) == 68, "sao_info_t mirror");

#ifdef __cplusplus
// uvg_epol_args : src/strategies/strategies-ipol.h:67-92 (PX = uvg_pixel of the build)
template <typename PX> struct ref_epol_args {
  PX *src; int src_w, src_h, src_s;
  int blk_x, blk_y, blk_w, blk_h, pad_l, pad_r, pad_t, pad_b, pad_b_simd;
  PX *buf;
  PX **ext, **ext_origin;
  int *ext_s;
};
static_assert(sizeof(ref_epol_args<uint8_t>) == 88, "uvg_epol_args mirror");
#endif
