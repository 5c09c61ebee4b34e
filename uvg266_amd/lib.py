"""ctypes loader for libuvg266hip.so (built in-tree by uvg266_amd/csrc/Makefile)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UVGHIP_LIB") or os.path.join(_HERE, "libuvg266hip.so")      # UVGHIP_LIB: a variant build of the same library (profiling builds, tools/dev)


class LibraryMissing(RuntimeError):
    pass


class DeviceMissing(RuntimeError):
    pass


class Blk(ctypes.Structure):
    """uvghip_blk_t (include/uvg266_hip.h)."""
    _fields_ = [("cur_x", ctypes.c_int32), ("cur_y", ctypes.c_int32),
                ("ref_x", ctypes.c_int32), ("ref_y", ctypes.c_int32)]


class BandPlan(ctypes.Structure):
    """uvghip_band_plan_t."""
    _fields_ = [(n, ctypes.c_int32) for n in ("rank", "nranks", "ctu_rows", "ctu_row0", "ctu_row1", "y0", "y1", "up", "down")]


class Xfer(ctypes.Structure):
    """uvghip_xfer_t."""
    _fields_ = [("peer", ctypes.c_int32), ("reserved", ctypes.c_int32), ("send", ctypes.c_void_p), ("send_bytes", ctypes.c_uint64),
                ("recv", ctypes.c_void_p), ("recv_bytes", ctypes.c_uint64)]


class QrParams(ctypes.Structure):
    """uvghip_qr_params_t."""
    _fields_ = [(n, ctypes.c_int32) for n in ("width", "height", "color", "type_hor", "type_ver", "skip_width", "skip_height", "qp_scaled",
                                                "slice_is_intra", "cu_type", "use_trskip", "rdoq_enable", "rdoq_skip", "dep_quant", "cbf_u",
                                                "mts_idx", "lfnst_idx", "signhide_enable")] + [("lambda_", ctypes.c_double), ("ctx", ctypes.c_uint8 * 244)]


class CabacModels(ctypes.Structure):
    """uvghip_cabac_models_t: the coefficient coder's context models with their full state."""
    _fields_ = [("state0", ctypes.c_uint16 * 244), ("state1", ctypes.c_uint16 * 244), ("rate", ctypes.c_uint8 * 244)]


class StateView(ctypes.Structure):
    """uvghip_state_view_t: what the state-taking strategies read from encoder_state_t."""
    _fields_ = [(n, ctypes.c_int32) for n in ("bitdepth", "qp", "slice_is_intra", "rdoq_enable", "rdoq_skip", "dep_quant", "signhide_enable",
                                                "scaling_list_enabled", "lfnst", "mts", "lmcs_chroma_adj_enabled", "collocated_luma_mode", "jccr_sign", "reserved")] + \
               [("lambda_", ctypes.c_double), ("c_lambda", ctypes.c_double), ("qp_map", ctypes.c_int8 * 64), ("cabac", ctypes.c_uint8 * 244)]


class CuView(ctypes.Structure):
    """uvghip_cu_view_t."""
    _fields_ = [(n, ctypes.c_int8) for n in ("type", "tr_idx", "lfnst_idx", "cr_lfnst_idx", "log2_width", "log2_height", "intra_mode",
                                               "intra_mode_chroma", "mip_flag", "isp_mode")] + [("cbf", ctypes.c_uint16), ("joint_cb_cr", ctypes.c_int8),
                                                                                          ("reserved", ctypes.c_int8 * 3)]


class CtuParams(ctypes.Structure):
    """uvghip_ctu_params_t: what the closed-loop CTU search reads from encoder_state_t / encoder_control_t."""
    _fields_ = [(n, ctypes.c_int32) for n in ("pic_w", "pic_h", "qp", "qp_c", "depth_min", "depth_max", "wpp", "combine_intra_cus", "rough_levels",
                                                "rd")] + \
               [(n, ctypes.c_double) for n in ("lambda_", "lambda_sqrt", "c_lambda", "chroma_weight_u", "chroma_weight_v", "c_lambda_tu")]


class CtuPicture(ctypes.Structure):
    """uvghip_ctu_picture_t."""
    _fields_ = [("src_y", ctypes.c_void_p), ("src_u", ctypes.c_void_p), ("src_v", ctypes.c_void_p), ("src_stride", ctypes.c_int32),
                ("src_stride_c", ctypes.c_int32), ("rec_y", ctypes.c_void_p), ("rec_u", ctypes.c_void_p), ("rec_v", ctypes.c_void_p),
                ("rec_stride", ctypes.c_int32), ("rec_stride_c", ctypes.c_int32), ("cu", ctypes.c_void_p), ("cu_stride", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("coeff", ctypes.c_void_p), ("models", ctypes.c_void_p)]


class SlicePb(ctypes.Structure):
    """uvghip_slice_pb_t."""
    _fields_ = [("slice_type", ctypes.c_int32), ("poc", ctypes.c_int32), ("n_refs", ctypes.c_int32), ("ref_pocs", ctypes.c_int32 * 16),
                ("l_size", ctypes.c_int32 * 2), ("l", (ctypes.c_int32 * 16) * 2), ("tmvp", ctypes.c_int32), ("max_merge", ctypes.c_int32),
                ("merge_level", ctypes.c_int32), ("frame_qp", ctypes.c_int32), ("col", ctypes.c_void_p), ("inter4", ctypes.c_void_p),
                ("models_inter", ctypes.c_void_p), ("col_stride", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class CtuPbPicture(ctypes.Structure):
    """uvghip_ctu_pb_picture_t: one P / B picture of the closed-loop CTU search."""
    _fields_ = [("params", CtuParams), ("pic", CtuPicture), ("slice_type", ctypes.c_int32), ("poc", ctypes.c_int32), ("n_refs", ctypes.c_int32),
                ("ref_pocs", ctypes.c_int32 * 16), ("l_size", ctypes.c_int32 * 2), ("l", (ctypes.c_int32 * 16) * 2), ("tmvp", ctypes.c_int32),
                ("max_merge", ctypes.c_int32), ("merge_level", ctypes.c_int32), ("frame_qp", ctypes.c_int32), ("bipred", ctypes.c_int32),
                ("fme_level", ctypes.c_int32), ("early_skip", ctypes.c_int32), ("depth_inter_min", ctypes.c_int32), ("depth_inter_max", ctypes.c_int32),
                ("ref_stride", ctypes.c_int32), ("ref_stride_c", ctypes.c_int32), ("ref_motion_stride", ctypes.c_int32), ("inflight_margin", ctypes.c_int32),
                ("ref_y", ctypes.c_void_p * 16), ("ref_u", ctypes.c_void_p * 16), ("ref_v", ctypes.c_void_p * 16), ("ref_motion", ctypes.c_void_p * 16),
                ("inter4", ctypes.c_void_p), ("models_inter", ctypes.c_void_p), ("trees", ctypes.c_void_p), ("motion_out", ctypes.c_void_p)]


class LoopPbPicture(ctypes.Structure):
    """uvghip_loop_pb_picture_t."""
    _fields_ = [("search", CtuPbPicture), ("out_y", ctypes.c_void_p), ("out_u", ctypes.c_void_p), ("out_v", ctypes.c_void_p), ("out_stride", ctypes.c_int32),
                ("out_stride_c", ctypes.c_int32)]


class InflightExternal(ctypes.Structure):
    """uvghip_inflight_external_t."""
    _fields_ = [("searched_flags", ctypes.c_void_p), ("sao_info", ctypes.c_void_p), ("sao_models", ctypes.c_void_p)]


class AlfPicture(ctypes.Structure):
    """uvghip_alf_picture_t."""
    _fields_ = [("in_y", ctypes.c_void_p), ("in_u", ctypes.c_void_p), ("in_v", ctypes.c_void_p), ("in_stride", ctypes.c_int32), ("in_stride_c", ctypes.c_int32),
                ("out_y", ctypes.c_void_p), ("out_u", ctypes.c_void_p), ("out_v", ctypes.c_void_p), ("out_stride", ctypes.c_int32), ("out_stride_c", ctypes.c_int32),
                ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("slice_enabled", ctypes.c_int32 * 3), ("n_luma_aps", ctypes.c_int32),
                ("ctu_flags", ctypes.c_void_p), ("filter_set_idx", ctypes.c_void_p), ("luma_aps", ctypes.c_void_p), ("chroma_aps", ctypes.c_void_p),
                ("alf_full", ctypes.c_int32), ("cc_alf_enabled", ctypes.c_int32 * 2), ("cc_coeff", ctypes.c_void_p), ("classification_shift", ctypes.c_int32)]


class SliceAlf(ctypes.Structure):
    """uvghip_slice_alf_t."""
    _fields_ = [("alf_type", ctypes.c_int32), ("enabled", ctypes.c_int32 * 3), ("n_luma_aps", ctypes.c_int32), ("n_alternatives_chroma", ctypes.c_int32),
                ("cc_enabled", ctypes.c_int32 * 2), ("cc_filter_count", ctypes.c_int32 * 2), ("ctu_flags", ctypes.c_void_p), ("filter_set_idx", ctypes.c_void_p)]


class AlfSlice(ctypes.Structure):
    """uvghip_alf_slice_t."""
    _fields_ = [("alf_type", ctypes.c_int32), ("enabled", ctypes.c_int32 * 3), ("n_luma_aps", ctypes.c_int32), ("luma_aps_id", ctypes.c_int32 * 8),
                ("chroma_aps_id", ctypes.c_int32), ("cc_enabled", ctypes.c_int32 * 2), ("cc_aps_id", ctypes.c_int32 * 2)]


class AlfAps(ctypes.Structure):
    """uvghip_alf_aps_t."""
    _fields_ = [("aps_id", ctypes.c_int32), ("new_filter", ctypes.c_int32 * 2), ("non_linear", ctypes.c_int32 * 2), ("num_luma_filters", ctypes.c_int32),
                ("num_alternatives_chroma", ctypes.c_int32), ("new_cc_filter", ctypes.c_int32 * 2), ("cc_filter_count", ctypes.c_int32 * 2),
                ("luma", ctypes.c_void_p), ("chroma", ctypes.c_void_p), ("cc", ctypes.c_void_p)]


class MeJob(ctypes.Structure):
    """uvghip_me_job_t."""
    _fields_ = [("x", ctypes.c_int32), ("y", ctypes.c_int32), ("ref", ctypes.c_int32), ("mv_cand", (ctypes.c_int32 * 2) * 2), ("extra_mv", ctypes.c_int32 * 2),
                ("n_start", ctypes.c_int32), ("start", (ctypes.c_int32 * 2) * 6)]


class MeResult(ctypes.Structure):
    """uvghip_me_result_t."""
    _fields_ = [("mv", ctypes.c_int32 * 2), ("int_mv", ctypes.c_int32 * 2), ("cost", ctypes.c_double), ("bits", ctypes.c_double), ("int_cost", ctypes.c_double),
                ("int_bits", ctypes.c_double), ("mv_cand", ctypes.c_int32), ("skipped_hexagon", ctypes.c_int32)]


class LoopPicture(ctypes.Structure):
    """uvghip_loop_picture_t."""
    _fields_ = [("search", CtuPicture), ("out_y", ctypes.c_void_p), ("out_u", ctypes.c_void_p), ("out_v", ctypes.c_void_p),
                ("out_stride", ctypes.c_int32), ("out_stride_c", ctypes.c_int32)]


class AlfDecision(ctypes.Structure):
    """uvghip_alf_decision_t: what the host's ALF derivation returns for one picture (host arrays)."""
    _fields_ = [("alf_type", ctypes.c_int32), ("enabled", ctypes.c_int32 * 3), ("n_luma_aps", ctypes.c_int32), ("luma_aps", ctypes.c_void_p), ("chroma_aps", ctypes.c_void_p),
                ("cc_enabled", ctypes.c_int32 * 2), ("cc_filter_count", ctypes.c_int32 * 2), ("cc_coeff", ctypes.c_void_p), ("ctu_flags", ctypes.c_void_p),
                ("filter_set_idx", ctypes.c_void_p)]


class AlfPlanes(ctypes.Structure):
    """uvghip_alf_planes_t."""
    _fields_ = [("y", ctypes.c_void_p), ("u", ctypes.c_void_p), ("v", ctypes.c_void_p), ("stride", ctypes.c_int32), ("stride_c", ctypes.c_int32)]


ALF_DECIDE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(LoopPicture), ctypes.POINTER(AlfDecision))      # uvghip_alf_decide_fn

_lib = None
_inited_device = None

c_int, c_vp, c_u32p = ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p

# name -> (restype, argtypes).  Must list every symbol include/uvg266_hip.h declares
# (tests/test_abi.py checks the header against this table and against the .so).
SIGNATURES = {
    "uvghip_init": (c_int, [c_int]),
    "uvghip_last_error": (ctypes.c_char_p, []),
    "uvghip_abi_version": (c_int, []),
    "uvghip_set_register_fn": (None, [c_vp]),
    "uvghip_graph_begin": (c_int, [c_vp]),
    "uvghip_graph_end": (c_int, [c_vp, c_vp]),
    "uvghip_graph_launch": (c_int, [c_vp, c_vp]),
    "uvghip_graph_destroy": (c_int, [c_vp]),
    "uvg_strategy_register_picture_hip": (c_int, [c_vp, ctypes.c_uint8]),
    "uvg_strategy_register_dct_hip": (c_int, [c_vp, ctypes.c_uint8]),
    "uvghip_transform_batch": (c_int, [c_int] * 8 + [c_vp, c_vp, c_int, c_vp]),
    "uvghip_mts_select": (c_int, [c_int] * 9 + [c_vp] * 4),
    "uvg_strategy_register_quant_hip": (c_int, [c_vp, ctypes.c_uint8]),
    "uvghip_quant_batch": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "uvghip_quant_lfnst_batch": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "uvghip_quant_signhide_batch": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "uvghip_dequant_batch": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "uvghip_coeff_abs_sum_batch": (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    "uvghip_fast_coeff_cost_batch": (c_int, [c_vp, c_int, c_int, c_int, ctypes.c_uint64, c_vp, c_vp]),
    "uvghip_rdoq_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "uvghip_rdoq_batch": (c_int, [c_int, c_vp, c_vp] + [c_int] * 9 + [ctypes.c_double, c_vp, c_vp, ctypes.c_size_t, c_vp, c_vp, c_vp]),
    "uvghip_rdoq_signhide_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "uvghip_rdoq_signhide_batch": (c_int, [c_int, c_vp, c_vp] + [c_int] * 9 + [ctypes.c_double, c_vp, c_vp, ctypes.c_size_t, c_vp, c_vp, c_vp]),
    "uvghip_tu_forward_batch": (c_int, [c_int] * 8 + [c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp]),
    "uvghip_tu_inverse_batch": (c_int, [c_int] * 8 + [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp]),
    "uvghip_tu_dequant_inverse_batch": (c_int, [c_int] * 6 + [c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp]),
    "uvghip_quantize_residual_workspace_bytes": (ctypes.c_size_t, [c_vp, c_int]),
    "uvghip_quantize_residual_batch": (c_int, [c_int, c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp,
                                               ctypes.c_size_t, c_vp]),
    "uvghip_tu_roundtrip_batch": (c_int, [c_int] * 9 + [c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp]),
    "uvg_strategy_register_intra_hip": (c_int, [c_vp, ctypes.c_uint8]),
    "uvghip_intra_pred_batch": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp]),
    "uvghip_mip_pred_batch": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_vp]),
    "uvghip_intra_search_batch": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp]),
    "uvghip_intra_search_best_batch": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]),
    "uvghip_intra_pred_plane_batch": (c_int, [c_int, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_vp]),
    "uvghip_intra_pred_plane_chroma_batch": (c_int, [c_int, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_vp]),
    "uvghip_intra_search_best_stacked_batch": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_vp]),
    "uvghip_intra_pred_plane_stacked_batch": (c_int, [c_int, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "uvghip_intra_select_best": (c_int, [c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp]),
    "uvghip_mc_batch": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_int, c_vp, c_vp]),
    "uvghip_extended_block_batch": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int] + [c_int] * 7 + [c_vp, c_int, c_vp, c_vp]),
    "uvghip_frac_satd_batch": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp]),
    "uvghip_bipred_average_batch": (c_int, [c_int, c_vp, c_vp, c_int, ctypes.c_size_t, c_vp, c_vp]),
    "uvg_strategy_register_sao_hip": (c_int, [c_vp, ctypes.c_uint8]),
    "uvg_strategy_register_ipol_hip": (c_int, [c_vp, ctypes.c_uint8]),
    "uvghip_sao_stats_batch": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp]),
    "uvghip_sao_apply_batch": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp]),
    "uvghip_sao_edge_offsets_batch": (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
    "uvghip_deblock_frame": (c_int, [c_int, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "uvghip_deblock_frame_sao_snapshot": (c_int, [c_int, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "uvghip_lfnst_batch": (c_int, [c_int, c_vp, c_int, c_int, c_vp, c_int, c_vp]),
    "uvghip_alf_classify_frame": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp]),
    "uvghip_alf_filter_batch": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_vp]),
    "uvghip_alf_stats_batch": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]),
    "uvghip_crc32c_batch": (c_int, [c_int, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_vp]),
    "uvghip_pixel_var_batch": (c_int, [c_int, c_vp, ctypes.c_uint32, c_int, c_vp, c_vp]),
    "uvghip_sad_batch": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp]),
    "uvghip_satd_batch": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp]),
    "uvghip_ssd_batch": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp]),
    "uvghip_sad_surface": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "uvghip_alf_stats_compact_batch": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp]),
    "uvghip_alf_cov_expand": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "uvghip_alf_cov_reduce": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    "uvghip_band_plan": (c_int, [c_int, c_int, c_int, c_vp]),
    "uvghip_deblock_band": (c_int, [c_int, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp,
                                    c_int, c_int, c_int, c_vp]),
    "uvghip_cc_alf_filter_batch": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp]),
    "uvghip_slice_rows_alf_workspace_bytes": (ctypes.c_size_t, [c_int]),
    "uvghip_encode_slice_rows_alf": (c_int, [c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    "uvghip_write_idr_nals_alf": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp, ctypes.c_size_t, c_vp, c_int, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "uvghip_cc_alf_stats_batch": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]),
    "uvghip_alf_expand_tables": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "uvghip_alf_reconstruct_workspace_bytes": (ctypes.c_size_t, [c_int, c_int]),
    "uvghip_alf_reconstruct_picture": (c_int, [c_int, c_vp, c_vp, c_vp]),
    "uvghip_alf_classify_band": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_int, c_int, c_vp]),
    "uvghip_sao_decide_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "uvghip_sao_decide_pictures": (c_int, [c_int, c_int, c_int, c_int, c_int, ctypes.c_double, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                           c_vp, c_vp, c_vp, c_vp, c_vp]),
    "uvghip_sao_decide_pictures_slice": (c_int, [c_int, c_int, c_int, c_int, c_int, ctypes.c_double, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                 c_vp, c_vp, c_vp, c_vp, c_vp]),
    "uvghip_ctu_search_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "uvghip_ctu_search_intra": (c_int, [c_int, c_vp, c_vp, c_int, c_vp, c_vp]),
    "uvghip_ctu_plan_create": (c_int, [c_int, c_vp, c_vp, c_int, c_vp, c_vp]),
    "uvghip_ctu_plan_create_rows": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "uvghip_ctu_plan_run": (c_int, [c_vp, c_vp]),
    "uvghip_ctu_plan_destroy": (None, [c_vp]),
    "uvghip_slice_rows_workspace_bytes": (ctypes.c_size_t, [c_int]),
    "uvghip_slice_rows_prepare": (c_int, [c_vp, c_vp, c_int, c_vp]),
    "uvghip_loop_plan_slice_data": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    "uvghip_encode_slice_rows": (c_int, [c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    "uvghip_loop_plan_picture_nals": (c_int, [c_vp, c_int, c_int, c_vp, ctypes.c_size_t, c_vp, c_vp]),
    "uvghip_loop_plan_group_nals": (c_int, [c_vp, c_int, c_vp, ctypes.c_size_t, c_vp, c_vp]),
    "uvghip_tile_grid": (c_int, [c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "uvghip_tiles_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "uvghip_tiles_plan_create": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "uvghip_tiles_plan_run": (c_int, [c_vp, c_vp]),
    "uvghip_tiles_plan_layout": (c_int, [c_vp, c_vp, c_vp, c_vp]),
    "uvghip_tiles_plan_tile": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "uvghip_tiles_plan_nals": (c_int, [c_vp, c_int, c_int, c_int, c_vp, ctypes.c_size_t, c_vp, c_vp]),
    "uvghip_tiles_plan_destroy": (None, [c_vp]),
    "uvghip_frame_pool_create": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_vp]),
    "uvghip_frame_pool_create_tiles": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp]),
    "uvghip_frame_pool_begin": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int]),
    "uvghip_frame_pool_finish": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "uvghip_frame_pool_destroy": (None, [c_vp]),
    "uvghip_tiles_workspace_bytes_owned": (ctypes.c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "uvghip_tiles_plan_create_owned": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp]),
    "uvghip_tile_grid_split": (c_int, [c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp]),
    "uvghip_tiles_workspace_bytes_split": (ctypes.c_size_t, [c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp]),
    "uvghip_tiles_plan_create_split": (c_int, [c_int, c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp]),
    "uvghip_tiles_plan_substreams": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, ctypes.c_size_t, c_vp, c_vp, c_vp]),
    "uvghip_picture_checksum_rect": (c_int, [c_int, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "uvghip_loop_plan_alf_workspace_bytes": (ctypes.c_size_t, [c_vp]),
    "uvghip_loop_plan_alf_stage": (c_int, [c_vp, ALF_DECIDE_FN, c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    "uvghip_picture_checksum": (c_int, [c_int, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "uvghip_write_picture_nals": (c_int, [c_int, c_int, c_vp, ctypes.c_size_t, c_vp, c_int, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "uvghip_write_idr_nals": (c_int, [c_int, c_int, c_int, c_vp, ctypes.c_size_t, c_vp, c_int, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "uvghip_write_picture_nals_pb": (c_int, [c_int, c_int, c_int, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, ctypes.c_size_t, c_vp, c_int, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "uvghip_write_picture_nals_gop": (c_int, [c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, ctypes.c_size_t, c_vp, c_int, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "uvghip_write_idr_nals_ra": (c_int, [c_int, c_int, c_int, c_int, c_vp, ctypes.c_size_t, c_vp, c_int, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "uvghip_write_picture_nals_ra": (c_int, [c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, ctypes.c_size_t, c_vp, c_int, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "uvghip_slice_rows_pb_workspace_bytes": (ctypes.c_size_t, [c_int]),
    "uvghip_encode_slice_rows_pb": (c_int, [c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    "uvghip_ctu_search_pb_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "uvghip_ctu_search_pb": (c_int, [c_int, c_vp, c_int, c_vp, c_vp]),
    "uvghip_loop_pb_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int, c_int]),
    "uvghip_loop_pb_run": (c_int, [c_int, c_vp, c_int, c_int, c_vp, c_vp]),
    "uvghip_loop_pb_results": (c_int, [c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "uvghip_filter_pictures_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "uvghip_filter_pictures_prepare": (c_int, [c_int, c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "uvghip_filter_pictures_run": (c_int, [c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "uvghip_ctu_search_pb_inflight_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "uvghip_ctu_search_pb_inflight": (c_int, [c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    "uvghip_loop_pb_run_inflight_ext": (c_int, [c_int, c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp]),
    "uvghip_ctu_search_pb_inflight_ext": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    "uvghip_loop_plan_search_reset": (c_int, [c_vp, c_vp]),
    "uvghip_loop_plan_search_launch": (c_int, [c_vp, c_vp]),
    "uvghip_loop_plan_set_search_grid": (c_int, [c_vp, c_int]),
    "uvghip_loop_plan_searched_flags": (c_vp, [c_vp]),
    "uvghip_loop_plan_run_coder": (c_int, [c_vp, c_vp]),
    "uvghip_loop_plan_run_coder_behind": (c_int, [c_vp, c_vp, c_vp]),
    "uvghip_encode_slice_rows_behind": (c_int, [c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    "uvghip_loop_pb_inflight_final_flags": (c_vp, [c_int, c_int, c_int, c_int, c_vp]),
    "uvghip_ctu_search_pb_inflight_final_flags": (c_vp, [c_int, c_int, c_int, c_vp]),
    "uvghip_ctu_plan_reset": (c_int, [c_vp, c_vp]),
    "uvghip_ctu_plan_launch": (c_int, [c_vp, c_vp]),
    "uvghip_ctu_plan_set_grid": (c_int, [c_vp, c_int]),
    "uvghip_ctu_plan_done_flags": (c_vp, [c_vp]),
    "uvghip_loop_pb_inflight_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int, c_int]),
    "uvghip_loop_pb_run_inflight": (c_int, [c_int, c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "uvghip_loop_pb_inflight_results": (c_int, [c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "uvghip_merge_cand_batch": (c_int, [c_vp, c_vp, c_vp, ctypes.c_long, c_vp, c_int, c_vp, c_vp, c_vp]),
    "uvghip_amvp_cand_batch": (c_int, [c_vp, c_vp, c_vp, ctypes.c_long, c_vp, c_int, c_vp, c_vp]),
    "uvghip_inter_pred_satd_batch": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_vp]),
    "uvghip_me_search_batch": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, ctypes.c_double, c_int, c_int, c_vp, c_int, c_vp, c_vp]),
    "uvghip_loop_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int, c_int]),
    "uvghip_loop_plan_create": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    "uvghip_loop_plan_run": (c_int, [c_vp, c_vp]),
    "uvghip_loop_plan_run_overlapped": (c_int, [c_vp, c_vp]),
    "uvghip_encode_slice_rows_behind_capped": (c_int, [c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_vp]),
    "uvghip_filter_pictures_reset": (c_int, [c_int, c_int, c_int, c_vp, c_vp]),
    "uvghip_filter_pictures_run_behind": (c_int, [c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp]),
    "uvghip_filter_pictures_final_flags": (c_vp, [c_int, c_int, c_int, c_vp]),
    "uvghip_loop_plan_run_search": (c_int, [c_vp, c_vp]),
    "uvghip_loop_plan_run_filters": (c_int, [c_vp, c_vp]),
    "uvghip_loop_plan_results": (c_int, [c_vp, c_vp, c_vp]),
    "uvghip_loop_plan_destroy": (None, [c_vp]),
    "uvghip_comm_unique_id": (c_int, [c_vp]),
    "uvghip_comm_create": (c_int, [c_vp, c_int, c_int, c_vp]),
    "uvghip_comm_destroy": (c_int, [c_vp]),
    "uvghip_comm_exchange": (c_int, [c_vp, c_vp, c_int, c_vp]),
    "uvghip_comm_allreduce_i64": (c_int, [c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "uvghip_quant_cbcr_residual_workspace_bytes": (ctypes.c_size_t, [c_vp, c_int]),
    "uvghip_quant_cbcr_residual_batch": (c_int, [c_int, c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_int,
                                                 c_vp, c_vp, c_vp, c_int, c_vp, ctypes.c_size_t, c_vp]),
    "uvghip_coeff_cost_batch": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "uvghip_quant_percall": (ctypes.c_uint, [c_vp, c_vp, c_vp, ctypes.c_int32, ctypes.c_int32] + [c_int] * 5),
    "uvghip_dequant_percall": (ctypes.c_uint, [c_vp, c_vp, c_vp, ctypes.c_int32, ctypes.c_int32] + [c_int] * 3),
    "uvghip_quantize_residual_percall": (c_int, [c_vp, c_vp] + [c_int] * 7 + [c_vp] * 4 + [c_int] * 3),
    "uvghip_quant_cbcr_residual_percall": (c_int, [c_vp, c_vp] + [c_int] * 5 + [c_vp] * 7 + [c_int] * 3),
    "uvghip_bipred_average_percall": (None, [c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, ctypes.c_uint, ctypes.c_uint]),
    "uvghip_alf_classify_percall": (c_int, [c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "uvghip_alf_filter_percall": (c_int, [c_int, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp]),
    "uvghip_alf_stats_percall": (c_int, [c_int, c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_vp]),
    "uvghip_comm_allgather": (c_int, [c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "uvghip_residual_plane": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp]),
}


def load_library():
    """dlopen the in-tree library and declare every signature.  No device needed."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)")
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def init(device=0):
    """uvghip_init on `device`; raises DeviceMissing when there is no gfx950 GPU."""
    global _inited_device
    lib = load_library()
    if _inited_device == device:
        return lib
    rc = lib.uvghip_init(device)
    if rc != 0:
        raise DeviceMissing(f"uvghip_init({device}) failed: {lib.uvghip_last_error().decode()}")
    _inited_device = device
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {load_library().uvghip_last_error().decode()}")
