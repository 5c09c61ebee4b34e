// The picture-level NAL units that depend on what the hot path produced: the slice NAL (its header carries the entry points = the
// lengths of the WPP rows' substreams) and the decoded-picture-hash SEI (a checksum of the picture after the in-loop filters).
// With them, the bytes behind the parameter sets of a one-picture all-intra .266 come entirely from device outputs
// (tests/test_picture_nal.py: the encoder's whole file = its SPS / PPS / version-SEI bytes + what this file writes).
//
// replaces, for an IDR picture of an all-intra (-p 1) stream with WPP on, one slice, picture header in the slice header, no ALF /
// LMCS / dependent quantisation / sign hiding / transform skip (what --preset medium configures):
//   uvg_encoder_state_write_bitstream_slice_header + _picture_header (src/encoder_state-bitstream.c:1009-1139, 1248-1411),
//   encoder_state_write_bitstream_entry_points_write (:993-1007), uvg_nal_write (src/nal.c:43-74), add_checksum (:1420-1477) with
//   uvg_image_checksum / array_checksum_generic (src/nal.c:91-115, src/strategies/generic/nal-generic.c:68-92), and the emulation
//   prevention of uvg_bitstream_put_byte (src/bitstream.c:215-226) on the header and SEI bytes.
// The parameter sets and the version SEI are the encoder's (control plane): they do not depend on the picture.
#include "uvghip_common.h"
#include <cstring>

namespace {

// sum over the samples of (low byte ^ mask) [+ (high byte ^ mask)], mask = (x ^ y ^ x >> 8 ^ y >> 8) & 0xff, modulo 2^32
template <typename PX>
__global__ void __launch_bounds__(256) checksum_kernel(const PX *__restrict__ plane, int stride, int x0, int y0, int width, int height, uint32_t *__restrict__ sum)
{
  uint32_t acc = 0;
  const size_t n = (size_t)width * height;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int ry = (int)(i / width), y = y0 + ry, x = x0 + (int)(i - (size_t)ry * width);
    const uint32_t v = plane[(size_t)y * stride + x];
    const uint32_t mask = (uint32_t)((x & 0xff) ^ (y & 0xff) ^ (x >> 8) ^ (y >> 8)) & 0xffu;
    acc += (v & 0xffu) ^ mask;
    if (sizeof(PX) == 2) acc += ((v >> 8) & 0xffu) ^ mask;
  }
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(sum, acc);
}

// bytes of one NAL unit's payload with the reference's emulation prevention: 00 00 0x (x < 4) gets a 03 in front of x
struct nal_writer {
  uint8_t *out;
  size_t cap, n;
  int zeros;
  uint32_t acc;       // bits not yet a whole byte, MSB first
  int nacc;
  void raw(uint8_t b) { if (n < cap) out[n] = b; ++n; }
  void byte(uint8_t b)
  {
    if (zeros == 2 && b < 4) { raw(3); zeros = 0; }
    zeros = b == 0 ? zeros + 1 : 0;
    raw(b);
  }
  void bits(uint32_t v, int len)
  {
    for (int i = len - 1; i >= 0; --i) {
      acc = acc << 1 | ((v >> i) & 1u);
      if (++nacc == 8) { byte((uint8_t)acc); acc = 0; nacc = 0; }
    }
  }
  void ue(uint32_t v)
  {
    int len = 0;
    for (uint32_t t = v + 1; t > 1; t >>= 1) ++len;
    bits(0, len);
    bits(v + 1, len + 1);
  }
  void se(int v) { ue(v > 0 ? 2u * (uint32_t)v - 1u : 2u * (uint32_t)(-v)); }
  void align_with_one() { bits(1, 1); while (nacc) bits(0, 1); }
  void start(int nal_type, bool long_code = false)       // start code, two header bytes: none of them counts towards the emulation prevention
  {
    if (long_code) raw(0);
    raw(0); raw(0); raw(1); raw(0); raw((uint8_t)(nal_type << 3 | 1));
  }
};

enum { NAL_TRAIL = 0, NAL_RASL = 3, NAL_IDR_W_RADL = 7, NAL_IDR_N_LP = 8, NAL_CRA = 9, NAL_PREFIX_APS = 17, NAL_SUFFIX_SEI = 24 };

int ceil_log2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
int floor_log2(int v) { int l = 0; while ((2 << l) <= v) ++l; return l; }

// encode_alf_aps_filter (alf.c:1415-1450): the coefficients of a filter set as ue(|c|) + sign, then two bits per clip index
void write_alf_filters(nal_writer &w, const int16_t *coeff, const int16_t *clipp, int n_filters, int n_coeff, int non_linear)
{
  for (int f = 0; f < n_filters; ++f)
    for (int i = 0; i < n_coeff - 1; ++i) {
      const int c = coeff[f * 13 + i];
      w.ue((uint32_t)(c < 0 ? -c : c));
      if (c) w.bits(c < 0, 1);
    }
  if (non_linear)
    for (int f = 0; f < n_filters; ++f)
      for (int i = 0; i < n_coeff - 1; ++i) w.bits((uint32_t)clipp[f * 13 + i], 2);
}

// one ALF APS NAL unit: encoder_state_write_adaptation_parameter_set + encode_alf_aps_flags (alf.c:1547-1573, 1452-1545), 4:2:0
void write_alf_aps(nal_writer &w, const uvghip_alf_aps_t &a, int alf_full, bool long_code)
{
  w.zeros = 0;
  w.start(NAL_PREFIX_APS, long_code);
  w.bits(0, 3);                      // aps_params_type: ALF
  w.bits((uint32_t)a.aps_id, 5);     // adaptation_parameter_set_id
  w.bits(1, 1);                      // aps_chroma_present_flag
  w.bits(a.new_filter[0] != 0, 1);   // alf_luma_new_filter
  w.bits(a.new_filter[1] != 0, 1);   // alf_chroma_new_filter
  w.bits(alf_full && a.new_cc_filter[0], 1);      // alf_cc_cb_filter_signal_flag
  w.bits(alf_full && a.new_cc_filter[1], 1);      // alf_cc_cr_filter_signal_flag
  if (a.new_filter[0]) {
    w.bits(a.non_linear[0] != 0, 1);              // alf_luma_clip
    w.ue((uint32_t)a.num_luma_filters - 1);
    if (a.num_luma_filters > 1) {
      const int length = ceil_log2(a.num_luma_filters);
      for (int i = 0; i < 25; ++i) w.bits((uint32_t)a.luma[2 * 325 + i], length);          // alf_luma_coeff_delta_idx
    }
    write_alf_filters(w, a.luma, a.luma + 325, a.num_luma_filters, 13, a.non_linear[0]);
  }
  if (a.new_filter[1]) {
    w.bits(a.non_linear[1] != 0, 1);
    w.ue((uint32_t)a.num_alternatives_chroma - 1);
    for (int t = 0; t < a.num_alternatives_chroma; ++t) {
      // (the chroma arrays are [alternative][7]: one "filter" of seven at a time)
      const int16_t *co = a.chroma + t * 7, *cl = a.chroma + (8 + t) * 7;
      for (int i = 0; i < 6; ++i) { const int c = co[i]; w.ue((uint32_t)(c < 0 ? -c : c)); if (c) w.bits(c < 0, 1); }
      if (a.non_linear[1]) for (int i = 0; i < 6; ++i) w.bits((uint32_t)cl[i], 2);
    }
  }
  if (alf_full)
    for (int c = 0; c < 2; ++c) {
      if (!a.new_cc_filter[c]) continue;
      w.ue((uint32_t)a.cc_filter_count[c] - 1);
      for (int f = 0; f < a.cc_filter_count[c]; ++f)
        for (int i = 0; i < 7; ++i) {
          const int v = a.cc[(c * 4 + f) * 8 + i];
          if (v == 0) w.bits(0, 3);
          else { w.bits((uint32_t)(1 + floor_log2(v < 0 ? -v : v)), 3); w.bits(v < 0, 1); }
        }
    }
  w.bits(0, 1);                      // aps_extension_flag
  w.align_with_one();                // rbsp_trailing_bits
}

// entry points, byte alignment, the rows' substreams, then the decoded picture hash SEI (shared by both writers)
void finish_picture(nal_writer &w, uint8_t *out, size_t cap, const uint8_t *rows, size_t row_pitch, const int32_t *row_bytes, int n_rows, int32_t longest,
                    const uint32_t *checksum)
{
  if (n_rows > 1) {                  // entry points: every row but the last, in offset_len bits each (:1386-1404)
    int offset_len = 0;
    for (int32_t t = longest; t; t >>= 1) ++offset_len;
    w.ue((uint32_t)offset_len - 1);
    for (int r = 0; r + 1 < n_rows; ++r) w.bits((uint32_t)row_bytes[r] - 1, offset_len);
  }
  w.align_with_one();
  for (int r = 0; r < n_rows; ++r) {
    const size_t nb = (size_t)row_bytes[r];
    if (w.n + nb <= cap) memcpy(out + w.n, rows + (size_t)r * row_pitch, nb);
    w.n += nb;
  }
  if (checksum) {
    w.zeros = 0;                     // (the last row ends with its stop bit: the zero run does not reach across)
    w.start(NAL_SUFFIX_SEI);
    w.bits(132, 8);                  // payload type: decoded picture hash
    w.bits(2 + 3 * 4, 8);            // payload size
    w.bits(2, 8);                    // hash type: checksum
    w.bits(0, 8);                    // dph_sei_single_component_flag = 0, seven reserved bits
    for (int c = 0; c < 3; ++c) w.bits(checksum[c], 32);
    w.align_with_one();              // (already aligned: the trailing bits are a byte of their own)
  }
}

}  // namespace

extern "C" int uvghip_picture_checksum(int bitdepth, const void *plane_y, int stride_y, const void *plane_u, const void *plane_v, int stride_c,
                                       int width, int height, uint32_t *sums, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (!plane_y || !plane_u || !plane_v || !sums || width <= 0 || height <= 0 || (width & 1) || (height & 1) || stride_y < width || stride_c < width / 2)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  hipStream_t st = uvghip_stream(stream);
  UVGHIP_TRY(hipMemsetAsync(sums, 0, 3 * sizeof(uint32_t), st));
  return uvghip_picture_checksum_rect(bitdepth, plane_y, stride_y, plane_u, plane_v, stride_c, 0, 0, width, height, sums, stream);
}

// ... of a rectangle of the picture, ADDED to sums: the checksum is a sum over samples of a function of the sample and its position in the
// PICTURE, so the sums of rectangles that tile the picture add up to the picture's (tiles on several devices: an all-reduce of three words).
extern "C" int uvghip_picture_checksum_rect(int bitdepth, const void *plane_y, int stride_y, const void *plane_u, const void *plane_v, int stride_c,
                                            int x0, int y0, int width, int height, uint32_t *sums, void *stream)
{
  UVGHIP_REQUIRE_READY();
  UVGHIP_REQUIRE_DEPTH(bitdepth);
  if (!plane_y || !plane_u || !plane_v || !sums || width <= 0 || height <= 0 || x0 < 0 || y0 < 0 || ((width | height | x0 | y0) & 1) || stride_y < x0 + width ||
      stride_c < (x0 + width) / 2)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  hipStream_t st = uvghip_stream(stream);
  const void *planes[3] = {plane_y, plane_u, plane_v};
  for (int c = 0; c < 3; ++c) {
    const int w = c ? width / 2 : width, h = c ? height / 2 : height, stride = c ? stride_c : stride_y, ox = c ? x0 / 2 : x0, oy = c ? y0 / 2 : y0;
    const size_t n = (size_t)w * h;
    const int blocks = (int)((n + 256 * 16 - 1) / (256 * 16) < 2048 ? (n + 256 * 16 - 1) / (256 * 16) : 2048);
    if (bitdepth == 8) checksum_kernel<uint8_t><<<blocks, 256, 0, st>>>(static_cast<const uint8_t *>(planes[c]), stride, ox, oy, w, h, sums + c);
    else checksum_kernel<uint16_t><<<blocks, 256, 0, st>>>(static_cast<const uint16_t *>(planes[c]), stride, ox, oy, w, h, sums + c);
  }
  UVGHIP_CHECK_LAUNCH();
}

// Host function (no device needed): rows / row_bytes / checksum are HOST memory.
extern "C" int uvghip_write_picture_nals(int poc, int sao, const uint8_t *rows, size_t row_pitch, const int32_t *row_bytes, int n_rows,
                                         const uint32_t *checksum, uint8_t *out, size_t cap, size_t *len)
{
  return uvghip_write_idr_nals(poc, 0, sao, rows, row_pitch, row_bytes, n_rows, checksum, out, cap, len);
}
// ... with the slice QP offset (sh_qp_delta = state->frame->QP - cfg.qp: the intra QP offset of a low-delay stream's first picture)
extern "C" int uvghip_write_idr_nals(int poc, int qp_delta, int sao, const uint8_t *rows, size_t row_pitch, const int32_t *row_bytes, int n_rows,
                                     const uint32_t *checksum, uint8_t *out, size_t cap, size_t *len)
{
  return uvghip_write_idr_nals_ra(poc, 4, qp_delta, sao, rows, row_pitch, row_bytes, n_rows, checksum, out, cap, len);
}
// ... and with the POC width of the stream's SPS (encoder_control->poc_lsb_bits, src/encoder.c:242: 4 without a GOP structure or with a
// low-delay one of up to 7 pictures, 6 for --gop 16)
extern "C" int uvghip_write_idr_nals_ra(int poc, int poc_lsb_bits, int qp_delta, int sao, const uint8_t *rows, size_t row_pitch, const int32_t *row_bytes, int n_rows,
                                        const uint32_t *checksum, uint8_t *out, size_t cap, size_t *len)
{
  if (!rows || !row_bytes || n_rows <= 0 || !len || (!out && cap) || poc < 0 || poc_lsb_bits < 4 || poc_lsb_bits > 16)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  int32_t longest = 0;
  for (int r = 0; r < n_rows; ++r) {
    if (row_bytes[r] <= 0 || (size_t)row_bytes[r] > row_pitch) return uvghip_set_error(hipErrorInvalidValue, "uvghip_write_picture_nals: a row is empty or longer than its slot");
    if (row_bytes[r] > longest) longest = row_bytes[r];
  }
  nal_writer w = {out, cap, 0, 0, 0, 0};
  // ---- the slice: header (picture header inside), byte alignment, the rows' substreams as they are ----
  // picture 0 of the stream follows the parameter sets in its access unit: IDR_N_LP, short start code; every later picture of a
  // -p 1 stream is IDR_W_RADL (src/encoderstate.c:1965-1966) and the first NAL unit of its access unit: long start code
  // (src/encoder_state-bitstream.c:1519-1531)
  if (poc == 0) w.start(NAL_IDR_N_LP); else w.start(NAL_IDR_W_RADL, true);
  w.bits(1, 1);                      // sh_picture_header_in_slice_header_flag
  w.bits(1, 1);                      // ph_gdr_or_irap_pic_flag
  w.bits(0, 1);                      // ph_non_ref_pic_flag
  w.bits(0, 1);                      // ph_gdr_pic_flag
  w.bits(0, 1);                      // ph_inter_slice_allowed_flag
  w.ue(0);                           // ph_pic_parameter_set_id
  w.bits((uint32_t)poc & ((1u << poc_lsb_bits) - 1u), poc_lsb_bits);      // ph_pic_order_cnt_lsb
  w.bits(0, 1);                      // sh_no_output_of_prior_pics_flag
  w.se(qp_delta);                    // sh_qp_delta
  if (sao) w.bits(3, 2);             // sh_sao_luma_used_flag, sh_sao_chroma_used_flag
  finish_picture(w, out, cap, rows, row_pitch, row_bytes, n_rows, longest, checksum);
  *len = w.n;
  if (w.n > cap) return uvghip_set_error(hipErrorInvalidValue, "uvghip_write_picture_nals: the output buffer is too small (see *len)");
  return 0;
}

// ... of an --alf on / --alf full stream: the APS NAL units the picture is preceded by (uvg_encode_alf_adaptive_parameter_set, alf.c:1610; written
// between the parameter sets / the start of the access unit and the slice, encoder_state-bitstream.c:1562) and the slice header's ALF
// fields (:1283-1330).  Host function.
extern "C" int uvghip_write_idr_nals_alf(int poc, int qp_delta, int sao, const uvghip_alf_slice_t *alf, const uvghip_alf_aps_t *aps, int n_aps, const uint8_t *rows,
                                         size_t row_pitch, const int32_t *row_bytes, int n_rows, const uint32_t *checksum, uint8_t *out, size_t cap, size_t *len)
{
  if (!rows || !row_bytes || n_rows <= 0 || !len || (!out && cap) || poc < 0 || !alf || n_aps < 0 || (n_aps && !aps) || (alf->alf_type != 1 && alf->alf_type != 2) ||
      alf->n_luma_aps < 0 || alf->n_luma_aps > 7)
    return uvghip_set_error(hipErrorInvalidValue, __func__);
  for (int i = 0; i < n_aps; ++i)
    if (aps[i].aps_id < 0 || aps[i].aps_id > 7 || (aps[i].new_filter[0] && (!aps[i].luma || aps[i].num_luma_filters < 1 || aps[i].num_luma_filters > 25)) ||
        (aps[i].new_filter[1] && (!aps[i].chroma || aps[i].num_alternatives_chroma < 1 || aps[i].num_alternatives_chroma > 8)) ||
        ((aps[i].new_cc_filter[0] || aps[i].new_cc_filter[1]) && !aps[i].cc) || aps[i].cc_filter_count[0] > 4 || aps[i].cc_filter_count[1] > 4)
      return uvghip_set_error(hipErrorInvalidValue, "uvghip_write_idr_nals_alf: a parameter set");
  int32_t longest = 0;
  for (int r = 0; r < n_rows; ++r) {
    if (row_bytes[r] <= 0 || (size_t)row_bytes[r] > row_pitch) return uvghip_set_error(hipErrorInvalidValue, "uvghip_write_idr_nals_alf: a row is empty or longer than its slot");
    if (row_bytes[r] > longest) longest = row_bytes[r];
  }
  nal_writer w = {out, cap, 0, 0, 0, 0};
  bool first_nal = poc != 0;         // picture 0 follows the parameter sets in its access unit; later pictures open theirs (long start code on the first NAL unit)
  for (int i = 0; i < n_aps; ++i) { write_alf_aps(w, aps[i], alf->alf_type == 2, first_nal); first_nal = false; }
  w.zeros = 0;
  if (poc == 0) w.start(NAL_IDR_N_LP); else w.start(NAL_IDR_W_RADL, first_nal);
  w.bits(1, 1);                      // sh_picture_header_in_slice_header_flag
  w.bits(1, 1);                      // ph_gdr_or_irap_pic_flag
  w.bits(0, 1);                      // ph_non_ref_pic_flag
  w.bits(0, 1);                      // ph_gdr_pic_flag
  w.bits(0, 1);                      // ph_inter_slice_allowed_flag
  w.ue(0);                           // ph_pic_parameter_set_id
  w.bits((uint32_t)poc & 15u, 4);    // ph_pic_order_cnt_lsb
  w.bits(0, 1);                      // sh_no_output_of_prior_pics_flag
  w.bits(alf->enabled[0] != 0, 1);   // slice_alf_enabled_flag
  if (alf->enabled[0]) {
    w.bits((uint32_t)alf->n_luma_aps, 3);
    for (int i = 0; i < alf->n_luma_aps; ++i) w.bits((uint32_t)alf->luma_aps_id[i], 3);
    w.bits(alf->enabled[1] != 0, 1); w.bits(alf->enabled[2] != 0, 1);
    if (alf->enabled[1] || alf->enabled[2]) w.bits((uint32_t)alf->chroma_aps_id, 3);
    if (alf->alf_type == 2)
      for (int c = 0; c < 2; ++c) {
        w.bits(alf->cc_enabled[c] != 0, 1);
        if (alf->cc_enabled[c]) w.bits((uint32_t)alf->cc_aps_id[c], 3);
      }
  }
  w.se(qp_delta);                    // sh_qp_delta
  if (sao) w.bits(3, 2);             // sh_sao_luma_used_flag, sh_sao_chroma_used_flag
  finish_picture(w, out, cap, rows, row_pitch, row_bytes, n_rows, longest, checksum);
  *len = w.n;
  if (w.n > cap) return uvghip_set_error(hipErrorInvalidValue, "uvghip_write_idr_nals_alf: the output buffer is too small (see *len)");
  return 0;
}

// The same for a P / B picture (TRAIL pictures, temporal id 0: uvg266 only signals a sub-layer for STSA pictures, :1484): the picture
// header's inter flags, the slice type, the reference picture list syntax, the collocated picture, the slice QP offset.
// replaces: uvg_encoder_state_write_bitstream_slice_header (:1248-1411) with _picture_header (:1009-1139) and _ref_pic_list (:1141-1246)
// for pictype TRAIL.  List 0 carries the references in the past, list 1 those in the future (gop_len != 0 && !gop_lowdelay), or is
// signalled as a copy of list 0's entries (copy_rpl1 = (gop_lowdelay || !gop_len) && bipred, :1165).  delta_neg / delta_pos: the POC
// distance of each reference picture, in the order of the GOP structure's ref_neg[] / ref_pos[] (src/gop.h, uvg_config_process_lp_gop
// src/cfg.c:1640-1720: ascending) restricted to the pictures that are in the reference buffer (:1176-1190).
static int write_inter_picture(const char *who, int nal_type, int poc, int poc_lsb_bits, int slice_type, int n_ref_neg, const int32_t *delta_neg, int n_ref_pos, const int32_t *delta_pos,
                               int copy_rpl1, int tmvp, int qp_delta, int sao, const uint8_t *rows, size_t row_pitch, const int32_t *row_bytes, int n_rows,
                               const uint32_t *checksum, uint8_t *out, size_t cap, size_t *len)
{
  // (a CRA picture is an I slice with the reference buffer of the pictures that follow it in its lists; every other picture here is P or B)
  if (!rows || !row_bytes || n_rows <= 0 || !len || (!out && cap) || poc < 0 || poc_lsb_bits < 4 || poc_lsb_bits > 16 ||
      (nal_type == NAL_CRA ? slice_type != 2 : (slice_type != 0 && slice_type != 1)) || (nal_type != NAL_TRAIL && nal_type != NAL_RASL && nal_type != NAL_CRA) ||
      n_ref_neg < 0 || n_ref_neg > 15 || n_ref_pos < 0 || n_ref_pos > 15 || (n_ref_neg + n_ref_pos < 1 && slice_type != 2) || (n_ref_neg && !delta_neg) || (n_ref_pos && !delta_pos) ||
      (copy_rpl1 && n_ref_pos))
    return uvghip_set_error(hipErrorInvalidValue, who);
  // the distances of a list: positive and ascending, as the GOP structures list them (a reference at distance 0 is the picture itself)
  for (int list = 0; list < 2; ++list) {
    const int32_t *d = list ? delta_pos : delta_neg;
    const int n = list ? n_ref_pos : n_ref_neg;
    for (int j = 0; j < n; ++j)
      if (d[j] < 1 || (j && d[j] <= d[j - 1]) || (!list && d[j] > poc)) return uvghip_set_error(hipErrorInvalidValue, who);          // (reference distances must be positive, ascending and inside the stream: a no-GOP stream passes 1, 2, 3, ...)
  }
  int32_t longest = 0;
  for (int r = 0; r < n_rows; ++r) {
    if (row_bytes[r] <= 0 || (size_t)row_bytes[r] > row_pitch) return uvghip_set_error(hipErrorInvalidValue, who);          // (a row is empty or longer than its slot)
    if (row_bytes[r] > longest) longest = row_bytes[r];
  }
  nal_writer w = {out, cap, 0, 0, 0, 0};
  w.start(nal_type, true);           // the first NAL unit of its access unit: long start code; temporal id 0
  w.bits(1, 1);                      // sh_picture_header_in_slice_header_flag
  w.bits(0, 1);                      // ph_gdr_or_irap_pic_flag
  w.bits(0, 1);                      // ph_non_ref_pic_flag
  w.bits(1, 1);                      // ph_inter_slice_allowed_flag
  w.bits(1, 1);                      // ph_intra_slice_allowed_flag
  w.ue(0);                           // ph_pic_parameter_set_id
  w.bits((uint32_t)poc & ((1u << poc_lsb_bits) - 1u), poc_lsb_bits);
  if (tmvp) w.bits(1, 1);            // ph_pic_temporal_mvp_enabled_flag
  w.bits(0, 1);                      // ph_mvd_l1_zero_flag
  w.ue((uint32_t)slice_type);        // sh_slice_type
  if (nal_type == NAL_CRA) w.bits(0, 1);             // sh_no_output_of_prior_pics_flag (:1278)
  const int lists = 1 + (copy_rpl1 ? 1 : 0);
  for (int list = 0; list < lists; ++list) {
    w.ue((uint32_t)n_ref_neg);       // num_ref_entries[0]
    int last = 0;
    for (int j = 0; j < n_ref_neg; ++j) {
      const int d = delta_neg[j];
      w.ue((uint32_t)(d - last - 1));               // abs_delta_poc_st (distances are >= 1 and ascending: checked above)
      w.bits(1, 1);                                  // strp_entry_sign_flag: in the past
      last = d;
    }
  }
  if (!copy_rpl1) {
    w.ue((uint32_t)n_ref_pos);       // num_ref_entries[1]
    int last = 0;
    for (int j = 0; j < n_ref_pos; ++j) {
      const int d = delta_pos[j];
      w.ue((uint32_t)(d - last - 1));               // abs_delta_poc_st
      w.bits(0, 1);                                  // strp_entry_sign_flag: in the future
      last = d;
    }
  }
  if ((slice_type != 2 && n_ref_neg > 1) || n_ref_pos > 1) {        // (:1237)
    w.bits(1, 1);                    // sh_num_ref_idx_active_override_flag
    if (n_ref_neg > 1) for (int list = 0; list < lists; ++list) w.ue((uint32_t)n_ref_neg - 1);
    if (!copy_rpl1 && n_ref_pos > 1) w.ue((uint32_t)n_ref_pos - 1);
  }
  if (tmvp && slice_type != 2) {
    if (slice_type == 0) w.bits(1, 1);               // sh_collocated_from_l0_flag
    if (n_ref_neg > 1) w.ue(0);                      // sh_collocated_ref_idx
  }
  w.se(qp_delta);                    // sh_qp_delta
  if (sao) w.bits(3, 2);             // sh_sao_luma_used_flag, sh_sao_chroma_used_flag
  finish_picture(w, out, cap, rows, row_pitch, row_bytes, n_rows, longest, checksum);
  *len = w.n;
  if (w.n > cap) return uvghip_set_error(hipErrorInvalidValue, "uvghip_write_picture_nals_pb / _ra: the output buffer is too small (see *len)");
  return 0;
}

// a picture of a low-delay stream (--gop lp-*, or no GOP structure): every reference in the past
extern "C" int uvghip_write_picture_nals_pb(int poc, int poc_lsb_bits, int slice_type, int n_ref_neg, const int32_t *delta_neg, int copy_rpl1, int tmvp, int qp_delta,
                                            int sao, const uint8_t *rows, size_t row_pitch, const int32_t *row_bytes, int n_rows, const uint32_t *checksum,
                                            uint8_t *out, size_t cap, size_t *len)
{
  if (n_ref_neg < 1) return uvghip_set_error(hipErrorInvalidValue, __func__);
  return write_inter_picture(__func__, NAL_TRAIL, poc, poc_lsb_bits, slice_type, n_ref_neg, delta_neg, 0, nullptr, copy_rpl1 != 0, tmvp, qp_delta, sao, rows, row_pitch, row_bytes, n_rows,
                             checksum, out, cap, len);
}

// a picture of a random-access stream (--gop 8 / 16, the hierarchical structures of src/gop.h): references in the past and in the future
extern "C" int uvghip_write_picture_nals_ra(int poc, int poc_lsb_bits, int slice_type, int n_ref_neg, const int32_t *delta_neg, int n_ref_pos, const int32_t *delta_pos,
                                            int tmvp, int qp_delta, int sao, const uint8_t *rows, size_t row_pitch, const int32_t *row_bytes, int n_rows,
                                            const uint32_t *checksum, uint8_t *out, size_t cap, size_t *len)
{
  return write_inter_picture(__func__, NAL_TRAIL, poc, poc_lsb_bits, slice_type, n_ref_neg, delta_neg, n_ref_pos, delta_pos, 0, tmvp, qp_delta, sao, rows, row_pitch, row_bytes, n_rows,
                             checksum, out, cap, len);
}

// ... of a random-access stream with more than one intra period and an open GOP (cfg.open_gop, the default): the I picture that opens a
// later period is a CRA picture (pictype, src/encoderstate.c:1957-1972) -- an I slice whose header still carries the reference picture lists
// (the buffer the following pictures use, :1326-1329) and sh_no_output_of_prior_pics_flag (:1278) --, the pictures before it in display
// order that are coded after it are RASL pictures (the same syntax as TRAIL under another NAL unit type).  nal_type: 0 TRAIL, 3 RASL, 9 CRA
// (slice_type 2 with CRA only).
extern "C" int uvghip_write_picture_nals_gop(int nal_type, int poc, int poc_lsb_bits, int slice_type, int n_ref_neg, const int32_t *delta_neg, int n_ref_pos,
                                             const int32_t *delta_pos, int tmvp, int qp_delta, int sao, const uint8_t *rows, size_t row_pitch, const int32_t *row_bytes, int n_rows,
                                             const uint32_t *checksum, uint8_t *out, size_t cap, size_t *len)
{
  return write_inter_picture(__func__, nal_type, poc, poc_lsb_bits, slice_type, n_ref_neg, delta_neg, n_ref_pos, delta_pos, 0, tmvp, qp_delta, sao, rows, row_pitch, row_bytes, n_rows,
                             checksum, out, cap, len);
}
