/* Dev-time only (tools/refcheck/run.sh): the reference's src/alf.c compiled INTO the tool, from where it lies under the reference tree,
 * so that its static functions can be called next to the oracle's restatements.  The library the tool links (oracle/_ref) holds alf.c's
 * seven public functions already: they are renamed in this translation unit.  Nothing of the reference is copied or modified. */
#define uvg_reset_cc_alf_aps_param rcdup_uvg_reset_cc_alf_aps_param
#define uvg_set_aps_map rcdup_uvg_set_aps_map
#define uvg_encode_alf_bits rcdup_uvg_encode_alf_bits
#define uvg_encode_alf_adaptive_parameter_set rcdup_uvg_encode_alf_adaptive_parameter_set
#define uvg_alf_create rcdup_uvg_alf_create
#define uvg_alf_destroy rcdup_uvg_alf_destroy
#define uvg_alf_enc_process rcdup_uvg_alf_enc_process
#include "alf.c"

void rc_get_blk_stats_cc_alf(encoder_state_t *const state, alf_covariance *cov, const uvg_picture *org_yuv, int comp_id, int x_pos, int y_pos, int width, int height)
{
  get_blk_stats_cc_alf(state, cov, org_yuv, (alf_component_id)comp_id, x_pos, y_pos, width, height);
}
