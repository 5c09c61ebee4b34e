#!/bin/bash
# Regenerates oracle/orc_ctx_init.h (and the device copy uvg266_amd/csrc/vvc_ctx_init.h, if present) from the reference's
# uvg_init_contexts, through oracle/_ref (oracle/build_ref.sh).
set -e
cd "$(dirname "$0")/../.."
REF=${UVG_REF_SRC:-/root/reference}
oracle/build_ref.sh "$REF" >/dev/null
gcc -O1 -std=gnu11 -w -Ioracle/_ref/gen -I$REF/src -I$REF/src/extras -I$REF/src/strategies tools/refcheck/gen_ctx_init.c oracle/_ref/libuvg266_8.a \
    -Wl,--wrap=uvg_search_lcu -Wl,--wrap=uvg_encode_coding_tree -Wl,--wrap=uvg_sao_search_lcu -Wl,--wrap=uvg_bitstream_put_byte -Wl,--wrap=uvg_cabac_finish \
    -Wl,--wrap=uvg_bitstream_align_zero -Wl,--wrap=uvg_inter_get_merge_cand -Wl,--wrap=uvg_inter_get_mv_cand -Wl,--wrap=uvg_search_cu_inter -Wl,--wrap=uvg_alf_enc_process -Wl,--wrap=uvg_encode_alf_adaptive_parameter_set -lm -lpthread -o /tmp/gen_ctx_init
/tmp/gen_ctx_init > /tmp/orc_ctx_init.h
cp /tmp/orc_ctx_init.h oracle/orc_ctx_init.h
[ -f uvg266_amd/csrc/vvc_ctx_init.h ] && sed -e "1i \#pragma once" -e "s#in the model order of oracle/orc_search.c#in the model order of uvg266_amd/csrc/ctu_core.h#" -e "s#^static const unsigned char k_ctx_init\[#__device__ static const unsigned char k_ctx_init[#" -e "s#^static const unsigned char k_ctx_init_inter#__device__ static const unsigned char k_ctx_init_inter#" -e "s#^static const unsigned char k_ctx_init_alf#__device__ static const unsigned char k_ctx_init_alf#" /tmp/orc_ctx_init.h > uvg266_amd/csrc/vvc_ctx_init.h
echo "wrote oracle/orc_ctx_init.h"
