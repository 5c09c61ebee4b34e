#!/bin/bash
# usage: ctu_dump.sh <depth 8|10> <in.yuv> <W> <H> <frames> <out prefix> [option value]...
# Builds tools/refcheck/ctu_dump.c against oracle/_ref (the gcc-only build of /root/reference, oracle/build_ref.sh) and runs it.
set -e
cd "$(dirname "$0")/../.."
REF=${UVG_REF_SRC:-/root/reference}
oracle/build_ref.sh "$REF" >/dev/null
D=$1; shift
if [ $D = 8 ]; then DEF=""; else DEF="-DUVG_BIT_DEPTH=10"; fi; LIB=oracle/_ref/libuvg266_$D.a
BIN=/tmp/ctu_dump$D
# (re)build only when the dumper's source is newer; build to a private name and rename, so that concurrent runs never see a half-written binary
if [ ! -x $BIN ] || [ tools/refcheck/ctu_dump.c -nt $BIN ] || [ tools/refcheck/ctu_dump.sh -nt $BIN ] || [ $LIB -nt $BIN ]; then
  gcc -O1 -g -std=gnu11 -w $DEF -Ioracle/_ref/gen -I$REF/src -I$REF/src/extras -I$REF/src/strategies tools/refcheck/ctu_dump.c $LIB \
      -Wl,--wrap=uvg_search_lcu -Wl,--wrap=uvg_encode_coding_tree -Wl,--wrap=uvg_sao_search_lcu -Wl,--wrap=uvg_bitstream_put_byte -Wl,--wrap=uvg_cabac_finish \
      -Wl,--wrap=uvg_bitstream_align_zero -Wl,--wrap=uvg_inter_get_merge_cand -Wl,--wrap=uvg_inter_get_mv_cand -Wl,--wrap=uvg_search_cu_inter -Wl,--wrap=uvg_alf_enc_process -Wl,--wrap=uvg_encode_alf_adaptive_parameter_set -lm -lpthread -o $BIN.$$ && mv -f $BIN.$$ $BIN
fi
IN=$1; W=$2; H=$3; N=$4; OUT=$5; shift 5
$BIN $IN $W $H $N $OUT.bin $OUT.266 "$@"
