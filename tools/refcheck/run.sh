#!/bin/bash
# usage: run.sh [dump]
set -e
cd "$(dirname "$0")/../.."
REF=${UVG_REF_SRC:-/root/reference}
oracle/build_ref.sh "$REF"      # gcc-only build of the reference into oracle/_ref/
make -s -C oracle
for D in 8 10; do
  if [ $D = 8 ]; then DEF=""; else DEF="-DUVG_BIT_DEPTH=10"; fi; LIB=oracle/_ref/libuvg266_$D.a
  gcc -O1 -g -std=gnu11 -w $DEF -DORC_BIT_DEPTH=$D -Ioracle/_ref/gen -I$REF/src -I$REF/src/extras -I$REF/src/strategies -Ioracle -Iinclude \
      -DHAVE_DCT -DHAVE_QUANT -DHAVE_INTRA -DHAVE_IPOL -DHAVE_SAO -DHAVE_DEBLOCK -DHAVE_ALF -DHAVE_LFNST tools/refcheck/refcheck.c tools/refcheck/rc_alfstatic.c oracle/_build/*.$D.o $LIB -lm -lpthread -fopenmp -o /tmp/refcheck$D
  /tmp/refcheck$D "$@"
done
