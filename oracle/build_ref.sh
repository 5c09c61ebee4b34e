#!/bin/bash
# oracle/build_ref.sh -- builds the REAL reference (ultravideo/uvg266 at /root/reference) with plain gcc, from the sources
# where they lie, into oracle/_ref/ (git-ignored; travels to the GPU box with the snapshot).  TEST INFRASTRUCTURE ONLY:
# the product never links or loads anything produced here; tools/refcheck/ (golden generators), bench.py's
# `cpu_baseline` leg and __graft_entry__.build() are the only users.
#
# The reference's own build system is NOT run.  What its CMakeLists.txt does is restated here in shell:
#   * src/version.h is instantiated from src/version.h.in (CMakeLists.txt:125 `configure_file(... @ONLY)`): the three
#     @VARS@ are substituted with sed into oracle/_ref/gen/version.h; nothing is written under /root/reference;
#   * library sources = src/*.c minus encmain.c cli.c yuv_io.c, + src/strategies/**/*.c minus
#     avx2/encode_coding_tree-avx2.c, + src/extras/libmd5.c   (CMakeLists.txt:127-141);
#   * per-directory ISA flags: avx2/ -> -mavx2 -mbmi -mpopcnt -mlzcnt -mbmi2, sse41/ -> -msse4.1, sse42/ -> -msse4.2
#     (CMakeLists.txt:210-215); -DUVG_DLL_EXPORTS (:144); 10-bit build = -DUVG_BIT_DEPTH=10;
#   * CLI = encmain.c cli.c yuv_io.c linked against the library, -lm -lpthread.
# Outputs: oracle/_ref/libuvg266_{8,10}.a, oracle/_ref/uvg266_{8,10} (the encoder CLI), oracle/_ref/gen/version.h,
#          oracle/_ref/STAMP (sha1 of the reference's src tree the build was made from).
# usage: oracle/build_ref.sh [reference root, default /root/reference]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${1:-${UVG_REF_SRC:-/root/reference}}"
OUT="$HERE/_ref"
[ -d "$REF/src" ] || { echo "build_ref.sh: no reference at $REF" >&2; exit 3; }
SRCSUM=$(cd "$REF" && find src -type f \( -name '*.c' -o -name '*.h' -o -name '*.in' \) | LC_ALL=C sort | xargs sha1sum | sha1sum | cut -d' ' -f1)
if [ -f "$OUT/STAMP" ] && [ "$(cat "$OUT/STAMP")" = "$SRCSUM" ] && [ -x "$OUT/uvg266_8" ] && [ -x "$OUT/uvg266_10" ] \
   && [ -f "$OUT/libuvg266_8.a" ] && [ -f "$OUT/libuvg266_10.a" ]; then
  exit 0
fi
mkdir -p "$OUT/gen"
VERSION=$(sed -n 's/^VERSION \([0-9.]*\).*/\1/p' "$REF/CMakeLists.txt" | head -1)
sed -e "s/@PROJECT_VERSION@/${VERSION:-0.0.0}/" -e "s/@UVG_COMPILER_STRING@/GNU $(gcc -dumpfullversion)/" \
    -e "s/@CMAKE_BUILD_DATE@/reproducible/" "$REF/src/version.h.in" > "$OUT/gen/version.h"
CC=${CC:-gcc}
BASE="-O3 -g0 -w -DNDEBUG -DUVG_DLL_EXPORTS -I$OUT/gen -I$REF/src -I$REF/src/extras -I$REF/src/strategies"
LIBSRC=$(cd "$REF" && ls src/*.c | grep -v -e 'src/encmain.c' -e 'src/cli.c' -e 'src/yuv_io.c'; \
         cd "$REF" && find src/strategies -name '*.c' | grep -v 'avx2/encode_coding_tree-avx2.c' | LC_ALL=C sort; echo src/extras/libmd5.c)
JOBS=${JOBS:-$(nproc)}
for D in 8 10; do
  OBJ="$OUT/obj$D"; rm -rf "$OBJ"; mkdir -p "$OBJ"
  DEF=""; [ $D = 10 ] && DEF="-DUVG_BIT_DEPTH=10"
  for f in $LIBSRC encmain cli yuv_io; do
    case "$f" in encmain|cli|yuv_io) f=src/$f.c;; esac
    isa=""
    case "$f" in
      src/strategies/avx2/*)  isa="-mavx2 -mbmi -mpopcnt -mlzcnt -mbmi2";;
      src/strategies/sse41/*) isa="-msse4.1";;
      src/strategies/sse42/*) isa="-msse4.2";;
    esac
    o="$OBJ/$(echo "$f" | tr '/' '_' | sed 's/\.c$/.o/')"
    echo "$CC $BASE $DEF $isa -c $REF/$f -o $o"
  done | xargs -P "$JOBS" -I{} sh -c '{}'
  CLI="$OBJ/src_encmain.o $OBJ/src_cli.o $OBJ/src_yuv_io.o"
  rm -f "$OUT/libuvg266_$D.a"
  ar rcs "$OUT/libuvg266_$D.a" $(ls "$OBJ"/*.o | grep -v -e src_encmain.o -e src_cli.o -e src_yuv_io.o)
  $CC -o "$OUT/uvg266_$D" $CLI "$OUT/libuvg266_$D.a" -lm -lpthread
  rm -rf "$OBJ"
done
echo "$SRCSUM" > "$OUT/STAMP"
echo "build_ref.sh: built oracle/_ref/{libuvg266_8.a,libuvg266_10.a,uvg266_8,uvg266_10} from $REF ($SRCSUM)"
