#!/bin/bash
# tools/refcheck/build_ref_hip.sh -- the drop-in, dropped in: the REAL reference encoder with the "hip" strategy backend linked.
#   1. /root/reference/src is copied to a scratch directory under /tmp (never into the repository, nothing is written under
#      /root/reference);
#   2. INTEGRATION.md section 1 is applied by tools/refcheck/patch_ref_hip.py (one registration block per strategy group) and the
#      section-2 shim (uvg266_amd/csrc/shim/strategies-hip-state.c) is copied to src/strategies/hip/; so are section 10's frame-level
#      hand-over (uvg266_amd/csrc/shim/frame-hip.c) and its two statements in uvg_encode_one_frame / the bitstream job;
#   3. everything is compiled with oracle/build_ref.sh's flags + -DUVG_HAVE_HIP (+ -DDEBUG_STRATEGYSELECTOR for strategyselector.c so
#      that the selector prints which strategy it chose per type) and linked against uvg266_amd/libuvg266hip.so.
# Output: oracle/_ref/uvg266_{8,10}_hip (binaries only; git-ignored; they travel to the GPU box with the snapshot like oracle/_ref's
# other binaries).  TEST INFRASTRUCTURE: tests/test_gpu_dropin.py runs it; the product never loads it.
# usage: tools/refcheck/build_ref_hip.sh [reference root, default /root/reference]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
REF="${1:-${UVG_REF_SRC:-/root/reference}}"
OUT="$ROOT/oracle/_ref"
[ -d "$REF/src" ] || { echo "build_ref_hip.sh: no reference at $REF" >&2; exit 3; }
[ -f "$ROOT/uvg266_amd/libuvg266hip.so" ] || { echo "build_ref_hip.sh: build uvg266_amd/libuvg266hip.so first" >&2; exit 3; }
"$ROOT/oracle/build_ref.sh" "$REF"                      # oracle/_ref/gen/version.h and the plain build beside it
SUM=$( (cd "$REF" && find src -type f \( -name '*.c' -o -name '*.h' -o -name '*.in' \) | LC_ALL=C sort | xargs sha1sum; \
        sha1sum "$HERE/patch_ref_hip.py" "$HERE/build_ref_hip.sh" "$ROOT/uvg266_amd/csrc/shim/strategies-hip-state.c" "$ROOT/uvg266_amd/csrc/shim/frame-hip.c" "$ROOT/include/uvg266_hip.h") | sha1sum | cut -d' ' -f1)
if [ -f "$OUT/STAMP_HIP" ] && [ "$(cat "$OUT/STAMP_HIP")" = "$SUM" ] && [ -x "$OUT/uvg266_8_hip" ] && [ -x "$OUT/uvg266_10_hip" ]; then exit 0; fi
SCR=$(mktemp -d /tmp/uvg266_hip_src.XXXXXX)
trap 'rm -rf "$SCR"' EXIT
cp -r "$REF/src" "$SCR/src"
chmod -R u+w "$SCR/src"
mkdir -p "$SCR/src/strategies/hip"
cp "$ROOT/uvg266_amd/csrc/shim/strategies-hip-state.c" "$ROOT/uvg266_amd/csrc/shim/frame-hip.c" "$SCR/src/strategies/hip/"
python3 "$HERE/patch_ref_hip.py" "$SCR/src"
CC=${CC:-gcc}
BASE="-O3 -g0 -w -DNDEBUG -DUVG_DLL_EXPORTS -DUVG_HAVE_HIP -I$OUT/gen -I$SCR/src -I$SCR/src/extras -I$SCR/src/strategies -I$ROOT/include"
SRCS=$(cd "$SCR" && ls src/*.c; cd "$SCR" && find src/strategies -name '*.c' | grep -v 'avx2/encode_coding_tree-avx2.c' | LC_ALL=C sort; echo src/extras/libmd5.c)
JOBS=${JOBS:-$(nproc)}
for D in 8 10; do
  OBJ="$SCR/obj$D"; mkdir -p "$OBJ"
  DEF=""; [ $D = 10 ] && DEF="-DUVG_BIT_DEPTH=10"
  for f in $SRCS; do
    isa=""
    case "$f" in
      src/strategies/avx2/*)  isa="-mavx2 -mbmi -mpopcnt -mlzcnt -mbmi2";;
      src/strategies/sse41/*) isa="-msse4.1";;
      src/strategies/sse42/*) isa="-msse4.2";;
      src/strategyselector.c) isa="-DDEBUG_STRATEGYSELECTOR";;
    esac
    o="$OBJ/$(echo "$f" | tr '/' '_' | sed 's/\.c$/.o/')"
    echo "$CC $BASE $DEF $isa -c $SCR/$f -o $o"
  done | xargs -P "$JOBS" -I{} sh -c '{}'
  # -rdynamic: libuvg266hip.so resolves uvg_strategyselector_register (a weak reference there) from the executable
  $CC -rdynamic -o "$OUT/uvg266_${D}_hip" "$OBJ"/*.o -L"$ROOT/uvg266_amd" -luvg266hip -Wl,-rpath,'$ORIGIN/../../uvg266_amd' \
      -Wl,-rpath-link,/opt/rocm/lib -lm -lpthread
done
echo "$SUM" > "$OUT/STAMP_HIP"
echo "build_ref_hip.sh: built oracle/_ref/uvg266_{8,10}_hip (reference + hip strategy backend)"
