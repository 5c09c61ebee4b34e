#!/bin/bash
# Developer probe: builds libuvg266hip variants with different LDS strides of the search kernel into gpurun_variants/.
# usage: build_stride_variants.sh "b16 p16 b32 p32" ...
cd "$(dirname "$0")/../.."
make -s -j8 -C uvg266_amd/csrc
mkdir -p gpurun_variants
for V in "$@"; do
  set -- $V
  TAG="${1}_${2}_${3}_${4}"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wno-unused-function \
     -DUVGHIP_BRS16_PAD=$1 -DUVGHIP_PS16=$2 -DUVGHIP_BRS32_PAD=$3 -DUVGHIP_PS32=$4 -c uvg266_amd/csrc/intra.hip -o /tmp/intra_$TAG.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_variants/lib_$TAG.so /tmp/intra_$TAG.o $(ls uvg266_amd/csrc/_build/*.o | grep -v intra.o) && echo built $TAG &
done
wait
