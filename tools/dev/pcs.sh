#!/bin/bash
# dev: PC sampling of the CTU search kernel (rocprofv3 --pc-sampling-beta-enabled) -> gpurun_out/pcs/<tag>_{stochastic,host_trap}.txt
# usage (on the GPU box, from the repo root): tools/dev/pcs.sh <tag> W H depth n_pictures
set -u
TAG=${1:-r05}; W=${2:-1920}; H=${3:-1080}; D=${4:-8}; N=${5:-64}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pcs; mkdir -p $OUT
export UVGHIP_LIB=$ROOT/uvg266_amd/libuvg266hip_g.so
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
cd /tmp; export TMPDIR=/tmp
for method in stochastic host_trap; do
  if [ $method = stochastic ]; then unit=cycles; interval=1048576; else unit=time; interval=2000; fi
  rm -rf /tmp/pcs_$method
  timeout 900 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $interval \
      --output-format csv -d /tmp/pcs_$method -- python $ROOT/tools/dev/ctu_run.py $W $H $D $N 2 > $OUT/${TAG}_$method.log 2>&1
  echo "$method rc=$?" >> $OUT/${TAG}_$method.log
  find /tmp/pcs_$method -name "*.csv" -size +0 | head >> $OUT/${TAG}_$method.log
  python $ROOT/tools/dev/pcs_aggregate.py /tmp/pcs_$method $OUT/${TAG}_$method.txt > /dev/null 2>> $OUT/${TAG}_$method.log
  if [ -s $OUT/${TAG}_$method.txt ] && grep -q "^samples: [1-9]" $OUT/${TAG}_$method.txt; then break; fi
done
ls -la $OUT
