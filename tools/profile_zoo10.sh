#!/bin/bash
# tools/kernel_zoo.py on 10-bit planes under the kernel trace (finds entry points whose 10-bit path lags the 8-bit one).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_zoo10
ZOO_DEPTH=10 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_zoo10 -- python tools/kernel_zoo.py > gpurun_out/prof_zoo10.log 2>&1
tail -3 gpurun_out/prof_zoo10.log
cp gpurun_out/zoo_plan.json gpurun_out/zoo_plan10.json
