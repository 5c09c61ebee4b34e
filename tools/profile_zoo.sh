#!/bin/bash
# Kernel-trace statistics of tools/kernel_zoo.py (every entry point at a 1080p-sized workload).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_zoo
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_zoo -- python tools/kernel_zoo.py > gpurun_out/prof_zoo.log 2>&1
tail -3 gpurun_out/prof_zoo.log
