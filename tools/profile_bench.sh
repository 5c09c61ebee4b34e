#!/bin/bash
# Round profile of the bench command on the GPU box: kernel-trace statistics, then two PMC passes
# (FETCH_SIZE and WRITE_SIZE separately, counters only).  Outputs under gpurun_out/prof_*; copy the
# summaries to profiles/ afterwards (tools/summarize_profiles.py).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
CMD="python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- $CMD > gpurun_out/prof_stats.log 2>&1
tail -1 gpurun_out/prof_stats.log > gpurun_out/prof_bench_line.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- $CMD > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- $CMD > gpurun_out/prof_write.log 2>&1
ls gpurun_out/prof_stats/*/ gpurun_out/prof_fetch/*/ 2>/dev/null | head -30
