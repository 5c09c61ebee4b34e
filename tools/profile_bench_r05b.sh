#!/bin/bash
# Round-5 profile, final kernels (the 10-bit LDS diet, the coder's staged context words; no occupancy pass) -- otherwise tools/profile_bench_r05.sh:
# Round-5 profile (same passes as round 3, plus an SQ occupancy pass) of the bench command (the closed loop: CTU search -> deblock -> SAO) on the GPU box.  Outputs under
# gpurun_out/prof5_*; tools/summarize_profiles_r05.py condenses them into profiles/r05_*.
#   prof5_stats        rocprofv3 --kernel-trace --stats of the bench as it runs by default (two launches in flight)
#   prof5_stats_1      the same with one launch in flight (--groups 1): the search kernel alone on the device
#   prof5_fetch/_write PMC passes (counters only), --groups 1 so that the device-wide TCC counters belong to one kernel at a time
#   prof5_sq           SQ instruction / wave counters per launch (--groups 1)
# The side measurements (2160p extra workload, open-loop line, CPU baseline) are off: they launch other kernels.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof5_stats gpurun_out/prof5_stats_1 gpurun_out/prof5_fetch gpurun_out/prof5_write gpurun_out/prof5_sq
CMD="python bench.py --steps 4 --warmup 2 --no-extra --no-open-loop --no-cpu-baseline"
echo "$CMD" > gpurun_out/prof5_command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof5_stats -- $CMD > gpurun_out/prof5_stats.log 2>&1
grep '^{' gpurun_out/prof5_stats.log | tail -1 > gpurun_out/prof5_bench_line.json
CMD1="python bench.py --steps 2 --warmup 1 --groups 1 --no-extra --no-open-loop --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof5_stats_1 -- $CMD1 > gpurun_out/prof5_stats_1.log 2>&1
grep '^{' gpurun_out/prof5_stats_1.log | tail -1 > gpurun_out/prof5_bench_line_1.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof5_fetch -- $CMD1 > gpurun_out/prof5_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof5_write -- $CMD1 > gpurun_out/prof5_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/prof5_sq -- $CMD1 > gpurun_out/prof5_sq.log 2>&1

ls gpurun_out/prof5_stats/*/ gpurun_out/prof5_fetch/*/ 2>/dev/null | head
