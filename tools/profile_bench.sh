#!/bin/bash
# Round profile of the bench command on the GPU box.  Outputs under gpurun_out/prof_*; condense them into
# profiles/ afterwards with tools/summarize_profiles.py <tag>.
#   prof_stats         kernel-trace statistics of the bench (pictures in flight on several streams / hipGraphs: a launch's
#                      duration includes the time it shares the GPU with other pictures' kernels)
#   prof_stats_serial  the same with --serial (one stream, no graphs): each launch alone on the GPU
#   prof_fetch/_write  PMC passes (counters only), --serial so that the device-wide TCC counters belong to one kernel
#   prof_valu          SQ instruction counts per launch (--serial)
#   prof_stats_4k      kernel-trace statistics of the extra workload (3840x2160 10-bit + ALF), serial
#   prof_mfma_4k       matrix-core counters of the extra workload (ALF covariance statistics), serial
# The profiled command is the default bench without its side measurements (the 2160p extra workload and the closed-loop
# probe launch the same kernel symbols at other sizes and would blur the per-kernel averages).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_stats gpurun_out/prof_stats_serial gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_valu gpurun_out/prof_stats_4k gpurun_out/prof_mfma_4k
CMD="python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-extra --no-closed-loop"
echo "$CMD" > gpurun_out/prof_command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- $CMD > gpurun_out/prof_stats.log 2>&1
grep '^{' gpurun_out/prof_stats.log | tail -1 > gpurun_out/prof_bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats_serial -- $CMD --serial > gpurun_out/prof_stats_serial.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- $CMD --serial > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- $CMD --serial > gpurun_out/prof_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/prof_valu -- $CMD --serial > gpurun_out/prof_valu.log 2>&1
CMD4K="python bench.py --workload 2160p10alf --steps 12 --warmup 4 --no-cpu-baseline --no-closed-loop --serial"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats_4k -- $CMD4K > gpurun_out/prof_stats_4k.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES --output-format csv -d gpurun_out/prof_mfma_4k -- $CMD4K > gpurun_out/prof_mfma_4k.log 2>&1
ls gpurun_out/prof_stats/*/ gpurun_out/prof_valu/*/ gpurun_out/prof_mfma_4k/*/ 2>/dev/null | head -30
