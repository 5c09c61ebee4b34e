#!/bin/bash
# Round-6 profile: the passes of tools/profile_bench_r05b.sh for the judged line (the all-intra closed loop; its kernels did not change
# this round) + a kernel trace of the two one-clip workloads whose kernel is this round's work (ctu_search_pb_kernel<PX, 256>: four
# waves per CTU, the in-loop filters inside, pictures in flight).  Outputs under gpurun_out/prof5_* / prof6_*;
# tools/summarize_profiles_r05.py r06 condenses the former into profiles/r06_*, the latter are copied as they are.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof5_stats gpurun_out/prof5_stats_1 gpurun_out/prof5_fetch gpurun_out/prof5_write gpurun_out/prof5_sq gpurun_out/prof6_c3 gpurun_out/prof6_ra
CMD="python bench.py --steps 4 --warmup 2 --no-extra --no-cpu-baseline"
echo "$CMD" > gpurun_out/prof5_command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof5_stats -- $CMD > gpurun_out/prof5_stats.log 2>&1
grep '^{' gpurun_out/prof5_stats.log | tail -1 > gpurun_out/prof5_bench_line.json
CMD1="python bench.py --steps 2 --warmup 1 --groups 1 --no-extra --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof5_stats_1 -- $CMD1 > gpurun_out/prof5_stats_1.log 2>&1
grep '^{' gpurun_out/prof5_stats_1.log | tail -1 > gpurun_out/prof5_bench_line_1.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof5_fetch -- $CMD1 > gpurun_out/prof5_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof5_write -- $CMD1 > gpurun_out/prof5_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/prof5_sq -- $CMD1 > gpurun_out/prof5_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof6_c3 -- python bench.py --only-c3-clip --no-cpu-baseline > gpurun_out/prof6_c3.log 2>&1
grep '^{' gpurun_out/prof6_c3.log | tail -1 > gpurun_out/prof6_c3_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof6_ra -- python bench.py --only-ra-clip --no-cpu-baseline > gpurun_out/prof6_ra.log 2>&1
grep '^{' gpurun_out/prof6_ra.log | tail -1 > gpurun_out/prof6_ra_line.json
cp $(ls -t gpurun_out/prof6_c3/*/*kernel_stats.csv | head -1) gpurun_out/prof6_c3_kernel_stats.csv
cp $(ls -t gpurun_out/prof6_ra/*/*kernel_stats.csv | head -1) gpurun_out/prof6_ra_kernel_stats.csv
rm -rf gpurun_out/prof6_c3 gpurun_out/prof6_ra gpurun_out/prof5_stats/*/*kernel_trace.csv gpurun_out/prof5_stats_1/*/*kernel_trace.csv
ls gpurun_out/prof5_stats/*/ gpurun_out/prof5_fetch/*/ 2>/dev/null | head
