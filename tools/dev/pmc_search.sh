cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
CMD="python bench.py --steps 1 --warmup 1 --groups 1 --no-extra --no-open-loop --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/pmc_a -- $CMD > gpurun_out/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_b -- $CMD > gpurun_out/pmc_b.log 2>&1
tail -2 gpurun_out/pmc_a.log | cut -c1-200; tail -2 gpurun_out/pmc_b.log | cut -c1-200
