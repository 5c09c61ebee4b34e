cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_w1 gpurun_out/pmc_w2
CMD1="python bench.py --steps 1 --warmup 1 --groups 1 --no-extra --no-open-loop --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY --output-format csv -d gpurun_out/pmc_w1 -- $CMD1 > gpurun_out/pmc_w1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_w2 -- $CMD1 > gpurun_out/pmc_w2.log 2>&1
python - <<'PY'
import csv, glob
for d in ("pmc_w1", "pmc_w2"):
    for f in glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv"):
        tot = {}
        for r in csv.DictReader(open(f)):
            if "ctu_search" not in r["Kernel_Name"]: continue
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        print(d, tot)
PY
tail -3 gpurun_out/pmc_w2.log | cut -c1-300
