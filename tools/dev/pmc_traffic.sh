cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof3_fetch gpurun_out/prof3_write
CMD1="python bench.py --steps 1 --warmup 1 --groups 1 --no-extra --no-open-loop --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof3_fetch -- $CMD1 > gpurun_out/prof3_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof3_write -- $CMD1 > gpurun_out/prof3_write.log 2>&1
python - <<'PY'
import csv, glob
for what in ("fetch", "write"):
    for f in glob.glob(f"gpurun_out/prof3_{what}/*/*counter_collection.csv"):
        tot = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            tot.setdefault(k, [0, 0.0]); tot[k][0] += 1; tot[k][1] += float(r["Counter_Value"])
        for k, (n, v) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:3]:
            print(what, k, n, "launch rows", v, "sum")
PY
