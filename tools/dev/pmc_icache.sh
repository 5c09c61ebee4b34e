#!/bin/bash
# Instruction-cache and issue counters of the CTU search kernel at 4 / 2 / 1 workgroups per CU (UVGHIP_CTU_LDS_PAD pads the
# dynamic LDS so that fewer workgroups fit).  Run on the GPU box:  tools/dev/pmc_icache.sh [tag]
# Writes gpurun_out/pmc_icache_<tag>/{pad*_a,pad*_b}/ (rocprofv3 csv) and gpurun_out/pmc_icache_<tag>.txt (one summary line per pass).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r04}
OUT=gpurun_out/pmc_icache_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o -i -E '\b(SQC_[A-Z_0-9]*ICACHE[A-Z_0-9]*|SQ_IFETCH[A-Z_0-9]*|SQC_[A-Z_0-9]*INST[A-Z_0-9]*)\b' | sort -u > $OUT/available.txt
cat $OUT/available.txt | tr '\n' ' '; echo
have() { grep -q -x "$1" $OUT/available.txt; }
A=""; for c in SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH; do have $c && A="$A $c"; done
B="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"
CMD="python bench.py --steps 1 --warmup 1 --groups 1 --no-extra --no-open-loop --no-cpu-baseline --no-parity"
for PAD in 0 50000 100000; do
  export UVGHIP_CTU_LDS_PAD=$PAD
  [ -n "$A" ] && rocprofv3 --kernel-trace --pmc $A --output-format csv -d $OUT/pad${PAD}_a -- $CMD > $OUT/pad${PAD}_a.log 2>&1
  rocprofv3 --kernel-trace --pmc $B --output-format csv -d $OUT/pad${PAD}_b -- $CMD > $OUT/pad${PAD}_b.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pad${PAD}_t -- $CMD > $OUT/pad${PAD}_t.log 2>&1
done
unset UVGHIP_CTU_LDS_PAD
python - "$OUT" <<'PY' | tee gpurun_out/pmc_icache_$TAG.txt
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/pad*_[abt]")):
    tot = collections.defaultdict(float); n = 0
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ctu_search_kernel" in r.get("Kernel_Name", ""):
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "ctu_search_kernel" in r.get("Name", ""):
                tot["calls"] = float(r["Calls"]); tot["avg_ns"] = float(r["AverageNs"])
    print(d.split("/")[-1], dict(tot))
PY
