#!/bin/bash
# usage: pmc.sh <kernel-name-substring> <python script> [args]   -- SQ counters per launch of the matching kernels (developer probe)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PAT=$1; shift
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_out -o p --output-format csv -- python $R/"$@" > /dev/null 2>&1
  PAT="$PAT" python3 - <<PY
import csv, glob, collections, os
pat = os.environ["PAT"]
f = glob.glob('/tmp/pmc_out/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if pat not in k: continue
        key = (k[:48], r['Grid_Size'])
        acc[key][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(key, r['Counter_Name'])] += 1
for key, d in acc.items():
    print(key, {c: round(v / cnt[(key, c)]) for c, v in d.items()})
PY
done
