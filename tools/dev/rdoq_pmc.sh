#!/bin/bash
# SQ counters of the RDOQ kernel (developer probe): instruction mix and stall split per launch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_out -o p --output-format csv -- python $R/tools/dev/rdoq_real.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob('/tmp/pmc_out/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if 'rdoq' not in k: continue
        key = (k[-40:], r['Grid_Size'])
        acc[key][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(key, r['Counter_Name'])] += 1
for key, d in acc.items():
    print(key, {c: round(v / cnt[(key, c)]) for c, v in d.items()})
PY
done
