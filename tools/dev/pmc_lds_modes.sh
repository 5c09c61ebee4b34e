#!/bin/bash
# Developer probe: LDS conflict counters of the search kernel per single candidate mode (64 copies of it).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
N=${1:-16}
for M in ${MODES:-0 1 18 22 30 34 40 50 54 60 66}; do
  rm -rf gpurun_out/pmcm
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d gpurun_out/pmcm -- python tools/dev/search_mode_cost.py $N $M 64 2 > /dev/null 2>&1
  python - "$M" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmcm/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'search' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('mode', sys.argv[1], {k: round(sum(v) / len(v) / 1e6, 2) for k, v in sorted(acc.items())})
PY
done
rm -rf gpurun_out/pmcm
