#!/bin/bash
# Developer probe: LDS counters of the search kernel with the full 67-candidate list, per block size.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for N in "$@"; do
  rm -rf gpurun_out/pmcm
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d gpurun_out/pmcm -- python tools/dev/search_mode_cost.py $N all67 64 2 > /dev/null 2>&1
  python - "$N" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmcm/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'search' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('n', sys.argv[1], {k: round(sum(v) / len(v) / 1e6, 2) for k, v in sorted(acc.items())})
PY
done
rm -rf gpurun_out/pmcm
for N in "$@"; do python tools/dev/search_mode_cost.py $N all67 64 20 | grep "n="; done
