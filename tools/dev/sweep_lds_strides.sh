#!/bin/bash
# Developer probe (runs on the GPU box): LDS conflict counters of the search kernel for every library variant in
# gpurun_variants/, separately for the candidate lists that read block pair rows (pos) and per-wave strips (neg).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp uvg266_amd/libuvg266hip.so /tmp/lib_keep.so
for LIB in gpurun_variants/lib_*.so; do
  cp $LIB uvg266_amd/libuvg266hip.so
  for N in 16 32; do for W in pos neg; do
    rm -rf gpurun_out/pmcm
    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmcm -- python tools/dev/search_mode_cost.py $N $W 64 2 > /dev/null 2>&1
    python - "$LIB" "$N" "$W" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmcm/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'search' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[1], 'n', sys.argv[2], sys.argv[3], {k: round(sum(v) / len(v) / 1e6, 2) for k, v in sorted(acc.items())})
PY
  done; done
done
cp /tmp/lib_keep.so uvg266_amd/libuvg266hip.so
rm -rf gpurun_out/pmcm
