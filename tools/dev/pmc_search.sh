#!/bin/bash
# PMC pass over the search probe (counters only, kernel-trace; no sys/hip traces).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-8}; WHAT=${2:-all67}
cd $R
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc1 -- python tools/dev/search_mode_cost.py $N $WHAT 64 2 > gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc2 -- python tools/dev/search_mode_cost.py $N $WHAT 64 2 > gpurun_out/pmc2.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ('gpurun_out/pmc1', 'gpurun_out/pmc2'):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'search' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(d, k, sum(v) / len(v), len(v))
PY
