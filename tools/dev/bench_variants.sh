#!/bin/bash
# developer probe: bench.py under different concurrency knobs; prints value / ms_per_step per variant
run() {
  (timeout 300 env $1 python bench.py --steps 80 --no-cpu-baseline --no-extra $2 2>gpurun_out/bv.err) > gpurun_out/bv.json
  python - "$1 $2" <<'PY'
import json, sys
try:
    d = json.load(open("gpurun_out/bv.json"))
    print(sys.argv[1], d["value"], d["ms_per_step"])
except Exception as e:
    print(sys.argv[1], "FAILED", open("gpurun_out/bv.err").read()[-400:])
PY
}
run "A=1" ""
run "A=1" "--no-fork"
run "GPU_MAX_HW_QUEUES=8" ""
run "GPU_MAX_HW_QUEUES=8" "--resident 8 --streams 8"
run "A=1" "--no-graphs"
run "GPU_MAX_HW_QUEUES=8" "--no-graphs --resident 8 --streams 8"
