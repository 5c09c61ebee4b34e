#!/bin/bash
# Round-6 profile of the frame-level hand-over: rocprofv3 --kernel-trace --stats around the REFERENCE ENCODER'S OWN PROCESS
# (oracle/_ref/uvg266_8_hip, UVG266_HIP_FRAME=1, 64 pictures 1080p -p 1 --preset medium, --owf 31: groups of 16 pictures) -- the kernels
# that run inside it are the judged line's three (ctu_search_kernel, ctu_filter_kernel, slice_rows_kernel), launched by uvghip_frame_pool_*.
# Output: gpurun_out/prof6_dropin_kernel_stats.csv (+ .log); copied to profiles/r06_frame_dropin_kernel_stats.csv.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import helpers as H
base = [H.varied_picture(1920, 1080, t, 8) for t in range(4)]
with open("/tmp/dropin_in.yuv", "wb") as f:
    for t in range(64):
        for plane in base[t % 4]:
            f.write(np.ascontiguousarray(plane).tobytes())
PY
rm -rf gpurun_out/prof6_dropin
UVG266_HIP_FRAME=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof6_dropin -- oracle/_ref/uvg266_8_hip -i /tmp/dropin_in.yuv -o /tmp/dropin_out.266 \
  --input-res 1920x1080 -n 64 -p 1 --preset medium -q 22 --threads 8 --owf 31 > gpurun_out/prof6_dropin.log 2>&1
cp $(ls -t gpurun_out/prof6_dropin/*/*kernel_stats.csv | head -1) gpurun_out/prof6_dropin_kernel_stats.csv
rm -rf gpurun_out/prof6_dropin
tail -4 gpurun_out/prof6_dropin.log
cat gpurun_out/prof6_dropin_kernel_stats.csv | cut -c1-200 | head -12
