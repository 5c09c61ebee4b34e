#!/usr/bin/env python3
"""The frame-level hand-over timed inside the reference encoder's own CLI (needs a GPU and oracle/_ref/uvg266_8{,_hip}):
a 1920x1080 8-bit clip, -p 1 --preset medium (BASELINE configs[1]'s settings), once with the encoder's CPU search on all host threads
(AVX2 strategies) and once with UVG266_HIP_FRAME=1 at several --owf (frames in flight = slots of the frame pool; groups of half of them per launch).
Prints the CLI's wall time per run and whether the two files are the same file.  DEVELOPMENT TOOL (test infrastructure binaries).
usage: frame_dropin_time.py [frames=64] [qp=22] [width=1920 height=1080 depth=8] [owf,owf,...] [extra encoder options, e.g. --tiles 6x4 --wpp]"""
import hashlib
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402


def run(binary, yuv, out, args, env_extra):
    env = dict(os.environ)
    for k in [k for k in env if k.startswith("UVG266_")]:
        del env[k]
    env.update(env_extra)
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", binary), "-i", yuv, "-o", out] + args, env=env, capture_output=True, text=True)
    dt = time.time() - t0
    if r.returncode:
        sys.exit(r.stderr[-2000:])
    fps = [l for l in r.stderr.splitlines() if l.strip().startswith("FPS:")]
    return hashlib.md5(open(out, "rb").read()).hexdigest(), dt, (fps[0].split()[1] if fps else "?")


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    qp = int(sys.argv[2]) if len(sys.argv) > 2 else 22
    w, h, depth = (int(a) for a in sys.argv[3:6]) if len(sys.argv) > 5 else (1920, 1080, 8)
    owfs = [int(a) for a in sys.argv[6].split(",")] if len(sys.argv) > 6 else [0, 3, 7, 15, 31, 63]
    with tempfile.TemporaryDirectory() as d:
        yuv = os.path.join(d, "in.yuv")
        with open(yuv, "wb") as f:
            base = [H.varied_picture(w, h, t, depth) for t in range(4)]
            for t in range(frames):
                for plane in base[t % 4]:
                    f.write(np.ascontiguousarray(plane).tobytes())
        args = ["--input-res", f"{w}x{h}", "-n", str(frames), "-p", "1", "--preset", "medium", "-q", str(qp)] + (["--input-bitdepth", "10"] if depth == 10 else []) + sys.argv[7:]
        threads = os.cpu_count()
        want, dt, fps = run(f"uvg266_{depth}", yuv, os.path.join(d, "cpu.266"), args + ["--threads", str(threads)], {})
        print(f"{frames} pictures {w}x{h} {depth}-bit qp {qp} {' '.join(sys.argv[7:])}: the encoder's CPU search, {threads} threads, --owf auto: {dt:.2f} s wall, the CLI's own FPS {fps}")
        for owf in owfs:
            got, dt, fps = run(f"uvg266_{depth}_hip", yuv, os.path.join(d, f"hip{owf}.266"), args + ["--threads", "8", "--owf", str(owf)], {"UVG266_HIP_FRAME": "1"})
            print(f"  UVG266_HIP_FRAME=1 --owf {owf:2d}: {dt:.2f} s wall, the CLI's own FPS {fps}; the same file: {got == want}")


if __name__ == "__main__":
    main()
