#!/usr/bin/env python3
"""Dev-time: N all-intra `--alf full` encodes of the real reference (sizes incl. partial CTUs, depths, QPs, kinds of content drawn from
`seed`); for every picture the oracle's ALF reconstruction (oracle/orc_alf_picture.c) from the recorded decisions against the picture
uvg_alf_enc_process left.  Nothing is written.   python tools/refcheck/sweep_alf.py N seed"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers as H
from make_ctu_goldens import read_records

if __name__ == "__main__":
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.default_rng(seed)
    orc = H.load_oracle()
    fixed = np.ascontiguousarray(np.load(os.path.join(H.GOLDEN, "ref_alf_fixed.npy")), np.int16)
    bad = 0
    tot = dict(pictures=0, on=0, fixed=0, cc=0, undefined=0)
    for case in range(n):
        W = int(rng.choice([64, 72, 128, 136, 192, 200, 256, 264, 320])); Hh = int(rng.choice([64, 72, 128, 136, 192]))
        depth, qp, frames, t0, kind = int(rng.choice([8, 10])), int(rng.integers(12, 42)), int(rng.integers(2, 5)), int(rng.integers(0, 60)), int(rng.choice([0, 1, 2, 3]))
        px = H.px_dtype(depth)
        yuv, out = f"/tmp/sweep_alf_{os.getpid()}.yuv", f"/tmp/sweep_alf_{os.getpid()}"
        with open(yuv, "wb") as f:
            for t in range(frames):
                for p in H.varied_picture(W, Hh, kind * 1000 + t0 + t, depth):
                    f.write(p.astype(px).tobytes())
        subprocess.check_call([os.path.join(ROOT, "tools/refcheck/ctu_dump.sh"), str(depth), yuv, str(W), str(Hh), str(frames), out,
                               "preset", "medium", "period", "1", "qp", str(qp), "alf", "full"], stderr=subprocess.DEVNULL)
        ok = True
        for name, r in read_records(out + ".bin"):
            if name != "alf":
                continue
            pre = [np.ascontiguousarray(r[1 + c]) for c in range(3)]
            res = [np.zeros_like(p) for p in pre]
            rc = orc.fn(depth, "alf_reconstruct_picture", ctypes.c_int)(*(H.ptr(p) for p in pre), W, Hh, *(H.ptr(o) for o in res), H.ptr(np.ascontiguousarray(r[0])), H.ptr(np.ascontiguousarray(r[7])),
                                                                        H.ptr(np.ascontiguousarray(r[8])), H.ptr(np.ascontiguousarray(r[9])), H.ptr(np.ascontiguousarray(r[10])), H.ptr(np.ascontiguousarray(r[11])), H.ptr(fixed))
            tot["pictures"] += 1; tot["on"] += int(r[0][4]); tot["cc"] += int(r[0][17] or r[0][18])
            nn = len(r[8]); tot["fixed"] += int(r[0][4] and (r[8][r[7][:nn] > 0] < 16).any())
            if rc == -1:
                tot["undefined"] += 1
                continue
            if rc != 0 or any(not np.array_equal(o, r[4 + c]) for c, o in enumerate(res)):
                ok = False
        print((W, Hh, depth, qp, frames, t0, kind), "OK" if ok else "MISMATCH", flush=True)
        bad += not ok
    print(f"{n} encodes, {bad} with a mismatch;", tot)
