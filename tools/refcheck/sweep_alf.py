#!/usr/bin/env python3
"""Dev-time: N all-intra `--alf full` encodes of the real reference (sizes incl. partial CTUs, depths, QPs, kinds of content drawn from
`seed`); for every picture the oracle's ALF reconstruction (oracle/orc_alf_picture.c) from the recorded decisions against the picture
uvg_alf_enc_process left, and the oracle's frame statistics against the covariances the encoder gathered.  Nothing is written.   python tools/refcheck/sweep_alf.py N seed"""
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
            # the frame statistics alf_derive_stats_for_filtering gathered (record arrays 14, 15: per luma class / chroma plane, summed over the
            # CTUs) against the oracle's get_blk_stats on the picture ALF got and the source, summed the same way
            if int(r[0][29]):
                t = int(r[0][0])
                src = H.varied_picture(W, Hh, kind * 1000 + t0 + t, depth)
                pre2 = [p.reshape(Hh >> (c > 0), W >> (c > 0)) for c, p in enumerate(pre)]
                cls = orc.alf_classify_frame(depth, pre2[0], W, Hh, int(r[0][28]) + 4)
                luma, chroma = np.zeros((25, 1509), np.int64), np.zeros((2, 1509), np.int64)
                for y in range(0, Hh, 64):
                    for x in range(0, W, 64):
                        w, h = min(64, W - x), min(64, Hh - y)
                        e, yv, pa = orc.alf_stats_rect(depth, np.ascontiguousarray(src[0]), pre2[0], W, Hh, x, y, w, h, False, cls)
                        luma += H.alf_sum_layout(e, yv, pa, 13)
                        for c in (1, 2):
                            e, yv, pa = orc.alf_stats_rect(depth, np.ascontiguousarray(src[c]), pre2[c], W // 2, Hh // 2, x // 2, y // 2, w // 2, h // 2, True, None)
                            chroma[c - 1] += H.alf_sum_layout(e, yv, pa, 7)[0]
                tot["stats"] = tot.get("stats", 0) + 1
                if not np.array_equal(luma, r[14].reshape(25, 1509)) or not np.array_equal(chroma, r[15].reshape(2, 1509)):
                    ok = False
                    print("  frame statistics differ, picture", t)
        print((W, Hh, depth, qp, frames, t0, kind), "OK" if ok else "MISMATCH", flush=True)
        bad += not ok
    print(f"{n} encodes, {bad} with a mismatch;", tot)
