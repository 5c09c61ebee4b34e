#!/usr/bin/env python3
"""Dev-time: N encodes of the real reference over GOP structures and the round-5 options -- random access (--gop 8 / 16, with and without
a short intra period: CRA / RASL pictures), low delay, --owf 0 / 1, rd 0 / 1, sizes, depths, QPs, three kinds of content -- through
make_ctu_goldens.inter (records into a scratch directory) and the oracle's chain, as tests/test_oracle_inter_search.py does for the
committed goldens; SWEEP_EMUL=1: also the P / B kernel's host emulation (tests/emul) against the same records.  Nothing is kept.
    python tools/refcheck/sweep_gop.py N seed [procs]"""
import os, sys, tempfile
import numpy as np
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def one(case):
    import io, contextlib
    import make_ctu_goldens as M
    import helpers as H
    W, Hh, depth, qp, frames, extra, clip = case
    tmp = tempfile.mkdtemp()
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            tag = M.inter(W, Hh, depth, qp, frames, extra=extra, suffix="_s", out_dir=tmp, clip=clip)
        g = np.load(os.path.join(tmp, f"ref_inter_{tag}.npz"))
        orc = H.load_oracle()
        W2, H2, d2, pics, P = H.inter_pictures_from_golden(g)
        calls = 0
        for fr, d, r, buf, ntr in H.run_inter_oracle(orc, W2, H2, d2, pics, P):
            msgs = H.compare_inter_picture(W2, H2, d, r, buf, ntr)
            if msgs:
                return case, f"picture {fr}: {msgs[:2]}"
            calls += ntr
        # the whole .266 from the rows' bytes and the library's host NAL writers (tests/test_picture_nal.py's whole-file test)
        import ctypes
        from uvg266_amd import lib
        L = lib.load_library()
        qp0, hc = int(g["dims"][3]), (H2 + 63) // 64
        first = {}
        for k in range(len(g["meta"])):
            first.setdefault(int(g["meta"][k][0]), k)
        stream, mine, irap_poc = g["bitstream"].tobytes(), b"", 0
        for f in range(int(g["dims"][4])):
            k = first[f]
            off = g["row_off"][f * hc:f * hc + hc + 1]
            sizes = np.diff(off).astype(np.int32)
            rows = np.zeros((hc, int(sizes.max())), np.uint8)
            for r in range(hc):
                rows[r, :sizes[r]] = g["row_bytes"][off[r]:off[r + 1]]
            sums = np.ascontiguousarray([H.picture_checksum(g[nme][f], d2) for nme in ("final_y", "final_u", "final_v")], np.uint32)
            slice_type, frame_qp, poc = int(g["meta"][k][6]), int(g["meta"][k][7]), int(g["refs"][k][51])
            cap = int(sizes.sum()) + 128 + 4 * hc
            out, n = np.zeros(cap, np.uint8), ctypes.c_size_t(0)
            n_refs, cfg = int(g["refs"][k][0]), g["cfg"]
            if slice_type == 2 and poc == 0:
                rc = L.uvghip_write_idr_nals_ra(poc, H.poc_lsb_bits(g), frame_qp - qp0, 1, H.ptr(rows), rows.shape[1], H.ptr(sizes), hc, H.ptr(sums), H.ptr(out), cap, ctypes.byref(n))
            else:
                rc = H.write_inter_nals(L, g, poc, slice_type, [int(p) for p in g["refs"][k][1:1 + n_refs]], int(cfg[3]), int(cfg[0]), frame_qp - qp0, rows, sizes, sums, out, n, irap_poc=irap_poc)
            if slice_type == 2:
                irap_poc = poc
            if rc != 0:
                return case, f"NAL writer refused picture {f} (poc {poc})"
            mine += out[:n.value].tobytes()
        at = stream.find(b"\x00\x00\x01\x00\x41")
        if at <= 0 or stream[:at] + mine != stream:
            return case, "whole file: the parameter sets + the library's NAL units differ from the encoder's .266"
        if os.environ.get("SWEEP_EMUL"):          # ... and the P / B kernel's source on the host (tests/emul), against the same records
            for fr, d, prm, F, keep in H.iter_inter_frames(W2, H2, P):
                if int(d["meta"][6]) == 2:
                    continue
                got = H.emul_search_inter_picture(d2, prm, F, *pics[fr])
                msgs = H.compare_device_inter_picture(W2, H2, d, got)
                if msgs:
                    return case, f"kernel emulation, picture {fr}: {msgs[:2]}"
        return case, None
    except Exception as e:          # noqa: BLE001
        return case, f"{type(e).__name__}: {e}"
    finally:
        for f in os.listdir(tmp):
            os.remove(os.path.join(tmp, f))
        os.rmdir(tmp)


if __name__ == "__main__":
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n):
        W = int(rng.choice([64, 72, 136, 192, 200, 264]))
        Hh = int(rng.choice([64, 72, 136, 200]))
        gop = str(rng.choice(["lp", "8", "16", "16"]))
        extra = () if gop == "lp" else ("gop", gop)
        frames = int(rng.integers(3, 9)) if gop == "lp" else int(gop) + 1 + int(rng.integers(0, int(gop) + 1))
        if gop != "lp" and rng.random() < 0.4:
            extra += ("period", gop)          # a second intra period: CRA + RASL pictures
        if rng.random() < 0.4:
            extra += ("owf", "1")
        if rng.random() < 0.3:
            extra += ("rd", "1")
        clip = int(rng.choice([1, 1, 2, 3]))
        if clip == 3:
            frames = min(frames, 12)
        cases.append((W, Hh, int(rng.choice([8, 10])), int(rng.integers(17, 40)), frames, extra, clip))
    bad = 0
    with ProcessPoolExecutor(procs) as ex:
        for case, err in ex.map(one, cases):
            print(case, "OK" if err is None else "MISMATCH " + err, flush=True)
            bad += err is not None
    print(f"{n} encodes, {bad} with a mismatch")
    sys.exit(1 if bad else 0)
