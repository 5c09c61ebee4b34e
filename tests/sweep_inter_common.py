"""Shared by tests/test_gpu_inter_sweep.py (device) and tests/test_ctu_pb_emulation.py (the kernel's source on the host): the cases of the
P / B search sweep, their synthetic sequences, and the oracle's chain over a case's pictures."""
import os
import numpy as np
import helpers as H


def sequence(kind, W, Hh, depth, frames, seed):
    """Synthetic sequences: 'pan' a noisy picture moving by whole and fractional samples, 'fast' large motion (vectors leave the picture),
    'noise' independent noise on a still (little inter gain: intra CUs inside B pictures), 'still' identical pictures (everything skipped)."""
    rng = np.random.default_rng(seed)
    mx = (1 << depth) - 1
    big = H.varied_picture(4 * (W + 160), 4 * (Hh + 160), 2100 + seed, depth)
    out = []
    for t in range(frames):
        pic = []
        for b, c in zip(big, (0, 1, 1)):
            w, h = W >> c, Hh >> c
            if kind == "pan":
                sx, sy = (80 + 7 * t) >> c, (120 + 5 * t) >> c
            elif kind == "fast":
                sx, sy = ((40 + 150 * t) % 640) >> c, ((400 - 90 * t) % 640) >> c
            else:
                sx, sy = 64 >> c, 64 >> c
            a = b.astype(np.int64)[sy:sy + 4 * h, sx:sx + 4 * w]
            p = (a.reshape(h, 4, w, 4).sum(axis=(1, 3)) + 8) >> 4
            if kind == "noise":
                p = p + rng.integers(-24, 25, p.shape) * (1 << (depth - 8))
            pic.append(np.clip(p, 0, mx).astype(b.dtype))
        out.append(tuple(pic))
    return out


CASES = [  # kind, W, H, golden the frame states come from, tools (tmvp, max_merge, merge_level, bipred, fme_level, early_skip)
    ("pan", 128, 64, "ref_inter_264x136_8_qp32_9frames", (1, 6, 2, 1, 4, 1)),
    ("fast", 136, 72, "ref_inter_192x128_8_qp17_5frames", (1, 6, 2, 1, 4, 1)),
    ("noise", 72, 136, "ref_inter_192x128_8_qp17_5frames", (1, 6, 2, 1, 4, 1)),
    ("still", 64, 64, "ref_inter_264x136_8_qp32_9frames", (1, 6, 2, 1, 4, 1)),
    ("pan", 200, 72, "ref_inter_136x72_10_qp22_4frames", (1, 6, 2, 1, 4, 1)),
    ("fast", 64, 128, "ref_inter_136x72_10_qp22_4frames", (0, 6, 2, 1, 0, 1)),
    ("noise", 136, 72, "ref_inter_264x136_8_qp32_9frames", (1, 5, 2, 0, 4, 0)),
    ("pan", 72, 72, "ref_inter_192x128_8_qp17_5frames", (0, 5, 2, 1, 4, 1)),
    ("fast", 192, 64, "ref_inter_264x136_8_qp32_9frames", (1, 6, 2, 1, 4, 0)),
    ("noise", 64, 72, "ref_inter_136x72_10_qp22_4frames", (1, 6, 2, 1, 4, 1)),
]



def oracle_chain(case):
    """-> (W, H, depth, pics, jobs): jobs = [(picture index, frame state, SearchParams, InterFrame, keep-alive list, the oracle's result)] of the
    case's P / B pictures; every picture's references are the oracle's own earlier (unfiltered) reconstructions and motion."""
    kind, W, Hh, gname, tools = CASES[case]
    g = np.load(os.path.join(H.GOLDEN, gname + ".npz"))
    depth, frames = int(g["dims"][2]), int(g["dims"][4])
    first = {}
    for k in range(len(g["meta"])):
        first.setdefault(int(g["meta"][k][0]), k)
    ks = [first[f] for f in range(frames)]
    states = H.frame_states_from_records(g["meta"][ks], g["lam"][ks], g["refs"][ks])
    pics = sequence(kind, W, Hh, depth, frames, case)
    orc = H.load_oracle()
    wc, hc = (W + 63) // 64, (Hh + 63) // 64
    n4, ctus = hc * 16 * wc * 16, wc * hc
    by_poc, jobs = {}, []
    for f, fs in enumerate(states):
        prm = H.SearchParams(W, Hh, fs["qp"], fs["qp"], 1, 4, 1, 1, 2, 0, fs["lam"], fs["lam_sqrt"], fs["c_lam"], fs["cw_u"], fs["cw_v"])
        if fs["slice_type"] == 2:
            r = H.oracle_search_picture(orc, depth, prm, *pics[f])
            mot = np.zeros((hc * 16, wc * 16, 8), np.int32)
        else:
            F = H.InterFrame()
            F.slice_type, F.poc, F.n_refs = fs["slice_type"] if tools[3] else 1, fs["poc"], fs["n_refs"]
            for i in range(16):
                F.ref_pocs[i], F.l[0][i], F.l[1][i] = fs["ref_pocs"][i], fs["lists"][0][i], fs["lists"][1][i]
            F.l_size[0], F.l_size[1] = fs["l_size"][0], (fs["l_size"][1] if tools[3] else 0)
            F.tmvp, F.max_merge, F.merge_level, F.bipred, F.fme_level, F.early_skip = tools
            F.depth_inter_min, F.depth_inter_max, F.ref_cu_stride, F.frame_qp = 0, 3, wc * 16, fs["frame_qp"]
            keep = []
            for i in range(F.n_refs):
                planes, rm = by_poc[F.ref_pocs[i]]
                keep += list(planes) + [rm]
                F.ref_y[i], F.ref_u[i], F.ref_v[i], F.ref_cu[i] = planes[0].ctypes.data, planes[1].ctypes.data, planes[2].ctypes.data, rm.ctypes.data
            r = H.oracle_search_inter_picture(orc, depth, prm, F, *pics[f], keep)
            mot = r["motion"]
            jobs.append((f, fs, prm, F, keep, r))
        own = ([fs["ref_pocs"][fs["lists"][0][i]] for i in range(fs["l_size"][0])], [fs["ref_pocs"][fs["lists"][1][i]] for i in range(fs["l_size"][1])])
        by_poc[fs["poc"]] = ([np.ascontiguousarray(r[k]) for k in ("rec_y", "rec_u", "rec_v")], H.ref_cu_table(r["cu"], mot, own))
    return W, Hh, depth, pics, jobs


def oracle_as_record(r):
    """The oracle's result of a P / B picture in the layout of the encoder's records (what compare_device_inter_picture takes)."""
    m = r["motion"].copy()
    e = r["extra"].astype(np.int32)
    m[:, :, 7] = m[:, :, 7] | (e[:, :, 1] << 8) | (e[:, :, 2] << 11) | (e[:, :, 0] << 14)
    cu12 = np.zeros(r["cu"].shape[:2] + (12,), np.uint8)
    cu12[:, :, :11] = r["cu"]
    return dict(cu=cu12, trees=r["trees"], motion=m, rec=[r["rec_y"], r["rec_u"], r["rec_v"]], coeff=r["coeff"], models=r["models"], models_inter=r["models_inter"])
