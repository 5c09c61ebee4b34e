"""The reconstruction half of the per-picture ALF process (oracle/orc_alf_picture.c: alf_reconstruct, the APS / fixed-set tables, CC-ALF)
against the real encoder: `--alf full` all-intra runs with the picture uvg_alf_enc_process got, the decisions it took and the picture it
left (tools/refcheck/ctu_dump.c record "alf", tests/golden/ref_alf_*.npz)."""
import ctypes
import os

import numpy as np
import pytest

import helpers as H

GOLDENS = ["ref_alf_320x192_10_qp27_3frames", "ref_alf_192x128_8_qp27_3frames", "ref_alf_192x128_10_qp23_2frames"]


def reconstruct(orc, g, f):
    W, Hh, depth = (int(a) for a in g["dims"][:3])
    px = H.px_dtype(depth)
    pre = [np.ascontiguousarray(g[k][f], px) for k in ("pre_y", "pre_u", "pre_v")]
    out = [np.zeros_like(p) for p in pre]
    fixed = np.ascontiguousarray(np.load(os.path.join(H.GOLDEN, "ref_alf_fixed.npy")), np.int16)
    fn = orc.fn(depth, "alf_reconstruct_picture", ctypes.c_int)
    rc = fn(*(H.ptr(p) for p in pre), W, Hh, *(H.ptr(o) for o in out), H.ptr(np.ascontiguousarray(g["meta"][f], np.int32)),
            H.ptr(np.ascontiguousarray(g["flags"][f], np.uint8)), H.ptr(np.ascontiguousarray(g["set_idx"][f], np.int16)),
            H.ptr(np.ascontiguousarray(g["luma_aps"][f], np.int16)), H.ptr(np.ascontiguousarray(g["chroma_aps"][f], np.int16)),
            H.ptr(np.ascontiguousarray(g["cc_coeff"][f], np.int16)), H.ptr(fixed))
    assert rc == 0
    return out


def check(orc, name):
    g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
    frames = int(g["dims"][4])
    seen = dict(off=0, new_aps=0, fixed=0, alternatives=0, cc=0)
    for f in range(frames):
        out = reconstruct(orc, g, f)
        for o, k in zip(out, ("post_y", "post_u", "post_v")):
            bad = np.argwhere(o != g[k][f])
            assert bad.size == 0, (name, f, k, bad[:4].tolist())
        m, fl = g["meta"][f], g["flags"][f]
        if m[4] and fl[0].any():          # the classification the encoder's luma filter worked from (taken through the strategy pointer)
            W, Hh, depth = (int(a) for a in g["dims"][:3])
            mine = orc.alf_classify_frame(depth, np.ascontiguousarray(g["pre_y"][f]), W, Hh, int(m[28]) + 4)
            assert np.array_equal(mine, g["cls"][f]), (name, f, "classification")
        seen["off"] += int(m[4] == 0)
        seen["new_aps"] += int(m[4] and (g["set_idx"][f][fl[0] > 0] >= 16).any())
        seen["fixed"] += int(m[4] and (g["set_idx"][f][fl[0] > 0] < 16).any())
        seen["alternatives"] += int(m[5] and fl[3].max() > 0)
        seen["cc"] += int(m[17] and fl[5].max() > 1)
    return seen


@pytest.mark.parametrize("name", GOLDENS)
def test_picture_after_alf_equals_the_encoders(orc, name):
    check(orc, name)


def test_the_goldens_cover_the_branches(orc):
    """A picture ALF leaves alone, filters from a new APS, the fixed filter sets, chroma alternatives, CC-ALF with several filters."""
    tot = {}
    for name in GOLDENS:
        for k, v in check(orc, name).items():
            tot[k] = tot.get(k, 0) + v
    assert all(v > 0 for v in tot.values()), tot


@pytest.mark.parametrize("name", GOLDENS)
def test_the_frame_statistics_equal_the_encoders(orc, name):
    """What alf_derive_stats_for_filtering (alf.c:4227) gathered in a real --alf full run -- the per-CTU covariances of every luma class and of
    the two chroma planes, summed over the picture (ctu_dump.c takes them while the process still holds them; cov_luma / cov_chroma) -- against
    the oracle's get_blk_stats on the picture ALF got and the run's source picture, CTU by CTU, summed the same way.  int64 sums and pix_acc
    exact."""
    g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
    W, Hh, depth, qp, frames, t0, kind = (int(a) for a in g["dims"])
    checked = 0
    for f in range(frames):
        if not int(g["meta"][f][29]):
            continue                      # (no CTU of the picture was filtered: the run's statistics were freed unseen)
        src = H.varied_picture(W, Hh, kind * 1000 + t0 + f, depth)
        import zlib
        assert zlib.crc32(b"".join(p.tobytes() for p in src)) == int(g["src_crc"][f])
        pre = [np.ascontiguousarray(g[k][f]) for k in ("pre_y", "pre_u", "pre_v")]
        cls = orc.alf_classify_frame(depth, pre[0], W, Hh, int(g["meta"][f][28]) + 4)
        luma, chroma = np.zeros((25, 1509), np.int64), np.zeros((2, 1509), np.int64)
        for y in range(0, Hh, 64):
            for x in range(0, W, 64):
                w, h = min(64, W - x), min(64, Hh - y)
                e, yv, pa = orc.alf_stats_rect(depth, np.ascontiguousarray(src[0]), pre[0], W, Hh, x, y, w, h, False, cls)
                luma += H.alf_sum_layout(e, yv, pa, 13)
                for c in (1, 2):
                    e, yv, pa = orc.alf_stats_rect(depth, np.ascontiguousarray(src[c]), pre[c], W // 2, Hh // 2, x // 2, y // 2, w // 2, h // 2, True, None)
                    chroma[c - 1] += H.alf_sum_layout(e, yv, pa, 7)[0]
        assert np.array_equal(luma, g["cov_luma"][f]), (name, f, "luma", np.argwhere(luma != g["cov_luma"][f])[:4].tolist())
        assert np.array_equal(chroma, g["cov_chroma"][f]), (name, f, "chroma", np.argwhere(chroma != g["cov_chroma"][f])[:4].tolist())
        checked += 1
    assert checked >= 2
