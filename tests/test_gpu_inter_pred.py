"""uvghip_inter_pred_satd_batch (csrc/inter_pred.hip): luma motion compensation of candidate motions (uni- and bi-predicted, integer and
fractional vectors, blocks reaching outside the picture) + SATD against the source, against the oracle's uvg_inter_pred_pu +
uvg_satd_any_size (orcN_inter_pred_satd; inter_recon_unipred / uvg_inter_recon_bipred / uvg_bipred_average restated step by step, where
the device takes one code path: an integer vector is phase 0 of the interpolation filter)."""
import ctypes

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth,W,Hh", [(8, 192, 128), (10, 136, 72)])
def test_prediction_and_satd_equal_the_oracle(hip, orc, depth, W, Hh):
    import torch
    from uvg266_amd import api
    rng = np.random.default_rng(11 * depth + W)
    cur = np.ascontiguousarray(H.moving_picture(W, Hh, 4, depth)[0])
    refs = [np.ascontiguousarray(H.moving_picture(W, Hh, t, depth)[0]) for t in (3, 1, 0)]
    dcur, drefs = torch.from_numpy(cur).cuda(), [torch.from_numpy(r).cuda() for r in refs]
    tab = api.ref_table(drefs)
    ptrs = (ctypes.c_void_p * len(refs))(*[r.ctypes.data for r in refs])
    fn = orc.fn(depth, "inter_pred_satd")
    kinds = dict(bi=0, frac=0, outside=0, mixed=0)
    for size in (8, 16, 32, 64):
        n = {8: 300, 16: 160, 32: 60, 64: 24}[size]
        m = np.zeros(n, api.MOTION_NP)
        xs, ys = np.arange(0, W - size + 1, 8), np.arange(0, Hh - size + 1, 8)
        m["x"], m["y"] = rng.choice(xs, n), rng.choice(ys, n)
        m["dir"] = rng.integers(1, 4, n)
        m["ref"] = rng.integers(0, len(refs), (n, 2))
        m["mv"] = rng.integers(-200, 201, (n, 2, 2))
        whole = rng.random((n, 2)) < 0.35                       # integer vectors on a list (bi-prediction: pixels on one side, intermediates on the other)
        m["mv"][whole] = (m["mv"][whole] >> 4) << 4
        far = rng.random(n) < 0.1
        m["mv"][far] *= 12                                        # far outside the picture
        satd, pred = api.inter_pred_satd_batch(dcur, drefs, tab, torch.from_numpy(m.view(np.uint8)).cuda(), size, want_pred=True)
        torch.cuda.synchronize()
        satd, pred = satd.cpu().numpy(), pred.cpu().numpy()
        for k in range(n):
            mot = np.array([m["x"][k], m["y"][k], m["dir"][k], *m["ref"][k], *m["mv"][k].ravel()], np.int32)
            want_pred = np.zeros((size, size), cur.dtype)
            want = fn(W, Hh, H.ptr(cur), ptrs, len(refs), H.ptr(mot), size, H.ptr(want_pred))
            assert np.array_equal(pred[k], want_pred), (size, k, mot.tolist())
            assert int(satd[k]) == want, (size, k, mot.tolist())
        bi = m["dir"] == 3
        kinds["bi"] += int(bi.sum())
        kinds["frac"] += int(((m["mv"] & 15) != 0).any(axis=(1, 2)).sum())
        kinds["mixed"] += int((bi & (((m["mv"][:, 0] & 15) == 0).all(axis=1) != ((m["mv"][:, 1] & 15) == 0).all(axis=1))).sum())
        kinds["outside"] += int(far.sum())
    assert kinds["bi"] > 100 and kinds["frac"] > 200 and kinds["mixed"] > 20 and kinds["outside"] > 20, kinds
