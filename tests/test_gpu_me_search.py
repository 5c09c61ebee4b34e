"""uvghip_me_search_batch (csrc/me_search.hip): the motion search of one reference picture per prediction unit on the device -- starting
points, early termination, hexagon search, four fractional steps on SATD, motion-vector-difference costs against both AMVP predictors --
against the oracle's restatement of select_starting_point / early_terminate / hexagon_search / search_frac / select_mv_cand
(orcN_me_search_job; the oracle's search equals the real encoder call by call, tests/test_oracle_inter_search.py).  Every field of every
result: both vectors, both costs and bit counts (doubles, bit for bit), the chosen predictor, the early-termination flag."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def make_jobs(rng, W, Hh, size, n, n_refs):
    from uvg266_amd import api
    jobs = np.zeros(n, api.ME_JOB_NP)
    xs, ys = np.arange(0, W - size + 1, size), np.arange(0, Hh - size + 1, size)
    for j in jobs:
        j["x"], j["y"] = int(rng.choice(xs)), int(rng.choice(ys))
        j["ref"] = int(rng.integers(0, n_refs))
        far = rng.random() < 0.2                                   # predictors and starting points well outside the picture now and then
        span = 2000 if far else 160
        j["mv_cand"] = (rng.integers(-span, span + 1, (2, 2)) // 4) * 4
        if rng.random() < 0.3:
            j["mv_cand"][1] = j["mv_cand"][0]
        j["extra_mv"] = rng.integers(-span, span + 1, 2) if rng.random() < 0.7 else 0
        j["n_start"] = int(rng.integers(0, 7))
        j["start"] = rng.integers(-span, span + 1, (6, 2))
        if j["n_start"] and rng.random() < 0.3:
            j["start"][0] = (j["extra_mv"] >> 4) << 4                # the extra vector is one of the merge vectors: it is not tried twice
        if rng.random() < 0.15:                                      # search_frac alone around a given integer vector
            j["n_start"] = -1
            j["extra_mv"] = (j["extra_mv"] >> 4) << 4
    return jobs


def oracle_results(orc, depth, W, Hh, lam_sqrt, fme, cur, refs, jobs, size):
    from uvg266_amd import api
    out = np.zeros(len(jobs), api.ME_RESULT_NP)
    import ctypes
    ptrs = (ctypes.c_void_p * len(refs))(*[r.ctypes.data for r in refs])
    fn = orc.fn(depth, "me_search_job", None)
    for k, j in enumerate(jobs):
        job = np.zeros(24, np.int32)
        job[0:4] = j["x"], j["y"], size, j["ref"]
        job[4:8] = j["mv_cand"].ravel()
        job[8:10] = j["extra_mv"]
        job[10] = j["n_start"]
        job[11:23] = j["start"].ravel()
        oi, od = np.zeros(6, np.int32), np.zeros(4, np.float64)
        fn(W, Hh, ctypes.c_double(lam_sqrt), fme, H.ptr(cur), ptrs, len(refs), H.ptr(job), H.ptr(oi), H.ptr(od))
        o = out[k]
        o["mv"], o["int_mv"], o["mv_cand"], o["skipped_hexagon"] = oi[0:2], oi[2:4], oi[4], oi[5]
        o["cost"], o["bits"], o["int_cost"], o["int_bits"] = od
    return out


@pytest.mark.parametrize("depth,W,Hh,qp", [(8, 192, 128, 27), (10, 136, 72, 22), (8, 264, 136, 37)])
@pytest.mark.parametrize("fme", [4, 0])
def test_search_equals_the_oracle(hip, orc, depth, W, Hh, qp, fme):
    import torch
    from uvg266_amd import api
    rng = np.random.default_rng(7 * W + depth + fme)
    cur = np.ascontiguousarray(H.moving_picture(W, Hh, 3, depth)[0])
    refs = [np.ascontiguousarray(H.moving_picture(W, Hh, t, depth)[0]) for t in (2, 0)]
    top = (1 << depth) - 1
    refs.append(np.clip(refs[0].astype(np.int32) + rng.integers(-20, 21, refs[0].shape), 0, top).astype(refs[0].dtype))
    lam_sqrt = float(np.sqrt(0.57 * 2.0 ** ((qp - 12) / 3.0)))
    dcur, drefs = torch.from_numpy(cur).cuda(), [torch.from_numpy(r).cuda() for r in refs]
    tab = api.ref_table(drefs)
    seen = dict(frac=0, moved=0, skipped=0, outside=0)
    for size in (8, 16, 32, 64):
        n = {8: 400, 16: 200, 32: 80, 64: 30}[size]
        jobs = make_jobs(rng, W, Hh, size, n, len(refs))
        got = api.me_search_batch(dcur, drefs, tab, torch.from_numpy(jobs.view(np.uint8)).cuda(), size, lam_sqrt, fme)
        torch.cuda.synchronize()
        got = got.cpu().numpy().view(api.ME_RESULT_NP)
        want = oracle_results(orc, depth, W, Hh, lam_sqrt, fme, cur, refs, jobs, size)
        for k in range(n):
            assert got[k].tobytes() == want[k].tobytes(), (size, k, jobs[k], got[k], want[k])
        seen["frac"] += int(((want["mv"] & 15) != 0).any(axis=1).sum())
        seen["moved"] += int((want["int_mv"] != 0).any(axis=1).sum())
        seen["skipped"] += int(want["skipped_hexagon"].sum())
        seen["outside"] += int(((jobs["x"] + (want["int_mv"][:, 0] >> 4) < 0) | (jobs["y"] + (want["int_mv"][:, 1] >> 4) < 0) |
                                (jobs["x"] + (want["int_mv"][:, 0] >> 4) + size > W) | (jobs["y"] + (want["int_mv"][:, 1] >> 4) + size > Hh)).sum())
    assert seen["moved"] > 50 and seen["skipped"] > 10 and seen["outside"] > 5 and (fme == 0 or seen["frac"] > 50), seen


def test_arguments_are_checked(hip):
    import torch
    from uvg266_amd import lib
    L = lib.init(0)
    z = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    p = z.data_ptr()
    assert L.uvghip_me_search_batch(8, p, 64, p, 64, 64, 64, 1.0, 4, 8, p, 0, p, None) == 0          # nothing to do
    assert L.uvghip_me_search_batch(8, p, 64, p, 64, 64, 64, 1.0, 4, 12, p, 1, p, None) != 0         # not a CU size
    assert L.uvghip_me_search_batch(8, p, 32, p, 64, 64, 64, 1.0, 4, 8, p, 1, p, None) != 0          # stride below the picture width
    assert L.uvghip_me_search_batch(12, p, 64, p, 64, 64, 64, 1.0, 4, 8, p, 1, p, None) != 0         # not a depth of this build
    assert L.uvghip_me_search_batch(8, p, 64, p, 64, 64, 64, 0.0, 4, 8, p, 1, p, None) != 0
