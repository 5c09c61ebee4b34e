"""uvghip_ctu_search_pb against the oracle on content, sizes and tools the goldens do not hold: per case a short low-delay sequence --
picture 0 through the oracle's intra search, every later picture through the oracle's inter search with the oracle's earlier
reconstructions (unfiltered: any picture serves as a reference for this purpose) and motion as references -- and the device kernel on the
same inputs, all P / B pictures of a case in one launch.  Compared per picture: side information, used-list motion and flags,
reconstruction, levels, the 257 + 18 models at the three points of every CTU.  The frame-level state (QP / lambda per GOP position,
reference lists) is that of the reference encoder's runs kept in the goldens; the tools vary per case."""
import ctypes
import os
import numpy as np
import pytest
import helpers as H

pytestmark = pytest.mark.gpu


from sweep_inter_common import CASES, oracle_chain, oracle_as_record  # noqa: E402


@pytest.mark.parametrize("case", range(len(CASES)))
def test_device_search_equals_the_oracle(hip, case):
    import torch
    from uvg266_amd import api, lib
    W, Hh, depth, pics, jobs = oracle_chain(case)
    wc, hc = (W + 63) // 64, (Hh + 63) // 64
    n4, ctus = hc * 16 * wc * 16, wc * hc
    assert jobs
    # ---- the device on the same inputs, every P / B picture in one launch ----
    tdt = torch.uint8 if depth == 8 else torch.uint16
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    cache, descs, tens = {}, [], []
    for f, fs, prm, F, keep, r in jobs:
        q = lib.CtuPbPicture()
        cp = H.ctu_params(prm)
        ctypes.memmove(ctypes.byref(q.params), ctypes.byref(cp), ctypes.sizeof(cp))
        t = dict(src=[dev(p) for p in pics[f]], rec=[torch.zeros((Hh >> c, W >> c), dtype=tdt, device="cuda") for c in (0, 1, 1)],
                 scu=torch.zeros(n4 * 32, dtype=torch.uint8, device="cuda"), i4=torch.zeros(n4 * 8, dtype=torch.uint8, device="cuda"),
                 trees=torch.zeros(n4, dtype=torch.int32, device="cuda"), mot=torch.zeros(n4 * 8, dtype=torch.int32, device="cuda"),
                 co=torch.zeros(ctus * 6144, dtype=torch.int16, device="cuda"), mo=torch.zeros(ctus * 3 * 257, dtype=torch.int32, device="cuda"),
                 mi=torch.zeros(ctus * 3 * 18, dtype=torch.int32, device="cuda"), refs=[])
        c = q.pic
        c.src_y, c.src_u, c.src_v = (a.data_ptr() for a in t["src"])
        c.rec_y, c.rec_u, c.rec_v = (a.data_ptr() for a in t["rec"])
        c.src_stride = c.rec_stride = W
        c.src_stride_c = c.rec_stride_c = W // 2
        c.cu, c.cu_stride, c.coeff, c.models = t["scu"].data_ptr(), wc * 16, t["co"].data_ptr(), t["mo"].data_ptr()
        for nme in ("slice_type", "poc", "n_refs", "tmvp", "max_merge", "merge_level", "frame_qp", "bipred", "fme_level", "early_skip", "depth_inter_min", "depth_inter_max"):
            setattr(q, nme, getattr(F, nme))
        for i in range(16):
            q.ref_pocs[i], q.l[0][i], q.l[1][i] = F.ref_pocs[i], F.l[0][i], F.l[1][i]
        q.l_size[0], q.l_size[1] = F.l_size[0], F.l_size[1]
        q.ref_stride, q.ref_stride_c, q.ref_motion_stride = W, W // 2, wc * 16
        for i in range(F.n_refs):
            key = F.ref_pocs[i]
            if key not in cache:
                cache[key] = [dev(a) for a in keep[4 * i:4 * i + 4]]
            pl = cache[key]
            q.ref_y[i], q.ref_u[i], q.ref_v[i], q.ref_motion[i] = (a.data_ptr() for a in pl)
        q.inter4, q.models_inter, q.trees, q.motion_out = t["i4"].data_ptr(), t["mi"].data_ptr(), t["trees"].data_ptr(), t["mot"].data_ptr()
        descs.append(q); tens.append(t)
    ws = api.ctu_search_pb(descs, depth)
    torch.cuda.synchronize()
    seen = dict(inter=0, intra=0, skipped=0, bi=0)
    for (f, fs, prm, F, keep, r), t in zip(jobs, tens):
        got = H.inter_result_from_device_layout(
            W, Hh, *(a.cpu().numpy() for a in t["rec"]), t["scu"].cpu().numpy().view(H.SCU_NP), t["i4"].cpu().numpy().view(H.INTER4_NP),
            t["trees"].cpu().numpy().view(np.uint32), t["mot"].cpu().numpy().reshape(n4, 8), t["co"].cpu().numpy(), t["mo"].cpu().numpy().view(np.uint32),
            t["mi"].cpu().numpy().view(np.uint32))
        d = oracle_as_record(r)
        m = d["motion"]
        msgs = H.compare_device_inter_picture(W, Hh, d, got)
        assert msgs == [], (CASES[case], "picture", f, msgs[:4])
        h4, w4 = Hh // 4, W // 4
        typ = r["cu"][:h4, :w4, 0]
        seen["inter"] += int((typ == 2).sum()); seen["intra"] += int((typ == 1).sum())
        seen["skipped"] += int(((m[:h4, :w4, 7] & 1) != 0)[typ == 2].sum()); seen["bi"] += int((m[:h4, :w4, 6] == 3)[typ == 2].sum())
    print(CASES[case], seen)
    assert seen["inter"] > 0
