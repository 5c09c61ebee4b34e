"""uvghip_merge_cand_batch / uvghip_amvp_cand_batch (csrc/inter_cand.hip, inter_cand_dev.h): the merge and AMVP candidate lists of
inter CUs on the device, one lane per call, against the calls the real encoder made during low-delay encodes (tests/golden/ref_merge_*,
ref_amvp_*: 3 047 calls, each recorded with everything uvg_inter_get_merge_cand / uvg_inter_get_mv_cand read) -- and against the oracle,
call by call, including the side effect on the neighbours' unused lists (inter_clear_cu_unused)."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("name", ["ref_merge_192x128_8_qp17_6frames", "ref_merge_136x72_10_qp27_8frames"])
def test_merge_candidates_equal_the_encoders(hip, orc, name):
    import torch
    from uvg266_amd import api
    g = {k: v for k, v in H.ctu_golden(name).items()}
    n = len(g["ctx"])
    lcu = _dev(g["lcu"].reshape(n, -1, 8).astype(np.int32))
    cands, counts = api.merge_cand_batch(_dev(g["ctx"]), lcu, _dev(g["col"]), _dev(g["hmvp"]))
    torch.cuda.synchronize()
    cands, counts, lcu_after = cands.cpu().numpy(), counts.cpu().numpy(), lcu.cpu().numpy()
    assert np.array_equal(counts, g["ctx"][:, 48])
    fn = orc.fn(8, "merge_candidates")
    for k in range(n):
        want, m = g["out"][k], int(counts[k])
        for i in range(m):                    # fields of a list a candidate does not use are whatever the encoder's array held before
            assert cands[k, i, 0] == want[i, 0], (k, i)
            for l in (0, 1):
                if want[i, 0] & (1 << l):
                    assert cands[k, i, 1 + l] == want[i, 1 + l] and (cands[k, i, 3 + 2 * l:5 + 2 * l] == want[i, 3 + 2 * l:5 + 2 * l]).all(), (k, i, l)
        # the oracle on the same call: every field of every candidate, and the table it leaves behind
        o_lcu = np.ascontiguousarray(g["lcu"][k]).copy()
        out = np.zeros((6, 7), np.int32)
        assert fn(H.ptr(np.ascontiguousarray(g["ctx"][k])), H.ptr(o_lcu), H.ptr(np.ascontiguousarray(g["col"][k])), H.ptr(np.ascontiguousarray(g["hmvp"][k])), H.ptr(out)) == m
        assert np.array_equal(cands[k], out), k
        assert np.array_equal(lcu_after[k], o_lcu.reshape(-1, 8)), k


@pytest.mark.parametrize("name", ["ref_amvp_192x128_8_qp17_6frames", "ref_amvp_136x72_10_qp27_8frames"])
def test_amvp_predictors_equal_the_encoders(hip, name):
    import torch
    from uvg266_amd import api
    g = {k: v for k, v in H.ctu_golden(name).items()}
    n = len(g["ctx"])
    out = api.amvp_cand_batch(_dev(g["ctx"]), _dev(g["lcu"].reshape(n, -1, 8).astype(np.int32)), _dev(g["col"]), _dev(g["hmvp"]))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().reshape(n, 4), g["out"].reshape(n, 4))


def test_empty_and_bad_arguments(hip):
    import ctypes
    import torch
    from uvg266_amd import lib
    L = lib.init(0)
    z = torch.zeros(64, dtype=torch.int32, device="cuda")
    assert L.uvghip_merge_cand_batch(z.data_ptr(), z.data_ptr(), z.data_ptr(), 0, z.data_ptr(), 0, z.data_ptr(), z.data_ptr(), None) == 0
    assert L.uvghip_merge_cand_batch(None, z.data_ptr(), z.data_ptr(), 0, z.data_ptr(), 1, z.data_ptr(), z.data_ptr(), None) != 0
    assert L.uvghip_amvp_cand_batch(z.data_ptr(), z.data_ptr(), z.data_ptr(), -1, z.data_ptr(), 1, z.data_ptr(), None) != 0
