"""uvghip_alf_reconstruct_picture (csrc/alf_picture.hip: classification, luma / chroma ALF by the encoder's decisions, CC-ALF) against the
real encoder's `--alf full` runs: the picture uvg_alf_enc_process got + its decisions -> the picture it left (tests/golden/ref_alf_*.npz,
the same goldens the oracle is held to in tests/test_oracle_alf_picture.py)."""
import os

import numpy as np
import pytest

import helpers as H
from test_oracle_alf_picture import GOLDENS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", GOLDENS)
def test_picture_after_alf_equals_the_encoders(hip, name):
    import torch
    from uvg266_amd import api
    g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
    W, Hh, depth, qp, frames = (int(a) for a in g["dims"][:5])
    filtered = 0
    for f in range(frames):
        m = g["meta"][f]
        pre = [torch.from_numpy(np.ascontiguousarray(g[k][f])).cuda() for k in ("pre_y", "pre_u", "pre_v")]
        n_aps = int(m[7])
        out = api.alf_reconstruct_picture(pre, m[4:7], g["flags"][f], g["set_idx"][f], g["luma_aps"][f][:n_aps], g["chroma_aps"][f], alf_full=int(m[3]) == 2,
                                          cc_alf_enabled=m[17:19], cc_coeff=g["cc_coeff"][f], classification_shift=int(m[28]) + 4)
        torch.cuda.synchronize()
        for o, k in zip(out, ("post_y", "post_u", "post_v")):
            bad = np.argwhere(o.cpu().numpy() != g[k][f])
            assert bad.size == 0, (name, f, k, len(bad), bad[:4].tolist())
        filtered += int(m[4])
    assert filtered > 0


def test_against_the_oracle_on_a_partial_ctu_picture(hip, orc):
    """264 x 136 (8-sample CTUs at both edges), decisions drawn at random: fixed sets, two APSs, every alternative and CC-ALF filter."""
    import ctypes
    import torch
    from uvg266_amd import api
    rng = np.random.default_rng(5)
    for depth in (8, 10):
        W, Hh = 264, 136
        n = 5 * 3
        pre = [np.ascontiguousarray(p) for p in H.varied_picture(W, Hh, 2003, depth)]
        flags = np.zeros((7, n), np.uint8)
        flags[0:3] = rng.integers(0, 2, (3, n)); flags[3:5] = rng.integers(0, 8, (2, n)); flags[5:7] = rng.integers(0, 5, (2, n))
        set_idx = rng.integers(0, 18, n).astype(np.int16)
        luma_aps = np.zeros((2, 677), np.int16)
        for a in range(2):
            luma_aps[a, :325] = rng.integers(-20, 21, 325); luma_aps[a, 325:650] = rng.integers(0, 4, 325)
            luma_aps[a, 650:675] = rng.integers(0, 3, 25); luma_aps[a, 675] = 3; luma_aps[a, 676] = a
        chroma_aps = np.zeros(114, np.int16)
        chroma_aps[:56] = rng.integers(-20, 21, 56); chroma_aps[56:112] = rng.integers(0, 4, 56); chroma_aps[112] = 8; chroma_aps[113] = 1
        cc = rng.integers(-64, 65, (2, 4, 8)).astype(np.int16)
        meta = np.zeros(32, np.int32)
        meta[3] = 2; meta[4:7] = 1; meta[7] = 2; meta[17:19] = 1; meta[28] = 8
        want = [np.zeros_like(p) for p in pre]
        fixed = np.ascontiguousarray(np.load(os.path.join(H.GOLDEN, "ref_alf_fixed.npy")), np.int16)
        rc = orc.fn(depth, "alf_reconstruct_picture", ctypes.c_int)(*(H.ptr(p) for p in pre), W, Hh, *(H.ptr(o) for o in want), H.ptr(meta), H.ptr(flags), H.ptr(set_idx),
                                                                    H.ptr(luma_aps), H.ptr(chroma_aps), H.ptr(np.ascontiguousarray(cc)), H.ptr(fixed))
        assert rc == 0
        out = api.alf_reconstruct_picture([torch.from_numpy(p).cuda() for p in pre], (1, 1, 1), flags, set_idx, luma_aps, chroma_aps, alf_full=True, cc_alf_enabled=(1, 1),
                                          cc_coeff=cc, classification_shift=12)
        torch.cuda.synchronize()
        for o, w_, k in zip(out, want, "yuv"):
            assert np.array_equal(o.cpu().numpy(), w_), (depth, k)


@pytest.mark.parametrize("name", ["ref_ctu_320x192_10_qp27_alf", "ref_ctu_192x128_8_qp22_alf"])
def test_slice_data_with_the_alf_syntax_equals_the_encoders(hip, name):
    """uvghip_encode_slice_rows_alf: the device's own search + filters of the golden's source (ALF does not change them), then the arithmetic
    coder with the CTU-level ALF syntax from the encoder's recorded decisions -> the slice data of the encoder's --alf full .266."""
    import torch
    from uvg266_amd import api
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    prm = H.search_params(W, Hh, qp)
    cl = api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v))])
    cl.run()
    m = g["alf_meta"]
    alf = dict(alf_type=m[3], enabled=m[4:7], n_luma_aps=m[7], n_alternatives_chroma=g["alf_chroma_aps"][112], cc_enabled=m[17:19], cc_filter_count=m[19:21],
               ctu_flags=g["alf_flags"], filter_set_idx=g["alf_set_idx"])
    out, nbytes = cl.encode_rows_alf([alf])
    nb = nbytes.cpu().numpy()[0]
    assert np.array_equal(np.concatenate([[0], np.cumsum(nb)]), g["row_off"])
    data = np.concatenate([out[0, r, :nb[r]].cpu().numpy() for r in range(len(nb))])
    assert np.array_equal(data, g["row_bytes"])
    # ... and the plain coder on the same picture is unchanged by the ALF code in the kernel: its rows are the oracle's / the other tests' business,
    # here only that they differ from the ALF rows (the bins are really written)
    out2, nb2 = cl.encode_rows()
    torch.cuda.synchronize()
    nb2 = nb2.cpu().numpy()[0]
    plain = np.concatenate([out2[0, r, :nb2[r]].cpu().numpy() for r in range(len(nb2))])
    assert len(plain) != len(data) or not np.array_equal(plain, data)
