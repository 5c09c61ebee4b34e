"""Deblocking of P / B pictures of real low-delay encodes (BASELINE configs[2], SAO off so that the encoder's output picture is the
deblocked one): from the reconstruction before the filters and the side information with motion (boundary strength from motion
vector differences and reference pictures, src/filter.c:734-818) the whole-picture deblocking gives the encoder's picture --
through the oracle (CPU) and through uvghip_deblock_frame (GPU).  Records: tests/golden/ref_inter_*_nosao.npz."""
import numpy as np
import pytest

import helpers as H

NAMES = ["ref_inter_192x128_8_qp32_5frames_nosao", "ref_inter_136x72_10_qp22_4frames_nosao"]


def frames_of(g):
    W, Hh, depth, qp, frames = (int(a) for a in g["dims"])
    for fr in range(frames):
        meta = g["meta"][[k for k in range(len(g["meta"])) if int(g["meta"][k][0]) == fr][0]]
        yield fr, W, Hh, depth, int(meta[6]) == 0, H.inter_scu_table(g, fr)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_deblocks_inter_pictures_as_the_encoder(orc, name):
    g = H.ctu_golden(name)
    strengths = 0
    for fr, W, Hh, depth, is_b, scu in frames_of(g):
        y, u, v = (np.ascontiguousarray(g[k][fr]).copy() for k in ("rec_y", "rec_u", "rec_v"))
        orc.deblock_frame(depth, y, u, v, W, Hh, scu.view(np.uint8).reshape(scu.shape[0], -1), scu.shape[1], 0, 0, is_b, -1, None)
        for a, k in ((y, "final_y"), (u, "final_u"), (v, "final_v")):
            assert np.array_equal(a, g[k][fr]), (name, fr, k)
        strengths += int((y != g["rec_y"][fr]).sum())
    assert strengths > 100           # the filter really changed samples (10 143 at QP 32)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_deblocks_inter_pictures_as_the_encoder(hip, name):
    import torch
    from uvg266_amd import api
    g = H.ctu_golden(name)
    for fr, W, Hh, depth, is_b, scu in frames_of(g):
        y, u, v = (torch.from_numpy(np.ascontiguousarray(g[k][fr])).cuda() for k in ("rec_y", "rec_u", "rec_v"))
        table = torch.from_numpy(np.ascontiguousarray(scu).view(np.uint8).reshape(scu.shape[0], -1)).cuda()
        api.deblock_frame(y, u, v, table, W, Hh, slice_is_b=is_b)
        for a, k in ((y, "final_y"), (u, "final_u"), (v, "final_v")):
            assert np.array_equal(a.cpu().numpy(), g[k][fr]), (name, fr, k)
