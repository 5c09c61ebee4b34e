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


@pytest.mark.parametrize("name", ["ref_inter_192x128_8_qp17_5frames", "ref_inter_136x72_10_qp22_4frames"])
def test_oracle_filters_inter_pictures_with_sao_as_the_encoder(orc, name):
    """The whole in-loop chain on P / B pictures with SAO on: per-CTU deblocking in the encoder's order (B-slice rule), the SAO
    decision of every CTU with the slice type's context initialisation and the picture's own lambda, SAO of the deblocked picture
    -> the encoder's decisions and the picture it outputs (which the next pictures predict from)."""
    g = {k: v for k, v in H.ctu_golden(name).items()}
    W, Hh, depth, qp0, frames = (int(a) for a in g["dims"])
    decided = 0
    for fr in range(frames):
        ks = [k for k in range(len(g["meta"])) if int(g["meta"][k][0]) == fr]
        meta = g["meta"][ks[0]]
        src = H.moving_picture(W, Hh, fr, depth)
        import zlib
        assert zlib.crc32(b"".join(p.tobytes() for p in src)) == int(g["src_crc"][fr])
        slice_type = int(meta[6])                       # UVG_SLICE_B = 0, P = 1, I = 2: also the row of the initialisation table
        r = H.oracle_sao_picture(orc, depth, W, Hh, int(meta[7]), float(g["lam"][ks[0]][0]), src, (g["rec_y"][fr], g["rec_u"][fr], g["rec_v"][fr]),
                                 H.inter_scu_table(g, fr), slice_type=slice_type)
        want = np.stack([g["sao"][k] for k in ks])
        assert np.array_equal(H.sao_info_comparable(r["sao"]), H.sao_info_comparable(want)), (name, fr)
        for k in ("final_y", "final_u", "final_v"):
            assert np.array_equal(r[k], g[k][fr]), (name, fr, k)
        decided += int((want[:, :, 0] != 0).sum())
    assert decided > 5
