"""The closed-loop CTU search of ONE picture sharded over ranks by CTU rows (uvghip_ctu_plan_create_rows +
bands.BandLayout.halo_search): N emulated ranks on one GPU, each with its own full-size buffers POISONED outside what it produces or
receives, the halo of the row above (last reconstruction line, last row of side information, the models after that row's first CTU)
moved between them by the same exchange lists the RCCL transport executes.  Every band's rows equal the records of the reference
encoder's whole-picture run (tests/golden/ref_ctu*.npz)."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def poison(t, seed):
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    if t.dtype in (torch.uint16,):
        t.copy_(torch.randint(0, 1024, t.shape, generator=g, device="cuda", dtype=torch.int32).to(torch.uint16))
    elif t.dtype == torch.uint8:
        t.copy_(torch.randint(0, 256, t.shape, generator=g, device="cuda", dtype=torch.int32).to(torch.uint8))
    else:
        t.copy_(torch.randint(-20000, 20000, t.shape, generator=g, device="cuda", dtype=torch.int32).to(t.dtype))


@pytest.mark.parametrize("name,nranks", [("ref_ctu_832x480_8_qp22", 2), ("ref_ctu_832x480_8_qp22", 3), ("ref_ctu_832x480_8_qp22", 4), ("ref_ctu_416x240_10_qp37", 2),
                                         ("ref_ctu_416x240_10_qp37", 4)])
def test_bands_of_the_closed_loop_search_equal_the_reference_run(hip, name, nranks):
    import torch
    from uvg266_amd import api, bands
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    prm = H.search_params(W, Hh, qp)
    wc, hc = (W + 63) // 64, (Hh + 63) // 64
    P = api.ctu_params(W, Hh, qp, lam=prm.lam)
    src = tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v))
    ranks, lays, specs = [], [], []
    for r in range(nranks):
        lay = bands.BandLayout(Hh, nranks, r)
        cs = api.CtuSearch(P, [src], rows=(lay.ctu_row0, lay.ctu_row1))
        for k, t in enumerate(list(cs.rec[0]) + [cs.cu[0], cs.coeff[0], cs.models[0]]):
            poison(t, 1000 * r + k)                       # nothing of a previous run, nothing another rank should have produced
        scu2d = cs.cu[0].view(hc * 16, wc * 16 * 32)
        mod2d = cs.models[0].view(wc * hc, 3 * 257)
        ranks.append(cs); lays.append(lay)
        specs.append(lay.halo_search(cs.rec[0][0], cs.rec[0][1], cs.rec[0][2], scu2d, mod2d, wc))
    sent = 0
    for r in range(nranks):
        if r > 0:                                         # what the band above sends down, as the transports would move it
            pair = [[(1 if p == r else 0, s, rc) for p, s, rc in specs[r - 1] if p == r], [(0 if p == r - 1 else 1, s, rc) for p, s, rc in specs[r] if p == r - 1]]
            bands.emulate(pair)
            sent += bands.spec_bytes(specs[r])[1]
        ranks[r].run()
        torch.cuda.synchronize()
    assert nranks == 1 or sent > 0
    h4, w4 = Hh // 4, W // 4
    for r, (cs, lay) in enumerate(zip(ranks, lays)):
        ry, ru, rv = (t.cpu().numpy() for t in cs.rec[0])
        scu = cs.cu[0].cpu().numpy().reshape(-1).view(H.SCU_NP)
        res = H.search_result_from_device_layout(W, Hh, ry, ru, rv, scu, cs.coeff[0].cpu().numpy(), cs.models[0].cpu().numpy().view(np.uint32))
        y0, y1 = lay.y0, min(lay.y1, Hh)
        assert np.array_equal(res["rec_y"][y0:y1], g["rec_y"][y0:y1]), (r, "rec_y")
        assert np.array_equal(res["rec_u"][y0 // 2:y1 // 2], g["rec_u"][y0 // 2:y1 // 2]) and np.array_equal(res["rec_v"][y0 // 2:y1 // 2], g["rec_v"][y0 // 2:y1 // 2]), (r, "chroma")
        assert np.array_equal(res["cu"][y0 // 4:y1 // 4, :w4], g["cu"][y0 // 4:y1 // 4, :w4]), (r, "cu")
        assert np.array_equal(res["trees"][y0 // 4:y1 // 4, :w4], g["trees"][y0 // 4:y1 // 4, :w4]), (r, "trees")
        k0, k1 = lay.ctu_row0 * wc, lay.ctu_row1 * wc
        assert np.array_equal(res["models"][k0:k1], g["models"][k0:k1]), (r, "models")
        crc = H.ctu_crcs(dict(res), W, Hh)[:, 2]
        want = H.ctu_crcs(dict(res, coeff=g["coeff"]), W, Hh)[:, 2]
        assert np.array_equal(crc[k0:k1], want[k0:k1]), (r, "levels")
