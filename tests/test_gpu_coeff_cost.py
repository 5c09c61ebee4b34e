"""uvghip_coeff_cost_batch (CABAC bit cost of coefficient blocks, one lane per block) vs the reference-run vectors and
the oracle: bits must be the same doubles."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", [8, 10])
def test_goldens(hip, depth):
    import torch
    from uvg266_amd import api
    for c in H.coeffcost_goldens(depth):
        blk = np.stack([c["coeff"].reshape(c["h"], c["w"])] * 3)
        bits, flags = api.coeff_cost_batch(torch.from_numpy(blk).cuda(), c["color"], c["models"])
        assert bits.cpu().numpy().tolist() == [c["bits"]] * 3, (c["w"], c["h"], c["color"], c["style"])
        assert flags.cpu().numpy().tolist() == [c["flags"]] * 3


@pytest.mark.parametrize("w,h", [(4, 4), (8, 8), (16, 16), (32, 32), (8, 4), (4, 16), (32, 8), (16, 32)])
@pytest.mark.parametrize("color", [0, 1])
def test_random_batches_vs_oracle(hip, orc, w, h, color):
    """Ragged batch sizes (partial workgroups), mixed level statistics incl. dense blocks that spend the regular bins."""
    import torch
    from uvg266_amd import api
    rng = np.random.default_rng(w * 64 + h + color)
    models = H.coeffcost_goldens(8)[3 + color]["models"]
    n = 203 if w * h <= 256 else 37
    fall = 1.0 / (1.0 + 0.3 * (np.arange(w)[None, :] + np.arange(h)[:, None]))
    coeff = np.zeros((n, h, w), np.int16)
    for i in range(n):
        s = i % 5
        if s == 0:
            coeff[i] = rng.integers(-3, 4, (h, w))
        elif s == 1:
            coeff[i] = np.where(rng.random((h, w)) < fall, rng.integers(-2, 3, (h, w)), 0)
        elif s == 2:
            coeff[i] = np.where(rng.random((h, w)) < fall ** 2, rng.integers(-60, 61, (h, w)), 0)
        elif s == 3:
            coeff[i, 0, 0] = rng.integers(-2000, 2001)
        # s == 4: empty
    bits, flags = api.coeff_cost_batch(torch.from_numpy(coeff).cuda(), color, models)
    bits, flags = bits.cpu().numpy(), flags.cpu().numpy()
    for i in range(n):
        wb, wf, _ = orc.coeff_cost(8, coeff[i].ravel(), w, h, color, models)
        assert bits[i] == wb and flags[i] == wf, (i, i % 5)
    assert (bits[4::5] == 0).all() and bits.max() > 100


def test_picture_sized_batch(hip, orc):
    """All 8x8 luma blocks of a 1080p picture's worth of levels in one launch; 512 sampled blocks vs the oracle."""
    import torch
    from uvg266_amd import api
    rng = np.random.default_rng(9)
    n = 32400
    models = H.coeffcost_goldens(8)[0]["models"]
    coeff = np.where(rng.random((n, 8, 8)) < 0.2, rng.integers(-4, 5, (n, 8, 8)), 0).astype(np.int16)
    bits, _ = api.coeff_cost_batch(torch.from_numpy(coeff).cuda(), 0, models)
    bits = bits.cpu().numpy()
    for i in rng.permutation(n)[:512]:
        assert bits[i] == orc.coeff_cost(8, coeff[i].ravel(), 8, 8, 0, models)[0]
