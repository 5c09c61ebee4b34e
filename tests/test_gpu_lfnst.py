"""HIP LFNST vs the reference goldens and the oracle (bit-exact)."""
import numpy as np
import pytest

import helpers as H
from test_gpu_picture import dev

pytestmark = pytest.mark.gpu


def test_vs_reference_goldens(hip):
    from uvg266_amd import api
    n = 0
    for name, (hdr, src, want) in H.read_golden("lfnst", 8):
        w, h, mode, idx, inverse = (int(v) for v in hdr)
        c = dev(src.reshape(1, h, w))
        api.lfnst_batch(c, api.make_lfnst_tus([[mode, idx, w.bit_length() - 1, h.bit_length() - 1]]), bool(inverse))
        assert np.array_equal(c.cpu().numpy().ravel(), want), (w, h, mode, idx, inverse)
        n += 1
    assert n >= 300


@pytest.mark.parametrize("shape", [(4, 4), (8, 8), (16, 16), (32, 32), (4, 16), (32, 8), (8, 4), (16, 32)])
@pytest.mark.parametrize("inverse", [False, True])
def test_batch_vs_oracle(hip, orc, shape, inverse):
    """Batches with every mode, both kernels, idx 0 (untouched), CU shapes that differ from the TU (chroma-style
    wide-angle inputs) and large magnitudes (16-bit wrap of the inverse)."""
    from uvg266_amd import api
    w, h = shape
    rng = np.random.default_rng(w * 7 + h + int(inverse))
    n = 67 * 3 + 40
    coeffs = rng.integers(-1200, 1201, (n, h, w)).astype(np.int16)
    coeffs[::9] = (coeffs[::9].astype(np.int32) * 27).clip(-32768, 32767).astype(np.int16)
    rows = []
    for i in range(n):
        mode = i % 67
        idx = (i // 67) % 3
        lw, lh = w.bit_length() - 1, h.bit_length() - 1
        if i >= 67 * 3:                              # other CU shapes for the wide-angle rule
            lw, lh = 2 + int(rng.integers(0, 4)), 2 + int(rng.integers(0, 4)); idx = 1 + i % 2
        rows.append([mode, idx, lw, lh])
    want = coeffs.copy()
    f = orc.lib.orc_lfnst_inv if inverse else orc.lib.orc_lfnst_fwd
    for i, (mode, idx, lw, lh) in enumerate(rows):
        c = np.ascontiguousarray(want[i]); f(H.ptr(c), w, h, mode, lw, lh, idx); want[i] = c
    got = dev(coeffs)
    api.lfnst_batch(got, api.make_lfnst_tus(rows), inverse)
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(want[67:134:1][0], want[67]) and not np.array_equal(want[67], coeffs[67])   # idx 1 changes the TU
    assert np.array_equal(want[:67], coeffs[:67])                                                      # idx 0 does not


def test_full_size_4k(hip, orc):
    """3840x2160 worth of 8x8 TUs (129600): spot-check 64 TUs against the oracle, everything else through the
    untouched-region property."""
    import torch
    from uvg266_amd import api
    n, w, h = 129600, 8, 8
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    coeffs = torch.randint(-500, 501, (n, h, w), generator=g, device="cuda", dtype=torch.int16)
    modes = torch.arange(n, device="cuda") % 67
    rows = torch.stack([modes, 1 + (torch.arange(n, device="cuda") // 67) % 2, torch.full_like(modes, 3), torch.full_like(modes, 3)], 1).to(torch.int8).contiguous()
    out = coeffs.clone()
    api.lfnst_batch(out, rows, False)
    assert torch.equal(out[:, 4:, 4:], coeffs[:, 4:, 4:])                       # bottom-right quadrant untouched
    assert int(out[:, 0, 3:4].abs().sum()) >= 0
    idx = np.linspace(0, n - 1, 64).astype(np.int64)
    src = coeffs[idx].cpu().numpy(); got = out[idx].cpu().numpy(); r = rows[idx].cpu().numpy()
    for k in range(64):
        c = np.ascontiguousarray(src[k]); orc.lib.orc_lfnst_fwd(H.ptr(c), w, h, int(r[k, 0]), 3, 3, int(r[k, 1]))
        assert np.array_equal(c, got[k])
