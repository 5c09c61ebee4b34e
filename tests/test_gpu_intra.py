"""HIP intra prediction + fused rough search vs oracle / reference goldens (bit-exact)."""
import ctypes

import numpy as np
import pytest

import helpers as H
from test_gpu_picture import Registry, dev, rand_plane

pytestmark = pytest.mark.gpu
ALL_MODES = list(range(67))


@pytest.mark.parametrize("depth", [8, 10])
def test_pred_and_search_vs_reference_goldens(hip, depth):
    from uvg266_amd import api
    k = 0
    for frame, x, y, n, at, al, orig, preds, costs in H.intra_golden_blocks(depth):
        blks = api.make_intra_blocks([[x, y, at, al]])
        got = api.intra_pred_batch(dev(frame), blks, n, n, api.make_modes(ALL_MODES)).cpu().numpy()
        assert np.array_equal(got.reshape(67, -1), preds), (x, y, n)
        # the search kernel reads the original block from a plane at the same (x, y)
        oplane = np.zeros_like(frame)
        oplane[y:y + n, x:x + n] = orig.reshape(n, n)
        c = api.intra_search_batch(dev(frame), dev(oplane), blks, n, api.make_modes(ALL_MODES)).cpu().numpy()
        assert np.array_equal(c.ravel().astype(np.uint32), costs), (x, y, n)
        k += 1
    assert k >= 25


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_search_batch_vs_oracle(hip, orc, depth, n):
    """A picture's worth of blocks incl. picture edges, partial CTUs and z-order availability."""
    from uvg266_amd import api
    rng = np.random.default_rng(n + depth)
    Hh, W = 136, 200                      # not multiples of 64: partial CTUs at the right/bottom
    rec = rand_plane(rng, Hh, W, depth)
    rec = (rec.astype(np.int32) // 4 + np.arange(W)[None, :] // 2 + (1 << (depth - 2))).clip(0, (1 << depth) - 1).astype(rec.dtype)
    orig = rand_plane(rng, Hh, W, depth)
    xs, ys = np.meshgrid(np.arange(0, W - n + 1, n), np.arange(0, Hh - n + 1, n))
    xy = np.stack([xs.ravel(), ys.ravel()], 1)
    sel = np.concatenate([np.arange(min(7, len(xy))), rng.permutation(len(xy))[:50]])
    xy = xy[np.unique(sel)]
    rows = [[x, y, *H.zorder_avail(int(x), int(y), n, W, Hh)] for x, y in xy]
    modes = ALL_MODES if n <= 8 else [0, 1] + list(range(2, 67, 3)) + [18, 34, 50, 66]
    got = api.intra_search_batch(dev(rec), dev(orig), api.make_intra_blocks(rows), n, api.make_modes(modes)).cpu().numpy()
    best, bcost, full = api.intra_search_best_batch(dev(rec), dev(orig), api.make_intra_blocks(rows), n, api.make_modes(modes), True)
    best2, bcost2 = api.intra_search_best_batch(dev(rec), dev(orig), api.make_intra_blocks(rows), n, api.make_modes(modes))
    best, bcost, full = best.cpu().numpy(), bcost.cpu().numpy(), full.cpu().numpy()
    for i, (x, y, at, al) in enumerate(rows):
        o = np.ascontiguousarray(orig[y:y + n, x:x + n]).ravel()
        want = orc.intra_mode_costs(depth, rec, W, Hh, x, y, n, at, al, o, modes)
        assert np.array_equal(got[i].astype(np.uint32), want), (x, y, at, al)
        # fused arg-min: first minimum of the candidate list (strict "<", search_intra.c:1089-1101)
        j = int(np.argmin(want))
        assert best[i] == modes[j] and bcost[i] == want[j], (x, y, best[i], modes[j])
    assert np.array_equal(full, got)
    assert np.array_equal(best2.cpu().numpy(), best) and np.array_equal(bcost2.cpu().numpy(), bcost)


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("shape", [(4, 4), (8, 8), (16, 16), (32, 32), (8, 4), (4, 16), (32, 8), (16, 32), (32, 4), (4, 32)])
@pytest.mark.parametrize("chroma", [False, True])
def test_pred_batch_vs_oracle_incl_wide_angles(hip, orc, depth, shape, chroma):
    from uvg266_amd import api
    w, h = shape
    rng = np.random.default_rng(w * 64 + h + depth)
    Hh, W = 96, 160
    rec = rand_plane(rng, Hh, W, depth)
    rows = [[0, 0, 0, 0], [w, 0, 0, h], [0, h, w, 0], [w, h, 2 * w, 2 * h], [2 * w, h, w, h], [W - w, Hh - h, w, h], [3 * w, 2 * h, w + 4, h + 4]]
    rows = [[x, y, min(at, W - x), min(al, Hh - y)] for x, y, at, al in rows if x + w <= W and y + h <= Hh]
    got = api.intra_pred_batch(dev(rec), api.make_intra_blocks(rows), w, h, api.make_modes(ALL_MODES), chroma).cpu().numpy()
    for i, (x, y, at, al) in enumerate(rows):
        top, left = orc.intra_build_refs(depth, rec, W, Hh, x, y, w, h, at, al)
        ft, fl = orc.intra_filter_refs(depth, top, left, w, h)
        for m in ALL_MODES:
            want = orc.intra_predict(depth, m, chroma, w, h, top, left, ft, fl)
            assert np.array_equal(got[i, m].ravel(), want), (x, y, m)


@pytest.mark.parametrize("depth", [8, 10])
def test_strategy_pointers(hip, orc, depth):
    """angular_pred / intra_pred_planar / pdpc_planar_dc through the registered 'hip' pointers."""
    reg = Registry(hip)
    assert hip.uvg_strategy_register_intra_hip(None, depth) == 1
    assert set(reg.table) == {"angular_pred", "intra_pred_planar", "pdpc_planar_dc", "mip_predict", "intra_pred_filtered_dc"}
    VP, I, I8, U8 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int8, ctypes.c_uint8
    ang = ctypes.CFUNCTYPE(None, VP, I8, I8, VP, VP, VP, U8, U8, I)(reg.table["angular_pred"])
    k = 0
    for name, arrs in H.read_golden("intra", depth):
        if name != "angular":
            continue
        (w, h, pm, chroma), ra, rl, want = arrs
        above = np.zeros(358, ra.dtype); above[: len(ra)] = ra
        left = np.zeros(358, rl.dtype); left[: len(rl)] = rl
        loc = np.array([0, 0], np.int16).tobytes() + bytes([0, 0, w, h, w // 2, h // 2])   # cu_loc_t, src/cu.h:200-209
        locb = ctypes.create_string_buffer(loc, 10)
        got = np.zeros(w * h, ra.dtype)
        ang(ctypes.cast(locb, VP), int(pm), 0, H.ptr(above), H.ptr(left), H.ptr(got), 0, 0, int(w))
        assert np.array_equal(got, want)
        k += 1
    assert k >= 20
    # planar + pdpc vs oracle on one random reference pair
    rng = np.random.default_rng(2)
    dt = H.px_dtype(depth)
    ref = rng.integers(0, 1 << depth, 2 * 358).astype(dt)       # uvg_intra_ref: left[358] then top[358]
    loc = np.array([0, 0], np.int16).tobytes() + bytes([0, 0, 16, 16, 8, 8])
    locb = ctypes.create_string_buffer(loc, 10)
    got = np.zeros(256, dt)
    pl = ctypes.CFUNCTYPE(None, VP, I, VP, VP, VP)(reg.table["intra_pred_planar"])
    top = np.ascontiguousarray(ref[358:]); left = np.ascontiguousarray(ref[:358])
    pl(ctypes.cast(locb, VP), 0, H.ptr(top), H.ptr(left), H.ptr(got))
    t400 = np.zeros(400, dt); t400[:358] = top; l400 = np.zeros(400, dt); l400[:358] = left
    want = np.zeros(256, dt)
    orc.fn(depth, "intra_pred_planar", None)(16, 16, H.ptr(t400), H.ptr(l400), H.ptr(want))
    assert np.array_equal(got, want)
    pd = ctypes.CFUNCTYPE(None, I, VP, I, VP, VP)(reg.table["pdpc_planar_dc"])
    pd(0, ctypes.cast(locb, VP), 0, H.ptr(ref), H.ptr(got))
    orc.fn(depth, "pdpc_planar_dc", None)(16, 16, H.ptr(t400), H.ptr(l400), H.ptr(want))
    assert np.array_equal(got, want)


def test_full_size_properties(hip):
    """1080p: all 8x8 blocks x 67 modes.  Flat picture -> every cost is |orig - c| based and equal across
    modes; a picture equal to its own prediction source has a zero-cost vertical mode where rows repeat."""
    import torch
    from uvg266_amd import api
    Hh, W, n = 1080, 1920, 8
    xs, ys = np.meshgrid(np.arange(0, W, n), np.arange(0, Hh - n + 1, n))
    rows = [[x, y, *H.zorder_avail(int(x), int(y), n, W, Hh)] for x, y in zip(xs.ravel(), ys.ravel())]
    blks = api.make_intra_blocks(rows)
    flat = torch.full((Hh, W), 90, dtype=torch.uint8, device="cuda")
    c = api.intra_search_batch(flat, flat, blks, n, api.make_modes(ALL_MODES))
    b, bc = api.intra_search_best_batch(flat, flat, blks, n, api.make_modes(ALL_MODES))
    assert torch.equal(bc, c.min(1).values) and torch.equal(b.long(), c.argmin(1))      # ties -> first candidate (mode == index here)
    first_row_or_col = torch.tensor([(r[0] == 0 and r[1] == 0) for r in rows], device="cuda")
    assert int(c[~first_row_or_col].abs().sum()) == 0              # interior: flat refs predict the flat block exactly
    # vertical stripes: mode 50 (pure vertical) reproduces every block below the first row exactly
    stripes = (torch.arange(W, device="cuda") * 7 % 251).to(torch.uint8).repeat(Hh, 1).contiguous()
    c = api.intra_search_batch(stripes, stripes, blks, n, api.make_modes([50, 18]))
    not_top = torch.tensor([r[1] > 0 for r in rows], device="cuda")
    assert int(c[not_top, 0].sum()) == 0 and int(c[not_top, 1].sum()) > 0


@pytest.mark.parametrize("depth", [8, 10])
def test_filtered_dc_pointer_vs_reference_golden(hip, depth):
    """intra_pred_filtered_dc (registered upstream, no caller): the 'hip' pointer against vectors dumped from the
    reference's generic function (tools/refcheck/rc_dcfilt.inc)."""
    reg = Registry(hip)
    assert hip.uvg_strategy_register_intra_hip(None, depth) == 1
    f = ctypes.CFUNCTYPE(None, ctypes.c_int8, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint8)(reg.table["intra_pred_filtered_dc"])
    n = 0
    for name, (meta, top, left, want) in H.read_golden("dcfilt", depth):
        log2w, mrl = int(meta[0]), int(meta[1])
        got = np.zeros(want.size, want.dtype)
        f(log2w, H.ptr(top), H.ptr(left), H.ptr(got), mrl)
        assert np.array_equal(got, want), (log2w, mrl)
        n += 1
    assert n >= 24
