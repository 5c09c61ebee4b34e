"""HIP picture kernels vs the oracle / reference KATs / reference goldens (bit-exact)."""
import ctypes

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rand_plane(rng, h, w, depth):
    return rng.integers(0, 1 << depth, size=(h, w)).astype(H.px_dtype(depth))


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("dim", H.SAD_DIMS + [(4, 4), (12, 12), (16, 12), (8, 12), (12, 8), (64, 16), (4, 16)])
def test_sad_satd_ssd_batch_vs_oracle(hip, orc, depth, dim):
    from uvg266_amd import api
    bw, bh = dim
    rng = np.random.default_rng(1000 + bw * 131 + bh + depth)
    Hh, W = 200, 264
    cur, ref = rand_plane(rng, Hh, W, depth), rand_plane(rng, Hh, W, depth)
    n = 333
    cxy = np.stack([rng.integers(0, W - bw + 1, n), rng.integers(0, Hh - bh + 1, n)], 1)
    rxy = np.stack([rng.integers(-2 * bw, W + bw, n), rng.integers(-2 * bh, Hh + bh, n)], 1)
    rxy[: n // 2] = np.stack([rng.integers(0, W - bw + 1, n // 2), rng.integers(0, Hh - bh + 1, n // 2)], 1)
    blks = api.make_blocks(cxy, rxy)
    got = api.sad_batch(dev(cur), dev(ref), bw, bh, blks).cpu().numpy()
    want = [orc.image_calc_sad(depth, cur, ref, W, Hh, *c, *r, bw, bh) for c, r in zip(cxy, rxy)]
    assert np.array_equal(got.astype(np.uint32), np.array(want, np.uint32))
    # SATD: interior blocks against the oracle; border blocks against a clamped gather
    got = api.satd_batch(dev(cur), dev(ref), bw, bh, blks).cpu().numpy().astype(np.uint32)
    want = []
    for (cx, cy), (rx, ry) in zip(cxy, rxy):
        ys = np.clip(np.arange(ry, ry + bh), 0, Hh - 1)
        xs = np.clip(np.arange(rx, rx + bw), 0, W - 1)
        blk = np.ascontiguousarray(ref[np.ix_(ys, xs)])
        want.append(orc.satd_any_size(depth, bw, bh, cur, W, blk, bw, a_off=cy * W + cx))
    assert np.array_equal(got, np.array(want, np.uint32))
    # SSD on interior pairs
    m = n // 2
    got = api.ssd_batch(dev(cur), dev(ref), bw, bh, blks[:m]).cpu().numpy().astype(np.uint32)
    want = []
    for (cx, cy), (rx, ry) in zip(cxy[:m], rxy[:m]):
        a = np.ascontiguousarray(cur[cy:cy + bh, cx:cx + bw]); b = np.ascontiguousarray(ref[ry:ry + bh, rx:rx + bw])
        want.append(orc.pixels_calc_ssd(depth, a, b, bw, bw, bw, bh))
    assert np.array_equal(got, np.array(want, np.uint32))


@pytest.mark.parametrize("case", H.SAD_BORDER_KAT)
def test_sad_border_kat(hip, case):
    """tests/sad_tests.c:144-285 through the batched kernel."""
    from uvg266_amd import api
    (rx, ry), expect = case
    got = api.sad_batch(dev(H.SAD_PIC_8x8), dev(H.SAD_REF_8x8), 8, 8, api.make_blocks([[0, 0]], [[rx, ry]]))
    assert int(got[0]) == expect


class Registry:
    """Stand-in for the encoder's strategy_list_t: collects what the registrar registers."""

    def __init__(self, hip_lib):
        self.table = {}
        CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p)

        def cb(opaque, type_, name, prio, fptr):
            assert name == b"hip" and prio == 50
            self.table[type_.decode()] = fptr
            return 1
        self.cb = CB(cb)
        hip_lib.uvghip_set_register_fn(ctypes.cast(self.cb, ctypes.c_void_p))


@pytest.fixture(scope="module")
def strategies(hip):
    reg = Registry(hip)
    assert hip.uvg_strategy_register_picture_hip(None, 8) == 1
    t8 = dict(reg.table)
    reg.table.clear()
    assert hip.uvg_strategy_register_picture_hip(None, 10) == 1
    return {8: t8, 10: dict(reg.table), "keep": reg}


U, VP, I = ctypes.c_uint, ctypes.c_void_p, ctypes.c_int


@pytest.mark.parametrize("test", [0, 1, 2])
@pytest.mark.parametrize("log_w", [2, 3, 4, 5, 6])
def test_satd_kat_through_strategy_pointer(strategies, test, log_w):
    """tests/satd_tests.c:118-170 run against the registered 'hip' strategy pointers."""
    n = 1 << log_w
    f = ctypes.CFUNCTYPE(U, VP, VP)(strategies[8][f"satd_{n}x{n}"])
    b1, b2 = H.satd_kat_buffers(test, log_w)
    r1, r2 = f(H.ptr(b1), H.ptr(b2)), f(H.ptr(b2), H.ptr(b1))
    assert r1 == r2 == H.SATD_KAT[test][log_w - 2]


@pytest.mark.parametrize("dim", H.SAD_DIMS)
def test_reg_sad_through_strategy_pointer(strategies, dim):
    """tests/sad_tests.c:296-343."""
    w, h = dim
    f = ctypes.CFUNCTYPE(U, VP, VP, I, I, U, U)(strategies[8]["reg_sad"])
    a, b = H.sad_big_planes()
    assert f(H.ptr(a), H.ptr(b), w, h, 64, 64) == int(np.abs(a[:h, :w].astype(np.int64) - b[:h, :w]).sum())
    z, m = np.zeros((64, 64), np.uint8), np.full((64, 64), 255, np.uint8)
    assert f(H.ptr(z), H.ptr(m), w, h, 64, 64) == 255 * w * h


@pytest.mark.parametrize("depth", [8, 10])
def test_strategy_pointers_vs_reference_goldens(strategies, depth):
    t = strategies[depth]
    for name, arrs in H.read_golden("picture", depth):
        if name == "strided":
            (w, h, S, qoff), a, b, qbase, outs, res = arrs
            w, h, S = int(w), int(h), int(S)
            assert ctypes.CFUNCTYPE(U, VP, VP, I, I, U, U)(t["reg_sad"])(H.ptr(a), H.ptr(b), w, h, S, S) == outs[0]
            assert ctypes.CFUNCTYPE(U, I, I, VP, I, VP, I)(t["satd_any_size"])(w, h, H.ptr(a), S, H.ptr(b), S) == outs[1]
            assert ctypes.CFUNCTYPE(U, VP, VP, I, I, I, I)(t["pixels_calc_ssd"])(H.ptr(a), H.ptr(b), S, S, w, h) == outs[2]
            got = np.zeros(w * h, np.int16)
            ctypes.CFUNCTYPE(None, VP, VP, VP, I, I, I, I)(t["generate_residual"])(H.ptr(a), H.ptr(b), H.ptr(got), w, h, S, S)
            assert np.array_equal(got, res)
            # satd_any_size_quad incl. the reference's h % 8 == 4 tiling quirk (picture-generic.c:412-479)
            qoff = int(qoff)
            preds = (VP * 4)(*[qbase.ctypes.data + k * qoff * qbase.itemsize for k in range(4)])
            q = np.zeros(4, np.uint32)
            valid = np.ones(4, np.int8)
            ctypes.CFUNCTYPE(None, I, I, VP, I, VP, I, U, VP, VP)(t["satd_any_size_quad"])(w, h, preds, 64, H.ptr(a), S, 4, H.ptr(q), H.ptr(valid))
            assert list(q) == list(outs[3:7]), (w, h)
        elif name == "nxn":
            (n,), pr64, p0, p1, orig, outs = arrs
            n = int(n)
            b2 = np.ascontiguousarray(pr64[64 * 64:])
            assert ctypes.CFUNCTYPE(U, VP, VP)(t[f"sad_{n}x{n}"])(H.ptr(pr64), H.ptr(b2)) == outs[0]
            assert ctypes.CFUNCTYPE(U, VP, VP)(t[f"satd_{n}x{n}"])(H.ptr(pr64), H.ptr(b2)) == outs[1]
            preds = np.concatenate([p0, p1])
            o = np.zeros(2, np.uint32)
            ctypes.CFUNCTYPE(None, VP, VP, U, VP)(t[f"sad_{n}x{n}_dual"])(H.ptr(preds), H.ptr(orig), 2, H.ptr(o))
            assert list(o) == list(outs[2:4])
            ctypes.CFUNCTYPE(None, VP, VP, U, VP)(t[f"satd_{n}x{n}_dual"])(H.ptr(preds), H.ptr(orig), 2, H.ptr(o))
            assert list(o) == list(outs[4:6])


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("cfg", [(16, 16, 8), (8, 8, 4), (32, 32, 16), (64, 64, 8), (16, 8, 7)])
def test_sad_surface_vs_oracle(hip, orc, depth, cfg):
    from uvg266_amd import api
    bw, bh, r = cfg
    rng = np.random.default_rng(7 + bw + r + depth)
    Hh, W = 128, 192
    cur, ref = rand_plane(rng, Hh, W, depth), rand_plane(rng, Hh, W, depth)
    got = api.sad_surface(dev(cur), dev(ref), W, Hh, bw, bh, r).cpu().numpy()
    bx_n = W // bw
    for blk in rng.integers(0, got.shape[0], 12):
        by, bx = (blk // bx_n) * bh, (blk % bx_n) * bw
        for dy, dx in [(-r, -r), (0, 0), (r, r), (-r, r), (1, -2), (r, 0)]:
            want = orc.image_calc_sad(depth, cur, ref, W, Hh, bx, by, bx + dx, by + dy, bw, bh)
            assert got[blk, dy + r, dx + r] == want
    # the corner blocks exercise the replicated border
    for blk in (0, bx_n - 1, got.shape[0] - bx_n, got.shape[0] - 1):
        by, bx = (blk // bx_n) * bh, (blk % bx_n) * bw
        want = np.array([[orc.image_calc_sad(depth, cur, ref, W, Hh, bx, by, bx + dx, by + dy, bw, bh)
                          for dx in range(-r, r + 1)] for dy in range(-r, r + 1)])
        assert np.array_equal(got[blk], want)


def test_full_size_properties(hip):
    """BASELINE config size (1080p): size-independent properties instead of an oracle sweep."""
    import torch
    from uvg266_amd import api
    rng = np.random.default_rng(5)
    Hh, W = 1080, 1920
    cur = rand_plane(rng, Hh, W, 8)
    ref = np.roll(cur, (3, -5), (0, 1))
    dcur, dref = dev(cur), dev(ref)
    ys, xs = np.meshgrid(np.arange(0, Hh - 15, 16), np.arange(0, W - 15, 16), indexing="ij")
    cxy = np.stack([xs.ravel(), ys.ravel()], 1)
    # identity: SAD/SATD/SSD of a plane with itself is zero for every block
    same = api.make_blocks(cxy, cxy)
    assert int(api.sad_batch(dcur, dcur, 16, 16, same).abs().sum()) == 0
    assert int(api.satd_batch(dcur, dcur, 16, 16, same).abs().sum()) == 0
    # a known shift is found exactly by the surface's argmin (true motion recovered)
    surf = api.sad_surface(dcur, dref, W, Hh, 16, 16, 8)
    inner = surf.view(Hh // 16, W // 16, 17, 17)[2:-2, 2:-2]
    assert int(inner[:, :, 8 + 3, 8 - 5].sum()) == 0
    # checksum of checksums: sum of block SADs == whole-plane SAD (numpy) over the tiled area
    blk = api.sad_batch(dcur, dref, 16, 16, same).to(torch.int64).sum().item()
    hh, ww = (Hh // 16) * 16, W
    assert blk == int(np.abs(cur[:hh, :ww].astype(np.int64) - ref[:hh, :ww]).sum())
    # residual: a - b reconstructs a
    res = api.residual_plane(dcur, dref).cpu().numpy()
    assert np.array_equal((res + ref.astype(np.int16)).astype(np.uint8), cur)


@pytest.mark.parametrize("depth", [8, 10])
def test_border_sad_strategy_pointers(strategies, orc, depth):
    """ver_sad / hor_sad / get_optimized_sad (strategies-picture.h:128-134) through the registered pointers,
    against the oracle's restatement of picture-generic.c:1250-1332."""
    t = strategies[depth]
    rng = np.random.default_rng(40 + depth)
    S = 80
    pic, ref = rand_plane(rng, 70, S, depth), rand_plane(rng, 70, S, depth)
    ver = ctypes.CFUNCTYPE(U, VP, VP, I, I, U)(t["ver_sad"])
    hor = ctypes.CFUNCTYPE(U, VP, VP, I, I, U, U, U, U)(t["hor_sad"])
    gos = ctypes.CFUNCTYPE(VP, I)(t["get_optimized_sad"])
    over = orc.fn(depth, "ver_sad", ctypes.c_uint)
    ohor = orc.fn(depth, "hor_sad", ctypes.c_uint)
    oreg = orc.fn(depth, "reg_sad", ctypes.c_uint)
    es = pic.itemsize
    for (w, h) in [(4, 4), (8, 8), (16, 16), (64, 64), (32, 8), (12, 16), (24, 32), (6, 5), (8, 1)]:
        p = pic.ctypes.data + (3 * S + 5) * es
        r = ref.ctypes.data + (2 * S + 7) * es
        assert ver(p, r, w, h, S) == over(ctypes.c_void_p(p), ctypes.c_void_p(r), w, h, S)
        for left, right in [(0, 0), (1, 0), (w // 2, 0), (w - 1, 0), (0, 1), (0, w // 2), (0, w - 1), (2, 3)]:
            if left >= w or right >= w:
                continue
            assert hor(p, r, w, h, S, S, left, right) == ohor(ctypes.c_void_p(p), ctypes.c_void_p(r), w, h, S, S, left, right), (w, h, left, right)
    for w in range(0, 70):
        fp = gos(w)
        if w in (4, 8, 12, 16, 24, 32, 64):
            f = ctypes.CFUNCTYPE(U, VP, VP, I, U, U)(fp)
            for h in (4, 16, 64):
                assert f(H.ptr(pic), H.ptr(ref), h, S, S) == oreg(H.ptr(pic), H.ptr(ref), w, h, S, S)
        else:
            assert not fp          # NULL: the caller falls back to reg_sad (image.c:259-265)


@pytest.mark.parametrize("depth", [8, 10])
def test_dual_64_strategy_pointers(strategies, orc, depth):
    """sad/satd_64x64_dual (picture-generic.c:1087,1475,1481): preds[1] = preds[0] + 32*32 as upstream's pred_buffer."""
    t = strategies[depth]
    rng = np.random.default_rng(64 + depth)
    preds = rand_plane(rng, 1, 1024 + 4096, depth).ravel()
    orig = rand_plane(rng, 1, 4096, depth).ravel()
    for name, want in (("sad_64x64_dual", orc.sad_nxn_dual(depth, preds, orig, 64)),
                       ("satd_64x64_dual", orc.satd_nxn_dual(depth, preds, orig, 64))):
        o = np.zeros(2, np.uint32)
        ctypes.CFUNCTYPE(None, VP, VP, U, VP)(t[name])(H.ptr(preds), H.ptr(orig), 2, H.ptr(o))
        assert list(o) == list(want), name
