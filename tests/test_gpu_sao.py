"""HIP SAO statistics / apply vs oracle and reference goldens (bit-exact)."""
import ctypes

import numpy as np
import pytest

import helpers as H
from test_gpu_picture import Registry, dev, rand_plane

pytestmark = pytest.mark.gpu


def ctu_rects(pw, ph, ctu):
    return [[x, y, min(ctu, pw - x), min(ctu, ph - y)] for y in range(0, ph, ctu) for x in range(0, pw, ctu)]


@pytest.mark.parametrize("depth", [8, 10])
def test_vs_reference_goldens(hip, depth):
    from uvg266_amd import api
    nr = ns = 0
    for name, arrs in H.read_golden("sao", depth):
        if name == "recon":
            (pw, ph, ps, fx, fy, w, h, typ, eo, is_v), plane, bp, offs, want = arrs
            pw, ph, ps = int(pw), int(ph), int(ps)
            rec = dev(plane.reshape(ph + 1, ps))[1:]              # keep the spare row in front, like the reference buffer
            out = dev(np.full((ph, ps), 0x55, plane.dtype))
            o = offs[5:] if is_v else offs[:5]
            params = api.make_sao_params([[typ, eo, bp[1 if is_v else 0], *o]])
            # torch slices keep the parent storage: pass the offset view directly
            api.sao_apply_batch(rec, out, api.make_rects([[fx, fy, w, h]]), params, pw, ph)
            # The reference writes only the samples it filters (the golden keeps the 0x55 fill elsewhere); the kernel
            # also copies rec into the rest of the rectangle: type 0, and for edge classes the picture's outermost
            # row / column on the sides the class looks at (sao.c:321-348).
            fx, fy, w, h = int(fx), int(fy), int(w), int(h)
            got = out.cpu().numpy()
            expect = want.reshape(ph, ps).copy()
            recn = plane.reshape(ph + 1, ps)[1:]
            skip = np.zeros((ph, ps), bool)
            if typ == 0:
                skip[fy:fy + h, fx:fx + w] = True
            elif typ == 2:
                ofs = {0: (-1, 0, 1, 0), 1: (0, -1, 0, 1), 2: (-1, -1, 1, 1), 3: (1, -1, -1, 1)}[int(eo)]
                ax, ay, bx, by = ofs
                if min(ax, bx) < 0 and fx == 0: skip[fy:fy + h, 0] = True
                if max(ax, bx) > 0 and fx + w == pw: skip[fy:fy + h, pw - 1] = True
                if min(ay, by) < 0 and fy == 0: skip[0, fx:fx + w] = True
                if max(ay, by) > 0 and fy + h == ph: skip[ph - 1, fx:fx + w] = True
            assert (expect[skip] == 0x55).all()                 # the reference left exactly these alone
            expect[skip] = recn[skip]
            assert np.array_equal(got, expect)
            nr += 1
        elif name == "stats":
            (PW, PH), po, pr, rects, edge, band = arrs
            e, b = api.sao_stats_batch(dev(po.reshape(PH, PW)), dev(pr.reshape(PH, PW)), api.make_rects(rects.reshape(-1, 4)))
            assert np.array_equal(e.cpu().numpy().ravel(), edge) and np.array_equal(b.cpu().numpy().ravel(), band)
            ns += 1
    assert nr >= 8 and ns == 1


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("chroma", [False, True])
def test_frame_stats_and_apply_vs_oracle(hip, orc, depth, chroma):
    from uvg266_amd import api
    rng = np.random.default_rng(40 + depth + chroma)
    pw, ph, ctu = (200, 136, 64) if not chroma else (100, 68, 32)
    rec = rand_plane(rng, ph, pw, depth)
    rec = (rec // 8 + np.linspace(0, (1 << depth) * 0.8, pw).astype(rec.dtype)[None, :]).astype(rec.dtype)
    orig = np.clip(rec.astype(np.int32) + rng.integers(-5, 6, rec.shape), 0, (1 << depth) - 1).astype(rec.dtype)
    rects = ctu_rects(pw, ph, ctu)
    # the reference hands SAO the CTU minus a few delayed rows/columns: also test cropped rectangles
    rects += [[x, y, max(4, w - 10), max(4, h - 10)] for x, y, w, h in rects[:4]]
    e, b = api.sao_stats_batch(dev(orig), dev(rec), api.make_rects(rects))
    we, wb = orc.sao_stats_rects(depth, orig, rec, rects)
    assert np.array_equal(e.cpu().numpy(), we) and np.array_equal(b.cpu().numpy(), wb)
    # apply: random parameters per CTU
    rects = ctu_rects(pw, ph, ctu)
    rows = [[rng.integers(0, 3), rng.integers(0, 4), rng.integers(0, 29), *rng.integers(-7, 8, 5)] for _ in rects]
    out = dev(rec).clone()
    api.sao_apply_batch(dev(rec), out, api.make_rects(rects), api.make_sao_params(rows))
    want = rec.copy()
    for (x, y, w, h), (typ, eo, bp, *offs) in zip(rects, rows):
        orc.sao_reconstruct_rect(depth, rec, want, pw, ph, x, y, w, h, int(typ), int(eo), [bp, bp], list(offs) + list(offs), False)
    assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("depth", [8, 10])
def test_strategy_pointers(hip, orc, depth):
    reg = Registry(hip)
    assert hip.uvg_strategy_register_sao_hip(None, depth) == 1
    assert set(reg.table) == {"calc_sao_edge_dir", "sao_edge_ddistortion", "sao_band_ddistortion", "sao_reconstruct_color"}
    VP, I = ctypes.c_void_p, ctypes.c_int
    rng = np.random.default_rng(depth)
    bw, bh = 54, 61
    rec = rand_plane(rng, bh, bw, depth) // 4 + (1 << (depth - 2))
    rec = rec.astype(H.px_dtype(depth))
    orig = np.clip(rec.astype(np.int32) + rng.integers(-6, 7, rec.shape), 0, (1 << depth) - 1).astype(rec.dtype)
    f = ctypes.CFUNCTYPE(None, VP, VP, I, I, I, VP)(reg.table["calc_sao_edge_dir"])
    g = ctypes.CFUNCTYPE(I, VP, VP, I, I, I, VP)(reg.table["sao_edge_ddistortion"])
    hb = ctypes.CFUNCTYPE(I, VP, VP, VP, I, I, I, VP)(reg.table["sao_band_ddistortion"])
    for c in range(4):
        got = np.full((2, 5), 3, np.int32)                       # the reference accumulates into the caller's array
        f(H.ptr(orig), H.ptr(rec), c, bw, bh, H.ptr(got))
        want = np.full((2, 5), 3, np.int32)
        orc.fn(depth, "calc_sao_edge_dir", None)(H.ptr(orig), H.ptr(rec), c, bw, bh, H.ptr(want))
        assert np.array_equal(got, want)
        offs = np.array([0, 3, 1, -1, -4], np.int32)
        assert g(H.ptr(orig), H.ptr(rec), bw, bh, c, H.ptr(offs)) == orc.fn(depth, "sao_edge_ddistortion")(H.ptr(orig), H.ptr(rec), bw, bh, c, H.ptr(offs))
    bands = np.array([2, -3, 0, 5], np.int32)
    for bp in (0, 7, 28, 30):
        assert hb(None, H.ptr(orig), H.ptr(rec), bw, bh, bp, H.ptr(bands)) == orc.fn(depth, "sao_band_ddistortion")(H.ptr(orig), H.ptr(rec), bw, bh, bp, H.ptr(bands))
    # sao_reconstruct_color on a bordered buffer (sao_info_t: src/sao.h:55-63, 17 ints)
    rc = ctypes.CFUNCTYPE(None, VP, VP, VP, VP, I, I, I, I, I)(reg.table["sao_reconstruct_color"])
    big = rand_plane(rng, bh + 2, bw + 2, depth)
    for typ in (1, 2):
        for eo in range(4):
            for color in (0, 2):
                info = np.array([typ, eo, 0, 0, 0, 5, 11, 0, 2, 1, -1, -2, 0, -3, 2, 4, -5], np.int32)
                got = np.zeros((bh, bw), big.dtype); want = np.zeros((bh, bw), big.dtype)
                inner = big[1:, 1:]
                rc(None, ctypes.c_void_p(big.ctypes.data + (bw + 2 + 1) * big.itemsize), H.ptr(got), H.ptr(info), bw + 2, bw, bw, bh, color)
                orc.fn(depth, "sao_reconstruct_color", None)(ctypes.c_void_p(big.ctypes.data + (bw + 2 + 1) * big.itemsize), H.ptr(want),
                                                             typ, eo, H.ptr(info[5:7]), H.ptr(info[7:]), bw + 2, bw, bw, bh, int(color == 2))
                assert np.array_equal(got, want), (typ, eo, color)


def test_full_size_properties(hip):
    """1080p: counts add up to the interior area; zero offsets are the identity; stats of rec==orig have zero sums."""
    import torch
    from uvg266_amd import api
    rng = np.random.default_rng(9)
    rec = dev(rand_plane(rng, 1080, 1920, 8))
    rects = ctu_rects(1920, 1080, 64)
    dr = api.make_rects(rects)
    e, b = api.sao_stats_batch(rec, rec, dr)
    assert int(e[:, :, 0].abs().sum()) == 0 and int(b[:, 0].abs().sum()) == 0
    area = torch.tensor([(w - 2) * (h - 2) for _, _, w, h in rects], device="cuda")
    assert torch.equal(e[:, :, 1].sum(-1), area[:, None].expand(-1, 4).to(torch.int32))
    assert torch.equal(b[:, 1].sum(-1), torch.tensor([w * h for _, _, w, h in rects], device="cuda", dtype=torch.int32))
    out = torch.zeros_like(rec)
    params = api.make_sao_params([[1, 0, 3, 0, 0, 0, 0, 0]] * len(rects))
    api.sao_apply_batch(rec, out, dr, params)
    assert torch.equal(out, rec)


def _random_edge_stats(n, seed):
    rng = np.random.default_rng(seed)
    cnt = rng.integers(0, 4096, (n, 4, 1, 5)).astype(np.int32)
    cnt[rng.random(cnt.shape) < 0.1] = 0
    mean = rng.normal(0, 3, (n, 4, 1, 5))
    s = np.rint(cnt * mean).astype(np.int32)
    s[cnt == 0] = 0
    return np.ascontiguousarray(np.concatenate([s, cnt], 2))


@pytest.mark.parametrize("with_rate", [False, True])
def test_edge_offsets_vs_oracle(hip, orc, with_rate):
    """sao_search_edge_sao's offset derivation + class choice (sao.c:380-439), bit-exact with the oracle."""
    from uvg266_amd import api
    n = 777
    edge = _random_edge_stats(n, 5)
    rate = np.random.default_rng(6).integers(0, 200, (n, 4)).astype(np.int32) if with_rate else None
    want_p, want_d = np.zeros((n, 8), np.int32), np.zeros(n, np.int32)
    orc.lib.orc_sao_edge_offsets(H.ptr(edge), H.ptr(rate) if with_rate else None, n, H.ptr(want_p), H.ptr(want_d))
    dd = dev(np.zeros(n, np.int32))
    got = api.sao_edge_offsets_batch(dev(edge), dev(rate) if with_rate else None, None, dd)
    assert np.array_equal(got.cpu().numpy(), want_p)
    assert np.array_equal(dd.cpu().numpy(), want_d)
    # sign constraints and range (sao.c:402-411)
    o = want_p[:, 3:]
    assert (o[:, 0] == 0).all() and (o[:, 1:3] >= 0).all() and (o[:, 3:5] <= 0).all() and (np.abs(o) <= 7).all()
