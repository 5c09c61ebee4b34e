"""HIP interpolation (MC, fractional-ME SATD, bi-pred average) vs oracle / reference goldens."""
import numpy as np
import pytest

import helpers as H
from test_gpu_picture import dev, rand_plane

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", [8, 10])
def test_vs_reference_goldens(hip, depth):
    from uvg266_amd import api
    ns = nf = 0
    for name, arrs in H.read_golden("ipol", depth):
        if name == "sample":
            (PW, PH, x0, y0, w, h, fx, fy, chroma, _), plane, px, hi = arrs
            w, h = int(w), int(h)
            d = dev(plane.reshape(PH, PW))
            blk = api.make_mc_blocks([[x0, y0, fx, fy]])
            assert np.array_equal(api.mc_batch(d, blk, w, h, is_chroma=bool(chroma)).cpu().numpy().ravel(), px)
            assert np.array_equal(api.mc_batch(d, blk, w, h, is_chroma=bool(chroma), hi=True).cpu().numpy().ravel(), hi)
            ns += 1
        elif name == "fme":
            (PW, PH, bx, by, w, h), plane, cur, cands, costs = arrs
            got = api.frac_satd_batch(dev(cur.reshape(64, 64)), dev(plane.reshape(PH, PW)), api.make_blocks([[0, 0]], [[bx, by]]),
                                      int(w), int(h), dev(cands.reshape(-1, 2))).cpu().numpy().ravel()
            assert np.array_equal(got.astype(np.uint32), costs)
            nf += 1
    assert ns >= 12 and nf >= 6


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("shape", [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 32), (12, 16), (4, 8)])
def test_mc_batch_vs_oracle(hip, orc, depth, shape):
    from uvg266_amd import api
    w, h = shape
    rng = np.random.default_rng(w + h * 3 + depth)
    PH, PW = 120, 168
    ref = rand_plane(rng, PH, PW, depth)
    n = 29
    for chroma in (False, True):
        if chroma and w > 32:
            continue
        rows = np.stack([rng.integers(-w - 4, PW + 4, n), rng.integers(-h - 4, PH + 4, n),
                         rng.integers(0, 32 if chroma else 16, n), rng.integers(0, 32 if chroma else 16, n)], 1)
        got = api.mc_batch(dev(ref), api.make_mc_blocks(rows), w, h, is_chroma=chroma).cpu().numpy()
        goth = api.mc_batch(dev(ref), api.make_mc_blocks(rows), w, h, is_chroma=chroma, hi=True).cpu().numpy()
        for i, (x0, y0, fx, fy) in enumerate(rows):
            assert np.array_equal(got[i].ravel(), orc.ipol_sample(depth, ref, PW, PH, x0, y0, w, h, fx, fy, chroma, False))
            assert np.array_equal(goth[i].ravel(), orc.ipol_sample(depth, ref, PW, PH, x0, y0, w, h, fx, fy, chroma, True))


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("shape", [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 16), (12, 12), (4, 4), (16, 4)])
def test_frac_satd_batch_vs_oracle(hip, orc, depth, shape):
    from uvg266_amd import api
    w, h = shape
    rng = np.random.default_rng(w * 5 + h + depth)
    PH, PW = 136, 200
    cur = rand_plane(rng, PH, PW, depth)
    ref = np.roll(cur, (1, -2), (0, 1))
    ref = np.clip(ref.astype(np.int32) + rng.integers(-3, 4, ref.shape), 0, (1 << depth) - 1).astype(cur.dtype)
    n = 17
    cxy = np.stack([rng.integers(0, PW - w + 1, n), rng.integers(0, PH - h + 1, n)], 1)
    rxy = cxy + rng.integers(-6, 7, (n, 2))
    rxy[:3] = [[-3, -2], [PW - w + 2, PH - h + 3], [0, PH - h + 1]]
    cands = [(dx, dy) for dy in (-12, -8, -4, 0, 4, 8, 12) for dx in (-12, -8, -4, 0, 4, 8, 12)]   # the whole quarter-sample grid
    got = api.frac_satd_batch(dev(cur), dev(ref), api.make_blocks(cxy, rxy), w, h, dev(np.array(cands, np.int16))).cpu().numpy()
    for i in range(n):
        want = orc.frac_satd(depth, cur, cxy[i, 0], cxy[i, 1], ref, PW, PH, rxy[i, 0], rxy[i, 1], w, h, cands)
        assert np.array_equal(got[i].astype(np.uint32), want), i


@pytest.mark.parametrize("depth", [8, 10])
def test_bipred_average(hip, orc, depth):
    from uvg266_amd import api
    rng = np.random.default_rng(depth)
    w, h = 16, 8
    px0 = rng.integers(0, 1 << depth, w * h).astype(H.px_dtype(depth)); px1 = rng.integers(0, 1 << depth, w * h).astype(H.px_dtype(depth))
    im0 = rng.integers(-2000, 18000, w * h).astype(np.int16); im1 = rng.integers(-2000, 18000, w * h).astype(np.int16)
    for a, b in ((px0, px1), (im0, im1), (px0, im1), (im0, px1)):
        if depth == 10 and (a.dtype == np.uint16 or b.dtype == np.uint16):
            # 10-bit pixels are uint16 on the host; the api distinguishes operands by dtype (int16 = intermediate)
            pass
        got = api.bipred_average_batch(dev(a), dev(b), depth).cpu().numpy()
        assert np.array_equal(got, orc.bipred_average(depth, a, b, w, h))


def test_full_size_shift_property(hip):
    """1080p: a reference that is the current picture shifted by a whole sample is found with zero SATD at
    the matching integer candidate, and the half-sample candidates cost more (size-independent property)."""
    import torch
    from uvg266_amd import api
    rng = np.random.default_rng(3)
    Hh, W = 1080, 1920
    cur = rand_plane(rng, Hh, W, 8)
    ref = np.roll(cur, -1, 1)                     # ref(x) = cur(x+1)  -> best displacement is -1 sample = -16
    xs, ys = np.meshgrid(np.arange(16, W - 32, 16), np.arange(16, Hh - 32, 16))
    cxy = np.stack([xs.ravel(), ys.ravel()], 1)
    cands = np.array([[0, 0], [-8, 0], [8, 0], [-12, 0], [-4, 0]], np.int16)
    c = api.frac_satd_batch(dev(cur), dev(ref), api.make_blocks(cxy, cxy - [1, 0]), 16, 16, dev(cands))
    assert int(c[:, 0].sum()) == 0 and int(c[:, 1:].min()) > 0


def _ipol_registry(hip, depth):
    from test_gpu_picture import Registry
    reg = Registry(hip)
    assert hip.uvg_strategy_register_ipol_hip(None, depth) == 1
    assert set(reg.table) == {"filter_hpel_blocks_hor_ver_luma", "filter_hpel_blocks_diag_luma",
                              "filter_qpel_blocks_hor_ver_luma", "filter_qpel_blocks_diag_luma",
                              "sample_quarterpel_luma", "sample_octpel_chroma",
                              "sample_quarterpel_luma_hi", "sample_octpel_chroma_hi",
                              "get_extended_block", "get_extended_block_wraparound"}      # all ten of strategies-ipol.h:116-139
    return reg


@pytest.mark.parametrize("depth", [8, 10])
def test_registered_sample_functions(hip, orc, depth):
    """sample_quarterpel_luma / sample_octpel_chroma (+ _hi) through the registered 'hip' pointers
    (typedefs strategies-ipol.h:95-114): src points into a caller-padded block, dst is strided."""
    import ctypes
    reg = _ipol_registry(hip, depth)
    VP, I, I16, I8 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int16, ctypes.c_int8
    sig = ctypes.CFUNCTYPE(None, VP, VP, I16, I, I, VP, I16, I8, I8, VP)
    rng = np.random.default_rng(depth)
    PH, PW = 90, 100
    plane = rand_plane(rng, PH, PW, depth)
    es = plane.itemsize
    for name, chroma, hi in (("sample_quarterpel_luma", False, False), ("sample_octpel_chroma", True, False),
                             ("sample_quarterpel_luma_hi", False, True), ("sample_octpel_chroma_hi", True, True)):
        f = sig(reg.table[name])
        for (w, h) in ((8, 8), (16, 4), (4, 16), (64, 64), (32, 16), (2, 2) if chroma else (12, 12)):
            x0, y0 = int(rng.integers(4, PW - w - 5)), int(rng.integers(4, PH - h - 5))
            mv = np.array(rng.integers(-200, 200, 2), np.int32)
            ds = w + 5
            dst = np.full((h, ds), -7 if hi else 3, np.int16 if hi else plane.dtype)
            f(None, plane.ctypes.data + (y0 * PW + x0) * es, PW, w, h, H.ptr(dst), ds, 1, 1, H.ptr(mv))
            mask = 31 if chroma else 15
            want = orc.ipol_sample(depth, plane, PW, PH, x0, y0, w, h, int(mv[0]) & mask, int(mv[1]) & mask, chroma, hi)
            assert np.array_equal(dst[:, :w].ravel(), want), (name, w, h)
            assert np.all(dst[:, w:] == (-7 if hi else 3))          # nothing written beyond the block


@pytest.mark.parametrize("depth", [8, 10])
def test_registered_fme_blocks_vs_reference_costs(hip, orc, depth):
    """filter_{hpel,qpel}_blocks_* through the registered pointers, called in search_frac's order
    (search_inter.c:1142-1166) on the reference-dumped cases: the SATD of each returned candidate block must
    equal the cost the reference got from its own generic functions; the blocks also equal the oracle's samples."""
    import ctypes
    reg = _ipol_registry(hip, depth)
    VP, I, I16, I8 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int16, ctypes.c_int8
    sig = ctypes.CFUNCTYPE(None, VP, VP, I16, I, I, VP, VP, I8, VP, I8, I8)
    steps = [sig(reg.table[n]) for n in ("filter_hpel_blocks_hor_ver_luma", "filter_hpel_blocks_diag_luma",
                                         "filter_qpel_blocks_hor_ver_luma", "filter_qpel_blocks_diag_luma")]
    seen = 0
    for name, arrs in H.read_golden("ipol", depth):
        if name != "fme":
            continue
        (PW, PH, bx, by, w, h), plane, cur, cands, costs = arrs
        PW, PH, bx, by, w, h = (int(v) for v in (PW, PH, bx, by, w, h))
        plane = plane.reshape(PH, PW)
        cands = cands.reshape(16, 2).astype(int)
        # what uvg_get_extended_block hands to search_frac: block + 1-sample ME border + 3/4 filter pad, edges replicated
        ys = np.clip(np.arange(by - 4, by + h + 4), 0, PH - 1)
        xs = np.clip(np.arange(bx - 4, bx + w + 4), 0, PW - 1)
        ext = np.ascontiguousarray(plane[np.ix_(ys, xs)])
        es, S = ext.itemsize, ext.shape[1]
        hx, hy = (cands[8][0] + 4) // 8, cands[8][1] // 8
        filtered = np.zeros((4, 64 * 64), plane.dtype)
        for step, f in enumerate(steps):
            filtered[:] = 0
            f(None, ext.ctypes.data + (3 * S + 3) * es, S, w, h, H.ptr(filtered), None, 4, None,
              0 if step < 2 else hx, 0 if step < 2 else hy)
            for j in range(4):
                mvx, mvy = cands[step * 4 + j]
                blk = filtered[j].reshape(64, 64)
                want = orc.ipol_sample(depth, plane, PW, PH, bx + (mvx >> 4), by + (mvy >> 4), w, h, mvx & 15, mvy & 15, False, False)
                assert np.array_equal(blk[:h, :w].ravel(), want), (step, j, w, h)
                assert orc.satd_any_size(depth, w, h, cur, 64, np.ascontiguousarray(blk), 64) == costs[step * 4 + j]
        seen += 1
    assert seen >= 6


class _Epol(__import__("ctypes").Structure):
    """uvg_epol_args (strategies-ipol.h:67-92)."""
    import ctypes as _c
    _fields_ = [("src", _c.c_void_p), ("src_w", _c.c_int), ("src_h", _c.c_int), ("src_s", _c.c_int),
                ("blk_x", _c.c_int), ("blk_y", _c.c_int), ("blk_w", _c.c_int), ("blk_h", _c.c_int),
                ("pad_l", _c.c_int), ("pad_r", _c.c_int), ("pad_t", _c.c_int), ("pad_b", _c.c_int), ("pad_b_simd", _c.c_int),
                ("buf", _c.c_void_p), ("ext", _c.c_void_p), ("ext_origin", _c.c_void_p), ("ext_s", _c.c_void_p)]


def _ext_cases(rng, PW, PH):
    """Block positions around every border and corner, far outside, and inside; the pads of the reference's callers
    (inter.c:90-100: filter taps; search_inter.c:1085-1100: +1 ME border; image.c:530)."""
    pads = [(3, 4, 3, 4, 0), (4, 4, 4, 4, 0), (1, 2, 1, 2, 5), (0, 0, 0, 0, 0), (3, 4, 3, 4, 3)]
    xs = [-40, -9, -3, 0, 5, PW // 2, PW - 20, PW - 8, PW - 3, PW + 6]
    ys = [-30, -4, 0, 7, PH // 2, PH - 16, PH - 5, PH + 3]
    for k in range(60):
        yield (int(rng.choice(xs)), int(rng.choice(ys)), int(rng.choice([4, 8, 16, 32])), int(rng.choice([4, 8, 16, 64])), pads[k % len(pads)])


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("wrap", [0, 1])
def test_extended_block_batch_and_registered_pointers(hip, orc, depth, wrap):
    """uvg_get_extended_block(_wraparound): the batched copy and the registered per-call pointers vs the oracle's
    restatement of ipol-generic.c:761-883, including the 'inside: pointers into the frame' shortcut."""
    import ctypes
    import torch
    from uvg266_amd import lib
    reg = _ipol_registry(hip, depth)
    f = ctypes.CFUNCTYPE(None, ctypes.c_void_p)(reg.table["get_extended_block_wraparound" if wrap else "get_extended_block"])
    rng = np.random.default_rng(7 + depth + wrap)
    PH, PW, S = 70, 96, 104
    plane = rand_plane(rng, PH, S, depth)
    dplane = dev(plane)
    es = plane.itemsize
    st = torch.cuda.current_stream().cuda_stream
    n_in = n_out = 0
    for bx, by, bw, bh, (pl, pr, pt, pb, pbs) in _ext_cases(rng, PW, PH):
        if wrap and (bx - pl < -PW or bx + bw + pr > 2 * PW):
            continue
        inside, off, s_want, buf_want = orc.get_extended_block(depth, wrap, plane, PW, PH, bx, by, bw, bh, pl, pr, pt, pb, pbs)
        rows, wdt = pt + bh + pb + pbs, pl + bw + pr
        # per-call pointer
        buf = np.full(rows * wdt, 0x5a, plane.dtype)
        ext, org, ext_s = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int()
        a = _Epol(plane.ctypes.data, PW, PH, S, bx, by, bw, bh, pl, pr, pt, pb, pbs, buf.ctypes.data,
                  ctypes.addressof(ext), ctypes.addressof(org), ctypes.addressof(ext_s))
        f(ctypes.addressof(a))
        if inside:
            n_in += 1
            assert ext.value == plane.ctypes.data + off * es and ext_s.value == S == s_want
            assert org.value == plane.ctypes.data + (by * S + bx) * es
            assert np.all(buf == 0x5a)
        else:
            n_out += 1
            assert ext.value == buf.ctypes.data and ext_s.value == wdt == s_want
            assert org.value == buf.ctypes.data + (pt * wdt + pl) * es
            assert np.array_equal(buf, buf_want), (bx, by, bw, bh, pl, pr, pt, pb, pbs)
            # batched entry point: always the copy
            pos = torch.tensor([[bx, by], [bx, by]], dtype=torch.int32, device="cuda")
            out = torch.full((2, rows, wdt), 0x33, dtype=dplane.dtype, device="cuda")
            lib.check(hip.uvghip_extended_block_batch(depth, dplane.data_ptr(), S, PW, PH, wrap, bw, bh, pl, pr, pt, pb, pbs,
                                                      pos.data_ptr(), 2, out.data_ptr(), st), "ext batch")
            assert np.array_equal(out[1].cpu().numpy().ravel(), buf_want)
    assert n_in >= 3 and n_out >= 25
