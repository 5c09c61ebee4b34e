"""Closed-loop intra coding (uvg266_amd.pipeline.ClosedLoopIntra): references come from the reconstruction of the CUs coded
before, so the picture is only right if every level of the wavefront schedule ran after everything it depends on.  The
oracle restates the loop block by block in coding order (search -> predict -> residual/DCT -> RDOQ -> dequant/IDCT ->
reconstruct, then the chroma blocks with the derived mode) on its own reconstruction planes."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def run_gpu(hip, W, Hh, depth, n, rdoq, graph):
    import torch
    from uvg266_amd import api, pipeline
    wl = dict(W=W, H=Hh, depth=depth, alf=False, rdoq=rdoq)
    cl = pipeline.ClosedLoopIntra(hip, wl, 4, n, "cuda", api.make_modes(pipeline.MODES))
    st = torch.cuda.Stream()             # (the null stream cannot be captured)
    torch.cuda.synchronize()
    if graph:
        g = pipeline.Graph(hip, cl.launches, st)
        g.launch(st.cuda_stream)          # twice: a replay must start from the same (cleared) state and end in the same picture
        torch.cuda.synchronize()
        cl.clear()
        g.launch(st.cuda_stream)
        torch.cuda.synchronize()
        g.destroy()
    else:
        pipeline.run(cl.launches, st.cuda_stream)
        torch.cuda.synchronize()
    return cl


def oracle_block(orc, cl, i, rec, rec_u, rec_v):
    """One CU by the oracle with references from the given planes -> (best, cost, luma rec block, levels, [chroma rec blocks])."""
    from uvg266_amd import pipeline
    d, n, W, Hh = cl.depth, cl.n, cl.W, cl.H
    y, u, v = cl.host
    x, yy, at, al = (int(t) for t in cl.blocks[i])
    mx = (1 << d) - 1
    qps = cl.qp + 6 * (d - 8)
    lam = pipeline.intra_lambda(cl.qp)
    ctx = pipeline.synthetic_rdoq_ctx()

    def tu(src_blk, p, c, color):
        res = (src_blk.astype(np.int32) - p.astype(np.int32)).astype(np.int16).ravel()
        coef = orc.tr(d, d, False, 0, 0, c, c, 0, 0, res)
        if cl.rdoq:
            q, _ = orc.rdoq(d, coef, c, c, color, 1, 0, 0, 0, qps, lam * (1.0 if color == 0 else 0.9), ctx)
        else:
            q = orc.quant(d, coef, c, c, d, qps, 0, 1)
        out = p.astype(np.int32)
        if q.any():
            r = orc.tr(d, d, True, 0, 0, c, c, 0, 0, orc.dequant(d, q, c, c, d, qps, 0)).reshape(c, c)
            out = np.clip(r.astype(np.int32) + p.astype(np.int32), 0, mx)
        return q, out

    o = np.ascontiguousarray(y[yy:yy + n, x:x + n])
    costs, preds = orc.intra_mode_costs(d, rec, W, Hh, x, yy, n, at, al, o.ravel(), pipeline.MODES, True)
    j = int(np.argmin(costs))
    p = preds.reshape(len(pipeline.MODES), n, n)[j]
    q, r = tu(o, p, n, 0)
    chroma = []
    if cl.chroma:
        c = n // 2
        cx, cy, cat, cal = x // 2, yy // 2, at // 2, al // 2
        for color, src, rc in ((1, u, rec_u), (2, v, rec_v)):
            top, left = orc.intra_build_refs(d, rc, W // 2, Hh // 2, cx, cy, c, c, cat, cal)
            ftop, fleft = orc.intra_filter_refs(d, top, left, c, c)
            pc = orc.intra_predict(d, pipeline.MODES[j], True, c, c, top, left, ftop, fleft).reshape(c, c)
            chroma.append(tu(src[cy:cy + c, cx:cx + c], pc, c, color))
    return pipeline.MODES[j], int(costs[j]), q, r, chroma


@pytest.mark.parametrize("W,Hh,depth,n,rdoq,graph", [(832, 480, 8, 16, True, True), (832, 480, 8, 32, True, False), (416, 240, 10, 8, False, True),
                                                       (416, 240, 8, 4, True, False)])
def test_whole_picture_equals_the_oracle_loop(hip, orc, W, Hh, depth, n, rdoq, graph):
    cl = run_gpu(hip, W, Hh, depth, n, rdoq, graph)
    px = H.px_dtype(depth)
    rec, rec_u, rec_v = np.zeros((Hh, W), px), np.zeros((Hh // 2, W // 2), px), np.zeros((Hh // 2, W // 2), px)
    best, cost, lev = cl.best.cpu().numpy(), cl.cost.cpu().numpy(), cl.jobs[0]["lev"].cpu().numpy()
    assert cl.n_levels > 20 and (np.diff(cl.level) >= 0).all()
    for i in range(len(cl.blocks)):                        # level order is a valid coding order
        b, c, q, r, chroma = oracle_block(orc, cl, i, rec, rec_u, rec_v)
        x, yy = int(cl.blocks[i, 0]), int(cl.blocks[i, 1])
        assert best[i] == b and cost[i] == c, (i, x, yy)
        assert np.array_equal(lev[i].ravel(), q), (i, x, yy)
        rec[yy:yy + n, x:x + n] = r
        for (qc, rc), plane in zip(chroma, (rec_u, rec_v)):
            plane[yy // 2:yy // 2 + n // 2, x // 2:x // 2 + n // 2] = rc
    assert np.array_equal(cl.rec.cpu().numpy(), rec)
    if cl.chroma:
        assert np.array_equal(cl.rec_u.cpu().numpy(), rec_u) and np.array_equal(cl.rec_v.cpu().numpy(), rec_v)
    # closed loop != open loop: the prediction really came from the reconstruction
    assert not np.array_equal(rec, cl.host[0])


def test_1080p_sampled_blocks_are_consistent_with_their_neighbours(hip, orc):
    """BASELINE configs[1] geometry, 8x8 CUs (32400 blocks, > 1000 wavefront levels): every sampled block must be what the
    oracle computes from the GPU's reconstruction of the blocks coded before it (induction over the coding order)."""
    cl = run_gpu(hip, 1920, 1080, 8, 8, True, True)
    rec, rec_u, rec_v = (t.cpu().numpy() for t in (cl.rec, cl.rec_u, cl.rec_v))
    best, cost, lev = cl.best.cpu().numpy(), cl.cost.cpu().numpy(), cl.jobs[0]["lev"].cpu().numpy()
    rng = np.random.default_rng(3)
    n = cl.n
    for i in np.sort(rng.permutation(len(cl.blocks))[:768]):
        b, c, q, r, chroma = oracle_block(orc, cl, int(i), rec, rec_u, rec_v)
        x, yy = int(cl.blocks[i, 0]), int(cl.blocks[i, 1])
        assert best[i] == b and cost[i] == c and np.array_equal(lev[i].ravel(), q), (i, x, yy)
        assert np.array_equal(rec[yy:yy + n, x:x + n], r), (i, x, yy)
        for (qc, rc), plane in zip(chroma, (rec_u, rec_v)):
            assert np.array_equal(plane[yy // 2:yy // 2 + n // 2, x // 2:x // 2 + n // 2], rc), (i, x, yy)
