"""Oracle intra path vs vectors dumped from the reference (uvg_intra_build_reference +
uvg_intra_predict + dual SATD/SAD costs).  No upstream unit test exists for intra prediction."""
import numpy as np
import pytest

import helpers as H


@pytest.mark.parametrize("depth", [8, 10])
def test_ref_goldens_blocks(orc, depth):
    k = 0
    for frame, x, y, n, at, al, orig, preds, costs in H.intra_golden_blocks(depth):
        FH, FW = frame.shape
        got_costs, got_preds = orc.intra_mode_costs(depth, frame, FW, FH, x, y, n, at, al, orig, range(67), True)
        assert np.array_equal(got_preds.reshape(67, -1), preds), (x, y, n)
        assert np.array_equal(got_costs, costs)
        k += 1
    assert k >= 25


@pytest.mark.parametrize("depth", [8, 10])
def test_ref_goldens_angular(orc, depth):
    k = 0
    for name, arrs in H.read_golden("intra", depth):
        if name != "angular":
            continue
        (w, h, pm, chroma), ra, rl, want = arrs
        above = np.zeros(400, ra.dtype); above[: len(ra)] = ra
        left = np.zeros(400, rl.dtype); left[: len(rl)] = rl
        assert np.array_equal(orc.angular_pred(depth, w, h, pm, chroma, above, left), want)
        k += 1
    assert k >= 20


def test_flat_references_give_flat_predictions(orc):
    """Every mode of a block whose neighbourhood is constant predicts that constant (H.266 sanity)."""
    rec = np.full((64, 64), 77, np.uint8)
    top, left = orc.intra_build_refs(8, rec, 64, 64, 16, 16, 16, 16, 32, 32)
    ft, fl = orc.intra_filter_refs(8, top, left, 16, 16)
    for mode in range(67):
        assert (orc.intra_predict(8, mode, False, 16, 16, top, left, ft, fl) == 77).all()


def test_picture_corner_uses_mid_grey(orc):
    rec = np.zeros((64, 64), np.uint8)
    top, left = orc.intra_build_refs(8, rec, 64, 64, 0, 0, 8, 8, 0, 0)
    assert (top[:25] == 128).all() and (left[:25] == 128).all()      # intra.c:790,889,1054
