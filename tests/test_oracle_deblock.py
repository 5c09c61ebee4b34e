"""Oracle deblocking (two whole-picture passes) vs pictures deblocked by the reference's
uvg_filter_deblock_lcu CTU by CTU (random quad-tree partitions, intra / P / B side information)."""
import numpy as np
import pytest

import helpers as H


def golden_frames(depth):
    for name, arrs in H.read_golden("deblock", depth):
        meta, tab, qmap, iy, iu, iv, oy, ou, ov = arrs
        W, Hh, ts, th, is_b, frame_qp = [int(v) for v in meta[:6]]
        yield (W, Hh, ts, th, is_b, frame_qp, tab, qmap.view(np.int8), iy.reshape(Hh, W), iu.reshape(Hh // 2, W // 2),
               iv.reshape(Hh // 2, W // 2), oy.reshape(Hh, W), ou.reshape(Hh // 2, W // 2), ov.reshape(Hh // 2, W // 2))


@pytest.mark.parametrize("depth", [8, 10])
def test_ref_goldens(orc, depth):
    k = 0
    for W, Hh, ts, th, is_b, fqp, tab, qmap, iy, iu, iv, oy, ou, ov in golden_frames(depth):
        y, u, v = iy.copy(), iu.copy(), iv.copy()
        orc.deblock_frame(depth, y, u, v, W, Hh, tab, ts, 0, 0, is_b, fqp, qmap)
        assert np.array_equal(y, oy) and np.array_equal(u, ou) and np.array_equal(v, ov)
        assert (y != iy).mean() > 0.02            # the filter really did something
        k += 1
    assert k >= 5
