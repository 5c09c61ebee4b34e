"""Host side of the closed-loop driver: the wavefront levels respect every reference a block reads."""
import numpy as np
import pytest

from uvg266_amd import layout


@pytest.mark.parametrize("W,Hh,n", [(832, 480, 16), (832, 480, 32), (416, 240, 8), (200, 136, 4), (1920, 1080, 8)])
def test_levels_respect_reference_samples(W, Hh, n):
    blks = layout.intra_availability(layout.block_grid(W - W % n, Hh - Hh % n, n), n, W, Hh)
    lv = layout.dependency_levels(blks, n)
    assert lv.min() == 0 and len(np.unique(lv)) == lv.max() + 1
    grid = -np.ones((Hh // n + 2, W // n + 2), np.int64)
    grid[blks[:, 1] // n, blks[:, 0] // n] = lv
    for (x, y, at, al), l in zip(blks[:: max(1, len(blks) // 3000)], lv[:: max(1, len(blks) // 3000)]):
        refs = [(x - 1, y - 1)] + [(x + k, y - 1) for k in range(0, at, 4)] + [(x - 1, y + k) for k in range(0, al, 4)]
        for rx, ry in refs:
            if rx >= 0 and ry >= 0:
                assert 0 <= grid[ry // n, rx // n] < l, (x, y, rx, ry)
    # the first block has no reference; a block directly right of it needs exactly one more level
    assert lv[0] == 0 and lv[1] == 1
    # coding order is a topological order of the levels' dependencies (what dependency_levels asserts internally), and the
    # schedule is far shorter than the block count
    assert lv.max() + 1 < len(blks) // 4
