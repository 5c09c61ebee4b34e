"""Oracle intra_pred_filtered_dc vs vectors dumped from the reference's generic strategy (no upstream unit test)."""
import numpy as np
import pytest

import helpers as H


@pytest.mark.parametrize("depth", [8, 10])
def test_ref_goldens(orc, depth):
    f = orc.fn(depth, "intra_pred_filtered_dc", None)
    sizes = set()
    for name, (meta, top, left, want) in H.read_golden("dcfilt", depth):
        log2w, mrl = int(meta[0]), int(meta[1])
        got = np.zeros(want.size, want.dtype)
        f(log2w, H.ptr(top), H.ptr(left), H.ptr(got), mrl)
        assert np.array_equal(got, want), (log2w, mrl)
        sizes.add((log2w, mrl))
    assert len(sizes) >= 12


def test_flat_references_give_flat_block(orc):
    """All references equal v: dc = v and every smoothed boundary sample is v again."""
    f = orc.fn(8, "intra_pred_filtered_dc", None)
    ref = np.full(80, 93, np.uint8)
    for log2w in (2, 3, 4, 5):
        out = np.zeros(1 << (2 * log2w), np.uint8)
        f(log2w, H.ptr(ref), H.ptr(ref), H.ptr(out), 0)
        assert np.all(out == 93)
