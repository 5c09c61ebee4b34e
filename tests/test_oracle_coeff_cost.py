"""The oracle's count-mode coefficient coder (oracle/orc_coeff_cost.c) against the reference-run vectors of
uvg_encode_coeff_nxn on a counting CABAC copy: bits (exact doubles), the adapted models, the constraint flags."""
import numpy as np
import pytest

import helpers as H


@pytest.mark.parametrize("depth", [8, 10])
def test_coeff_cost_goldens(orc, depth):
    g = H.coeffcost_goldens(depth)
    assert len(g) == 160
    seen = set()
    for c in g:
        bits, flags, after = orc.coeff_cost(depth, c["coeff"], c["w"], c["h"], c["color"], c["models"])
        tag = (c["w"], c["h"], c["color"], c["style"])
        assert bits == c["bits"] and flags == c["flags"], tag
        assert np.array_equal(after, c["after"]), tag
        seen.add((c["color"] != 0, c["bits"] > 0, c["style"]))
    assert len(seen) >= 8 and any(s[0] for s in seen)
    assert any(c["bits"] == 0.0 for c in g) and max(c["bits"] for c in g) > 1000


def test_f_entropy_table_is_the_integer_table_scaled(orc):
    t = np.zeros(512, np.float32)
    orc.lib.orc8_f_entropy_table(H.ptr(t))
    assert t[1] == 9.0 and abs(float(t[0]) - 92 / 32768) < 1e-12
    assert np.array_equal(t * 32768, np.round(t * 32768))             # multiples of 2^-15
    assert t[255 * 2] == 9.0 or t[255 * 2 + 1] < 0.01


def test_empty_block_and_dc_only(orc):
    g = H.coeffcost_goldens(8)
    models = g[0]["models"]
    bits, flags, after = orc.coeff_cost(8, np.zeros(64, np.int16), 8, 8, 0, models)
    assert bits == 0.0 and flags == 0 and np.array_equal(after, models)
    c = np.zeros(64, np.int16); c[0] = 1
    bits, flags, _ = orc.coeff_cost(8, c, 8, 8, 0, models)
    assert 0 < bits < 20 and flags == 0
