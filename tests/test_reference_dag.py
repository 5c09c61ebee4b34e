"""api.reference_dag: which coded pictures a picture waits for, from the reference encoder's own frame-level records (the reference
buffer of every picture of a --preset medium --gop 16 run and of a --gop lp-g4d3t1 run; tools/refcheck/make_ctu_goldens.py
lowdelay_states).  What api.LowDelayLoop(by_level=True) and bench.py's ra_clip schedule by."""
import os

import numpy as np

import helpers as H


def states(name):
    g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
    return H.frame_states_from_records(g["meta"], g["lam"], g["refs"]), g


def test_random_access_gop16():
    from uvg266_amd import api
    fr, g = states("ref_gop16_states_qp27_65frames")
    deps, level = api.reference_dag(fr)
    display = [int(a) for a in g["display"]]
    assert display[:17] == [0, 16, 8, 4, 2, 1, 3, 6, 5, 7, 12, 10, 9, 11, 14, 13, 15]
    poc_of = [f["poc"] for f in fr]
    for f in range(len(fr)):
        assert all(d < f for d in deps[f])                                   # only pictures coded earlier
        assert sorted(poc_of[d] for d in deps[f]) == sorted(fr[f]["ref_pocs"][:fr[f]["n_refs"]]) or fr[f]["slice_type"] == 2
        assert level[f] == (0 if not deps[f] else 1 + max(level[d] for d in deps[f]))
    # the first GOP: I, then 16 -> 8 -> 4 -> 2 -> {1, 3}; 6 waits for 4 and 8 only, 5 and 7 for 6
    by_poc = {poc_of[f]: level[f] for f in range(17)}
    assert [by_poc[p] for p in (0, 16, 8, 4, 2, 1, 3, 6, 5, 7, 12, 10)] == [0, 1, 2, 3, 4, 5, 5, 4, 5, 5, 3, 4]
    # 65 pictures are 11 levels: the chain through the GOPs' anchor pictures, not the picture count
    assert 1 + max(level) == 11
    widest = max(sum(1 for l in level if l == k) for k in range(1 + max(level)))
    assert widest >= 10


def test_low_delay_is_a_chain():
    from uvg266_amd import api
    fr, _ = states("ref_lowdelay_states_qp27_120frames")
    deps, level = api.reference_dag(fr)
    n_i = sum(1 for f in fr if f["slice_type"] == 2)
    assert n_i == 2
    at = [f for f in range(len(fr)) if fr[f]["slice_type"] == 2]
    for f in range(len(fr)):
        start = max(a for a in at if a <= f)
        assert level[f] == f - start                                           # every picture waits for the one before it
