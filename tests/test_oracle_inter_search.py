"""The oracle's inter search (oracle/orc_search.c + orc_search_inter.inc: uvg_search_cu_inter with merge analysis, early skip, hexagon
search, fractional search and bi-prediction competing with the intra search inside search_cu, the coder's model adaptation and history
table between CTUs) against records of low-delay encodes of the real encoder (BASELINE configs[2]: --gop lp-g4d3t1 --preset medium;
tests/golden/ref_inter_*.npz): every call of uvg_search_cu_inter with the motion it decided and both costs (doubles, bit for bit), and
per CTU the side information, motion, split trees, levels, reconstruction before the in-loop filters and all three sets of context
models (257 + the 18 of the inter syntax) of every picture.  tools/refcheck/sweep_inter.py repeats this on fresh encodes of the
reference (sizes 64..320, both depths, QP 10..44, 2..10 pictures, four kinds of content)."""
import numpy as np
import pytest

import helpers as H


@pytest.mark.parametrize("name", ["ref_inter_192x128_8_qp17_5frames", "ref_inter_136x72_10_qp22_4frames", "ref_inter_264x136_8_qp32_9frames",
                                  "ref_inter_136x72_8_qp27_4frames_p_notmvp", "ref_inter_192x128_10_qp24_4frames_subme0_noskip",
           "ref_inter_136x72_8_qp27_17frames_ra16", "ref_inter_136x72_10_qp22_17frames_ra16", "ref_inter_136x72_8_qp27_9frames_ra8", "ref_inter_136x72_8_qp27_5frames_rd1", "ref_inter_136x72_8_qp27_33frames_ra16p16"])
def test_every_picture_of_a_low_delay_encode(name):
    orc = H.load_oracle()
    g = H.ctu_golden(name)
    W, Hh, depth, pics, P = H.inter_pictures_from_golden(g)
    seen = dict(pictures=0, calls=0, skip=0, merge=0, amvp=0, bi=0, intra_in_b=0)
    for fr, d, r, buf, ntr in H.run_inter_oracle(orc, W, Hh, depth, pics, P):
        msgs = H.compare_inter_picture(W, Hh, d, r, buf, ntr)
        assert not msgs, (name, fr, msgs)
        assert ntr == len(d["cuinter"])
        seen["pictures"] += 1
        seen["calls"] += ntr
        m, c = d["motion"][:Hh // 4, :W // 4], d["cu"][:Hh // 4, :W // 4]
        inter = c[:, :, 0] == 2
        seen["skip"] += int(((m[:, :, 7] & 1) != 0)[inter].sum())
        seen["merge"] += int(((m[:, :, 7] & 2) != 0)[inter].sum())
        seen["amvp"] += int(((m[:, :, 7] & 3) == 0)[inter].sum())
        seen["bi"] += int((m[:, :, 6] == 3)[inter].sum())
        if int(d["meta"][6]) != 2:
            seen["intra_in_b"] += int((c[:, :, 0] == 1).sum())
    # what the golden exercises (counts of 4x4 units): every kind of decision in the two 8-bit encodes
    print(name, seen)
    variant = "cfg" in g.files          # a golden with other tools than --preset medium's: P slices only / no early skip
    assert seen["calls"] > 100 and seen["amvp"] and (seen["bi"] or variant), seen
    if depth == 8 and not variant:
        assert seen["merge"] and seen["skip"], seen


def test_frames_in_flight_restrict_the_vectors():
    """cfg.owf != 0: pictures are coded while their reference pictures are still in the making, and no candidate vector may reach beyond what
    is final there (fracmv_within_tile, search_inter.c:94-149: one CTU row below, two CTUs down-right, a margin for the interpolation and the
    in-loop filters).  tests/golden/ref_inter_136x200_8_qp27_11frames_owf1 is an `--owf 1` run on content that rises ever faster
    (helpers.rising_picture), where that changes the stream; the oracle with the restriction reproduces every call and every CTU, and
    without it does not."""
    name = "ref_inter_136x200_8_qp27_11frames_owf1"
    orc = H.load_oracle()
    g = H.ctu_golden(name)
    assert int(g["cfg"][7]) == 1
    W, Hh, depth, pics, P = H.inter_pictures_from_golden(g)
    calls = 0
    for fr, d, r, buf, ntr in H.run_inter_oracle(orc, W, Hh, depth, pics, P):
        msgs = H.compare_inter_picture(W, Hh, d, r, buf, ntr)
        assert not msgs, (name, fr, msgs)
        calls += ntr
    assert calls > 500
    # the same records against the unrestricted search (the --owf 0 of every other golden): somewhere a decision differs
    for d in P.values():
        d["cfg"] = np.concatenate([np.asarray(d["cfg"][:7]), [0]])
    differs = False
    for fr, d, r, buf, ntr in H.run_inter_oracle(orc, W, Hh, depth, pics, P):
        differs = differs or bool(H.compare_inter_picture(W, Hh, d, r, buf, ntr))
    assert differs, "the golden does not exercise the restriction"
