"""The P / B CTU search kernel's LOGIC without a GPU: uvg266_amd/csrc/ctu_pb.h built for the host with one emulated lane (tests/emul/)
against the reference encoder's own records of low-delay encodes (tests/golden/ref_inter_*: decisions, motion, reconstruction, levels,
the three model sets of every CTU, through the real coder's model adaptation and history table)."""
import os
import numpy as np
import pytest
import helpers as H

GOLDENS = ["ref_inter_136x72_10_qp22_4frames", "ref_inter_192x128_8_qp17_5frames", "ref_inter_264x136_8_qp32_9frames",
           # other tools than --preset medium's: P slices (no bi-prediction) without the temporal candidate; no fractional search, no early skip
           "ref_inter_136x72_8_qp27_4frames_p_notmvp", "ref_inter_192x128_10_qp24_4frames_subme0_noskip",
           "ref_inter_136x72_8_qp27_17frames_ra16", "ref_inter_136x72_10_qp22_17frames_ra16", "ref_inter_136x72_8_qp27_9frames_ra8", "ref_inter_136x200_8_qp27_11frames_owf1", "ref_inter_136x72_8_qp27_5frames_rd1", "ref_inter_136x72_8_qp27_33frames_ra16p16"]


# (leaf wave, depth waves, lazy): one wave; two waves; three waves with every evaluation in at once / only when the walk waits for it; four waves
BUILDS = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (1, 1, 1), (1, 2, 0), (1, 2, 1)]


@pytest.mark.parametrize("leafwave,depthwave,lazy", BUILDS)
@pytest.mark.parametrize("name", GOLDENS)
def test_emulated_pb_kernel_equals_the_reference_run(name, leafwave, depthwave, lazy):
    """leafwave 1: the two-wave build's order of work -- the four 4x4 CUs of an 8x8 area are all evaluated, before the cost of the area's unsplit
    CU is known (on the device: beside it, on the second wave); the reference's cuts are re-applied afterwards (ctu_pb.h post_leaves)."""
    g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
    W, Hh, depth, pics, P = H.inter_pictures_from_golden(g)
    n = 0
    for fr, d, prm, F, keep in H.iter_inter_frames(W, Hh, P):
        if int(d["meta"][6]) == 2:
            continue
        r = H.emul_search_inter_picture(depth, prm, F, *pics[fr], leafwave=leafwave, depthwave=depthwave, lazy=lazy)
        assert H.compare_device_inter_picture(W, Hh, d, r) == [], f"frame {fr}"
        n += 1
    assert n >= 3


from sweep_inter_common import CASES, oracle_chain, oracle_as_record  # noqa: E402


@pytest.mark.parametrize("case", range(len(CASES)))
def test_emulated_pb_kernel_equals_the_oracle_on_other_content(case):
    """The sweep of tests/test_gpu_inter_sweep.py with the kernel's source on the host: content, sizes and tools the goldens do not hold
    (large motion that leaves the picture, noise with intra CUs inside B pictures, stills, partial CTUs, P slices, no temporal candidate,
    five merge candidates, no fractional search, no early skip), the oracle's search as the expected result on the same references.
    (Found this way: an intra CU that won against its split left the depth's previous INTER candidate in the history table.)"""
    W, Hh, depth, pics, jobs = oracle_chain(case)
    assert jobs
    for f, fs, prm, F, keep, r in jobs:
        for build in BUILDS:
            got = H.emul_search_inter_picture(depth, prm, F, *pics[f], leafwave=build[0], depthwave=build[1], lazy=build[2])
            assert H.compare_device_inter_picture(W, Hh, oracle_as_record(r), got) == [], (CASES[case], "picture", f, "build", build)
