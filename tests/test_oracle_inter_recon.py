"""Reconstruction of the encoder's INTER decisions (BASELINE configs[2]: --gop lp-g4d3t1 --preset medium) with the oracle's block
functions: for every inter CU of a low-delay encode, motion compensation from the reference pictures as uvg_inter_pred_pu does
it (src/inter.c:400-530, 532-602, 685-748: integer copies, the 8-tap / 4-tap fractional samplers, high-precision
intermediates and uvg_bipred_average for bi-prediction, border replication) + the residual (dequantisation, inverse DCT) must
give the encoder's reconstruction before the in-loop filters.  Records: tests/golden/ref_inter_*.npz (tools/refcheck/ctu_dump.c:
side information with motion per 4x4, reference lists, levels, reconstruction per CTU, the filtered picture per frame).
This is the orchestration the device inter path (next round) has to reproduce; the block kernels it needs are rows a16-a18, a7, a9,
a13 of the path and already on the GPU."""
import pytest

import helpers as H


@pytest.mark.parametrize("name", ["ref_inter_192x128_8_qp17_5frames", "ref_inter_136x72_10_qp22_4frames"])
def test_motion_compensation_plus_residual_gives_the_encoders_reconstruction(orc, name):
    g = H.ctu_golden(name)
    seen = H.inter_reconstruct(g, H.OracleBlocks(orc, int(g["dims"][2])))
    # the encode really exercises what the test is about (461 + 166 inter CUs, 272 bi-predicted, 635 fractional vectors, 524 residual blocks)
    assert seen["inter"] > 100 and seen["bi"] > 20 and seen["frac"] > 50 and seen["resid"] > 50, seen
