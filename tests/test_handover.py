"""The hand-over to the bitstream coder (first slice).  What the search returns per CTU -- the reference's
lcu_coeff_t levels and the cu_info_t fields uvg_encode_coding_tree reads (src/encoderstate.c:863-976) -- is fed to a count-mode
coder (orcN_count_picture_bits: the coding tree's bins in the reference's order through the arithmetic coder's range arithmetic)
and must give, CTU by CTU, the number of bits the real encoder's arithmetic coder consumed for that CTU's coding tree, its range
afterwards and its models afterwards (tests/golden/ref_ctu*.npz: coder, models[:, 2]; tools/refcheck/ctu_dump.c)."""
import numpy as np
import pytest

import helpers as H


@pytest.mark.parametrize("name", ["ref_ctu_832x480_8_qp22", "ref_ctu_416x240_10_qp37", "ref_ctu_320x192_8_qp42", "ref_ctu_192x128_10_qp12", "ref_ctu_256x128_8_qp7", "ref_ctu_264x136_10_qp32"])
def test_golden_handover_counts_the_encoders_bits(orc, name):
    """The consumer itself, on the encoder's own hand-over (the golden's cu fields, trees, levels, start models)."""
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    res = dict(cu=g["cu"], trees=g["trees"], coeff=g["coeff"], models=g["models"])
    bits, rng, after = H.oracle_count_bits(orc, depth, H.search_params(W, Hh, qp), res, g["coder"][:, 1])
    assert np.array_equal(bits, g["coder"][:, 0])
    assert np.array_equal(rng, g["coder"][:, 2])
    assert np.array_equal(after, g["models"][:, 2])
    assert bits.sum() < 8 * len(g["bitstream"])       # (the rest of the stream: parameter sets, slice header, SAO syntax, row ends)


@pytest.mark.parametrize("name", ["ref_ctu_832x480_8_qp22", "ref_ctu_416x240_10_qp37", "ref_ctu_320x192_8_qp42", "ref_ctu_192x128_10_qp12", "ref_ctu_256x128_8_qp7", "ref_ctu_264x136_10_qp32"])
def test_golden_handover_gives_the_encoders_bytes(orc, name):
    """The consumer as a real coder: every CTU's coding tree through the arithmetic coder (low, carries, bypass bins with their
    values) from the coder state the encoder had at that point -> the payload bytes the encoder's coder emitted during that tree
    and its state afterwards.  The trees are 99 % of the stream (56 519 of 56 892 bytes at 832x480)."""
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    res = dict(cu=g["cu"], trees=g["trees"], coeff=g["coeff"], models=g["models"])
    sout, data, off = H.oracle_encode_ctus(orc, depth, H.search_params(W, Hh, qp), res, g["coder_state"][:, 0])
    assert np.array_equal(off, g["tree_off"])
    assert np.array_equal(data, g["tree_bytes"])
    assert np.array_equal(sout, g["coder_state"][:, 1])


@pytest.mark.parametrize("name", ["ref_ctu_832x480_8_qp22", "ref_ctu_416x240_10_qp37", "ref_ctu_320x192_8_qp42", "ref_ctu_192x128_10_qp12", "ref_ctu_256x128_8_qp7", "ref_ctu_264x136_10_qp32"])
def test_golden_handover_gives_the_slice_data_of_the_encoders_stream(orc, name):
    """Everything between the slice header and the end of the slice NAL: the WPP rows' substreams -- SAO syntax and coding tree of
    every CTU through the arithmetic coder, models carried CTU to CTU and row to row, end_of_sub_stream_one_bit, the coder's
    flush, alignment, emulation prevention -- from the hand-over + the SAO decisions.  They are found, byte for byte, inside the
    .266 the encoder wrote (all but 268 of its 56 892 bytes at 832x480)."""
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    res = dict(cu=g["cu"], trees=g["trees"], coeff=g["coeff"])
    data, off, after = H.oracle_encode_rows(orc, depth, H.search_params(W, Hh, qp), res, g["sao"])
    assert np.array_equal(off, g["row_off"])
    assert np.array_equal(data, g["row_bytes"])
    assert np.array_equal(after, g["models"][:, 2])
    stream = g["bitstream"].tobytes()
    at = stream.find(data.tobytes())
    assert at > 0 and len(stream) - at - len(data) < 64          # headers in front, a few bytes (the SEI hash) behind


@pytest.mark.parametrize("name", ["ref_ctucrc_1920x1080_8_qp22"])
def test_oracle_search_handover_counts_the_encoders_bits(orc, name):
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    prm = H.search_params(W, Hh, qp)
    s = H.oracle_search_picture(orc, depth, prm, y, u, v)
    bits, rng, _ = H.oracle_count_bits(orc, depth, prm, s, g["coder"][:, 1])
    assert np.array_equal(bits, g["coder"][:, 0]) and np.array_equal(rng, g["coder"][:, 2])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ref_ctu_832x480_8_qp22", "ref_ctu_192x128_10_qp12", "ref_ctu_256x128_8_qp7", "ref_ctu_264x136_10_qp32", "ref_ctucrc_1920x1080_8_qp22",
                                  "ref_ctucrc_1920x1080_10_qp27", "ref_ctucrc_3840x2160_10_qp22"])
def test_device_handover_counts_the_encoders_bits(hip, orc, name):
    """From the device's outputs: uvghip_scu_t table + levels + start models of uvghip_ctu_plan_run -> the count-mode coder."""
    import torch
    from uvg266_amd import api
    g = H.ctu_golden(name)
    W, Hh, depth, qp, y, u, v = H.golden_source(g)
    prm = H.search_params(W, Hh, qp)
    cs = api.CtuSearch(api.ctu_params(W, Hh, qp, lam=prm.lam), [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v))])
    cs.run()
    torch.cuda.synchronize()
    ry, ru, rv = (t.cpu().numpy() for t in cs.rec[0])
    scu = cs.cu[0].cpu().numpy().reshape(-1).view(H.SCU_NP)
    res = H.search_result_from_device_layout(W, Hh, ry, ru, rv, scu, cs.coeff[0].cpu().numpy(), cs.models[0].cpu().numpy().view(np.uint32))
    bits, rng, _ = H.oracle_count_bits(orc, depth, prm, res, g["coder"][:, 1])
    assert np.array_equal(bits, g["coder"][:, 0]) and np.array_equal(rng, g["coder"][:, 2])
    # ... and the bytes: the arithmetic coder over the device's hand-over emits what the encoder's emitted, CTU by CTU
    import zlib
    sout, data, off = H.oracle_encode_ctus(orc, depth, prm, res, g["coder_state"][:, 0])
    assert np.array_equal(off, g["tree_off"]) and np.array_equal(sout, g["coder_state"][:, 1])
    if "tree_bytes" in g.files:
        assert np.array_equal(data, g["tree_bytes"])
    else:
        assert np.array_equal(np.array([zlib.crc32(data[off[k]:off[k + 1]].tobytes()) for k in range(len(off) - 1)], np.uint32), g["tree_crc"])
    # ... and the whole slice data: the rows' substreams from the device's search outputs + the device's SAO decisions
    cl = api.ClosedLoop(api.ctu_params(W, Hh, qp, lam=prm.lam), [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in (y, u, v))])
    cl.run()
    info, _ = cl.results()
    rows, roff, _ = H.oracle_encode_rows(orc, depth, prm, res, info[0])
    assert np.array_equal(roff, g["row_off"])
    if "row_bytes" in g.files:
        assert np.array_equal(rows, g["row_bytes"])
    else:
        assert np.array_equal(np.array([zlib.crc32(rows[roff[k]:roff[k + 1]].tobytes()) for k in range(len(roff) - 1)], np.uint32), g["row_crc"])
