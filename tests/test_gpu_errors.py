"""Error behaviour of the batched C ABI on a GPU box: invalid arguments return a non-zero code and leave a message in
uvghip_last_error (the strategy typedefs have no error channel, the batched ABI does); empty batches are a successful
no-op that touches no output; nothing falls back to a CPU path."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def test_invalid_arguments_are_reported(hip):
    from uvg266_amd import api
    y = torch.zeros((64, 64), dtype=torch.uint8, device="cuda")
    out = torch.full((4,), 77, dtype=torch.int32, device="cuda")
    blks = api.make_blocks([[0, 0]], [[0, 0]])
    bad = [
        ("uvghip_sad_batch", (8, _p(y), 64, _p(y), 64, 64, 64, 0, 16, _p(blks), 1, _p(out), None)),            # width 0
        ("uvghip_satd_batch", (8, _p(y), 64, _p(y), 64, 64, 64, 6, 8, _p(blks), 1, _p(out), None)),            # width % 4
        ("uvghip_transform_batch", (8, 0, 0, 0, 3, 8, 0, 0, _p(y), _p(y), 1, None)),                            # 3-point transform
        ("uvghip_mc_batch", (8, _p(y), 64, 64, 64, 0, 65, 8, _p(blks), 1, 0, _p(out), None)),                   # block wider than 64
        ("uvghip_deblock_frame", (8, _p(y), 64, None, None, 0, 62, 64, _p(blks), 16, 0, 0, 0, 22, None, None)),  # width % 4
        ("uvghip_crc32c_batch", (8, _p(y), 64, 5, _p(blks), 1, _p(out), None)),                                 # only 4x4 / 8x8
    ]
    for name, args in bad:
        fn = getattr(hip, name)
        assert len(args) == len(fn.argtypes), name
        rc = fn(*args)
        assert rc != 0, name
        assert hip.uvghip_last_error(), name
    torch.cuda.synchronize()
    assert torch.all(out == 77)                      # a refused call launched nothing


def test_intra_search_argument_checks(hip):
    from uvg266_amd import api
    y = torch.zeros((64, 64), dtype=torch.uint8, device="cuda")
    blks = api.make_intra_blocks(np.array([[0, 0, 0, 0]], np.int32))
    modes = api.make_modes(list(range(67)))
    best = torch.zeros(1, dtype=torch.int8, device="cuda")
    cost = torch.zeros(1, dtype=torch.int32, device="cuda")
    f = hip.uvghip_intra_search_best_batch
    assert f(8, _p(y), 64, _p(y), 64, 12, _p(blks), 1, _p(modes), 67, _p(best), _p(cost), None, None) != 0      # 12x12 blocks
    assert f(8, _p(y), 64, _p(y), 64, 8, _p(blks), 1, _p(modes), 0, _p(best), _p(cost), None, None) != 0        # no candidates
    assert f(8, _p(y), 64, _p(y), 64, 8, _p(blks), 1, _p(modes), 67, None, _p(cost), None, None) != 0           # no output
    assert f(8, _p(y), 64, _p(y), 64, 8, _p(blks), 1, _p(modes), 67, _p(best), _p(cost), None, None) == 0
    torch.cuda.synchronize()


def test_empty_batches_are_noops(hip):
    from uvg266_amd import api
    y = torch.zeros((64, 64), dtype=torch.uint8, device="cuda")
    out = torch.full((4,), 55, dtype=torch.int32, device="cuda")
    blks = api.make_blocks([[0, 0]], [[0, 0]])
    modes = api.make_modes([0, 1])
    assert hip.uvghip_sad_batch(8, _p(y), 64, _p(y), 64, 64, 64, 16, 16, _p(blks), 0, _p(out), None) == 0
    assert hip.uvghip_satd_batch(8, _p(y), 64, _p(y), 64, 64, 64, 16, 16, _p(blks), 0, _p(out), None) == 0
    assert hip.uvghip_mc_batch(8, _p(y), 64, 64, 64, 0, 8, 8, _p(blks), 0, 0, _p(out), None) == 0
    assert hip.uvghip_intra_search_best_batch(8, _p(y), 64, _p(y), 64, 8, _p(blks), 0, _p(modes), 2, _p(out), _p(out), None, None) == 0
    assert hip.uvghip_alf_stats_batch(8, _p(y), 64, _p(y), 64, 64, 64, 0, _p(blks), 0, None, 0, _p(out), _p(out), _p(out), None) == 0
    torch.cuda.synchronize()
    assert torch.all(out == 55)


def test_unsupported_bit_depth_is_refused(hip):
    """Every entry point that picks the pixel type from `bitdepth` refuses anything but 8 and 10 (12 used to be treated as 10)."""
    from uvg266_amd import api
    y = torch.zeros((64, 64), dtype=torch.uint8, device="cuda")
    out = torch.full((4,), 77, dtype=torch.int32, device="cuda")
    blks = api.make_blocks([[0, 0]], [[0, 0]])
    ib = api.make_intra_blocks([[0, 0, 0, 0]])
    modes = api.make_modes([0])
    calls = [
        ("uvghip_sad_batch", (12, _p(y), 64, _p(y), 64, 64, 64, 8, 8, _p(blks), 1, _p(out), None)),
        ("uvghip_satd_batch", (12, _p(y), 64, _p(y), 64, 64, 64, 8, 8, _p(blks), 1, _p(out), None)),
        ("uvghip_ssd_batch", (9, _p(y), 64, _p(y), 64, 8, 8, _p(blks), 1, _p(out), None)),
        ("uvghip_mc_batch", (12, _p(y), 64, 64, 64, 0, 8, 8, _p(blks), 1, 0, _p(out), None)),
        ("uvghip_intra_pred_batch", (12, _p(y), 64, 0, 8, 8, _p(ib), 1, _p(modes), 1, _p(out), None)),
        ("uvghip_deblock_frame", (12, _p(y), 64, None, None, 0, 64, 64, _p(blks), 16, 0, 0, 0, 22, None, None)),
        ("uvghip_transform_batch", (12, 0, 0, 0, 8, 8, 0, 0, _p(y), _p(y), 1, None)),
        ("uvghip_sao_stats_batch", (12, _p(y), 64, _p(y), 64, _p(blks), 1, _p(out), _p(out), None)),
        ("uvghip_alf_classify_frame", (12, _p(y), 64, 64, 64, 12, _p(out), 16, None)),
    ]
    for name, args in calls:
        fn = getattr(hip, name)
        assert len(args) == len(fn.argtypes), name
        assert fn(*args) != 0, name
    torch.cuda.synchronize()
    assert torch.all(out == 77)


def test_tiny_quant_blocks_are_refused(hip):
    """Blocks of fewer than 8 coefficients make the dequantiser's rounding shift non-positive."""
    c = torch.zeros((4, 2, 2), dtype=torch.int16, device="cuda")
    assert hip.uvghip_quant_batch(8, _p(c), _p(c), 2, 2, 4, 22, 0, 1, None) != 0
    assert hip.uvghip_dequant_batch(8, _p(c), _p(c), 1, 2, 4, 22, 0, None) != 0
