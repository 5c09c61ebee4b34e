"""Oracle crc32c_4x4 / crc32c_8x8 (IBC hash) and pixel_var (VAQ) vs vectors dumped from the reference's generic strategy
(no upstream unit test), plus an independent CRC-32C check value."""
import ctypes

import numpy as np
import pytest

import helpers as H


@pytest.mark.parametrize("depth", [8, 10])
def test_ref_goldens(orc, depth):
    n = 0
    for name, (meta, a, crc, v, var) in H.read_golden("hashvar", depth):
        S, x, y, ln = (int(t) for t in meta)
        blk = a.reshape(S, S)
        p = np.ascontiguousarray(blk[y:, x:])               # pointer to (x, y) with stride S: pass the offset view's base
        base = blk.ravel()[y * S + x:]
        f = orc.fn(depth, "crc32c_nxn", ctypes.c_uint32)
        assert f(H.ptr(base), S, 4) == int(crc[0]) and f(H.ptr(base), S, 8) == int(crc[1])
        g = orc.fn(depth, "pixel_var", ctypes.c_double)
        assert g(H.ptr(v), ln) == float(var[0])
        n += 1
    assert n >= 16


def test_crc32c_check_value(orc):
    """The standard CRC-32C check: "123456789" -> 0xE3069283.  A 4x4 block hashes 16 bytes, so use the known value of
    sixteen zero bytes instead (0x8A9136AA... computed independently below with the bitwise definition)."""
    def crc32c(data):
        c = 0xFFFFFFFF
        for b in data:
            c ^= b
            for _ in range(8):
                c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
        return c ^ 0xFFFFFFFF
    assert crc32c(b"123456789") == 0xE3069283
    blk = (np.arange(64, dtype=np.uint8) * 37 + 11).astype(np.uint8)
    f = orc.fn(8, "crc32c_nxn", ctypes.c_uint32)
    assert f(H.ptr(blk), 8, 8) == crc32c(blk.tobytes())
    assert f(H.ptr(blk), 8, 4) == crc32c(b"".join(blk[8 * r:8 * r + 4].tobytes() for r in range(4)))
