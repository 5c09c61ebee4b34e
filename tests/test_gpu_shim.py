"""State-taking strategies behind the plain-value views (uvghip_*_percall), replayed from the records the encoder-tree
shim produced inside the reference (tools/refcheck/rc_shim.inc): shim extraction + per-call entry point == the generic
strategy's result.  And the concurrency contract of the drop-in pointers: the encoder calls them from all of its
threadqueue workers at once (src/threadqueue.c:275)."""
import ctypes
import threading

import numpy as np
import pytest

import helpers as H
from test_gpu_picture import Registry

pytestmark = pytest.mark.gpu
U, VP, I, I8 = ctypes.c_uint, ctypes.c_void_p, ctypes.c_int, ctypes.c_int8


def run_quantize_residual(hip, r, depth):
    px = H.px_dtype(depth)
    w, h, so = r["w"], r["h"], r["out_stride"]
    rec = np.full(r["rec"].size, 7 if depth == 8 else 0x0707, px)
    coeff = np.zeros(w * h, np.int16)
    has = hip.uvghip_quantize_residual_percall(ctypes.byref(r["sv"]), ctypes.byref(r["cv"]), w, h, r["color"], r["scan_order"], r["trskip"],
                                               r["in_stride"], so, H.ptr(r["ref"]), H.ptr(r["pred"]), H.ptr(rec), H.ptr(coeff),
                                               r["early_skip"], r["lmcs_adj"], r["tree"])
    return has, coeff, rec


@pytest.mark.parametrize("depth", [8, 10])
def test_quantize_residual_percall_replays_shim_records(hip, depth):
    seen = set()
    for r in H.shim_goldens(depth)["sqr"]:
        has, coeff, rec = run_quantize_residual(hip, r, depth)
        tag = (r["branch"], r["w"], r["h"], r["color"], r["tree"], r["trskip"], r["early_skip"])
        assert has == r["has"], tag
        assert np.array_equal(coeff, r["q"]), tag
        assert np.array_equal(rec, r["rec"]), tag          # the whole output buffer: nothing outside the w x h block is touched
        seen.add((r["branch"], has))
    assert len(seen) >= 10


@pytest.mark.parametrize("depth", [8, 10])
def test_quant_dequant_percall_replay_shim_records(hip, depth):
    for r in H.shim_goldens(depth)["sq"]:
        out = np.full(r["w"] * r["h"], 0x3333, np.int16)
        if r["inverse"]:
            hip.uvghip_dequant_percall(ctypes.byref(r["sv"]), H.ptr(r["src"]), H.ptr(out), r["w"], r["h"], r["color"], r["block_type"], r["ts"])
        else:
            hip.uvghip_quant_percall(ctypes.byref(r["sv"]), H.ptr(r["src"]), H.ptr(out), r["w"], r["h"], r["color"], r["scan_idx"],
                                     r["block_type"], r["ts"], r["lfnst"])
        assert np.array_equal(out, r["want"]), (r["inverse"], r["w"], r["h"], r["color"], r["ts"], r["lfnst"])


@pytest.mark.parametrize("depth", [8, 10])
def test_bipred_percall_replays_shim_records(hip, depth):
    px = H.px_dtype(depth)
    for r in H.shim_goldens(depth)["sbp"]:
        w, h, s = r["w"], r["h"], r["stride"]
        dst = np.full(h * s, 9, px)
        hip.uvghip_bipred_average_percall(depth, H.ptr(dst), s, H.ptr(r["l0"]), r["i0"], H.ptr(r["l1"]), r["i1"], w, h)
        d2 = dst.reshape(h, s)
        assert np.array_equal(d2[:, :w].ravel(), r["want"]), (w, h, r["i0"], r["i1"])
        assert (d2[:, w:] == 9).all()


@pytest.mark.parametrize("depth", [8, 10])
def test_quant_cbcr_percall_replays_shim_records(hip, depth):
    px = H.px_dtype(depth)
    for r in H.shim_goldens(depth)["sjc"]:
        w, h = r["w"], r["h"]
        fill = 7 if depth == 8 else 0x0707
        urec, vrec = np.full(r["urec"].size, fill, px), np.full(r["vrec"].size, fill, px)
        coeff = np.zeros(w * h, np.int16)
        ret = hip.uvghip_quant_cbcr_residual_percall(ctypes.byref(r["sv"]), ctypes.byref(r["cv"]), w, h, r["scan_order"], r["in_stride"], r["out_stride"],
                                                     H.ptr(r["uref"]), H.ptr(r["vref"]), H.ptr(r["upred"]), H.ptr(r["vpred"]), H.ptr(urec), H.ptr(vrec),
                                                     H.ptr(coeff), r["early_skip"], r["lmcs_adj"], r["tree"])
        tag = (w, r["cv"].joint_cb_cr, r["sv"].jccr_sign, r["early_skip"])
        assert ret == r["ret"] and np.array_equal(coeff, r["q"]), tag
        assert np.array_equal(urec, r["urec"]) and np.array_equal(vrec, r["vrec"]), tag


def test_sixteen_threads_call_the_registered_pointers_concurrently(hip, orc):
    """Eight registered strategy pointers plus the three state-taking per-call entry points, hammered from 16 threads at
    once; every single result is checked.  ctypes drops the GIL around foreign calls, so the calls do overlap in the library:
    per-thread streams and staging arenas (percall.h), no shared mutable state."""
    reg = Registry(hip)
    for g in ("picture", "dct", "quant", "intra", "sao"):
        assert getattr(hip, f"uvg_strategy_register_{g}_hip")(None, 8) == 1
    t = dict(reg.table)
    f_satd8 = ctypes.CFUNCTYPE(U, VP, VP)(t["satd_8x8"])
    f_satd32 = ctypes.CFUNCTYPE(U, VP, VP)(t["satd_32x32"])
    f_sad = ctypes.CFUNCTYPE(U, VP, VP, I, I, U, U)(t["reg_sad"])
    f_ssd = ctypes.CFUNCTYPE(U, VP, VP, I, I, I, I)(t["pixels_calc_ssd"])
    f_dct = ctypes.CFUNCTYPE(None, I8, VP, VP)(t["dct_16x16"])
    f_idct = ctypes.CFUNCTYPE(None, I8, VP, VP)(t["idct_8x8"])
    f_abs = ctypes.CFUNCTYPE(ctypes.c_uint32, VP, ctypes.c_size_t)(t["coeff_abs_sum"])
    f_cost = ctypes.CFUNCTYPE(ctypes.c_uint32, VP, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64)(t["fast_coeff_cost"])
    f_any = ctypes.CFUNCTYPE(U, I, I, VP, I, VP, I)(t["satd_any_size"])
    shim = H.shim_goldens(8)
    sqr = shim["sqr"][:96]
    sq = [r for r in shim["sq"] if not r["lfnst"]][:96]
    sbp = shim["sbp"][:96]
    # expectations up front, single-threaded, from the oracle (the records carry the reference's own)
    rng = np.random.default_rng(2024)
    cases = []
    for k in range(48):
        a = rng.integers(0, 256, (64, 64), dtype=np.uint8)
        b = np.clip(a.astype(np.int32) + rng.integers(-20, 21, (64, 64)), 0, 255).astype(np.uint8)
        w, h = int(rng.choice([8, 16, 24, 32, 48, 64])), int(rng.choice([8, 16, 32, 64]))
        c16 = rng.integers(-255, 256, 256).astype(np.int16)
        c8 = rng.integers(-2000, 2001, 64).astype(np.int16)
        a8, b8 = np.ascontiguousarray(a[:8, :8]), np.ascontiguousarray(b[:8, :8])
        a32, b32 = np.ascontiguousarray(a[:32, :32]), np.ascontiguousarray(b[:32, :32])
        cases.append(dict(a=a, b=b, w=w, h=h, c16=c16, c8=c8, a8=a8, b8=b8, a32=a32, b32=b32,
                          satd8=orc.satd_nxn(8, a8, b8, 8), satd32=orc.satd_nxn(8, a32, b32, 32), sad=orc.reg_sad(8, a, b, w, h, 64, 64),
                          ssd=orc.pixels_calc_ssd(8, a, b, 64, 64, w, h), dct=orc.dct_nxn(8, 8, 16, c16), idct=orc.dct_nxn(8, 8, 8, c8, True),
                          abs=orc.coeff_abs_sum(8, c16), cost=orc.fast_coeff_cost(8, c16, 16, 16, 0x0004000300020001),
                          any=orc.satd_any_size(8, w, h, a, 64, b, 64)))
    errors, counts = [], [0] * 16
    start = threading.Barrier(16)

    def worker(tid):
        try:
            start.wait()
            order = np.random.default_rng(tid).permutation(len(cases))
            for rep in range(3):
                for j, k in enumerate(order):
                    c = cases[k]
                    ok = [f_satd8(H.ptr(c["a8"]), H.ptr(c["b8"])) == c["satd8"],
                          f_satd32(H.ptr(c["a32"]), H.ptr(c["b32"])) == c["satd32"],
                          f_sad(H.ptr(c["a"]), H.ptr(c["b"]), c["w"], c["h"], 64, 64) == c["sad"],
                          f_ssd(H.ptr(c["a"]), H.ptr(c["b"]), 64, 64, c["w"], c["h"]) == c["ssd"],
                          f_abs(H.ptr(c["c16"]), 256) == c["abs"],
                          f_cost(H.ptr(c["c16"]), 16, 16, 0x0004000300020001) == c["cost"],
                          f_any(c["w"], c["h"], H.ptr(c["a"]), 64, H.ptr(c["b"]), 64) == c["any"]]
                    o16, o8 = np.zeros(256, np.int16), np.zeros(64, np.int16)
                    f_dct(8, H.ptr(c["c16"]), H.ptr(o16)); f_idct(8, H.ptr(c["c8"]), H.ptr(o8))
                    ok += [np.array_equal(o16, c["dct"]), np.array_equal(o8, c["idct"])]
                    r = sqr[(tid * 7 + j + rep * 31) % len(sqr)]
                    has, coeff, rec = run_quantize_residual(hip, r, 8)
                    ok += [has == r["has"] and np.array_equal(coeff, r["q"]) and np.array_equal(rec, r["rec"])]
                    r = sq[(tid * 5 + j + rep * 17) % len(sq)]
                    out = np.zeros(r["w"] * r["h"], np.int16)
                    if r["inverse"]:
                        hip.uvghip_dequant_percall(ctypes.byref(r["sv"]), H.ptr(r["src"]), H.ptr(out), r["w"], r["h"], r["color"], r["block_type"], r["ts"])
                    else:
                        hip.uvghip_quant_percall(ctypes.byref(r["sv"]), H.ptr(r["src"]), H.ptr(out), r["w"], r["h"], r["color"], 0, r["block_type"], r["ts"], 0)
                    ok += [np.array_equal(out, r["want"])]
                    r = sbp[(tid * 3 + j + rep * 13) % len(sbp)]
                    dst = np.zeros(r["h"] * r["stride"], np.uint8)
                    hip.uvghip_bipred_average_percall(8, H.ptr(dst), r["stride"], H.ptr(r["l0"]), r["i0"], H.ptr(r["l1"]), r["i1"], r["w"], r["h"])
                    ok += [np.array_equal(dst.reshape(r["h"], r["stride"])[:, : r["w"]].ravel(), r["want"])]
                    if not all(ok):
                        errors.append((tid, rep, int(k), ok))
                    counts[tid] += len(ok)
        except Exception as e:      # noqa: BLE001 -- surfaced below
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(16)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]
    assert all(c == 3 * 48 * 12 for c in counts)
