"""The frame-level hand-over, dropped in: the REAL reference encoder (oracle/_ref/uvg266_{8,10}_hip, built by
tools/refcheck/build_ref_hip.sh from /root/reference's sources + INTEGRATION.md section 10's two statements + uvg266_amd/csrc/shim/frame-hip.c)
gives every all-intra frame to the device's closed loop where it would queue its per-CTU jobs (uvg_encode_one_frame,
src/encoderstate.c:2051-2091 -> uvghip_frame_pool_begin) and takes the picture and the WPP rows' substreams back in its bitstream job
(src/encoder_state-bitstream.c:1609 -> uvghip_frame_pool_finish).  Parameter sets, slice header, entry points and the hash SEI are
written by the encoder itself around them: the .266 must be the file its own CPU search writes.

The binaries are test infrastructure built where /root/reference exists (__graft_entry__.build()); they travel with the snapshot.
"""
import hashlib
import os
import subprocess
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def need(path):
    if not os.path.exists(path):
        pytest.skip(f"{os.path.relpath(path, ROOT)} not built (tools/refcheck/build_ref_hip.sh needs /root/reference)")
    return path


def clip(d, name, w, h, frames, depth):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    p = d / name
    with open(p, "wb") as f:
        for t in range(frames):
            for plane in H.varied_picture(w, h, t, depth):
                f.write(np.ascontiguousarray(plane).tobytes())
    return str(p)


def encode(binary, yuv, out, env_extra, args, threads=4, check=True):
    env = dict(os.environ)
    for k in [k for k in env if k.startswith("UVG266_")]:
        del env[k]
    env.update(env_extra)
    t0 = time.time()
    r = subprocess.run([binary, "-i", yuv, "-o", out, "--threads", str(threads)] + list(args), env=env, capture_output=True, text=True, timeout=1500)
    if not check:
        return r
    assert r.returncode == 0, r.stderr[-2000:]
    return hashlib.md5(open(out, "rb").read()).hexdigest(), time.time() - t0


@pytest.mark.parametrize("depth,w,h,frames,qp,owf,group", [(8, 416, 240, 5, 27, 2, None), (8, 264, 136, 3, 22, 0, None), (10, 416, 240, 3, 32, 4, None),
                                                           (8, 264, 136, 23, 27, 7, None), (8, 264, 136, 12, 32, 5, 1), (10, 264, 136, 9, 27, 3, 4)])
def test_all_intra_frames_through_the_closed_loop_write_the_encoders_own_file(tmp_path, depth, w, h, frames, qp, owf, group):
    """--preset medium -p 1 (BASELINE configs[1]'s settings), partial CTUs at the right and lower edges, several frames in flight (--owf + 1
    slots of the frame pool; the frames that are begun before one is asked for share a launch: groups of (owf + 2) / 2 pictures, of one
    picture -- a launch per frame --, of all slots)."""
    yuv = clip(tmp_path, "in.yuv", w, h, frames, depth)
    args = ["--input-res", f"{w}x{h}", "-n", str(frames), "-p", "1", "--preset", "medium", "-q", str(qp), "--owf", str(owf)] + (["--input-bitdepth", "10"] if depth == 10 else [])
    env = {"UVG266_HIP_FRAME": "1"}
    if group is not None:
        env["UVG266_HIP_FRAME_GROUP"] = str(group)
    want, _ = encode(need(os.path.join(REF, f"uvg266_{depth}")), yuv, str(tmp_path / "cpu.266"), {}, args + ["--no-cpuid"])
    got, _ = encode(need(os.path.join(REF, f"uvg266_{depth}_hip")), yuv, str(tmp_path / "hip.266"), env, args)
    assert got == want
    # ... and beside the per-call strategies of the same backend (they are not called for these frames: nothing is left to call them)
    both, _ = encode(os.path.join(REF, f"uvg266_{depth}_hip"), yuv, str(tmp_path / "hip_both.266"), dict(env, UVG266_HIP="1"), args)
    assert both == want


@pytest.mark.parametrize("threads,owf", [(0, None), (0, 2), (16, None)])
def test_the_encoders_own_scheduling_choices_do_not_matter(tmp_path, threads, owf):
    """--threads 0 (no worker threads: the bitstream job runs on the thread that waits for it) and --owf auto (the encoder derives the frames in
    flight from its thread count): begin() and finish() on one thread or two, the same file."""
    w, h, frames = 264, 136, 7
    yuv = clip(tmp_path, "in.yuv", w, h, frames, 8)
    args = ["--input-res", f"{w}x{h}", "-n", str(frames), "-p", "1", "--preset", "medium", "-q", "27"] + ([] if owf is None else ["--owf", str(owf)])
    want, _ = encode(need(os.path.join(REF, "uvg266_8")), yuv, str(tmp_path / "cpu.266"), {}, args + ["--no-cpuid"], threads=threads)
    got, _ = encode(need(os.path.join(REF, "uvg266_8_hip")), yuv, str(tmp_path / "hip.266"), {"UVG266_HIP_FRAME": "1"}, args, threads=threads)
    assert got == want


@pytest.mark.parametrize("depth,w,h,frames,qp,owf,tiles", [(8, 456, 264, 5, 27, 3, ["--tiles", "3x2"]), (10, 416, 240, 3, 32, 2, ["--tiles", "2x2"]),
                                                           (8, 456, 264, 4, 27, 1, ["--tiles-width-split", "64,320", "--tiles-height-split", "192"]),
                                                           (8, 256, 4480, 2, 32, 1, ["--tiles", "4x2"])])      # 280 WPP leaf states (2160p in 8 x 4 tiles has 272)
def test_tiled_frames_through_the_closed_loop_write_the_encoders_own_file(tmp_path, depth, w, h, frames, qp, owf, tiles):
    """--tiles CxR --wpp (and an explicit grid): a group is one tiles plan, the substreams of all tiles go to the tiles' WPP leaf states in
    the order of the bitstream; the PPS with the grid, the slice header with all entry points and the hash SEI are the encoder's."""
    yuv = clip(tmp_path, "in.yuv", w, h, frames, depth)
    args = (["--input-res", f"{w}x{h}", "-n", str(frames), "-p", "1", "--preset", "medium", "-q", str(qp), "--owf", str(owf)] + tiles + ["--wpp"]
            + (["--input-bitdepth", "10"] if depth == 10 else []))
    want, _ = encode(need(os.path.join(REF, f"uvg266_{depth}")), yuv, str(tmp_path / "cpu.266"), {}, args + ["--no-cpuid"])
    got, _ = encode(need(os.path.join(REF, f"uvg266_{depth}_hip")), yuv, str(tmp_path / "hip.266"), {"UVG266_HIP_FRAME": "1"}, args)
    assert got == want


def test_a_configuration_the_closed_loop_does_not_cover_is_refused_loudly(tmp_path):
    """UVG266_HIP_FRAME=1 with P / B pictures: the encoder stops with the reason instead of quietly searching on the CPU."""
    w, h = 264, 136
    yuv = clip(tmp_path, "in.yuv", w, h, 2, 8)
    args = ["--input-res", f"{w}x{h}", "-n", "2", "--preset", "medium", "-q", "27", "--gop", "lp-g4d3t1", "-p", "64"]
    r = encode(need(os.path.join(REF, "uvg266_8_hip")), yuv, str(tmp_path / "x.266"), {"UVG266_HIP_FRAME": "1"}, args, check=False)
    assert r.returncode != 0
    assert "does not cover this configuration" in r.stderr and "intra period" in r.stderr
