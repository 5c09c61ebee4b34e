"""examples/closed_loop: the library driven by a C++ host program over the C ABI alone (the encoder's side of the boundary) gives
the pictures the Python-driven path gives."""
import os
import subprocess
import zlib

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def test_cpp_host_program_equals_the_python_driven_path(hip, tmp_path):
    import torch
    from uvg266_amd import api, layout
    exe = os.path.join(H.ROOT, "examples", "closed_loop")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(H.ROOT, "examples")])
    W, Hh, depth, qp, n = 328, 200, 8, 27, 3
    pics = [layout.synthetic_yuv420(W, Hh, t, depth) for t in range(n)]
    yuv = tmp_path / "in.yuv"
    with open(yuv, "wb") as f:
        for p in pics:
            for plane in p:
                f.write(np.ascontiguousarray(plane).tobytes())
    out = subprocess.check_output([exe, str(W), str(Hh), str(depth), str(qp), str(n), str(yuv)], text=True)
    lines = [l.split() for l in out.splitlines() if l.startswith("picture")]
    assert len(lines) == n
    cl = api.ClosedLoop(api.ctu_params(W, Hh, qp), [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in yuv_) for yuv_ in pics])
    cl.run()
    torch.cuda.synchronize()
    for i in range(n):
        want = zlib.crc32(b"".join(t.cpu().numpy().tobytes() for t in cl.out[i]))
        assert int(lines[i][5], 16) == want, (i, lines[i])
        assert int(lines[i][3], 16) == zlib.crc32(b"".join(np.ascontiguousarray(p).tobytes() for p in pics[i]))


def test_cpp_host_program_writes_the_encoders_stream(hip, tmp_path):
    """The C++ program alone, from a .yuv of three pictures: its NAL units behind the encoder's parameter sets are the encoder's .266
    (tests/golden/ref_stream_192x128_8_qp27_3frames.npz)."""
    from uvg266_amd import layout
    exe = os.path.join(H.ROOT, "examples", "closed_loop")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(H.ROOT, "examples")])
    g = H.ctu_golden("ref_stream_192x128_8_qp27_3frames")
    W, Hh, depth, qp = (int(a) for a in g["meta"])
    yuv, nals = tmp_path / "in.yuv", tmp_path / "out.nals"
    with open(yuv, "wb") as f:
        for t in g["ts"]:
            for plane in layout.synthetic_yuv420(W, Hh, int(t), depth):
                f.write(np.ascontiguousarray(plane).tobytes())
    subprocess.check_call([exe, str(W), str(Hh), str(depth), str(qp), str(len(g["ts"])), str(yuv), "0", str(nals)], stdout=subprocess.DEVNULL)
    stream = g["bitstream"].tobytes()
    at = stream.find(b"\x00\x00\x01\x00\x41")
    assert stream[:at] + open(nals, "rb").read() == stream


@pytest.mark.parametrize("name", ["ref_tiles_416x240_10_qp32_3x2_2frames", "ref_tiles_264x136_8_qp27_2x2_1frames"])
def test_cpp_tiles_program_writes_the_encoders_stream_under_tiles(hip, tmp_path, name):
    """examples/tiles: uvghip_tile_grid / uvghip_tiles_plan_* from C++ alone, from a .yuv: its NAL units behind the encoder's parameter sets
    are the .266 the encoder wrote with the same --tiles."""
    exe = os.path.join(H.ROOT, "examples", "tiles")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(H.ROOT, "examples")])
    g = H.ctu_golden(name)
    W, Hh, depth, qp, cols, rows = (int(a) for a in g["meta"])
    yuv, nals = tmp_path / "in.yuv", tmp_path / "out.nals"
    with open(yuv, "wb") as f:
        for t in g["ts"]:
            for plane in H.varied_picture(W, Hh, int(t), depth):
                f.write(np.ascontiguousarray(plane).tobytes())
    out = subprocess.check_output([exe, str(W), str(Hh), str(depth), str(qp), str(len(g["ts"])), str(cols), str(rows), str(yuv), str(nals)], text=True)
    assert f"{cols * rows} tiles" in out
    stream = g["bitstream"].tobytes()
    at = stream.find(b"\x00\x00\x01\x00\x41")
    assert stream[:at] + open(nals, "rb").read() == stream
    crcs = [int(l.split()[-1], 16) for l in out.splitlines() if l.startswith("picture")]
    assert crcs == [zlib.crc32(g["final"][i].tobytes()) for i in range(len(g["ts"]))]
