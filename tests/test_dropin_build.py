"""The drop-in's build, without a GPU: tools/refcheck/build_ref_hip.sh applies INTEGRATION.md's registration blocks to a scratch copy of the
reference and links it against libuvg266hip.so (needs /root/reference: skipped where it is absent).  Here: the patched encoder with no
request behaves like the plain one (same .266, no hip strategy in the selector's table), and a request without a gfx950 device is an error,
not a fall-back.  What it does WITH a device is tests/test_gpu_dropin.py."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = os.environ.get("UVG_REF_SRC", "/root/reference")
REF = os.path.join(ROOT, "oracle", "_ref")
W, H, FRAMES = 192, 128, 3
ARGS = ["--input-res", f"{W}x{H}", "-n", str(FRAMES), "-p", "1", "--preset", "ultrafast", "--no-sao", "--no-deblock", "-q", "27", "--threads", "2"]


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    if not os.path.isdir(os.path.join(REF_SRC, "src")):
        pytest.skip("no reference tree here")
    if not os.path.exists(os.path.join(ROOT, "uvg266_amd", "libuvg266hip.so")):
        pytest.skip("libuvg266hip.so not built")
    subprocess.check_call([os.path.join(ROOT, "tools", "refcheck", "build_ref_hip.sh"), REF_SRC], stdout=subprocess.DEVNULL)
    import sys
    sys.path.insert(0, ROOT)
    from uvg266_amd import layout
    d = tmp_path_factory.mktemp("dropin_build")
    yuv = d / "in.yuv"
    with open(yuv, "wb") as f:
        for t in range(FRAMES):
            for plane in layout.synthetic_yuv420(W, H, t, 8):
                f.write(np.ascontiguousarray(plane).tobytes())
    return d, str(yuv)


def run(binary, yuv, out, env_extra, extra=()):
    env = {k: v for k, v in os.environ.items() if not k.startswith("UVG266_")}
    env.update(env_extra)
    return subprocess.run([binary, "-i", yuv, "-o", out] + ARGS + list(extra), env=env, capture_output=True, text=True, timeout=300)


def test_the_patched_encoder_without_a_request_is_the_plain_encoder(built):
    d, yuv = built
    a = run(os.path.join(REF, "uvg266_8"), yuv, str(d / "plain.266"), {}, ["--no-cpuid"])
    b = run(os.path.join(REF, "uvg266_8_hip"), yuv, str(d / "patched.266"), {})
    assert a.returncode == 0 and b.returncode == 0, (a.stderr[-500:], b.stderr[-500:])
    md5 = lambda p: hashlib.md5(open(p, "rb").read()).hexdigest()
    assert md5(str(d / "plain.266")) == md5(str(d / "patched.266"))
    assert "> hip (" not in b.stderr and "Choosing strategy for sad_8x8" in b.stderr          # the selector's table is printed, hip is not in it


def test_a_request_without_a_device_fails_loudly(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_gpu_dropin.py runs the request")
    d, yuv = built
    r = run(os.path.join(REF, "uvg266_8_hip"), yuv, str(d / "x.266"), {"UVG266_HIP": "1"})
    assert r.returncode != 0
    assert "failed" in r.stderr.lower()
