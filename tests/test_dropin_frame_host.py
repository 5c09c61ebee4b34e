"""The frame-level hand-over's host logic without a GPU (csrc/shim/frame-hip.c inside oracle/_ref/uvg266_8_hip): asked for with
UVG266_HIP_FRAME=1 the encoder never ends in its CPU search -- it stops with the reason for a configuration the closed loop does not cover
and with the library's error when there is no device.  (The parity tests proper: tests/test_gpu_dropin_frame.py.)"""
import os
import subprocess

import numpy as np
import pytest

import helpers as H

EXE = os.path.join(H.ROOT, "oracle", "_ref", "uvg266_8_hip")


def run(tmp_path, extra, env_extra):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/uvg266_8_hip not built (tools/refcheck/build_ref_hip.sh needs /root/reference)")
    w, h = 136, 72
    yuv = tmp_path / "in.yuv"
    with open(yuv, "wb") as f:
        for t in range(2):
            for plane in H.varied_picture(w, h, t, 8):
                f.write(np.ascontiguousarray(plane).tobytes())
    env = {k: v for k, v in os.environ.items() if not k.startswith("UVG266_")}
    env.update(env_extra)
    return subprocess.run([EXE, "-i", str(yuv), "-o", str(tmp_path / "out.266"), "--threads", "2", "--input-res", f"{w}x{h}", "-n", "2", "--preset", "medium", "-q", "27"] + extra,
                          env=env, capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("extra,reason", [(["--gop", "lp-g4d3t1", "-p", "64"], "intra period"), (["-p", "1", "--no-wpp"], "--no-wpp"), (["-p", "1", "--alf", "full"], "ALF"),
                                          (["-p", "1", "--rd", "2"], "rd >= 2"), (["-p", "1", "--bitrate", "500000"], "rate control")])
def test_uncovered_configurations_stop_the_encoder_with_the_reason(tmp_path, extra, reason):
    r = run(tmp_path, extra, {"UVG266_HIP_FRAME": "1"})
    assert r.returncode != 0
    assert "does not cover this configuration" in r.stderr and reason in r.stderr


def test_without_the_request_the_same_binary_encodes_on_the_cpu(tmp_path):
    r = run(tmp_path, ["-p", "1"], {})
    assert r.returncode == 0 and os.path.getsize(tmp_path / "out.266") > 0


def test_no_device_is_an_error_not_a_fallback(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    r = run(tmp_path, ["-p", "1"], {"UVG266_HIP_FRAME": "1"})
    assert r.returncode != 0
    assert "hip frame backend: uvghip_init" in r.stderr
    out = tmp_path / "out.266"
    assert not os.path.exists(out) or os.path.getsize(out) == 0
