"""csrc/shim/frame-hip.c's host logic against the reference's own bitstream code, without a GPU: tools/refcheck/rc_frame_append.c includes the
shim as it is, links /root/reference/src/bitstream.c and holds hip_append (a WPP row's bytes into its leaf state's stream, a chunk at a time)
to what a uvg_bitstream_writebyte per byte leaves -- chunks, length, and the zerocount uvg_bitstream_put_byte had left.  Needs the reference's
sources (skipped where /root/reference does not exist: the GPU box); the parity tests proper are tests/test_gpu_dropin_frame.py."""
import os
import subprocess

import pytest

import helpers as H

REF = os.environ.get("UVG_REF_SRC", "/root/reference")


def test_rows_appended_by_the_shim_leave_the_stream_as_the_encoders_own_writer_does(tmp_path):
    gen = os.path.join(H.ROOT, "oracle", "_ref", "gen")
    if not os.path.isdir(os.path.join(REF, "src")) or not os.path.exists(os.path.join(gen, "version.h")):
        pytest.skip("no reference sources / oracle/_ref/gen here")
    exe = str(tmp_path / "rc_frame_append")
    subprocess.check_call(["gcc", "-O1", "-w", "-DUVG_HAVE_HIP", "-DUVG_DLL_EXPORTS", f"-I{gen}", f"-I{REF}/src", f"-I{REF}/src/extras", f"-I{REF}/src/strategies",
                           f"-I{H.ROOT}/include", os.path.join(H.ROOT, "tools", "refcheck", "rc_frame_append.c"), f"{REF}/src/bitstream.c", "-o", exe, "-lm", "-lpthread"])
    out = subprocess.check_output([exe], text=True)
    assert out.startswith("ok 400"), out


def test_integration_md_shows_the_statements_the_build_applies():
    """INTEGRATION.md section 1 / 10 are what tools/refcheck/patch_ref_hip.py writes into the scratch copy of the encoder: the registration block
    of a strategy group and the three statements of the frame-level hand-over, text for text."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("patch_ref_hip", os.path.join(H.ROOT, "tools", "refcheck", "patch_ref_hip.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    doc = open(os.path.join(H.ROOT, "INTEGRATION.md")).read()
    for rel, _fn, old, new in mod.FRAME:
        added = new.replace(old, "") if new.endswith(old) else new
        for line in added.strip().splitlines():
            assert line.strip() in doc, (rel, line)
        assert rel in doc
