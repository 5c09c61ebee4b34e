"""The drop-in, dropped in: the REAL reference encoder (oracle/_ref/uvg266_8_hip = /root/reference's sources + INTEGRATION.md section 1's
registration blocks + the section-2 shim, linked against uvg266_amd/libuvg266hip.so by tools/refcheck/build_ref_hip.sh) runs BASELINE.json
configs[0] -- 832x480 8-bit, 10 frames, -p 1 --preset ultrafast --no-sao --no-deblock -- with the "hip" strategies selected by its own
strategy selector, and writes the same .266 as its generic-C strategies (--no-cpuid).  Whole backend at once, then group by group
(UVG266_HIP=<group>), then the selector's own override variable on top (strategyselector.c:293-314).

The binaries are test infrastructure built where /root/reference exists (__graft_entry__.build()); they travel with the snapshot.
"""
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
W, H, FRAMES = 832, 480, 10
ARGS = ["--input-res", f"{W}x{H}", "-n", str(FRAMES), "-p", "1", "--preset", "ultrafast", "--no-sao", "--no-deblock", "-q", "27"]
GROUPS = ("picture", "dct", "intra", "sao", "quant", "ipol", "alf")
# the strategy types each group of the backend registers
TYPES = {
    "picture": ["reg_sad", "sad_8x8", "sad_16x16", "satd_4x4", "satd_8x8", "satd_16x16", "satd_32x32", "satd_any_size", "pixels_calc_ssd", "generate_residual",
                "bipred_average", "pixel_var"],
    "dct": ["dct_4x4", "dct_8x8", "dct_16x16", "dct_32x32", "idct_4x4", "idct_8x8", "idct_16x16", "idct_32x32", "fast_forward_dst_4x4"],
    "intra": ["angular_pred", "intra_pred_planar", "pdpc_planar_dc", "intra_pred_filtered_dc"],
    "sao": ["sao_edge_ddistortion", "calc_sao_edge_dir", "sao_reconstruct_color", "sao_band_ddistortion"],
    "quant": ["quant", "dequant", "quantize_residual", "coeff_abs_sum", "fast_coeff_cost"],
    "ipol": ["filter_hpel_blocks_hor_ver_luma", "sample_quarterpel_luma", "sample_octpel_chroma", "get_extended_block"],
    "alf": ["alf_derive_classification_blk", "alf_filter_5x5_blk", "alf_filter_7x7_blk", "alf_get_blk_stats"],
}


def need(path):
    if not os.path.exists(path):
        pytest.skip(f"{os.path.relpath(path, ROOT)} not built (tools/refcheck/build_ref_hip.sh needs /root/reference)")
    return path


@pytest.fixture(scope="module")
def clip(tmp_path_factory):
    import sys
    sys.path.insert(0, ROOT)
    from uvg266_amd import layout
    d = tmp_path_factory.mktemp("dropin")
    p = d / "c0.yuv"
    with open(p, "wb") as f:
        for t in range(FRAMES):
            for plane in layout.synthetic_yuv420(W, H, t, 8):
                f.write(np.ascontiguousarray(plane).tobytes())
    return d, str(p)


def encode(binary, yuv, out, env_extra, extra=(), threads=4, args=None):
    env = dict(os.environ)
    for k in [k for k in env if k.startswith("UVG266_")]:
        del env[k]
    env.update(env_extra)
    r = subprocess.run([binary, "-i", yuv, "-o", out, "--threads", str(threads)] + (ARGS if args is None else list(args)) + list(extra), env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    return hashlib.md5(open(out, "rb").read()).hexdigest(), r.stderr


def chosen(stderr):
    """{type: name of the strategy the selector picked} from DEBUG_STRATEGYSELECTOR's table (strategyselector.c:316-327)."""
    out, cur = {}, None
    for line in stderr.splitlines():
        m = re.match(r"Choosing strategy for (\w+):", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"> (\w+) \(", line)
        if m and cur:
            out[cur] = m.group(1)
    return out


@pytest.fixture(scope="module")
def generic_md5(clip):
    d, yuv = clip
    md5, _ = encode(need(os.path.join(REF, "uvg266_8")), yuv, str(d / "generic.266"), {}, ["--no-cpuid"])
    return md5


def test_configs0_with_every_hip_strategy_selected_writes_the_generic_bitstream(clip, generic_md5):
    d, yuv = clip
    md5, err = encode(need(os.path.join(REF, "uvg266_8_hip")), yuv, str(d / "hip_all.266"), {"UVG266_HIP": "1"})
    sel = chosen(err)
    for g in GROUPS:
        for t in TYPES[g]:
            assert sel.get(t) == "hip", (t, sel.get(t))
    assert md5 == generic_md5


@pytest.mark.parametrize("group", GROUPS)
def test_configs0_group_by_group(clip, generic_md5, group):
    d, yuv = clip
    md5, err = encode(need(os.path.join(REF, "uvg266_8_hip")), yuv, str(d / f"hip_{group}.266"), {"UVG266_HIP": group})
    sel = chosen(err)
    for g in GROUPS:
        for t in TYPES[g]:
            assert (sel.get(t) == "hip") == (g == group), (t, sel.get(t))
    assert md5 == generic_md5


def test_the_selectors_override_variable_still_works(clip, generic_md5):
    """UVG266_OVERRIDE_<type>=<name> (strategyselector.c:293-314) picks by strategy NAME: "hip" is a name like any other."""
    d, yuv = clip
    md5, err = encode(need(os.path.join(REF, "uvg266_8_hip")), yuv, str(d / "hip_ovr.266"),
                      {"UVG266_HIP": "dct,picture", "UVG266_OVERRIDE_dct_8x8": "generic", "UVG266_OVERRIDE_satd_8x8": "hip"})
    assert "UVG266_OVERRIDE_dct_8x8 environment variable present, choosing dct_8x8:generic" in err
    assert "UVG266_OVERRIDE_satd_8x8 environment variable present, choosing satd_8x8:hip" in err
    assert md5 == generic_md5


def test_without_a_request_the_backend_stays_out(clip, generic_md5):
    d, yuv = clip
    md5, err = encode(need(os.path.join(REF, "uvg266_8_hip")), yuv, str(d / "hip_off.266"), {})
    assert "hip" not in set(chosen(err).values())
    assert md5 == generic_md5


def _small_clip(d, name, w, h, frames, depth):
    import sys
    sys.path.insert(0, ROOT)
    from uvg266_amd import layout
    p = d / name
    with open(p, "wb") as f:
        for t in range(frames):
            for plane in layout.synthetic_yuv420(w, h, t, depth):
                f.write(np.ascontiguousarray(plane).tobytes())
    return str(p)


def test_the_10_bit_build_of_the_drop_in(clip):
    """oracle/_ref/uvg266_10_hip (-DUVG_BIT_DEPTH=10: uvg_pixel is 16 bits wide, every registrar is called with bitdepth 10) on a
    configs[0]-shaped 10-bit clip: the same .266 as the 10-bit generic-C build."""
    d, _ = clip
    w, h, frames = 416, 240, 4
    yuv = _small_clip(d, "c0_10.yuv", w, h, frames, 10)
    args = ["--input-res", f"{w}x{h}", "--input-bitdepth", "10", "-n", str(frames), "-p", "1", "--preset", "ultrafast", "--no-sao", "--no-deblock", "-q", "27"]
    want, _ = encode(need(os.path.join(REF, "uvg266_10")), yuv, str(d / "generic10.266"), {}, ["--no-cpuid"], args=args)
    got, err = encode(need(os.path.join(REF, "uvg266_10_hip")), yuv, str(d / "hip10.266"), {"UVG266_HIP": "1"}, args=args)
    sel = chosen(err)
    for g in GROUPS:
        for t in TYPES[g]:
            assert sel.get(t) == "hip", (t, sel.get(t))
    assert got == want


def test_preset_medium_through_the_per_call_strategies(clip):
    """--preset medium -p 1 (RDOQ on, SAO and deblocking on, the full intra search: BASELINE configs[1]'s settings) on a short small clip
    with every hip strategy selected: uvg_quantize_residual's RDOQ branch goes through quantize_residual_hip, the SAO group through its
    four pointers -- the .266 of the generic-C run."""
    d, _ = clip
    w, h, frames = 192, 128, 2
    yuv = _small_clip(d, "medium.yuv", w, h, frames, 8)
    args = ["--input-res", f"{w}x{h}", "-n", str(frames), "-p", "1", "--preset", "medium", "-q", "27"]
    want, _ = encode(need(os.path.join(REF, "uvg266_8")), yuv, str(d / "generic_medium.266"), {}, ["--no-cpuid"], args=args)
    got, err = encode(need(os.path.join(REF, "uvg266_8_hip")), yuv, str(d / "hip_medium.266"), {"UVG266_HIP": "1"}, args=args)
    assert chosen(err).get("quantize_residual") == "hip" and chosen(err).get("sao_edge_ddistortion") == "hip"
    assert got == want


ALF_TYPES = TYPES["alf"]


@pytest.mark.parametrize("depth", [8, 10])
def test_the_alf_group_in_an_alf_full_run(clip, depth):
    """uvg_strategy_register_state_hip_alf (the shim) + csrc/alf_percall.hip: classification, the 7x7 / 5x5 filters and the covariance statistics
    of a -p 1 --alf full run go through the "hip" strategies -- the .266 of the generic-C run (the derivation in between works on the
    device's statistics: any difference in a covariance would change the filters it derives)."""
    d, _ = clip
    w, h, frames = 192, 128, 3
    yuv = _small_clip(d, f"alf{depth}.yuv", w, h, frames, depth)
    args = ["--input-res", f"{w}x{h}", "-n", str(frames), "-p", "1", "--preset", "ultrafast", "-q", "27", "--alf", "full"] + (["--input-bitdepth", "10"] if depth == 10 else [])
    want, _ = encode(need(os.path.join(REF, f"uvg266_{depth}")), yuv, str(d / f"generic_alf{depth}.266"), {}, ["--no-cpuid"], args=args, threads=1)
    got, err = encode(need(os.path.join(REF, f"uvg266_{depth}_hip")), yuv, str(d / f"hip_alf{depth}.266"), {"UVG266_HIP": "alf"}, args=args, threads=1)
    sel = chosen(err)
    for t in ALF_TYPES:
        assert sel.get(t) == "hip", (t, sel.get(t))
    assert got == want
