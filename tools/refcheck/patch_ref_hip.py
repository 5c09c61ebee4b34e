#!/usr/bin/env python3
"""Apply INTEGRATION.md section 1 + 2 to a SCRATCH COPY of the reference's src/ tree (never to /root/reference, never into the repo):
one registration block per strategy group, in front of the `return success;` of uvg_strategy_register_<group>().
usage: patch_ref_hip.py <scratch>/src

The block is the text INTEGRATION.md shows, with one extension used by the tests: UVG266_HIP may name the groups to register
("picture,dct"); "1" / "all" registers every group.
"""
import re
import sys

GROUPS = {
    # group: (file, registrars called)
    "picture": ("strategies/strategies-picture.c", ["uvg_strategy_register_picture_hip", "uvg_strategy_register_state_hip_picture"]),
    "dct": ("strategies/strategies-dct.c", ["uvg_strategy_register_dct_hip"]),
    "intra": ("strategies/strategies-intra.c", ["uvg_strategy_register_intra_hip"]),
    "sao": ("strategies/strategies-sao.c", ["uvg_strategy_register_sao_hip"]),
    "quant": ("strategies/strategies-quant.c", ["uvg_strategy_register_quant_hip", "uvg_strategy_register_state_hip_quant"]),
    "ipol": ("strategies/strategies-ipol.c", ["uvg_strategy_register_ipol_hip"]),
    "alf": ("strategies/strategies-alf.c", ["uvg_strategy_register_state_hip_alf"]),
}


def block(group, regs):
    lines = ["#if defined(UVG_HAVE_HIP)", "  {", "    extern int uvg_hip_group_enabled(const char *group);      /* strategies/hip/strategies-hip-state.c */"]
    for fn in regs:
        lines.append(f"    extern int {fn}(void *opaque, uint8_t bitdepth);")
    for fn in regs:
        lines.append(f"    if (uvg_hip_group_enabled(\"{group}\")) success &= {fn}(opaque, bitdepth);")
    lines += ["  }", "#endif"]
    return "\n".join(lines) + "\n"


def main(src):
    for group, (rel, regs) in GROUPS.items():
        path = f"{src}/{rel}"
        text = open(path).read()
        m = re.search(r"int\s+uvg_strategy_register_%s\s*\(" % group, text)
        if not m:
            sys.exit(f"patch_ref_hip: uvg_strategy_register_{group} not found in {rel}")
        r = text.find("return success;", m.end())
        if r < 0:
            sys.exit(f"patch_ref_hip: no `return success;` behind uvg_strategy_register_{group}")
        line_start = text.rfind("\n", 0, r) + 1
        text = text[:line_start] + block(group, regs) + text[line_start:]
        if "stdlib.h" not in text:
            text = "#include <stdlib.h>\n" + text
        open(path, "w").write(text)
        print(f"patched {rel}")


# the frame-level hand-over (uvg266_amd/csrc/shim/frame-hip.c).  Two statements + the pool's release.
FRAME = [
    ("encoderstate.c", r"void\s+uvg_encode_one_frame\s*\(", "  encoder_state_encode(state);\n",
     "#if defined(UVG_HAVE_HIP)\n"
     "  extern int uvg_hip_frame_enabled(const encoder_state_t *state);       /* strategies/hip/frame-hip.c */\n"
     "  extern void uvg_hip_frame_begin(encoder_state_t *state);\n"
     "  if (uvg_hip_frame_enabled(state)) uvg_hip_frame_begin(state); else\n"
     "#endif\n"
     "  encoder_state_encode(state);\n"),
    ("encoder_state-bitstream.c", r"void\s+uvg_encoder_state_worker_write_bitstream\s*\(", "  uvg_encoder_state_write_bitstream((encoder_state_t *) opaque);\n",
     "#if defined(UVG_HAVE_HIP)\n"
     "  { extern void uvg_hip_frame_finish(encoder_state_t *state); uvg_hip_frame_finish((encoder_state_t *) opaque); }\n"
     "#endif\n"
     "  uvg_encoder_state_write_bitstream((encoder_state_t *) opaque);\n"),
    # the pool goes with the encoder instance: behind the stop of the thread queue (no bitstream job is running any more)
    ("uvg266.c", r"static\s+void\s+uvg266_close\s*\(", "    if (encoder->states) {\n",
     "#if defined(UVG_HAVE_HIP)\n"
     "    { extern void uvg_hip_frame_close(const encoder_control_t *ctrl); uvg_hip_frame_close(encoder->control); }\n"
     "#endif\n"
     "    if (encoder->states) {\n"),
]


def patch_frame(src):
    for rel, fn, old, new in FRAME:
        path = f"{src}/{rel}"
        text = open(path).read()
        m = re.search(fn, text)
        if not m:
            sys.exit(f"patch_ref_hip: {fn} not found in {rel}")
        at = text.find(old, m.end())
        if at < 0 or at - m.end() > 4000:
            sys.exit(f"patch_ref_hip: `{old.strip()}` not found behind {fn} in {rel}")
        text = text[:at] + new + text[at + len(old):]
        open(path, "w").write(text)
        print(f"patched {rel} (frame hand-over)")


if __name__ == "__main__":
    main(sys.argv[1])
    patch_frame(sys.argv[1])
