#!/usr/bin/env python3
"""Every batched entry point once at a 1080p-sized workload (8-bit unless noted), five launches each, so that a
`rocprofv3 --kernel-trace --stats` run of this script (tools/profile_zoo.sh) yields one duration per kernel.  With
--summarize <kernel_stats.csv> it prints/writes the table: kernel, calls, average us, algorithmic MB, GB/s, % of 8 TB/s.
Not a benchmark line -- bench.py is; this is the per-kernel evidence for the rows bench.py does not exercise."""
import csv
import json
import sys

import os
W, H = 1920, 1080
REPS = 5
DEPTH = int(os.environ.get("ZOO_DEPTH", "8"))      # ZOO_DEPTH=10: the same workloads on 10-bit planes (uint16)


def workloads():
    import numpy as np
    import torch
    sys.path.insert(0, ".")
    from uvg266_amd import api, layout
    dev = torch.device("cuda:0")
    y0, u0, v0 = layout.synthetic_yuv420(W, H, 0, DEPTH)
    y1, _, _ = layout.synthetic_yuv420(W, H, 1, DEPTH)
    Y0, Y1 = torch.from_numpy(y0).to(dev), torch.from_numpy(y1).to(dev)
    U0, V0 = torch.from_numpy(u0).to(dev), torch.from_numpy(v0).to(dev)
    rng = np.random.default_rng(0)
    out = []

    def add(name, fn, alg_bytes, launches=1):
        out.append((name, fn, alg_bytes, launches))

    # --- picture group: 16x16 blocks of the frame against the previous frame at a small random displacement
    xy = layout.block_grid(W, H, 16)
    mv = rng.integers(-8, 9, xy.shape)
    blks = api.make_blocks(xy, xy + mv)
    n = len(xy)
    add("sad_batch 16x16", lambda: api.sad_batch(Y0, Y1, 16, 16, blks), n * (2 * 256 + 4))
    add("satd_batch 16x16", lambda: api.satd_batch(Y0, Y1, 16, 16, blks), n * (2 * 256 + 4))
    add("ssd_batch 16x16", lambda: api.ssd_batch(Y0, Y1, 16, 16, blks), n * (2 * 256 + 4))
    add("sad_surface 16x16 r8", lambda: api.sad_surface(Y0, Y1, W, H - H % 16, 16, 16, 8), 2 * W * H + n * 17 * 17 * 4)
    add("residual_plane", lambda: api.residual_plane(Y0, Y1), W * H * 4)
    xy8 = layout.block_grid(W, H, 8)
    tus8 = api.make_tus(xy8)
    add("crc32c 8x8", lambda: api.crc32c_batch(Y0, tus8, 8), len(xy8) * 68)
    ctus = Y0[:1024].reshape(16, 64, 30, 64).permute(0, 2, 1, 3).reshape(480, 4096).contiguous()
    add("pixel_var 64x64", lambda: api.pixel_var_batch(ctus), 480 * 4096)
    # --- transforms / quant on 2 Mi coefficients
    for nsz in (4, 8, 16, 32):
        nb = (W // nsz) * (H // nsz)
        blocks = torch.randint(-255, 256, (nb, nsz, nsz), dtype=torch.int16, device=dev)
        add(f"transform fwd DCT2 {nsz}", lambda b=blocks: api.transform_batch(b, DEPTH, False), nb * nsz * nsz * 4)
        add(f"transform inv DST7 {nsz}", lambda b=blocks: api.transform_batch(b, DEPTH, True, 2, 2), nb * nsz * nsz * 4)
    coef = torch.randint(-2000, 2001, ((W // 8) * (H // 8), 8, 8), dtype=torch.int16, device=dev)
    add("quant 8x8", lambda: api.quant_batch(coef, 8, 22), coef.numel() * 4)
    add("dequant 8x8", lambda: api.dequant_batch(coef, 8, 22), coef.numel() * 4)
    add("coeff_abs_sum 8x8", lambda: api.coeff_abs_sum_batch(coef), coef.numel() * 2)
    lt = api.make_lfnst_tus(np.stack([np.arange(len(xy8)) % 67, 1 + np.arange(len(xy8)) % 2, np.full(len(xy8), 3), np.full(len(xy8), 3)], 1))
    add("lfnst fwd 8x8", lambda: api.lfnst_batch(coef, lt, False), len(xy8) * 2 * 96)
    # --- intra prediction
    for nsz in (8, 32):
        ib = api.make_intra_blocks(layout.intra_availability(layout.block_grid(W, H, nsz), nsz, W, H))
        modes = api.make_modes([0, 1, 18, 50, 34, 66, 2, 60])
        add(f"intra_pred_batch {nsz} x8 modes", lambda ib=ib, nsz=nsz, m=modes: api.intra_pred_batch(Y0, ib, nsz, nsz, m),
            ib.shape[0] * (4 * nsz + 1 + 8 * nsz * nsz))
    ib16 = api.make_intra_blocks(layout.intra_availability(layout.block_grid(W, H, 16), 16, W, H))
    mt = (torch.arange(ib16.shape[0], device=dev) % 6).to(torch.uint8)
    add("mip_pred_batch 16", lambda: api.mip_pred_batch(Y0, ib16, 16, 16, mt), ib16.shape[0] * (32 + 256))
    # --- interpolation
    mcb = api.make_mc_blocks(np.concatenate([xy + mv, rng.integers(0, 16, xy.shape)], 1))
    add("mc_batch 16x16 luma", lambda: api.mc_batch(Y1, mcb, 16, 16), n * (23 * 23 + 256))
    cand = torch.from_numpy(np.array([[0, 0], [8, 0], [-8, 0], [0, 8], [0, -8], [4, 4], [-4, 4], [4, -4], [-4, -4]], np.int16) ).to(dev)
    fb = api.make_blocks(xy, xy + mv)
    add("frac_satd 16x16 x9", lambda: api.frac_satd_batch(Y0, Y1, fb, 16, 16, cand), n * (23 * 23 + 256 + 36))
    l0 = (Y0.reshape(-1) if DEPTH == 8 else Y0.reshape(-1))
    add("bipred_average px/px", lambda: api.bipred_average_batch(l0, l0, DEPTH), W * H * 3)
    # --- in-loop filters
    scu = api.make_scu_table(layout.quadtree_scu_table(W, H, seed=0, qp=22))
    rec = Y0.clone()
    add("deblock_frame (2 launches)", lambda: api.deblock_frame(rec, U0.clone(), V0.clone(), scu, W, H, frame_qp=22), int(2 * 1.5 * W * H + 32 * W * H / 16), 2)
    rects_np = layout.ctu_rects(W, H)
    rects = api.make_rects(rects_np)
    add("sao_stats", lambda: api.sao_stats_batch(Y0, Y1, rects), 2 * W * H + len(rects_np) * 416)
    edge, _ = api.sao_stats_batch(Y0, Y1, rects)
    params = api.sao_edge_offsets_batch(edge)
    sout = torch.zeros_like(Y0)
    add("sao_apply", lambda: api.sao_apply_batch(Y1, sout, rects, params), 2 * W * H)
    cls = api.alf_classify_frame(Y1, W, H)
    add("alf_classify", lambda: api.alf_classify_frame(Y1, W, H), W * H + W * H // 16)
    coefs = torch.zeros((1, 25, 13), dtype=torch.int16, device=dev); coefs[:, :, 12] = 0
    clips = torch.full((1, 25, 13), (1 << DEPTH) - 1, dtype=torch.int16, device=dev)
    sidx = torch.zeros(len(rects_np), dtype=torch.int32, device=dev)
    add("alf_filter luma 7x7", lambda: api.alf_filter_batch(Y1, sout, rects, sidx, coefs, clips, cls), 2 * W * H)
    add("alf_stats luma", lambda: api.alf_stats_batch(Y0, Y1, rects, cls), 2 * W * H + len(rects_np) * 25 * (13 * 13 * 16 * 8 + 13 * 4 * 4 + 8))
    return out


def run():
    import torch
    items = workloads()
    names = []
    for name, fn, alg, launches in items:
        for _ in range(REPS):
            fn()
        torch.cuda.synchronize()
        names.append({"name": name, "alg_bytes": int(alg), "kernels_per_call": launches, "reps": REPS})
    json.dump(names, open("gpurun_out/zoo_plan.json", "w"))
    print("ran", len(names), "workloads")


def summarize(trace_csv, plan_json, tag):
    """Walk the kernel trace in time order, skip torch/runtime kernels, hand `reps * kernels_per_call` kernels to each
    workload of the plan in turn."""
    rows = sorted(csv.DictReader(open(trace_csv)), key=lambda r: int(r["Start_Timestamp"]))
    ours = [r for r in rows if "at::native" not in r["Kernel_Name"] and "rocclr" not in r["Kernel_Name"]]
    plan = json.load(open(plan_json))
    # workloads() itself launches three of our kernels while it prepares inputs (sao_stats, sao_edge_offsets,
    # alf_classify), before any timed call
    table, i = [], N_SETUP
    for w in plan:
        k = w["reps"] * w["kernels_per_call"]
        seg = ours[i:i + k]; i += k
        per_call_us = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1000.0 / w["reps"]
        gbs = w["alg_bytes"] / (per_call_us * 1e-6) / 1e9
        table.append({"workload": w["name"], "kernel": seg[0]["Kernel_Name"].split("(")[0][:60], "us_per_call": round(per_call_us, 1),
                      "alg_MB": round(w["alg_bytes"] / 1e6, 2), "GBps": round(gbs, 1), "pct_of_8TBps": round(100 * gbs / 8000, 2)})
    depth = 10 if "10bit" in tag else 8
    json.dump({"note": f"rocprofv3 --kernel-trace durations of tools/kernel_zoo.py, 1080p {depth}-bit, one launch alone on the GPU; "
                       "algorithmic bytes as in DESIGN.md section 4", "kernels": table}, open(f"profiles/{tag}_kernel_zoo.json", "w"), indent=1)
    for t in table:
        print(f'{t["workload"]:32s} {t["kernel"][:44]:44s} {t["us_per_call"]:8.1f} us {t["alg_MB"]:8.2f} MB {t["GBps"]:8.1f} GB/s {t["pct_of_8TBps"]:6.2f} %')


N_SETUP = 3


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "r01")
    else:
        run()
