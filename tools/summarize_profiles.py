#!/usr/bin/env python3
"""Condense gpurun_out/prof_* (tools/profile_bench.sh) into profiles/<tag>_*: the rocprofv3 kernel statistics tables as they
are, HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes, SQ instruction counts per launch, and the matrix-core
counters of the ALF covariance kernel.  rocprofv3 reports FETCH_SIZE / WRITE_SIZE in kilobytes: bytes = value * 1024, and
FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md: it counts 64-byte units of 128-byte requests).
Launches are keyed by the name bench.py gives them: kernel symbol (+ grid size where one symbol serves several block sizes)."""
import collections, csv, glob, json, os, re, shutil, sys


def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
shutil.copy(newest("gpurun_out/prof_stats/**/*kernel_stats.csv"), f"profiles/{tag}_bench_kernel_stats.csv")
shutil.copy(newest("gpurun_out/prof_stats_serial/**/*kernel_stats.csv"), f"profiles/{tag}_bench_kernel_stats_serial.csv")
shutil.copy("gpurun_out/prof_bench_line.json", f"profiles/{tag}_bench_line_profiled.json")
if os.path.exists("gpurun_out/prof_command.txt"):
    shutil.copy("gpurun_out/prof_command.txt", f"profiles/{tag}_profiled_command.txt")
try:
    shutil.copy(newest("gpurun_out/prof_stats_4k/**/*kernel_stats.csv"), f"profiles/{tag}_bench4k_kernel_stats_serial.csv")
except ValueError:
    pass

W, H = 1920, 1080
SEARCH_GRID = {}
for n, bpg, thr in ((32, 4, 512), (16, 16, 512), (8, 64, 512), (4, 64, 256)):
    for F in (1, 2, 4, 5, 8, 10, 16, 20):            # one launch covers the F pictures of a group (bench --group)
        SEARCH_GRID.setdefault(str(-(-(F * (W // n) * (H // n)) // bpg) * thr), f"intra_search_{n}")
RDOQ = {("5", "0"): "rdoq_32", ("4", "0"): "rdoq_16", ("3", "0"): "rdoq_8", ("2", "0"): "rdoq_4",
        ("4", "1"): "rdoq_chroma_32", ("3", "1"): "rdoq_chroma_16", ("2", "1"): "rdoq_chroma_8"}


def short(kernel):
    m = re.search(r"(\w+)(<[^(]*>)?\(", kernel)
    return (m.group(1) + (m.group(2) or "")) if m else kernel.split("(")[0]


def bench_name(kernel, grid):
    k = short(kernel)
    if k.startswith("intra_search_kernel"):
        return SEARCH_GRID.get(grid)
    m = re.match(r"rdoq_kernel<16, (\d), (\d)(?:, \d)?>", k)
    if m:
        return RDOQ.get((m.group(1), m.group(2)))
    if k.startswith("rdoq_pre_kernel"):          # the decision-free pass in front of the 32x32 walk: same entry point, same bench name
        return "rdoq_32"
    return None


def per_kernel(dirname, counter):
    f = newest(f"gpurun_out/{dirname}/**/*counter_collection.csv")
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[(short(r["Kernel_Name"]), r["Grid_Size"], bench_name(r["Kernel_Name"], r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


fetch, nf = per_kernel("prof_fetch", "FETCH_SIZE")
write, _ = per_kernel("prof_write", "WRITE_SIZE")
rows, by_bench = [], {}
for k in sorted(fetch, key=lambda k: -fetch[k]):
    fb, wb = fetch[k] * 1024 * 2, write.get(k, 0.0) * 1024
    rows.append({"kernel": k[0], "grid": k[1], "bench_name": k[2], "launches": nf[k], "fetch_bytes_per_launch": round(fb),
                 "write_bytes_per_launch": round(wb), "hbm_bytes_per_launch": round(fb + wb)})
    if k[2]:
        by_bench[k[2]] = by_bench.get(k[2], 0) + round(fb + wb)      # (an entry point may launch two kernels)
json.dump(by_bench, open("profiles/hbm_traffic_latest.json", "w"), indent=1)
json.dump({"note": "FETCH_SIZE x 1024 x 2 (gfx950 correction) + WRITE_SIZE x 1024, averaged per launch; separate --pmc passes of the --serial bench",
           "kernels": rows}, open(f"profiles/{tag}_bench_hbm_traffic.json", "w"), indent=1)
valu, _ = per_kernel("prof_valu", "SQ_INSTS_VALU")
salu, _ = per_kernel("prof_valu", "SQ_INSTS_SALU")
lds, _ = per_kernel("prof_valu", "SQ_INSTS_LDS")
waves, _ = per_kernel("prof_valu", "SQ_WAVES")
vb = {}
for k, v in valu.items():
    if k[2]:
        e = vb.setdefault(k[2], {"valu_insts": 0, "salu_insts": 0, "lds_insts": 0, "waves": 0})
        e["valu_insts"] += round(v); e["salu_insts"] += round(salu.get(k, 0)); e["lds_insts"] += round(lds.get(k, 0)); e["waves"] += round(waves.get(k, 0))
json.dump(vb, open("profiles/valu_latest.json", "w"), indent=1)
json.dump({"note": "SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS / SQ_WAVES per launch (wave-level instruction counts), --serial bench",
           "kernels": [{"kernel": k[0], "grid": k[1], "bench_name": k[2], "valu_insts": round(v), "salu_insts": round(salu.get(k, 0)),
                        "lds_insts": round(lds.get(k, 0)), "waves": round(waves.get(k, 0))} for k, v in sorted(valu.items(), key=lambda kv: -kv[1])]},
          open(f"profiles/{tag}_bench_sq_insts.json", "w"), indent=1)
try:
    mf, _ = per_kernel("prof_mfma_4k", "SQ_INSTS_VALU_MFMA_I8")
    busy, _ = per_kernel("prof_mfma_4k", "SQ_VALU_MFMA_BUSY_CYCLES")
    sqb, _ = per_kernel("prof_mfma_4k", "SQ_BUSY_CYCLES")
    va, _ = per_kernel("prof_mfma_4k", "SQ_INSTS_VALU")
    wv, _ = per_kernel("prof_mfma_4k", "SQ_WAVES")
    out = [{"kernel": k[0], "grid": k[1], "mfma_i8_insts": round(v), "mfma_busy_cycles": round(busy.get(k, 0)), "sq_busy_cycles": round(sqb.get(k, 0)),
            "valu_insts": round(va.get(k, 0)), "waves": round(wv.get(k, 0)),
            "mfma_busy_over_sq_busy": round(busy.get(k, 0) / sqb[k], 4) if sqb.get(k) else None} for k, v in mf.items() if v > 0]
    json.dump({"note": "2160p10alf --serial: matrix-core counters per launch.  v_mfma_i32_32x32x32_i8 = 32768 MACs; SQ_VALU_MFMA_BUSY_CYCLES and "
                       "SQ_BUSY_CYCLES are summed over the shader engines as rocprofv3 reports them", "kernels": out},
              open(f"profiles/{tag}_bench4k_mfma.json", "w"), indent=1)
except (IndexError, ValueError, KeyError) as e:
    print("no mfma capture:", e)
for r in rows[:12]:
    print(r)
print(json.dumps(vb, indent=0)[:600])
