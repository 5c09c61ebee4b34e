#!/usr/bin/env python3
"""Condense gpurun_out/prof_* (tools/profile_bench.sh) into profiles/rNN_*: the rocprofv3 kernel
statistics table as is, and per-kernel HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes.
FETCH_SIZE is doubled (MI355X_MICROARCH.md: on gfx950 it counts 64-byte units of 128-byte requests);
both counters are in KiB-like units of 1024 B?  -- no: rocprofv3 reports FETCH_SIZE/WRITE_SIZE in
kilobytes (derived metric), so bytes = value * 1024 (* 2 for FETCH_SIZE on gfx950)."""
import collections, csv, glob, json, os, shutil, sys

def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
stats = newest("gpurun_out/prof_stats/**/*kernel_stats.csv")
shutil.copy(stats, f"profiles/{tag}_bench_kernel_stats.csv")
shutil.copy(newest("gpurun_out/prof_stats_serial/**/*kernel_stats.csv"), f"profiles/{tag}_bench_kernel_stats_serial.csv")
shutil.copy("gpurun_out/prof_bench_line.json", f"profiles/{tag}_bench_line_profiled.json")
try:
    shutil.copy(newest("gpurun_out/prof_stats_4k/**/*kernel_stats.csv"), f"profiles/{tag}_bench4k_kernel_stats_serial.csv")
except ValueError:
    pass

def per_kernel(dirname, counter):
    f = newest(f"gpurun_out/{dirname}/**/*counter_collection.csv")
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[(r["Kernel_Name"].split("(")[0], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}

fetch, nf = per_kernel("prof_fetch", "FETCH_SIZE")
write, _ = per_kernel("prof_write", "WRITE_SIZE")
rows = []
for k in sorted(fetch, key=lambda k: -fetch[k]):
    fb = fetch[k] * 1024 * 2
    wb = write.get(k, 0.0) * 1024
    rows.append({"kernel": k[0], "grid": k[1], "launches": nf[k], "fetch_bytes_per_launch": round(fb),
                 "write_bytes_per_launch": round(wb), "hbm_bytes_per_launch": round(fb + wb)})
# bench.py kernel names of the search launches, by grid size (1080p: blocks per size / blocks per workgroup x threads)
W, H = 1920, 1080
names = {}
for n, bpg, thr in ((32, 4, 512), (16, 16, 512), (8, 64, 512), (4, 64, 256)):
    blocks = (W // n) * (H // n)
    names[str(-(-blocks // bpg) * thr)] = f"intra_search_{n}"
by_bench = {}
for r in rows:
    if "intra_search_kernel" in r["kernel"] and r["grid"] in names:
        r["bench_name"] = names[r["grid"]]
        by_bench[names[r["grid"]]] = r["hbm_bytes_per_launch"]
json.dump(by_bench, open("profiles/hbm_traffic_latest.json", "w"), indent=1)
try:
    valu, _ = per_kernel("prof_valu", "SQ_INSTS_VALU")
    lds, _ = per_kernel("prof_valu", "SQ_INSTS_LDS")
    waves, _ = per_kernel("prof_valu", "SQ_WAVES")
    vb = {}
    for k, v in valu.items():
        if "intra_search_kernel" in k[0] and k[1] in names:
            vb[names[k[1]]] = {"valu_insts": round(v), "lds_insts": round(lds.get(k, 0)), "waves": round(waves.get(k, 0))}
    json.dump(vb, open("profiles/valu_latest.json", "w"), indent=1)
    json.dump({"note": "SQ_INSTS_VALU / SQ_INSTS_LDS / SQ_WAVES per launch (wave-level instruction counts), --serial bench",
               "kernels": [{"kernel": k[0], "grid": k[1], "valu_insts": round(v), "lds_insts": round(lds.get(k, 0)), "waves": round(waves.get(k, 0))}
                           for k, v in sorted(valu.items(), key=lambda kv: -kv[1])]},
              open(f"profiles/{tag}_bench_sq_insts.json", "w"), indent=1)
except (IndexError, ValueError):
    pass
json.dump({"note": "FETCH_SIZE x 1024 x 2 (gfx950 correction) + WRITE_SIZE x 1024, averaged per launch; separate --pmc passes",
           "kernels": rows}, open(f"profiles/{tag}_bench_hbm_traffic.json", "w"), indent=1)
for r in rows[:24]:
    print(r)
