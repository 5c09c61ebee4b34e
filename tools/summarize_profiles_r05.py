#!/usr/bin/env python3
"""Condense gpurun_out/prof5_* (tools/profile_bench_r05.sh) into profiles/r05_*: the rocprofv3 kernel statistics tables as they
are, HBM bytes per launch of every kernel of the closed loop from the FETCH_SIZE / WRITE_SIZE passes, SQ instruction counts per
launch.  rocprofv3 reports FETCH_SIZE / WRITE_SIZE in kilobytes: bytes = value * 1024, and FETCH_SIZE is doubled on gfx950
(MI355X_MICROARCH.md: it counts 64-byte units of 128-byte requests).  profiles/hbm_traffic_latest.json gets the key
"ctu_search" (bytes per launch of ctu_search_kernel) that bench.py reports as roofline.traffic."""
import collections, csv, glob, json, os, re, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"


def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def short(kernel):
    m = re.search(r"(\w+)(<[^(]*>)?\(", kernel)
    return (m.group(1) + (m.group(2) or "")) if m else kernel.split("(")[0]


shutil.copy(newest("gpurun_out/prof5_stats/**/*kernel_stats.csv"), f"profiles/{tag}_bench_kernel_stats.csv")
shutil.copy(newest("gpurun_out/prof5_stats_1/**/*kernel_stats.csv"), f"profiles/{tag}_bench_kernel_stats_one_launch_in_flight.csv")
for a, b in (("gpurun_out/prof5_bench_line.json", f"profiles/{tag}_bench_line_profiled.json"),
             ("gpurun_out/prof5_bench_line_1.json", f"profiles/{tag}_bench_line_profiled_one_launch_in_flight.json")):
    if os.path.getsize(a):          # (empty: the profiled run did not print its line -- the kernel tables are complete all the same)
        shutil.copy(a, b)
shutil.copy("gpurun_out/prof5_command.txt", f"profiles/{tag}_profiled_command.txt")


def per_kernel(dirname, counters):
    f = newest(f"gpurun_out/{dirname}/**/*counter_collection.csv")
    acc = {c: collections.defaultdict(list) for c in counters}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in acc:
            acc[r["Counter_Name"]][short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {c: {k: (sum(v) / len(v), len(v)) for k, v in d.items()} for c, d in acc.items()}


fetch = per_kernel("prof5_fetch", ["FETCH_SIZE"])["FETCH_SIZE"]
write = per_kernel("prof5_write", ["WRITE_SIZE"])["WRITE_SIZE"]
try:
    pics = json.loads(open("gpurun_out/prof5_bench_line_1.json").read())["config"]["pictures_per_step"]
except Exception:          # noqa: BLE001
    pics = 224          # bench.py's default group (--in-flight)
traffic = {}
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, (0.0, 0))
    w, nw = write.get(k, (0.0, 0))
    traffic[k] = {"launches": max(nf, nw), "fetch_bytes_per_launch": round(f * 1024 * 2), "write_bytes_per_launch": round(w * 1024),
                  "hbm_bytes_per_launch": round(f * 1024 * 2 + w * 1024)}
out = {"command": open("gpurun_out/prof5_command.txt").read().strip() + "  (with --groups 1 --steps 2 --warmup 1 for the counter passes)",
       "pictures_per_search_launch": pics, "kernels": traffic,
       "note": "FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 (rocprofv3 reports KB; FETCH_SIZE counts 64-byte halves of 128-byte requests on gfx950)"}
json.dump(out, open(f"profiles/{tag}_bench_hbm_traffic.json", "w"), indent=1)
latest = {}
if os.path.exists("profiles/hbm_traffic_latest.json"):
    latest = json.load(open("profiles/hbm_traffic_latest.json"))
import hashlib
sha = hashlib.sha1(b"".join(open(os.path.join("uvg266_amd", "csrc", f), "rb").read() for f in ("ctu_core.h", "ctu_leaf4.h", "ctu_search.hip"))).hexdigest()
ck = [k for k in traffic if k.startswith("ctu_search_kernel")]
if ck:
    t = traffic[ck[0]]
    latest["ctu_search"] = {"bytes_per_launch": t["hbm_bytes_per_launch"], "pictures_per_launch": pics,
                            "bytes_per_picture": round(t["hbm_bytes_per_launch"] / pics), "tag": tag,
                            "source_sha1": sha}          # of ctu_core.h + ctu_search.hip as profiled: bench.py refuses the number for other sources
json.dump(latest, open("profiles/hbm_traffic_latest.json", "w"), indent=1)
try:
    sq = per_kernel("prof5_sq", ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"])
    json.dump({c: {k: {"per_launch": round(v[0]), "launches": v[1]} for k, v in d.items()} for c, d in sq.items()},
              open(f"profiles/{tag}_bench_sq_insts.json", "w"), indent=1)
except Exception as e:          # noqa: BLE001 -- an optional pass
    print("no SQ instruction pass:", e, file=sys.stderr)
try:
    occ = per_kernel("prof5_occ", ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"])
    json.dump({c: {k: {"per_launch": round(v[0]), "launches": v[1]} for k, v in d.items()} for c, d in occ.items()},
              open(f"profiles/{tag}_bench_sq_occupancy.json", "w"), indent=1)
except Exception as e:          # noqa: BLE001 -- an optional pass
    print("no occupancy pass:", e, file=sys.stderr)
print(json.dumps(latest.get("ctu_search")), file=sys.stderr)
