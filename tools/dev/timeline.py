"""Developer probe: concurrency summary of a rocprofv3 --kernel-trace CSV (which kernels overlap, GPU idle gaps)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
t0 = rows[len(rows) // 2][0]
sel = [r for r in rows if r[0] >= t0][:400]
def short(n):
    for k in ('rdoq', 'intra_search', 'tu_roundtrip', 'intra_pred_plane', 'deblock', 'sao_stats', 'sao_apply', 'sao_edge', 'quant_kernel', 'alf', 'lfnst', 'has_coeffs'):
        if k in n: return k
    return n[:24]
span = sel[-1][1] - sel[0][0]
busy = collections.Counter(); 
ev = []
for s, e, n in sel: ev += [(s, 1, short(n)), (e, -1, short(n))]
ev.sort()
cur = collections.Counter(); last = ev[0][0]; conc_time = collections.Counter()
for tt, d, n in ev:
    k = sum(cur.values())
    conc_time[k] += tt - last; last = tt
    cur[n] += d
print("span us", span / 1e3, "kernels", len(sel))
print("time by number of concurrently running kernels (us):", {k: round(v / 1e3, 1) for k, v in sorted(conc_time.items())})
dur = collections.defaultdict(list)
for s, e, n in sel: dur[short(n)].append((e - s) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])): print(f"{k:18s} n={len(v):4d} avg {sum(v)/len(v):8.1f} us total {sum(v):9.1f}")
