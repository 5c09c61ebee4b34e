"""Developer probe: per-queue view of a rocprofv3 --kernel-trace CSV around the middle of the run."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rd = list(csv.DictReader(open(f)))
print(list(rd[0].keys()))
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id'), r.get('Stream_Id'), int(r.get('Grid_Size', 0)), int(r.get('LDS_Block_Size', 0) or 0)) for r in rd)
mid = rows[len(rows) * 2 // 3][0]
sel = [r for r in rows if r[0] >= mid][:120]
t0 = sel[0][0]
for s, e, n, q, st, g, lds in sel:
    nm = 'rdoq' if 'rdoq' in n else ('search' if 'intra_search' in n else n[:28])
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} q{q} s{st} grid {g:8d} lds {lds:6d} {nm}")
