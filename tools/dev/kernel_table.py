import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], d['unit'], d['ms_per_step'])
ks=d['kernels']
print('sum',sum(r['per_step_ms'] for r in ks.values()))
for name,row in sorted(ks.items(), key=lambda kv:-kv[1]['per_step_ms'])[:int(sys.argv[2]) if len(sys.argv)>2 else 12]:
    print(f"{name:28s} {row['per_step_ms']:.4f} avg {row['avg_ms']:.4f} x{row['launches']}")
for w,e in d.get('extra_workloads',{}).items():
    print(w, e['value'], e['ms_per_step'])
    for name,row in list(e['kernels_top'].items())[:10]: print(f"   {name:26s} {row['per_step_ms']:.4f} avg {row['avg_ms']:.4f}")
