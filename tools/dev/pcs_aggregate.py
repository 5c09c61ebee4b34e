"""dev: fold a rocprofv3 PC-sampling CSV into a histogram per instruction (and per source line when the code object has line tables).
usage: pcs_aggregate.py <dir with *pc_sampling*.csv> <out.txt> [kernel substring]"""
import sys, os, csv, glob, collections
csv.field_size_limit(1 << 30)
d, out = sys.argv[1], sys.argv[2]
files = [f for f in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True) if "pc_sampl" in os.path.basename(f)]
lines = []
lines.append("files: " + " ".join(files))
for f in files:
    with open(f, newline="") as fh:
        rd = csv.DictReader(fh)
        cols = rd.fieldnames
        lines.append(f"== {f}\ncolumns: {cols}")
        by_inst = collections.Counter(); by_line = collections.Counter(); by_stall = collections.Counter(); by_type = collections.Counter()
        issued = collections.Counter(); lanes = collections.Counter()
        n = 0
        for r in rd:
            n += 1
            inst = r.get("Instruction", ""); cm = r.get("Instruction_Comment", "")
            key = (r.get("Code_Object_Id", r.get("Codeobj", "")), r.get("Code_Object_Offset", r.get("Vaddr", "")), inst, cm)
            by_inst[key] += 1
            by_line[cm] += 1
            if "Stall_Reason" in r: by_stall[r["Stall_Reason"]] += 1
            if "Instruction_Type" in r: by_type[r["Instruction_Type"]] += 1
            if "Wave_Issued_Instruction" in r: issued[r["Wave_Issued_Instruction"]] += 1
            em = r.get("Exec_Mask", "")
            try: lanes[bin(int(em, 0)).count("1") // 8 * 8] += 1
            except Exception: pass
        lines.append(f"samples: {n}")
        for name, c in (("stall reason", by_stall), ("instruction type", by_type), ("issued", issued), ("active lanes (floor 8)", lanes)):
            if c: lines.append(f"-- {name}: " + ", ".join(f"{k}={v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
        lines.append("-- top source lines")
        for k, v in by_line.most_common(400): lines.append(f"{v:9d} {100.0 * v / max(n, 1):6.2f}%  {k}")
        lines.append("-- top instructions")
        for k, v in by_inst.most_common(1500): lines.append(f"{v:9d} {100.0 * v / max(n, 1):6.2f}%  {k[0]}:{k[1]}  {k[2]}   ; {k[3]}")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
