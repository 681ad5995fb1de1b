"""Per-queue, per-kernel time of the steady-state steps in a rocprofv3 kernel-trace CSV (usage: trace_by_queue.py <kernel_trace.csv> [first_step] [n_steps])."""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ad = [i for i, r in enumerate(rows) if 'adamw_kernel' in r['Kernel_Name']]
k0 = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
sel = rows[ad[k0] + 1:ad[k0 + n] + 1]
wall = (rows[ad[k0 + n]]['e'] - rows[ad[k0]]['e']) / n / 1e6
def short(nm):
    nm = re.sub(r'\(anonymous namespace\)::', '', nm); nm = nm.replace('void ', '')
    m = re.match(r'([\w:]+)(<[^(]*>)?', nm)
    return (m.group(1) + (m.group(2) or ''))[:70] if m else nm[:70]
agg = collections.defaultdict(lambda: [0, 0.0])
qtot = collections.defaultdict(float)
for r in sel:
    key = (r['Queue_Id'], short(r['Kernel_Name']))
    agg[key][0] += 1; agg[key][1] += (r['e'] - r['s'])
    qtot[r['Queue_Id']] += (r['e'] - r['s'])
print(f"step wall {wall:.2f} ms (profiled); kernel time per queue per step:", {q: round(v / n / 1e6, 2) for q, v in qtot.items()})
for (q, name), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 40]:
    print(f"q{q} {t / n / 1e6:7.3f} ms {c / n:6.1f} x {t / c / 1e3:7.1f} us  {name}")
