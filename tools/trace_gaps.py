"""Where the main HIP queue of the training step sits idle: gaps between consecutive kernels of one steady-state step in a rocprofv3
kernel-trace CSV, grouped by the kernel that FOLLOWS the gap (usage: trace_gaps.py <kernel_trace.csv> [step index] [min gap us])."""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ad = [i for i, r in enumerate(rows) if 'adamw_kernel' in r['Kernel_Name']]
k0 = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ming = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
sel = rows[ad[k0] + 1:ad[k0 + 1] + 1]
def short(nm):
    nm = re.sub(r'\(anonymous namespace\)::', '', nm).replace('void ', '')
    m = re.match(r'([\w:]+)(<[^(]*>)?', nm)
    return (m.group(1) + (m.group(2) or ''))[:60] if m else nm[:60]
t0 = sel[0]['s']
qs = collections.Counter(r['Queue_Id'] for r in sel)
main = qs.most_common()[0][0] if False else min(qs, key=lambda q: int(q))
for q in sorted(qs):
    rq = [r for r in sel if r['Queue_Id'] == q]
    busy = sum(r['e'] - r['s'] for r in rq) / 1e6
    span = (rq[-1]['e'] - rq[0]['s']) / 1e6
    gaps = collections.defaultdict(lambda: [0, 0.0])
    tot = 0.0
    big = []
    for a, b in zip(rq, rq[1:]):
        g = (b['s'] - a['e']) / 1e3
        if g > 0:
            tot += g
            gaps[short(b['Kernel_Name'])][0] += 1; gaps[short(b['Kernel_Name'])][1] += g
            if g > 60: big.append((round((a['e'] - t0) / 1e6, 2), round(g), short(a['Kernel_Name'])[:36], short(b['Kernel_Name'])[:36]))
    print(f"queue {q}: {len(rq)} kernels, busy {busy:.2f} ms, span {span:.2f} ms, idle inside span {tot / 1e3:.2f} ms")
    for name, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"     {t / 1e3:6.3f} ms in {c:4d} gaps (avg {t / c:6.1f} us) before {name}")
    for b in big[:25]:
        print("     big gap at", b)
print("step wall", (sel[-1]['e'] - sel[0]['s']) / 1e6, "ms")
