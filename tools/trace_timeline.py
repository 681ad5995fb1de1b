"""Per-queue timeline summary of one step from a rocprofv3 kernel trace CSV (usage: trace_timeline.py <kernel_trace.csv> [step])."""
import csv, collections, re, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ad = [i for i, r in enumerate(rows) if 'adamw_kernel' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
step = rows[ad[k] + 1:ad[k + 1] + 1]
t0, t1 = rows[ad[k]]['e'], step[-1]['e']
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'\(.*', '', n); return n.replace('void ', '')[:48]
print("step wall ms", (t1 - t0) / 1e6, "kernels", len(step))
byq = collections.defaultdict(list)
for r in step: byq[r['Queue_Id']].append(r)
for q, l in sorted(byq.items()):
    busy = sum(r['e'] - r['s'] for r in l)
    gaps = [(l[i]['s'] - l[i - 1]['e'], i) for i in range(1, len(l))]
    print(f"queue {q}: n {len(l)} busy {busy/1e6:.2f} ms span {(l[0]['s']-t0)/1e6:.2f}..{(l[-1]['e']-t0)/1e6:.2f} gaps {sum(max(g,0) for g,_ in gaps)/1e6:.2f} ms")
    for g, i in sorted(gaps, reverse=True)[:8]:
        print(f"      gap {g/1e3:8.1f} us at {(l[i]['s']-t0)/1e6:6.2f} ms  {short(l[i-1]['Kernel_Name'])} -> {short(l[i]['Kernel_Name'])}")
ev = sorted((r['s'], r['e']) for r in step)
cs, ce = ev[0]; tot = 0
for s, e in ev[1:]:
    if s > ce: tot += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
print("union busy ms", (tot + ce - cs) / 1e6)
if len(sys.argv) > 3:
    for r in step:
        print(f"{r['Queue_Id']} {(r['s']-t0)/1e3:9.1f} {(r['e']-r['s'])/1e3:7.1f} {short(r['Kernel_Name'])}")
