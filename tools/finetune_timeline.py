import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ad = [i for i, r in enumerate(rows) if 'adamw_groups_kernel' in r['Kernel_Name']]
k = 8
step = rows[ad[k] + 1:ad[k + 1] + 1]
t0, t1 = rows[ad[k]]['e'], step[-1]['e']
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'\(.*', '', n); return n.replace('void ', '')[:44]
print("step wall ms", (t1 - t0) / 1e6, "kernels", len(step))
byq = collections.defaultdict(list)
for r in step: byq[r['Queue_Id']].append(r)
for q, l in sorted(byq.items()):
    busy = sum(r['e'] - r['s'] for r in l)
    agg = collections.Counter()
    for r in l: agg[short(r['Kernel_Name'])] += r['e'] - r['s']
    print(f"queue {q}: n {len(l)} busy {busy/1e6:.2f} ms span {(l[0]['s']-t0)/1e6:.2f}..{(l[-1]['e']-t0)/1e6:.2f}")
    for n, v in agg.most_common(12): print(f"     {v/1e6:6.2f} ms {n}")
