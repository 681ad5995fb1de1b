"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel: mean of each counter over the dispatches of a kernel."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'\(.*', '', n); return n.replace('void ', '')[:62]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'at::' in k or 'rocclr' in k: continue
    print(f"{k:64s}", "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())))
