"""Per-step summary of a rocprofv3 kernel_stats CSV (usage: stats_summary.py <kernel_stats.csv> <steps_incl_warmup_and_probe>)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 27.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"sum of kernel time per step: {tot / n / 1e6:.2f} ms")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:78]
    print(f"{float(r['TotalDurationNs']) / n / 1e6:7.3f} ms {int(r['Calls']) / n:6.1f} x {float(r['AverageNs']) / 1e3:7.1f} us  {name}")
