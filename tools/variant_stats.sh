#!/bin/bash
# rocprofv3 kernel statistics of one bench.py variant, per step:   bash tools/variant_stats.sh <tag> [bench.py flags ...]
#   -> gpurun_out/<tag>_kernel_stats.txt (top 30 kernels: launches per step x average us = ms per step)
set -u
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python $ROOT/bench.py --steps 10 --warmup 4 --no-step-graph --no-mim-only --no-cpu-baseline "$@" > $OUT/${TAG}_bench_under_rocprof.json 2> /dev/null
cd $ROOT
python - "$(ls $OUT/prof_$TAG/*/*kernel_stats.csv | head -1)" > $OUT/${TAG}_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = sum(int(r["Calls"]) for r in rows if "adamw" in r["Name"])
tot = sum(int(r["TotalDurationNs"]) for r in rows) / steps / 1e6
print(f"{steps} steps in the trace; kernel time per step {tot:.2f} ms (sum over both queues)")
for r in rows[:30]:
    print(f"{int(r['Calls']) / steps:6.1f} x {float(r['AverageNs']) / 1e3:8.1f} us = {int(r['TotalDurationNs']) / steps / 1e6:6.3f} ms  {r['Name'][:110]}")
PY
rm -rf $OUT/prof_$TAG
cat $OUT/${TAG}_kernel_stats.txt
