#!/bin/bash
# The step with one environment switch off / on, alternating on ONE box:
#   gpurun --timeout 1200 -- 'bash tools/experiments/env_ab.sh DIG_ATTN_BWD_PROJ 0 1 [bench.py flags]'   -> gpurun_out/env_ab_<name>.txt
set -u
NAME=$1; A=$2; B=$3; shift 3
mkdir -p gpurun_out
OUT=gpurun_out/env_ab_$NAME.txt
: > $OUT
run() {
  python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-step-graph --no-mim-only "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
bv = d['roofline']['by_variant']
print('$TAG', round(d['ms_per_step'], 3), 'ms', round(d['value'], 1), 'images/s |', ' '.join(f\"{k} {v['avg_launch_us']:.1f}\" for k, v in bv.items() if v['ms_per_step'] > 0.3))"
}
for i in 1 2 3; do
  export $NAME=$A; TAG="$NAME=$A" run "$@" >> $OUT
  export $NAME=$B; TAG="$NAME=$B" run "$@" >> $OUT
done
cat $OUT
