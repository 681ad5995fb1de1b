#!/bin/bash
# The step with the kernel library of another revision (A, built by build_rev_lib.sh) and the working tree's (B), alternating on ONE box:
#   gpurun --timeout 1200 -- 'bash tools/experiments/lib_ab.sh HEAD [bench.py flags]'   -> gpurun_out/lib_ab.txt
set -u
REV=${1:-HEAD}; shift
mkdir -p gpurun_out
OUT=gpurun_out/lib_ab.txt
: > $OUT
cp dig_amd/lib/libdig_hip.so build/ab/libdig_hip_new.so
run() {
  python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-step-graph --no-mim-only "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
bv = d['roofline']['by_variant']
print('$TAG', round(d['ms_per_step'], 3), 'ms', round(d['value'], 1), 'images/s |', ' '.join(f\"{k} {v['avg_launch_us']:.1f}\" for k, v in bv.items() if v['ms_per_step'] > 0.3))"
}
for i in 1 2 3; do
  cp build/ab/$REV/dig_amd/lib/libdig_hip.so dig_amd/lib/libdig_hip.so; TAG="A $REV   " run "$@" >> $OUT
  cp build/ab/libdig_hip_new.so dig_amd/lib/libdig_hip.so; TAG="B worktree" run "$@" >> $OUT
done
cat $OUT
