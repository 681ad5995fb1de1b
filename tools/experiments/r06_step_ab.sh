#!/bin/bash
# Round 6: the step with this round's two plan changes off / on, alternating on ONE box (boxes of the pool differ by 3 %):
#   A  DIG_ADAMW_FOLD=0 DIG_DGRAD_DIRECT=0 DIG_BN_FUSED=0   the round-5 plan: cast + transposes at the head of every forward, transpose-read data
#                                                          gradients, three launches per BatchNorm layer
#   B  DIG_DGRAD_DIRECT=0 DIG_BN_FUSED=0    the optimizer launch leaves the bf16 shadow and the transposed weight copies
#   C  DIG_BN_FUSED=0                       ... and the proj / qkv data gradients run in their direct form on those copies
#   D  (default)                            ... and the heads' few-row BatchNorm layers are one launch each
#   gpurun --timeout 1200 -- 'bash tools/experiments/r06_step_ab.sh'   -> gpurun_out/r06_step_ab.txt
set -u
mkdir -p gpurun_out
OUT=gpurun_out/r06_step_ab.txt
: > $OUT
run() {
  python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-step-graph --no-mim-only 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['ms_per_step'], 3), 'ms', round(d['value'], 1), 'images/s')"
}
for i in 1 2 3; do
  DIG_ADAMW_FOLD=0 DIG_DGRAD_DIRECT=0 DIG_BN_FUSED=0 run "A round-5 plan         " >> $OUT
  DIG_DGRAD_DIRECT=0 DIG_BN_FUSED=0 run "B + optimizer fold      " >> $OUT
  DIG_BN_FUSED=0 run "C + direct-form dgrads  " >> $OUT
  run "D + fused few-row BN    " >> $OUT
done
cat $OUT
