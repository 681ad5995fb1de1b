#!/bin/bash
# Round 6 (second session): how the attention backward's results leave, in the step, alternating on ONE box:
#   DIG_ATTN_BWD_STORE=0   16-byte row stores, default cache policy (the round-5 form)
#   DIG_ATTN_BWD_STORE=1   the same stores, non-temporal
#   DIG_ATTN_BWD_STORE=3   full 128-byte lines through 2 KiB of LDS per wave, non-temporal (the default)
#   gpurun --timeout 1200 -- 'bash tools/experiments/r06_attn_store_ab.sh [extra bench.py flags]'   -> gpurun_out/r06_attn_store_ab.txt
set -u
mkdir -p gpurun_out
OUT=gpurun_out/r06_attn_store_ab.txt
: > $OUT
run() {
  python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-step-graph --no-mim-only "${EXTRA[@]}" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']['by_variant'].get('attn_bwd', {})
print('$1', round(d['ms_per_step'], 3), 'ms', round(d['value'], 1), 'images/s; attn_bwd in the step', round(r.get('avg_launch_us', 0), 1), 'us')"
}
EXTRA=("$@")
for i in 1 2 3; do
  DIG_ATTN_BWD_STORE=0 run "0 row stores          " >> $OUT
  DIG_ATTN_BWD_STORE=1 run "1 row stores, nt      " >> $OUT
  DIG_ATTN_BWD_STORE=3 run "3 line stores, nt     " >> $OUT
done
cat $OUT
