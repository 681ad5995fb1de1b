#!/bin/bash
# gpurun --timeout 1500 -- 'bash tools/experiments/knob_sweep.sh'  -> gpurun_out/knob_sweep.txt : ms per step of the default plan and of single-knob variants, one box
mkdir -p gpurun_out
OUT=gpurun_out/knob_sweep.txt
: > $OUT
run() {
  local tag="$1"; shift
  local ms=$(env "$@" python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-mim-only --no-step-graph 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$tag $ms" | tee -a $OUT
}
run default X=1
run DIG_WGRAD_WA=1 DIG_WGRAD_WA=1
run DIG_WGRAD_WA=3 DIG_WGRAD_WA=3
run default X=1
run DIG_FWD_MOM_PRIO=normal DIG_FWD_MOM_PRIO=normal
run DIG_BWD_SIDE_PRIO=normal DIG_BWD_SIDE_PRIO=normal
run DIG_BATCH_REDUCE=1 DIG_BATCH_REDUCE=1
run default X=1
run DIG_WGRAD_DEFER=0 DIG_WGRAD_DEFER=0
run DIG_BLOCK_CALLS=0 DIG_BLOCK_CALLS=0
run DIG_DGRAD_BK=0 DIG_DGRAD_BK=0
run default X=1
