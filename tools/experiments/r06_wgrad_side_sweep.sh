mkdir -p gpurun_out
OUT=gpurun_out/r06_wgrad_side_sweep.txt
: > $OUT
run() {
  local tag="$1"; shift
  local ms=$(env "$@" python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-mim-only --no-step-graph 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "$tag $ms" | tee -a $OUT
}
run default X=1
run "inline0 defer0 side=high" DIG_WGRAD_INLINE=0 DIG_WGRAD_DEFER=0
run "inline0 defer0 side=normal" DIG_WGRAD_INLINE=0 DIG_WGRAD_DEFER=0 DIG_BWD_SIDE_PRIO=normal
run "inline0 defer0 side=low" DIG_WGRAD_INLINE=0 DIG_WGRAD_DEFER=0 DIG_BWD_SIDE_PRIO=low
run default X=1
run "inline0 defer0 side=low blockcalls0" DIG_WGRAD_INLINE=0 DIG_WGRAD_DEFER=0 DIG_BWD_SIDE_PRIO=low DIG_BLOCK_CALLS=0
run "blockcalls0" DIG_BLOCK_CALLS=0
