#!/bin/bash
# Phase timeline of the attention backward for a list of compile-time variants (each a quoted set of -D flags), product compile flags:
#   bash tools/experiments/run_attn_bwd_lab.sh build "" "-DDIG_ATTN_B_PIPE=1"          # in the build container: compile only (the binaries travel)
#   gpurun --timeout 600 -- 'bash tools/experiments/run_attn_bwd_lab.sh r06 "" "-DDIG_ATTN_B_PIPE=1"'   -> gpurun_out/r06_attn_bwd_lab.txt
set -u
TAG=${1:-r06}; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -ffp-contract=fast -munsafe-fp-atomics -w -I include -I dig_amd/csrc"
mkdir -p build/lab gpurun_out
OUT=gpurun_out/${TAG}_attn_bwd_lab.txt
[ "$TAG" = build ] || : > $OUT
for defs in "$@"; do
  name=build/lab/attn_bwd_lab_$(echo "x$defs" | md5sum | cut -c1-8)
  if [ ! -x $name ] || [ $name -ot dig_amd/csrc/attention.hip ] || [ $name -ot tools/experiments/attn_bwd_lab.hip ]; then
    hipcc $FLAGS $defs tools/experiments/attn_bwd_lab.hip dig_amd/csrc/probe.hip -o $name || exit 1
  fi
  [ "$TAG" = build ] && continue
  echo "== variant: ${defs:-product}" >> $OUT
  timeout 120 $name >> $OUT 2>&1
done
[ "$TAG" = build ] || cat $OUT
