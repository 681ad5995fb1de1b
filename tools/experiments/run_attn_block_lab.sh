#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/experiments/run_attn_block_lab.sh r05 "0 1 2 4 8 16"'   -> gpurun_out/<tag>_attn_block_lab.txt
# (LAB_DEFS in the environment: extra -D flags, e.g. LAB_DEFS="-DDIG_AB_NSPLIT=4")
set -u
TAG=${1:-r05}
ABLS=${2:-"0"}
mkdir -p build/lab gpurun_out
OUT=gpurun_out/${TAG}_attn_block_lab.txt
: > $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -ffp-contract=fast -munsafe-fp-atomics -w -I include -I dig_amd/csrc"
SRC="tools/experiments/attn_block_lab.hip dig_amd/csrc/attn_block.hip"
fresh() { [ -x "$1" ] && [ -z "${LAB_DEFS:-}" ] && [ "$1" -nt tools/experiments/attn_block_lab.hip ] && [ "$1" -nt dig_amd/csrc/attn_block.hip ]; }   # (a binary built in the build container travels with the snapshot)
fresh build/lab/attn_block_lab_ts || hipcc $FLAGS -DLAB_TS ${LAB_DEFS:-} tools/experiments/attn_block_lab.hip -o build/lab/attn_block_lab_ts || exit 1
timeout 120 build/lab/attn_block_lab_ts >> $OUT 2>&1
for a in $ABLS; do
  fresh build/lab/attn_block_lab_$a || hipcc $FLAGS -DDIG_AB_ABL=$a ${LAB_DEFS:-} tools/experiments/attn_block_lab.hip -o build/lab/attn_block_lab_$a || exit 1
  timeout 120 build/lab/attn_block_lab_$a >> $OUT 2>&1
done
cat $OUT
