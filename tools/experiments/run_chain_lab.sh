#!/bin/bash
# gpurun --timeout 600 -- 'bash tools/experiments/run_chain_lab.sh r03 "0 1 2 4 8"'   -> gpurun_out/<tag>_chain_lab.txt
# CHAIN_TL=1 in the environment adds the per-tick timeline of one workgroup.
set -u
TAG=${1:-r03}
ABLS=${2:-"0"}
mkdir -p build/lab gpurun_out
: > gpurun_out/${TAG}_chain_lab.txt
for a in $ABLS; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w -I include -I dig_amd/csrc -DDIG_CHAIN_ABL=$a ${CHAIN_DEFS:-} tools/experiments/chain_lab.hip dig_amd/csrc/probe.hip -o build/lab/chain_lab_$a || exit 1
  timeout 120 build/lab/chain_lab_$a >> gpurun_out/${TAG}_chain_lab.txt 2>&1
done
cat gpurun_out/${TAG}_chain_lab.txt
