#!/bin/bash
# Round 6: the online fused MLP launch with the S-waves bringing all ring pieces (DIG_CHAIN_SDMA=1) against every wave bringing its
# share (=0): the chain lab's wall time and per-wave accounting per mode.   gpurun --timeout 900 -- 'bash tools/experiments/r06_sdma_ab.sh'
set -u
mkdir -p gpurun_out
for v in 0 1; do
  CHAIN_DEFS="-DDIG_CHAIN_SDMA=$v" bash tools/experiments/run_chain_lab.sh r06_sdma$v "0" > /dev/null 2>&1
  echo "== SDMA $v"; grep -h -A2 "mode" gpurun_out/r06_sdma${v}_chain_lab.txt | grep -v "^--" | cut -c1-220
done
