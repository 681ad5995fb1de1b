#!/bin/bash
# Builds the three round-2 lab harnesses (hipcc, no Python) and runs them on the GPU box:
#   gpurun --timeout 600 -- 'bash tools/experiments/run_labs.sh r02'
# -> gpurun_out/<tag>_gemm_lab.txt, <tag>_attn_bwd_lab.txt, <tag>_dma_rate_lab.txt (copy into profiles/ to keep them).
set -u
TAG=${1:-r02}
mkdir -p build/lab gpurun_out
for lab in gemm_lab attn_bwd_lab dma_rate_lab; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I include -I dig_amd/csrc tools/experiments/$lab.hip -o build/lab/$lab || exit 1
  timeout 300 build/lab/$lab > gpurun_out/${TAG}_$lab.txt 2>&1
  tail -5 gpurun_out/${TAG}_$lab.txt
done
