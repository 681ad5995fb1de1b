#!/bin/bash
# bench_finetune.py with the kernel library of another revision (A) and the working tree's (B), alternating on one box -> gpurun_out/lib_ab_finetune.txt
set -u
REV=${1:-HEAD}; shift
mkdir -p gpurun_out
OUT=gpurun_out/lib_ab_finetune.txt
: > $OUT
cp dig_amd/lib/libdig_hip.so build/ab/libdig_hip_new.so
run() { python bench_finetune.py "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$TAG', round(d['ms_per_step'], 3), 'ms', round(d['value'], 1), 'images/s')"; }
for i in 1 2; do
  cp build/ab/$REV/dig_amd/lib/libdig_hip.so dig_amd/lib/libdig_hip.so; TAG="A $REV   " run "$@" >> $OUT
  cp build/ab/libdig_hip_new.so dig_amd/lib/libdig_hip.so; TAG="B worktree" run "$@" >> $OUT
done
cat $OUT
