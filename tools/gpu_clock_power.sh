#!/bin/bash
# Shader clock and package power while the step runs (gpurun -- 'bash tools/gpu_clock_power.sh'): rocm-smi is polled every 1.5 s beside
# `python bench.py --steps 1500`; the idle samples before / after the run are kept for contrast.  -> gpurun_out/clock_power.txt
mkdir -p gpurun_out
OUT=gpurun_out/clock_power.txt
: > $OUT
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | sed 's/GPU\[0\]\s*: //' >> $OUT
python bench.py --steps 1500 --warmup 20 --no-cpu-baseline --no-mim-only --no-step-graph > /tmp/b.json 2>/dev/null &
BP=$!
SECONDS=0
while kill -0 $BP 2>/dev/null; do
  echo "t=${SECONDS}s $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\|Power (W)' | sed 's/GPU\[0\]\s*: //' | tr '\n' ' ')" >> $OUT
  sleep 1.5
done
python - <<'PY' >> gpurun_out/clock_power.txt
import json
d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print(f"bench: {d['ms_per_step']:.2f} ms per step, {d['value']:.0f} images/s over {d['steps']} steps")
PY
cat $OUT
