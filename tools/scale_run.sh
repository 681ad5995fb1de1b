#!/bin/bash
# Scaling series of the pre-training step on ONE node: N = 1, 2, 4, 8 ranks (one process per GPU, RCCL over xGMI), the same
# commands the driver uses.  Writes one JSON line per N into gpurun_out/scale_<N>.json and checks that the process group
# really had N ranks on N distinct devices (NCCL_DEBUG=INFO banner + the line's n_gpus / parallelism fields).
#   bash tools/scale_run.sh [steps] [warmup]
set -u
STEPS=${1:-20}
WARMUP=${2:-5}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
echo "visible GPUs: $NGPU"
for N in 1 2 4 8; do
  if [ "$N" -gt "$NGPU" ]; then echo "N=$N: skipped (only $NGPU GPUs visible)"; continue; fi
  if [ "$N" -eq 1 ]; then
    python $ROOT/bench.py --gpus 1 --steps $STEPS --warmup $WARMUP --no-cpu-baseline > $OUT/scale_$N.json 2> $OUT/scale_$N.err
  else
    # the plain command: bench.py starts its own N ranks under torch.distributed.run (RCCL, one device per rank)
    NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT python $ROOT/bench.py --gpus $N --steps $STEPS --warmup $WARMUP --no-cpu-baseline > $OUT/scale_$N.json 2> $OUT/scale_$N.err
    ranks=$(grep -c "Init COMPLETE\|init complete\|comm .* rank" $OUT/scale_$N.err || true)
    echo "N=$N: RCCL init lines in the log: $ranks"
  fi
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/scale_$N.json").read().strip().splitlines()[-1])
    assert d["n_gpus"] == $N and d["config"]["parallelism"] == "dp$N", (d["n_gpus"], d["config"]["parallelism"])
    print(f"N=$N: {d['value']:.0f} images/s, {d['ms_per_step']:.2f} ms/step, per GPU {d['value'] / $N:.0f}")
except Exception as e:
    print("N=$N: no valid JSON line:", e)
PY
done
