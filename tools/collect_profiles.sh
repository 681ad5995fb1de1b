#!/bin/bash
# Collects the round's measurement files on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r02
# -> gpurun_out/<tag>_final_bench.json            python bench.py (the driver's command, default flags)
#    gpurun_out/<tag>_final_kernel_stats.csv      rocprofv3 --kernel-trace --stats of bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-mim-only --no-step-graph
#    gpurun_out/<tag>_final_bench_under_rocprof.json
#    gpurun_out/<tag>_final_by_queue.txt / _timeline.txt / _gaps.txt   the same kernel trace by queue, as a timeline of one step, idle gaps
#    gpurun_out/<tag>_pmc_traffic.json            FETCH_SIZE / WRITE_SIZE passes (separate), tagged with the kernel-source hash
#    gpurun_out/<tag>_pmc_kernel_counters.txt     MFMA / LDS utilisation, occupancy, instruction mix passes
# Copy what should be judged into profiles/ afterwards.
set -u
TAG=${1:-r05}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python $ROOT/bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-mim-only --no-step-graph > $OUT/${TAG}_final_bench_under_rocprof.json 2> /dev/null
cp $(ls $OUT/prof_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_final_kernel_stats.csv
f=$(ls $OUT/prof_stats/*/*kernel_trace.csv | head -1)
python $ROOT/tools/trace_by_queue.py $f 10 10 60 > $OUT/${TAG}_final_by_queue.txt
python $ROOT/tools/trace_timeline.py $f 12 full > $OUT/${TAG}_final_timeline.txt
python $ROOT/tools/trace_gaps.py $f 12 8 > $OUT/${TAG}_final_gaps.txt
rm -rf $OUT/prof_stats
SHORT="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-mim-only --no-step-graph"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/prof_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/prof_$c -- $SHORT > /dev/null 2>&1
done
python $ROOT/tools/pmc_traffic.py $(ls $OUT/prof_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls $OUT/prof_WRITE_SIZE/*/*counter_collection.csv | head -1) $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_traffic.txt
python - <<PY
import hashlib, json
p = "$OUT/${TAG}_pmc_traffic.json"
d = json.load(open(p))
d["steps_in_run"] = d["kernels"]["adamw"]["launches"]      # one AdamW launch per step: warm-up + timed + the roofline probe's extra steps
d["lib_sha256_16"] = hashlib.sha256(open("$ROOT/dig_amd/lib/libdig_hip.so", "rb").read()).hexdigest()[:16]
import sys
sys.path.insert(0, "$ROOT")
from dig_amd import build
d["src_sha256_16"] = build.source_hash()          # what bench.py re-checks: the .so bytes differ between builds of identical sources
json.dump(d, open(p, "w"), indent=1)
PY
rm -rf $OUT/prof_FETCH_SIZE $OUT/prof_WRITE_SIZE
# the default bench run comes AFTER the traffic passes: its roofline.traffic is read from profiles/<tag>_pmc_traffic.json (hash-gated)
cp $OUT/${TAG}_pmc_traffic.json $ROOT/profiles/${TAG}_pmc_traffic.json
python $ROOT/bench.py > $OUT/${TAG}_final_bench.json 2> $OUT/${TAG}_final_bench.err
: > $OUT/${TAG}_pmc_kernel_counters.txt
for set in "MfmaUtil LdsUtil" "VmemLatency OccupancyPercent" "MemUnitStalled" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL"; do
  rm -rf $OUT/prof_pmc
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/prof_pmc -- $SHORT > /dev/null 2>&1
  echo "== $set" >> $OUT/${TAG}_pmc_kernel_counters.txt
  python $ROOT/tools/pmc_summary.py $(ls $OUT/prof_pmc/*/*counter_collection.csv | head -1) | grep -E "gemm|attn|ln_|reduce_partials|mlp_chain|wgrad" >> $OUT/${TAG}_pmc_kernel_counters.txt
done
rm -rf $OUT/prof_pmc
cd $ROOT
python tools/roofline_check.py $OUT/${TAG}_final_kernel_stats.csv $OUT/${TAG}_final_bench_under_rocprof.json $OUT/${TAG}_final_bench.json > $OUT/${TAG}_final_roofline_check.txt
cat $OUT/${TAG}_final_roofline_check.txt
