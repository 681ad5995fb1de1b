set -u
bash tools/collect_profiles.sh r06 > gpurun_out/collect.log 2>&1
tail -12 gpurun_out/collect.log
B="--no-cpu-baseline --no-step-graph --no-mim-only"
python bench.py --model base --batch 256 --steps 30 --warmup 8 $B 2>/dev/null | tail -1 > gpurun_out/r06_base_b256_bench.json
python bench.py --model-kind simmim --steps 60 --warmup 10 $B 2>/dev/null | tail -1 > gpurun_out/r06_gen_only_bench.json
python bench.py --model-kind moco --steps 60 --warmup 10 $B 2>/dev/null | tail -1 > gpurun_out/r06_dis_only_bench.json
python bench.py --num-windows 5 --patchnet-name regular --steps 60 --warmup 10 $B 2>/dev/null | tail -1 > gpurun_out/r06_cli_defaults_bench.json
python bench_finetune.py 2>/dev/null | tail -1 > gpurun_out/r06_finetune_bench_tf_decoder.json
bash tools/variant_stats.sh r06_base --model base --batch 256 > /dev/null 2>&1
for f in base_b256 gen_only dis_only cli_defaults finetune_bench_tf_decoder; do python - gpurun_out/r06_${f}*.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d['value'],1), round(d['ms_per_step'],3))
PY
done
head -14 gpurun_out/r06_base_kernel_stats.txt | cut -c1-150
