cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
rm -rf $R/gpurun_out/prof_bq
DIG_BATCH_REDUCE=$v rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_bq -- python $R/bench.py --steps 16 --warmup 6 --no-cpu-baseline --no-mim-only --no-step-graph > /dev/null 2>&1
f=$(ls $R/gpurun_out/prof_bq/*/*kernel_trace.csv | head -1)
echo "== DIG_BATCH_REDUCE=$v"
python $R/tools/trace_by_queue.py $f 8 8 70 | grep -i "step wall\|reduce\|finalize\|colsum"
done
rm -rf $R/gpurun_out/prof_bq
