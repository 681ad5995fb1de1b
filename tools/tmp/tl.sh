cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_tl
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tl -- python $R/bench.py --steps 16 --warmup 6 --no-cpu-baseline --no-mim-only --no-step-graph > /dev/null 2>&1
f=$(ls $R/gpurun_out/prof_tl/*/*kernel_trace.csv | head -1)
python $R/tools/trace_timeline.py $f 12 full > $R/gpurun_out/r02_timeline_step.txt
head -30 $R/gpurun_out/r02_timeline_step.txt
rm -rf $R/gpurun_out/prof_tl
