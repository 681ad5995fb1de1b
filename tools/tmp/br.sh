cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k grad_reduce 2>&1 | tail -3
export PROBE_REPS=2
for v in 0 1; do echo "== DIG_BATCH_REDUCE=$v"; DIG_BATCH_REDUCE=$v timeout 200 python tools/gpu_step_graph_probe.py 2>/dev/null | tail -5; done
