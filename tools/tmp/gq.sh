cd $GRAFT_REPO_ROOT
export PROBE_REPS=2
for sg in 0 2 1; do
for q in 2 3; do echo "== SIDE_GROUP=$sg QUEUES=$q"; DIG_SIDE_GROUP=$sg DEBUG_HIP_FORCE_GRAPH_QUEUES=$q timeout 200 python tools/gpu_step_graph_probe.py 2>/dev/null | tail -4; done
done
