cd $GRAFT_REPO_ROOT
b() { timeout 200 python bench.py --no-cpu-baseline --no-mim-only --no-step-graph --steps 30 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "== main 0 side -1 (current)"; b
echo "== main -1 side 0"; DIG_MAIN_PRIORITY=-1 DIG_SIDE_PRIORITY=0 b
echo "== main -1 side -1"; DIG_MAIN_PRIORITY=-1 DIG_SIDE_PRIORITY=-1 b
echo "== main 0 side -1 (current)"; b
echo "== main -1 side 0"; DIG_MAIN_PRIORITY=-1 DIG_SIDE_PRIORITY=0 b
