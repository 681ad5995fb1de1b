cd $GRAFT_REPO_ROOT
python -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)
for p in (-2,-1,0,1,2):
    s=torch.cuda.Stream(priority=p); print(p, s.priority)
"
for p in -1 1 0 -1 1; do echo "== side priority $p"; DIG_SIDE_PRIORITY=$p timeout 200 python bench.py --no-cpu-baseline --no-mim-only --no-step-graph --steps 30 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
