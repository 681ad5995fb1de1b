cd $GRAFT_REPO_ROOT
F=ffffffff
run() { echo "== side mask: $1"; DIG_SIDE_CU_MASK=$2 timeout 200 python bench.py --no-cpu-baseline --no-mim-only --no-step-graph --steps 30 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "== baseline (priority -1, all CUs)"; timeout 200 python bench.py --no-cpu-baseline --no-mim-only --no-step-graph --steps 30 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
run "all 256 (external stream, normal priority)" $F,$F,$F,$F,$F,$F,$F,$F
run "low 128" $F,$F,$F,$F
run "low 192" $F,$F,$F,$F,$F,$F
run "every other CU (128)" 55555555,55555555,55555555,55555555,55555555,55555555,55555555,55555555
run "3 of 4 CUs (192)" 77777777,77777777,77777777,77777777,77777777,77777777,77777777,77777777
echo "== baseline again"; timeout 200 python bench.py --no-cpu-baseline --no-mim-only --no-step-graph --steps 30 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
