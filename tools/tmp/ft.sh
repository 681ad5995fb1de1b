cd $GRAFT_REPO_ROOT/tools/tmp/old && python -m dig_amd.build > /dev/null 2>&1; ls -la dig_amd/lib/*.so | head -2
cd $GRAFT_REPO_ROOT
f() { timeout 300 python bench_finetune.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
echo "== HEAD"; (cd $GRAFT_REPO_ROOT && f)
echo "== 3b5a1f6 (side stream priority commit)"; (cd $GRAFT_REPO_ROOT/tools/tmp/old && f)
done
