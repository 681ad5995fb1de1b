#!/bin/bash
# The step of another REVISION (whole tree: Python + kernel library, built by `tree_ab.sh build <rev>` in the build container) against the working
# tree's, alternating on ONE box:
#   bash tools/experiments/tree_ab.sh build 7e6bbcb                       (build container: build/ab/tree_<rev>/ with its own libdig_hip.so)
#   gpurun --timeout 1500 -- 'bash tools/experiments/tree_ab.sh 7e6bbcb [bench.py flags]'   -> gpurun_out/tree_ab_<rev>.txt
set -u
if [ "$1" = build ]; then
  REV=$2; DST=build/ab/tree_$REV
  rm -rf $DST && mkdir -p $DST && git archive $REV | tar -x -C $DST
  (cd $DST && python -c "import sys; sys.path.insert(0, '.'); from dig_amd import build; build.build(verbose=False)") && ls -la $DST/dig_amd/lib/libdig_hip.so
  exit
fi
REV=$1; shift
mkdir -p gpurun_out
OUT=$(pwd)/gpurun_out/tree_ab_$REV.txt
[ -n "${APPEND:-}" ] || : > $OUT
run() {
  (cd $1 && python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-step-graph --no-mim-only "${@:3}" 2>/dev/null) | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2', round(d['ms_per_step'], 3), 'ms', round(d['value'], 1), 'images/s')" >> $OUT
}
echo "# bench.py $*" >> $OUT
for i in 1 2 3; do
  run build/ab/tree_$REV "A $REV   " "$@"
  run . "B worktree" "$@"
done
cat $OUT
