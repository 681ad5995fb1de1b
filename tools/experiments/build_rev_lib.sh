#!/bin/bash
# The kernel library of another revision beside the working tree's, for A/B runs of the step on ONE box (build container):
#   bash tools/experiments/build_rev_lib.sh HEAD        -> build/ab/HEAD/dig_amd/lib/libdig_hip.so   (travels to the GPU box with the snapshot)
# The A/B scripts copy it over dig_amd/lib/libdig_hip.so of the box's scratch copy between runs (tools/experiments/lib_ab.sh).
set -eu
REV=${1:-HEAD}
DST=build/ab/$REV
rm -rf $DST && mkdir -p $DST
git archive $REV dig_amd include | tar -x -C $DST
(cd $DST && python -c "import sys; sys.path.insert(0, '.'); from dig_amd import build; build.build(verbose=False)")
ls -la $DST/dig_amd/lib/libdig_hip.so
