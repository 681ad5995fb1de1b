"""In-situ sweep of the R-split count of the tall weight-gradient GEMMs (monkeypatches ops.wgrad_splits; prints ms per step)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = '''
import sys, io, json, contextlib
sys.path.insert(0, "{root}")
from dig_amd import ops
want, small = {want}, {small}
orig = ops.wgrad_splits
def patched(rows, tiles):
    sp, bk = orig(rows, tiles)
    if want and rows >= 32768 and tiles >= 24:
        return ops.L.lib().dig_gemm_effective_splits(rows, want), bk
    if small and rows >= 32768 and tiles < 24:
        return ops.L.lib().dig_gemm_effective_splits(rows, small), bk
    return sp, bk
ops.wgrad_splits = patched
sys.argv = ["bench.py", "--steps", "30", "--warmup", "10", "--no-cpu-baseline", "--no-mim-only"]
import runpy
runpy.run_path("{root}/bench.py", run_name="__main__")
'''
for want, small in ((16, 0), (8, 0), (16, 24), (16, 16), (16, 56), (16, 0), (12, 0)):
    out = subprocess.run([sys.executable, "-c", code.format(root=ROOT, want=want, small=small)], capture_output=True, text=True).stdout.strip().splitlines()
    d = json.loads(out[-1])
    print("splits big", want or "default(24)", "small", small or "default(40)", round(d["ms_per_step"], 3), flush=True)
