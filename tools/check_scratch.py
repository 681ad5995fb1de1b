"""Build-time check: no hot kernel of libdig_hip.so may use scratch memory (private_segment_fixed_size > 0 = spilled registers or stack arrays:
every spill is a round trip through the scratch aperture inside a loop that is priced in MFMA slots).

    python tools/check_scratch.py            # table of every kernel with scratch; exit 1 if a HOT kernel has any
    python tools/check_scratch.py --all      # ... the whole table (vgprs, sgprs, scratch bytes)

Reads the gfx950 code objects out of dig_amd/lib/obj/*.o (llvm-objcopy --dump-section .hip_fatbin, clang-offload-bundler --unbundle,
llvm-readelf --notes); `dig_amd.build.check_scratch()` is the same check as a function (tests/test_abi_symbols.py runs it)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from dig_amd import build
    rows = build.kernel_resources()
    bad = build.check_scratch(rows, raise_on_fail=False)
    show = rows if "--all" in sys.argv else [r for r in rows if r["scratch"] > 0]
    for r in sorted(show, key=lambda r: (-r["scratch"], r["name"])):
        print(f"{r['scratch']:6d} B scratch  {r['vgpr']:4d} vgpr {r['sgpr']:4d} sgpr  {'HOT ' if r['hot'] else '    '}{r['name'][:150]}")
    if bad:
        print("FAIL: hot kernels with scratch:", *[b["name"] for b in bad], sep="\n  ")
        sys.exit(1)
    print(f"OK: {sum(1 for r in rows if r['hot'])} hot kernels, none uses scratch ({len(rows)} kernels in the library)")
