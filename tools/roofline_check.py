"""profiles/rNN_final_roofline_check.txt: dig_gemm_bf16 launches of the rocprofv3 kernel statistics grouped by family, against the
HIP-event figures bench.py prints (usage: roofline_check.py kernel_stats.csv bench_under_rocprof.json bench.json)."""
import csv, json, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
u, b = json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
fam = {"fwd": [0, 0.0], "dgrad": [0, 0.0], "wgrad": [0, 0.0]}
for r in rows:
    m = re.search(r"gemm_(?:wide_)?kernel<(true|false), (true|false), (\d)", r["Name"])
    if "gemm_pwide_kernel" in r["Name"]:
        ta, tb = False, False                                   # persistent forward tiles
    elif not m:
        continue
    else:
        ta, tb = m.group(1) == "true", m.group(2) == "true"
    k = "wgrad" if ta else ("dgrad" if tb else "fwd")
    fam[k][0] += int(r["Calls"]); fam[k][1] += float(r["TotalDurationNs"])
out = [f"rocprofv3 --kernel-trace --stats of `python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-mim-only --no-step-graph` ({sys.argv[1].split(chr(47))[-1]}):",
       "dig_gemm_bf16 launches grouped by family (template arguments TA, TB of gemm_kernel / gemm_wide_kernel), against the HIP-event figures",
       "bench.py prints in the same process (..._bench_under_rocprof.json) and in the unprofiled default run (..._final_bench.json).",
       "In the timed steps two HIP streams overlap, so a kernel's rocprof duration includes the slowdown from its neighbour on the other",
       "stream; the bench's event brackets are taken in two extra steps with the overlap off and are the lower figures.", ""]
out.append(f"{'family':8s} {'rocprof calls':>14s} {'rocprof avg us':>15s} | {'bench avg us (same process)':>28s} | {'bench avg us (unprofiled)':>26s}")
for k, (c, t) in fam.items():
    bu, bb = u["roofline"]["by_variant"][k], b["roofline"]["by_variant"][k]
    out.append(f"{k:8s} {c:14d} {t / c / 1e3:15.1f} | {bu['ms_per_step'] * 1e3 / bu['launches_per_step']:28.1f} | {bb['ms_per_step'] * 1e3 / bb['launches_per_step']:26.1f}")
out += ["", f"step: {b['ms_per_step']:.2f} ms unprofiled ({b['value']:.0f} images/s), {u['ms_per_step']:.2f} ms under rocprofv3; roofline.frac "
            f"{b['roofline']['frac']:.3f} ({b['roofline']['bound']}), hbm view {b['roofline']['hbm']['frac']:.3f}"]
dom = max(fam, key=lambda k: fam[k][1])
c, t = fam[dom]
fl = b["roofline"]["flops_per_launch"]
out.append(f"dominant family by rocprof time: {dom}; with the bench's {fl / 1e9:.1f} GFLOP per launch its rocprof average of {t / c / 1e3:.1f} us is "
           f"{fl / (t / c * 1e-9) / 1e12:.0f} TFLOP/s = {fl / (t / c * 1e-9) / 2.5e15:.3f} of the 2.5 PF bf16 roof (under stream overlap; the bench line's "
           f"{b['roofline']['frac']:.3f} is measured with the overlap off)")
print("\n".join(out))
