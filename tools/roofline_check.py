"""profiles/rNN_final_roofline_check.txt: the matrix-core launches of the rocprofv3 kernel statistics grouped by family, against the
HIP-event figures bench.py prints for the same families (usage: roofline_check.py kernel_stats.csv bench_under_rocprof.json bench.json).
bench.py's `roofline.frac` is the dominant family's FLOP over its IN-STEP launch durations (both streams running), which is what the
rocprofv3 average of the same command measures; `roofline.alone` is the same launches with the stream overlap off."""
import csv, json, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
u, b = json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))


def family(name):
    """Kernel name (rocprofv3) -> the family key bench.py's probe uses."""
    import re
    if "mlp_chain_kernel<2" in name:
        return "mlp_chain_bwd"
    if "mlp_chain_kernel<1" in name:
        return "mlp_chain_online"
    if "mlp_chain_kernel" in name:
        return "mlp_chain_momentum"
    if "wgrad_wide_kernel" in name or "wgrad_group_kernel" in name:
        return "wgrad_group"
    if "attn_block_kernel<true" in name:
        return "attn_block_online"
    if "attn_block_kernel" in name:
        return "attn_block_momentum"
    if re.search(r"attn_bwd_kernel<(true|false), \d, true>", name):
        return "attn_bwd_proj"
    if "attn_bwd" in name:
        return "attn_bwd"
    if "attn_fwd" in name:
        return "attn_fwd"
    m = re.search(r"gemm_pwide_kernel<(\d)", name)
    if m:
        return "fwd:" + {"4": "544", "3": "564"}.get(m.group(1), "5xx")
    m = re.search(r"gemm_wide_kernel<(true|false), (true|false), \d, (\d), (\d)", name)
    if m:
        form = "wgrad" if m.group(1) == "true" else ("dgrad" if m.group(2) == "true" else "fwd")
        return form + ":" + {"44": "244", "43": "264", "12": "212", "21": "221"}.get(m.group(3) + m.group(4), "2xx")
    m = re.search(r"gemm_kernel<(true|false), (true|false), \d, (\d+)", name)
    if m:
        form = "wgrad" if m.group(1) == "true" else ("dgrad" if m.group(2) == "true" else "fwd")
        return form + ":" + m.group(3)
    return None


fam = {}
for r in rows:
    k = family(r["Name"])
    if k:
        d = fam.setdefault(k, [0, 0.0])
        d[0] += int(r["Calls"]); d[1] += float(r["TotalDurationNs"])
out = [f"rocprofv3 --kernel-trace --stats of `python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-mim-only --no-step-graph` ({sys.argv[1].split(chr(47))[-1]}):",
       "matrix-core launches grouped by family, against the HIP-event figures bench.py prints in the same process (..._bench_under_rocprof.json)",
       "and in the unprofiled default run (..._final_bench.json): `in-step` = event brackets with both streams running (roofline.frac), `alone` =",
       "the same launches with the stream overlap off (roofline.alone).  (The fold-only wgrad_group launch of every step carries no FLOP: the",
       "rocprof call count of that family is one per block + one per step.)", ""]
out.append(f"{'family':14s} {'rocprof calls':>14s} {'rocprof avg us':>15s} | {'bench in-step us (same process)':>32s} {'alone':>8s} | {'bench in-step us (unprofiled)':>30s} {'alone':>8s}")
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    def cell(j, key):
        v = j["roofline"].get(key, {}).get(k)
        return f"{v['avg_launch_us']:.1f}" if v else "-"
    out.append(f"{k:14s} {c:14d} {t / c / 1e3:15.1f} | {cell(u, 'by_variant'):>32s} {cell(u, 'by_variant_alone'):>8s} | {cell(b, 'by_variant'):>30s} {cell(b, 'by_variant_alone'):>8s}")
out += ["", f"step: {b['ms_per_step']:.2f} ms unprofiled ({b['value']:.0f} images/s), {u['ms_per_step']:.2f} ms under rocprofv3; roofline.frac "
            f"{b['roofline']['frac']:.3f} ({b['roofline']['bound']}; alone {b['roofline']['alone']['frac']:.3f}), hbm view {b['roofline']['hbm']['frac']:.3f}"]
dom = b["roofline"].get("family") or max(fam, key=lambda k: fam[k][1])
c, t = fam[dom]
v = b["roofline"]["by_variant"].get(dom)
if v:
    lps = v["launches_per_step"]
    c_fl = c if dom != "wgrad_group" else c - c // (lps + 1)    # (the fold-only launch of every step carries no FLOP and is listed by rocprofv3 under the same kernel name)
    t_fl = t if dom != "wgrad_group" else t - (c - c_fl) * 27e3  # (~27 us each: `wgrad_fold` in the bench line)
    fl = v["flops_per_launch"]
    rp = fl / (t_fl / c_fl * 1e-9)
    out.append(f"bench line's dominant family: {dom} ({fl / 1e9:.1f} GFLOP per launch).  rocprofv3 average over its {c_fl} FLOP-carrying launches: {t_fl / c_fl / 1e3:.1f} us = "
               f"{rp / 1e12:.0f} TFLOP/s = {rp / 2.5e15:.3f} of the 2.5 PF bf16 roof; the bench line's roofline.frac (in-step, unprofiled run) is {b['roofline']['frac']:.3f}, "
               f"ratio {b['roofline']['frac'] / (rp / 2.5e15):.2f}")
# the family the bench line names must be the one that owns the step: the kernel template instantiation with the largest total duration in the
# rocprofv3 summary (every kernel of the trace, matrix-core or not)
top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
top_family = family(top["Name"])
out.append(f"top row of the rocprofv3 summary: {top['Name'][:60]} ({float(top['Percentage']):.2f} % of kernel time) = family {top_family}; "
           f"bench line's roofline.family: {b['roofline'].get('family')} (profiled process: {u['roofline'].get('family')})")
print("\n".join(out))
assert top_family == b["roofline"].get("family") == u["roofline"].get("family"), "roofline.family is not the top kernel of the rocprofv3 summary"
