"""HBM traffic per launch by kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py.

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
FETCH_SIZE / WRITE_SIZE are in KB; gfx950 tallies 128-byte read requests as 64 B, so fetch is doubled
(/opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv, json, re, sys, collections


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
    for k in ("attn_fwd", "attn_bwd", "ln_bwd_kernel", "ln_fwd_kernel", "reduce_partials", "colsum_kernel", "bn_fwd_apply", "bn_bwd_apply", "adamw", "ema_kernel"):
        if k in name:
            return k.replace("_kernel", "")
    return None


def load(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        f = family(r["Kernel_Name"])
        if f:
            acc[f][0] += float(r["Counter_Value"]) * 1024.0
            acc[f][1] += 1
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 1 --no-cpu-baseline "
               "--no-mim-only`; bytes per launch averaged over all launches of the kernel family; FETCH_SIZE doubled per "
               "MI355X_MICROARCH.md (gfx950 tallies 128-B requests as 64 B)", "kernels": {}}
for f in sorted(set(fetch) | set(write)):
    fb = 2.0 * fetch[f][0] / max(fetch[f][1], 1)
    wb = write[f][0] / max(write[f][1], 1)
    out["kernels"][f] = {"launches": fetch[f][1], "fetch_bytes_per_launch_corrected": fb, "write_bytes_per_launch": wb,
                         "hbm_bytes_per_launch": fb + wb}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for f, v in out["kernels"].items():
    print(f"{f:18s} launches {v['launches']:5d}  fetch {v['fetch_bytes_per_launch_corrected']/1e6:8.1f} MB  write {v['write_bytes_per_launch']/1e6:8.1f} MB")
