"""Per-tile fixed cost of the forward GEMM variants: time vs reduction length R at fixed output size (T = T0 + c*R)."""
import sys, time, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd import ops
dev = torch.device("cuda:0")
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
I = 65536
for name, J, kw in (("plain bf16 out", 1536, {}), ("bias", 1536, dict(bias=1)), ("bias+gelu", 1536, dict(bias=1, act=1)), ("bias+gelu+pre", 1536, dict(bias=1, act=1, pre=1)),
                    ("bias+resid J=384", 384, dict(bias=1, resid=1)), ("plain J=384", 384, {})):
    for bk in (244, 64):
        row = []
        for R in (64, 128, 384, 768, 1536):
            x = torch.randn(I, R, device=dev).bfloat16(); w = torch.randn(J, R, device=dev).bfloat16()
            y = torch.empty(I, J, device=dev, dtype=torch.bfloat16)
            args = dict(out=y)
            if kw.get("bias"): args["bias"] = torch.randn(J, device=dev)
            if kw.get("act"): args["act"] = 1
            if kw.get("pre"): args["pre"] = torch.empty_like(y)
            if kw.get("resid"): args["resid"] = torch.randn(I, J, device=dev).bfloat16()
            row.append(bench(lambda: ops.gemm(x, w, I, J, R, bk=bk, **args)))
        print(f"{name:18s} bk{bk:3d} " + " ".join(f"R{r}:{t:6.1f}" for r, t in zip((64, 128, 384, 768, 1536), row)), flush=True)
