"""LayerNorm forward / backward alone at the step's shape (65 536 x 384 bf16; `python tools/gpu_ln_probe.py R D` for another, e.g. 131072 512 =
ViT-Base at B = 256): microseconds and achieved HBM rate."""
import sys, torch
sys.path.insert(0, ".")
from dig_amd import ops
dev = torch.device("cuda:0")
R, D = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (65536, 384)
print(f"rows {R} width {D}")
x = torch.randn(R, D, device=dev).bfloat16(); dy = torch.randn(R, D, device=dev).bfloat16(); dres = torch.randn(R, D, device=dev).bfloat16()
g, b = torch.randn(D, device=dev), torch.randn(D, device=dev)
dg, db, dc = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
def bench(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = torch.empty_like(x)
t = bench(lambda: ops.layernorm_fwd(x, g, b, 1e-6))
print(f"ln_fwd            {t:6.1f} us  {2 * R * D * 2 / t / 1e6:5.2f} TB/s (read x, write y)")
t = bench(lambda: ops.layernorm_bwd(dy, x, g, b, mean, rstd, dres, dg, db, out=out, dres_colsum=dc, defer=True))
print(f"ln_bwd partials   {t:6.1f} us  {4 * R * D * 2 / t / 1e6:5.2f} TB/s (read dy, x, dres; write dx)")
t = bench(lambda: ops.layernorm_bwd(dy, x, g, b, mean, rstd, None, dg, db, out=out, defer=True))
print(f"ln_bwd no dres    {t:6.1f} us  {3 * R * D * 2 / t / 1e6:5.2f} TB/s")
