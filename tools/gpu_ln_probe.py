import sys, time, torch
sys.path.insert(0, ".")
from dig_amd import ops
dev = torch.device("cuda:0")
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
rows, D = 65536, 384
x = torch.randn(rows, D, device=dev).bfloat16(); dy = torch.randn_like(x); dres = torch.randn_like(x)
g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev); dc = torch.zeros(D, device=dev); dx = torch.empty_like(x)
print("ln fwd %.1f us" % bench(lambda: ops.layernorm_fwd(x, g, b, 1e-6)))
print("ln bwd %.1f us" % bench(lambda: ops.layernorm_bwd(dy, x, g, b, mean, rstd, dres, dg, db, out=dx, dres_colsum=dc)))
