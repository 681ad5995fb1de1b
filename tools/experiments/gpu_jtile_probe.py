"""Forward GEMMs whose output width is not a multiple of 256 (J = 384: proj / fc2, J = 1152: qkv): the 256x256 tile wastes 25 % / 11 % of
its columns there -- does an exactly-tiling shape win?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dig_amd import ops
dev = torch.device("cuda:0"); I = 65536
def bench(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for name, J, R, resid in (("proj", 384, 384, True), ("fc2", 384, 1536, True), ("qkv", 1152, 384, False), ("fc1", 1536, 384, False)):
    x = torch.randn(I, R, device=dev).bfloat16(); w = (torch.randn(J, R, device=dev) * 0.05).bfloat16(); bias = torch.randn(J, device=dev)
    res = torch.randn(I, J, device=dev).bfloat16() if resid else None
    out = torch.empty(I, J, device=dev, dtype=torch.bfloat16)
    ref = ops.gemm(x, w, I, J, R, bias=bias, resid=res, bk=0).float()
    chk = ops.gemm(x, w, I, J, R, bias=bias, resid=res, bk=264).float()
    line = [f"264 vs 128x128 max|d| {(chk - ref).abs().max().item():.3g}"]
    for bk in (244, 264, 242, 0, 332):
        try:
            t = bench(lambda: ops.gemm(x, w, I, J, R, bias=bias, resid=res, out=out, bk=bk))
            line.append(f"{bk}: {t:6.1f} us ({2*I*J*R/t/1e6:4.0f} TF)")
        except Exception as e:
            line.append(f"{bk}: n/a")
    print(name, " | ".join(line))
