"""dgrad GEMMs (dx = dy @ W, transpose-read weight operand) of the encoder at 65536 rows: 128x128/BK32 (current) vs the wide tiles."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dig_amd import ops
dev = torch.device("cuda:0"); I = 65536
def bench(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for name, R, J in (("dgrad fc1 (1536->384)", 1536, 384), ("dgrad qkv (1152->384)", 1152, 384), ("dgrad proj (384->384)", 384, 384), ("dgrad fc2 (384->1536)", 384, 1536)):
    dy = torch.randn(I, R, device=dev).bfloat16(); w = (torch.randn(R, J, device=dev) * 0.05).bfloat16()
    out = torch.empty(I, J, device=dev, dtype=torch.bfloat16)
    ref = ops.gemm(dy, w, I, J, R, tb=True, bk=32).float()
    line = []
    for bk in (32, 64, 244, 264, 242):
        try:
            d = (ops.gemm(dy, w, I, J, R, tb=True, bk=bk).float() - ref).abs().max().item()
            t = bench(lambda: ops.gemm(dy, w, I, J, R, tb=True, out=out, bk=bk))
            line.append(f"{bk}: {t:6.1f} us ({2*I*J*R/t/1e6:4.0f} TF, d {d:.2g})")
        except Exception as e:
            line.append(f"{bk}: n/a")
    print(name, " | ".join(line))
