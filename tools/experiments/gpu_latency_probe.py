import sys, time, torch
sys.path.insert(0, ".")
from dig_amd import ops
dev = torch.device("cuda:0")
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
# one tile per CU (256 tiles), then 2, 4 per CU
for tiles in (256, 512, 1024, 2048):
    I, J = 128 * tiles, 128
    for R in (64, 128, 384, 1536):
        x = torch.randn(I, R, device=dev).bfloat16(); w = torch.randn(J, R, device=dev).bfloat16()
        y = torch.empty(I, J, device=dev, dtype=torch.bfloat16)
        t = bench(lambda: ops.gemm(x, w, I, J, R, out=y, bk=64))
        t32 = bench(lambda: ops.gemm(x, w, I, J, R, out=y, bk=32))
        print(f"tiles={tiles} R={R}: bk64 {t:.1f} us  bk32 {t32:.1f} us")
e = torch.empty(1, device=dev)
print("empty kernel-ish (fill 4B):", bench(lambda: e.zero_()))
