import sys, time, torch
sys.path.insert(0, ".")
from dig_amd import ops
dev = torch.device("cuda:0")
def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
I = 65536
for J, R in [(1536, 64), (1536, 384), (384, 64), (384, 384), (1152, 64), (1152, 384), (384, 1536)]:
    x = torch.randn(I, R, device=dev).bfloat16(); w = torch.randn(J, R, device=dev).bfloat16()
    bias = torch.randn(J, device=dev); res = torch.randn(I, J, device=dev).bfloat16()
    y = torch.empty(I, J, device=dev, dtype=torch.bfloat16); pre = torch.empty_like(y)
    t0 = bench(lambda: ops.linear_fwd(x, w, out=y))
    t1 = bench(lambda: ops.linear_fwd(x, w, bias=bias, resid=res, out=y))
    t2 = bench(lambda: ops.linear_fwd(x, w, bias=bias, act=1, out=y))
    t3 = bench(lambda: ops.linear_fwd(x, w, bias=bias, act=1, pre=pre, out=y))
    mb = I * J * 2 / 1e6
    print(f"J={J} R={R}: plain {t0:.1f} us (write {mb/t0:.2f} TB/s) | +bias+resid {t1:.1f} | +gelu {t2:.1f} | +gelu+pre {t3:.1f}")
# pure copy bandwidth reference
a = torch.empty(I * 1536, device=dev, dtype=torch.bfloat16); b = torch.empty_like(a)
t = bench(lambda: b.copy_(a)); print(f"torch copy 201MB: {t:.1f} us -> {2*201.3/t:.2f} TB/s (r+w)")
t = bench(lambda: b.zero_()); print(f"torch fill 201MB: {t:.1f} us -> {201.3/t:.2f} TB/s (w)")
print("---- BK=32 forward")
ops.GEMM_BK_FWD = 32
for J, R in [(1536, 384), (384, 384), (1152, 384), (384, 1536)]:
    x = torch.randn(I, R, device=dev).bfloat16(); w = torch.randn(J, R, device=dev).bfloat16()
    bias = torch.randn(J, device=dev); res = torch.randn(I, J, device=dev).bfloat16()
    y = torch.empty(I, J, device=dev, dtype=torch.bfloat16); pre = torch.empty_like(y)
    t0 = bench(lambda: ops.linear_fwd(x, w, out=y))
    t1 = bench(lambda: ops.linear_fwd(x, w, bias=bias, resid=res, out=y))
    t2 = bench(lambda: ops.linear_fwd(x, w, bias=bias, act=1, out=y))
    t3 = bench(lambda: ops.linear_fwd(x, w, bias=bias, act=1, pre=pre, out=y))
    print(f"J={J} R={R}: plain {t0:.1f} us | +bias+resid {t1:.1f} | +gelu {t2:.1f} | +gelu+pre {t3:.1f}")
