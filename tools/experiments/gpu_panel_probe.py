import os, sys, time, torch
sys.path.insert(0, ".")
from dig_amd import ops
dev = torch.device("cuda:0")
def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
R = 384
for I in (128 * 64, 128 * 128, 128 * 256, 128 * 512, 128 * 1024):
    for J in (128, 768, 1536):
        x = torch.randn(I, R, device=dev).bfloat16(); w = torch.randn(J, R, device=dev).bfloat16()
        y = torch.empty(I, J, device=dev, dtype=torch.bfloat16)
        t = bench(lambda: ops.gemm_panel(x, w, I, J, R, out=y))
        print(f"dbg {os.environ.get('DIG_PANEL_DBG')} WGs={I//128} tiles/WG={J//128}: {t:.1f} us")
