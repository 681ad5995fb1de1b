"""GEMM shapes of the BN-MLP heads (few rows, wide layers) across tile variants; InfoNCE sgemm; mse."""
import sys, time, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd import ops
dev = torch.device("cuda:0")
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
def shape(name, I, J, R, tb=False, bks=(64, 32, 212, 221, 232, 242, 244)):
    x = torch.randn(I, R, device=dev).bfloat16()
    w = (torch.randn(R, J, device=dev) if tb else torch.randn(J, R, device=dev)).bfloat16()
    y = torch.empty(I, J, device=dev, dtype=torch.bfloat16)
    ts = {bk: bench(lambda: ops.gemm(x, w, I, J, R, out=y, tb=tb, bk=bk)) for bk in bks}
    b = min(ts, key=ts.get)
    print(f"{name:30s}", " ".join(f"{k}:{v:6.1f}" for k, v in ts.items()), f"| best {b} {2*I*J*R/ts[b]/1e6:.0f} TF", flush=True)
shape("proj fwd 1024x4096x4096", 1024, 4096, 4096)
shape("proj fwd 1024x4096x384", 1024, 4096, 384)
shape("proj fwd 1024x256x4096", 1024, 256, 4096)
shape("proj dgrad 1024x4096x4096", 1024, 4096, 4096, tb=True)
shape("proj dgrad 1024x384x4096", 1024, 384, 4096, tb=True)
shape("pixproj fwd 32768x4096x384", 32768, 4096, 384)
shape("pixproj fwd 32768x384x4096", 32768, 384, 4096)
shape("pixproj dgrad 32768x4096x384", 32768, 4096, 384, tb=True)
shape("pixproj dgrad 32768x384x4096", 32768, 384, 4096, tb=True)
a = torch.randn(512, 256, device=dev); k = torch.randn(512, 256, device=dev); lg = torch.empty(512, 512, device=dev); dq = torch.empty(512, 256, device=dev)
print("sgemm logits 512x512x256:", f"{bench(lambda: ops.sgemm(a, k, lg, 512, 512, 256, False, 5.0)):.1f} us;  dq 512x256x512:", f"{bench(lambda: ops.sgemm(lg, k, dq, 512, 256, 512, True, 5.0)):.1f} us")
k8 = torch.randn(4096, 256, device=dev); lg8 = torch.empty(512, 4096, device=dev)
print("sgemm (8 ranks) logits 512x4096x256:", f"{bench(lambda: ops.sgemm(a, k8, lg8, 512, 4096, 256, False, 5.0)):.1f} us;  dq:", f"{bench(lambda: ops.sgemm(lg8, k8, dq, 512, 256, 4096, True, 5.0)):.1f} us")
