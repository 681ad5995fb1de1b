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
def run(name, J, R, **kw):
    x = torch.randn(I, R, device=dev).bfloat16(); w = torch.randn(J, R, device=dev).bfloat16()
    bias = torch.randn(J, device=dev); res = torch.randn(I, J, device=dev).bfloat16()
    y = torch.empty(I, J, device=dev, dtype=torch.bfloat16); pre = torch.empty_like(y)
    args = dict(out=y, bias=bias)
    if kw.get("pre"): args["pre"] = pre
    if kw.get("resid"): args["resid"] = res
    if kw.get("act"): args["act"] = 1
    if kw.get("alpha"): args.update(alpha=0.125, alpha_cols=J // 3)
    out = []
    for bk in (64, 244, 344, 343):
        out.append(f"bk{bk} {bench(lambda: ops.gemm(x, w, I, J, R, bk=bk, **args)):.1f}")
    print(name, " | ".join(out))
run("qkv bias+alpha    ", 1152, 384, alpha=True)
run("proj bias+resid   ", 384, 384, resid=True)
run("fc1 gelu+pre      ", 1536, 384, act=True, pre=True)
run("fc1 gelu (momentum)", 1536, 384, act=True)
run("fc2 bias+resid    ", 384, 1536, resid=True)
# pix_projector-like: rows 32768
I = 32768
run("pixproj 384->512  ", 512, 384)
run("pixproj 512->512  ", 512, 512)

I = 8192
x = torch.randn(I, I, device=dev).bfloat16(); w = torch.randn(I, I, device=dev).bfloat16(); y = torch.empty(I, I, device=dev, dtype=torch.bfloat16)
for bk in (64, 244, 344, 343):
    t = bench(lambda: ops.gemm(x, w, I, I, I, out=y, bk=bk), n=10)
    print(f"8k^3 bk{bk}: {t:.1f} us {2*I**3/t/1e6:.0f} TF")
