"""All GEMM shapes of one ViT-S step x tile variants (us per launch, TFLOP/s of the best)."""
import sys, time, torch
sys.path.insert(0, ".")
from dig_amd import ops
dev = torch.device("cuda:0")
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
ROWS = 65536
def fwd(name, J, R, bks=(64, 244, 448), **kw):
    I = kw.pop("rows", ROWS)
    x = torch.randn(I, R, device=dev).bfloat16(); w = torch.randn(J, R, device=dev).bfloat16()
    bias = torch.randn(J, device=dev); res = torch.randn(I, J, device=dev).bfloat16()
    y = torch.empty(I, J, device=dev, dtype=torch.bfloat16); pre = torch.empty_like(y)
    args = dict(out=y, bias=bias)
    if kw.get("pre"): args["pre"] = pre
    if kw.get("resid"): args["resid"] = res
    if kw.get("act"): args["act"] = 1
    ts = {bk: bench(lambda: ops.gemm(x, w, I, J, R, bk=bk, **args)) for bk in bks}
    b = min(ts, key=ts.get)
    print(f"fwd   {name:22s}", " ".join(f"{bk}:{t:6.1f}" for bk, t in ts.items()), f"| best {b} {2*I*J*R/ts[b]/1e6:.0f} TF", flush=True)
def dgrad(name, J, R, gelu=False, bks=(32, 64, 244, 422, 424, 423)):
    I = ROWS
    dy = torch.randn(I, R, device=dev).bfloat16(); w = torch.randn(R, J, device=dev).bfloat16()
    y = torch.empty(I, J, device=dev, dtype=torch.bfloat16); pre = torch.randn(I, J, device=dev).bfloat16()
    args = dict(out=y, tb=True)
    if gelu: args.update(act=2, resid=pre)
    ts = {bk: bench(lambda: ops.gemm(dy, w, I, J, R, bk=bk, **args)) for bk in bks}
    b = min(ts, key=ts.get)
    print(f"dgrad {name:22s}", " ".join(f"{bk}:{t:6.1f}" for bk, t in ts.items()), f"| best {b} {2*I*J*R/ts[b]/1e6:.0f} TF", flush=True)
def wgrad(name, I, J, bks=(32, 64, 244, 264, 242), splits=(8, 16, 24, 32, 40, 48)):
    R = ROWS
    dy = torch.randn(R, I, device=dev).bfloat16(); x = torch.randn(R, J, device=dev).bfloat16()
    dw = torch.zeros(I, J, device=dev)
    out = []
    for bk in bks:
        for sp0 in splits:
            sp = ops.L.lib().dig_gemm_effective_splits(R, sp0)
            ws = ops._workspace(dev, sp * I * J)
            tg = bench(lambda: ops.gemm(dy, x, I, J, R, ta=True, tb=True, out=ws, out_kind=ops.OUT_F32_PARTIAL, splits=sp, ldc=J, bk=bk))
            t = bench(lambda: (ops.gemm(dy, x, I, J, R, ta=True, tb=True, out=ws, out_kind=ops.OUT_F32_PARTIAL, splits=sp, ldc=J, bk=bk),
                               ops.L.call("dig_reduce_partials", ops.L.ptr(ws), sp, ops.cll(I * J), ops.L.ptr(dw), 1, ops.L.stream())))
            out.append((t, tg, bk, sp))
    b = min(out)
    print(f"wgrad {name:18s}", " ".join(f"{bk}/s{sp}:{tg:5.1f}+{t-tg:4.1f}" for t, tg, bk, sp in out), f"| best {b[2]}/s{b[3]} {b[0]:.1f} us {2*I*J*R/b[0]/1e6:.0f} TF", flush=True)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "fwd"):
    fwd("qkv 384->1152 bias", 1152, 384)
    fwd("proj 384->384 +res", 384, 384, resid=True)
    fwd("fc1 gelu+pre", 1536, 384, act=True, pre=True)
    fwd("fc1 gelu", 1536, 384, act=True)
    fwd("fc2 1536->384 +res", 384, 1536, resid=True)
if which in ("all", "dgrad"):
    dgrad("fc2 (gelu') ->1536", 1536, 384, gelu=True)
    dgrad("fc1 ->384 K1536", 384, 1536)
    dgrad("proj ->384 K384", 384, 384)
    dgrad("qkv ->384 K1152", 384, 1152)
if which in ("all", "wgrad"):
    wgrad("fc2 384x1536", 384, 1536)
    wgrad("fc1 1536x384", 1536, 384)
    wgrad("proj 384x384", 384, 384)
    wgrad("qkv 1152x384", 1152, 384)
if which in ("all", "big"):
    I = 8192
    x = torch.randn(I, I, device=dev).bfloat16(); w = torch.randn(I, I, device=dev).bfloat16(); y = torch.empty(I, I, device=dev, dtype=torch.bfloat16)
    for bk in (64, 244, 422, 424, 423):
        t = bench(lambda: ops.gemm(x, w, I, I, I, out=y, bk=bk), n=10)
        print(f"8k^3 bk{bk}: {t:.1f} us {2*I**3/t/1e6:.0f} TF", flush=True)
