"""GPU-side check + timing of dig_gemm_bf16 against torch (run on the MI355X box)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from dig_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-20)).item()


ok = True
for (I, J, R) in [(256, 256, 128), (2048, 1152, 384), (716, 48, 192), (65536, 1536, 384), (1024, 4096, 4096), (300, 200, 64), (32, 64, 256)]:
    x = torch.randn(I, R, device=dev).bfloat16()
    w = torch.randn(J, R, device=dev).bfloat16() * 0.05
    bias = torch.randn(J, device=dev)
    res = torch.randn(I, J, device=dev).bfloat16()
    y = ops.linear_fwd(x, w, bias=bias, resid=res)
    ref = x.float() @ w.float().t() + bias + res.float()
    e1 = rel(y, ref)
    pre = torch.empty(I, J, device=dev, dtype=torch.bfloat16)
    y2 = ops.linear_fwd(x, w, bias=bias, pre=pre, act=1)
    h = x.float() @ w.float().t() + bias
    e2 = rel(y2, torch.nn.functional.gelu(h)); e2b = rel(pre, h)
    ac = (J // 16) * 8
    y3 = ops.linear_fwd(x, w, bias=bias, alpha=0.125, alpha_cols=ac, out_kind=ops.OUT_F32)
    ref3 = h.clone(); ref3[:, :ac] *= 0.125
    e3 = rel(y3, ref3)
    if J % 64 == 0:
        dy = torch.randn(I, J, device=dev).bfloat16()
        dx = ops.linear_dgrad(dy, w)
        e4 = rel(dx, dy.float() @ w.float())
    else:
        e4 = 0.0
    dy = torch.randn(I, J, device=dev).bfloat16()
    dW = torch.randn(J, R, device=dev); dW0 = dW.clone()
    ops.linear_wgrad(dy, x, dW)
    refw = dW0 + dy.float().t() @ x.float()
    e5 = rel(dW, refw)
    if J % 64 == 0:
        prex = torch.randn(I, R, device=dev).bfloat16()
        dxg = ops.linear_dgrad(dy2 := torch.randn(I, J, device=dev).bfloat16(), w, gelu_pre=prex)
        pp = prex.float().requires_grad_(True)
        torch.nn.functional.gelu(pp).backward(dy2.float() @ w.float())
        e4 = max(e4, rel(dxg, pp.grad))
    good = max(e1, e2, e2b, e3, e4) < 1e-2 and e5 < 2e-3
    ok &= good
    print(f"I={I} J={J} R={R}: fwd {e1:.2e} gelu {e2:.2e} pre {e2b:.2e} alpha/f32 {e3:.2e} dgrad {e4:.2e} wgrad {e5:.2e} {'OK' if good else 'FAIL'}")


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


import os
if os.environ.get("DIG_BK"):
    ops.GEMM_BK_FWD = ops.GEMM_BK_BWD = int(os.environ["DIG_BK"])
print("GEMM_BK fwd/bwd", ops.GEMM_BK_FWD, ops.GEMM_BK_BWD)
for (I, J, R, name) in [(65536, 1152, 384, "qkv"), (65536, 384, 384, "proj"), (65536, 1536, 384, "fc1"), (65536, 384, 1536, "fc2"), (1024, 4096, 4096, "head"), (8192, 8192, 8192, "8k")]:
    x = torch.randn(I, R, device=dev).bfloat16()
    w = torch.randn(J, R, device=dev).bfloat16()
    bias = torch.randn(J, device=dev)
    res = torch.randn(I, J, device=dev).bfloat16()
    y = torch.empty(I, J, device=dev, dtype=torch.bfloat16)
    pre = torch.empty(I, J, device=dev, dtype=torch.bfloat16)
    t = bench(lambda: ops.linear_fwd(x, w, out=y))
    tb = bench(lambda: ops.linear_fwd(x, w, bias=bias, resid=res, out=y))
    tg = bench(lambda: ops.linear_fwd(x, w, bias=bias, act=1, pre=pre, out=y))
    tt = bench(lambda: torch.matmul(x, w.t()))
    fl = 2.0 * I * J * R
    dy = torch.randn(I, J, device=dev).bfloat16()
    dx = torch.empty(I, R, device=dev, dtype=torch.bfloat16)
    t2 = bench(lambda: ops.linear_dgrad(dy, w, out=dx))
    dW = torch.zeros(J, R, device=dev)
    t3 = bench(lambda: ops.linear_wgrad(dy, x, dW))
    print(f"{name}: fwd {t*1e6:.1f} us {fl/t/1e12:.0f} TF | +bias+resid {tb*1e6:.1f} | +bias+gelu+pre {tg*1e6:.1f} | torch {tt*1e6:.1f} us {fl/tt/1e12:.0f} TF | dgrad {t2*1e6:.1f} us {fl/t2/1e12:.0f} TF | wgrad {t3*1e6:.1f} us {fl/t3/1e12:.0f} TF")
print("ALL_OK" if ok else "SOME_FAIL")
