"""GPU-side check + timing of dig_gemm_bf16 against torch (run on the MI355X box)."""
import ctypes
import sys
import time

import torch

sys.path.insert(0, ".")
from dig_amd import _lib as L

dev = torch.device("cuda:0")
torch.manual_seed(0)


def gemm(A, B, I, J, R, ta, tb, out_kind, bias=None, resid=None, pre=None, alpha=1.0, alpha_cols=0, act=0, splits=1, C=None):
    if C is None:
        C = torch.empty((I, J), device=dev, dtype=torch.bfloat16 if out_kind == 0 else torch.float32)
    L.call("dig_gemm_bf16", L.ptr(A), L.ptr(B), L.ptr(C), I, J, R, A.stride(0), B.stride(0), C.stride(0),
           int(ta), int(tb), out_kind, L.ptr(bias), L.ptr(resid), resid.stride(0) if resid is not None else 0,
           L.ptr(pre), pre.stride(0) if pre is not None else 0, ctypes.c_float(alpha), alpha_cols, act, splits, 0, 0, L.stream())
    return C


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-20)).item()


ok = True
for (I, J, R) in [(256, 256, 128), (2048, 1152, 384), (716, 48, 192), (65536, 1536, 384), (1024, 4096, 4096), (300, 200, 64)]:
    x = torch.randn(I, R, device=dev).bfloat16()
    w = torch.randn(J, R, device=dev).bfloat16() * 0.05
    bias = torch.randn(J, device=dev)
    res = torch.randn(I, J, device=dev).bfloat16()
    # forward NT with bias+resid
    y = gemm(x, w, I, J, R, False, False, 0, bias=bias, resid=res)
    ref = x.float() @ w.float().t() + bias + res.float()
    e1 = rel(y, ref)
    # gelu + pre
    pre = torch.empty(I, J, device=dev, dtype=torch.bfloat16)
    y2 = gemm(x, w, I, J, R, False, False, 0, bias=bias, pre=pre, act=1)
    h = x.float() @ w.float().t() + bias
    e2 = rel(y2, torch.nn.functional.gelu(h)); e2b = rel(pre, h)
    # alpha cols, fp32 out
    y3 = gemm(x, w, I, J, R, False, False, 1, bias=bias, alpha=0.125, alpha_cols=(J // 8) * 4)
    ref3 = h.clone(); ref3[:, :(J // 8) * 4] *= 0.125
    e3 = rel(y3, ref3)
    # dgrad: dx[I,R'] = dy[I,J] @ w[J,R']  -> A=dy direct (reduction J), B=w transposed storage
    if J % 64 == 0:
        dy = torch.randn(I, J, device=dev).bfloat16()
        dx = gemm(dy, w, I, R, J, False, True, 0)
        e4 = rel(dx, dy.float() @ w.float())
    else:
        e4 = 0.0
    # wgrad: dW[J,R] = dy^T[J,I] @ x[I,R]  (reduction I) both transposed, atomic split
    dy = torch.randn(I, J, device=dev).bfloat16()
    dW = torch.zeros(J, R, device=dev)
    gemm(dy, x, J, R, I, True, True, 2, splits=8, C=dW)
    refw = dy.float().t() @ x.float()
    e5 = rel(dW, refw)
    dW1 = gemm(dy, x, J, R, I, True, True, 1)
    e6 = rel(dW1, refw)
    good = max(e1, e2, e2b, e3, e4) < 1e-2 and max(e5, e6) < 2e-3
    ok &= good
    print(f"I={I} J={J} R={R}: fwd {e1:.2e} gelu {e2:.2e} pre {e2b:.2e} alpha/f32 {e3:.2e} dgrad {e4:.2e} wgrad-atomic {e5:.2e} wgrad {e6:.2e} {'OK' if good else 'FAIL'}")

# timing
def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n

for (I, J, R, name) in [(65536, 1152, 384, "qkv"), (65536, 384, 384, "proj"), (65536, 1536, 384, "fc1"), (65536, 384, 1536, "fc2"), (8192, 8192, 8192, "8k")]:
    x = torch.randn(I, R, device=dev).bfloat16()
    w = torch.randn(J, R, device=dev).bfloat16()
    y = torch.empty(I, J, device=dev, dtype=torch.bfloat16)
    t = bench(lambda: gemm(x, w, I, J, R, False, False, 0, C=y))
    tt = bench(lambda: torch.matmul(x, w.t()))
    fl = 2.0 * I * J * R
    dy = torch.randn(I, J, device=dev).bfloat16()
    dx = torch.empty(I, R, device=dev, dtype=torch.bfloat16)
    t2 = bench(lambda: gemm(dy, w, I, R, J, False, True, 0, C=dx))
    dW = torch.zeros(J, R, device=dev)
    sp = max(1, min(64, 1024 // (((J + 127) // 128) * ((R + 127) // 128))))
    t3 = bench(lambda: gemm(dy, x, J, R, I, True, True, 2, splits=sp, C=dW))
    print(f"{name}: fwd {t*1e6:.1f} us {fl/t/1e12:.0f} TF | torch {tt*1e6:.1f} us {fl/tt/1e12:.0f} TF | dgrad {t2*1e6:.1f} us {fl/t2/1e12:.0f} TF | wgrad(split {sp}) {t3*1e6:.1f} us {fl/t3/1e12:.0f} TF")
print("ALL_OK" if ok else "SOME_FAIL")
