"""GPU-side check + timing of dig_attn_fwd/bwd against torch fp32 (run on the MI355X box)."""
import ctypes
import sys
import time

import torch

sys.path.insert(0, ".")
from dig_amd import _lib as L

dev = torch.device("cuda:0")
torch.manual_seed(0)


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-20)).item()


def run(Bn, H, scale, spike=False):
    D = H * 64
    qkv = (torch.randn(Bn * 256, 3 * D, device=dev) * 1.0).bfloat16()
    if spike:
        qkv[5, :64] *= 6.0
        qkv[77, D:D + 64] *= 6.0
    ctx = torch.empty(Bn * 256, D, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(Bn * H, 256, device=dev)
    L.call("dig_attn_fwd", L.ptr(qkv), L.ptr(ctx), L.ptr(lse), Bn, H, D, L.stream())
    x = qkv.float().requires_grad_(True)
    t = x.reshape(Bn, 256, 3, H, 64).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]
    s = q @ k.transpose(-2, -1)
    a = s.softmax(-1)
    o = (a @ v).transpose(1, 2).reshape(Bn * 256, D)
    ref_lse = torch.logsumexp(s, -1).reshape(Bn * H, 256)
    e_o, e_l = rel(ctx, o), (lse - ref_lse).abs().max().item()
    dctx = torch.randn(Bn * 256, D, device=dev).bfloat16()
    o.backward(dctx.float())
    g = x.grad.clone()
    g[:, :D] *= scale
    dqkv = torch.empty_like(qkv)
    L.call("dig_attn_bwd", L.ptr(qkv), L.ptr(ctx), L.ptr(dctx), L.ptr(lse), L.ptr(dqkv), Bn, H, D, ctypes.c_float(scale), None, None, L.stream())
    e_q, e_k, e_v = rel(dqkv[:, :D], g[:, :D]), rel(dqkv[:, D:2 * D], g[:, D:2 * D]), rel(dqkv[:, 2 * D:], g[:, 2 * D:])
    ok = max(e_o, e_q, e_k, e_v) < 2e-2 and e_l < 2e-2
    print(f"Bn={Bn} H={H} scale={scale} spike={spike}: ctx {e_o:.2e} lse {e_l:.2e} dq {e_q:.2e} dk {e_k:.2e} dv {e_v:.2e} {'OK' if ok else 'FAIL'}")
    return ok


ok = run(2, 2, 1.0) & run(4, 6, 0.125) & run(3, 8, 0.125, spike=True)


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


Bn, H = 256, 6
D = H * 64
qkv = torch.randn(Bn * 256, 3 * D, device=dev).bfloat16()
ctx = torch.empty(Bn * 256, D, device=dev, dtype=torch.bfloat16)
dctx = torch.randn(Bn * 256, D, device=dev).bfloat16()
dqkv = torch.empty_like(qkv)
lse = torch.empty(Bn * H, 256, device=dev)
tf = bench(lambda: L.call("dig_attn_fwd", L.ptr(qkv), L.ptr(ctx), L.ptr(lse), Bn, H, D, L.stream()))
tb = bench(lambda: L.call("dig_attn_bwd", L.ptr(qkv), L.ptr(ctx), L.ptr(dctx), L.ptr(lse), L.ptr(dqkv), Bn, H, D, ctypes.c_float(0.125), None, None, L.stream()))
fl = 4.0 * 256 * 256 * 64 * Bn * H
print(f"attn fwd {tf*1e6:.1f} us {fl/tf/1e12:.0f} TF | bwd {tb*1e6:.1f} us {2.5*fl/tb/1e12:.0f} TF (algorithmic 2.5x fwd)")
qq = qkv.view(Bn, 256, 3, H, 64).permute(2, 0, 3, 1, 4)
ts = bench(lambda: torch.nn.functional.scaled_dot_product_attention(qq[0], qq[1], qq[2], scale=1.0))
print(f"torch sdpa fwd {ts*1e6:.1f} us {fl/ts/1e12:.0f} TF")
print("ALL_OK" if ok else "SOME_FAIL")
