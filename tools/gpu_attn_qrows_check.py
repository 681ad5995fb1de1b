"""dig_attn_fwd / _bwd with q_rows < 32 (a handful of queries per image against its 256 keys: PatchNet's cross-attention) against fp32 torch:
relative error and norm ratio of ctx, dq, dk, dv.   python tools/gpu_attn_qrows_check.py [nw] [spread]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dig_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 5
spread = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
n_img, H, D, N = 16, 6, 384, 256
torch.manual_seed(0)
scale = 64 ** -0.5
q = (torch.randn(n_img, nw, D, device=dev) * spread).bfloat16()
k = (torch.randn(n_img, N, D, device=dev) * spread).bfloat16()
v = torch.randn(n_img, N, D, device=dev).bfloat16()
da = torch.randn(n_img, nw, D, device=dev).bfloat16()
fused = torch.zeros(n_img * N, 3 * D, device=dev, dtype=torch.bfloat16)
fv = fused.view(n_img, N, 3 * D)
fv[:, :nw, :D] = (q.float() * scale).bfloat16()
fv[:, :, D:2 * D] = k
fv[:, :, 2 * D:] = v
ctx, lse = ops.attn_fwd(fused, n_img, H, D, q_rows=nw)
dctx = torch.zeros(n_img * N, D, device=dev, dtype=torch.bfloat16)
dctx.view(n_img, N, D)[:, :nw] = da
dfused = ops.attn_bwd(fused, ctx, dctx, lse, n_img, H, D, scale, q_rows=nw)
dq = dfused.view(n_img, N, 3 * D)[:, :nw, :D].float()
dk = dfused.view(n_img, N, 3 * D)[:, :, D:2 * D].float()
dv = dfused.view(n_img, N, 3 * D)[:, :, 2 * D:].float()
# fp32 reference on the same bf16 inputs
qf = q.float().requires_grad_(True)
kf, vf = k.float().requires_grad_(True), v.float().requires_grad_(True)
qs = ((qf.view(n_img, nw, H, 64) * scale).bfloat16().float() - (qf.view(n_img, nw, H, 64) * scale)).detach() + qf.view(n_img, nw, H, 64) * scale
s = torch.einsum("bqhd,bkhd->bhqk", qs, kf.view(n_img, N, H, 64))
o = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), vf.view(n_img, N, H, 64)).reshape(n_img, nw, D)
o.backward(da.float())


def cmp(name, a, b):
    print(f"{name}: rel err {((a - b).norm() / b.norm()).item():.2e}   |a|/|b| {(a.norm() / b.norm()).item():.4f}")


cmp("ctx", ctx.view(n_img, N, D)[:, :nw].float(), o.detach())
cmp("dq ", dq, qf.grad)
cmp("dk ", dk, kf.grad)
cmp("dv ", dv, vf.grad)
