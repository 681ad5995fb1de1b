"""GPU-side checks of the non-GEMM kernels against torch fp32 (run on the MI355X box)."""
import ctypes
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from dig_amd import _lib as L

dev = torch.device("cuda:0")
torch.manual_seed(0)
cf = ctypes.c_float
ok_all = True


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-20)).item()


def report(name, errs, tol):
    global ok_all
    ok = all(e < tol for e in errs)
    ok_all &= ok
    print(f"{name}: " + " ".join(f"{e:.2e}" for e in errs) + (" OK" if ok else " FAIL"))


# ---- LayerNorm ----
for D, gelu in [(384, 0), (512, 0), (128, 0), (192, 1), (64, 1)]:
    rows = 1000
    x = torch.randn(rows, D, device=dev).bfloat16()
    g = torch.randn(D, device=dev) * 0.2 + 1
    b = torch.randn(D, device=dev) * 0.1
    y = torch.empty_like(x); mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    L.call("dig_layernorm_fwd", L.ptr(x), L.ptr(g), L.ptr(b), L.ptr(y), L.ptr(mean), L.ptr(rstd), rows, D, cf(1e-6), gelu, L.stream())
    xf = x.float().requires_grad_(True); gf = g.clone().requires_grad_(True); bf = b.clone().requires_grad_(True)
    ref = F.layer_norm(xf, (D,), gf, bf, 1e-6)
    if gelu:
        ref = F.gelu(ref)
    dy = torch.randn(rows, D, device=dev).bfloat16()
    dres = torch.randn(rows, D, device=dev).bfloat16()
    ref.backward(dy.float())
    dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev); dc = torch.zeros(D, device=dev)
    from dig_amd import ops as _ops
    dx = _ops.layernorm_bwd(dy, x, g, b, mean, rstd, dres, dg, db, gelu=bool(gelu), dres_colsum=dc)
    report(f"layernorm D={D} gelu={gelu}", [rel(y, ref), rel(dx, xf.grad + dres.float()), rel(dg, gf.grad), rel(db, bf.grad), rel(dc, dres.float().sum(0))], 1e-2)

# ---- BatchNorm ----
for rows, C, affine, relu in [(1024, 4096, 1, 1), (4096, 512, 1, 1), (333, 256, 0, 0), (32, 64, 0, 0)]:
    x = (torch.randn(rows, C, device=dev) * 2 + 0.5).bfloat16()
    gamma = (torch.randn(C, device=dev) * 0.2 + 1) if affine else None
    beta = (torch.randn(C, device=dev) * 0.1) if affine else None
    sums = torch.empty(2, C, device=dev)
    _ops.bn_stats(x, sums)
    y = torch.empty_like(x); mean = torch.empty(C, device=dev); rstd = torch.empty(C, device=dev)
    L.call("dig_bn_fwd_apply", L.ptr(x), L.ptr(sums), cf(rows), cf(1e-5), L.ptr(gamma), L.ptr(beta), relu, L.ptr(y), L.ptr(mean), L.ptr(rstd), rows, C, L.stream())
    xf = x.float().requires_grad_(True)
    gf = gamma.clone().requires_grad_(True) if affine else None
    bf = beta.clone().requires_grad_(True) if affine else None
    ref = F.batch_norm(xf, None, None, gf, bf, True, 0.1, 1e-5)
    if relu:
        ref = F.relu(ref)
    dy = torch.randn(rows, C, device=dev).bfloat16()
    ref.backward(dy.float())
    s2 = torch.empty(2, C, device=dev)
    _ops.bn_bwd_stats(dy, x, mean, rstd, gamma, beta, relu, s2)
    s2g = s2.clone()
    dx = torch.empty_like(x)
    L.call("dig_bn_bwd_apply", L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(beta), relu, L.ptr(s2g), cf(rows), L.ptr(dx), rows, C, L.stream())
    errs = [rel(y, ref), rel(dx, xf.grad), rel(mean, x.float().mean(0))]
    if affine:
        errs += [rel(s2[1], gf.grad), rel(s2[0], bf.grad)]
    report(f"batchnorm rows={rows} C={C} affine={affine} relu={relu}", errs, 1.5e-2)

# ---- patch embed ----
for Bn, D in [(6, 384), (3, 128)]:
    img = torch.rand(Bn, 3, 32, 128, device=dev) * 2 - 1
    W = torch.randn(D, 3, 4, 4, device=dev) * 0.1
    bias = torch.randn(D, device=dev) * 0.1
    mt = torch.randn(D, device=dev) * 0.1
    pos = torch.randn(256, D, device=dev)
    mask = (torch.rand(Bn, 256, device=dev) < 0.5)
    m8 = mask.to(torch.uint8)
    out = torch.empty(Bn * 256, D, device=dev, dtype=torch.bfloat16)
    L.call("dig_patch_embed_fwd", L.ptr(img), L.ptr(W), L.ptr(bias), L.ptr(m8), L.ptr(mt), L.ptr(pos), L.ptr(out), Bn, 8, 32, D, L.stream())
    Wf = W.clone().requires_grad_(True); bf = bias.clone().requires_grad_(True); mtf = mt.clone().requires_grad_(True)
    xe = F.conv2d(img, Wf, bf, stride=4).flatten(2).transpose(1, 2)
    mm = mask.unsqueeze(-1).float()
    ref = xe * (1 - mm) + mtf * mm + pos
    dy = torch.randn(Bn * 256, D, device=dev).bfloat16()
    ref.reshape(-1, D).backward(dy.float())
    dW = torch.zeros(D, 48, device=dev); dbias = torch.zeros(D, device=dev); dmt = torch.zeros(D, device=dev)
    L.call("dig_patch_embed_bwd", L.ptr(dy), L.ptr(img), L.ptr(m8), L.ptr(dW), L.ptr(dbias), L.ptr(dmt), Bn, 8, 32, D, L.stream())
    report(f"patch_embed Bn={Bn} D={D}", [rel(out, ref.reshape(-1, D)), rel(dW, Wf.grad.reshape(D, 48)), rel(dbias, bf.grad), rel(dmt, mtf.grad)], 5e-3)
    from dig_amd import ops
    dW2 = torch.zeros(D, 48, device=dev); db2 = torch.zeros(D, device=dev); dm2 = torch.zeros(D, device=dev)
    ops.patch_embed_bwd_mfma(dy, img, m8, dW2, db2, dm2, D, 8, 32)
    report(f"patch_embed_mfma Bn={Bn} D={D}", [rel(dW2, Wf.grad.reshape(D, 48)), rel(db2, bf.grad), rel(dm2, mtf.grad)], 5e-3)

# ---- window pool ----
Bn, D = 6, 384
x = torch.randn(Bn, 256, D, device=dev).bfloat16()
out = torch.empty(Bn * 4, D, device=dev, dtype=torch.bfloat16)
L.call("dig_window_pool_fwd", L.ptr(x), L.ptr(out), 0, Bn, 8, 32, 4, D, L.stream())
ref = x.float().reshape(Bn, 8, 4, 8, D).mean(dim=(1, 3)).reshape(Bn * 4, D)
dp = torch.randn(Bn * 4, D, device=dev).bfloat16()
dx = torch.randn(Bn, 256, D, device=dev).bfloat16(); dx0 = dx.clone()
L.call("dig_window_pool_bwd", L.ptr(dp), L.ptr(dx), Bn, 8, 32, 4, D, 1, L.stream())
refdx = dx0.float() + (dp.float().reshape(Bn, 1, 4, 1, D) / 64).expand(Bn, 8, 4, 8, D).reshape(Bn, 256, D)
report("window_pool", [rel(out, ref), rel(dx, refdx)], 5e-3)

# ---- mask_to_index / gather / scatter / mim_target / mse ----
B = 5
rng = np.random.RandomState(3)
mask = torch.zeros(B, 256, dtype=torch.uint8)
for b in range(B):
    mask[b, rng.permutation(256)[:179]] = 1
maskd = mask.to(dev)
idx = torch.full((B, 179), -1, device=dev, dtype=torch.int32); cnt = torch.zeros(B, device=dev, dtype=torch.int32)
L.call("dig_mask_to_index", L.ptr(maskd), L.ptr(idx), L.ptr(cnt), B, 256, 179, L.stream())
ref_idx = torch.nonzero(mask.reshape(-1)).squeeze(1).to(torch.int32).reshape(B, 179)
exact = torch.equal(idx.cpu(), ref_idx) and bool((cnt.cpu() == 179).all())
print("mask_to_index exact:", exact); ok_all &= exact
src = torch.randn(B * 256, 384, device=dev).bfloat16()
M = B * 179; Mp = ((M + 63) // 64) * 64
dst = torch.full((Mp, 384), 7.0, device=dev, dtype=torch.bfloat16)
L.call("dig_gather_rows", L.ptr(src), L.ptr(idx), L.ptr(dst), M, Mp, 384, L.stream())
g_ok = torch.equal(dst[:M], src[idx.reshape(-1).long()]) and bool((dst[M:] == 0).all())
acc = torch.randn(B * 256, 384, device=dev).bfloat16(); acc0 = acc.clone()
L.call("dig_scatter_rows_add", L.ptr(dst), L.ptr(idx), L.ptr(acc), M, 384, L.stream())
refacc = acc0.float(); refacc[idx.reshape(-1).long()] += dst[:M].float()
print("gather exact:", g_ok, "scatter err", rel(acc, refacc)); ok_all &= g_ok and rel(acc, refacc) < 5e-3
img = torch.rand(B, 3, 32, 128, device=dev) * 2 - 1
tgt = torch.empty(M, 48, device=dev)
L.call("dig_mim_target", L.ptr(img), L.ptr(idx), L.ptr(tgt), M, 8, 32, L.stream())
pp = (img * 0.5 + 0.5).reshape(B, 3, 8, 4, 32, 4).permute(0, 2, 4, 3, 5, 1).reshape(B * 256, 48)
t_ok = torch.equal(tgt, pp[idx.reshape(-1).long()])
print("mim_target exact:", t_ok); ok_all &= t_ok
pred = torch.randn(Mp, 64, device=dev)
loss = torch.zeros(1, device=dev); dpred = torch.empty(Mp, 64, device=dev, dtype=torch.bfloat16)
L.call("dig_mse_fwd_bwd", L.ptr(pred), 64, L.ptr(tgt), M, 48, cf(0.7), L.ptr(loss), L.ptr(dpred), 64, L.stream())
pf = pred[:M, :48].clone().requires_grad_(True)
rl = F.mse_loss(pf, tgt); (rl * 0.7).backward()
report("mse", [abs(loss.item() - rl.item()) / rl.item(), rel(dpred[:M, :48], pf.grad), float(dpred[:M, 48:].abs().max())], 5e-3)

# ---- gelu bwd, add, colsum ----
n = 4096 * 96
a = torch.randn(n, device=dev).bfloat16(); b2 = torch.randn(n, device=dev).bfloat16(); o = torch.empty_like(a)
L.call("dig_gelu_bwd", L.ptr(a), L.ptr(b2), L.ptr(o), ctypes.c_longlong(n), L.stream())
bb = b2.float().requires_grad_(True); F.gelu(bb).backward(a.float())
e1 = rel(o, bb.grad)
L.call("dig_add_bf16", L.ptr(a), L.ptr(b2), L.ptr(o), ctypes.c_longlong(n), L.stream())
e2 = rel(o, a.float() + b2.float())
xx = torch.randn(3000, 48, device=dev).bfloat16(); cs = torch.zeros(48, device=dev)
from dig_amd import ops as _ops
_ops.colsum(xx, cs)
report("gelu_bwd/add/colsum", [e1, e2, rel(cs, xx.float().sum(0))], 5e-3)

# ---- InfoNCE pieces ----
nq, mk, C = 512, 2048, 256
q = torch.randn(nq, C, device=dev); k = torch.randn(mk, C, device=dev)
qn = torch.empty_like(q); qi = torch.empty(nq, device=dev); kn = torch.empty_like(k); ki = torch.empty(mk, device=dev)
L.call("dig_l2norm_fwd", L.ptr(q), L.ptr(qn), L.ptr(qi), nq, C, cf(1e-12), L.stream())
L.call("dig_l2norm_fwd", L.ptr(k), L.ptr(kn), L.ptr(ki), mk, C, cf(1e-12), L.stream())
logits = torch.empty(nq, mk, device=dev)
T = 0.2
L.call("dig_sgemm", L.ptr(qn), L.ptr(kn), L.ptr(logits), nq, mk, C, C, C, mk, 0, cf(1.0 / T), L.stream())
qf = q.clone().requires_grad_(True)
rlog = F.normalize(qf, dim=1) @ F.normalize(k, dim=1).t() / T
off = 512
labels = torch.arange(nq, device=dev) + off
rloss = F.cross_entropy(rlog, labels) * 2 * T
rloss.backward()
e_log = rel(logits, rlog)
out3 = torch.zeros(3, device=dev)
gs = 2 * T / nq
L.call("dig_ce_rows", L.ptr(logits), nq, mk, off, cf(gs), L.ptr(out3), L.stream())
dqn = torch.empty(nq, C, device=dev)
L.call("dig_sgemm", L.ptr(logits), L.ptr(kn), L.ptr(dqn), nq, C, mk, mk, C, C, 1, cf(1.0 / T), L.stream())
dq = torch.empty_like(q)
L.call("dig_l2norm_bwd", L.ptr(dqn), L.ptr(qn), L.ptr(qi), L.ptr(dq), nq, C, L.stream())
top = rlog.topk(5, 1)[1]; hit = top.eq(labels[:, None])
report("infonce", [e_log, abs(out3[0].item() * 2 * T / nq - rloss.item()) / rloss.item(), rel(dq, qf.grad),
                   abs(out3[1].item() - hit[:, :1].sum().item()), abs(out3[2].item() - hit.sum().item())], 1e-4)

# ---- optimizer ----
n = 1 << 20
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 1e-2; m = torch.randn(n, device=dev) * 1e-3; v = torch.rand(n, device=dev) * 1e-5
p0, m0, v0 = p.clone(), m.clone(), v.clone()
sh = torch.empty(n, device=dev, dtype=torch.bfloat16)
segb = (ctypes.c_longlong * 2)(0, n // 2); sege = (ctypes.c_longlong * 2)(n // 2, n); segw = (ctypes.c_float * 2)(0.1, 0.0)
L.call("dig_adamw_step", L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.ptr(sh), ctypes.c_longlong(n), 2, segb, sege, segw, cf(1e-3), cf(0.9), cf(0.999), cf(1e-8), 3, cf(1.0), None, L.stream())
import math
rp = p0.clone(); rp[: n // 2] *= (1 - 1e-3 * 0.1)
rm = m0 * 0.9 + g * 0.1; rv = v0 * 0.999 + g * g * 0.001
den = rv.sqrt() / math.sqrt(1 - 0.999 ** 3) + 1e-8
rp -= (1e-3 / (1 - 0.9 ** 3)) * rm / den
report("adamw", [rel(p, rp), rel(m, rm), rel(v, rv), rel(sh, rp)], 2e-3)
report("adamw-tight", [float((p - rp).abs().max())], 1e-6)
pm = torch.randn(n, device=dev); pm0 = pm.clone()
L.call("dig_ema_update", L.ptr(pm), L.ptr(p), L.ptr(sh), ctypes.c_longlong(n), cf(0.99), L.stream())
report("ema", [float((pm - (pm0 * 0.99 + p * (1 - 0.99))).abs().max())], 1e-6)
ws = torch.empty(1024, device=dev); o1 = torch.empty(1, device=dev)
L.call("dig_sumsq", L.ptr(g), ctypes.c_longlong(n), L.ptr(ws), L.ptr(o1), L.stream())
report("sumsq", [abs(o1.item() - (g.double() ** 2).sum().item()) / (g.double() ** 2).sum().item()], 1e-5)
print("ALL_OK" if ok_all else "SOME_FAIL")
