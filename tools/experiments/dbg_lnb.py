import os, sys, torch
sys.path.insert(0, "/root/repo")
from dig_amd import ops
dev = torch.device("cuda:0")
R, D, Fh = 128, 384, 128
g = torch.Generator(device="cpu").manual_seed(1)
rn = lambda *s: torch.randn(*s, generator=g).to(dev)
x_mid = (rn(R, D) * 1.3 + 0.2).bfloat16()
gam, bet = 1.0 + 0.3 * rn(D), 0.1 * rn(D)
w1 = (rn(Fh, D) * 0.06).bfloat16(); b1 = rn(Fh) * 0.5
w2 = (rn(D, Fh) * 0.04).bfloat16()
dy = rn(R, D).bfloat16()
ln2, mu, rs = ops.layernorm_fwd(x_mid, gam, bet, 1e-6)
pre = torch.empty((R, Fh), device=dev, dtype=torch.bfloat16)
ops.linear_fwd(ln2, w1, bias=b1, act=1, pre=pre)
w2t, w1t = ops.transpose_bf16(w2), ops.transpose_bf16(w1)
dln2, dpre0, parts0 = ops.mlp_chain_bwd(dy, w2t, pre, w1t)
dgam0, dbet0, dcol0 = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
dxm0 = ops.layernorm_bwd(dln2, x_mid, gam, bet, mu, rs, dy, dgam0, dbet0, dres_colsum=dcol0)
dxm, dpre, parts, lnp = ops.mlp_chain_bwd_ln(dy, w2t, pre, w1t, x_mid, gam, mu, rs)
torch.cuda.synchronize()
d = (dxm.float() - dxm0.float()).abs()
print("dxm max diff", d.max().item(), "mean", d.mean().item(), "ref mean abs", dxm0.float().abs().mean().item())
print("per column block of 32 (mean diff):", [round(d[:, i*32:(i+1)*32].mean().item(), 4) for i in range(12)])
print("per row block of 32:", [round(d[i*32:(i+1)*32].mean().item(), 4) for i in range(4)])
print("cols within block (mean diff by col%32):", [round(d[:, c::32].mean().item(), 3) for c in range(32)])
s = lnp.sum(0)
for nm, a, b in (("dgam", s[0], dgam0), ("dbet", s[1], dbet0), ("dcol", s[2], dcol0)):
    e = (a - b).abs()
    print(nm, "max diff", e.max().item(), "ref max", b.abs().max().item(), "by col%32:", [round(e[c::32].mean().item(), 3) for c in range(0, 32, 4)])
# is dxm == dy + something? check dxm - dy vs dxm0 - dy
print("row 0 first 8 cols: new", dxm[0, :8].float().tolist(), "\n old", dxm0[0, :8].float().tolist(), "\n dy ", dy[0, :8].float().tolist())
