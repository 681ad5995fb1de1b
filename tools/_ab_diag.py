import os, sys, torch
sys.path.insert(0, "/root/repo")
from dig_amd import ops
dev = torch.device("cuda:0")
n_img, D, H = 4, 384, 6
R = n_img * 256
scale = (D // H) ** -0.5
torch.manual_seed(0)
ln1 = torch.randn(R, D, device=dev).bfloat16()
x = torch.randn(R, D, device=dev).bfloat16()
wq = (torch.randn(3 * D, D, device=dev) * 0.05).bfloat16()
bq = torch.randn(3 * D, device=dev) * 0.3
bq[D:2 * D] = 0
wp = (torch.randn(D, D, device=dev) * 0.05).bfloat16()
bp = torch.randn(D, device=dev) * 0.3
qkv = ops.linear_fwd(ln1, wq, bias=bq, alpha=scale, alpha_cols=D)
ctx3, lse3 = ops.attn_fwd(qkv, n_img, H, D)
xm1, ctx1, qkv1, lse1 = ops.attn_block_fwd(ln1, x, wq, bq, wp, bp, n_img, H, D, scale, save=True)
torch.cuda.synchronize()
d = (lse1 - lse3).abs().view(n_img, H, 8, 32)
print("lse err by head:", d.amax(dim=(0, 2, 3)).tolist())
print("lse err by wave:", d.amax(dim=(0, 1, 3)).tolist())
print("lse err by lane&31 (img0 head0 wave0):", [round(v, 3) for v in d[0, 0, 0].tolist()])
# which keys are mis-scored?  reference scores for img 0 head 0
q = qkv[:256, 0:64].float(); k = qkv[:256, D:D + 64].float()
S = q @ k.t()
ref_lse = torch.logsumexp(S, dim=1)
print("old kernel lse vs torch:", (lse3[0] - ref_lse).abs().max().item())
# hypothesis tests: lse with subsets of keys or permuted d
for name, Sx in (("keys 0..127", S[:, :128]), ("keys 128..255", S[:, 128:])):
    print(name, (lse1[0] - torch.logsumexp(Sx, dim=1)).abs().max().item())
c = (ctx1.float() - ctx3.float()).abs().view(n_img, 256, H, 64)
print("ctx err by head:", c.amax(dim=(0, 1, 3)).tolist())
print("ctx err by d (img0, head0):", [round(v, 2) for v in c[0, :, 0].amax(dim=0).tolist()])
print("ctx err by query block:", c.view(n_img, 8, 32, H, 64).amax(dim=(0, 2, 3, 4)).tolist())
