"""A few launches of each production GEMM flavour on the encoder shapes + attention + LayerNorm (for rocprofv3 --pmc runs)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd import ops
dev = torch.device("cuda:0")
I, D, F = 65536, 384, 1536
bf = lambda *s: torch.randn(*s, device=dev).bfloat16()
x, ln = bf(I, D), bf(I, D)
w_qkv, w_proj, w_fc1, w_fc2 = bf(3 * D, D), bf(D, D), bf(F, D), bf(D, F)
b3, b1, bF = torch.randn(3 * D, device=dev), torch.randn(D, device=dev), torch.randn(F, device=dev)
N = 3
for _ in range(N):
    qkv = ops.linear_fwd(ln, w_qkv, bias=b3, alpha=0.125, alpha_cols=D)               # wide, no resid
    ctx, lse = ops.attn_fwd(qkv, 256, 6, D)
    xm = ops.linear_fwd(ctx, w_proj, bias=b1, resid=x)                                # wide, resid
    pre = torch.empty(I, F, device=dev, dtype=torch.bfloat16)
    act = ops.linear_fwd(ln, w_fc1, bias=bF, act=1, pre=pre)                          # wide gelu+pre
    xo = ops.linear_fwd(act, w_fc2, bias=b1, resid=xm)                                # wide K=1536
    dact = ops.linear_dgrad(xo, w_fc2, gelu_pre=pre)                                  # dgrad + gelu'
    dln = ops.linear_dgrad(dact, w_fc1)                                               # dgrad K=1536
    dctx = ops.linear_dgrad(xo, w_proj)
    dqkv = ops.attn_bwd(qkv, ctx, dctx, lse, 256, 6, D, 0.125)
    dl1 = ops.linear_dgrad(dqkv, w_qkv)
    g1, g2 = torch.zeros(F, D, device=dev), torch.zeros(D, F, device=dev)
    ops.linear_wgrad(dact, ln, g1)
    ops.linear_wgrad(xo, act, g2)
torch.cuda.synchronize()
