"""The fused attention sub-block (csrc/attn_block.hip) against the three launches it replaces (qkv GEMM -> dig_attn_fwd -> proj GEMM +
residual), ViT-S shapes, alone on the GPU: (1) results -- qkv must be bit-identical, ctx / lse / x_mid agree to bf16 rounding, both against
the three-launch path and against an fp32 torch reference; (2) microseconds per launch for both branches (momentum: nothing kept; online:
qkv + lse kept) and the effective TFLOP/s on the algorithmic FLOP (2 R D 3D + 4 R 256 D + 2 R D D)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd import ops  # noqa: E402


def timeit(fn, iters=20, reps=7):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / iters * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def reference(ln1, x, wq, bq, wp, bp, n_img, H, D, scale):
    """fp32 torch on the bf16 inputs (intermediates NOT rounded)."""
    qkv = ln1.float() @ wq.float().t() + bq
    q, k, v = qkv.view(n_img, 256, 3, H, D // H).permute(2, 0, 3, 1, 4)
    s = (q * scale) @ k.transpose(-1, -2)
    lse = torch.logsumexp(s, dim=-1)
    ctx = (torch.softmax(s, dim=-1) @ v).permute(0, 2, 1, 3).reshape(n_img * 256, D)
    return x.float() + ctx @ wp.float().t() + bp, ctx, lse.reshape(n_img * H, 256)


def main():
    dev = torch.device("cuda:0")
    n_img = int(os.environ.get("N_IMG", 256)); D, H = 384, 6
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

    def three(save=True):
        qkv = ops.linear_fwd(ln1, wq, bias=bq, alpha=scale, alpha_cols=D)
        ctx, lse = ops.attn_fwd(qkv, n_img, H, D)
        return ops.linear_fwd(ctx, wp, bias=bp, resid=x), ctx, qkv, lse

    xm3, ctx3, qkv3, lse3 = three()
    xm1, ctx1, qkv1, lse1 = ops.attn_block_fwd(ln1, x, wq, bq, wp, bp, n_img, H, D, scale, save=True)
    xm0, ctx0, _, _ = ops.attn_block_fwd(ln1, x, wq, bq, wp, bp, n_img, H, D, scale, save=False)
    torch.cuda.synchronize()
    print("qkv bit-identical to the GEMM launch:", torch.equal(qkv1, qkv3))
    if not torch.equal(qkv1, qkv3):
        d = (qkv1.float() - qkv3.float()).abs()
        print("   max |diff|", d.max().item(), "mismatches", int((d > 0).sum()), "of", d.numel(),
              "per part", [int((d[:, i * D:(i + 1) * D] > 0).sum()) for i in range(3)])
    print("momentum form == online form (ctx, x_mid):", torch.equal(ctx0, ctx1), torch.equal(xm0, xm1))
    for name, a, b in (("ctx", ctx1, ctx3), ("x_mid", xm1, xm3), ("lse", lse1, lse3)):
        d = (a.float() - b.float()).abs()
        print(f"{name:6s} vs three launches: max |diff| {d.max().item():.4e}  mean {d.mean().item():.3e}  (max |value| {b.float().abs().max().item():.3f})")
    nref = min(n_img, 16)
    rx, rctx, rlse = reference(ln1[:nref * 256], x[:nref * 256], wq, bq, wp, bp, nref, H, D, scale)
    for name, a, b in (("fused x_mid", xm1[:nref * 256], rx), ("three x_mid", xm3[:nref * 256], rx), ("fused ctx", ctx1[:nref * 256], rctx),
                       ("three ctx", ctx3[:nref * 256], rctx), ("fused lse", lse1[:nref * H], rlse), ("three lse", lse3[:nref * H], rlse)):
        d = (a.float() - b).abs()
        print(f"{name:12s} vs fp32 torch: max |diff| {d.max().item():.4e}  mean {d.mean().item():.3e}")
    # run-to-run reproducibility
    again = ops.attn_block_fwd(ln1, x, wq, bq, wp, bp, n_img, H, D, scale, save=True)
    print("bit-reproducible run to run:", all(torch.equal(a, b) for a, b in zip(again, (xm1, ctx1, qkv1, lse1))))

    flop = 2.0 * R * D * 3 * D + 4.0 * R * 256 * D + 2.0 * R * D * D
    rows = [("three launches (online: same work)", lambda: three()),
            ("fused, momentum form", lambda: ops.attn_block_fwd(ln1, x, wq, bq, wp, bp, n_img, H, D, scale, save=False)),
            ("fused, online form", lambda: ops.attn_block_fwd(ln1, x, wq, bq, wp, bp, n_img, H, D, scale, save=True))]
    for name, fn in rows:
        med, best = timeit(fn)
        print(f"{name:40s} {med:8.1f} us (best {best:.1f})  {flop / med * 1e-6:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
