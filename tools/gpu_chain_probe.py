"""Timing of the fused MLP chain kernels (csrc/mlp_chain.hip) against the two-GEMM path they replace, ViT-S shapes (R = 2 B 256 rows,
D = 384, F = 1536), alone on the GPU.  Prints one line per variant: microseconds per launch (median of `reps` timed batches) and the
effective TFLOP/s on the algorithmic 4 R D F FLOP."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd import ops  # noqa: E402


def timeit(fn, iters=20, reps=5):
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
    return ts[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    R = int(os.environ.get("R", 65536)); D = 384; Fh = 1536
    torch.manual_seed(0)
    x = torch.randn(R, D, device=dev).bfloat16()
    w1 = (torch.randn(Fh, D, device=dev) * 0.06).bfloat16(); b1 = torch.randn(Fh, device=dev) * 0.5
    w2 = (torch.randn(D, Fh, device=dev) * 0.04).bfloat16(); b2 = torch.randn(D, device=dev) * 0.5
    res = torch.randn(R, D, device=dev).bfloat16()
    dy = torch.randn(R, D, device=dev).bfloat16()
    w2t, w1t = ops.transpose_bf16(w2), ops.transpose_bf16(w1)
    flop = 4.0 * R * D * Fh
    pre = torch.empty(R, Fh, device=dev, dtype=torch.bfloat16)

    def unfused_mom():
        a = ops.linear_fwd(x, w1, bias=b1, act=1)
        return ops.linear_fwd(a, w2, bias=b2, resid=res)

    def unfused_online():
        a = ops.linear_fwd(x, w1, bias=b1, act=1, pre=pre)
        return ops.linear_fwd(a, w2, bias=b2, resid=res)

    def unfused_bwd():
        dact, parts = ops.linear_dgrad(dy, w2, gelu_pre=pre, colsum=True)
        return ops.linear_dgrad(dact, w1)

    gam = 1.0 + 0.3 * torch.randn(D, device=dev); bet = torch.zeros(D, device=dev)
    _, mu, rs = ops.layernorm_fwd(res, gam, bet, 1e-6)

    wp = (torch.randn(D, D, device=dev) * 0.05).bfloat16()
    wpt = ops.transpose_bf16(wp)

    def chain_ln_then_proj():
        dxm = ops.mlp_chain_bwd_ln(dy, w2t, pre, w1t, res, gam, mu, rs)[0]
        return ops.linear_dgrad(dxm, wp)

    def chain_then_ln():
        dln2, dpre, parts = ops.mlp_chain_bwd(dy, w2t, pre, w1t)
        return ops.layernorm_bwd(dln2, res, gam, bet, mu, rs, dy, None, None, out=dln2, defer=True)

    unfused_online()
    rows = [("fwd momentum: fc1+gelu, fc2+res (2 launches)", unfused_mom),
            ("fwd momentum: chain", lambda: ops.mlp_chain_fwd(x, w1, b1, w2, b2, res)),
            ("fwd online: fc1+gelu+pre, fc2+res (2 launches)", unfused_online),
            ("fwd online: chain + pre/act", lambda: ops.mlp_chain_fwd(x, w1, b1, w2, b2, res, save=True)),
            ("bwd: fc2 dgrad*gelu'+colsum, fc1 dgrad (2 launches)", unfused_bwd),
            ("bwd: chain", lambda: ops.mlp_chain_bwd(dy, w2t, pre, w1t)),
            ("bwd: chain, then norm2's backward (2 launches)", chain_then_ln),
            ("bwd: chain with norm2's backward in it", lambda: ops.mlp_chain_bwd_ln(dy, w2t, pre, w1t, res, gam, mu, rs)),
            ("bwd: chain + norm2's backward, then proj dgrad (2 launches)", chain_ln_then_proj),
            ("bwd: chain + norm2's backward + proj dgrad", lambda: ops.mlp_chain_bwd_ln(dy, w2t, pre, w1t, res, gam, mu, rs, projt=wpt))]
    for name, fn in rows:
        us = timeit(fn)
        print(f"{name:58s} {us:8.1f} us  {flop / us * 1e-6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
