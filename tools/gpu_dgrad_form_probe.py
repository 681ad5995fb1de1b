"""The two data-gradient GEMMs of an encoder block (proj: dctx = dx_mid Wproj, K = N = 384; qkv: dln1 = dqkv Wqkv, K = 1152, N = 384) in their
transpose-read form (dig_gemm_bf16 with trans_b = 1 on the [out, in] weight: what the step launches) against the direct form on a
K-contiguous copy W^T (which the optimizer launch can leave for free, dig_adamw_step_tr): microseconds per launch, R = 65536.
   python tools/gpu_dgrad_form_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dig_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 384                    # 384: ViT-S (R = 65 536 token rows at B = 128), 512: ViT-"Base" (R = 131 072 at B = 256)
R = 65536 if D == 384 else 131072


def timeit(fn, n=60):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, K in (("proj", D), ("qkv", 3 * D)):
    dy = torch.randn(R, K, device=dev).bfloat16()
    w = (torch.randn(K, D, device=dev) * 0.05).bfloat16()            # [out = K, in = D] as the Linear stores it
    wt = w.t().contiguous()                                           # [in = D, out = K]: K-contiguous for the data gradient
    out1 = torch.empty(R, D, device=dev, dtype=torch.bfloat16)
    out2 = torch.empty_like(out1)
    t_tr = timeit(lambda: ops.linear_dgrad(dy, w, out=out1))
    res = {}
    for code in ((0, 264, 244, 544, 64) if D == 384 else (0, 244, 544, 64)):
        try:
            t = timeit(lambda: ops.gemm(dy, wt, R, D, K, out=out2, bk=code) if code else ops.linear_fwd(dy, wt, out=out2))
            err = ((out1.float() - out2.float()).norm() / out1.float().norm()).item()
            res[code] = (round(t, 1), f"{err:.1e}")
        except Exception as e:  # noqa: BLE001
            res[code] = str(e)[:40]
    fl = 2.0 * R * D * K
    print(f"{name}: transpose-read form {t_tr:.1f} us ({fl / t_tr / 1e6:.0f} TFLOP/s); direct form on W^T by tile code (us, rel. diff): {res}")


# ---- the MLP's two data gradients where they are separate launches (D = 512: no fused MLP backward): fc2 with the GELU' epilogue and the
# fc1 bias column sums, then fc1
if D != 384:
    Fh = 4 * D
    dy = torch.randn(R, D, device=dev).bfloat16()
    pre = torch.randn(R, Fh, device=dev).bfloat16()
    w2 = (torch.randn(D, Fh, device=dev) * 0.03).bfloat16()           # fc2.weight [out = D, in = F]
    w2t = w2.t().contiguous()                                         # [F, D]: K (= D) contiguous for dact[R, F] = dy w2
    t_tr = timeit(lambda: ops.linear_dgrad(dy, w2, gelu_pre=pre, colsum=True), 20)
    a1, p1 = ops.linear_dgrad(dy, w2, gelu_pre=pre, colsum=True)
    res = {}
    for code in (0, 244, 544, 64):
        try:
            parts = torch.empty_like(p1)
            out2 = torch.empty_like(a1)
            fn = lambda: ops.gemm(dy, w2t, R, Fh, D, out=out2, act=2, resid=pre, bk=code or ops.dgrad_gelu_tile_code(R, Fh), colsum_partials=parts)  # noqa: E731
            t = timeit(fn, 20)
            res[code] = (round(t, 1), f"{((a1.float() - out2.float()).norm() / a1.float().norm()).item():.1e}", f"{((p1 - parts).norm() / p1.norm()).item():.1e}")
        except Exception as e:  # noqa: BLE001
            res[code] = str(e)[:60]
    print(f"fc2 + GELU': transpose-read form {t_tr:.1f} us ({2.0 * R * D * Fh / t_tr / 1e6:.0f} TFLOP/s); direct form by tile code: {res}")
    dact = torch.randn(R, Fh, device=dev).bfloat16()
    w1 = (torch.randn(Fh, D, device=dev) * 0.03).bfloat16()           # fc1.weight [out = F, in = D]
    w1t = w1.t().contiguous()                                         # [D, F]
    o1 = torch.empty(R, D, device=dev, dtype=torch.bfloat16)
    o2 = torch.empty_like(o1)
    t_tr = timeit(lambda: ops.linear_dgrad(dact, w1, out=o1), 20)
    res = {}
    for code in (0, 244, 544, 64):
        try:
            t = timeit(lambda: ops.gemm(dact, w1t, R, D, Fh, out=o2, bk=code) if code else ops.linear_fwd(dact, w1t, out=o2), 20)
            res[code] = (round(t, 1), f"{((o1.float() - o2.float()).norm() / o1.float().norm()).item():.1e}")
        except Exception as e:  # noqa: BLE001
            res[code] = str(e)[:60]
    print(f"fc1: transpose-read form {t_tr:.1f} us ({2.0 * R * D * Fh / t_tr / 1e6:.0f} TFLOP/s); direct form by tile code: {res}")
