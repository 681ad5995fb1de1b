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
R, D = 65536, 384


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
    for code in (0, 264, 244, 544, 64):
        try:
            t = timeit(lambda: ops.gemm(dy, wt, R, D, K, out=out2, bk=code) if code else ops.linear_fwd(dy, wt, out=out2))
            err = ((out1.float() - out2.float()).norm() / out1.float().norm()).item()
            res[code] = (round(t, 1), f"{err:.1e}")
        except Exception as e:  # noqa: BLE001
            res[code] = str(e)[:40]
    fl = 2.0 * R * D * K
    print(f"{name}: transpose-read form {t_tr:.1f} us ({fl / t_tr / 1e6:.0f} TFLOP/s); direct form on W^T by tile code (us, rel. diff): {res}")
