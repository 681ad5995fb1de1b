"""Shader clock and package power of the step's kernel families, each looped alone for a few seconds (rocm-smi polled beside it):
which kernels pull the chip into its power limit, and at what clock each one really runs.

    gpurun -- 'python tools/gpu_kernel_power.py'
"""
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
from dig_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
R, D, Fh, H = 65536, 384, 1536, 6
g = torch.Generator(device="cpu").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g).to(dev)
x = rn(R, D).bfloat16(); res = rn(R, D).bfloat16()
w1 = (rn(Fh, D) * 0.05).bfloat16(); b1 = rn(Fh); w2 = (rn(D, Fh) * 0.03).bfloat16(); b2 = rn(D)
wq = (rn(3 * D, D) * 0.05).bfloat16(); bq = rn(3 * D)
g1, be1 = torch.ones(D, device=dev), torch.zeros(D, device=dev)
act = rn(R, Fh).bfloat16(); dy = rn(R, D).bfloat16(); dact = rn(R, Fh).bfloat16()
qkv = (rn(R, 3 * D) * 0.5).bfloat16()
ctx, lse = ops.attn_fwd(qkv, R // 256, H, D)
dW1 = torch.zeros(Fh, D, device=dev)
ln, mu, rs = ops.layernorm_fwd(x, g1, be1, 1e-6)
dg, db, dc = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
out = torch.empty_like(x)

KERNELS = [
    ("fused MLP forward (momentum form, + LayerNorms)", 4.0 * R * D * Fh, lambda: ops.mlp_chain_fwd_ln(x, g1, be1, 1e-6, w1, b1, w2, b2, g1, be1)),
    ("fused MLP forward (online form, + LayerNorms)", 4.0 * R * D * Fh, lambda: ops.mlp_chain_fwd_ln(x, g1, be1, 1e-6, w1, b1, w2, b2, g1, be1, save=True)),
    ("qkv forward GEMM (persistent 256x256)", 2.0 * R * D * 3 * D, lambda: ops.linear_fwd(x, wq, bias=bq, alpha=0.125, alpha_cols=D)),
    ("attention forward", 4.0 * 256 * 256 * 64 * (R // 256) * H, lambda: ops.attn_fwd(qkv, R // 256, H, D)),
    ("attention backward", 10.0 * 256 * 256 * 64 * (R // 256) * H, lambda: ops.attn_bwd(qkv, ctx, dy, lse, R // 256, H, D, 0.125, bias_sums=True)),
    ("fc1 data gradient (128x128)", 2.0 * R * D * Fh, lambda: ops.linear_dgrad(dact, w1)),
    ("fc2 data gradient x GELU'", 2.0 * R * D * Fh, lambda: ops.linear_dgrad(dy, w2, gelu_pre=act, colsum=True)),
    ("fc1 weight gradient (16 splits + slab sum)", 2.0 * R * D * Fh, lambda: ops.linear_wgrad(dact, ln, dW1)),
    ("LayerNorm backward", 0.0, lambda: ops.layernorm_bwd(dy, x, g1, be1, mu, rs, res, dg, db, out=out, dres_colsum=dc, defer=True)),
]


def poll(stop, samples):
    while not stop.is_set():
        try:
            t = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            m1, m2 = re.search(r"sclk clock level.*\((\d+)Mhz\)", t), re.search(r"Package Power \(W\): ([\d.]+)", t)
            if m1 and m2:
                samples.append((int(m1.group(1)), float(m2.group(1))))
        except Exception:  # noqa: BLE001
            pass
        time.sleep(0.3)


for name, flops, fn in KERNELS:
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, samples)); th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_end, n = time.time() + 5.0, 0
    e0.record()
    while time.time() < t_end:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    us = e0.elapsed_time(e1) / n * 1e3
    s = sorted(samples[len(samples) // 3:])                                   # (drop the ramp)
    clk = sorted(c for c, _ in s)[len(s) // 2] if s else 0
    pw = sorted(p for _, p in s)[len(s) // 2] if s else 0
    tf = f"{flops / us / 1e6:6.0f} TFLOP/s" if flops else "              "
    print(f"{name:48s} {us:7.1f} us  {tf}   sclk {clk:4d} MHz   {pw:6.0f} W   ({len(samples)} samples)", flush=True)
    time.sleep(2.0)
