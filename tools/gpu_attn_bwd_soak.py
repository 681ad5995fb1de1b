"""Race screen of the attention backward (csrc/attention.hip, round 6): the restage now mixes an LDS-DMA of the even K / V blocks with register ->
LDS writes of the odd ones, both query blocks' fragments are lifted out of LDS before it, and the results leave through a wave-private LDS staging
area -- orderings that a missing wait would break only sometimes.  So: many launches on changing data and image counts, a second stream hammering
HBM beside every third one (uneven load moves the DMA landing times), and every dqkv compared BIT FOR BIT with the same launch from a build of the
plain data path (build/lab/libattn_ref.so: -DDIG_ATTN_LIFT=1 -DDIG_ATTN_BWD_STORE=0: all of K and V restaged by the LDS-DMA, 16-byte row stores, sums
from the accumulators; LIFT=0 does not build beside the fused projection form, which has no d(ctx) rows to re-read);
the v_bias sums bit for bit as well, the q_bias sums to 2e-3 (bf16 rows against fp32 accumulators) and bit for bit run to run; the fused
projection form (dig_attn_bwd_proj) bit for bit run to run and to 4e-3 against the pair.
usage: python tools/gpu_attn_bwd_soak.py [iterations]      (builds the reference library with hipcc if it is missing)"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dig_amd import ops  # noqa: E402

REF = os.path.join(ROOT, "build", "lab", "libattn_ref.so")
SRC = [os.path.join(ROOT, "dig_amd", "csrc", f) for f in ("attention.hip", "attn_tiles.h")]
if not os.path.exists(REF) or any(os.path.getmtime(REF) < os.path.getmtime(f) for f in SRC):
    os.makedirs(os.path.dirname(REF), exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", "-ffp-contract=fast",
                    "-munsafe-fp-atomics", "-w", "-DDIG_ATTN_LIFT=1", "-DDIG_ATTN_BWD_STORE=0", "-I", os.path.join(ROOT, "include"),
                    SRC[0], os.path.join(ROOT, "dig_amd", "csrc", "probe.hip"), "-o", REF], check=True)
ref = ctypes.CDLL(REF)
ref.dig_attn_bwd.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20))


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda:0")
    noise = torch.cuda.Stream()
    junk_a, junk_b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)
    bad = 0
    for it in range(iters):
        H = (6, 8, 2)[it % 3]
        D = 64 * H
        n_img = (1, 7, 64, 256, 300)[it % 5] if H == 6 else (3, 40, 128)[it % 3]
        R = n_img * 256
        g = torch.Generator(device=dev).manual_seed(5000 + it)
        rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc)
        qkv, dy = rn(R, 3 * D).bfloat16(), rn(R, D).bfloat16()
        wp = rn(D, D, sc=D ** -0.5).bfloat16()
        ctx, lse = ops.attn_fwd(qkv, n_img, H, D)
        dctx = (dy.float() @ wp.float()).bfloat16()
        projt = wp.t().contiguous()
        if it % 3 == 0:
            with torch.cuda.stream(noise):
                for _ in range(4):
                    junk_b.copy_(junk_a)
        d1, q1, v1 = ops.attn_bwd(qkv, ctx, dctx, lse, n_img, H, D, 0.125, bias_sums=True)
        d2, q2, v2 = ops.attn_bwd(qkv, ctx, dctx, lse, n_img, H, D, 0.125, bias_sums=True)
        d0 = torch.empty_like(qkv)
        q0, v0 = torch.empty_like(q1), torch.empty_like(v1)
        torch.cuda.current_stream().synchronize()
        rc = ref.dig_attn_bwd(qkv.data_ptr(), ctx.data_ptr(), dctx.data_ptr(), lse.data_ptr(), d0.data_ptr(), n_img, H, D, 0.125, q0.data_ptr(),
                              v0.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        dn = ops.attn_bwd(qkv, ctx, dctx, lse, n_img, H, D, 0.125)                       # (no bias sums: another instantiation of the epilogue)
        p1, pq, pv = ops.attn_bwd_proj(qkv, ctx, dy, projt, lse, n_img, H, D, 0.125, bias_sums=True) if D % 128 == 0 else (None, None, None)
        p2 = ops.attn_bwd_proj(qkv, ctx, dy, projt, lse, n_img, H, D, 0.125) if p1 is not None else None
        torch.cuda.synchronize()
        ok = (torch.equal(d1, d0) and torch.equal(d2, d0) and torch.equal(dn, d0) and torch.equal(q1, q2) and torch.equal(v1, v2)
              and rel(q1, q0) < 2e-3 and rel(v1, v0) < 2e-3)
        if p1 is not None:
            ok = ok and torch.equal(p1, p2) and rel(p1, d0) < 4e-3 and rel(pq, q0) < 6e-3 and rel(pv, v0) < 6e-3
        if not ok:
            bad += 1
            print(f"iteration {it}: MISMATCH (n_img {n_img}, heads {H}): dqkv {torch.equal(d1, d0)} {torch.equal(d2, d0)} {torch.equal(dn, d0)} "
                  f"sums {torch.equal(q1, q2)} {torch.equal(v1, v2)} {rel(q1, q0):.1e} {rel(v1, v0):.1e}"
                  + (f" proj {torch.equal(p1, p2)} {rel(p1, d0):.1e}" if p1 is not None else ""))
    print(f"{iters} iterations, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
