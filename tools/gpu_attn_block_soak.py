"""Race screen of the fused attention sub-block (csrc/attn_block.hip): its LDS ring is ordered by COUNTED s_waitcnt vmcnt(N) waits, and a count
that is one too high would read a slot before its DMA has landed -- only sometimes.  So: many launches on changing data, with a second stream
hammering HBM beside it (uneven load moves the DMA landing times), every output compared bit for bit with
  (a) the same launch from a build whose every counted wait is vmcnt(0) (build/lab/libab_safe.so: -DDIG_AB_SAFE_WAITS=1), and
  (b) for the q | k | v rows, the dig_gemm_bf16 launch of the product library.
usage: python tools/gpu_attn_block_soak.py [iterations]      (builds the safe-wait library with hipcc if it is missing)"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dig_amd import ops  # noqa: E402

SAFE = os.path.join(ROOT, "build", "lab", "libab_safe.so")
if not os.path.exists(SAFE) or os.path.getmtime(SAFE) < os.path.getmtime(os.path.join(ROOT, "dig_amd", "csrc", "attn_block.hip")):
    os.makedirs(os.path.dirname(SAFE), exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", "-ffp-contract=fast",
                    "-munsafe-fp-atomics", "-w", "-DDIG_AB_SAFE_WAITS=1", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "dig_amd", "csrc", "attn_block.hip"), os.path.join(ROOT, "dig_amd", "csrc", "probe.hip"), "-o", SAFE], check=True)
safe = ctypes.CDLL(SAFE)
safe.dig_attn_block_fwd.argtypes = [ctypes.c_void_p] * 10 + [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_void_p]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda:0")
    D, H, scale = 384, 6, 0.125
    noise_stream = torch.cuda.Stream()
    junk_a, junk_b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)
    bad = 0
    for it in range(iters):
        n_img = (1, 7, 64, 256, 300)[it % 5]
        R = n_img * 256
        g = torch.Generator(device=dev).manual_seed(1000 + it)
        rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc)
        ln1, x = rn(R, D).bfloat16(), rn(R, D).bfloat16()
        wq, wp = rn(3 * D, D, sc=0.05).bfloat16(), rn(D, D, sc=0.05).bfloat16()
        bq, bp = rn(3 * D, sc=0.3), rn(D, sc=0.3)
        if it % 3 == 0:                                                   # uneven load beside the launch
            with torch.cuda.stream(noise_stream):
                for _ in range(4):
                    junk_b.copy_(junk_a)
        save = it % 2 == 0
        xm, ctx, qkv, lse = ops.attn_block_fwd(ln1, x, wq, bq, wp, bp, n_img, H, D, scale, save=save)
        xm2, ctx2 = torch.empty_like(xm), torch.empty_like(ctx)
        qkv2 = torch.empty_like(qkv) if save else None
        lse2 = torch.empty_like(lse) if save else None
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        rc = safe.dig_attn_block_fwd(p(ln1), p(x), p(wq), p(bq), p(wp), p(bp), p(qkv2), p(ctx2), p(lse2), p(xm2), n_img, H, D, scale,
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        ok = torch.equal(xm, xm2) and torch.equal(ctx, ctx2)
        if save:
            ok = ok and torch.equal(qkv, qkv2) and torch.equal(lse, lse2) and torch.equal(qkv, ops.linear_fwd(ln1, wq, bias=bq, alpha=scale, alpha_cols=D))
        if not ok:
            bad += 1
            print(f"iteration {it} (n_img {n_img}, save {save}): MISMATCH", flush=True)
    torch.cuda.synchronize()
    print(f"{iters} launches, {bad} mismatches against the vmcnt(0) build / the GEMM launch")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
