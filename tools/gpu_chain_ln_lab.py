"""Lab of norm2's backward inside the fused MLP backward launch (dig_mlp_chain_bwd_ln, csrc/mlp_chain.hip): the launch alone at the step's
shape (R = 65536, F = 1536) from builds of mlp_chain.hip with -DDIG_CHAIN_LNB_ABL=<mask> (1: dy re-read from x_mid's addresses, 2: no row loads,
4: no dx_mid stores; projection phase: 8 no MFMAs, 16 no weight DMA, 32 no dctx stores), next to dig_mlp_chain_bwd (no LayerNorm phase) from the same build.  What the phase costs and what bounds it.
usage: python tools/gpu_chain_ln_lab.py [masks ...]     (default: 0 1 2 4 6)"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LAB = os.path.join(ROOT, "build", "lab")
STUB = "#include <hip/hip_runtime.h>\nbool dig_probe_on() { return false; }\nvoid dig_probe_events(hipEvent_t*, hipEvent_t*) {}\n"


def build(mask):
    os.makedirs(LAB, exist_ok=True)
    so = os.path.join(LAB, f"libchain_lnb_{mask}.so")
    src = os.path.join(ROOT, "dig_amd", "csrc", "mlp_chain.hip")
    if os.path.exists(so) and os.path.getmtime(so) > os.path.getmtime(src):
        return so
    stub = os.path.join(LAB, "probe_stub.hip")
    with open(stub, "w") as f:
        f.write(STUB)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize", "-ffp-contract=fast",
                    "-munsafe-fp-atomics", "-w", f"-DDIG_CHAIN_LNB_ABL={mask}", "-I", os.path.join(ROOT, "include"), src, stub, "-o", so], check=True)
    return so


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
    return sorted(ts)[len(ts) // 2]


def main():
    masks = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 6]
    if not torch.cuda.is_available():                                  # build container: compile the variants so that they travel with the snapshot
        for m in masks:
            print(build(m))
        return
    dev = torch.device("cuda:0")
    R, D, Fh = int(os.environ.get("R", 65536)), 384, 1536
    torch.manual_seed(0)
    bf = lambda *s, k=1.0: (torch.randn(*s, device=dev) * k).bfloat16()
    dy, x_mid, pre = bf(R, D), bf(R, D), bf(R, Fh)
    w2t, w1t = bf(Fh, D, k=0.04), bf(D, Fh, k=0.06)
    gam = 1.0 + 0.3 * torch.randn(D, device=dev)
    mu, rs = torch.randn(R, device=dev) * 0.1, 1.0 + 0.1 * torch.rand(R, device=dev)
    dpre, dx = torch.empty(R, Fh, device=dev, dtype=torch.bfloat16), torch.empty(R, D, device=dev, dtype=torch.bfloat16)
    parts = torch.empty((R + 127) // 128 * 4, Fh, device=dev)
    lnp = torch.empty((R + 127) // 128, 3, D, device=dev)
    projt, dctx = bf(D, D, k=0.05), torch.empty(R, D, device=dev, dtype=torch.bfloat16)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for m in masks:
        lib = ctypes.CDLL(build(m))

        def fused():
            rc = lib.dig_mlp_chain_bwd_ln(vp(dy), vp(w2t), vp(pre), vp(w1t), vp(dpre), vp(x_mid), vp(gam), vp(mu), vp(rs), vp(dx), vp(parts), vp(lnp),
                                          R, D, Fh, st)
            assert rc == 0, rc

        def fused_proj():
            rc = lib.dig_mlp_chain_bwd_ln_proj(vp(dy), vp(w2t), vp(pre), vp(w1t), vp(dpre), vp(x_mid), vp(gam), vp(mu), vp(rs), vp(dx), vp(parts),
                                               vp(lnp), vp(projt), vp(dctx), R, D, Fh, st)
            assert rc == 0, rc

        def plain():
            rc = lib.dig_mlp_chain_bwd(vp(dy), vp(w2t), vp(pre), vp(w1t), vp(dpre), vp(dx), vp(parts), R, D, Fh, st)
            assert rc == 0, rc
        print(f"LNB_ABL={m:2d}: dig_mlp_chain_bwd_ln_proj {timeit(fused_proj):7.1f} us   dig_mlp_chain_bwd_ln {timeit(fused):7.1f} us   "
              f"dig_mlp_chain_bwd {timeit(plain):7.1f} us", flush=True)


if __name__ == "__main__":
    main()
