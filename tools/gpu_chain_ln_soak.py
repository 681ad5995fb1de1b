"""Race screen of the phases behind the fused MLP backward's main loop (dig_mlp_chain_bwd_ln / _bwd_ln_proj, csrc/mlp_chain.hip): the tile stores,
the lane-swap reductions and the LDS hand-overs between the role split, the LayerNorm phase and the projection phase are ordered by explicit waits
and workgroup barriers -- a missing one shows as a result that depends on timing.  Many launches at the step's shape while another stream keeps
the chip unevenly busy; every result bit for bit against the first launch's.
usage: python tools/gpu_chain_ln_soak.py [iterations]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd import ops  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda:0")
    R, D, Fh = 65536, 384, 1536
    torch.manual_seed(0)
    bf = lambda *s, k=1.0: (torch.randn(*s, device=dev) * k).bfloat16()
    dy, x_mid, pre = bf(R, D), bf(R, D, k=1.3), bf(R, Fh)
    w2t, w1t, projt = bf(Fh, D, k=0.04), bf(D, Fh, k=0.06), bf(D, D, k=0.05)
    gam = 1.0 + 0.3 * torch.randn(D, device=dev)
    _, mu, rs = ops.layernorm_fwd(x_mid, gam, torch.zeros(D, device=dev), 1e-6)
    side = torch.cuda.Stream()
    a, b = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16), torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    bad = 0
    for proj in (False, True):
        ref = ops.mlp_chain_bwd_ln(dy, w2t, pre, w1t, x_mid, gam, mu, rs, projt=projt if proj else None)
        torch.cuda.synchronize()
        for i in range(iters):
            with torch.cuda.stream(side):                              # uneven background load: 0..3 GEMMs of varying size
                for _ in range(i % 4):
                    n = 1024 * (1 + (i * 7) % 4)
                    torch.mm(a[:n], b)
            out = ops.mlp_chain_bwd_ln(dy, w2t, pre, w1t, x_mid, gam, mu, rs, projt=projt if proj else None)
            ok = all(torch.equal(u, v) for u, v in zip(out, ref))
            if not ok:
                bad += 1
                print(f"mismatch: launch {i}, proj={proj}: " + ", ".join(str(int((u != v).sum())) for u, v in zip(out, ref)), flush=True)
        torch.cuda.synchronize()
        print(f"{'dig_mlp_chain_bwd_ln_proj' if proj else 'dig_mlp_chain_bwd_ln'}: {iters} launches under load, mismatches so far: {bad}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
