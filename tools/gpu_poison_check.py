"""Does any kernel of the training step read memory it (or an earlier kernel of the step) never wrote?

The same seeded trajectory is run three times in one process: (a) as is, (b) again (run-to-run determinism: the step uses no
atomics, so the meters must repeat bit for bit), (c) after every block the caching allocator holds has been filled with NaN
patterns (fp32 and bf16 quiet NaNs), so that a read of a `torch.empty` region the step never wrote shows up as a NaN or as a
changed meter instead of going unnoticed because the block happened to hold finite values.  Configurations: the tiny model in
the two MIM modes (one view / both views masked) and ViT-S at B = 8 (fused MLP chain, persistent GEMM tiles).

    gpurun -- 'python tools/gpu_poison_check.py'          (prints one line per configuration; exit code 1 on any difference)
"""
import dataclasses
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import dig_oracle as O                                                   # noqa: E402  (test infrastructure: seeded inputs / initial weights)
from gpu_util import build_model, run_engine_steps                        # noqa: E402


def poison():
    """Fill what the caching allocator has cached (and some more) with NaN bit patterns, then give it back."""
    torch.cuda.synchronize()
    held = []
    for nbytes, count in ((64 << 20, 48), (4 << 20, 256), (256 << 10, 1024), (16 << 10, 2048), (512, 4096)):
        for i in range(count):
            t = torch.empty(nbytes // 4, device="cuda:0", dtype=torch.int32)
            t.fill_(0x7FC07FC0 if i % 2 else 0x7FC00001)                  # two bf16 NaNs / one fp32 NaN
            held.append(t)
    torch.cuda.synchronize()
    del held


def trajectory(cfg, hp, seed, B, n):
    model = build_model(cfg, *O.det_state(cfg, seed))
    batches = [O.synthetic_batch(B, cfg, 9000 + s) for s in range(n)]
    stats, _ = run_engine_steps(model, batches, hp)
    torch.cuda.synchronize()
    keys = ("loss", "loss_pixel", "loss_contrast", "grad_norm")
    return [tuple(float(s[k]) for k in keys) for s in stats], float(model.flat_params.double().sum())


def main():
    sys.stdout = os.fdopen(os.dup(1), "w")
    devnull = open(os.devnull, "w")
    bad = 0
    cases = [("tiny", O.DiGConfig(**O.TINY), O.StepHyper(lr=1e-3), 4, 12),
             ("tiny both views masked", O.DiGConfig(**O.TINY), O.StepHyper(lr=1e-3, only_mim_on_ori_img=False), 4, 12),
             ("vit_small B=8", O.DiGConfig(), O.StepHyper(lr=1e-3), 8, 4)]
    for name, cfg, hp, B, n in cases:
        runs = []
        for mode in ("first", "repeat", "poisoned"):
            if mode == "poisoned":
                poison()
            out, sys.stdout = sys.stdout, devnull                         # (the engine prints its meters)
            try:
                runs.append(trajectory(cfg, hp, 23, B, n))
            finally:
                sys.stdout = out
        # what must repeat bit for bit: every step's gradient norm and the final weights.  The loss METERS are sums of per-block partials
        # added with one fp32 atomic per block (mse_fwd_bwd_kernel, ce_rows_kernel): their last bits depend on the arrival order
        state = [([row[3] for row in r[0]], r[1]) for r in runs]
        rep, poi = state[0] == state[1], state[0] == state[2]
        if name == cases[0][0]:
            first_state = state[0]
        meters = runs[0][0] == runs[1][0] == runs[2][0]
        finite = all(all(v == v for v in row) for row in runs[2][0])
        print(f"{name}: {n} steps; gradient norms + final weights: repeat identical {rep}, after NaN-poisoning the allocator's blocks "
              f"identical {poi}, finite {finite}; loss meters bit-identical {meters}", flush=True)
        if not (rep and poi and finite):
            bad += 1
            for s, (a, b, c) in enumerate(zip(*(r[0] for r in runs))):
                if not (a == b == c):
                    print(f"   step {s}: first {a}\n           repeat {b}\n           poisoned {c}", flush=True)
                    break
    # state that outlives a model (cached workspaces, streams, tile choices) must not change results: the first configuration again, after
    # the larger ones have run in this process
    name, cfg, hp, B, n = cases[0]
    out, sys.stdout = sys.stdout, devnull
    try:
        again = trajectory(cfg, hp, 23, B, n)
    finally:
        sys.stdout = out
    same = ([row[3] for row in again[0]], again[1]) == first_state
    print(f"{name} again after the other configurations: identical {same}", flush=True)
    sys.exit(1 if bad or not same else 0)


if __name__ == "__main__":
    main()
