"""--patchnet_name regular on the GPU against the fp32 oracle, tensor by tensor (patch_extractor.* and what it feeds): cosine and norm ratio
of every gradient beside the oracle's own bf16-autocast run.   python tools/gpu_patchnet_probe.py [B]"""
import dataclasses
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import dig_oracle as O  # noqa: E402
from gpu_util import build_model, run_engine_steps  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = dataclasses.replace(O.DiGConfig(**O.TINY), patchnet="regular", num_windows=5)
seed = 35
hp = O.StepHyper(lr=1e-3)
im, au, mk = O.synthetic_batch(B, cfg, seed * 1000)
model = build_model(cfg, *O.det_state(cfg, seed))
(st,), _ = run_engine_steps(model, [(im, au, mk)], hp)
grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}
hp0 = dataclasses.replace(hp, moco_m=O.adjust_moco_momentum(0.0, 10, hp.moco_m))
ref_m, ref_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
with torch.autocast("cpu", dtype=torch.bfloat16):
    _, bf_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
print({k: (round(st[k], 5), round(ref_m[k], 5)) for k in ("loss", "loss_pixel", "loss_contrast", "grad_norm")})
cos = torch.nn.functional.cosine_similarity
for n, g in grads.items():
    if not (n.startswith(("patch_extractor", "pix_projector.0", "encoder_projection_layer.0")) or n.endswith("blocks.1.mlp.fc2.weight")):
        continue
    r = ref_g[n].reshape(1, -1)
    c, cb = cos(g.reshape(1, -1), r).item(), cos(bf_g[n].float().reshape(1, -1), r).item()
    q, qb = (g.norm() / r.norm()).item(), (bf_g[n].float().norm() / r.norm()).item()
    print(f"{n:56s} cos {c:.5f} (bf16 {cb:.5f})  |g|/|ref| {q:.4f} (bf16 {qb:.4f})  |ref| {float(r.norm()):.3e}")
