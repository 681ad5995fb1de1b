"""End-to-end parity of one engine step (HIP path) against the CPU oracle on the tiny config (GPU box)."""
import dataclasses
import sys
import types

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "oracle")
import dig_oracle as O
from dig_amd.modeling_pretrain_moco_mim_ori import MoCo_ViT
from dig_amd.optim_factory import create_optimizer
from dig_amd.engine_for_pretraining_moco import train_one_epoch
from dig_amd.utils import NativeScalerWithGradNormCount

dev = torch.device("cuda:0")
cfg = O.DiGConfig(**O.TINY)
seed, B = 3, 4
P, S = O.det_state(cfg, seed)
model = MoCo_ViT(encoder_embed_dim=cfg.embed_dim, encoder_depth=cfg.depth, encoder_num_heads=cfg.heads,
                 decoder_embed_dim=cfg.dec_dim, mlp_dim=cfg.moco_mlp_dim, dim=cfg.moco_dim, T=cfg.T, num_windows=4,
                 use_pixel_target=True, patchnet_name='no_patchtrans')
names = [n for n, _ in model.named_parameters()]
assert names == list(P.keys()), "parameter order differs"
model.load_state_dict({**P, **S})
model.to(dev)
hp = O.StepHyper(lr=1e-3)
args = types.SimpleNamespace(num_view=2, moco_m=hp.moco_m, use_moco_m_cos=1, epochs=10, contrast_start_epoch=0,
                             contrast_warmup_steps=0, loss_weight_contrast=hp.w_contrast, loss_weight_pixel=hp.w_pixel,
                             only_mim_on_ori_img=True, eval_freq=500, opt='adamw', lr=hp.lr, weight_decay=hp.weight_decay,
                             opt_eps=hp.eps, opt_betas=None)
opt = create_optimizer(args, model)
scaler = NativeScalerWithGradNormCount()
im, au, mk = O.synthetic_batch(B, cfg, seed * 1000)
loader = [([im, au, mk], torch.ones(1), torch.ones(1))]
stats = train_one_epoch(model, None, None, loader, None, opt, dev, 0, scaler, None, patch_size=4, normlize_target=False,
                        start_steps=0, lr_schedule_values=np.full(4, hp.lr), wd_schedule_values=np.full(4, hp.weight_decay), args=args)
grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}

tr = O.OracleTrainer(cfg, *O.det_state(cfg, seed))
metrics, ograds, out, labels = tr.step(im, au, mk, dataclasses.replace(hp, moco_m=O.adjust_moco_momentum(0.0, 10, hp.moco_m)))
print("engine:", {k: round(v, 5) for k, v in stats.items()})
print("oracle:", {k: round(v, 5) for k, v in metrics.items()})
ok = True
for k in ("loss", "loss_pixel", "loss_contrast", "grad_norm"):
    e = abs(stats[k] - metrics[k]) / abs(metrics[k])
    ok &= e < 2e-2
    print(f"{k}: rel err {e:.2e}")
worst = []
for n, g in grads.items():
    og = ograds[n]
    cos = torch.nn.functional.cosine_similarity(g.reshape(1, -1), og.reshape(1, -1)).item() if og.norm() > 1e-7 else 1.0
    ratio = (g.norm() / (og.norm() + 1e-12)).item() if og.norm() > 1e-7 else 1.0
    worst.append((cos, ratio, n, og.norm().item()))
worst.sort()
for w in worst[:12]:
    print("cos %.5f ratio %.4f %s (|g|=%.3e)" % w)
bad = [w for w in worst if w[0] < 0.99 or abs(w[1] - 1) > 0.05]
print("n_bad", len(bad), "of", len(worst))
ok &= len(bad) == 0
post = {n: p.detach().float().cpu() for n, p in model.named_parameters()}
e_m = max((post[n] - tr.P[n]).abs().max().item() for n in post if not O.is_trainable(n))
print("momentum params max abs diff", e_m)
print("STEP_PARITY_OK" if ok else "STEP_PARITY_FAIL")
