import ctypes
import types

import numpy as np
import torch

import dig_oracle as O


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-20)).item()


def engine_args(hp, epochs=10):
    return types.SimpleNamespace(num_view=2, moco_m=hp.moco_m, use_moco_m_cos=1, epochs=epochs, contrast_start_epoch=0,
                                 contrast_warmup_steps=0, loss_weight_contrast=hp.w_contrast, loss_weight_pixel=hp.w_pixel,
                                 only_mim_on_ori_img=bool(getattr(hp, "only_mim_on_ori_img", True)), eval_freq=500, opt='adamw', lr=hp.lr, weight_decay=hp.weight_decay,
                                 opt_eps=hp.eps, opt_betas=None)


def build_model(cfg, P=None, S=None, device="cuda:0", hp=None):
    """hp (an oracle StepHyper): its drop_path rate and drop_seed (stochastic depth, --drop_path) go to the model."""
    from dig_amd.modeling_pretrain_moco_mim_ori import MoCo_ViT
    m = MoCo_ViT(encoder_embed_dim=cfg.embed_dim, encoder_depth=cfg.depth, encoder_num_heads=cfg.heads,
                 decoder_embed_dim=cfg.dec_dim, mlp_dim=cfg.moco_mlp_dim, dim=cfg.moco_dim, T=cfg.T, num_windows=cfg.num_windows,
                 use_pixel_target=cfg.use_pixel, use_moco_target=cfg.use_moco, patchnet_name=getattr(cfg, "patchnet", "no_patchtrans"),
                 drop_path_rate=float(getattr(hp, "drop_path", 0.0) or 0.0))
    if hp is not None and getattr(hp, "drop_path", 0.0):
        m.drop_seed = int(hp.drop_seed)
    if P is not None:
        m.load_state_dict({**P, **S})
    return m.to(torch.device(device))


def run_engine_steps(model, batches, hp, start=0, opt=None):
    from dig_amd.optim_factory import create_optimizer
    from dig_amd.engine_for_pretraining_moco import train_one_epoch
    from dig_amd.utils import NativeScalerWithGradNormCount
    args = engine_args(hp)
    opt = opt or create_optimizer(args, model)
    out = []
    n = len(batches)
    for s, (im, au, mk) in enumerate(batches):
        st = train_one_epoch(model, None, None, [([im, au, mk], torch.ones(1), torch.ones(1))], None, opt, torch.device("cuda:0"),
                             start + s, NativeScalerWithGradNormCount(), getattr(hp, "clip_grad", None), patch_size=4, normlize_target=False,
                             start_steps=start + s, lr_schedule_values=np.full(start + n + 2, hp.lr),
                             wd_schedule_values=np.full(start + n + 2, hp.weight_decay), args=args)
        out.append(st)
    return out, opt
