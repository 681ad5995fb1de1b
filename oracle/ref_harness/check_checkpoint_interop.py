"""Row N2 (checkpoint compatibility both ways), checked with the reference's OWN code in the build container (CPU):

  A. reference model + custom_optim AdamW after one real reference step  --reference utils.save_model-->  checkpoint-0.pth
     --dig_amd.utils.auto_load_model-->  dig_amd MoCo_ViT + FusedAdamW:  every parameter / buffer / optimizer moment equal.
  B. that dig_amd model + optimizer  --dig_amd.utils.save_model-->  checkpoint-1.pth  --reference utils.auto_load_model-->
     a fresh reference model + optimizer:  state_dict and optimizer.state_dict() equal to the originals of A.

    python oracle/ref_harness/check_checkpoint_interop.py        # needs /root/reference; prints OK lines, exits non-zero on mismatch
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)
import dig_oracle as O
import refenv
import gen_golden as G


def main():
    cfg = O.DiGConfig(**O.TINY)
    hp = O.StepHyper(lr=1e-3)
    ref_utils = refenv.setup()
    import engine_for_pretraining_moco as E
    import optim_factory as ref_optim_factory

    def fresh_ref():
        m = G.build_ref_model(cfg)
        return m, ref_optim_factory.create_optimizer(G.ref_args(hp), m)

    model, opt = fresh_ref()
    P, S = O.det_state(cfg, 3)
    sd = model.state_dict()
    for k, v in {**P, **S}.items():
        sd[k].copy_(v)
    im, au, mk = O.synthetic_batch(2, cfg, 5)
    E.train_one_epoch(model, None, None, [([im, au, mk], torch.ones(1), torch.ones(1))], None, opt, torch.device('cpu'), 0,
                      G.ScalerCPU(ref_utils), None, patch_size=cfg.patch, normlize_target=False, start_steps=0,
                      lr_schedule_values=np.full(2, hp.lr), wd_schedule_values=np.full(2, hp.weight_decay), args=G.ref_args(hp))
    with tempfile.TemporaryDirectory() as tmp:
        args = types.SimpleNamespace(output_dir=tmp, auto_resume=True, resume="", model="pretrain_simmim_moco_ori_vit_small_patch4_32x128",
                                     start_epoch=0, model_ema=False)
        # ---- A: reference writes, dig_amd reads
        ref_utils.save_model(args=args, epoch=0, model=model, model_without_ddp=model, optimizer=opt, loss_scaler=G.ScalerCPU(ref_utils))
        from dig_amd import utils as U
        from dig_amd.modeling_pretrain_moco_mim_ori import MoCo_ViT
        from dig_amd.optim_factory import create_optimizer
        mine = MoCo_ViT(encoder_embed_dim=cfg.embed_dim, encoder_depth=cfg.depth, encoder_num_heads=cfg.heads, decoder_embed_dim=cfg.dec_dim,
                        mlp_dim=cfg.moco_mlp_dim, dim=cfg.moco_dim, T=cfg.T, num_windows=cfg.num_windows, use_pixel_target=True,
                        patchnet_name='no_patchtrans')
        my_args = types.SimpleNamespace(opt="adamw", lr=hp.lr, weight_decay=hp.weight_decay, opt_eps=hp.eps, opt_betas=None,
                                        output_dir=tmp, auto_resume=True, resume="", start_epoch=0)
        my_opt = create_optimizer(my_args, mine)
        U.auto_load_model(my_args, mine, mine, my_opt, U.NativeScalerWithGradNormCount())
        assert my_args.start_epoch == 1
        ref_sd, my_sd = model.state_dict(), mine.state_dict()
        assert list(ref_sd.keys()) == list(my_sd.keys())
        for k in ref_sd:
            assert ref_sd[k].dtype == my_sd[k].dtype and torch.equal(ref_sd[k], my_sd[k].cpu()), k
        ro, mo = opt.state_dict(), my_opt.state_dict()
        assert len(ro["state"]) == len(mo["state"]) and len(ro["param_groups"]) == len(mo["param_groups"])
        for i in ro["state"]:
            for f in ("exp_avg", "exp_avg_sq"):
                assert torch.equal(ro["state"][i][f], mo["state"][i][f].cpu()), (i, f)
            assert int(ro["state"][i]["step"]) == int(mo["state"][i]["step"])
        for g_r, g_m in zip(ro["param_groups"], mo["param_groups"]):
            assert g_r["params"] == g_m["params"] and g_r["weight_decay"] == g_m["weight_decay"] and g_r["lr"] == g_m["lr"]
        print("A OK: reference checkpoint -> dig_amd model + optimizer (", len(ref_sd), "tensors,", len(ro["state"]), "optimizer slots )")
        # ---- B: dig_amd writes, reference reads
        os.remove(os.path.join(tmp, "checkpoint-0.pth"))
        U.save_model(my_args, 1, mine, mine, my_opt, U.NativeScalerWithGradNormCount())
        model2, opt2 = fresh_ref()
        _load = torch.load                       # the reference calls torch.load(path, map_location='cpu') (utils/utils.py:577); torch >= 2.6
        torch.load = lambda *a, **k: _load(*a, **{**k, "weights_only": False})   # defaults to weights_only=True, which rejects ITS OWN files too
        args2 = types.SimpleNamespace(output_dir=tmp, auto_resume=True, resume="", model=args.model, start_epoch=0, model_ema=False)
        ref_utils.auto_load_model(args=args2, model=model2, model_without_ddp=model2, optimizer=opt2, loss_scaler=G.ScalerCPU(ref_utils))
        torch.load = _load
        assert args2.start_epoch == 2
        sd2 = model2.state_dict()
        for k in ref_sd:
            assert torch.equal(ref_sd[k], sd2[k]), k
        ro2 = opt2.state_dict()
        for i in ro["state"]:
            for f in ("exp_avg", "exp_avg_sq"):
                assert torch.equal(ro["state"][i][f], ro2["state"][i][f]), (i, f)
            assert int(ro["state"][i]["step"]) == int(ro2["state"][i]["step"])
        print("B OK: dig_amd checkpoint -> reference auto_load_model (model.load_state_dict strict, optimizer.load_state_dict)")


if __name__ == "__main__":
    main()
