"""Generate golden fixtures by running the UNMODIFIED reference (imported from /root/reference) on CPU.

Build-container only.  Writes small .npz files to tests/golden/ and asserts, while generating, that the
oracle restatement (oracle/dig_oracle.py) reproduces the reference to fp32 round-off.  What is stored is
data only: seeds, scalar metrics, per-tensor gradient norms, and a few sampled elements.

    python oracle/ref_harness/gen_golden.py            # all single-rank fixtures
    python oracle/ref_harness/gen_golden.py --world 2  # SyncBN / all-gather fixtures over gloo (spawns ranks)
"""
import argparse
import os
import sys
import types
from functools import partial

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)
import dig_oracle as O  # noqa: E402
import refenv  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def sample_index(numel: int, k: int = 8):
    """Deterministic element positions sampled from a flat tensor."""
    if numel <= k:
        return np.arange(numel)
    return (np.arange(k, dtype=np.int64) * 2654435761 + 12345) % numel


def build_ref_model(cfg: O.DiGConfig, drop_path: float = 0.0):
    import torch.nn as nn
    import modeling_pretrain_moco_mim_ori as M
    return M.MoCo_ViT(drop_path_rate=drop_path, img_size=(cfg.img_h, cfg.img_w), patch_size=cfg.patch, encoder_embed_dim=cfg.embed_dim,
                      encoder_depth=cfg.depth, encoder_num_heads=cfg.heads, encoder_num_classes=0,
                      decoder_num_classes=cfg.dec_classes, decoder_embed_dim=cfg.dec_dim, decoder_depth=4,
                      decoder_num_heads=3, mlp_ratio=cfg.mlp_ratio, qkv_bias=True,
                      norm_layer=partial(nn.LayerNorm, eps=1e-6), use_pixel_target=cfg.use_pixel, use_moco_target=cfg.use_moco,
                      mlp_dim=cfg.moco_mlp_dim, dim=cfg.moco_dim, T=cfg.T, num_windows=cfg.num_windows,
                      patchnet_name=cfg.patchnet)


def ref_args(hp: O.StepHyper, epochs=10):
    a = types.SimpleNamespace()
    a.num_view = 2
    a.moco_m = hp.moco_m
    a.use_moco_m_cos = 1
    a.epochs = epochs
    a.contrast_start_epoch = 0
    a.contrast_warmup_steps = 0
    a.loss_weight_contrast = hp.w_contrast
    a.loss_weight_pixel = hp.w_pixel
    a.only_mim_on_ori_img = bool(hp.only_mim_on_ori_img)
    a.eval_freq = 500
    a.opt = 'adamw'
    a.lr = hp.lr
    a.weight_decay = hp.weight_decay
    a.opt_eps = hp.eps
    a.opt_betas = None
    a.momentum = 0.9
    return a


class ScalerCPU:
    """NativeScalerWithGradNormCount on CPU: GradScaler is disabled without CUDA; add the `scale` key the
    engine reads (engine_for_pretraining_moco.py:157)."""

    def __init__(self, ref_utils):
        self.inner = ref_utils.NativeScalerWithGradNormCount()

    def __call__(self, *a, **k):
        return self.inner(*a, **k)

    def state_dict(self):
        d = dict(self.inner.state_dict())
        d.setdefault("scale", 1.0)
        return d


def patch_drop_paths(model, holder):
    """--drop_path > 0: every DropPath INSTANCE of the unmodified model (encoder.blocks.i.drop_path and momentum_encoder.blocks.i.drop_path,
    each called twice per forward: attention branch, then MLP branch -- modeling_finetune.py:156-158) gets its forward replaced by the keyed
    per-sample mask of its site, as the device and the oracle draw it (torch's own generator stream cannot be reproduced outside torch)."""
    import modeling_finetune as MF
    import finetune_oracle as FO
    counts, done = {}, []
    for name, m in model.named_modules():
        if not isinstance(m, MF.DropPath):
            continue
        parts = name.split(".")
        assert parts[0] in ("encoder", "momentum_encoder") and parts[1] == "blocks" and parts[3] == "drop_path", name
        i = int(parts[2])
        j = i + (O.MOMENTUM_SITE_OFFSET if parts[0] == "momentum_encoder" else 0)

        def f(x, name=name, i=i, j=j):
            n = counts.get(name, 0)
            counts[name] = n + 1
            dr = holder["dr"]
            return dr.path(FO.enc_site(j, 2 if n % 2 == 0 else 4), x, dr.dpr[i])
        m.forward = f
        done.append(name)
    return done


def run_reference_steps(cfg, seed, B, n_steps, hp, world=1, rank=0, sync_bn=False):
    """Drive the unmodified engine for n_steps one-batch 'epochs-worth' loaders; capture per-step values."""
    ref_utils = refenv.setup(rank=rank, world_size=world)
    import engine_for_pretraining_moco as E
    import optim_factory
    model = build_ref_model(cfg, hp.drop_path)
    drop_holder = {}
    if hp.drop_path:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        patched = patch_drop_paths(model, drop_holder)
        assert len(patched) == (2 if cfg.use_moco else 1) * (cfg.depth - 1), patched      # (block 0's rate is 0: nn.Identity, modeling_finetune.py:137)
    P, S = O.det_state(cfg, seed)
    sd = model.state_dict()
    for k, v in P.items():
        assert sd[k].shape == v.shape, k
        sd[k].copy_(v)
    assert list(P.keys()) == [n for n, _ in model.named_parameters()], "param order differs from reference"
    args = ref_args(hp)
    opt = optim_factory.create_optimizer(args, model)
    scaler = ScalerCPU(ref_utils)
    caps = {}
    model.encoder.register_forward_hook(lambda m, i, o: caps.__setitem__("enc", o.detach().clone()))
    if cfg.use_moco:
        model.predictor.register_forward_hook(lambda m, i, o: caps.__setitem__("qs", o.detach().clone()))
        model.momentum_projection_layer.register_forward_hook(lambda m, i, o: caps.__setitem__("ks", o.detach().clone()))
    fwd = model.forward

    def wrapped(*a, **k):
        out = fwd(*a, **k)
        assert ("vis_out" in out) == cfg.use_pixel and ("contra_loss" in out) == cfg.use_moco
        if "vis_out" in out:
            caps["vis_out"] = out["vis_out"][0].detach().clone()
            if len(out["vis_out"]) > 1:                 # only_mim_on_ori_img=False: the second view's masked predictions
                caps["vis_out1"] = out["vis_out"][1].detach().clone()
        return out
    model.forward = wrapped
    run_model = model
    if world > 1:
        import torch.distributed as dist
        import torch.nn.functional as F
        import torch.distributed.nn.functional as dnf
        # SyncBatchNorm refuses CPU modules: keep BatchNorm1d and replace F.batch_norm by its definition over
        # all ranks; wrap in gloo DDP for gradient averaging.
        def synced_bn(x, rm, rv, weight=None, bias=None, training=False, momentum=0.1, eps=1e-5):
            C = x.shape[1]
            st = torch.cat([x.sum(0), (x * x).sum(0), x.new_tensor([float(x.shape[0])])])
            st = dnf.all_reduce(st)
            n = st[-1]
            mean = st[:C] / n
            var = st[C:2 * C] / n - mean * mean
            with torch.no_grad():
                rm.mul_(1 - momentum).add_(mean, alpha=momentum)
                rv.mul_(1 - momentum).add_(var * (n / (n - 1)), alpha=momentum)
            y = (x - mean) * torch.rsqrt(var + eps)
            if weight is not None:
                y = y * weight + bias
            return y
        F.batch_norm = synced_bn
        run_model = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)
    steps = []
    iters_per_epoch = 1
    lr_sched = np.full(n_steps * 4, hp.lr)
    wd_sched = np.full(n_steps * 4, hp.weight_decay)
    for s in range(n_steps):
        im, au, mk = O.synthetic_batch(B, cfg, seed * 1000 + 17 * s + rank)
        loader = [([im, au, mk], torch.ones(1), torch.ones(1))]
        if hp.drop_path:
            import finetune_oracle as FO
            drop_holder["dr"] = FO.DropOracle(hp.drop_seed, s, drop_path=hp.drop_path, depth=cfg.depth)
        stats = E.train_one_epoch(run_model, None, None, loader, None, opt, torch.device('cpu'), s, scaler, hp.clip_grad,
                                  patch_size=cfg.patch, normlize_target=False, start_steps=s,
                                  lr_schedule_values=lr_sched, wd_schedule_values=wd_sched, args=args)
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        params = {n: p.detach().clone() for n, p in model.named_parameters()}
        bufs = {n: b.detach().clone() for n, b in model.named_buffers()}
        name_of = {id(p): n for n, p in model.named_parameters()}
        moments = {name_of[id(p)]: (st["exp_avg"].clone(), st["exp_avg_sq"].clone(), int(st["step"]))
                   for p, st in opt.state.items()}
        steps.append(dict(stats=dict(stats), grads=grads, params=params, bufs=bufs, caps=dict(caps), moments=moments))
    return steps


def run_oracle_steps(cfg, seed, B, n_steps, hp, comm=None, rank=0, teacher=None):
    """teacher: the reference's steps; when given, step s>0 starts from the reference's post-step-(s-1) state
    (params, BN buffers, Adam moments) so each step is compared from an identical starting point -- Adam's
    sign-like update otherwise amplifies 1e-9 gradient round-off into 1e-3 parameter differences."""
    P, S = O.det_state(cfg, seed)
    tr = O.OracleTrainer(cfg, P, S, comm=comm)
    steps = []
    for s in range(n_steps):
        if teacher is not None and s > 0:
            prev = teacher[s - 1]
            for k in tr.P:
                tr.P[k] = prev["params"][k].clone()
            for k in tr.S:
                tr.S[k] = prev["bufs"][k].clone()
            for k, (m1, m2, st) in prev["moments"].items():
                tr.exp_avg[k], tr.exp_avg_sq[k] = m1.clone(), m2.clone()
            tr.step_count = s
        im, au, mk = O.synthetic_batch(B, cfg, seed * 1000 + 17 * s + rank)
        import dataclasses
        hps = dataclasses.replace(hp, moco_m=O.adjust_moco_momentum(float(s), 10, hp.moco_m))
        taps = {}
        metrics, grads, out, labels = tr.step(im, au, mk, hps, taps)
        caps = dict(enc=taps["enc"].detach())
        if cfg.use_moco:
            caps.update(qs=torch.cat([taps["q1"], taps["q2"]]).detach(), ks=torch.cat([taps["k1"], taps["k2"]]).detach())
        if cfg.use_pixel:
            caps["vis_out"] = out["vis_out"][0].detach()
            if len(out["vis_out"]) > 1:
                caps["vis_out1"] = out["vis_out"][1].detach()
        # (the reference leaves `.grad = None` on a parameter its forward never reads -- Dis-only's mask_token: not in its gradient list)
        steps.append(dict(stats=metrics, grads={k: v.clone() for k, v in grads.items() if k not in tr.never_grad},
                          params={k: v.detach().clone() for k, v in tr.P.items()},
                          bufs={k: v.clone() for k, v in tr.S.items()}, caps=caps))
    return steps


def check_adamw_formula(ref_step0, cfg, seed, hp, m0):
    """Tight check of the optimizer restatement: feed the REFERENCE's step-0 gradients through the oracle's
    AdamW/EMA and compare with the reference's post-step parameters (Adam's sign-like first step makes a
    params-vs-params comparison ill-conditioned wherever |g| ~ eps, so gradients are taken as given)."""
    P, _ = O.det_state(cfg, seed)
    decay, no_decay = O.param_groups(P, 1.0)
    O.ema_update(P, m0)
    worst = 0.0
    for names, wd in ((decay, hp.weight_decay), (no_decay, 0.0)):
        for n in names:
            g = ref_step0["grads"].get(n)
            if g is None:
                continue
            O.adamw_update(P[n], g, torch.zeros_like(g), torch.zeros_like(g), 1, hp.lr, wd, hp.beta1, hp.beta2, hp.eps)
    for n, p in P.items():
        err = (p - ref_step0["params"][n]).abs().max().item()
        worst = max(worst, err)
        assert err <= 1e-7 + 1e-6 * p.abs().max().item(), (n, err)
    print(f"AdamW + EMA restatement == reference given reference grads; worst abs err {worst:.2e}")


def compare(ref_steps, ora_steps, rtol=2e-4, atol=2e-5, tag="", lr=1e-3, nonsmooth=None):
    """nonsmooth (ConvPatchNet: cfg): the loss is piecewise smooth in the weights with ~10^6 kinks per step (2x2 max-pool arg-maxima and ReLU
    signs over [2 B, 8 x 32 .. 1 x 4] maps) -- a 1e-7 round-off difference upstream (this file's BatchNorm sums vs torch's native kernel) flips a
    few of them, and each flip moves single gradient elements by their full size: the reference in fp32 differs from ITSELF in fp64 by 5e-2 of
    the maximum there (check_conv_module prints it).  Gradients are then compared in the L2 norm per tensor (<= 1e-2), everything else as
    before; the module itself is pinned tightly on identical inputs by check_conv_module."""
    worst = 0.0
    for s, (r, o) in enumerate(zip(ref_steps, ora_steps)):
        logged = ("loss", "loss_pixel", "loss_contrast", "q1_acc1", "q1_acc5", "q2_acc1", "q2_acc5")
        assert {k for k in logged if k in r["stats"]} == {k for k in logged if k in o["stats"]}, (r["stats"].keys(), o["stats"].keys())
        assert set(r["grads"]) == set(o["grads"]), set(r["grads"]) ^ set(o["grads"])
        for k in ("loss", "loss_pixel", "loss_contrast", "q1_acc1", "q1_acc5", "q2_acc1", "q2_acc5", "grad_norm"):
            if k not in r["stats"]:
                continue
            a, b = float(r["stats"][k]), float(o["stats"][k])
            assert abs(a - b) <= 1e-4 * max(1.0, abs(a)), (tag, s, k, a, b)
        for grp in ("caps", "grads", "params", "bufs"):
            for k, a in r[grp].items():
                b = o[grp][k]
                if a.dtype == torch.int64:
                    assert int(a) == int(b), (tag, s, grp, k)
                    continue
                err = (a - b).abs().max().item()
                ref = a.abs().max().item()
                if err / (ref + 1e-12) > worst:
                    worst, worst_at = err / (ref + 1e-12), (s, grp, k, err, ref)
                slack = 2.5 * lr if grp == "params" else 0.0   # Adam step is sign-like where |g|~eps
                if nonsmooth is not None and grp == "grads":
                    if O.bn_cancelled_bias(k, nonsmooth):         # a bias in front of a BatchNorm: its true gradient is zero, both sides hold round-off
                        assert a.norm().item() <= 1e-6 and b.norm().item() <= 1e-6, (tag, s, k, a.norm().item(), b.norm().item())
                        continue
                    l2 = (a - b).norm().item() / (a.norm().item() + 1e-30)
                    assert l2 <= 1e-2, (tag, s, grp, k, l2)
                    continue
                assert err <= atol + rtol * ref + slack, (tag, s, grp, k, err, ref)
    print(f"[{tag}] oracle == reference over {len(ref_steps)} steps; worst rel-to-max err {worst:.2e} at {worst_at}")


def pack(ref_steps, cfg, seed, B, hp, extra=None):
    d = {"seed": np.int64(seed), "B": np.int64(B), "n_steps": np.int64(len(ref_steps)),
         "cfg_keys": np.array([k for k, v in vars(cfg).items() if not isinstance(v, str)]),
         "cfg_vals": np.array([float(v) for v in vars(cfg).values() if not isinstance(v, str)]), "cfg_kind": np.array(cfg.kind),
         "cfg_patchnet": np.array(cfg.patchnet),
         "hp_keys": np.array([k for k, v in vars(hp).items() if isinstance(v, (int, float)) and v is not None]),
         "hp_vals": np.array([float(v) for k, v in vars(hp).items() if isinstance(v, (int, float)) and v is not None])}
    for s, r in enumerate(ref_steps):
        for k in ("loss", "loss_pixel", "loss_contrast", "q1_acc1", "q1_acc5", "q2_acc1", "q2_acc5", "grad_norm", "moco_m"):
            if k in r["stats"]:                     # (a single-objective model logs its own loss only)
                d[f"s{s}/stat/{k}"] = np.float64(r["stats"][k])
        names = sorted(r["grads"].keys())
        d[f"s{s}/grad_names"] = np.array(names)
        d[f"s{s}/grad_norms"] = np.array([r["grads"][n].double().norm().item() for n in names])
        d[f"s{s}/grad_samples"] = np.stack([np.resize(r["grads"][n].reshape(-1)[sample_index(r["grads"][n].numel())].numpy(), 8) for n in names])
        pn = sorted(r["params"].keys())
        d[f"s{s}/param_names"] = np.array(pn)
        d[f"s{s}/param_norms"] = np.array([r["params"][n].double().norm().item() for n in pn])
        d[f"s{s}/param_samples"] = np.stack([np.resize(r["params"][n].reshape(-1)[sample_index(r["params"][n].numel())].numpy(), 8) for n in pn])
        bn = sorted(k for k in r["bufs"] if not k.endswith("num_batches_tracked"))
        d[f"s{s}/buf_names"] = np.array(bn)
        d[f"s{s}/buf_norms"] = np.array([r["bufs"][n].double().norm().item() for n in bn])
        for k, v in r["caps"].items():
            d[f"s{s}/cap/{k}/norm"] = np.float64(v.double().norm().item())
            d[f"s{s}/cap/{k}/samples"] = v.reshape(-1)[sample_index(v.numel(), 64)].numpy()
        if "vis_out" in r["caps"]:
            d[f"s{s}/cap/vis_out/full"] = r["caps"]["vis_out"].numpy().astype(np.float32)
        if "vis_out1" in r["caps"]:
            d[f"s{s}/cap/vis_out1/full"] = r["caps"]["vis_out1"].numpy().astype(np.float32)
        if hp.clip_grad is not None:                # --clip_grad (utils/utils.py:487-493): the clip coefficient lives in the Adam moments
            mn = sorted(r["moments"].keys())
            d[f"s{s}/moment_names"] = np.array(mn)
            d[f"s{s}/exp_avg_norms"] = np.array([r["moments"][n][0].double().norm().item() for n in mn])
            d[f"s{s}/exp_avg_sq_norms"] = np.array([r["moments"][n][1].double().norm().item() for n in mn])
    if extra:
        d.update(extra)
    return d


def check_conv_module(cfg, seed, x0, go):
    """ConvPatchNet alone on the input / output gradient captured in the reference's step (identical bits on both sides): the oracle's
    restatement against the reference module, both fp32 -- and, for the record, the reference module against itself in fp64."""
    import modeling_pretrain_moco_mim_ori as M
    P, S = O.det_state(cfg, seed)
    res = {}
    with torch.enable_grad():
        for tg, dt in (("ref32", torch.float32), ("ref64", torch.float64)):
            net = M.ConvPatchNet(embed_dim=cfg.embed_dim, num_windows=cfg.num_windows, patch_shape=cfg.grid).to(dt)
            missing = net.load_state_dict({k[len("patch_extractor."):]: v.to(dt) for k, v in P.items() if k.startswith("patch_extractor.")}, strict=False)
            assert not missing.unexpected_keys and all(k.rsplit(".", 1)[-1].startswith(("running_", "num_batches")) for k in missing.missing_keys), missing
            net.train()
            x = x0.detach().clone().to(dt).requires_grad_()
            y = net(x)
            (y * go.to(dt)).sum().backward()
            res[tg] = [y.detach().double(), x.grad.double()] + [p.grad.double() for n, p in net.named_parameters() if n.endswith("weight")]
        Pd = {k: v.clone().requires_grad_() for k, v in P.items() if k.startswith("patch_extractor.")}
        x = x0.detach().clone().requires_grad_()
        y = O.conv_patch_extractor(x, Pd, {k: v.clone() for k, v in S.items()}, "patch_extractor.", cfg, O.LocalComm())
        (y * go).sum().backward()
        res["ora32"] = [y.detach().double(), x.grad.double()] + [g.grad.double() for n, g in Pd.items() if n.endswith("weight")]
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()          # noqa: E731
    tight = max(rel(a, b) for a, b in zip(res["ora32"], res["ref32"]))
    loose = max(rel(a, b) for a, b in zip(res["ref32"], res["ref64"]))
    assert tight <= 2e-5, tight
    print(f"ConvPatchNet: oracle == reference module on the step's own input, worst rel-to-max err {tight:.2e} over output, d input and "
          f"{len(res['ref32']) - 2} weight gradients (the reference in fp32 vs itself in fp64: {loose:.2e})")


def gen_single(tag, cfg, seed, B, n_steps, hp):
    global build_ref_model
    cap, orig = {}, build_ref_model
    if cfg.patchnet == "conv":
        def build_ref_model(c, dp=0.0):
            m = orig(c, dp)

            def grab_in(mod, inp):
                cap.setdefault("x", inp[0].detach().clone())

            def grab_go(mod, gi, go):
                cap.setdefault("go", go[0].detach().clone())
            m.patch_extractor.register_forward_pre_hook(grab_in)
            m.patch_extractor.register_full_backward_hook(grab_go)
            return m
    try:
        ref_steps = run_reference_steps(cfg, seed, B, n_steps, hp)
    finally:
        build_ref_model = orig
    ora_steps = run_oracle_steps(cfg, seed, B, n_steps, hp, teacher=ref_steps)
    if cfg.patchnet == "conv":
        check_conv_module(cfg, seed, cap["x"], cap["go"])
    compare(ref_steps, ora_steps, tag=tag, lr=hp.lr, nonsmooth=cfg if cfg.patchnet == "conv" else None)
    check_adamw_formula(ref_steps[0], cfg, seed, hp, O.adjust_moco_momentum(0.0, 10, hp.moco_m))
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, f"{tag}.npz"), **pack(ref_steps, cfg, seed, B, hp))
    print("wrote", tag)


def gen_masks_and_schedules():
    """Mask generator and schedule goldens straight from the reference functions."""
    ref_utils = refenv.setup()
    from masking_generator import RandomMaskingGenerator
    d = {}
    for seed in (0, 1234, 99):
        np.random.seed(seed)
        gen = RandomMaskingGenerator((8, 32), 0.7, num_view=2)
        ref = np.stack([gen() for _ in range(4)])
        mine = O.random_masks(4, O.DiGConfig(), 0.7, np.random.RandomState(seed)).numpy()
        assert np.array_equal(ref, mine), "mask stream differs"
        d[f"mask/{seed}"] = ref.astype(np.uint8)
    d["sched/lr"] = ref_utils.cosine_scheduler(1.5e-4 * 4, 1e-5, 10, 50, warmup_epochs=1, warmup_steps=-1)
    d["sched/lr_ws"] = ref_utils.cosine_scheduler(6e-4, 1e-5, 10, 50, warmup_epochs=1, warmup_steps=20)
    d["sched/wd"] = ref_utils.cosine_scheduler(0.1, 0.1, 10, 50)
    assert np.array_equal(d["sched/lr"], O.cosine_scheduler(1.5e-4 * 4, 1e-5, 10, 50, warmup_epochs=1))
    assert np.array_equal(d["sched/lr_ws"], O.cosine_scheduler(6e-4, 1e-5, 10, 50, warmup_epochs=1, warmup_steps=20))
    a = types.SimpleNamespace(epochs=10, moco_m=0.99)
    d["sched/moco_m"] = np.array([ref_utils.adjust_moco_momentum(e / 7.0, a) for e in range(70)])
    assert np.array_equal(d["sched/moco_m"], np.array([O.adjust_moco_momentum(e / 7.0, 10, 0.99) for e in range(70)]))
    np.savez_compressed(os.path.join(GOLD, "masks_schedules.npz"), **d)
    print("wrote masks_schedules")


def worker(rank, world, cfg, seed, B, n_steps, hp, q):
    ref_steps = run_reference_steps(cfg, seed, B, n_steps, hp, world=world, rank=rank)
    comm = O.DistComm()
    ora_steps = run_oracle_steps(cfg, seed, B, n_steps, hp, comm=comm, rank=rank, teacher=ref_steps)
    compare(ref_steps, ora_steps, tag=f"w{world}r{rank}", lr=hp.lr)
    q.put((rank, pack(ref_steps, cfg, seed, B, hp)))


def gen_multi(world, cfg, seed, B, n_steps, hp):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, cfg, seed, B, n_steps, hp, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get() for _ in range(world))
    for p in procs:
        p.join()
        assert p.exitcode == 0
    for r, d in res.items():
        np.savez_compressed(os.path.join(GOLD, f"tiny_w{world}_rank{r}.npz"), **d)
    print("wrote multi-rank fixtures, world", world)


def gen_checkpoint_structure():
    """Structure (not values) of the reference optimizer's state_dict after one step of the tiny model: per-index tensor
    shapes and the parameter index lists of each group -- what a checkpoint interchange has to reproduce."""
    cfg = O.DiGConfig(**O.TINY)
    hp = O.StepHyper(lr=1e-3)
    ref_utils = refenv.setup()
    import engine_for_pretraining_moco as E
    import optim_factory
    model = build_ref_model(cfg)
    P, S = O.det_state(cfg, 3)
    sd = model.state_dict()
    for k, v in P.items():
        sd[k].copy_(v)
    args = ref_args(hp)
    opt = optim_factory.create_optimizer(args, model)
    im, au, mk = O.synthetic_batch(2, cfg, 5)
    E.train_one_epoch(model, None, None, [([im, au, mk], torch.ones(1), torch.ones(1))], None, opt, torch.device('cpu'), 0,
                      ScalerCPU(ref_utils), None, patch_size=cfg.patch, normlize_target=False, start_steps=0,
                      lr_schedule_values=np.full(2, hp.lr), wd_schedule_values=np.full(2, hp.weight_decay), args=args)
    osd = opt.state_dict()
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    name_of = {id(p): n for n, p in model.named_parameters()}
    idx_names = [name_of[id(p)] for g in opt.param_groups for p in g["params"]]
    d = {"n_state": np.int64(len(osd["state"])), "group_sizes": np.array([len(g["params"]) for g in osd["param_groups"]]),
         "group_wd": np.array([g["weight_decay"] for g in osd["param_groups"]]),
         "group_keys": np.array(sorted(osd["param_groups"][0].keys())),
         "idx_names": np.array(idx_names), "steps": np.array([osd["state"][i]["step"] for i in range(len(idx_names))]),
         "exp_avg_norms": np.array([osd["state"][i]["exp_avg"].double().norm().item() for i in range(len(idx_names))])}
    assert all(tuple(osd["state"][i]["exp_avg"].shape) == tuple(P[n].shape) for i, n in enumerate(idx_names))
    np.savez_compressed(os.path.join(GOLD, "optimizer_state_structure.npz"), **d)
    print("wrote optimizer_state_structure")



if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--only", default="")
    ap.add_argument("--ckpt-structure", action="store_true")
    a = ap.parse_args()
    if a.ckpt_structure:
        torch.manual_seed(0)
        gen_checkpoint_structure()
        sys.exit(0)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tiny = O.DiGConfig(**O.TINY)
    hp = O.StepHyper(lr=1e-3)
    if a.world > 1:
        gen_multi(a.world, tiny, 3, 4, 2, hp)
    else:
        if a.only in ("", "masks"):
            gen_masks_and_schedules()
        if a.only in ("", "tiny"):
            gen_single("tiny_w1", tiny, 3, 4, 2, hp)
        if a.only in ("", "small"):
            small = O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128")
            gen_single("vit_small_b4_w1", small, 7, 4, 1, O.StepHyper(lr=1.5e-4 * 4 / 256))
        if a.only in ("", "base"):                  # BASELINE configs[3]'s model (D=512, 8 heads), smallest batch BatchNorm accepts
            base = O.make_config("pretrain_simmim_moco_ori_vit_base_patch4_32x128")
            gen_single("vit_base_b2_w1", base, 11, 2, 1, O.StepHyper(lr=1.5e-4 * 2 / 256))
        if a.only in ("", "c0"):                    # BASELINE configs[1]'s loss: loss_weight_contrast = 0 (MIM-only gradients)
            gen_single("tiny_w1_c0", tiny, 5, 4, 1, O.StepHyper(lr=1e-3, w_contrast=0.0))
        if a.only in ("", "clip"):                  # --clip_grad 1.0 (grad norm ~5: the clip is active), two steps: the second step's moments mix
            gen_single("tiny_w1_clip", tiny, 13, 4, 2, O.StepHyper(lr=1e-3, clip_grad=1.0))      # both steps' clip coefficients
        if a.only in ("", "dis"):                   # Dis-only (pretrain_moco_ori_*: use_pixel_target=False): no mask, no pix_projector, no decoder
            import dataclasses
            gen_single("tiny_dis_w1", dataclasses.replace(tiny, kind="moco"), 21, 4, 2, hp)
        if a.only in ("", "gen"):                   # Gen-only (pretrain_simmim_ori_*: use_moco_target=False): encoder + final norm + decoder
            import dataclasses
            gen_single("tiny_gen_w1", dataclasses.replace(tiny, kind="simmim"), 23, 4, 2, hp)
        if a.only in ("", "gen2"):                  # ... with the MIM loss on both views
            import dataclasses
            gen_single("tiny_gen_w1_mim2", dataclasses.replace(tiny, kind="simmim"), 25, 4, 1, O.StepHyper(lr=1e-3, only_mim_on_ori_img=False))
        if a.only in ("", "nw5"):                   # the argparse default --num_windows 5: uneven adaptive_avg_pool2d windows on 32 columns
            import dataclasses
            gen_single("tiny_w1_nw5", dataclasses.replace(tiny, num_windows=5), 27, 4, 1, hp)
        if a.only in ("", "regular"):               # the reference CLI's defaults: --patchnet_name regular with --num_windows 5 (run_mae_pretraining_moco.py:143-145)
            import dataclasses
            gen_single("tiny_w1_regular", dataclasses.replace(tiny, patchnet="regular", num_windows=5), 35, 4, 2, hp)
        if a.only in ("", "conv"):                  # --patchnet_name conv (ConvPatchNet, :207-260): conv3x3 / BatchNorm2d / max-pool stack, one patch per image
            import dataclasses
            gen_single("tiny_w1_conv", dataclasses.replace(tiny, patchnet="conv", num_windows=5), 41, 8, 2, hp)
        if a.only in ("", "dp"):                    # --drop_path 0.3 (run_mae_pretraining_moco.py:87): stochastic depth in both encoders under keyed masks
            gen_single("tiny_w1_dp", O.DiGConfig(**dict(O.TINY, depth=3)), 29, 8, 2, O.StepHyper(lr=1e-3, drop_path=0.3, drop_seed=1234))
        if a.only in ("", "mim2"):                  # only_mim_on_ori_img=False: both views masked, MIM loss on both (engine :100-111,138-141)
            gen_single("tiny_w1_mim2", tiny, 9, 4, 1, O.StepHyper(lr=1e-3, only_mim_on_ori_img=False))
