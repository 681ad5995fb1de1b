"""The oracle (oracle/dig_oracle.py) replayed against fixtures produced by the unmodified reference
(oracle/ref_harness/gen_golden.py).  CPU only; this is what pins the oracle."""
import dataclasses
import os

import numpy as np
import pytest
import torch

import dig_oracle as O


def sample_index(numel, k=8):
    if numel <= k:
        return np.arange(numel)
    return (np.arange(k, dtype=np.int64) * 2654435761 + 12345) % numel


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def cfg_from(g):
    kw = {k: v for k, v in zip(g["cfg_keys"].tolist(), g["cfg_vals"].tolist())}
    ints = {"img_h", "img_w", "patch", "in_chans", "embed_dim", "depth", "heads", "dec_dim", "dec_classes",
            "moco_dim", "moco_mlp_dim", "pix_mlp_dim", "num_windows", "patchnet_depth"}
    if "cfg_kind" in g:                                                   # fixtures of the single-objective models
        kw["kind"] = str(g["cfg_kind"])
    if "cfg_patchnet" in g:
        kw["patchnet"] = str(g["cfg_patchnet"])
    return O.DiGConfig(**{k: (int(v) if k in ints else v) for k, v in kw.items()})


def hp_from(g):
    kw = {k: v for k, v in zip(g["hp_keys"].tolist(), g["hp_vals"].tolist())}
    kw["only_mim_on_ori_img"] = bool(kw.get("only_mim_on_ori_img", 1.0))
    if "drop_seed" in kw:
        kw["drop_seed"] = int(kw["drop_seed"])
    return O.StepHyper(**kw)


def check_step0(g, rtol=3e-4, atol_scale=1.0, samples=True):
    cfg, hp = cfg_from(g), hp_from(g)
    seed, B = int(g["seed"]), int(g["B"])
    P, S = O.det_state(cfg, seed)
    tr = O.OracleTrainer(cfg, P, S)
    im, au, mk = O.synthetic_batch(B, cfg, seed * 1000)
    hp0 = dataclasses.replace(hp, moco_m=float(g["s0/stat/moco_m"]))
    taps = {}
    metrics, grads, out, labels = tr.step(im, au, mk, hp0, taps)
    logged = ("loss", "loss_pixel", "loss_contrast", "q1_acc1", "q1_acc5", "q2_acc1", "q2_acc5", "grad_norm")
    assert {k for k in logged if k in metrics} == {k for k in logged if f"s0/stat/{k}" in g}       # a single-objective model logs its own loss only
    for k in logged:
        if k in metrics:
            assert metrics[k] == pytest.approx(float(g[f"s0/stat/{k}"]), rel=1e-4, abs=1e-5), k
    names = g["s0/grad_names"].tolist()
    assert set(names) == {n for n in grads if n not in tr.never_grad}      # `.grad is None` in the reference <=> never read by its forward
    norms = g["s0/grad_norms"]
    gsamples = g["s0/grad_samples"]
    gmax = max(norms)
    for i, n in enumerate(names):
        gi = grads[n]
        assert gi.double().norm().item() == pytest.approx(norms[i], rel=rtol, abs=1e-6 * gmax), n
        got = np.resize(gi.reshape(-1)[sample_index(gi.numel())].numpy(), 8)
        if samples:
            # (floor of 1e-7 of the largest tensor's gradient norm: a gradient that is zero in exact arithmetic -- a per-channel constant in
            #  front of a BatchNorm, e.g. the last block's fc2 bias of the Dis-only model -- is round-off noise of that size on both sides)
            np.testing.assert_allclose(got, gsamples[i], rtol=rtol, atol=max(1e-7 * gmax, atol_scale * 1e-5 * max(1e-3, float(np.abs(gsamples[i]).max()) + norms[i] / np.sqrt(gi.numel()))))
    if cfg.use_pixel:
        np.testing.assert_allclose(out["vis_out"][0].detach().numpy(), g["s0/cap/vis_out/full"], rtol=1e-4, atol=2e-5)
    if "s0/cap/vis_out1/full" in g:
        np.testing.assert_allclose(out["vis_out"][1].detach().numpy(), g["s0/cap/vis_out1/full"], rtol=1e-4, atol=2e-5)
    enc = taps["enc"].detach()
    np.testing.assert_allclose(enc.reshape(-1)[sample_index(enc.numel(), 64)].numpy(), g["s0/cap/enc/samples"], rtol=1e-4, atol=1e-4)
    if cfg.use_moco:
        ks = torch.cat([taps["k1"], taps["k2"]]).detach()
        np.testing.assert_allclose(ks.reshape(-1)[sample_index(ks.numel(), 64)].numpy(), g["s0/cap/ks/samples"], rtol=1e-3, atol=1e-3)
    # post-step state: EMA'd momentum params are well-conditioned; online params only loosely (Adam sign step)
    pn = g["s0/param_names"].tolist()
    pnorm = g["s0/param_norms"]
    for i, n in enumerate(pn):
        tol = 1e-5 if not O.is_trainable(n) else 5e-3
        assert tr.P[n].double().norm().item() == pytest.approx(pnorm[i], rel=tol, abs=1e-6), n
    bn = g["s0/buf_names"].tolist()
    for i, n in enumerate(bn):
        assert tr.S[n].double().norm().item() == pytest.approx(g["s0/buf_norms"][i], rel=1e-4, abs=1e-5), n


def test_tiny_step_matches_reference(golden_dir):
    check_step0(load(golden_dir, "tiny_w1"))


def test_vit_small_b4_step_matches_reference(golden_dir):
    """BASELINE.json configs[0]: pretrain_simmim_moco_ori_vit_small_patch4_32x128, bs=4, CPU, world 1."""
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    check_step0(load(golden_dir, "vit_small_b4_w1"))


def test_vit_base_b2_step_matches_reference(golden_dir):
    """BASELINE.json configs[3]'s model (pretrain_simmim_moco_ori_vit_base_patch4_32x128: D=512, 8 heads), bs=2, CPU."""
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    check_step0(load(golden_dir, "vit_base_b2_w1"))


def test_zero_contrast_weight_step_matches_reference(golden_dir):
    """BASELINE.json configs[1]'s loss (loss_weight_contrast = 0): contrastive-only parameters get exactly-zero gradients
    in the reference, the logged loss_contrast / accuracies are still computed."""
    g = load(golden_dir, "tiny_w1_c0")
    check_step0(g)
    names, norms = g["s0/grad_names"].tolist(), g["s0/grad_norms"]
    zero = [n for n, v in zip(names, norms) if v == 0.0]
    assert any(n.startswith("predictor.") for n in zero) and any(n.startswith("pix_projector.") for n in zero)
    assert float(g["s0/stat/loss"]) == pytest.approx(float(g["s0/stat/loss_pixel"]), rel=1e-6) and float(g["s0/stat/loss_contrast"]) > 0


def test_both_views_mim_step_matches_reference(golden_dir):
    """only_mim_on_ori_img=False: both views masked in both encoders, one masked-pixel loss per view (the second against patches of the
    ORIGINAL crops, engine_for_pretraining_moco.py:106-108), loss_pixel their mean."""
    g = load(golden_dir, "tiny_w1_mim2")
    assert not hp_from(g).only_mim_on_ori_img and "s0/cap/vis_out1/full" in g
    check_step0(g)


def test_dis_only_step_matches_reference(golden_dir):
    """pretrain_moco_ori_* (use_pixel_target=False, modeling_pretrain_moco_mim_ori.py:627-653): no mask, no pix_projector, no decoder, only
    loss_contrast is logged; encoder.mask_token never gets a gradient and the reference's AdamW leaves it alone."""
    g = load(golden_dir, "tiny_dis_w1")
    cfg = cfg_from(g)
    assert cfg.kind == "moco" and "s0/stat/loss_pixel" not in g and "encoder.mask_token" not in g["s0/grad_names"].tolist()
    assert not any(n.startswith(("pix_", "encoder.norm")) for n in g["s0/param_names"].tolist())
    check_step0(g)


@pytest.mark.parametrize("name", ["tiny_gen_w1", "tiny_gen_w1_mim2"])
def test_gen_only_step_matches_reference(golden_dir, name):
    """pretrain_simmim_ori_* (use_moco_target=False, :655-681): encoder with its final LayerNorm + pix_decoder, nothing else; only loss_pixel
    is logged."""
    g = load(golden_dir, name)
    cfg = cfg_from(g)
    pn = g["s0/param_names"].tolist()
    assert cfg.kind == "simmim" and "s0/stat/loss_contrast" not in g and "encoder.norm.weight" in pn
    assert not any(n.startswith(("momentum_", "predictor", "encoder_projection", "pix_projector")) for n in pn)
    check_step0(g)


def test_uneven_windows_step_matches_reference(golden_dir):
    """--num_windows 5, the reference CLI's default (run_mae_pretraining_moco.py:143): adaptive_avg_pool2d bins of 7 / 7 / 8 / 7 / 7 columns."""
    g = load(golden_dir, "tiny_w1_nw5")
    assert cfg_from(g).num_windows == 5
    check_step0(g)


def test_regular_patchnet_step_matches_reference(golden_dir):
    """The reference CLI's defaults --patchnet_name regular --num_windows 5 (run_mae_pretraining_moco.py:143-145): PatchNet with its two
    cross-attention blocks (modeling_pretrain_moco_mim_ori.py:137-205, :21-135) in both the online and the EMA'd momentum branch."""
    g = load(golden_dir, "tiny_w1_regular")
    cfg = cfg_from(g)
    pn = g["s0/param_names"].tolist()
    assert cfg.patchnet == "regular" and cfg.num_windows == 5
    assert "patch_extractor.blocks.1.attn.linear_k.weight" in pn and "momentum_patch_extractor.norm.bias" in pn
    check_step0(g)


def test_conv_patchnet_step_matches_reference(golden_dir):
    """--patchnet_name conv (ConvPatchNet, modeling_pretrain_moco_mim_ori.py:207-260): conv3x3 / BatchNorm2d / ReLU x 4 with three max-pools, one
    patch per image.  Losses, activations, buffers as everywhere; gradients in the per-tensor norm at 1e-2 without element samples -- arg-max
    and ReLU kinks make single elements jump under 1e-7 round-off differences (the reference in fp32 vs itself in fp64: 5e-2 of the maximum;
    oracle/ref_harness/gen_golden.py::compare), and the module alone is pinned at 5e-6 on identical inputs while the fixture is generated
    (check_conv_module)."""
    g = load(golden_dir, "tiny_w1_conv")
    cfg = cfg_from(g)
    pn = g["s0/param_names"].tolist()
    assert cfg.patchnet == "conv" and cfg.n_patch == 1 and cfg.conv_channels == (128, 128, 192, 256, 256)
    assert "patch_extractor.conv_layers.4.0.weight" in pn and "momentum_patch_extractor.patches2global.3.bias" in pn
    assert "patch_extractor.conv_layers.6.1.running_var" in g["s0/buf_names"].tolist()
    check_step0(g, rtol=1e-2, samples=False)
    names, norms = g["s0/grad_names"].tolist(), g["s0/grad_norms"]
    for n, v in zip(names, norms):
        if O.bn_cancelled_bias(n, cfg):
            assert v <= 1e-6, (n, v)                                      # a bias in front of a BatchNorm: zero gradient, round-off in the reference too


def test_drop_path_step_matches_reference(golden_dir):
    """--drop_path 0.3 (run_mae_pretraining_moco.py:87): stochastic depth on both branches of blocks 1.. of BOTH encoders, the unmodified
    reference run with every DropPath instance drawing the keyed per-sample masks the device and the oracle draw (gen_golden.patch_drop_paths)."""
    g = load(golden_dir, "tiny_w1_dp")
    hp = hp_from(g)
    assert hp.drop_path == pytest.approx(0.3) and cfg_from(g).depth == 3
    check_step0(g, rtol=2e-3)                       # (fp32 summation order with a third of the batch's branches zeroed: one of 8 x 356 sampled elements sits at 1e-3 relative)
    # the masks bite: the same step without them is a different step
    cfg, seed, B = cfg_from(g), int(g["seed"]), int(g["B"])
    im, au, mk = O.synthetic_batch(B, cfg, seed * 1000)
    hp0 = dataclasses.replace(hp, moco_m=float(g["s0/stat/moco_m"]), drop_path=0.0)
    m0, _, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
    assert abs(m0["loss"] - float(g["s0/stat/loss"])) > 1e-3 * abs(m0["loss"])


def test_clip_grad_two_steps_match_reference(golden_dir):
    """--clip_grad 1.0 on the pre-training path (utils/utils.py:487-493: clip_grad_norm_ between backward and optimizer.step): the
    fixture comes from the unmodified reference engine called with max_norm = 1.0 (gradient norm ~4-5: the clip is active).  Adam's
    update is invariant to the scale of a single gradient, so the clip coefficient is pinned where it lives: in the two moments."""
    g = load(golden_dir, "tiny_w1_clip")
    hp = hp_from(g)
    assert hp.clip_grad == 1.0 and float(g["s0/stat/grad_norm"]) > 2.0
    # step 0: metrics, norms of the CLIPPED gradients, post-step state (single elements of this seed's BatchNorm-head gradients carry
    # ~5e-7 of fp32 summation-order noise between an 8-thread and a default-thread run: per-tensor norms instead of element samples)
    check_step0(g, rtol=1e-3, samples=False)
    cfg, seed, B = cfg_from(g), int(g["seed"]), int(g["B"])
    tr = O.OracleTrainer(cfg, *O.det_state(cfg, seed))
    for s in range(2):
        im, au, mk = O.synthetic_batch(B, cfg, seed * 1000 + 17 * s)
        m, _, _, _ = tr.step(im, au, mk, dataclasses.replace(hp, moco_m=float(g[f"s{s}/stat/moco_m"])))
        for k in ("loss", "loss_pixel", "loss_contrast", "grad_norm"):
            assert m[k] == pytest.approx(float(g[f"s{s}/stat/{k}"]), rel=2e-3 if s else 1e-4, abs=1e-5), (s, k)
        names = g[f"s{s}/moment_names"].tolist()
        a = np.array([tr.exp_avg[n].double().norm().item() for n in names])
        b = np.array([tr.exp_avg_sq[n].double().norm().item() for n in names])
        np.testing.assert_allclose(a, g[f"s{s}/exp_avg_norms"], rtol=3e-3 if s else 1e-4, atol=1e-9)
        np.testing.assert_allclose(b, g[f"s{s}/exp_avg_sq_norms"], rtol=6e-3 if s else 2e-4, atol=1e-12)
    # the clip coefficient is really in there: unclipped first moments would be grad_norm / max_norm times larger
    coef = 1.0 / (float(g["s0/stat/grad_norm"]) + 1e-6)
    gn = np.sqrt((g["s0/grad_norms"] ** 2).sum())                        # stored gradients are the clipped ones
    assert gn == pytest.approx(coef * float(g["s0/stat/grad_norm"]), rel=1e-4)


def test_param_inventory_matches_reference_counts():
    cfg = O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128")
    shapes = O.param_shapes(cfg)
    numel = {k: int(np.prod(v)) for k, v in shapes.items()}
    assert len(shapes) == 356                                            # 398 keys incl. 42 BN buffers
    assert len(shapes) + len(O.buffer_shapes(cfg)) == 398
    assert sum(v for k, v in numel.items() if O.is_trainable(k)) == 43606192
    assert sum(numel.values()) == 84986800
    assert sum(v for k, v in numel.items() if not O.is_trainable(k)) == 41380608


def test_masks_bit_exact(golden_dir):
    g = load(golden_dir, "masks_schedules")
    for seed in (0, 1234, 99):
        mine = O.random_masks(4, O.DiGConfig(), 0.7, np.random.RandomState(seed)).numpy()
        assert np.array_equal(mine.astype(np.uint8), g[f"mask/{seed}"])
        assert (mine.sum(-1) == 179).all()


def test_schedules_bit_exact(golden_dir):
    g = load(golden_dir, "masks_schedules")
    assert np.array_equal(g["sched/lr"], O.cosine_scheduler(1.5e-4 * 4, 1e-5, 10, 50, warmup_epochs=1))
    assert np.array_equal(g["sched/lr_ws"], O.cosine_scheduler(6e-4, 1e-5, 10, 50, warmup_epochs=1, warmup_steps=20))
    assert np.array_equal(g["sched/wd"], O.cosine_scheduler(0.1, 0.1, 10, 50))
    assert np.array_equal(g["sched/moco_m"], np.array([O.adjust_moco_momentum(e / 7.0, 10, 0.99) for e in range(70)]))


def test_mim_target_gather_order():
    """Gather indices are bit-exact: ascending token id per sample, (p1 p2 c) element order."""
    cfg = O.DiGConfig()
    im, au, mk = O.synthetic_batch(3, cfg, 11)
    mask, labels = O.mim_targets(im, mk, cfg)
    assert not mask[:, 1].any()
    for b in range(3):
        idx = np.nonzero(mk[b, 0].numpy())[0]
        assert len(idx) == 179 and (np.diff(idx) > 0).all()
        for j in (0, 57, 178):
            t = idx[j]
            h, w = divmod(int(t), 32)
            patch = im[b, :, h * 4:(h + 1) * 4, w * 4:(w + 1) * 4] * 0.5 + 0.5   # [c,p1,p2]
            assert torch.equal(labels[0][b, j], patch.permute(1, 2, 0).reshape(-1))
