"""Step-level parity on the MI355X: the HIP engine step against (1) the CPU oracle on the same seeded inputs, (2) the
committed golden fixtures produced by the unmodified reference, and (3) size-independent properties at BASELINE.json's
full size (ViT-S, 128 samples/GPU).

Tolerances (bf16 compute vs fp32 oracle): losses / grad-norm |d| <= 2e-2*|x| + 1e-3; accuracies may move
by one sample; per-tensor gradient direction is judged against a yardstick: the same oracle run under CPU bf16 autocast
(1 - cos_hip <= 2 * (1 - cos_autocast) + 5e-3, norm ratio within 2x the autocast deviation + 3 %)."""
import dataclasses
import os

import numpy as np
import pytest
import torch

import dig_oracle as O
from gpu_util import build_model, run_engine_steps

pytestmark = pytest.mark.gpu


def close(a, b, rtol=2e-2, atol=1e-3):
    return abs(a - b) <= rtol * abs(b) + atol


@pytest.mark.parametrize("variant", ["tiny", "vit_tiny_widths_conv", "dis_only_conv"])
def test_tiny_step_vs_oracle_with_bf16_yardstick(variant):
    """vit_tiny_widths_conv: ConvPatchNet at ViT-Tiny's widths (192 -> 192 -> 288 -> 384 -> 384, 3 heads) -- the 288-channel map's im2col matrix
    has 2592 columns, padded to the GEMM's 64-element reduction granule (2624), in the forward and in the data gradient."""
    cfg = O.DiGConfig(**O.TINY)
    seed, B = 3, 4
    if variant == "vit_tiny_widths_conv":
        cfg = dataclasses.replace(O.make_config("pretrain_simmim_moco_ori_vit_tiny_patch4_32x128"), depth=2, patchnet="conv", num_windows=5,
                                  moco_mlp_dim=512)
        B = 16
    elif variant == "dis_only_conv":
        # pretrain_moco_ori_* with --patchnet_name conv: no pix_projector -- the extractor reads the encoder's own rows of both views and its
        # backward hands their gradient straight to the encoder
        cfg, B = dataclasses.replace(cfg, kind="moco", patchnet="conv", num_windows=5), 16
    hp = O.StepHyper(lr=1e-3)
    im, au, mk = O.synthetic_batch(B, cfg, seed * 1000)
    model = build_model(cfg, *O.det_state(cfg, seed))
    (stats,), _ = run_engine_steps(model, [(im, au, mk)], hp)
    grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}
    hp0 = dataclasses.replace(hp, moco_m=O.adjust_moco_momentum(0.0, 10, hp.moco_m))
    ref_m, ref_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        _, bf_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
    for k in ("loss", "loss_pixel", "loss_contrast", "grad_norm"):
        if k in stats or k in ref_m:
            assert close(stats[k], ref_m[k]), (k, stats[k], ref_m[k])
    for k in ("q1_acc1", "q1_acc5", "q2_acc1", "q2_acc5"):
        assert abs(stats[k] - ref_m[k]) <= (2 if cfg.patchnet == "conv" else 1) * 100.0 / (cfg.n_patch * B) + 1e-6, (k, stats[k], ref_m[k])
    cos = torch.nn.functional.cosine_similarity
    bad = []
    for n, g in grads.items():
        r = ref_g[n].reshape(1, -1)
        if r.norm() < 1e-7:
            continue
        c_hip, c_bf = cos(g.reshape(1, -1), r).item(), cos(bf_g[n].float().reshape(1, -1), r).item()
        q_hip, q_bf = (g.norm() / r.norm()).item(), (bf_g[n].float().norm() / r.norm()).item()
        # (norm band + 2 x the yardstick's own turn, as in test_step_vs_reference_golden_fixture: independent noise adds to the norm in
        #  quadrature -- a ConvPatchNet gradient that bf16 turns by 1 - cos = 5 % is 10 % longer than the fp32 one on either side)
        if (1 - c_hip) > 2 * (1 - c_bf) + 5e-3 or abs(q_hip - 1) > 2 * abs(q_bf - 1) + 3e-2 + 2 * (1 - c_bf):
            bad.append((n, c_hip, c_bf, q_hip, q_bf))
    assert not bad, bad
    # EMA'd momentum parameters (fp32 path) match tightly
    tr = O.OracleTrainer(cfg, *O.det_state(cfg, seed))
    O.ema_update(tr.P, hp0.moco_m)
    for n, p in model.named_parameters():
        if not O.is_trainable(n):
            assert (p.detach().cpu() - tr.P[n]).abs().max().item() < 1e-6, n


def test_zero_contrast_weight_skips_branch_but_matches_oracle():
    """BASELINE config 2 / epochs before contrast_start_epoch: loss_weight_contrast = 0.  The reference back-propagates
    exact zeros through the contrastive branch; the engine launches nothing for it and runs the encoder backward on
    view 0 only.  Same losses/meters, same gradients (zero where the reference's are zero), same parameters after AdamW."""
    cfg = O.DiGConfig(**O.TINY)
    seed, B = 5, 4
    hp = O.StepHyper(lr=1e-3, w_contrast=0.0)
    im, au, mk = O.synthetic_batch(B, cfg, seed * 1000)
    model = build_model(cfg, *O.det_state(cfg, seed))
    (stats,), _ = run_engine_steps(model, [(im, au, mk)], hp)
    grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}
    hp0 = dataclasses.replace(hp, moco_m=O.adjust_moco_momentum(0.0, 10, hp.moco_m))
    ref_m, ref_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        _, bf_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
    for k in ("loss", "loss_pixel", "loss_contrast", "grad_norm"):
        assert close(stats[k], ref_m[k]), (k, stats[k], ref_m[k])
    assert close(stats["loss"], stats["loss_pixel"], rtol=1e-6, atol=1e-7)
    cos = torch.nn.functional.cosine_similarity
    n_zero = 0
    for n, g in grads.items():
        r = ref_g[n].reshape(1, -1)
        if r.abs().max() == 0:                                            # contrastive-only parameters: exact zeros on both sides
            assert g.abs().max().item() == 0.0, n
            n_zero += 1
            continue
        c_hip, c_bf = cos(g.reshape(1, -1), r).item(), cos(bf_g[n].float().reshape(1, -1), r).item()
        q_hip, q_bf = (g.norm() / r.norm()).item(), (bf_g[n].float().norm() / r.norm()).item()
        assert (1 - c_hip) <= 2 * (1 - c_bf) + 5e-3 and abs(q_hip - 1) <= 2 * abs(q_bf - 1) + 3e-2, (n, c_hip, c_bf, q_hip, q_bf)
    assert n_zero >= 10, n_zero                                           # predictor + projector + pix_projector weights


@pytest.mark.parametrize("w_contrast", [0.1, 0.0])
def test_both_views_mim_vs_oracle(w_contrast):
    """only_mim_on_ori_img=False (engine_for_pretraining_moco.py:100-111,138-141; modeling_pretrain_moco_mim_ori.py:572-577): view 1
    keeps its mask in both encoders, the decoder runs on both views' masked rows, loss_pixel is the mean of the two MSEs (view 1
    against patches of the ORIGINAL crops).  With a zero contrastive weight the encoder backward still covers both views."""
    cfg = O.DiGConfig(**O.TINY)
    seed, B = 9, 4
    hp = O.StepHyper(lr=1e-3, w_contrast=w_contrast, only_mim_on_ori_img=False)
    im, au, mk = O.synthetic_batch(B, cfg, seed * 1000 + 3)
    model = build_model(cfg, *O.det_state(cfg, seed))
    (stats,), _ = run_engine_steps(model, [(im, au, mk)], hp)
    grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}
    hp0 = dataclasses.replace(hp, moco_m=O.adjust_moco_momentum(0.0, 10, hp.moco_m))
    ref_m, ref_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        _, bf_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
    for k in ("loss", "loss_pixel", "loss_contrast", "grad_norm"):
        assert close(stats[k], ref_m[k]), (k, stats[k], ref_m[k])
    cos = torch.nn.functional.cosine_similarity
    tot = float(np.sqrt(sum(float(r.norm()) ** 2 for r in ref_g.values())))
    for n, g in grads.items():
        r = ref_g[n].reshape(1, -1)
        if r.abs().max() == 0:
            assert g.abs().max().item() == 0.0, n
            continue
        c_hip, c_bf = cos(g.reshape(1, -1), r).item(), cos(bf_g[n].float().reshape(1, -1), r).item()
        q_hip, q_bf = (g.norm() / r.norm()).item(), (bf_g[n].float().norm() / r.norm()).item()
        # norm band: twice the yardstick's own deviation + 3 %, widened for tensors that are noisy by construction -- by the yardstick's
        # direction error (a vector that bf16 turns by 1 - cos = 2 % has 4 % of its energy in noise: its norm cannot be pinned to 3 %) and
        # by an absolute floor of 1e-4 of the whole gradient's norm (bf16 rounding noise scales with the activations upstream, not with a
        # small tensor's own norm).  An instruction-selection change in one kernel (-fno-slp-vectorize, round 3) moved two tensors of this
        # model across the old flat band by 0.2 % of their norm.
        band = 2 * abs(q_bf - 1) + 3e-2 + (1 - c_bf) + 1e-4 * tot / float(r.norm())
        assert (1 - c_hip) <= 2 * (1 - c_bf) + 5e-3 and abs(q_hip - 1) <= band, (n, c_hip, c_bf, q_hip, q_bf, band)
    assert model._last_idx_views[1].min().item() >= 0 and model._last_idx_views[1].max().item() < B * 256


def _fixture_step0(name):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    kw = {k: v for k, v in zip(g["cfg_keys"].tolist(), g["cfg_vals"].tolist())}
    ints = {"img_h", "img_w", "patch", "in_chans", "embed_dim", "depth", "heads", "dec_dim", "dec_classes", "moco_dim",
            "moco_mlp_dim", "pix_mlp_dim", "num_windows", "patchnet_depth"}
    if "cfg_kind" in g:
        kw["kind"] = str(g["cfg_kind"])
    if "cfg_patchnet" in g:
        kw["patchnet"] = str(g["cfg_patchnet"])
    cfg = O.DiGConfig(**{k: (int(v) if k in ints else v) for k, v in kw.items()})
    hpk = {k: v for k, v in zip(g["hp_keys"].tolist(), g["hp_vals"].tolist())}
    hpk["only_mim_on_ori_img"] = bool(hpk.get("only_mim_on_ori_img", 1.0))
    if "drop_seed" in hpk:
        hpk["drop_seed"] = int(hpk["drop_seed"])
    return g, cfg, O.StepHyper(**hpk), int(g["seed"]), int(g["B"])


@pytest.mark.parametrize("name", ["tiny_w1", "vit_small_b4_w1", "vit_base_b2_w1", "tiny_w1_c0", "tiny_w1_mim2", "tiny_w1_nw5", "tiny_dis_w1",
                                  "tiny_gen_w1", "tiny_gen_w1_mim2", "tiny_w1_dp", "tiny_w1_regular", "tiny_w1_conv"])
def test_step_vs_reference_golden_fixture(name):
    """Fixtures come from the UNMODIFIED reference engine (tests/golden, oracle/ref_harness/gen_golden.py).
    vit_small_b4_w1 is BASELINE.json configs[0] (the reference's own CPU-runnable case), vit_base_b2_w1 the model of
    configs[3], tiny_w1_c0 the loss of configs[1] (loss_weight_contrast = 0: the engine skips the contrastive backward), tiny_w1_mim2 the
    only_mim_on_ori_img=False variant (both views masked, a masked-pixel loss on each, view 1's target cut from the original crops)."""
    g, cfg, hp, seed, B = _fixture_step0(name)
    im, au, mk = O.synthetic_batch(B, cfg, seed * 1000)
    model = build_model(cfg, *O.det_state(cfg, seed), hp=hp)          # (tiny_w1_dp: --drop_path 0.3 under the fixture's mask keys)
    cap = {}
    def grab(mod, inp, out):                                              # (a forward hook must return None, or it replaces the output)
        assert ("vis_out" in out) == cfg.use_pixel and ("contra_loss" in out) == cfg.use_moco      # the reference's out_dict keys
        if "vis_out" not in out:
            return
        cap.setdefault("vis_out", out["vis_out"][0].detach().float().cpu().clone())
        if len(out["vis_out"]) > 1:
            cap.setdefault("vis_out1", out["vis_out"][1].detach().float().cpu().clone())
    hook = model.register_forward_hook(grab)
    (stats,), _ = run_engine_steps(model, [(im, au, mk)], hp)
    hook.remove()
    logged = ("loss", "loss_pixel", "loss_contrast", "q1_acc1", "q1_acc5", "q2_acc1", "q2_acc5")
    assert {k for k in logged if k in stats} == {k for k in logged if f"s0/stat/{k}" in g}      # a single-objective model logs its own loss only
    for k in ("loss", "loss_pixel", "loss_contrast", "grad_norm"):
        if k in stats:
            assert close(stats[k], float(g[f"s0/stat/{k}"])), (k, stats[k], float(g[f"s0/stat/{k}"]))
    for k in ("q1_acc1", "q1_acc5", "q2_acc1", "q2_acc5"):
        if k in stats:
            # (one rank flip among the B * n_patch rows of a view; ConvPatchNet's fixture has 8 rows per view: two)
            assert abs(stats[k] - float(g[f"s0/stat/{k}"])) <= (2 if cfg.patchnet == "conv" else 1) * 100.0 / (cfg.n_patch * B) + 1e-6
    grads = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.requires_grad}
    names, norms = g["s0/grad_names"].tolist(), g["s0/grad_norms"]
    tot = float(np.sqrt((norms ** 2).sum()))
    # per-tensor gradient norms against the REFERENCE's; yardstick = the oracle of the same step under CPU bf16 autocast (the device may
    # be at most twice as far from the reference as that run is, + 3 %; the round-1 form of this check was a flat 8 %)
    hp0 = dataclasses.replace(hp, moco_m=float(g["s0/stat/moco_m"]))
    torch.set_num_threads(max(8, min(32, os.cpu_count() or 8)))
    with torch.autocast("cpu", dtype=torch.bfloat16):
        _, bf_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
    # tiny models: the yardstick's DIRECTION error widens the norm band, as in test_both_views_mim_vs_oracle (a vector that bf16 turns by
    # 1 - cos = 2.5 % -- every tensor behind the 40-row BatchNorms of the patch transformer's fixture -- cannot have its norm pinned to 3 %);
    # the fixture holds norms and samples only, so the directions come from the fp32 oracle (itself pinned to this fixture on the CPU side)
    ref_g = None
    if cfg.embed_dim <= 128:
        _, ref_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
    cosf = torch.nn.functional.cosine_similarity
    for i, n in enumerate(names):
        if norms[i] > 1e-3 * tot:                                        # tensors that carry the gradient
            q_hip, q_bf = grads[n].norm().item() / norms[i], bf_g[n].float().norm().item() / norms[i]
            turn = (1 - cosf(bf_g[n].float().reshape(1, -1), ref_g[n].reshape(1, -1)).item()) if ref_g is not None else 0.0
            # (+ an absolute floor of 1e-4 of the whole gradient's norm: see test_both_views_mim_vs_oracle)
            # (2 x turn: the device's rounding noise is as large as the yardstick's and independent of it; the patch transformer's fixture at
            #  B = 4 has every tensor behind two 40-row BatchNorm stacks turned by 2.5 % -- tools/gpu_patchnet_probe.py shows the yardstick
            #  itself 3-4 % long on the same tensors at B = 16, and test_patch_transformer_forward_backward_vs_oracle pins the module alone to 3 %)
            assert abs(q_hip - 1) <= 2 * abs(q_bf - 1) + 3e-2 + 2 * turn + 1e-4 * tot / norms[i], (n, q_hip, q_bf, turn, norms[i] / tot)
    # mask gather order is bit-exact => vis_out rows line up with the reference's: compare the full [B, 179, 48] tensor, captured
    # from the step's own forward (before the optimizer touched the weights)
    if cfg.use_pixel:
        vis_ref = torch.from_numpy(g["s0/cap/vis_out/full"]).float()
        assert cap["vis_out"].shape == vis_ref.shape
        assert ((cap["vis_out"] - vis_ref).norm() / vis_ref.norm()).item() < 2e-2
        assert (cap["vis_out"] - vis_ref).abs().max().item() < 3e-2 * max(1.0, vis_ref.abs().max().item())
    if cfg.use_pixel and not hp.only_mim_on_ori_img:
        vis_ref1 = torch.from_numpy(g["s0/cap/vis_out1/full"]).float()
        assert cap["vis_out1"].shape == vis_ref1.shape
        assert ((cap["vis_out1"] - vis_ref1).norm() / vis_ref1.norm()).item() < 2e-2
    bn, bnorm = g["s0/buf_names"].tolist(), g["s0/buf_norms"]
    sd = model.state_dict()
    for i, n in enumerate(bn):
        # (a BN output re-projected without bias has an exactly-zero column mean in fp32; bf16 leaves ~4e-5/element)
        assert abs(sd[n].double().norm().item() - bnorm[i]) <= 2e-2 * bnorm[i] + 1e-4 * np.sqrt(sd[n].numel()), n
    if cfg.use_moco:
        assert int(sd["predictor.1.num_batches_tracked"]) == 1
    # post-step parameters: the EMA'd momentum tensors are well-conditioned (fp32 path): norms against the reference's own
    pn, pnorm = g["s0/param_names"].tolist(), g["s0/param_norms"]
    for i, n in enumerate(pn):
        if not O.is_trainable(n):
            assert abs(sd[n].double().norm().item() - pnorm[i]) <= 1e-5 * pnorm[i] + 1e-6, n
    if cfg.kind == "moco":
        # encoder.mask_token never gets a gradient in the reference (its unmasked encoder does not read it) and its AdamW skips the
        # parameter altogether: no decay either -- bit-identical to the loaded value after the step
        P0, _ = O.det_state(cfg, seed)
        assert "encoder.mask_token" not in names and torch.equal(sd["encoder.mask_token"].cpu(), P0["encoder.mask_token"])


@pytest.mark.parametrize("kind", ["moco", "simmim"])
def test_single_objective_two_steps_vs_oracle(kind):
    """The Dis-only / Gen-only models (pretrain_moco_ori_* / pretrain_simmim_ori_*, modeling_pretrain_moco_mim_ori.py:627-681) through the
    engine for two steps against the fp32 oracle (itself pinned by tiny_dis_w1 / tiny_gen_w1): losses, every tensor's gradient with the
    bf16-autocast yardstick, and the state the second step starts from."""
    cfg = dataclasses.replace(O.DiGConfig(**O.TINY), kind=kind)
    seed, B = 31, 8
    hp = O.StepHyper(lr=5e-4)
    batches = [O.synthetic_batch(B, cfg, 700 + s) for s in range(2)]
    model = build_model(cfg, *O.det_state(cfg, seed))
    keys = ("loss", "loss_contrast") if kind == "moco" else ("loss", "loss_pixel")
    tr = O.OracleTrainer(cfg, *O.det_state(cfg, seed))
    hps = [dataclasses.replace(hp, moco_m=O.adjust_moco_momentum(float(s), 10, hp.moco_m)) for s in range(2)]
    # ---- step 0: losses and every tensor's gradient, with the oracle's own bf16-autocast run as the yardstick
    (st0,), opt = run_engine_steps(model, batches[:1], hp)
    grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        _, bf_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(*batches[0], hps[0])
    ref, ref_g, _, _ = tr.step(*batches[0], hps[0])
    assert ("loss_pixel" in st0) == (kind == "simmim") and ("loss_contrast" in st0) == (kind == "moco")
    for k in keys:
        assert close(st0[k], ref[k]), (0, k, st0[k], ref[k])
    cos = torch.nn.functional.cosine_similarity
    tot = float(np.sqrt(sum(float(r.norm()) ** 2 for r in ref_g.values())))
    checked = 0
    for n, gq in grads.items():
        r = ref_g[n].reshape(1, -1)
        if float(r.norm()) < 1e-3 * tot:
            assert float(gq.norm()) < 2e-3 * tot + 1e-6, n              # (never-read / BatchNorm-cancelled tensors stay small)
            continue
        c_hip, c_bf = cos(gq.reshape(1, -1), r).item(), cos(bf_g[n].float().reshape(1, -1), r).item()
        q_hip, q_bf = (gq.norm() / r.norm()).item(), (bf_g[n].float().norm() / r.norm()).item()
        band = 2 * abs(q_bf - 1) + 3e-2 + (1 - c_bf) + 1e-4 * tot / float(r.norm())
        assert (1 - c_hip) <= 2 * (1 - c_bf) + 5e-3 and abs(q_hip - 1) <= band, (n, c_hip, c_bf, q_hip, q_bf, band)
        checked += 1
    assert checked > 10
    # ---- step 1 starts from the state step 0 left (parameters, Adam moments, EMA'd momentum networks, BatchNorm buffers)
    (st1,), _ = run_engine_steps(model, batches[1:], hp, start=1, opt=opt)
    ref1, _, _, _ = tr.step(*batches[1], hps[1])
    for k in keys:
        assert close(st1[k], ref1[k], rtol=3e-2, atol=3e-3), (1, k, st1[k], ref1[k])
    assert opt._step == 2


@pytest.mark.parametrize("n_img,nw", [(16, 5), (8, 4)])
def test_patch_transformer_forward_backward_vs_oracle(n_img, nw):
    """dig_amd/patchnet.py (PatchNet with its patch transformer, --patchnet_name regular) in isolation: output, every parameter gradient and
    the gradients w.r.t. the image tokens and the pooled windows against fp32 autograd through oracle.patch_extractor on the same bf16
    inputs -- the attention on the encoder's MFMA kernels with q_rows = num_windows."""
    from dig_amd import engine_core, patchnet, ops
    cfg = dataclasses.replace(O.DiGConfig(**O.TINY), patchnet="regular", num_windows=nw)
    P, S = O.det_state(cfg, 51)
    model = build_model(cfg, P, S)
    dev = torch.device("cuda:0")
    D, N = cfg.embed_dim, 256
    torch.manual_seed(3)
    feat = torch.randn(n_img * N, D, device=dev).bfloat16()
    dout = (torch.randn(n_img * nw, D, device=dev) * 0.1).bfloat16()
    ops.cast_f32_to_bf16(model._flat["online"], model.shadow("online"))
    model.flat_grads.zero_()
    step = engine_core._Step(model)
    pooled = torch.empty((n_img * nw, D), device=dev, dtype=torch.bfloat16)
    ops.window_pool_fwd(feat, pooled, n_img, model.gh, model.gw, nw, D)
    out, saved = patchnet.forward(step, feat, pooled, "patch_extractor", "online", n_img, True)
    dpooled, dfeat = patchnet.backward(step, dout, "patch_extractor", saved, n_img)
    torch.cuda.synchronize()
    # fp32 reference: the oracle's patch_extractor (pinned to the reference by tiny_w1_regular) with the pooling cut out of the graph
    Pf = {k: v.clone().requires_grad_(k.startswith("patch_extractor.")) for k, v in P.items()}
    xf = feat.float().cpu().view(n_img, N, D).requires_grad_(True)
    pf = pooled.float().cpu().view(n_img, nw, D).requires_grad_(True)
    import unittest.mock as um
    with um.patch.object(O, "window_pool", lambda x, c: pf):
        ref = O.patch_extractor(xf, Pf, "patch_extractor.", cfg)
    ref.backward(dout.float().cpu().view(n_img, nw, D))

    def close_(a, b, tol, name):
        a, b = a.detach().float().cpu().reshape(-1), b.detach().float().reshape(-1)
        err = ((a - b).norm() / b.norm()).item()
        assert err < tol, (name, err, (a.norm() / b.norm()).item())
    close_(out, ref, 2e-2, "out")
    close_(dpooled, pf.grad, 3e-2, "d pooled")
    close_(dfeat, xf.grad, 3e-2, "d tokens")
    for n_, sp in model.specs.items():
        if n_.startswith("patch_extractor."):
            g = model.flat_grads[sp.offset:sp.offset + sp.numel]
            close_(g, Pf[n_].grad, 3e-2, n_)


def test_gen_only_skips_the_unread_view():
    """Gen-only + only_mim_on_ori_img: the reference runs the augmented view through the encoder and reads none of its rows; the engine
    does not launch it -- vis_out and every gradient are what they are with ANY augmented view."""
    cfg = dataclasses.replace(O.DiGConfig(**O.TINY), kind="simmim")
    seed, B = 33, 4
    hp = O.StepHyper(lr=1e-3)
    im, au, mk = O.synthetic_batch(B, cfg, 900)
    res = []
    for aug in (au, torch.randn_like(au)):
        model = build_model(cfg, *O.det_state(cfg, seed))
        (st,), _ = run_engine_steps(model, [(im, aug, mk)], hp)
        res.append((st["loss"], model.flat_grads.clone()))
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])


def test_two_steps_carry_state():
    cfg = O.DiGConfig(**O.TINY)
    seed, B = 11, 4
    hp = O.StepHyper(lr=5e-4)
    batches = [O.synthetic_batch(B, cfg, 500 + s) for s in range(2)]
    model = build_model(cfg, *O.det_state(cfg, seed))
    stats, opt = run_engine_steps(model, batches, hp)
    tr = O.OracleTrainer(cfg, *O.det_state(cfg, seed))
    for s in range(2):
        ref, _, _, _ = tr.step(*batches[s], dataclasses.replace(hp, moco_m=O.adjust_moco_momentum(float(s), 10, hp.moco_m)))
        for k in ("loss", "loss_pixel", "loss_contrast"):
            assert close(stats[s][k], ref[k], rtol=3e-2, atol=3e-3), (s, k, stats[s][k], ref[k])
    assert opt._step == 2 and int(model.state_dict()["pix_projector.1.num_batches_tracked"]) == 2
    # parameters moved, in the same direction as the oracle's.  Two sign-like Adam steps turn small gradient noise into large
    # update noise, so the yardstick is the same oracle run under CPU bf16 autocast: the device may be at most twice as far from
    # the fp32 updates as that run is (the round-1 form of this check was a bare cosine > 0.7)
    P0, _ = O.det_state(cfg, seed)
    tb = O.OracleTrainer(cfg, *O.det_state(cfg, seed))
    with torch.autocast("cpu", dtype=torch.bfloat16):
        for s in range(2):
            tb.step(*batches[s], dataclasses.replace(hp, moco_m=O.adjust_moco_momentum(float(s), 10, hp.moco_m)))
    cos = torch.nn.functional.cosine_similarity
    for n in ("encoder.blocks.0.mlp.fc1.weight", "pix_decoder.0.weight", "encoder_projection_layer.3.weight", "encoder.blocks.1.attn.qkv.weight",
              "predictor.0.weight", "encoder.patch_embed.proj.weight"):
        d_hip = dict(model.named_parameters())[n].detach().cpu() - P0[n]
        d_ref = tr.P[n] - P0[n]
        d_bf = tb.P[n].float() - P0[n]
        c_hip, c_bf = cos(d_hip.reshape(1, -1), d_ref.reshape(1, -1)).item(), cos(d_bf.reshape(1, -1), d_ref.reshape(1, -1)).item()
        assert (1 - c_hip) <= 2 * (1 - c_bf) + 2e-2, (n, c_hip, c_bf)


@pytest.mark.parametrize("B", [1, 3, 5])
def test_ragged_batch_sizes_vs_oracle(B):
    """Batch sizes that leave ragged tiles everywhere (rows = 2*B*256 tokens, 8*B pooled rows, B*179 decoder rows)."""
    cfg = O.DiGConfig(**O.TINY)
    seed = 20 + B
    hp = O.StepHyper(lr=1e-3)
    im, au, mk = O.synthetic_batch(B, cfg, seed * 1000)
    model = build_model(cfg, *O.det_state(cfg, seed))
    (stats,), _ = run_engine_steps(model, [(im, au, mk)], hp)
    hp0 = dataclasses.replace(hp, moco_m=O.adjust_moco_momentum(0.0, 10, hp.moco_m))
    ref_m, ref_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
    for k in ("loss", "loss_pixel", "loss_contrast", "grad_norm"):
        assert close(stats[k], ref_m[k], rtol=3e-2, atol=2e-3), (k, stats[k], ref_m[k])
    with torch.autocast("cpu", dtype=torch.bfloat16):                      # noise yardstick: the same oracle in bf16
        _, bf_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
    grads = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.requires_grad}
    cos = torch.nn.functional.cosine_similarity
    tot = float(torch.sqrt(sum((g.double() ** 2).sum() for g in ref_g.values())))
    for n, r in ref_g.items():
        if r.norm() > 2e-2 * tot:                                         # the tensors that carry the gradient
            c_hip = cos(grads[n].reshape(1, -1), r.reshape(1, -1)).item()
            c_bf = cos(bf_g[n].float().reshape(1, -1), r.reshape(1, -1)).item()
            assert (1 - c_hip) <= 2 * (1 - c_bf) + 5e-3, (n, c_hip, c_bf)


def test_error_behaviour_matches_reference_engine():
    """engine_for_pretraining_moco.py:146-150: a non-finite loss prints and sys.exit(1)s; masks that do not select the same
    number of tokens per sample cannot be reshaped to [B, -1, C] (:107-111) -- here a RuntimeError.  Both surface from
    train_one_epoch (one step late inside an epoch, at the latest at its end)."""
    cfg = O.DiGConfig(**O.TINY)
    hp = O.StepHyper(lr=1e-3)
    im, au, mk = O.synthetic_batch(4, cfg, 77)
    model = build_model(cfg, *O.det_state(cfg, 1))
    bad = im.clone(); bad[1, 0, 3, 5] = float("nan")
    before = model._flat["online"].clone()
    with pytest.raises(SystemExit) as e:
        run_engine_steps(model, [(bad, au, mk)], hp)
    assert e.value.code == 1
    # the exit is resolved one step late, but the optimizer launch was gated on the (non-finite) gradient norm: whoever catches the
    # SystemExit still holds the last good weights, as with the reference (its check precedes the update, engine...:146-150)
    assert torch.equal(before, model._flat["online"]) and bool(torch.isfinite(model._flat["online"]).all())
    model = build_model(cfg, *O.det_state(cfg, 1))
    run_engine_steps(model, [(im, au, mk)], hp)                            # caches the per-sample mask count (179)
    ragged = mk.clone(); ragged[2, 0, int(torch.nonzero(ragged[2, 0])[0])] = 0
    with pytest.raises(RuntimeError, match="same number of tokens"):
        run_engine_steps(model, [(im, au, ragged)], hp)


def test_checkpoint_resume_is_bit_exact(tmp_path):
    """save_model / auto_load_model (utils/utils.py:546-651 layout): resuming after step 1 reproduces step 2 of the
    uninterrupted run bit for bit (deterministic kernels, optimizer state and BN buffers carried through the file)."""
    import types
    from dig_amd import utils as U
    from dig_amd.optim_factory import create_optimizer
    from gpu_util import engine_args
    cfg = O.DiGConfig(**O.TINY)
    hp = O.StepHyper(lr=1e-3)
    batches = [O.synthetic_batch(4, cfg, 900 + s) for s in range(2)]
    ref = build_model(cfg, *O.det_state(cfg, 2))
    _, _ = run_engine_steps(ref, batches, hp)
    a = build_model(cfg, *O.det_state(cfg, 2))
    _, opt = run_engine_steps(a, batches[:1], hp)
    args = types.SimpleNamespace(output_dir=str(tmp_path), auto_resume=True, resume="")
    U.save_model(args=args, epoch=0, model=a, model_without_ddp=a, optimizer=opt, loss_scaler=U.NativeScalerWithGradNormCount())
    b = build_model(cfg)                                                    # fresh (random) model + optimizer, then resume
    opt_b = create_optimizer(engine_args(hp), b)
    U.auto_load_model(args=args, model=b, model_without_ddp=b, optimizer=opt_b, loss_scaler=U.NativeScalerWithGradNormCount())
    assert args.start_epoch == 1
    run_engine_steps(b, batches[1:], hp, start=1, opt=opt_b)
    sd_ref, sd_b = ref.state_dict(), b.state_dict()
    for k in sd_ref:
        assert torch.equal(sd_ref[k], sd_b[k]), k


def _run_epoch(model, batches, hp, lr_values, w_contrast=None):
    from dig_amd.optim_factory import create_optimizer
    from dig_amd.engine_for_pretraining_moco import train_one_epoch
    from dig_amd.utils import NativeScalerWithGradNormCount
    from gpu_util import engine_args
    args = engine_args(hp)
    if w_contrast is not None:
        args.loss_weight_contrast = w_contrast
    opt = create_optimizer(args, model)
    loader = [([im, au, mk], torch.ones(1), torch.ones(1)) for im, au, mk in batches]
    stats = train_one_epoch(model, None, None, loader, None, opt, torch.device("cuda:0"), 0, NativeScalerWithGradNormCount(), None,
                            patch_size=4, normlize_target=False, start_steps=0, lr_schedule_values=lr_values,
                            wd_schedule_values=np.linspace(0.05, 0.1, len(batches)), args=args)
    return stats, opt


@pytest.mark.parametrize("w_contrast", [0.1, 0.0])
def test_graphed_step_equals_eager(w_contrast):
    """dig_amd/step_graph.py: nine steps with a moving lr / weight-decay / EMA-momentum schedule and fresh inputs every step, once with
    every step launched eagerly and once with steps 5..9 replayed from the captured HIP graph (step 1 eager, 2-4 the eager warm-up of
    the captured body): weights, momentum weights, BN buffers, both Adam moments and the logged meters are bit-identical."""
    cfg = O.DiGConfig(**O.TINY)
    hp = O.StepHyper(lr=1e-3)
    n = 9
    batches = [O.synthetic_batch(4, cfg, 4200 + s) for s in range(n)]
    lr_values = np.linspace(2e-4, 1e-3, n)
    a = build_model(cfg, *O.det_state(cfg, 5))
    a.step_graph = False
    st_a, opt_a = _run_epoch(a, batches, hp, lr_values, w_contrast)
    assert getattr(a, "_step_graph", None) is None
    b = build_model(cfg, *O.det_state(cfg, 5))
    b.step_graph = True
    st_b, opt_b = _run_epoch(b, batches, hp, lr_values, w_contrast)
    assert b._step_graph.replays == n - 4 and len(b._step_graph.graphs) == 1
    assert opt_a._step == opt_b._step == n
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert torch.equal(opt_a.exp_avg, opt_b.exp_avg) and torch.equal(opt_a.exp_avg_sq, opt_b.exp_avg_sq)
    assert torch.equal(a.flat_grads, b.flat_grads)
    for k in st_a:            # (round 4: the reported loss / accuracy sums are fixed-order too -- dig_mse_fwd_bwd_ws / dig_ce_rows_ws -- so the log is bit-stable)
        assert st_a[k] == st_b[k], (k, st_a[k], st_b[k])


def test_vit_small_b32_step_vs_oracle():
    """The BASELINE model (ViT-S, dim 256 / mlp 4096 heads) at a quarter of the per-GPU batch, one full step against the fp32
    oracle on the same inputs: losses, grad-norm, and the gradient direction of every parameter bucket."""
    cfg = O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128")
    seed, B = 31, 32
    hp = O.StepHyper(lr=1.5e-4 * B / 256)
    im, au, mk = O.synthetic_batch(B, cfg, seed * 1000)
    model = build_model(cfg, *O.det_state(cfg, seed))
    (stats,), _ = run_engine_steps(model, [(im, au, mk)], hp)
    grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}
    hp0 = dataclasses.replace(hp, moco_m=O.adjust_moco_momentum(0.0, 10, hp.moco_m))
    torch.set_num_threads(max(8, min(64, os.cpu_count() or 8)))
    ref_m, ref_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
    for k in ("loss", "loss_pixel", "loss_contrast", "grad_norm"):
        assert close(stats[k], ref_m[k]), (k, stats[k], ref_m[k])
    cos = torch.nn.functional.cosine_similarity
    buckets = {}
    for n, r in ref_g.items():
        key = ".".join(n.split(".")[:3]) if n.startswith("encoder.blocks.") else n.split(".")[0]
        a, b = buckets.setdefault(key, ([], []))
        a.append(grads[n].reshape(-1)); b.append(r.reshape(-1))
    for key, (a, b) in buckets.items():
        a, b = torch.cat(a), torch.cat(b)
        c, q = cos(a[None], b[None]).item(), (a.norm() / b.norm()).item()
        # (the BN-MLP heads normalise over only 4*B = 128 pooled rows: bf16 noise is amplified there, as in the bf16 oracle)
        assert c > (0.99 if key.startswith("encoder.") else 0.975) and abs(q - 1) < 3e-2, (key, c, q)


@pytest.fixture(scope="module")
def full_size():
    """BASELINE.json configs[1]/[2] per-GPU shape: ViT-S, 128 samples (256 images) per step."""
    from dig_amd.registry import create_model
    torch.manual_seed(0)
    model = create_model("pretrain_simmim_moco_ori_vit_small_patch4_32x128", pretrained=False, drop_path_rate=0.0,
                         drop_block_rate=None, mlp_dim=4096, dim=256, T=0.2, num_windows=4, encoder_type='vit',
                         queue_size=65536, patchnet_name='no_patchtrans').to("cuda:0")
    cfg = O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128")
    batch = O.synthetic_batch(128, cfg, 4242)
    return model, cfg, batch


def test_full_size_properties(full_size):
    model, cfg, (im, au, mk) = full_size
    hp = O.StepHyper(lr=1.5e-4 * 128 / 256)
    mom0 = model._flat["momentum"].clone()
    on0 = model._flat["online"][:model.n_ema].clone()
    (s1,), opt = run_engine_steps(model, [(im, au, mk)], hp)
    assert all(np.isfinite(v) for v in s1.values())
    assert 0.0 < s1["loss_pixel"] < 2.0 and 4.0 < s1["loss_contrast"] < 2 * 0.2 * 2 * np.log(512) + 1.0
    # EMA linearity on 41.4 M parameters: p_m' == m p_m + (1-m) p  (fp32, elementwise)
    m = O.adjust_moco_momentum(0.0, 10, hp.moco_m)
    assert (model._flat["momentum"] - (mom0 * m + on0 * (1 - m))).abs().max().item() < 1e-6
    # grad-norm meter == L2 norm of the flat gradient arena; pads of the arena stay exactly zero
    g = model.flat_grads
    assert abs(float(g.double().norm()) - s1["grad_norm"]) <= 1e-4 * s1["grad_norm"]
    used = torch.zeros_like(g, dtype=torch.bool)
    for n, sp in model.specs.items():
        if sp.arena == "online":
            used[sp.offset:sp.offset + sp.numel] = True
    assert float(g[~used].abs().max()) == 0.0
    # bit-exact mask indexing at full size: gather indices == torch.nonzero order
    mask0 = mk[:, 0].bool()
    ref_idx = torch.nonzero(mask0.reshape(-1)).squeeze(1).to(torch.int32)
    assert torch.equal(model._last_idx.reshape(-1).cpu(), ref_idx)
    # every trainable tensor received a finite, non-zero gradient (find_unused_parameters would find none)
    for n, p in model.named_parameters():
        if p.requires_grad:
            assert torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0, n


def test_full_size_loss_decreases_and_is_reproducible(full_size):
    model, cfg, (im, au, mk) = full_size
    hp = O.StepHyper(lr=1.5e-4 * 128 / 256)
    stats, _ = run_engine_steps(model, [(im, au, mk)] * 4, hp, start=1)
    assert stats[-1]["loss_pixel"] < stats[0]["loss_pixel"]            # same batch 4x: the MIM loss must go down
    # same weights, same batch, no_grad forward twice: bit-identical outputs (no atomics on the data path)
    with torch.no_grad():
        msk = O.mim_targets(im, mk, cfg)[0].cuda()
        mom = model._flat["momentum"].clone()
        o1 = model(im.cuda(), au.cuda(), msk, 1.0, True)                # m = 1.0: the EMA leaves the momentum weights unchanged
        v1 = o1["vis_out"][0].clone()
        o2 = model(im.cuda(), au.cuda(), msk, 1.0, True)
    assert torch.equal(mom, model._flat["momentum"])
    assert torch.equal(v1, o2["vis_out"][0])
    assert abs(float(o1["contra_loss"]) - float(o2["contra_loss"])) <= 1e-6 * abs(float(o1["contra_loss"]))


@pytest.mark.parametrize("case", ["tiny", "tiny_both_views", "vit_small_b8"])
def test_training_state_is_bit_reproducible_and_reads_no_unwritten_memory(case):
    """Several optimisation steps from the same seeds, three times in one process: as is, again, and after every block the caching allocator
    holds has been filled with NaN bit patterns (a kernel that reads a `torch.empty` region the step never wrote would produce a NaN or a
    changed number).  What must repeat bit for bit is every step's gradient norm and the final weights; the loss METERS are per-block partials
    added with one fp32 atomic per block and may differ in the last bit.  (tools/gpu_poison_check.py is the stand-alone form.)"""
    cfg = O.DiGConfig(**O.TINY) if case.startswith("tiny") else O.DiGConfig()
    hp = O.StepHyper(lr=1e-3, only_mim_on_ori_img=(case != "tiny_both_views"))
    B, n = (4, 6) if case.startswith("tiny") else (8, 3)
    batches = [O.synthetic_batch(B, cfg, 9100 + s) for s in range(n)]

    def run():
        model = build_model(cfg, *O.det_state(cfg, 29))
        stats, _ = run_engine_steps(model, batches, hp)
        torch.cuda.synchronize()
        return [float(s["grad_norm"]) for s in stats], model.flat_params.clone(), [float(s["loss"]) for s in stats]

    def poison():
        held = []
        for nbytes, count in ((64 << 20, 32), (4 << 20, 128), (256 << 10, 512), (16 << 10, 1024), (512, 2048)):
            for i in range(count):
                t = torch.empty(nbytes // 4, device="cuda:0", dtype=torch.int32)
                t.fill_(0x7FC07FC0 if i % 2 else 0x7FC00001)              # two bf16 NaNs / one fp32 NaN
                held.append(t)
        torch.cuda.synchronize()
        del held

    g1, w1, l1 = run()
    g2, w2, l2 = run()
    poison()
    g3, w3, l3 = run()
    assert all(v == v for v in g3 + l3) and bool(torch.isfinite(w3).all())
    assert g1 == g2 == g3 and torch.equal(w1, w2) and torch.equal(w1, w3)
    assert max(abs(a - b) for a, b in zip(l1, l3)) <= 1e-5 * max(abs(a) for a in l1)


def _bucket_cosines(grads, ref_g):
    cos = torch.nn.functional.cosine_similarity
    buckets = {}
    for n, r in ref_g.items():
        key = ".".join(n.split(".")[:3]) if n.startswith("encoder.blocks.") else n.split(".")[0]
        a, b = buckets.setdefault(key, ([], []))
        a.append(grads[n].reshape(-1)); b.append(r.reshape(-1))
    out = {}
    for key, (a, b) in buckets.items():
        a, b = torch.cat(a), torch.cat(b)
        out[key] = (cos(a[None], b[None]).item(), (a.norm() / b.norm()).item() if b.norm() > 0 else 1.0, b.norm().item())
    return out


@pytest.mark.parametrize("w_contrast", [0.1, 0.0])
def test_full_size_b128_step_vs_oracle(w_contrast):
    """BASELINE.json configs[2] (w_contrast 0.1) and configs[1] (0.0) at their FULL per-GPU size -- ViT-S, 128 samples = 256
    images, 65 536 token rows -- one train_one_epoch step against the fp32 oracle on the same inputs: the four losses /
    grad-norm, the accuracies, and the gradient direction and magnitude of every parameter bucket (for configs[1]: exact
    zeros where the reference's gradients are exact zeros).  The oracle step takes about a minute of host CPU."""
    cfg = O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128")
    seed, B = 41, 128
    hp = O.StepHyper(lr=1.5e-4 * B / 256, w_contrast=w_contrast)
    im, au, mk = O.synthetic_batch(B, cfg, seed * 1000)
    model = build_model(cfg, *O.det_state(cfg, seed))
    (stats,), _ = run_engine_steps(model, [(im, au, mk)], hp)
    grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}
    del model
    torch.cuda.empty_cache()
    hp0 = dataclasses.replace(hp, moco_m=O.adjust_moco_momentum(0.0, 10, hp.moco_m))
    torch.set_num_threads(max(8, min(64, os.cpu_count() or 8)))
    ref_m, ref_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
    for k in ("loss", "loss_pixel", "loss_contrast", "grad_norm"):
        assert close(stats[k], ref_m[k]), (k, stats[k], ref_m[k])
    for k in ("q1_acc1", "q1_acc5", "q2_acc1", "q2_acc5"):
        assert abs(stats[k] - ref_m[k]) <= 3 * 100.0 / (4 * B) + 1e-6, (k, stats[k], ref_m[k])     # a few rank flips among 512 queries
    for key, (c, q, rn) in _bucket_cosines(grads, ref_g).items():
        if rn == 0.0:                                                      # configs[1]: the contrastive-only buckets
            assert all(float(grads[n].abs().max()) == 0.0 for n in ref_g if n.split(".")[0] == key), key
            continue
        assert c > (0.99 if key.startswith("encoder.") else 0.975) and abs(q - 1) < 3e-2, (key, c, q)


def test_vit_base_b256_full_size_properties():
    """BASELINE.json configs[3] at full per-GPU size (the 'base' model of the family, 256 samples = 512 images, 131 072 token rows):
    size-independent properties of one step, and the MIM loss going down on a repeated batch."""
    from dig_amd.registry import create_model
    torch.manual_seed(0)
    model = create_model("pretrain_simmim_moco_ori_vit_base_patch4_32x128", pretrained=False, drop_path_rate=0.0, drop_block_rate=None,
                         mlp_dim=4096, dim=256, T=0.2, num_windows=4, encoder_type='vit', queue_size=65536,
                         patchnet_name='no_patchtrans').to("cuda:0")
    cfg = O.make_config("pretrain_simmim_moco_ori_vit_base_patch4_32x128")
    B = 256
    im, au, mk = O.synthetic_batch(B, cfg, 5151)
    hp = O.StepHyper(lr=1.5e-4 * B / 256)
    mom0 = model._flat["momentum"].clone()
    on0 = model._flat["online"][:model.n_ema].clone()
    stats, _ = run_engine_steps(model, [(im, au, mk)], hp)
    s1 = stats[0]
    assert all(np.isfinite(v) for v in s1.values())
    assert 0.0 < s1["loss_pixel"] < 2.0 and 4.0 < s1["loss_contrast"] < 2 * 0.2 * 2 * np.log(4 * B) + 1.0
    m = O.adjust_moco_momentum(0.0, 10, hp.moco_m)
    assert (model._flat["momentum"] - (mom0 * m + on0 * (1 - m))).abs().max().item() < 1e-6
    g = model.flat_grads
    assert abs(float(g.double().norm()) - s1["grad_norm"]) <= 1e-4 * s1["grad_norm"]
    ref_idx = torch.nonzero(mk[:, 0].bool().reshape(-1)).squeeze(1).to(torch.int32)
    assert torch.equal(model._last_idx.reshape(-1).cpu(), ref_idx)
    for n, p in model.named_parameters():
        if p.requires_grad:
            assert torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0, n
    more, _ = run_engine_steps(model, [(im, au, mk)] * 3, hp, start=1)
    assert more[-1]["loss_pixel"] < s1["loss_pixel"]


@pytest.mark.parametrize("which", ["tiny", "vit_small", "tiny_conv"])
def test_rccl_path_world1_matches_local_path(which):
    """The data-parallel wrapper on a one-rank 'nccl' (= RCCL) group: every collective of the N>1 path is issued
    (BN-statistics all-reduce, key all-gather, 17 gradient-bucket all-reduces, 1/world averaging) and the step must equal
    the plain single-process step.  `vit_small`: the model of the BASELINE configs, whose encoder blocks run their weight gradients on
    the grouped launch (a block's bucket becomes final -- and its all-reduce is issued -- one launch later)."""
    import torch.distributed as dist
    from dig_amd.parallel import DistributedDataParallel
    seed = 21
    if which == "tiny":
        cfg, B = O.DiGConfig(**O.TINY), 4
        P0, S0 = O.det_state(cfg, seed)
    elif which == "tiny_conv":
        # ConvPatchNet: six more BatchNorm layers per extractor whose statistics cross the ranks (SyncBatchNorm converts BatchNorm2d too)
        cfg, B = dataclasses.replace(O.DiGConfig(**O.TINY), patchnet="conv", num_windows=5), 8
        P0, S0 = O.det_state(cfg, seed)
    else:
        cfg, B = O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128"), 8
        P0, S0 = O.init_state(cfg, seed)
    state = lambda: ({k: v.clone() for k, v in P0.items()}, {k: v.clone() for k, v in S0.items()})
    hp = O.StepHyper(lr=1e-3)
    im, au, mk = O.synthetic_batch(B, cfg, 900)
    base = build_model(cfg, *state())
    from dig_amd import ops
    fused_bn, ops.BN_FUSED = ops.BN_FUSED, False      # (the single-process plan folds a few-row BatchNorm layer into one launch: another summation
    try:                                              #  order over rows than the statistics / all-reduce / apply launches of the process-group path)
        (s0,), _ = run_engine_steps(base, [(im, au, mk)], hp)
    finally:
        ops.BN_FUSED = fused_bn
    g0 = base.flat_grads.clone()
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29633")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        m = build_model(cfg, *state())
        ddp = DistributedDataParallel(m)
        assert m.comm.world == 1
        from gpu_util import engine_args
        from dig_amd.optim_factory import create_optimizer
        from dig_amd.engine_for_pretraining_moco import train_one_epoch
        from dig_amd.utils import NativeScalerWithGradNormCount
        args = engine_args(hp)
        opt = create_optimizer(args, ddp)
        # force the collective code paths even though world == 1
        m.comm.world_override = True
        s1 = train_one_epoch(ddp, None, None, [([im, au, mk], torch.ones(1), torch.ones(1))], None, opt, torch.device("cuda:0"), 0,
                             NativeScalerWithGradNormCount(), None, patch_size=4, normlize_target=False, start_steps=0,
                             lr_schedule_values=np.full(2, hp.lr), wd_schedule_values=np.full(2, hp.weight_decay), args=args)
        for k in ("loss", "loss_pixel", "loss_contrast", "grad_norm"):
            assert abs(s1[k] - s0[k]) <= 1e-3 * abs(s0[k]) + 1e-5, (k, s1[k], s0[k])
        assert not m.comm._pending
        rel = ((m.flat_grads - g0).norm() / g0.norm()).item()
        # every reduction on the gradient path is deterministic; the ViT-S forward differs between the two paths in WHICH stream plan runs it
        # (same kernels, same order per stream), not in arithmetic
        assert rel < 1e-6, rel
        # the ORDER of collectives on the communicator (a second step, recorded): it must be what every rank issues, or the first real
        # multi-GPU step dead-locks.  The committed list was recorded from this path (tools/gpu_collective_order.py).
        import json
        m.comm.log = []
        train_one_epoch(ddp, None, None, [([im, au, mk], torch.ones(1), torch.ones(1))], None, opt, torch.device("cuda:0"), 1,
                        NativeScalerWithGradNormCount(), None, patch_size=4, normlize_target=False, start_steps=1,
                        lr_schedule_values=np.full(3, hp.lr), wd_schedule_values=np.full(3, hp.weight_decay), args=args)
        got = [[op, n] for op, n in m.comm.log]
        m.comm.log = None
        if which == "tiny_conv":
            # 8 + 8 of the heads, 6 + 6 of the two extractors' forward, 6 of the online extractor's backward; its gradients travel with the projector's
            assert sum(1 for op, _ in got if op == "all_reduce") == 34 and sum(1 for op, _ in got if op == "all_gather") == 1
            assert sum(1 for op, _ in got if op.startswith("all_reduce_async:")) == cfg.depth + 2
            return
        want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", f"collective_order_{which}.json")))
        assert got == want["step"], (got[:6], want["step"][:6])
        # depth + 5 gradient buckets in depth + 2 messages (14 for the 12-block models: small neighbours travel together, MoCo_ViT.bucket_groups),
        # one fused key gather, 8 + 8 BatchNorm-statistics messages
        assert sum(1 for op, _ in got if op.startswith("all_reduce_async:")) == cfg.depth + 2 and sum(1 for op, _ in got if op == "all_gather") == 1
        assert sum(1 for op, _ in got if op == "all_reduce") == 16
    finally:
        if created:
            dist.destroy_process_group()


def test_clip_grad_two_steps_vs_reference_fixture():
    """--clip_grad on the pre-training path (utils/utils.py:487-493) against tests/golden/tiny_w1_clip.npz, written by the unmodified reference
    engine with max_norm = 1.0 (gradient norm 4-5: the clip is active).  Adam's update direction does not depend on the scale of one
    gradient, so the check is where the coefficient lives: the norms of both Adam moments after each of two steps."""
    g, cfg, hp, seed, B = _fixture_step0("tiny_w1_clip")
    assert hp.clip_grad == 1.0
    batches = [O.synthetic_batch(B, cfg, seed * 1000 + 17 * s) for s in range(2)]
    model = build_model(cfg, *O.det_state(cfg, seed))
    stats, opt = run_engine_steps(model, batches, hp)
    for s in range(2):
        for k in ("loss", "loss_pixel", "loss_contrast", "grad_norm"):
            assert close(stats[s][k], float(g[f"s{s}/stat/{k}"]), rtol=3e-2, atol=3e-3), (s, k, stats[s][k], float(g[f"s{s}/stat/{k}"]))
    names = g["s1/moment_names"].tolist()
    spec = model.specs                                                    # name -> (offset, numel, ...) of the flat arenas the moments mirror
    m1 = np.array([opt.exp_avg[spec[n].offset:spec[n].offset + spec[n].numel].double().norm().item() for n in names])
    m2 = np.array([opt.exp_avg_sq[spec[n].offset:spec[n].offset + spec[n].numel].double().norm().item() for n in names])
    r1, r2 = g["s1/exp_avg_norms"], g["s1/exp_avg_sq_norms"]
    tot1, tot2 = np.sqrt((r1 ** 2).sum()), np.sqrt((r2 ** 2).sum())
    assert abs(np.sqrt((m1 ** 2).sum()) / tot1 - 1) < 3e-2 and abs(np.sqrt((m2 ** 2).sum()) / tot2 - 1) < 6e-2
    for i, n in enumerate(names):
        if r1[i] > 1e-2 * tot1:                                           # tensors that carry the update (bf16 noise yardstick: 8 %)
            assert abs(m1[i] / r1[i] - 1) < 8e-2, (n, m1[i], r1[i])
    # an unclipped run would be off by the clip coefficient (1 / grad_norm ~ 0.23 at both steps)
    assert float(g["s0/stat/grad_norm"]) > 2.0 and float(g["s1/stat/grad_norm"]) > 2.0


def test_vit_small_b32_every_tensor_gradient_vs_oracle():
    """Every one of the 183 trainable tensors of the BASELINE model (ViT-S, B = 32) -- weights, biases, LayerNorm and BatchNorm vectors,
    mask token -- against the fp32 oracle, judged one by one with the CPU bf16-autocast run of the same oracle as the noise yardstick (a
    bucketed cosine hides a wrong bias next to its block's weight gradients)."""
    cfg = O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128")
    seed, B = 37, 32
    hp = O.StepHyper(lr=1.5e-4 * B / 256)
    im, au, mk = O.synthetic_batch(B, cfg, seed * 1000)
    model = build_model(cfg, *O.det_state(cfg, seed))
    (stats,), _ = run_engine_steps(model, [(im, au, mk)], hp)
    grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}
    assert len(grads) == 183
    hp0 = dataclasses.replace(hp, moco_m=O.adjust_moco_momentum(0.0, 10, hp.moco_m))
    torch.set_num_threads(max(8, min(64, os.cpu_count() or 8)))
    ref_m, ref_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        _, bf_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
    cos = torch.nn.functional.cosine_similarity
    tot = float(torch.sqrt(sum((r.double() ** 2).sum() for r in ref_g.values())))
    bad, checked = [], 0
    for n, g in grads.items():
        r = ref_g[n].reshape(1, -1)
        if float(r.norm()) < 1e-7 * tot:
            continue                                                       # (exactly-zero reference gradients: nothing to compare a direction with)
        checked += 1
        c_hip, c_bf = cos(g.reshape(1, -1), r).item(), cos(bf_g[n].float().reshape(1, -1), r).item()
        q_hip, q_bf = (g.norm() / r.norm()).item(), (bf_g[n].float().norm() / r.norm()).item()
        if (1 - c_hip) > 2 * (1 - c_bf) + 5e-3 or abs(q_hip - 1) > 2 * abs(q_bf - 1) + 3e-2:
            bad.append((n, round(c_hip, 5), round(c_bf, 5), round(q_hip, 4), round(q_bf, 4)))
    assert checked >= 180 and not bad, bad


def test_twenty_step_trajectory_vs_oracle_band():
    """Twenty optimisation steps of the tiny model with MOVING schedules (warm-up + cosine lr, weight decay ramp, cosine EMA momentum) and a
    fresh batch per step: what drifts at step 10 and not at step 2 -- EMA-before-forward ordering, BatchNorm running buffers, schedule
    indexing, the lagged meter read-back.  Band: at every step the device loss may be at most twice as far from the fp32 oracle's as the
    CPU bf16-autocast oracle's is (+ 3 %); BatchNorm buffers and counters after step 20."""
    from dig_amd.optim_factory import create_optimizer
    from dig_amd.engine_for_pretraining_moco import train_one_epoch
    from dig_amd.utils import NativeScalerWithGradNormCount
    from gpu_util import engine_args
    cfg = O.DiGConfig(**O.TINY)
    seed, B, n, epochs = 41, 4, 20, 25
    # Peak lr 2.5e-4 (the recipe's own is 1.5e-4 at batch 256): at 1e-3 AdamW moves every weight by its own scale within 20 steps and the
    # trajectory is chaotic -- the fp32 CPU oracle itself then differs by 6e-4 in loss_contrast at step 16 between two runs with different
    # thread counts, and the device's distance from it flips between 0.84 and 1.29 of the band with the seed (tools: three seeds x two lr
    # scales measured; the device trajectory is bit-identical run to run and independent of what ran before in the process).
    hp = O.StepHyper(lr=2.5e-4)
    lr = 0.25 * np.concatenate([np.linspace(1e-4, 1e-3, 5), 5e-5 + 0.5 * (1e-3 - 5e-5) * (1 + np.cos(np.pi * np.arange(15) / 15))])
    wd = np.linspace(0.05, 0.1, n)
    batches = [O.synthetic_batch(B, cfg, 7000 + s) for s in range(n)]
    model = build_model(cfg, *O.det_state(cfg, seed))
    args = engine_args(hp, epochs=epochs)
    opt = create_optimizer(args, model)
    dev_stats = []
    for s, (im, au, mk) in enumerate(batches):
        dev_stats.append(train_one_epoch(model, None, None, [([im, au, mk], torch.ones(1), torch.ones(1))], None, opt, torch.device("cuda:0"), s,
                                         NativeScalerWithGradNormCount(), None, patch_size=4, normlize_target=False, start_steps=s,
                                         lr_schedule_values=lr, wd_schedule_values=wd, args=args))
    def oracle(autocast):
        tr = O.OracleTrainer(cfg, *O.det_state(cfg, seed))
        out = []
        for s, (im, au, mk) in enumerate(batches):
            hps = dataclasses.replace(hp, lr=float(lr[s]), weight_decay=float(wd[s]), moco_m=O.adjust_moco_momentum(float(s), epochs, hp.moco_m))
            if autocast:
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    m, _, _, _ = tr.step(im, au, mk, hps)
            else:
                m, _, _, _ = tr.step(im, au, mk, hps)
            out.append(m)
        return out, tr
    ref, tr = oracle(False)
    bf, tb = oracle(True)
    for s in range(n):
        assert dev_stats[s]["lr"] == pytest.approx(float(lr[s]), rel=1e-6) and dev_stats[s]["weight_decay"] == pytest.approx(float(wd[s]), rel=1e-6)
        assert dev_stats[s]["moco_m"] == pytest.approx(O.adjust_moco_momentum(float(s), epochs, hp.moco_m), rel=1e-6)
        for k in ("loss", "loss_pixel", "loss_contrast"):
            d_hip, d_bf = abs(dev_stats[s][k] - ref[s][k]), abs(bf[s][k] - ref[s][k])
            # (InfoNCE over 8 queries x 8 keys at T = 0.2 is the ill-conditioned meter of the three: 5 % instead of 3 %)
            assert d_hip <= 2 * d_bf + (5e-2 if k == "loss_contrast" else 3e-2) * abs(ref[s][k]) + 3e-3, (s, k, dev_stats[s][k], ref[s][k], bf[s][k])
    assert ref[-1]["loss_pixel"] < ref[0]["loss_pixel"] and dev_stats[-1]["loss_pixel"] < dev_stats[0]["loss_pixel"]      # it trains
    sd = model.state_dict()
    for k, v in tr.S.items():
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == n == int(v), k
        else:
            a, b, c = sd[k].float().cpu(), v.float(), tb.S[k].float()
            assert (a - b).norm() <= 2 * (c - b).norm() + 3e-2 * b.norm() + 1e-4 * np.sqrt(b.numel()), (k, float((a - b).norm()), float((c - b).norm()), float(b.norm()))
    assert opt._step == n
    # the momentum (EMA) weights followed the schedule: they are compared with the oracle's in the same band
    for name in ("momentum_encoder.blocks.0.mlp.fc1.weight", "momentum_projection_layer.0.weight", "momentum_encoder.patch_embed.proj.weight"):
        a, b, c = dict(model.named_parameters())[name].detach().float().cpu(), tr.P[name], tb.P[name].float()
        assert (a - b).norm() <= 2 * (c - b).norm() + 1e-3 * b.norm(), (name, float((a - b).norm()), float((c - b).norm()))


def test_vit_small_b32_twenty_step_trajectory_vs_oracle():
    """Twenty optimisation steps of the FULL ViT-S model at B = 32 (the batch averages over 32 x 179 masked patches and 128 x 128 logits,
    which damps the chaos that the 8 x 8-logit tiny trajectory shows) with the recipe's schedules in miniature: linear warm-up to the
    recipe's peak lr for its own global batch (1.5e-4 x 1024 / 256 = 6e-4), cosine decay, weight-decay ramp, cosine EMA momentum, a
    fresh batch every step.  Band: every loss of every step within 3 % of the fp32 oracle's (+ 3e-3 absolute); no bf16 yardstick."""
    from dig_amd.optim_factory import create_optimizer
    from dig_amd.engine_for_pretraining_moco import train_one_epoch
    from dig_amd.utils import NativeScalerWithGradNormCount
    from gpu_util import engine_args
    cfg = O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128")
    seed, B, n, epochs = 3, 32, 20, 25
    peak = 6e-4
    hp = O.StepHyper(lr=peak)
    lr = np.concatenate([np.linspace(peak / 10, peak, 5), 1e-5 + 0.5 * (peak - 1e-5) * (1 + np.cos(np.pi * np.arange(15) / 15))])
    wd = np.linspace(0.05, 0.1, n)
    batches = [O.synthetic_batch(B, cfg, 9000 + s) for s in range(n)]
    torch.set_num_threads(max(1, min(16, __import__("os").cpu_count() or 8)))
    tr = O.OracleTrainer(cfg, seed=seed)
    model = build_model(cfg, {k: v.detach().clone() for k, v in tr.P.items()}, {k: v.detach().clone() for k, v in tr.S.items()})
    args = engine_args(hp, epochs=epochs)
    opt = create_optimizer(args, model)
    worst = {}
    for s, (im, au, mk) in enumerate(batches):
        st = train_one_epoch(model, None, None, [([im, au, mk], torch.ones(1), torch.ones(1))], None, opt, torch.device("cuda:0"), s,
                             NativeScalerWithGradNormCount(), None, patch_size=4, normlize_target=False, start_steps=s,
                             lr_schedule_values=lr, wd_schedule_values=wd, args=args)
        hps = dataclasses.replace(hp, lr=float(lr[s]), weight_decay=float(wd[s]), moco_m=O.adjust_moco_momentum(float(s), epochs, hp.moco_m))
        ref, _, _, _ = tr.step(im, au, mk, hps)
        for k in ("loss", "loss_pixel", "loss_contrast"):
            d = abs(st[k] - ref[k])
            worst[k] = max(worst.get(k, 0.0), d / (3e-2 * abs(ref[k]) + 3e-3))
            assert d <= 3e-2 * abs(ref[k]) + 3e-3, (s, k, st[k], ref[k])
    print("worst fraction of the 3 % band used:", {k: round(v, 3) for k, v in worst.items()})
    assert opt._step == n


def test_vit_base_b16_every_tensor_gradient_vs_oracle():
    """BASELINE configs[3]'s model (the "base" factory: D = 512, 8 heads, F = 2048) at B = 16: every trainable tensor against the fp32 oracle
    with the bf16-autocast yardstick, as for ViT-S above -- the D = 512 forms of the grouped weight gradients (fn = 2) and the 256-row
    data-gradient tiles are on this path, the fused MLP / attention launches (D = 384 only) are not."""
    cfg = O.make_config("pretrain_simmim_moco_ori_vit_base_patch4_32x128")
    seed, B = 41, 16
    hp = O.StepHyper(lr=1.5e-4 * B / 256)
    im, au, mk = O.synthetic_batch(B, cfg, seed * 1000)
    model = build_model(cfg, *O.det_state(cfg, seed))
    (stats,), _ = run_engine_steps(model, [(im, au, mk)], hp)
    grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}
    hp0 = dataclasses.replace(hp, moco_m=O.adjust_moco_momentum(0.0, 10, hp.moco_m))
    torch.set_num_threads(max(8, min(64, os.cpu_count() or 8)))
    ref_m, ref_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        _, bf_g, _, _ = O.OracleTrainer(cfg, *O.det_state(cfg, seed)).loss_and_grads(im, au, mk, hp0)
    for k in ("loss", "loss_pixel", "loss_contrast"):
        assert close(stats[k], ref_m[k]), (k, stats[k], ref_m[k])
    cos = torch.nn.functional.cosine_similarity
    tot = float(torch.sqrt(sum((r.double() ** 2).sum() for r in ref_g.values())))
    bad, checked = [], 0
    for n, g in grads.items():
        r = ref_g[n].reshape(1, -1)
        if float(r.norm()) < 1e-7 * tot:
            continue
        checked += 1
        c_hip, c_bf = cos(g.reshape(1, -1), r).item(), cos(bf_g[n].float().reshape(1, -1), r).item()
        q_hip, q_bf = (g.norm() / r.norm()).item(), (bf_g[n].float().norm() / r.norm()).item()
        if (1 - c_hip) > 2 * (1 - c_bf) + 5e-3 or abs(q_hip - 1) > 2 * abs(q_bf - 1) + 3e-2:
            bad.append((n, round(c_hip, 5), round(c_bf, 5), round(q_hip, 4), round(q_bf, 4)))
    assert checked >= len(grads) - 4 and not bad, bad


def test_optimizer_launch_leaves_shadow_and_transposes_for_the_next_forward():
    """dig_adamw_step_tr (the default: optim_factory.FOLD_SHADOW) writes the bf16 operand shadow and the transposed MLP / projection
    weights, and the next forward launches neither its cast nor its four transposes: four steps are bit-identical to the plan that
    rebuilds both every step; a parameter written behind the optimizer's back (a torch in-place operation on the arena, load_state_dict)
    is seen and the copies are rebuilt."""
    from dig_amd import ops, optim_factory
    # (ViT-S width, two blocks: the fused MLP backward -- the reader of the transposed copies -- exists for D = 384 only)
    cfg = dataclasses.replace(O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128"), depth=2)
    seed, B = 41, 4
    hp = O.StepHyper(lr=1e-3)
    batches = [O.synthetic_batch(B, cfg, 300 + s) for s in range(4)]
    assert ops.mlp_chain_supported(cfg.embed_dim, cfg.hidden, 2 * B * 256)

    def run(fold, poke):
        old = optim_factory.FOLD_SHADOW
        optim_factory.FOLD_SHADOW = fold
        counts = {"cast": 0, "tr": 0}
        oc, ot = ops.cast_f32_to_bf16, ops.transpose_bf16_multi
        try:
            model = build_model(cfg, *O.init_state(cfg, seed))
            big = model.flat_params.numel()

            def cast(x, y):
                counts["cast"] += int(x.numel() == big)
                return oc(x, y)

            def tr(srcs, outs=None):
                counts["tr"] += 1
                return ot(srcs, outs)
            ops.cast_f32_to_bf16, ops.transpose_bf16_multi = cast, tr
            stats, opt = run_engine_steps(model, batches[:2], hp)
            if poke == "inplace":
                model.flat_params.mul_(1.0009765625)              # a torch in-place write: bumps the arena's version counter
            elif poke == "load":
                model.load_state_dict({k: v * 1.0009765625 if v.is_floating_point() and k in model.specs else v for k, v in model.state_dict().items()})
            st2, _ = run_engine_steps(model, batches[2:], hp, start=2, opt=opt)
            torch.cuda.synchronize()
            return [s_["loss"] for s_ in stats + st2], model.flat_params.clone(), dict(counts)
        finally:
            ops.cast_f32_to_bf16, ops.transpose_bf16_multi = oc, ot
            optim_factory.FOLD_SHADOW = old

    for poke in (None, "inplace", "load"):
        l1, p1, c1 = run(True, poke)
        l0, p0, c0 = run(False, poke)
        assert l1 == l0 and torch.equal(p1, p0), poke
        assert c0 == {"cast": 4, "tr": 16}                               # (four weight kinds: fc2, fc1, proj, qkv)
        # folded: the first forward (and the one behind a foreign write) rebuilds, the others launch nothing
        assert c1 == ({"cast": 1, "tr": 4} if poke is None else {"cast": 2, "tr": 8}), (poke, c1)


@pytest.mark.parametrize("switch", ["wgrad_off", "wgrad_pair", "wgrad_wa1", "wgrad_side_stream", "chain_mask3", "dgrad_128", "fwd_side", "fwd_serial",
                                    "chain_no_ln", "bwd_single", "chain_bwd_every2", "per_entry_point", "autograd_function", "attn_three_launches",
                                    "attn_bwd_single_pass", "wgrad_inline", "ln2_bwd_own_launch", "proj_dgrad_in_chain", "bn_three_launches",
                                    "dgrad_transpose_read", "adamw_plain", "head_dgrad_transpose_read", "proj_dgrad_in_attn_bwd",
                                    "attn_bwd_row_stores", "reductions_per_block", "per_entry_point_deferred"])
def test_engine_switches_agree_with_the_default_path(switch):
    """Every non-default execution plan of the step (environment switches of dig_amd/ops.py and engine_core.py: launch groupings, tile
    codes, stream plans, fusion masks) against the default plan on one ViT-S step from the same state and batch: the losses agree, and
    every gradient tensor agrees -- to fp32 summation order (1e-5) where only the grouping / placement of launches changes; to 2e-2
    where a different BACKWARD kernel produces the data gradients (same saved activations, bf16 rounding of the gradient tensors);
    and, where a different FORWARD kernel re-rounds every activation (the LayerNorm fusion), by direction and length per
    tensor (cosine >= 0.95, norm within 10 %): twelve blocks of softmax attention amplify a last-bit change of an activation into a
    ~25 % relative difference of the patch-embedding gradient at random init and B = 8 -- the same spread the fp32 oracle's own bf16
    autocast shows in test_vit_small_b32_every_tensor_gradient_vs_oracle.
    "per_entry_point" turns the one-call-per-encoder-block path (dig_encoder_block_fwd / _bwd, the default) off: the per-entry-point plan
    launches the same kernels with the same arguments in the same order, so every loss value and every gradient must agree BIT FOR BIT."""
    from dig_amd import ops, engine_core
    cfg = O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128")
    B = 8
    P, S = O.init_state(cfg, 11)
    batch = O.synthetic_batch(B, cfg, 77)
    hp = O.StepHyper(lr=1e-4)

    def one_step():
        model = build_model(cfg, {k: v.clone() for k, v in P.items()}, {k: v.clone() for k, v in S.items()})
        stats, _ = run_engine_steps(model, [batch], hp)
        torch.cuda.synchronize()
        return stats[0], model.flat_grads.detach().float().clone(), {n: sp for n, sp in model.specs.items() if sp.arena == "online"}

    if switch == "per_entry_point":
        engine_core.RED_DEFER, red_saved = False, engine_core.RED_DEFER   # (both sides of this comparison launch the reductions block by block)
    try:
        ref_stats, ref_g, specs = one_step()
    finally:
        if switch == "per_entry_point":
            engine_core.RED_DEFER = red_saved
    plans = {
        "wgrad_off": [(engine_core, "WGRAD_GROUPING", "off")], "wgrad_pair": [(engine_core, "WGRAD_GROUPING", "pair")],
        "wgrad_wa1": [(ops, "WGRAD_GROUP_WA", 1), (ops, "WGRAD_GROUP_SLOTS", 512)], "wgrad_side_stream": [(engine_core, "WGRAD_INLINE", False)],
        "chain_mask3": [(ops, "MLP_CHAIN_MASK", 3)], "dgrad_128": [(ops, "DGRAD_BK", 0)], "fwd_side": [(engine_core, "FWD_MODE", "side")],
        "fwd_serial": [(engine_core, "FWD_MODE", "serial")], "chain_no_ln": [(ops, "MLP_CHAIN_LN", False)],
        "bwd_single": [(engine_core, "BWD_SINGLE_STREAM", True)], "chain_bwd_every2": [(engine_core, "CHAIN_BWD_EVERY", 2)],
        # (the per-entry-point plan launches every block's reductions by themselves: compared with the block-call path doing the same)
        "per_entry_point": [(ops, "BLOCK_CALLS", False), (engine_core, "RED_DEFER", False)], "autograd_function": [(engine_core, "STEP_OPS", False)],
        # the blocks' bias / LayerNorm-parameter reductions per block on the second stream instead of one launch behind the last data gradient:
        # another grouping of the same partial rows (64 against 128 row groups for the LayerNorm vectors)
        "reductions_per_block": [(engine_core, "RED_DEFER", False)],
        # the per-entry-point plan with ITS held-back reductions (one launch behind the block loop) against the block-call path with its own:
        # the same partial rows through the same kernel
        "per_entry_point_deferred": [(ops, "BLOCK_CALLS", False)],
        # the attention sub-block as three launches (qkv GEMM -> dig_attn_fwd -> proj GEMM) instead of dig_attn_block_fwd: a different FORWARD kernel
        "attn_three_launches": [(ops, "ATTN_BLOCK", False)],
        "attn_bwd_single_pass": [],                                       # (a library-wide mode, set below: a different BACKWARD kernel)
        # norm2's backward as its own launch behind dig_mlp_chain_bwd instead of inside dig_mlp_chain_bwd_ln: a different BACKWARD kernel
        "ln2_bwd_own_launch": [(ops, "MLP_CHAIN_LNB", False)],
        # the projection's data gradient behind norm2's backward in dig_mlp_chain_bwd_ln_proj instead of its own GEMM launch (opt-in plan)
        "proj_dgrad_in_chain": [(ops, "MLP_CHAIN_PROJ", True)],
        # the heads' few-row BatchNorm layers as statistics / apply launches (the plan under a process group) instead of one fused launch
        # each: another summation order of the column statistics
        "bn_three_launches": [(ops, "BN_FUSED", False)],
        # the proj / qkv data gradients in their transpose-read form on the [out, in] weights instead of the direct form on the transposed copies:
        # bit-identical; and the plain optimizer launch (cast + transposes rebuilt by the forward): one step from a loaded state is the same step
        "dgrad_transpose_read": [(ops, "DGRAD_DIRECT", False)],
        "adamw_plain": [],
        "head_dgrad_transpose_read": [(ops, "HEAD_DGRAD_DIRECT", False)],  # the heads' data gradients on the [out, in] weights: another K split of the 4096-deep layers
        # the projection's data gradient inside the attention backward launch (dig_attn_bwd_proj, opt-in plan): d(ctx) per (image, head) workgroup,
        # another K order inside it -- a different BACKWARD kernel in front of the same attention backward
        "proj_dgrad_in_attn_bwd": [(ops, "ATTN_BWD_PROJ", True)],
        "attn_bwd_row_stores": [],                                        # (a library-wide mode, set below: the same results, bit for bit)
        "wgrad_inline": [(engine_core, "WGRAD_DEFER", "0")],              # the grouped launch of a block inside its data-gradient chain (the plan under a
                                                                          # process group) instead of all twelve behind the last data gradient: same sums
    }
    tight = switch in ("wgrad_off", "wgrad_pair", "wgrad_wa1", "wgrad_side_stream", "fwd_side", "fwd_serial", "bwd_single", "wgrad_inline",
                       "reductions_per_block")
    if switch == "adamw_plain":
        from dig_amd import optim_factory
        plans[switch] = [(optim_factory, "FOLD_SHADOW", False)]
    saved = [(m, k, getattr(m, k)) for m, k, _ in plans[switch]]
    prev_mode = ops.attn_bwd_mode(True) if switch == "attn_bwd_single_pass" else None
    prev_store = ops.attn_bwd_store(0) if switch == "attn_bwd_row_stores" else None
    try:
        for m, k, v in plans[switch]:
            setattr(m, k, v)
        if switch == "attn_three_launches":                               # (the cached block-call tables carry the fuse flag)
            assert ops.attn_block_supported(cfg.heads, cfg.embed_dim) is False
        stats, g, _ = one_step()
    finally:
        for m, k, v in saved:
            setattr(m, k, v)
        if prev_mode is not None:
            ops.attn_bwd_mode(prev_mode)
        if prev_store is not None:
            assert prev_store == 3, "full-line non-temporal stores are the default"
            ops.attn_bwd_store(prev_store)
    if switch == "autograd_function":
        # the default dispatches the step as the registered operators dig::pretrain_step_fwd / _bwd; the autograd.Function form runs the same code
        assert engine_core.STEP_OPS and torch.ops.dig.pretrain_step_fwd is not None and not engine_core._LIVE_STEPS
        assert stats["grad_norm"] == ref_stats["grad_norm"] and torch.equal(g, ref_g)
        return
    if switch == "attn_bwd_row_stores":
        # the same kernel, another way out for its results: every gradient bit for bit -- except the q_bias gradients, whose sums the full-line form
        # takes from the rows as stored (bf16) and the row-store form from the fp32 accumulators (2e-3: bf16 rounding of the summands)
        assert all(stats[k] == ref_stats[k] for k in ("loss", "loss_pixel", "loss_contrast")), (stats, ref_stats)
        assert abs(stats["grad_norm"] - ref_stats["grad_norm"]) <= 1e-5 * ref_stats["grad_norm"]
        for name, sp in specs.items():
            a, b = g[sp.offset:sp.offset + sp.numel], ref_g[sp.offset:sp.offset + sp.numel]
            if name.endswith("attn.q_bias"):
                assert float((a - b).norm() / b.norm()) <= 2e-3, name
            else:
                assert torch.equal(a, b), name
        return
    if switch in ("dgrad_transpose_read", "adamw_plain", "per_entry_point_deferred"):
        # same products, same K order per output element / the same step from a loaded state: bit for bit
        assert all(stats[k] == ref_stats[k] for k in ("loss", "loss_pixel", "loss_contrast", "grad_norm")), (stats, ref_stats)
        assert torch.equal(g, ref_g)
        return
    if switch == "per_entry_point":
        # gradients may be WRITTEN instead of added only right after zero_grad(): the flag is consumed by the backward
        m_ = build_model(cfg, {k: v.clone() for k, v in P.items()}, {k: v.clone() for k, v in S.items()})
        assert not getattr(m_, "_grads_fresh", False)
        _, opt_ = run_engine_steps(m_, [batch], hp)
        assert m_._grads_fresh is False
        opt_.zero_grad()
        assert m_._grads_fresh is True
        assert ops.BLOCK_CALLS, "the block-call path is the default"
        # (the reported loss values are fixed-order sums as well: dig_mse_fwd_bwd_ws / dig_ce_rows_ws)
        assert all(stats[k] == ref_stats[k] for k in ("loss", "loss_pixel", "loss_contrast", "grad_norm")), (stats, ref_stats)
        assert torch.equal(g, ref_g)
        return
    for k in ("loss", "loss_pixel", "loss_contrast", "grad_norm"):
        assert abs(stats[k] - ref_stats[k]) <= (1e-5 if tight else 2e-2) * abs(ref_stats[k]) + 1e-6, (k, stats[k], ref_stats[k])      # (2e-2: this file's bf16 band)
    tol = 1e-5 if tight else 2e-2
    fwd_plan = switch in ("chain_no_ln", "attn_three_launches", "bn_three_launches")     # (bn: a last-bit change of the heads' statistics re-rounds what follows)
    for name, sp in specs.items():
        a, b = g[sp.offset:sp.offset + sp.numel], ref_g[sp.offset:sp.offset + sp.numel]
        if float(b.norm()) == 0.0:
            assert float(a.norm()) == 0.0, name
            continue
        if fwd_plan:
            cos = float((a * b).sum() / (a.norm() * b.norm()))
            assert cos >= 0.95 and abs(float(a.norm() / b.norm()) - 1.0) <= 0.10, (name, cos, float(a.norm() / b.norm()))
        else:
            assert float((a - b).norm() / b.norm()) <= tol, (name, float((a - b).norm() / b.norm()))
