"""Row N1 (fine-tune training step, drop rates 0): the fp32 oracle against the fixture written from the unmodified reference
(CPU), and the device step against the oracle / fixture (GPU)."""
import os

import numpy as np
import pytest
import torch

import decode_oracle as D
import dig_oracle as O
import finetune_oracle as F

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sample_index(numel, k=8):
    if numel <= k:
        return np.arange(numel)
    return (np.arange(k, dtype=np.int64) * 2654435761 + 12345) % numel


def _fixture():
    g = np.load(os.path.join(GOLD, "finetune_tiny.npz"))
    c, ecfg = D.DecoderConfig(**D.TINY), O.DiGConfig(**O.TINY)
    P = {**D.det_encoder_state(ecfg, int(g["seed_enc"])), **D.det_decoder_state(c, int(g["seed_dec"]))}
    images = O.synthetic_batch(int(g["B"]), ecfg, int(g["batch_seed"]))[0]
    return g, c, ecfg, P, images, torch.from_numpy(g["targets"]), torch.from_numpy(g["lens"])


def test_oracle_finetune_step_matches_reference_fixture():
    g, c, ecfg, P, images, targets, lens = _fixture()
    loss, grads, logits = F.loss_and_grads(P, ecfg, c, images, targets, lens)
    assert abs(loss - float(g["loss"])) < 1e-5 * float(g["loss"])
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=3e-5)
    names = g["grad_names"].tolist()
    for i, n in enumerate(names):
        gi = grads[n]
        assert abs(gi.double().norm().item() - g["grad_norms"][i]) <= 3e-4 * g["grad_norms"][i] + 1e-7, n
        got = np.resize(gi.reshape(-1)[_sample_index(gi.numel())].numpy(), 8)
        np.testing.assert_allclose(got, g["grad_samples"][i], rtol=2e-3, atol=1e-5 * (np.abs(g["grad_samples"][i]).max() + 1e-3))
    assert "encoder.mask_token" not in names                                  # unused at fine-tune: no gradient, AdamW skips it
    groups = F.param_groups(P, ecfg.depth, float(g["layer_decay"]), float(g["weight_decay"]))
    for i, n in enumerate(names):
        assert groups[n] == (pytest.approx(float(g["group_scale"][i])), pytest.approx(float(g["group_wd"][i]))), n
    Pn = {k: v.clone() for k, v in P.items()}
    F.adamw_step(Pn, grads, {}, 1, float(g["lr"]), groups)
    for i, n in enumerate(names):
        assert abs(Pn[n].double().norm().item() - g["param_norms"][i]) <= 1e-5 * g["param_norms"][i] + 1e-7, n


def test_layer_ids_follow_reference_rule():
    assert F.layer_id("encoder.patch_embed.proj.weight", 12) == 0 and F.layer_id("encoder.mask_token", 12) == 0
    assert F.layer_id("encoder.blocks.0.attn.qkv.weight", 12) == 1 and F.layer_id("encoder.blocks.11.mlp.fc2.bias", 12) == 12
    assert F.layer_id("encoder.norm.weight", 12) == 13 and F.layer_id("decoder.layer_stack.0.norm1.weight", 12) == 13
    assert F.layer_id("linear_norm.0.weight", 12) == 13


# ------------------------------------------------------------------------------------------------ device kernels
@pytest.mark.gpu
@pytest.mark.parametrize("Lk,causal", [(25, True), (256, False), (9, True), (100, False)])
def test_seq_attention_fwd_bwd_vs_torch(Lk, causal):
    import ctypes
    from dig_amd import _lib as L
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    B, H, Lq = 5, 3, (Lk if causal else 25)
    hk = H * 64
    q = torch.randn(B * Lq, hk, device=dev).bfloat16()
    kv = torch.randn(B * Lk, 2 * hk, device=dev).bfloat16()
    lens = torch.randint(1, Lk + 1, (B,), device=dev) if causal else None
    out = torch.empty(B * Lq, hk, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B, H, Lq, device=dev)
    k, v = kv[:, :hk], kv[:, hk:]
    L.call("dig_seq_attn_fwd", L.ptr(q), hk, L.ptr(k), 2 * hk, L.ptr(v), 2 * hk, L.ptr(out), hk, L.ptr(lse), B, H, Lq, Lk, ctypes.c_float(0.125),
           int(causal), L.ptr(lens), L.stream())
    qf = q.float().requires_grad_(True); kvf = kv.float().requires_grad_(True)
    qh = qf.view(B, Lq, H, 64).permute(0, 2, 1, 3)
    kh = kvf[:, :hk].reshape(B, Lk, H, 64).permute(0, 2, 1, 3)
    vh = kvf[:, hk:].reshape(B, Lk, H, 64).permute(0, 2, 1, 3)
    logits = qh @ kh.transpose(-1, -2) * 0.125
    if causal:
        mask = (torch.arange(Lk, device=dev)[None, :] < lens[:, None])[:, None, :] & torch.tril(torch.ones(Lq, Lk, device=dev)).bool()[None]
        logits = logits.masked_fill(~mask[:, None], float("-inf"))
    ref = (logits.softmax(-1) @ vh).permute(0, 2, 1, 3).reshape(B * Lq, hk)
    assert (out.float() - ref).abs().max().item() < 3e-2
    assert (lse - torch.logsumexp(logits, -1)).abs().max().item() < 1e-3
    dout = torch.randn(B * Lq, hk, device=dev).bfloat16()
    ref.backward(dout.float())
    dq = torch.empty_like(q); dkv = torch.empty_like(kv)
    L.call("dig_seq_attn_bwd", L.ptr(q), hk, L.ptr(k), 2 * hk, L.ptr(v), 2 * hk, L.ptr(dout), hk, L.ptr(lse), L.ptr(dq), hk, L.ptr(dkv[:, :hk]), 2 * hk,
           L.ptr(dkv[:, hk:]), 2 * hk, B, H, Lq, Lk, ctypes.c_float(0.125), int(causal), L.ptr(lens), L.stream())
    rel = lambda a, b: ((a.float() - b).norm() / (b.norm() + 1e-12)).item()
    assert rel(dq, qf.grad) < 2e-2 and rel(dkv, kvf.grad) < 2e-2


@pytest.mark.gpu
def test_seq_embedding_and_cross_entropy_kernels_vs_torch():
    from dig_amd import _lib as L
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    B, T, d, V, C, Cp = 7, 25, 128, 98, 97, 104
    tok = torch.randint(0, V, (B, T), device=dev)
    emb = torch.randn(V, d, device=dev); pos = torch.randn(200, d, device=dev)
    x = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16)
    L.call("dig_seq_embed_fwd", L.ptr(tok), L.ptr(emb), L.ptr(pos), L.ptr(x), B, T, d, V, L.stream())
    ref = emb[tok] + pos[None, :T]
    assert (x.float().view(B, T, d) - ref).abs().max().item() < 3e-2
    dx = torch.randn(B * T, d, device=dev).bfloat16()
    demb = torch.randn(V, d, device=dev); d0 = demb.clone()
    L.call("dig_seq_embed_bwd", L.ptr(tok), L.ptr(dx), L.ptr(demb), B * T, d, V, L.stream())
    want = torch.zeros(V, d, device=dev).index_add_(0, tok.reshape(-1), dx.float())
    assert (demb - d0 - want).abs().max().item() < 1e-4
    logits = torch.randn(B, T, Cp, device=dev)
    tgt = torch.randint(0, C, (B, T), device=dev); lens = torch.randint(0, T + 1, (B,), device=dev)
    lf = logits[..., :C].clone().requires_grad_(True)
    mask = torch.arange(T, device=dev)[None, :] < lens[:, None]
    loss = -(torch.log_softmax(lf, -1).gather(-1, tgt[..., None])[..., 0] * mask).sum() / B
    (loss * 0.5).backward()
    dl = torch.empty(B * T, Cp, device=dev, dtype=torch.bfloat16)
    g = torch.tensor([0.5], device=dev)
    L.call("dig_seq_cross_entropy_bwd", L.ptr(logits), Cp, L.ptr(tgt), L.ptr(lens), L.ptr(g), B, T, C, L.ptr(dl), Cp, L.stream())
    assert (dl.float().view(B, T, Cp)[..., :C] - lf.grad).abs().max().item() < 2e-3 and float(dl.float().view(B, T, Cp)[..., C:].abs().max()) == 0.0
    rows = torch.empty(B * T, device=dev); out = torch.empty(1, device=dev)
    L.call("dig_seq_cross_entropy", L.ptr(logits[..., :C].contiguous()), L.ptr(tgt), L.ptr(lens), B, T, C, L.ptr(rows), L.ptr(out), L.stream())
    assert abs(out.item() - loss.item()) < 1e-4 * abs(loss.item())


def _device_model(c, ecfg, P):
    from dig_amd.finetune import RecModelTrain
    m = RecModelTrain(embed_dim=ecfg.embed_dim, depth=ecfg.depth, num_heads=ecfg.heads, n_layers=c.n_layers, d_model=c.d_model, n_head=c.n_head,
                      d_k=c.d_k, d_inner=c.d_inner, nb_classes=c.num_classes, max_len=c.max_seq_len)
    m.load_state_dict(P)
    m.to("cuda:0")
    return m.train()


@pytest.mark.gpu
def test_device_finetune_step_vs_reference_fixture():
    """Forward (teacher-forced logits), SeqCrossEntropyLoss, hand-written backward and the layer-decay AdamW step of
    dig_amd.finetune against the fixture written from the unmodified reference (bf16 kernels: the oracle under CPU bf16 autocast
    is the noise yardstick for the gradients)."""
    import types
    from dig_amd.finetune import SeqCrossEntropyLoss, LayerDecayValueAssigner, create_optimizer
    g, c, ecfg, P, images, targets, lens = _fixture()
    m = _device_model(c, ecfg, P)
    nl, ld = m.get_num_layers(), float(g["layer_decay"])
    assigner = LayerDecayValueAssigner([ld ** (nl + 1 - i) for i in range(nl + 2)])
    args = types.SimpleNamespace(opt="adamw", lr=float(g["lr"]), weight_decay=float(g["weight_decay"]), opt_eps=1e-8, opt_betas=None)
    opt = create_optimizer(args, m, get_num_layer=assigner.get_layer_id, get_layer_scale=assigner.get_scale)
    for grp in opt.param_groups:
        grp["lr"] = args.lr * grp["lr_scale"]
    opt.zero_grad()
    out = m((images.to("cuda:0"), targets, lens))
    logits = out[0]
    loss = SeqCrossEntropyLoss()(logits, targets, lens)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 2e-2 * float(g["loss"])
    ref_logits = torch.from_numpy(g["logits"])
    assert ((logits.detach().cpu() - ref_logits).norm() / ref_logits.norm()).item() < 2e-2
    _, ref_g, _ = F.loss_and_grads(P, ecfg, c, images, targets, lens)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        _, bf_g, _ = F.loss_and_grads(P, ecfg, c, images, targets, lens)
    grads = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()}
    cos = torch.nn.functional.cosine_similarity
    names, norms = g["grad_names"].tolist(), g["grad_norms"]
    tot = float(np.sqrt((norms ** 2).sum()))
    bad = []
    for i, n in enumerate(names):
        if norms[i] < 1e-3 * tot:
            continue
        r = ref_g[n].reshape(1, -1)
        c_hip, c_bf = cos(grads[n].reshape(1, -1), r).item(), cos(bf_g[n].float().reshape(1, -1), r).item()
        q_hip, q_bf = grads[n].norm().item() / norms[i], bf_g[n].float().norm().item() / norms[i]
        if (1 - c_hip) > 2 * (1 - c_bf) + 5e-3 or abs(q_hip - 1) > 2 * abs(q_bf - 1) + 3e-2:
            bad.append((n, c_hip, c_bf, q_hip, q_bf))
    assert not bad, bad
    # AdamW with layer decay: feed the reference's own gradients into the device optimizer -> parameters after the step
    for n, p in m.named_parameters():
        p.grad.copy_(ref_g[n].to("cuda:0"))
    opt.step()
    sd = m.state_dict()
    for i, n in enumerate(names):
        assert abs(sd[n].double().norm().item() - g["param_norms"][i]) <= 2e-5 * g["param_norms"][i] + 1e-6, n
    assert torch.equal(sd["encoder.mask_token"], P["encoder.mask_token"])        # no gradient: untouched, as in the reference


@pytest.mark.gpu
def test_device_finetune_engine_loop_vs_oracle():
    """dig_amd.engine_for_finetuning.train_one_epoch (schedules x lr_scale, gradient accumulation, class accuracy, lagged meters)
    over 4 micro-batches with update_freq=2, against the oracle running the same two optimizer steps."""
    import types
    from dig_amd.finetune import SeqCrossEntropyLoss, LayerDecayValueAssigner, create_optimizer
    from dig_amd.engine_for_finetuning import train_one_epoch
    from dig_amd.utils import NativeScalerWithGradNormCount
    g, c, ecfg, P, _, _, _ = _fixture()
    m = _device_model(c, ecfg, P)
    nl, ld, lr, wd = m.get_num_layers(), 0.75, 1e-3, 0.05
    assigner = LayerDecayValueAssigner([ld ** (nl + 1 - i) for i in range(nl + 2)])
    args = types.SimpleNamespace(opt="adamw", lr=lr, weight_decay=wd, opt_eps=1e-8, opt_betas=None, eval_freq=1000)
    opt = create_optimizer(args, m, get_num_layer=assigner.get_layer_id, get_layer_scale=assigner.get_scale)
    voc = D.vocabulary()
    rng = np.random.RandomState(3)
    batches = []
    for i in range(4):
        B = 5
        lens = torch.from_numpy(rng.randint(1, c.max_seq_len + 1, size=B))
        tg = torch.from_numpy(rng.randint(0, 94, size=(B, c.max_seq_len)))
        for b in range(B):
            tg[b, int(lens[b]) - 1] = 94
            tg[b, int(lens[b]):] = 95
        batches.append((O.synthetic_batch(B, ecfg, 800 + i)[0], tg, lens))
    loader = type("Ldr", (list,), {})(batches)
    loader.dataset = types.SimpleNamespace(idx_to_class={i: ch for i, ch in enumerate(voc)})
    stats = train_one_epoch(m, SeqCrossEntropyLoss(), loader, opt, torch.device("cuda:0"), 0, NativeScalerWithGradNormCount(), None, None, None,
                            None, start_steps=0, lr_schedule_values=np.array([lr, 0.5 * lr]), wd_schedule_values=np.array([wd, wd]),
                            num_training_steps_per_epoch=2, update_freq=2, args=args)
    # oracle: two optimizer steps, each on the mean gradient of two micro-batches
    Pn = {k: v.clone() for k, v in P.items()}
    state, losses = {}, []
    groups = F.param_groups(Pn, nl, ld, wd)
    for s in range(2):
        acc_g = None
        for mb in batches[2 * s:2 * s + 2]:
            l, gr, _ = F.loss_and_grads(Pn, ecfg, c, *mb)
            losses.append(l)
            acc_g = gr if acc_g is None else {k: acc_g[k] + gr[k] for k in gr}
        F.adamw_step(Pn, {k: v / 2 for k, v in acc_g.items()}, state, s + 1, [lr, 0.5 * lr][s], groups)
    assert abs(stats["loss"] - np.mean(losses)) < 3e-2 * np.mean(losses), (stats["loss"], losses)
    assert 0.0 <= stats["class_acc"] <= 1.0 and stats["lr"] == pytest.approx(0.75 * lr) and opt._step == 2   # meters average over the 4 micro-steps
    sd = m.state_dict()
    for n in ("encoder.blocks.0.mlp.fc1.weight", "decoder.layer_stack.1.enc_attn.linear_k.weight", "linear_norm.0.weight", "decoder.trg_word_emb.weight"):
        d_dev, d_ref = sd[n] - P[n], Pn[n] - P[n]
        cosv = torch.nn.functional.cosine_similarity(d_dev.reshape(1, -1), d_ref.reshape(1, -1)).item()
        assert cosv > 0.7 and abs(d_dev.norm().item() / d_ref.norm().item() - 1) < 0.2, (n, cosv)
