"""Row N1 (fine-tune training step, with and without dropout / drop-path): the fp32 oracle against the fixtures written from the
unmodified reference (CPU), and the device step against the oracle / fixtures (GPU)."""
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
@pytest.mark.parametrize("Lk,causal", [(25, True), (256, False), (9, True), (100, False)])
def test_seq_attention_fwd_bwd_vs_torch(abi_dev, Lk, causal):
    import ctypes
    from dig_amd import _lib as L
    dev = abi_dev
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


def test_seq_embedding_and_cross_entropy_kernels_vs_torch(abi_dev):
    from dig_amd import _lib as L
    dev = abi_dev
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
    # length-aware form (d = 640: three channel chunks): positions t >= lens[b] are skipped
    d2 = 640
    dx2 = torch.randn(B * T, d2, device=dev).bfloat16()
    ln = torch.tensor([25, 1, 0, 13, 7, 25, 2], device=dev)
    valid = (torch.arange(T, device=dev)[None, :] < ln[:, None]).reshape(-1)
    demb2 = torch.zeros(V, d2, device=dev)
    L.call("dig_seq_embed_bwd_lens", L.ptr(tok), L.ptr(dx2), L.ptr(demb2), B * T, d2, V, T, L.ptr(ln), L.stream())
    want2 = torch.zeros(V, d2, device=dev).index_add_(0, tok.reshape(-1)[valid], dx2.float()[valid])
    assert (demb2 - want2).abs().max().item() < 1e-4
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
    lc = logits[..., :C].contiguous()
    L.call("dig_seq_cross_entropy", L.ptr(lc), L.ptr(tgt), L.ptr(lens), B, T, C, L.ptr(rows), L.ptr(out), L.stream())
    assert abs(out.item() - loss.item()) < 1e-4 * abs(loss.item())


def _device_model(c, ecfg, P, decoder_dropout=0.0, **kw):
    kw["decoder_dropout"] = decoder_dropout
    from dig_amd.finetune import RecModelTrain
    m = RecModelTrain(embed_dim=ecfg.embed_dim, depth=ecfg.depth, num_heads=ecfg.heads, n_layers=c.n_layers, d_model=c.d_model, n_head=c.n_head,
                      d_k=c.d_k, d_inner=c.d_inner, nb_classes=c.num_classes, max_len=c.max_seq_len, **kw)
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


def _fixture_1d():
    g = np.load(os.path.join(GOLD, "finetune_tiny_1d.npz"))
    c, ecfg = D.DecoderConfig(**D.TINY), O.DiGConfig(**O.TINY)
    P = {**D.det_encoder_state(ecfg, int(g["seed_enc"])), **D.det_decoder_state(c, int(g["seed_dec"]))}
    images = O.synthetic_batch(int(g["B"]), ecfg, int(g["batch_seed"]))[0]
    return g, c, ecfg, P, images, torch.from_numpy(g["targets"]), torch.from_numpy(g["lens"])


def test_oracle_1d_decoder_matches_reference_fixture():
    """--use_1d_attdec (model_builder.py:145-148): the decoder attends over the 32 column means of the token grid; train step and
    greedy evaluation of the oracle against the fixture written from the reference modules."""
    g, c, ecfg, P, images, targets, lens = _fixture_1d()
    loss, grads, logits = F.loss_and_grads(P, ecfg, c, images, targets, lens, use_1d_attdec=True)
    assert abs(loss - float(g["loss"])) < 1e-5 * float(g["loss"])
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=3e-5)
    for i, n in enumerate(g["grad_names"].tolist()):
        assert abs(grads[n].double().norm().item() - g["grad_norms"][i]) <= 3e-4 * g["grad_norms"][i] + 1e-7, n
    probs, maps, tok = D.recognize(P, ecfg, c, images, cached=True, use_1d_attdec=True)
    np.testing.assert_allclose(probs.numpy(), g["eval_probs"], atol=3e-5)
    assert maps.shape[-1] == 32 and np.array_equal(tok.numpy(), g["eval_tokens"])


@pytest.mark.gpu
def test_device_1d_decoder_train_step_and_eval_vs_fixture():
    """The same on the device: column means by dig_window_pool_fwd / _bwd (nwin = 32), cross-attention over 32 memory tokens on the
    generic sequence-attention kernels (training) and dig_decode_cross_attn (greedy evaluation)."""
    from dig_amd.finetune import SeqCrossEntropyLoss
    g, c, ecfg, P, images, targets, lens = _fixture_1d()
    m = _device_model(c, ecfg, P, use_1d_attdec=True)
    assert m.n_mem == 32
    logits = m((images.to("cuda:0"), targets, lens))[0]
    loss = SeqCrossEntropyLoss()(logits, targets, lens)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 2e-2 * float(g["loss"])
    ref_logits = torch.from_numpy(g["logits"])
    assert ((logits.detach().cpu() - ref_logits).norm() / ref_logits.norm()).item() < 2e-2
    _, ref_g, _ = F.loss_and_grads(P, ecfg, c, images, targets, lens, use_1d_attdec=True)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        _, bf_g, _ = F.loss_and_grads(P, ecfg, c, images, targets, lens, use_1d_attdec=True)
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
    m.eval()
    probs, _, _, maps = m((images.to("cuda:0"), None, None))
    ref_p = torch.from_numpy(g["eval_probs"])
    assert maps.shape[-1] == 32 and (probs.float().cpu() - ref_p).abs().max().item() < 3e-2
    top2 = ref_p.topk(2, -1)[0]
    sure = (top2[..., 0] - top2[..., 1]) > 6e-2                                # tokens wherever the reference's margin exceeds the bf16 noise
    first_eos = torch.from_numpy(g["eval_tokens"]).eq(94).int().argmax(1)
    alive = torch.arange(ref_p.shape[1])[None, :] <= first_eos[:, None]
    same = probs.argmax(-1).cpu().eq(torch.from_numpy(g["eval_tokens"]))
    assert bool(same[sure & alive].all())


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


# ---------------------------------------------------------------------------------------------- dropout / drop-path
def _drop_fixture():
    g = np.load(os.path.join(GOLD, "finetune_tiny_drop.npz"))
    c, ecfg = D.DecoderConfig(**D.TINY), O.DiGConfig(**O.TINY)
    P = {**D.det_encoder_state(ecfg, int(g["seed_enc"])), **D.det_decoder_state(c, int(g["seed_dec"]))}
    images = O.synthetic_batch(int(g["B"]), ecfg, int(g["batch_seed"]))[0]
    r = g["rates"]
    dr = F.DropOracle(int(g["drop_seed"]), int(g["drop_step"]), drop=float(r[0]), attn_drop=float(r[1]), drop_path=float(r[2]), depth=ecfg.depth,
                      decoder_dropout=float(r[3]))
    return g, c, ecfg, P, images, torch.from_numpy(g["targets"]), torch.from_numpy(g["lens"]), dr


def test_oracle_dropout_step_matches_reference_fixture():
    """The train-mode restatement with the keyed masks against the unmodified reference run under the same masks."""
    g, c, ecfg, P, images, targets, lens, dr = _drop_fixture()
    loss, grads, logits = F.loss_and_grads(P, ecfg, c, images, targets, lens, drop=dr)
    assert abs(loss - float(g["loss"])) < 1e-5 * float(g["loss"])
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=3e-5)
    for i, n in enumerate(g["grad_names"].tolist()):
        gi = grads[n]
        assert abs(gi.double().norm().item() - g["grad_norms"][i]) <= 3e-4 * g["grad_norms"][i] + 1e-7, n
        got = np.resize(gi.reshape(-1)[_sample_index(gi.numel())].numpy(), 8)
        np.testing.assert_allclose(got, g["grad_samples"][i], rtol=2e-3, atol=1e-5 * (np.abs(g["grad_samples"][i]).max() + 1e-3))


def test_dropout_keys_and_mask_statistics():
    """Host key derivation (dig_amd.dropout) == the oracle's; the hash keeps 1-p of the elements, without visible correlation between
    neighbours or between sites."""
    from dig_amd import dropout as DR
    dr = F.DropOracle(99, 7, drop=0.1, attn_drop=0.1, drop_path=0.1, depth=12)
    plan = DR.DropPlan(99, 7)
    for site in (DR.ENC_POS, DR.enc_site(0, 0), DR.enc_site(11, 4), DR.DEC_TGT, DR.dec_site(5, 5)):
        assert DR.site_key(plan.step_seed, site) == dr.key(site)
    assert (DR.enc_site(3, 2), DR.dec_site(4, 1), DR.DEC_TGT) == (F.enc_site(3, 2), F.dec_site(4, 1), F.DEC_TGT)
    sp = plan.spec(DR.enc_site(2, 1), 0.1, DR.enc_site(2, 2), 0.05, 256)
    assert sp.thr == F.DropOracle.thr(0.1) == 429496729 and sp.pthr == F.DropOracle.thr(0.05) and sp.rows_per_sample == 256
    assert abs(sp.scale - 1 / 0.9) < 1e-6 and plan.spec(5, 0.0) is None
    assert DR.DropPlan(99, 8).step_seed != plan.step_seed                     # fresh masks every step
    k0, k1 = dr.key(DR.enc_site(1, 1))
    n = 1 << 21
    keep = F.keep_mask(k0, k1, np.arange(n, dtype=np.uint32), 0, sp.thr).astype(np.float64)
    assert abs(keep.mean() - 0.9) < 1e-3
    z = keep - keep.mean()
    for lag in (1, 2, 8, 384):
        assert abs((z[:-lag] * z[lag:]).mean() / z.var()) < 5e-3
    other = F.keep_mask(*dr.key(DR.enc_site(1, 3)), np.arange(n, dtype=np.uint32), 0, sp.thr).astype(np.float64)
    assert abs(np.corrcoef(keep, other)[0, 1]) < 5e-3


def test_dropout_kernels_reproduce_the_oracle_masks(abi_dev):
    """dig_dropout_apply, the GEMM dropout epilogue (forward with residual + drop-path, and the GELU' backward form) and the
    attention kernels with attention dropout: the keep/drop pattern is the oracle's, bit for bit."""
    import ctypes
    from dig_amd import _lib as L, ops, dropout as DR
    dev = abi_dev
    g = torch.Generator().manual_seed(4)
    dr = F.DropOracle(77, 3, drop=0.1, attn_drop=0.1, drop_path=0.25, depth=4, decoder_dropout=0.1)
    plan = DR.DropPlan(77, 3)
    # elementwise + drop-path, [rows, cols] with 8 samples of 24 rows
    rows, cols, rps = 192, 136, 24
    x = torch.randn(rows, cols, generator=g).bfloat16()
    sp = plan.spec(DR.enc_site(1, 1), 0.1, DR.enc_site(1, 2), dr.dpr[3], rps)
    y = ops.dropout_apply(x.to(dev), sp).float().cpu()
    want = dr.path(F.enc_site(1, 2), dr.elem(F.enc_site(1, 1), x.float(), 0.1).view(rows // rps, rps, cols), dr.dpr[3]).view(rows, cols)
    assert torch.equal(y == 0, want == 0) and (y - want).abs().max() <= 2e-2 * want.abs().max()
    assert 0.05 < (want == 0).float().mean() < 0.6
    # GEMM epilogue: y = resid + drop_path(dropout(x W^T + b)), both tile families
    for I, J, R, bk in ((256, 128, 64, 0), (8192, 384, 128, 244), (8192, 384, 128, 264)):
        a = torch.randn(I, R, generator=g).bfloat16(); w = torch.randn(J, R, generator=g).bfloat16()
        bias = torch.randn(J, generator=g); res = torch.randn(I, J, generator=g).bfloat16()
        sp = plan.spec(DR.enc_site(2, 3), 0.1, DR.enc_site(2, 4), 0.25, 64)
        out = ops.gemm(a.to(dev), w.to(dev), I, J, R, bias=bias.to(dev), resid=res.to(dev), bk=bk, drop=sp).float().cpu()
        lin = a.float() @ w.float().t() + bias
        want = res.float() + dr.path(F.enc_site(2, 4), dr.elem(F.enc_site(2, 3), lin, 0.1).view(I // 64, 64, J), 0.25).view(I, J)
        dropped = dr.path(F.enc_site(2, 4), dr.elem(F.enc_site(2, 3), torch.ones(I, J), 0.1).view(I // 64, 64, J), 0.25).view(I, J) == 0
        assert torch.equal(out[dropped], res.float()[dropped])                  # dropped elements: the residual passes through untouched
        assert (out - want).abs().max() <= 2e-2 * want.abs().max()
    # dropout(gelu(.)) forward + backward form (decoder feed-forward): act 1 then mask; act 2 (x gelu') with the same mask
    I, J, R = 320, 256, 128
    a = torch.randn(I, R, generator=g).bfloat16(); w = (torch.randn(J, R, generator=g) / R ** 0.5).bfloat16()
    sp = plan.spec(DR.dec_site(0, 4), 0.1)
    pre = torch.empty(I, J, device=dev, dtype=torch.bfloat16)
    u = ops.gemm(a.to(dev), w.to(dev), I, J, R, act=1, pre=pre, bk=32, drop=sp).float().cpu()
    want = dr.elem(F.dec_site(0, 4), torch.nn.functional.gelu(a.float() @ w.float().t()), 0.1)
    assert torch.equal(u == 0, want == 0) and (u - want).abs().max() <= 2e-2 * want.abs().max()
    dy = torch.randn(I, R, generator=g).bfloat16()
    du, parts = ops.linear_dgrad(dy.to(dev), w.t().contiguous().to(dev), gelu_pre=pre, colsum=True, drop=sp)
    pf = pre.float().cpu()
    dgelu = 0.5 * (1 + torch.erf(pf / 2 ** 0.5)) + pf * torch.exp(-0.5 * pf * pf) / (2 * np.pi) ** 0.5
    want = dr.elem(F.dec_site(0, 4), (dy.float() @ w.float().t()) * dgelu, 0.1)
    assert torch.equal(du.float().cpu() == 0, want == 0) and (du.float().cpu() - want).abs().max() <= 3e-2 * want.abs().max()
    assert (parts.sum(0).cpu() - want.sum(0)).abs().max() <= 2e-2 * want.sum(0).abs().max() + 0.2


def _attn_ref(q, k, v, mask, dr, site, p):
    """softmax(q k^T) -> keyed dropout -> @ v with autograd; q pre-scaled; q,k,v [B,H,L,64] fp32 leaves."""
    s = q @ k.transpose(-2, -1)
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
    return dr.attn(site, s.softmax(-1), p) @ v


def test_attention_dropout_kernels_vs_torch_with_oracle_masks(abi_dev):
    from dig_amd import _lib as L, ops, dropout as DR
    import ctypes
    dev = abi_dev
    g = torch.Generator().manual_seed(8)
    dr = F.DropOracle(5, 11, attn_drop=0.1, decoder_dropout=0.1)
    plan = DR.DropPlan(5, 11)
    cf = ctypes.c_float
    # ---- encoder / cross-attention MFMA kernels: 256 tokens, head dim 64
    B, H = 3, 2
    Dm = H * 64
    qkv = (torch.randn(B * 256, 3 * Dm, generator=g) * 0.7).bfloat16()
    qkv[:, :Dm] *= 0.125
    dctx = torch.randn(B * 256, Dm, generator=g).bfloat16()
    site = DR.enc_site(0, 0)
    sp = plan.spec(site, 0.1)
    ctx, lse = ops.attn_fwd(qkv.to(dev), B, H, Dm, drop=sp)
    dqkv = ops.attn_bwd(qkv.to(dev), ctx, dctx.to(dev), lse, B, H, Dm, 1.0, drop=sp).float().cpu()
    t = qkv.float().view(B, 256, 3, H, 64).permute(2, 0, 3, 1, 4)
    q, k, v = (t[i].clone().requires_grad_(True) for i in range(3))
    o = _attn_ref(q, k, v, None, dr, F.enc_site(0, 0), 0.1)
    want = o.transpose(1, 2).reshape(B * 256, Dm)
    assert (ctx.float().cpu() - want.detach()).abs().max() < 3e-2 * want.abs().max()
    o.backward(dctx.float().view(B, 256, H, 64).transpose(1, 2))
    wg = torch.stack([q.grad, k.grad, v.grad]).permute(1, 3, 0, 2, 4).reshape(B * 256, 3 * Dm)
    assert (dqkv - wg).abs().max() < 4e-2 * wg.abs().max()
    plain_ctx, _ = ops.attn_fwd(qkv.to(dev), B, H, Dm)
    assert (plain_ctx.float() - ctx.float()).abs().max() > 0.05 * want.abs().max()   # the mask really acts
    # q_rows: 40 existing queries per image (two 32-row blocks; rows 40..63 zero queries with zero dO), the rest never touched
    Tq = 40
    qk = qkv.clone().view(B, 256, 3 * Dm); qk[:, Tq:64, :Dm] = 0
    dc = dctx.clone().view(B, 256, Dm); dc[:, Tq:] = 0
    ctx_q, lse_q = ops.attn_fwd(qk.view(-1, 3 * Dm).to(dev), B, H, Dm, drop=sp, q_rows=Tq)
    ctx_f, lse_f = ops.attn_fwd(qk.view(-1, 3 * Dm).to(dev), B, H, Dm, drop=sp)
    assert torch.equal(ctx_q.view(B, 256, Dm)[:, :64], ctx_f.view(B, 256, Dm)[:, :64])
    dq_q = ops.attn_bwd(qk.view(-1, 3 * Dm).to(dev), ctx_f, dc.view(-1, Dm).to(dev), lse_f, B, H, Dm, 1.0, drop=sp, q_rows=Tq).view(B, 256, 3 * Dm)
    dq_f = ops.attn_bwd(qk.view(-1, 3 * Dm).to(dev), ctx_f, dc.view(-1, Dm).to(dev), lse_f, B, H, Dm, 1.0, drop=sp).view(B, 256, 3 * Dm)
    assert torch.equal(dq_q[:, :64, :Dm], dq_f[:, :64, :Dm])                          # dQ of the existing blocks
    assert (dq_q[:, :, Dm:].float() - dq_f[:, :, Dm:].float()).abs().max() <= 2e-2 * dq_f[:, :, Dm:].float().abs().max()   # dK, dV (fewer zero terms)
    # ---- decoder sequence kernels: causal + length mask (self) and plain (cross, 40 keys)
    for Lq, Lk, causal in ((8, 8, 1), (8, 40, 0)):
        B, H = 5, 2
        hk = H * 64
        qb = torch.randn(B * Lq, hk, generator=g).bfloat16(); kb = torch.randn(B * Lk, hk, generator=g).bfloat16()
        vb = torch.randn(B * Lk, hk, generator=g).bfloat16(); do = torch.randn(B * Lq, hk, generator=g).bfloat16()
        lens = torch.tensor([1, 3, 8, 5, 2]) if causal else None
        site = DR.dec_site(1, 0 if causal else 2)
        sp = plan.spec(site, 0.1)
        out = torch.empty(B * Lq, hk, device=dev, dtype=torch.bfloat16); lse = torch.empty(B, H, Lq, device=dev)
        qd, kd, vd, dod = qb.to(dev), kb.to(dev), vb.to(dev), do.to(dev)
        ld = lens.to(dev) if causal else None
        L.call("dig_seq_attn_fwd_dropout", L.ptr(qd), hk, L.ptr(kd), hk, L.ptr(vd), hk, L.ptr(out), hk, L.ptr(lse), B, H, Lq, Lk, cf(0.125),
               causal, L.ptr(ld), ctypes.byref(sp), L.stream())
        dq, dk, dv = (torch.empty_like(t_) for t_ in (qd, kd, vd))
        L.call("dig_seq_attn_bwd_dropout", L.ptr(qd), hk, L.ptr(kd), hk, L.ptr(vd), hk, L.ptr(dod), hk, L.ptr(lse), L.ptr(dq), hk, L.ptr(dk), hk,
               L.ptr(dv), hk, B, H, Lq, Lk, cf(0.125), causal, L.ptr(ld), ctypes.byref(sp), L.stream())
        q = qb.float().view(B, Lq, H, 64).transpose(1, 2).clone().requires_grad_(True)
        k = kb.float().view(B, Lk, H, 64).transpose(1, 2).clone().requires_grad_(True)
        v = vb.float().view(B, Lk, H, 64).transpose(1, 2).clone().requires_grad_(True)
        mask = None
        if causal:
            mask = ((torch.arange(Lk)[None, :] < lens[:, None])[:, None, :] & torch.tril(torch.ones(Lq, Lk)).bool()[None])[:, None]
        o = _attn_ref(q * 0.125, k, v, mask, dr, F.dec_site(1, 0 if causal else 2), 0.1)
        want = o.transpose(1, 2).reshape(B * Lq, hk)
        assert (out.float().cpu() - want.detach()).abs().max() < 3e-2 * want.abs().max()
        o.backward(do.float().view(B, Lq, H, 64).transpose(1, 2))
        for got, ref in ((dq, q.grad), (dk, k.grad), (dv, v.grad)):
            r = ref.transpose(1, 2).reshape(got.shape)
            assert (got.float().cpu() - r).abs().max() < 4e-2 * r.abs().max() + 1e-3


@pytest.mark.gpu
def test_device_finetune_step_with_dropout_vs_reference_fixture():
    """The whole training step with drop 0.1 / attn_drop 0.1 / drop_path 0.2 / decoder dropout 0.1 against the fixture produced by the
    unmodified reference under the same keyed masks; yardstick for the gradients = the oracle under CPU bf16 autocast."""
    from dig_amd.finetune import SeqCrossEntropyLoss
    g, c, ecfg, P, images, targets, lens, dr = _drop_fixture()
    r = g["rates"]
    m = _device_model(c, ecfg, P, decoder_dropout=float(r[3]), drop_rate=float(r[0]), attn_drop_rate=float(r[1]), drop_path_rate=float(r[2]),
                      drop_seed=int(g["drop_seed"]))
    m.drop_step = int(g["drop_step"])
    for p in m.parameters():
        p.grad.zero_()
    logits = m((images.to("cuda:0"), targets, lens))[0]
    assert m.drop_step == int(g["drop_step"]) + 1
    loss = SeqCrossEntropyLoss()(logits, targets, lens)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 2e-2 * float(g["loss"])
    ref_logits = torch.from_numpy(g["logits"])
    assert ((logits.detach().cpu() - ref_logits).norm() / ref_logits.norm()).item() < 2e-2
    det = F.train_logits(P, ecfg, c, images, targets, lens)                    # without the masks the logits are far away
    assert ((det - ref_logits).norm() / ref_logits.norm()).item() > 0.1
    _, ref_g, _ = F.loss_and_grads(P, ecfg, c, images, targets, lens, drop=dr)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        _, bf_g, _ = F.loss_and_grads(P, ecfg, c, images, targets, lens, drop=dr)
    grads = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()}
    cos = torch.nn.functional.cosine_similarity
    names, norms = g["grad_names"].tolist(), g["grad_norms"]
    tot = float(np.sqrt((norms ** 2).sum()))
    bad = []
    for i, n in enumerate(names):
        if norms[i] < 1e-3 * tot:
            continue
        rr = ref_g[n].reshape(1, -1)
        c_hip, c_bf = cos(grads[n].reshape(1, -1), rr).item(), cos(bf_g[n].float().reshape(1, -1), rr).item()
        q_hip, q_bf = grads[n].norm().item() / norms[i], bf_g[n].float().norm().item() / norms[i]
        if (1 - c_hip) > 2 * (1 - c_bf) + 5e-3 or abs(q_hip - 1) > 2 * abs(q_bf - 1) + 3e-2:
            bad.append((n, c_hip, c_bf, q_hip, q_bf))
    assert not bad, bad
    # a second forward draws new masks
    logits2 = m((images.to("cuda:0"), targets, lens))[0]
    assert (logits2 - logits).abs().max().item() > 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("B,lens", [(1, [8]), (3, [1, 8, 2]), (5, [1, 1, 1, 1, 1])])
def test_device_finetune_step_ragged_batches_and_length_extremes(B, lens):
    """Batch sizes that are no multiple of any tile, sequences of length 1 and of the maximum length: loss, logits and the global
    gradient norm against the fp32 oracle (rates 0)."""
    from dig_amd.finetune import SeqCrossEntropyLoss
    _, c, ecfg, P, _, _, _ = _fixture()
    images = O.synthetic_batch(B, ecfg, 4242 + B)[0]
    lens_t = torch.tensor(lens)
    rng = np.random.RandomState(B)
    targets = torch.from_numpy(rng.randint(0, 94, size=(B, c.max_seq_len)))
    for b in range(B):
        targets[b, lens[b] - 1] = 94
        targets[b, lens[b]:] = 95
    m = _device_model(c, ecfg, P)
    for p in m.parameters():
        p.grad.zero_()
    logits = m((images.to("cuda:0"), targets, lens_t))[0]
    loss = SeqCrossEntropyLoss()(logits, targets, lens_t)
    loss.backward()
    o_loss, o_grads, o_logits = F.loss_and_grads(P, ecfg, c, images, targets, lens_t)
    assert abs(loss.item() - o_loss) < 2e-2 * abs(o_loss)
    valid = (torch.arange(c.max_seq_len)[None, :] < lens_t[:, None])
    d = (logits.detach().cpu() - o_logits)[valid]
    assert (d.norm() / o_logits[valid].norm()).item() < 2e-2
    tot_ref = float(torch.sqrt(sum((g.double() ** 2).sum() for n, g in o_grads.items() if n != "encoder.mask_token")))
    tot_dev = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for _, p in m.named_parameters())))
    assert abs(tot_dev / tot_ref - 1) < 3e-2, (tot_dev, tot_ref)
    ge = m._view(m.flat_grads, "decoder.trg_word_emb.weight").float().cpu()
    assert float(ge[95].abs().max()) == 0.0                                   # the padding token never receives a gradient


@pytest.mark.gpu
def test_finetune_checkpoint_resume_is_bit_exact(tmp_path):
    """utils.save_model / auto_load_model around the fine-tune model + FineTuneAdamW (torch per-parameter optimizer layout, dropout
    key counter): two steps, save, third step == load into fresh objects and run the third step, bit for bit, dropout on."""
    import types
    from dig_amd.finetune import SeqCrossEntropyLoss, LayerDecayValueAssigner, create_optimizer
    from dig_amd.utils import NativeScalerWithGradNormCount, save_model, auto_load_model
    _, c, ecfg, P, images, targets, lens = _fixture()
    dev = "cuda:0"

    def make():
        m = _device_model(c, ecfg, P, decoder_dropout=0.1, drop_rate=0.1, attn_drop_rate=0.1, drop_path_rate=0.1, drop_seed=17)
        nl = m.get_num_layers()
        asg = LayerDecayValueAssigner([0.75 ** (nl + 1 - i) for i in range(nl + 2)])
        args = types.SimpleNamespace(opt="adamw", lr=1e-3, weight_decay=0.05, opt_eps=1e-8, opt_betas=None, output_dir=str(tmp_path), resume="",
                                     auto_resume=True, start_epoch=0)
        opt = create_optimizer(args, m, get_num_layer=asg.get_layer_id, get_layer_scale=asg.get_scale)
        for grp in opt.param_groups:
            grp["lr"] = args.lr * grp["lr_scale"]
        return m, opt, args

    def step(m, opt):
        opt.zero_grad()
        loss = SeqCrossEntropyLoss()(m((images.to(dev), targets, lens))[0], targets, lens)
        NativeScalerWithGradNormCount()(loss, opt, clip_grad=1.0, parameters=None)
        return loss.item()

    m, opt, args = make()
    step(m, opt); step(m, opt)
    save_model(args, 0, m, m, opt, NativeScalerWithGradNormCount())
    l3 = step(m, opt)
    want = m.state_dict()
    m2, opt2, args2 = make()
    for _, p in m2.named_parameters():
        p.add_(1.0)                                                            # make sure everything comes from the file
    auto_load_model(args2, m2, m2, opt2, NativeScalerWithGradNormCount())
    assert args2.start_epoch == 1 and m2.drop_step == 2 and opt2._step == 2
    ck = torch.load(os.path.join(str(tmp_path), "checkpoint-0.pth"), map_location="cpu", weights_only=False)
    assert set(ck) >= {"model", "optimizer", "epoch", "scaler", "args"} and len(ck["optimizer"]["state"]) == len(list(m.named_parameters()))
    assert sum(len(g_["params"]) for g_ in ck["optimizer"]["param_groups"]) == len(ck["optimizer"]["state"]) + 1      # + mask_token: listed, stateless
    assert [len(g["params"]) for g in ck["optimizer"]["param_groups"]] == [len(g["names"]) for g in opt.param_groups]
    l3b = step(m2, opt2)
    assert l3b == l3
    got = m2.state_dict()
    assert all(torch.equal(got[k], want[k]) for k in want), [k for k in want if not torch.equal(got[k], want[k])][:5]


# ---------------------------------------------------------------------------------------------- label smoothing (--smoothing > 0)
def _ls_cases():
    g = np.load(os.path.join(GOLD, "seq_ls_loss.npz"))
    for case in range(3):
        yield (torch.from_numpy(g[f"c{case}/logits"]), torch.from_numpy(g[f"c{case}/target"]), torch.from_numpy(g[f"c{case}/lens"]),
               float(g[f"c{case}/smoothing"]), float(g[f"c{case}/loss"]), torch.from_numpy(g[f"c{case}/grad"]))


def test_oracle_label_smoothing_loss_matches_reference_fixture():
    """Literal restatement (with the reference's [BT] x [BT,1] broadcast) against values and gradients of the reference class."""
    for x, t, l, sm, loss, grad in _ls_cases():
        xx = x.clone().requires_grad_(True)
        got = F.seq_label_smoothing_cross_entropy(xx, t, l, sm)
        got.backward()
        assert abs(got.item() - loss) <= 2e-6 * abs(loss)
        np.testing.assert_allclose(xx.grad.numpy(), grad.numpy(), rtol=1e-4, atol=1e-5 * float(grad.abs().max()))
        # the closed form the device kernels use
        B, T, C = x.shape
        lp = torch.log_softmax(x.reshape(-1, C), 1)
        mask = (torch.arange(T)[None, :] < l[:, None]).reshape(-1).float()
        nll = -lp.gather(1, t.reshape(-1, 1)).squeeze(1) * mask
        closed = ((1 - sm) * B * T * nll.sum() + sm * mask.sum() * (-lp.mean(1)).sum()) / B
        assert abs(closed.item() - loss) <= 2e-5 * abs(loss)


def test_device_label_smoothing_loss_vs_reference_fixture(abi_dev):
    from dig_amd.finetune import SeqLabelSmoothingCrossEntropyLoss
    for x, t, l, sm, loss, grad in _ls_cases():
        xd = x.clone().to(abi_dev).requires_grad_(True)
        got = SeqLabelSmoothingCrossEntropyLoss(sm)(xd, t, l)
        (got * 0.5).backward()
        assert abs(got.item() - loss) <= 2e-5 * abs(loss)
        gd = xd.grad.cpu() * 2
        assert (gd - grad).abs().max().item() <= 1e-2 * grad.abs().max().item()          # bf16 gradient rows
        assert (gd - grad).norm().item() <= 4e-3 * grad.norm().item()


# ---------------------------------------------------------------------------------------------- GRU attention head (--decoder_type attention)
import attn_decoder_oracle as AD


def _attn_fixture():
    g = np.load(os.path.join(GOLD, "attn_decoder_tiny.npz"))
    ecfg = O.DiGConfig(**O.TINY)
    c = AD.AttnDecConfig(**{**AD.TINY, "in_planes": ecfg.embed_dim})
    P = {**D.det_encoder_state(ecfg, int(g["seed_enc"])), **AD.det_state(c, int(g["seed_dec"]))}
    images = O.synthetic_batch(int(g["B"]), ecfg, int(g["batch_seed"]))[0]
    return g, c, ecfg, P, images, torch.from_numpy(g["targets"]), torch.from_numpy(g["lens"])


def test_oracle_gru_attention_head_matches_reference_fixture():
    g, c, ecfg, P, images, targets, lens = _attn_fixture()
    loss, grads, logits = AD.loss_and_grads(P, ecfg, c, images, targets, lens)
    assert abs(loss - float(g["loss"])) < 1e-5 * float(g["loss"])
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=3e-5)
    assert float(logits[:, int(lens.max()):].abs().max()) == 0.0              # steps past max(lengths) are never run: zero rows
    norms = g["grad_norms"]
    for i, n in enumerate(g["grad_names"].tolist()):
        gi = grads[n]
        assert abs(gi.double().norm().item() - norms[i]) <= 3e-4 * norms[i] + 1e-6 * norms.max(), n
        got = np.resize(gi.reshape(-1)[_sample_index(gi.numel())].numpy(), 8)
        np.testing.assert_allclose(got, g["grad_samples"][i], rtol=2e-3, atol=1e-5 * (np.abs(g["grad_samples"][i]).max() + 1e-3) + 1e-6 * norms.max())
    probs = AD.head_sample(P, c, D.encoder_features(P, ecfg, images))
    np.testing.assert_allclose(probs.numpy(), g["sample_probs"], atol=3e-5)


def test_gru_attention_kernels_vs_torch(abi_dev):
    """dig_addattn_fwd / _bwd / _bwd_tokens and dig_gru_cell_fwd / _bwd against torch autograd on the same (bf16-rounded) operands."""
    from dig_amd import _lib as L
    dev = abi_dev
    g = torch.Generator().manual_seed(12)
    B, N, A, X, T = 5, 256, 64, 128, 3
    xproj = torch.randn(B, N, A, generator=g).bfloat16(); x = torch.randn(B, N, X, generator=g).bfloat16()
    w = torch.randn(A, generator=g) * 0.3
    sproj = [torch.randn(B, A, generator=g).bfloat16() for _ in range(T)]
    dctx = [torch.randn(B, X, generator=g).bfloat16() for _ in range(T)]
    xp = xproj.float().requires_grad_(True); xx = x.float().requires_grad_(True); ww = w.clone().requires_grad_(True)
    sp = [s_.float().requires_grad_(True) for s_ in sproj]
    alphas, ctxs = [], []
    for t in range(T):
        v = torch.tanh(sp[t][:, None, :] + xp) @ ww
        al = v.softmax(1)
        c_ = torch.bmm(al[:, None, :], xx)[:, 0]
        alphas.append(al); ctxs.append(c_)
        c_.backward(dctx[t].float(), retain_graph=True)
    xd, xpd, wd = x.to(dev).view(B * N, X), xproj.to(dev).view(B * N, A), w.to(dev)
    alpha_all = torch.empty(T, B, N, device=dev); dv_all = torch.empty(T, B, N, device=dev)
    sproj_all = torch.stack(sproj).to(dev); dctx_all = torch.stack(dctx).to(dev)
    dsproj_all = torch.empty(T, B, A, device=dev, dtype=torch.bfloat16); dw_acc = torch.zeros(B, A, device=dev)
    for t in range(T):
        ctx = torch.empty(B, X, device=dev, dtype=torch.bfloat16)
        L.call("dig_addattn_fwd", L.ptr(xpd), L.ptr(sproj_all[t]), L.ptr(wd), L.ptr(xd), L.ptr(alpha_all[t]), L.ptr(ctx), X, B, N, A, X, L.stream())
        assert (alpha_all[t].cpu() - alphas[t].detach()).abs().max() < 2e-4 * alphas[t].max().item() + 1e-6
        assert (ctx.float().cpu() - ctxs[t].detach()).abs().max() < 2e-2 * ctxs[t].abs().max().item()
        L.call("dig_addattn_bwd", L.ptr(xpd), L.ptr(sproj_all[t]), L.ptr(wd), L.ptr(xd), L.ptr(alpha_all[t]), L.ptr(dctx_all[t]), X, L.ptr(dv_all[t]),
               L.ptr(dsproj_all[t]), L.ptr(dw_acc), B, N, A, X, L.stream())
        assert (dsproj_all[t].float().cpu() - sp[t].grad).abs().max() < 2e-2 * sp[t].grad.abs().max().item()
    assert (dw_acc.sum(0).cpu() - ww.grad).abs().max() < 1e-3 * ww.grad.abs().max().item()
    dxproj = torch.empty(B * N, A, device=dev, dtype=torch.bfloat16); dx = torch.empty(B * N, X, device=dev, dtype=torch.bfloat16)
    L.call("dig_addattn_bwd_tokens", L.ptr(xpd), L.ptr(sproj_all), L.ptr(wd), L.ptr(dv_all), L.ptr(alpha_all), L.ptr(dctx_all), X, L.ptr(dxproj), L.ptr(dx),
           T, B, N, A, X, L.stream())
    assert (dxproj.float().cpu().view(B, N, A) - xp.grad).abs().max() < 2e-2 * xp.grad.abs().max().item()
    assert (dx.float().cpu().view(B, N, X) - xx.grad).abs().max() < 2e-2 * xx.grad.abs().max().item()
    # GRU cell
    S = 128
    cell = torch.nn.GRUCell(7, S)
    gi = torch.randn(B, 3 * S, generator=g).bfloat16(); gh = torch.randn(B, 3 * S, generator=g).bfloat16(); s0 = torch.randn(B, S, generator=g)
    gif, ghf, s0f = gi.float().requires_grad_(True), gh.float().requires_grad_(True), s0.clone().requires_grad_(True)
    r = torch.sigmoid(gif[:, :S] + ghf[:, :S]); z = torch.sigmoid(gif[:, S:2 * S] + ghf[:, S:2 * S]); n = torch.tanh(gif[:, 2 * S:] + r * ghf[:, 2 * S:])
    s1 = (1 - z) * n + z * s0f
    ds = torch.randn(B, S, generator=g)
    s1.backward(ds)
    sd_ = torch.empty(B, S, device=dev); sbf = torch.empty(B, S, device=dev, dtype=torch.bfloat16); gates = torch.empty(B, 4 * S, device=dev)
    gid, ghd, s0d = gi.to(dev), gh.to(dev), s0.to(dev)                         # (keep the device copies alive across the launches)
    L.call("dig_gru_cell_fwd", L.ptr(gid), L.ptr(ghd), L.ptr(s0d), L.ptr(sd_), L.ptr(sbf), L.ptr(gates), B, S, L.stream())
    assert (sd_.cpu() - s1.detach()).abs().max() < 1e-5
    dgi = torch.empty(B, 3 * S, device=dev, dtype=torch.bfloat16); dgh = torch.empty_like(dgi); dsp = torch.empty(B, S, device=dev)
    half = (ds * 0.5).to(dev)
    L.call("dig_gru_cell_bwd", L.ptr(half), L.ptr(half), None, None, L.ptr(gates), L.ptr(s0d), L.ptr(dgi), L.ptr(dgh), L.ptr(dsp), B, S, L.stream())
    assert (dgi.float().cpu() - gif.grad).abs().max() < 1e-2 * gif.grad.abs().max().item()
    assert (dgh.float().cpu() - ghf.grad).abs().max() < 1e-2 * ghf.grad.abs().max().item()
    assert (dsp.cpu() - (ds * z.detach())).abs().max() < 1e-5


def _attn_device_model(c, ecfg, P):
    from dig_amd.attn_recognizer import AttnRecModelTrain
    m = AttnRecModelTrain(embed_dim=ecfg.embed_dim, depth=ecfg.depth, num_heads=ecfg.heads, nb_classes=c.num_classes, max_len=c.max_len, sDim=c.sDim,
                          attDim=c.attDim)
    m.load_state_dict(P)
    m.to("cuda:0")
    return m


@pytest.mark.gpu
def test_device_gru_attention_model_vs_reference_fixture():
    """AttnRecModel training step (teacher forcing, SeqCrossEntropyLoss, hand-written BPTT) and greedy sample against the fixture
    written from the unmodified reference; gradient yardstick = the oracle under CPU bf16 autocast."""
    from dig_amd.finetune import SeqCrossEntropyLoss
    g, c, ecfg, P, images, targets, lens = _attn_fixture()
    m = _attn_device_model(c, ecfg, P).train()
    assert list(m.state_dict().keys())[-len(AD.param_shapes(c)):] == list(AD.param_shapes(c))
    for p in m.parameters():
        p.grad.zero_()
    logits = m((images.to("cuda:0"), targets, lens))[0]
    loss = SeqCrossEntropyLoss()(logits, targets, lens)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 2e-2 * float(g["loss"])
    ref_logits = torch.from_numpy(g["logits"])
    assert ((logits.detach().cpu() - ref_logits).norm() / ref_logits.norm()).item() < 2e-2
    assert float(logits.detach()[:, int(lens.max()):].abs().max()) == 0.0
    _, ref_g, _ = AD.loss_and_grads(P, ecfg, c, images, targets, lens)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        _, bf_g, _ = AD.loss_and_grads(P, ecfg, c, images, targets, lens)
    grads = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()}
    cos = torch.nn.functional.cosine_similarity
    names, norms = g["grad_names"].tolist(), g["grad_norms"]
    tot = float(np.sqrt((norms ** 2).sum()))
    bad = []
    for i, n in enumerate(names):
        if norms[i] < 1e-3 * tot:
            continue
        rr = ref_g[n].reshape(1, -1)
        c_hip, c_bf = cos(grads[n].reshape(1, -1), rr).item(), cos(bf_g[n].float().reshape(1, -1), rr).item()
        q_hip, q_bf = grads[n].norm().item() / norms[i], bf_g[n].float().norm().item() / norms[i]
        if (1 - c_hip) > 2 * (1 - c_bf) + 5e-3 or abs(q_hip - 1) > 2 * abs(q_bf - 1) + 3e-2:
            bad.append((n, c_hip, c_bf, q_hip, q_bf))
    assert not bad, bad
    # greedy sample (eval)
    m.eval()
    probs = m((images.to("cuda:0"), None, None))[0].cpu()
    ref = torch.from_numpy(g["sample_probs"])
    srt = ref.sort(-1, descending=True)[0]
    margin = srt[..., 0] - srt[..., 1]
    same = probs.argmax(-1) == ref.argmax(-1)
    # steps are compared while the greedy paths agree (a near-tie flips a token and the later steps legitimately differ)
    for b in range(ref.shape[0]):
        for t in range(ref.shape[1]):
            if not same[b, t]:
                assert margin[b, t] < 5e-2, (b, t, float(margin[b, t]))
                break
            assert (probs[b, t] - ref[b, t]).abs().max() < 3e-2


@pytest.mark.gpu
def test_gru_attention_model_through_the_engine_loops():
    """AttnRecModel through dig_amd.engine_for_finetuning: train_one_epoch (2 micro-batches, update_freq 1) lowers nothing it should
    not -- first-step loss equals the oracle's, parameters move -- and evaluate() runs the greedy sample with the reference's meters."""
    import types
    from dig_amd.finetune import SeqCrossEntropyLoss, LayerDecayValueAssigner, create_optimizer
    from dig_amd.engine_for_finetuning import train_one_epoch, evaluate
    from dig_amd.utils import NativeScalerWithGradNormCount
    g, c, ecfg, P, images, targets, lens = _attn_fixture()
    m = _attn_device_model(c, ecfg, P)
    nl, lr, wd = m.get_num_layers(), 1e-3, 0.05
    asg = LayerDecayValueAssigner([0.75 ** (nl + 1 - i) for i in range(nl + 2)])
    args = types.SimpleNamespace(opt="adamw", lr=lr, weight_decay=wd, opt_eps=1e-8, opt_betas=None, eval_freq=1000, beam_width=0)
    opt = create_optimizer(args, m, get_num_layer=asg.get_layer_id, get_layer_scale=asg.get_scale)
    voc = D.vocabulary()
    loader = type("Ldr", (list,), {})([(images, targets, lens), (images.flip(0), targets.flip(0), lens.flip(0))])
    loader.dataset = types.SimpleNamespace(idx_to_class={i: ch for i, ch in enumerate(voc)})
    before = m.flat_params.detach().clone()
    stats = train_one_epoch(m, SeqCrossEntropyLoss(), loader, opt, torch.device("cuda:0"), 0, NativeScalerWithGradNormCount(), None, None, None,
                            None, start_steps=0, lr_schedule_values=np.array([lr, lr]), wd_schedule_values=np.array([wd, wd]),
                            num_training_steps_per_epoch=2, update_freq=1, args=args)
    assert np.isfinite(stats["loss"]) and not torch.equal(before, m.flat_params)
    o_loss = AD.loss_and_grads(P, ecfg, c, images, targets, lens)[0]
    assert stats["loss"] < 1.05 * o_loss                                       # mean of step 1 (= oracle loss) and step 2 (after one update)
    ev = evaluate(loader, m, torch.device("cuda:0"), args=args)
    assert set(ev) >= {"loss", "acc", "recognition_fmeasure"} and np.isfinite(ev["loss"]) and 0.0 <= ev["acc"] <= 100.0


# ---------------------------------------------------------------------------------------------- full-size models (README shapes)
def _bucket_check(grads, ref_g, cos_min=0.985, enc_cos_min=0.99):
    cos = torch.nn.functional.cosine_similarity
    buckets = {}
    for n, r in ref_g.items():
        if n == "encoder.mask_token" or n.endswith("wEmbed.bias"):
            continue
        key = ".".join(n.split(".")[:3]) if n.startswith(("encoder.blocks.", "decoder.layer_stack.")) else ".".join(n.split(".")[:2])
        a, b = buckets.setdefault(key, ([], []))
        a.append(grads[n].reshape(-1)); b.append(r.reshape(-1))
    bad = []
    for key, (a, b) in buckets.items():
        a, b = torch.cat(a), torch.cat(b)
        c, q = cos(a[None], b[None]).item(), (a.norm() / b.norm()).item()
        if c < (enc_cos_min if key.startswith("encoder.") else cos_min) or abs(q - 1) > 4e-2:
            bad.append((key, round(c, 4), round(q, 4)))
    return bad


def _full_batch(B, T, seed):
    rng = np.random.RandomState(seed)
    lens = torch.from_numpy(rng.randint(2, T + 1, size=B))
    tg = torch.from_numpy(rng.randint(0, 94, size=(B, T)))
    for b in range(B):
        tg[b, int(lens[b]) - 1] = 94
        tg[b, int(lens[b]):] = 95
    return tg, lens


@pytest.mark.gpu
def test_full_size_finetune_step_vs_oracle():
    """simmim_vit_small_patch4_32x128 + tf_decoder (6 layers, d 512, 8 heads) at B = 32 (8192 token rows: the tall-layer tiles of
    the encoder, the 25 x 256 cross-attention on the MFMA kernels), rates 0: loss, logits and the gradient of every block against the
    fp32 oracle on the same inputs."""
    import types
    from dig_amd.finetune import RecModelTrain, SeqCrossEntropyLoss
    ecfg = O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128")
    c = D.DecoderConfig()
    P = {**D.det_encoder_state(ecfg, 52), **D.det_decoder_state(c, 53)}
    B = 32
    images = O.synthetic_batch(B, ecfg, 5252)[0]
    tg, lens = _full_batch(B, c.max_seq_len, 7)
    args = types.SimpleNamespace(model="simmim_vit_small_patch4_32x128", decoder_name="tf_decoder", nb_classes=97, max_len=25, drop=0.0,
                                 attn_drop_rate=0.0, drop_path=0.0)
    m = RecModelTrain(args, decoder_dropout=0.0)
    m.load_state_dict(P); m.to("cuda:0"); m.train()
    for p in m.parameters():
        p.grad.zero_()
    logits = m((images.to("cuda:0"), tg, lens))[0]
    loss = SeqCrossEntropyLoss()(logits, tg, lens)
    loss.backward()
    torch.set_num_threads(max(8, min(64, os.cpu_count() or 8)))
    o_loss, o_grads, o_logits = F.loss_and_grads(P, ecfg, c, images, tg, lens)
    assert abs(loss.item() - o_loss) < 2e-2 * abs(o_loss)
    valid = torch.arange(c.max_seq_len)[None, :] < lens[:, None]
    assert ((logits.detach().cpu() - o_logits)[valid].norm() / o_logits[valid].norm()).item() < 3e-2
    grads = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()}
    bad = _bucket_check(grads, o_grads)
    assert not bad, bad


@pytest.mark.gpu
def test_full_size_b256_finetune_properties():
    """BASELINE.json configs[4] at its full per-GPU size (simmim_vit_small_patch4_32x128 + tf_decoder, 97 classes, 25 positions,
    B = 256 = 65 536 encoder token rows), README drop rates: size-independent properties of the training step -- finite loss and
    gradients for every parameter, the step is reproducible from (seed, step) (keyed dropout masks, deterministic reductions), the
    loss of a repeated batch goes down, and greedy decoding of that batch in eval mode returns [B, 25] tokens."""
    import types
    from dig_amd.finetune import RecModelTrain, SeqCrossEntropyLoss, create_optimizer
    ecfg = O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128")
    c = D.DecoderConfig()
    P = {**D.det_encoder_state(ecfg, 82), **D.det_decoder_state(c, 83)}
    B = 256
    images = O.synthetic_batch(B, ecfg, 8282)[0].to("cuda:0")
    tg, lens = _full_batch(B, c.max_seq_len, 11)
    args = types.SimpleNamespace(model="simmim_vit_small_patch4_32x128", decoder_name="tf_decoder", nb_classes=97, max_len=25, drop=0.1,
                                 attn_drop_rate=0.1, drop_path=0.1, opt="adamw", lr=1e-4, weight_decay=0.05, opt_eps=1e-8, opt_betas=None,
                                 layer_decay=0.75)
    crit = SeqCrossEntropyLoss()

    def one_model():
        m = RecModelTrain(args, decoder_dropout=0.1)
        m.load_state_dict(P); m.to("cuda:0"); m.train()
        return m

    def step_loss(m):
        for p in m.parameters():
            p.grad.zero_()
        loss = crit(m((images, tg, lens))[0], tg, lens)
        loss.backward()
        return loss

    m1, m2 = one_model(), one_model()
    l1, l2 = step_loss(m1), step_loss(m2)
    assert torch.isfinite(l1) and float(l1) == float(l2)
    for (n, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.isfinite(p.grad).all(), n
        assert torch.equal(p.grad, q.grad), n
    del m2
    opt = create_optimizer(args, m1)
    losses = []
    for _ in range(4):
        loss = step_loss(m1)
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0], losses
    m1.eval()
    with torch.no_grad():
        out = m1((images, tg, lens))
    pred = out[0] if isinstance(out, (tuple, list)) else out
    assert pred.shape[0] == B and pred.shape[1] == c.max_seq_len


@pytest.mark.gpu
def test_full_size_gru_attention_step_vs_oracle():
    """simmim_vit_small_patch4_32x128 + AttentionRecognitionHead (sDim = attDim = 512) at B = 32: loss, logits, gradients per block."""
    import types
    from dig_amd.attn_recognizer import AttnRecModelTrain
    from dig_amd.finetune import SeqCrossEntropyLoss
    ecfg = O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128")
    c = AD.AttnDecConfig()
    P = {**D.det_encoder_state(ecfg, 62), **AD.det_state(c, 63)}
    B = 32
    images = O.synthetic_batch(B, ecfg, 6262)[0]
    tg, lens = _full_batch(B, c.max_len, 8)
    args = types.SimpleNamespace(model="simmim_vit_small_patch4_32x128", nb_classes=97, max_len=25, drop=0.0, attn_drop_rate=0.0, drop_path=0.0)
    m = AttnRecModelTrain(args)
    m.load_state_dict(P); m.to("cuda:0"); m.train()
    for p in m.parameters():
        p.grad.zero_()
    logits = m((images.to("cuda:0"), tg, lens))[0]
    loss = SeqCrossEntropyLoss()(logits, tg, lens)
    loss.backward()
    torch.set_num_threads(max(8, min(64, os.cpu_count() or 8)))
    o_loss, o_grads, o_logits = AD.loss_and_grads(P, ecfg, c, images, tg, lens)
    assert abs(loss.item() - o_loss) < 2e-2 * abs(o_loss)
    valid = torch.arange(c.max_len)[None, :] < lens[:, None]
    assert ((logits.detach().cpu() - o_logits)[valid].norm() / o_logits[valid].norm()).item() < 3e-2
    grads = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()}
    bad = _bucket_check(grads, o_grads)
    assert not bad, bad


@pytest.mark.gpu
def test_full_size_finetune_step_with_readme_drop_rates_vs_oracle():
    """The README recipe's regularisers at full size (--drop 0.1 --attn_drop_rate 0.1 --drop_path 0.1, decoder dropout 0.1), B = 16:
    the device step against the fp32 oracle under the same keyed masks (12 encoder layers of 256 x 256 attention masks, drop-path
    rates from 0 to 0.1)."""
    import types
    from dig_amd.finetune import RecModelTrain, SeqCrossEntropyLoss
    ecfg = O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128")
    c = D.DecoderConfig()
    P = {**D.det_encoder_state(ecfg, 72), **D.det_decoder_state(c, 73)}
    B = 16
    images = O.synthetic_batch(B, ecfg, 7272)[0]
    tg, lens = _full_batch(B, c.max_seq_len, 9)
    args = types.SimpleNamespace(model="simmim_vit_small_patch4_32x128", decoder_name="tf_decoder", nb_classes=97, max_len=25, drop=0.1,
                                 attn_drop_rate=0.1, drop_path=0.1)
    m = RecModelTrain(args, drop_seed=4711)
    m.load_state_dict(P); m.to("cuda:0"); m.train()
    m.drop_step = 3
    for p in m.parameters():
        p.grad.zero_()
    logits = m((images.to("cuda:0"), tg, lens))[0]
    loss = SeqCrossEntropyLoss()(logits, tg, lens)
    loss.backward()
    torch.set_num_threads(max(8, min(64, os.cpu_count() or 8)))
    dr = F.DropOracle(4711, 3, drop=0.1, attn_drop=0.1, drop_path=0.1, depth=ecfg.depth, decoder_dropout=0.1)
    o_loss, o_grads, o_logits = F.loss_and_grads(P, ecfg, c, images, tg, lens, drop=dr)
    assert abs(loss.item() - o_loss) < 2e-2 * abs(o_loss)
    valid = torch.arange(c.max_seq_len)[None, :] < lens[:, None]
    assert ((logits.detach().cpu() - o_logits)[valid].norm() / o_logits[valid].norm()).item() < 3e-2
    grads = {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()}
    bad = _bucket_check(grads, o_grads, cos_min=0.98, enc_cos_min=0.985)
    assert not bad, bad


def test_finetune_optimizer_checkpoint_layout_matches_reference():
    """FineTuneAdamW.state_dict(): torch's per-parameter indices in the order of the REFERENCE optimizer (optim_factory.create_optimizer with
    the layer-decay assigner; the fixture holds which parameter sits at which index, the group sizes, lr_scale and weight_decay)."""
    import types
    from dig_amd.finetune import RecModelTrain, LayerDecayValueAssigner, create_optimizer
    g, c, ecfg, P, _, _, _ = _fixture()
    m = RecModelTrain(embed_dim=ecfg.embed_dim, depth=ecfg.depth, num_heads=ecfg.heads, n_layers=c.n_layers, d_model=c.d_model, n_head=c.n_head,
                      d_k=c.d_k, d_inner=c.d_inner, nb_classes=c.num_classes, max_len=c.max_seq_len, decoder_dropout=0.0)
    m.load_state_dict(P)
    nl, ld = m.get_num_layers(), float(g["layer_decay"])
    asg = LayerDecayValueAssigner([ld ** (nl + 1 - i) for i in range(nl + 2)])
    args = types.SimpleNamespace(opt="adamw", lr=float(g["lr"]), weight_decay=float(g["weight_decay"]), opt_eps=1e-8, opt_betas=None)
    opt = create_optimizer(args, m, get_num_layer=asg.get_layer_id, get_layer_scale=asg.get_scale)
    assert opt._ordered_names() == g["opt_index_names"].tolist()
    sd = opt.state_dict()
    assert [len(gr["params"]) for gr in sd["param_groups"]] == g["opt_group_sizes"].tolist()
    np.testing.assert_allclose([gr["lr_scale"] for gr in sd["param_groups"]], g["opt_group_lr_scale"], rtol=1e-12)
    np.testing.assert_allclose([gr["weight_decay"] for gr in sd["param_groups"]], g["opt_group_wd"], rtol=1e-12)
    # a reference-shaped state (one tensor per index) loads and round-trips
    opt._tables()
    state = {i: {"step": 4, "exp_avg": torch.full(m._offsets[n][2], float(i)), "exp_avg_sq": torch.full(m._offsets[n][2], 0.5 * i)}
             for i, n in enumerate(opt._ordered_names()) if n != "encoder.mask_token"}       # (listed, but stateless in the reference)
    opt.load_state_dict({"state": state, "param_groups": sd["param_groups"]})
    back = opt.state_dict()
    assert opt._step == 4 and set(back["state"]) == set(state) and all(torch.equal(back["state"][i]["exp_avg"], state[i]["exp_avg"]) for i in state)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 3])
def test_fixed_encoder_layers_match_oracle_with_frozen_parameters(k):
    """`--fixed_encoder_layers k` (run_class_finetuning.py:500-518): frozen parameters get no gradient, keep their values and are absent
    from the optimizer; every other gradient equals the oracle's (freezing a prefix does not change the gradients above it)."""
    import types
    from dig_amd.finetune import SeqCrossEntropyLoss, LayerDecayValueAssigner, create_optimizer
    from dig_amd.utils import NativeScalerWithGradNormCount
    g, c, ecfg, P, images, targets, lens = _fixture()
    m = _device_model(c, ecfg, P)
    frozen = m.fix_encoder_layers(k)
    want = [n for n in P if "encoder.patch_embed" in n] + ([n for n in P if n.startswith("encoder.blocks.") and int(n.split(".")[2]) < min(k, ecfg.depth + 1) - 1] if k > 1 else [])
    assert sorted(frozen) == sorted(want) and len(frozen) > 0
    nl = m.get_num_layers()
    asg = LayerDecayValueAssigner([0.75 ** (nl + 1 - i) for i in range(nl + 2)])
    args = types.SimpleNamespace(opt="adamw", lr=1e-3, weight_decay=0.05, opt_eps=1e-8, opt_betas=None)
    opt = create_optimizer(args, m, get_num_layer=asg.get_layer_id, get_layer_scale=asg.get_scale)
    assert not (set(opt._ordered_names()) & set(frozen))
    for grp in opt.param_groups:
        grp["lr"] = args.lr * grp["lr_scale"]
    opt.zero_grad()
    loss = SeqCrossEntropyLoss()(m((images.to("cuda:0"), targets, lens))[0], targets, lens)
    NativeScalerWithGradNormCount()(loss, opt, clip_grad=None, parameters=None)
    assert abs(loss.item() - float(g["loss"])) < 2e-2 * float(g["loss"])
    _, ref_g, _ = F.loss_and_grads(P, ecfg, c, images, targets, lens)
    sd = m.state_dict()
    for n, p in m.named_parameters():
        gr = p.grad.detach().float().cpu()
        if n in frozen:
            assert float(gr.abs().max()) == 0.0 and torch.equal(sd[n], P[n]), n
        elif ref_g[n].norm() > 1e-3 * max(v.norm() for v in ref_g.values()):
            cosv = torch.nn.functional.cosine_similarity(gr.reshape(1, -1), ref_g[n].reshape(1, -1)).item()
            assert cosv > 0.98, (n, cosv)
            assert not torch.equal(sd[n], P[n]), n


@pytest.mark.gpu
def test_model_ema_tracks_the_parameters():
    """`--model_ema`: after every optimizer step ema = decay * ema + (1 - decay) * model for every tensor; the averaged model evaluates."""
    import types
    from dig_amd.finetune import SeqCrossEntropyLoss, LayerDecayValueAssigner, create_optimizer, ModelEma
    from dig_amd.engine_for_finetuning import train_one_epoch
    from dig_amd.utils import NativeScalerWithGradNormCount
    g, c, ecfg, P, images, targets, lens = _fixture()
    m = _device_model(c, ecfg, P)
    ema = ModelEma(m, decay=0.9)
    nl, lr, wd = m.get_num_layers(), 1e-2, 0.05
    asg = LayerDecayValueAssigner([0.75 ** (nl + 1 - i) for i in range(nl + 2)])
    args = types.SimpleNamespace(opt="adamw", lr=lr, weight_decay=wd, opt_eps=1e-8, opt_betas=None, eval_freq=1000)
    opt = create_optimizer(args, m, get_num_layer=asg.get_layer_id, get_layer_scale=asg.get_scale)
    loader = type("Ldr", (list,), {})([(images, targets, lens)] * 3)
    loader.dataset = types.SimpleNamespace(idx_to_class={i: ch for i, ch in enumerate(D.vocabulary())})
    p0 = m.flat_params.detach().clone()
    snaps = []
    orig_update = ema.update
    ema.update = lambda mm: (orig_update(mm), snaps.append(mm.flat_params.detach().clone()))[0]
    train_one_epoch(m, SeqCrossEntropyLoss(), loader, opt, torch.device("cuda:0"), 0, NativeScalerWithGradNormCount(), None, ema, None, None,
                    start_steps=0, lr_schedule_values=np.full(3, lr), wd_schedule_values=np.full(3, wd), num_training_steps_per_epoch=3,
                    update_freq=1, args=args)
    assert len(snaps) == 3
    want = p0.clone()
    for s_ in snaps:
        want = 0.9 * want + 0.1 * s_
    torch.testing.assert_close(ema.ema.flat_params, want, rtol=1e-5, atol=1e-7)
    assert not torch.equal(ema.ema.flat_params, m.flat_params)
    probs = ema.ema((images.to("cuda:0"), None, None))[0]                      # the averaged model decodes (eval mode)
    assert probs.shape[0] == images.shape[0] and torch.isfinite(probs).all()
