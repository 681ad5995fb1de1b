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
