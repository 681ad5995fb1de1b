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
