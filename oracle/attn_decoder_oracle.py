"""CPU oracle (test infrastructure only) for the GRU attention recognition head -- SURVEY.md 8(f) row N1, second decoder:
`AttnRecModel` (models/model_builder.py:40-72, `--decoder_type attention`, run_class_finetuning.py:352-353) = the fine-tune encoder
followed by `AttentionRecognitionHead` (models/attn_decoder.py:11-78: forward_train under teacher forcing, greedy `sample`) whose step
is `DecoderUnit` (:236-272: additive attention `AttentionUnit` :197-233 over the encoder tokens, target embedding, one nn.GRU cell,
classifier).  Restated in fp32 torch with the GRU cell written out (gate order r, z, n; torch.nn.GRU); autograd supplies gradients.
Pinned against the unmodified reference classes by tests/golden/attn_decoder_tiny.npz (oracle/ref_harness/gen_attn_decoder_golden.py)."""
from collections import OrderedDict
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

import dig_oracle as O
import decode_oracle as D


@dataclass
class AttnDecConfig:
    num_classes: int = 97
    in_planes: int = 384
    sDim: int = 512
    attDim: int = 512
    max_len: int = 25


TINY = dict(num_classes=97, in_planes=128, sDim=128, attDim=64, max_len=8)
PRE = "decoder.decoder."


def param_shapes(c: AttnDecConfig):
    """state_dict keys of AttentionRecognitionHead inside AttnRecModel, registration order (attn_decoder.py:198-206,244-247)."""
    o = OrderedDict()
    o[PRE + "attention_unit.sEmbed.weight"] = (c.attDim, c.sDim); o[PRE + "attention_unit.sEmbed.bias"] = (c.attDim,)
    o[PRE + "attention_unit.xEmbed.weight"] = (c.attDim, c.in_planes); o[PRE + "attention_unit.xEmbed.bias"] = (c.attDim,)
    o[PRE + "attention_unit.wEmbed.weight"] = (1, c.attDim); o[PRE + "attention_unit.wEmbed.bias"] = (1,)
    o[PRE + "tgt_embedding.weight"] = (c.num_classes + 1, c.attDim)
    o[PRE + "gru.weight_ih_l0"] = (3 * c.sDim, c.in_planes + c.attDim); o[PRE + "gru.weight_hh_l0"] = (3 * c.sDim, c.sDim)
    o[PRE + "gru.bias_ih_l0"] = (3 * c.sDim,); o[PRE + "gru.bias_hh_l0"] = (3 * c.sDim,)
    o[PRE + "fc.weight"] = (c.num_classes, c.sDim); o[PRE + "fc.bias"] = (c.num_classes,)
    return o


def det_state(c: AttnDecConfig, seed: int):
    P = OrderedDict()
    for n, s in param_shapes(c).items():
        if n.endswith("bias") or n.endswith("bias_ih_l0") or n.endswith("bias_hh_l0"):
            P[n] = O.det_tensor(n, s, seed, 0.05)
        elif "tgt_embedding" in n:
            P[n] = O.det_tensor(n, s, seed, 0.5)
        else:
            P[n] = O.det_tensor(n, s, seed, 1.0 / np.sqrt(s[-1]))
    return P


def decoder_step(P, c, x, xproj, s_prev, y_prev):
    """DecoderUnit.forward (attn_decoder.py:258-272) with AttentionUnit.forward (:215-233); x [B,T,xDim], s_prev [B,sDim]."""
    sproj = s_prev @ P[PRE + "attention_unit.sEmbed.weight"].t() + P[PRE + "attention_unit.sEmbed.bias"]
    v = torch.tanh(sproj.unsqueeze(1) + xproj) @ P[PRE + "attention_unit.wEmbed.weight"].t() + P[PRE + "attention_unit.wEmbed.bias"]
    alpha = F.softmax(v.squeeze(-1), dim=1)
    context = torch.bmm(alpha.unsqueeze(1), x).squeeze(1)
    yproj = P[PRE + "tgt_embedding.weight"][y_prev.long()]
    inp = torch.cat([yproj, context], 1)
    gi = inp @ P[PRE + "gru.weight_ih_l0"].t() + P[PRE + "gru.bias_ih_l0"]
    gh = s_prev @ P[PRE + "gru.weight_hh_l0"].t() + P[PRE + "gru.bias_hh_l0"]
    S = c.sDim
    r = torch.sigmoid(gi[:, :S] + gh[:, :S])
    z = torch.sigmoid(gi[:, S:2 * S] + gh[:, S:2 * S])
    n = torch.tanh(gi[:, 2 * S:] + r * gh[:, 2 * S:])
    s = (1 - z) * n + z * s_prev
    return s @ P[PRE + "fc.weight"].t() + P[PRE + "fc.bias"], s, alpha


def head_forward_train(P, c, x, targets, lengths):
    """AttentionRecognitionHead.forward_train (:36-56): max(lengths) teacher-forced steps, outputs zero-padded to max_len."""
    B = x.shape[0]
    xproj = x @ P[PRE + "attention_unit.xEmbed.weight"].t() + P[PRE + "attention_unit.xEmbed.bias"]
    s = torch.zeros(B, c.sDim)
    outs = []
    steps = int(lengths.max())
    for i in range(steps):
        y_prev = torch.full((B,), c.num_classes, dtype=torch.long) if i == 0 else targets[:, i - 1]
        o, s, _ = decoder_step(P, c, x, xproj, s, y_prev)
        outs.append(o)
    out = torch.zeros(B, c.max_len, c.num_classes)
    out[:, :steps] = torch.stack(outs, 1)
    return out


def head_sample(P, c, x):
    """AttentionRecognitionHead.sample (:58-78): greedy, max_len steps, returns the per-step softmax [B, max_len, C]."""
    B = x.shape[0]
    xproj = x @ P[PRE + "attention_unit.xEmbed.weight"].t() + P[PRE + "attention_unit.xEmbed.bias"]
    s = torch.zeros(B, c.sDim)
    y_prev = torch.full((B,), c.num_classes, dtype=torch.long)
    outs = []
    for _ in range(c.max_len):
        o, s, _ = decoder_step(P, c, x, xproj, s, y_prev)
        prob = F.softmax(o, dim=1)
        y_prev = prob.max(1)[1]
        outs.append(prob)
    return torch.stack(outs, 1)


def train_logits(P, ecfg, c, images, targets, lengths):
    """AttnRecModel.forward in train mode (model_builder.py:67-72): encoder tokens [B, 256, D] straight into the head."""
    return head_forward_train(P, c, D.encoder_features(P, ecfg, images), targets, lengths)


def loss_and_grads(P, ecfg, c, images, targets, lengths):
    Q = OrderedDict((k, v.detach().clone().requires_grad_(k != "encoder.mask_token")) for k, v in P.items())
    logits = train_logits(Q, ecfg, c, images, targets, lengths)
    loss = D.seq_cross_entropy(logits, targets, lengths)
    loss.backward()
    grads = OrderedDict((k, (v.grad if v.grad is not None else torch.zeros_like(v))) for k, v in Q.items())
    return loss.item(), grads, logits.detach()
