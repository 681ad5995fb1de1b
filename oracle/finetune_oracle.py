"""CPU oracle (test infrastructure only) for the fine-tune TRAINING step -- SURVEY.md 8(f) row N1, deterministic part
(all drop rates 0: `--drop 0 --attn_drop_rate 0 --drop_path 0`, label smoothing 0 as in README.md:92-118).

Restates in fp32 torch (autograd supplies the gradients of the restated forward):
  engine_for_finetuning.py:26-46 `train_class_batch`: outputs = model((samples, target, tgt_lens)); loss = criterion(outputs, ...)
  models/model_builder.py:124-160 RecModel.forward in train mode -> TFDecoder.forward_train (models/decoder.py:196-222): BOS-shifted
      targets, `_attention` with the pad & causal masks, classifier logits [B, T, C]
  loss/seqCrossEntropyLoss.py:47-63 (sample_normalize)
  run_class_finetuning.py:471-520 + optim_factory.py:33-100: AdamW parameter groups with layer-wise lr decay
      (`layer_decay ** (num_layers + 1 - layer_id)`; encoder blocks i -> layer i+1, patch_embed / mask_token / pos_embed -> 0,
      everything else -> num_layers + 1), no weight decay for 1-D tensors and biases
  custom_optim/_functional.py:115-140 AdamW update (dig_oracle.adamw_update).
Pinned against the unmodified reference by tests/golden/finetune_tiny.npz (oracle/ref_harness/gen_finetune_golden.py)."""
from collections import OrderedDict

import torch

import dig_oracle as O
import decode_oracle as D


def train_logits(P, ecfg, c, images, targets, lens):
    """RecModel.forward (train) up to the classifier: [B, T, num_classes]."""
    enc = D.encoder_features(P, ecfg, images)
    mem = torch.nn.functional.layer_norm(enc @ P["linear_norm.0.weight"].t() + P["linear_norm.0.bias"], (c.d_model,),
                                         P["linear_norm.1.weight"], P["linear_norm.1.bias"], 1e-5)
    B = images.shape[0]
    bos = torch.full((B, 1), c.start_idx, dtype=targets.dtype)
    query = torch.cat([bos, targets], dim=-1)[:, :-1]                          # decoder.py:213-214
    out, _ = D.decoder_attention(P, c, query, lens, mem)
    return out @ P["decoder.classifier.weight"].t() + P["decoder.classifier.bias"]


def loss_and_grads(P, ecfg, c, images, targets, lens):
    Q = OrderedDict((k, v.detach().clone().requires_grad_(k != "encoder.mask_token")) for k, v in P.items())
    logits = train_logits(Q, ecfg, c, images, targets, lens)
    loss = D.seq_cross_entropy(logits, targets, lens)
    loss.backward()
    grads = OrderedDict((k, (v.grad if v.grad is not None else torch.zeros_like(v))) for k, v in Q.items())
    return loss.item(), grads, logits.detach()


def layer_id(name, num_layers):
    """optim_factory.py:33-45,71-76: `encoder.` prefix stripped first; len(values) = num_layers + 2."""
    n = name[len("encoder."):] if name.startswith("encoder") else name
    if n in ("cls_token", "mask_token", "pos_embed") or n.startswith("patch_embed"):
        return 0
    if n.startswith("blocks"):
        return int(n.split(".")[1]) + 1
    return num_layers + 1


def param_groups(P, num_layers, layer_decay, weight_decay):
    """-> {name: (lr_scale, weight_decay)} for every parameter (run_class_finetuning.py:471-520; skip list = encoder.pos_embed /
    encoder.cls_token, which are buffers / absent here)."""
    out = OrderedDict()
    for n, p in P.items():
        lid = layer_id(n, num_layers)
        scale = layer_decay ** (num_layers + 1 - lid) if layer_decay < 1.0 else 1.0
        wd = 0.0 if (p.ndim == 1 or n.endswith(".bias")) else weight_decay
        out[n] = (scale, wd)
    return out


def adamw_step(P, grads, state, step, lr, groups):
    """One optimizer step in place; state = {name: (exp_avg, exp_avg_sq)}.  mask_token receives a zero gradient in the reference
    only if it takes part in the graph; at fine-tune it does not (mask=None), its .grad is None and AdamW skips it."""
    for n, p in P.items():
        if n == "encoder.mask_token":
            continue
        m, v = state.setdefault(n, (torch.zeros_like(p), torch.zeros_like(p)))
        scale, wd = groups[n]
        O.adamw_update(p, grads[n], m, v, step, lr * scale, wd)
