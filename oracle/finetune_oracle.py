"""CPU oracle (test infrastructure only) for the fine-tune TRAINING step -- SURVEY.md 8(f) row N1 (label smoothing 0 as in
README.md:92-118), with and without the stochastic regularisers (`--drop`, `--attn_drop_rate`, `--drop_path`, the decoder's
hard-wired dropout 0.1).

Restates in fp32 torch (autograd supplies the gradients of the restated forward):
  engine_for_finetuning.py:26-46 `train_class_batch`: outputs = model((samples, target, tgt_lens)); loss = criterion(outputs, ...)
  models/model_builder.py:124-160 RecModel.forward in train mode -> TFDecoder.forward_train (models/decoder.py:196-222): BOS-shifted
      targets, `_attention` with the pad & causal masks, classifier logits [B, T, C]
  loss/seqCrossEntropyLoss.py:47-63 (sample_normalize)
  run_class_finetuning.py:471-520 + optim_factory.py:33-100: AdamW parameter groups with layer-wise lr decay
      (`layer_decay ** (num_layers + 1 - layer_id)`; encoder blocks i -> layer i+1, patch_embed / mask_token / pos_embed -> 0,
      everything else -> num_layers + 1), no weight decay for 1-D tensors and biases
  custom_optim/_functional.py:115-140 AdamW update (dig_oracle.adamw_update).
Dropout: the reference draws masks from torch's generator, which no other implementation can reproduce; the product draws them
from a keyed counter hash (include/dig_hip.h `dig_dropout_t`).  `DropOracle` restates that hash in numpy; WHERE each mask acts
(modeling_finetune.py:59,116,120,156-158; models/transformer_layer.py:271,275,399,401; models/decoder.py:180) is pinned by feeding
these very masks to the unmodified reference modules (oracle/ref_harness/gen_finetune_golden.py --drop -> finetune_tiny_drop.npz).
Pinned against the unmodified reference by tests/golden/finetune_tiny.npz (oracle/ref_harness/gen_finetune_golden.py)."""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as TF

import dig_oracle as O
import decode_oracle as D


# ---------------------------------------------------------------------------------------------- dropout masks
M64 = (1 << 64) - 1
ENC_POS, DEC_TGT = 1, 0x10000


def enc_site(layer, kind):
    """0 attn_drop, 1 proj_drop, 2 drop_path (attention branch), 3 Mlp.drop, 4 drop_path (MLP branch)."""
    return 0x100 * (layer + 1) + kind


def dec_site(layer, kind):
    """0 self attn_drop, 1 self proj_drop, 2 cross attn_drop, 3 cross proj_drop, 4 mlp dropout after act, 5 after w_2."""
    return 0x10000 + 0x100 * (layer + 1) + kind


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def keep_mask(k0, k1, a, b, thr):
    """include/dig_hip.h: uint32 hash of the coordinates (a, b) under the key (k0, k1); keep when hash >= thr."""
    a = np.asarray(a, dtype=np.uint32)
    b = np.asarray(b, dtype=np.uint32)
    with np.errstate(over="ignore"):
        x = a ^ np.uint32(k0)
        x = x ^ (x >> np.uint32(16)); x = x * np.uint32(0x7feb352d)
        x = x + np.uint32(k1) + b * np.uint32(0x9e3779b9)
        x = x ^ (x >> np.uint32(15)); x = x * np.uint32(0x846ca68b)
        x = x ^ (x >> np.uint32(16))
    return x >= np.uint32(thr)


class DropOracle:
    """Masks of one training step: `seed`, `step` as dig_amd.finetune.RecModelTrain.drop_seed / .drop_step."""

    def __init__(self, seed, step, drop=0.0, attn_drop=0.0, drop_path=0.0, depth=12, decoder_dropout=0.1):
        self.step_seed = splitmix64((int(seed) + int(step)) & M64)
        self.drop, self.attn_drop, self.decoder_dropout = drop, attn_drop, decoder_dropout
        self.dpr = [x.item() for x in torch.linspace(0, drop_path, depth)]     # modeling_pretrain_vit.py:50

    def key(self, site):
        k = splitmix64(self.step_seed ^ ((site * 0x9E3779B97F4A7C15) & M64))
        return k & 0xFFFFFFFF, k >> 32

    @staticmethod
    def thr(p):
        return min(int(float(p) * 4294967296.0), 0xFFFFFFFF)

    def elem(self, site, x, p):
        """nn.Dropout(p) on a [..., cols] tensor viewed as [rows, cols]."""
        if not p:
            return x
        k0, k1 = self.key(site)
        keep = keep_mask(k0, k1, np.arange(x.numel(), dtype=np.uint32), 0, self.thr(p))
        return x * torch.from_numpy(keep).view(x.shape).to(x.dtype) * (1.0 / (1.0 - p))

    def attn(self, site, w, p):
        """nn.Dropout(p) on attention probabilities [B, H, Lq, Lk]."""
        if not p:
            return w
        B, H, Lq, Lk = w.shape
        k0, k1 = self.key(site)
        a = (np.arange(Lq, dtype=np.uint32)[:, None] << np.uint32(16)) | np.arange(Lk, dtype=np.uint32)[None, :]
        b = np.arange(B * H, dtype=np.uint32)[:, None, None]
        keep = keep_mask(k0, k1, a[None], b, self.thr(p))
        return w * torch.from_numpy(keep).view(w.shape).to(w.dtype) * (1.0 / (1.0 - p))

    def path(self, site, x, p):
        """timm drop_path(x, p, training=True): one decision per sample."""
        if not p:
            return x
        k0, k1 = self.key(site)
        keep = keep_mask(k0, k1, np.arange(x.shape[0], dtype=np.uint32), 0, self.thr(p))
        return x * torch.from_numpy(keep).view(-1, *([1] * (x.dim() - 1))).to(x.dtype) * (1.0 / (1.0 - p))


def _encoder_train(P, cfg, images, dr):
    """PretrainVisionTransformerEncoder.forward_features (modeling_pretrain_vit.py:89-106; it has no pos_drop) with
    Block.forward (modeling_finetune.py:150-158), Attention.forward (:87-120) and Mlp.forward (:53-60) in train mode."""
    pre = "encoder."
    w = P[pre + "patch_embed.proj.weight"]
    x = O.patchify_cpp(images, cfg.patch) @ w.reshape(w.shape[0], -1).t() + P[pre + "patch_embed.proj.bias"]
    x = x + O.sinusoid_table(cfg.num_patches, cfg.embed_dim).to(x.dtype)[None]
    Bn, N, Dm = x.shape
    H, dh = cfg.heads, cfg.head_dim
    for i in range(cfg.depth):
        b = f"{pre}blocks.{i}."
        h = O.layer_norm(x, P[b + "norm1.weight"], P[b + "norm1.bias"], cfg.ln_eps)
        bias = torch.cat([P[b + "attn.q_bias"], torch.zeros_like(P[b + "attn.v_bias"]), P[b + "attn.v_bias"]])
        qkv = TF.linear(h, P[b + "attn.qkv.weight"], bias).reshape(Bn, N, 3, H, dh).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * (dh ** -0.5), qkv[1], qkv[2]
        a = dr.attn(enc_site(i, 0), (q @ k.transpose(-2, -1)).softmax(dim=-1), dr.attn_drop)
        o = TF.linear((a @ v).transpose(1, 2).reshape(Bn, N, Dm), P[b + "attn.proj.weight"], P[b + "attn.proj.bias"])
        x = x + dr.path(enc_site(i, 2), dr.elem(enc_site(i, 1), o, dr.drop), dr.dpr[i])
        h = TF.gelu(TF.linear(O.layer_norm(x, P[b + "norm2.weight"], P[b + "norm2.bias"], cfg.ln_eps), P[b + "mlp.fc1.weight"], P[b + "mlp.fc1.bias"]))
        h = dr.elem(enc_site(i, 3), TF.linear(h, P[b + "mlp.fc2.weight"], P[b + "mlp.fc2.bias"]), dr.drop)
        x = x + dr.path(enc_site(i, 4), h, dr.dpr[i])
    return TF.layer_norm(x, (cfg.embed_dim,), P["encoder.norm.weight"], P["encoder.norm.bias"], 1e-6)


def _mha_train(P, pre, c, q_in, kv_in, mask, dr, site_attn, site_proj):
    """MultiHeadAttention.forward (transformer_layer.py:238-281), train mode."""
    B, Lq, _ = q_in.shape
    Lk = kv_in.shape[1]
    q = (q_in @ P[pre + "linear_q.weight"].t()).view(B, Lq, c.n_head, c.d_k).permute(0, 2, 1, 3)
    k = (kv_in @ P[pre + "linear_k.weight"].t()).view(B, Lk, c.n_head, c.d_k).permute(0, 2, 3, 1)
    v = (kv_in @ P[pre + "linear_v.weight"].t()).view(B, Lk, c.n_head, c.d_k).permute(0, 2, 1, 3)
    logits = torch.matmul(q, k) * (c.d_k ** -0.5)
    if mask is not None:
        logits = logits.masked_fill(mask.unsqueeze(1) == 0, float("-inf"))
    w = dr.attn(site_attn, logits.softmax(dim=-1), dr.decoder_dropout)
    out = torch.matmul(w, v).transpose(1, 2).reshape(B, Lq, c.n_head * c.d_k)
    return dr.elem(site_proj, out @ P[pre + "fc.weight"].t(), dr.decoder_dropout)


def _decoder_train(P, c, trg_seq, tgt_lens, memory, dr):
    """TFDecoder._attention (decoder.py:173-194) + TransformerDecoderLayer.forward (transformer_layer.py:96-117) +
    PositionwiseFeedForward.forward (:396-403), train mode."""
    B, L = trg_seq.shape
    pd = dr.decoder_dropout
    x = dr.elem(DEC_TGT, P["decoder.trg_word_emb.weight"][trg_seq] + D.position_table(c.n_position, c.d_model)[None, :L], pd)
    pad = torch.arange(L)[None, :] < tgt_lens[:, None]
    sub = (1 - torch.triu(torch.ones(L, L), diagonal=1)).bool()
    mask = pad.unsqueeze(-2) & sub.unsqueeze(0)
    for i in range(c.n_layers):
        p = f"decoder.layer_stack.{i}."
        h = TF.layer_norm(x, (c.d_model,), P[p + "norm1.weight"], P[p + "norm1.bias"], 1e-5)
        x = x + _mha_train(P, p + "self_attn.", c, h, h, mask, dr, dec_site(i, 0), dec_site(i, 1))
        h = TF.layer_norm(x, (c.d_model,), P[p + "norm2.weight"], P[p + "norm2.bias"], 1e-5)
        x = x + _mha_train(P, p + "enc_attn.", c, h, memory, None, dr, dec_site(i, 2), dec_site(i, 3))
        h = TF.layer_norm(x, (c.d_model,), P[p + "norm3.weight"], P[p + "norm3.bias"], 1e-5)
        u = dr.elem(dec_site(i, 4), TF.gelu(h @ P[p + "mlp.w_1.weight"].t() + P[p + "mlp.w_1.bias"]), pd)
        x = x + dr.elem(dec_site(i, 5), u @ P[p + "mlp.w_2.weight"].t() + P[p + "mlp.w_2.bias"], pd)
    return TF.layer_norm(x, (c.d_model,), P["decoder.layer_norm.weight"], P["decoder.layer_norm.bias"], 1e-6)


def train_logits(P, ecfg, c, images, targets, lens, drop=None, use_1d_attdec=False):
    """RecModel.forward (train) up to the classifier: [B, T, num_classes].  drop: DropOracle (None = every rate 0).
    use_1d_attdec (model_builder.py:145-148): the decoder attends over the 32 column means of the 8 x 32 token grid."""
    enc = D.encoder_features(P, ecfg, images) if drop is None else _encoder_train(P, ecfg, images, drop)
    if use_1d_attdec:
        enc = D.columns_1d(enc, ecfg)
    mem = torch.nn.functional.layer_norm(enc @ P["linear_norm.0.weight"].t() + P["linear_norm.0.bias"], (c.d_model,),
                                         P["linear_norm.1.weight"], P["linear_norm.1.bias"], 1e-5)
    B = images.shape[0]
    bos = torch.full((B, 1), c.start_idx, dtype=targets.dtype)
    query = torch.cat([bos, targets], dim=-1)[:, :-1]                          # decoder.py:213-214
    out = D.decoder_attention(P, c, query, lens, mem)[0] if drop is None else _decoder_train(P, c, query, lens, mem, drop)
    return out @ P["decoder.classifier.weight"].t() + P["decoder.classifier.bias"]


def loss_and_grads(P, ecfg, c, images, targets, lens, drop=None, use_1d_attdec=False):
    Q = OrderedDict((k, v.detach().clone().requires_grad_(k != "encoder.mask_token")) for k, v in P.items())
    logits = train_logits(Q, ecfg, c, images, targets, lens, drop, use_1d_attdec)
    loss = D.seq_cross_entropy(logits, targets, lens)
    loss.backward()
    grads = OrderedDict((k, (v.grad if v.grad is not None else torch.zeros_like(v))) for k, v in Q.items())
    return loss.item(), grads, logits.detach()


def seq_label_smoothing_cross_entropy(logits, target, length, smoothing=0.1):
    """SeqLabelSmoothingCrossEntropyLoss.forward (loss/seqLabelSmoothingCrossEntropyLoss.py:48-70, sample_normalize), operation by
    operation -- including `-logprobs.mean(1) * mask`, a [BT] vector times the [BT, 1] mask, which broadcasts to [BT, BT]."""
    B, T = target.shape
    mask = (torch.arange(T)[None, :] < length[:, None]).reshape(-1, 1)
    logprobs = TF.log_softmax(logits.reshape(-1, logits.shape[2]), dim=1)
    nll_loss = -logprobs.gather(1, target.reshape(-1, 1).long()) * mask
    smooth_loss = -logprobs.mean(1) * mask
    loss = (1.0 - smoothing) * nll_loss + smoothing * smooth_loss
    return torch.sum(loss) / B


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
