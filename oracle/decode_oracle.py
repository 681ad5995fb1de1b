"""CPU oracle (test infrastructure only) for the recognition forward and greedy decode -- SURVEY.md 8(f) row N4.

Restates, in fp32 torch, the eval-mode forward of the reference's fine-tune model `RecModel` (models/model_builder.py:74-169):
  encoder  = PretrainVisionTransformerEncoder.forward_features with mask=None (modeling_pretrain_vit.py:89-106; the fine-tune
             factory `simmim_vit_small_patch4_32x128`, :123-128, is the pre-training encoder class): PatchEmbed, sinusoid
             table, 12 pre-norm blocks, final LayerNorm(eps 1e-6)
  linear_norm = Linear(384, d_embedding) + LayerNorm (model_builder.py:86-89)
  decoder  = models/decoder.py TFDecoder.forward_test (:224-252): 25 greedy steps, each re-running `_attention` (:173-194) over
             the whole (BOS-prefixed) sequence through 6 pre-norm TransformerDecoderLayers (models/transformer_layer.py:47-118:
             masked self-attention, cross-attention over the encoder memory, GELU feed-forward), final LayerNorm(eps 1e-6),
             classifier, softmax, argmax.
`greedy_decode_cached` is the same computation with a K/V cache (position t only depends on tokens <= t), which is what the
device path runs; both forms are pinned against the unmodified reference classes by tests/golden/decode_*.npz
(oracle/ref_harness/gen_decode_golden.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this."""
from collections import OrderedDict
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

import dig_oracle as O


@dataclass
class DecoderConfig:
    n_layers: int = 6
    d_model: int = 512            # = d_embedding
    n_head: int = 8
    d_k: int = 64                 # = d_v
    d_inner: int = 256
    n_position: int = 200
    num_classes: int = 97
    max_seq_len: int = 25
    enc_dim: int = 384            # encoder.num_features feeding linear_norm

    @property
    def start_idx(self):          # decoder.py:149: the extra last embedding row is <BOS>
        return self.num_classes


TINY = dict(n_layers=2, d_model=128, n_head=2, d_k=64, d_inner=64, max_seq_len=8, enc_dim=128)


def decoder_param_shapes(c: DecoderConfig) -> "OrderedDict[str, tuple]":
    """state_dict keys of `linear_norm` + `decoder` inside RecModel, in registration order (model_builder.py:80-89,
    decoder.py:152-170, transformer_layer.py:61-95,226-236,392-398)."""
    d, hk = c.d_model, c.n_head * c.d_k
    o = OrderedDict()
    o["linear_norm.0.weight"] = (d, c.enc_dim); o["linear_norm.0.bias"] = (d,)
    o["linear_norm.1.weight"] = (d,); o["linear_norm.1.bias"] = (d,)
    o["decoder.trg_word_emb.weight"] = (c.num_classes + 1, d)
    for i in range(c.n_layers):
        p = f"decoder.layer_stack.{i}."
        for n in ("norm1", "norm2", "norm3"):
            o[p + n + ".weight"] = (d,); o[p + n + ".bias"] = (d,)
        for a in ("self_attn", "enc_attn"):
            o[p + a + ".linear_q.weight"] = (hk, hk); o[p + a + ".linear_k.weight"] = (hk, hk)
            o[p + a + ".linear_v.weight"] = (hk, hk); o[p + a + ".fc.weight"] = (d, hk)
        o[p + "mlp.w_1.weight"] = (c.d_inner, d); o[p + "mlp.w_1.bias"] = (c.d_inner,)
        o[p + "mlp.w_2.weight"] = (d, c.d_inner); o[p + "mlp.w_2.bias"] = (d,)
    o["decoder.layer_norm.weight"] = (d,); o["decoder.layer_norm.bias"] = (d,)
    o["decoder.classifier.weight"] = (c.num_classes, d); o["decoder.classifier.bias"] = (c.num_classes,)
    return o


def det_decoder_state(c: DecoderConfig, seed: int):
    P = OrderedDict()
    for n, s in decoder_param_shapes(c).items():
        if n.endswith("norm.weight") or ".norm1.weight" in n or ".norm2.weight" in n or ".norm3.weight" in n or n == "linear_norm.1.weight":
            P[n] = O.det_tensor(n, s, seed, 0.1, 1.0)
        elif n.endswith(".bias"):
            P[n] = O.det_tensor(n, s, seed, 0.05)
        elif "trg_word_emb" in n:
            P[n] = O.det_tensor(n, s, seed, 0.5)
        else:
            P[n] = O.det_tensor(n, s, seed, 1.0 / np.sqrt(s[-1]))
    return P


def position_table(n_position, d_hid):
    """PositionalEncoding._get_sinusoid_encoding_table (transformer_layer.py:409-423), same dtype path (float64 denominators
    cast to float32, products and sin/cos in float32)."""
    den = torch.Tensor([1.0 / np.power(10000, 2 * (j // 2) / d_hid) for j in range(d_hid)]).view(1, -1)
    tab = torch.arange(n_position).unsqueeze(-1).float() * den
    tab[:, 0::2] = torch.sin(tab[:, 0::2])
    tab[:, 1::2] = torch.cos(tab[:, 1::2])
    return tab


def _mha(P, pre, c, q_in, kv_in, mask):
    """MultiHeadAttention.forward (transformer_layer.py:238-281), eval mode; returns (out, head-mean weights)."""
    B, Lq, _ = q_in.shape
    Lk = kv_in.shape[1]
    q = (q_in @ P[pre + "linear_q.weight"].t()).view(B, Lq, c.n_head, c.d_k).permute(0, 2, 1, 3)
    k = (kv_in @ P[pre + "linear_k.weight"].t()).view(B, Lk, c.n_head, c.d_k).permute(0, 2, 3, 1)
    v = (kv_in @ P[pre + "linear_v.weight"].t()).view(B, Lk, c.n_head, c.d_k).permute(0, 2, 1, 3)
    logits = torch.matmul(q, k) * (c.d_k ** -0.5)
    if mask is not None:
        logits = logits.masked_fill(mask.unsqueeze(1) == 0, float("-inf"))
    w = logits.softmax(dim=-1)
    out = torch.matmul(w, v).transpose(1, 2).reshape(B, Lq, c.n_head * c.d_k)
    return out @ P[pre + "fc.weight"].t(), w.mean(1)


def decoder_attention(P, c: DecoderConfig, trg_seq, tgt_lens, memory):
    """TFDecoder._attention (decoder.py:173-194): full-sequence decoder pass; returns (output [B,L,d], last layer's cross maps)."""
    B, L = trg_seq.shape
    x = P["decoder.trg_word_emb.weight"][trg_seq] + position_table(c.n_position, c.d_model)[None, :L]
    pad = torch.arange(L)[None, :] < tgt_lens[:, None]                                  # get_pad_mask (:431-441)
    sub = (1 - torch.triu(torch.ones(L, L), diagonal=1)).bool()                          # get_subsequent_mask (:444-450)
    mask = pad.unsqueeze(-2) & sub.unsqueeze(0)
    maps = None
    for i in range(c.n_layers):
        p = f"decoder.layer_stack.{i}."
        h = F.layer_norm(x, (c.d_model,), P[p + "norm1.weight"], P[p + "norm1.bias"], 1e-5)
        a, _ = _mha(P, p + "self_attn.", c, h, h, mask)
        x = x + a
        h = F.layer_norm(x, (c.d_model,), P[p + "norm2.weight"], P[p + "norm2.bias"], 1e-5)
        a, maps = _mha(P, p + "enc_attn.", c, h, memory, None)
        x = x + a
        h = F.layer_norm(x, (c.d_model,), P[p + "norm3.weight"], P[p + "norm3.bias"], 1e-5)
        x = x + F.gelu(h @ P[p + "mlp.w_1.weight"].t() + P[p + "mlp.w_1.bias"]) @ P[p + "mlp.w_2.weight"].t() + P[p + "mlp.w_2.bias"]
    return F.layer_norm(x, (c.d_model,), P["decoder.layer_norm.weight"], P["decoder.layer_norm.bias"], 1e-6), maps


def greedy_decode(P, c: DecoderConfig, memory):
    """TFDecoder.forward_test (decoder.py:224-252), literally: every step re-runs the decoder over max_seq_len+1 positions."""
    B = memory.shape[0]
    seq = torch.zeros((B, c.max_seq_len + 1), dtype=torch.long)
    seq[:, 0] = c.start_idx
    outs, maps = [], []
    for step in range(c.max_seq_len):
        o, m = decoder_attention(P, c, seq, torch.full((B,), step + 1, dtype=torch.long), memory)
        prob = F.softmax(o[:, step] @ P["decoder.classifier.weight"].t() + P["decoder.classifier.bias"], dim=-1)
        outs.append(prob); maps.append(m[:, step])
        seq[:, step + 1] = prob.argmax(-1)
    return torch.stack(outs, 1), torch.stack(maps, 1), seq[:, 1:]


def greedy_decode_cached(P, c: DecoderConfig, memory):
    """The same result with a K/V cache: step t feeds only token t; self-attention reads the cached keys/values 0..t, the
    cross-attention keys/values of the memory are projected once per layer."""
    B = memory.shape[0]
    tab = position_table(c.n_position, c.d_model)
    hk = c.n_head * c.d_k
    kc = [torch.zeros(B, c.max_seq_len, hk) for _ in range(c.n_layers)]
    vc = [torch.zeros(B, c.max_seq_len, hk) for _ in range(c.n_layers)]
    mk = [memory @ P[f"decoder.layer_stack.{i}.enc_attn.linear_k.weight"].t() for i in range(c.n_layers)]
    mv = [memory @ P[f"decoder.layer_stack.{i}.enc_attn.linear_v.weight"].t() for i in range(c.n_layers)]
    tok = torch.full((B,), c.start_idx, dtype=torch.long)
    outs, maps, toks = [], [], []

    def heads(t):
        return t.view(B, -1, c.n_head, c.d_k).permute(0, 2, 1, 3)

    for t in range(c.max_seq_len):
        x = P["decoder.trg_word_emb.weight"][tok] + tab[t]
        m = None
        for i in range(c.n_layers):
            p = f"decoder.layer_stack.{i}."
            h = F.layer_norm(x, (c.d_model,), P[p + "norm1.weight"], P[p + "norm1.bias"], 1e-5)
            kc[i][:, t] = h @ P[p + "self_attn.linear_k.weight"].t()
            vc[i][:, t] = h @ P[p + "self_attn.linear_v.weight"].t()
            q = heads((h @ P[p + "self_attn.linear_q.weight"].t())[:, None])
            w = (q @ heads(kc[i][:, :t + 1]).transpose(-1, -2) * c.d_k ** -0.5).softmax(-1)
            x = x + (w @ heads(vc[i][:, :t + 1])).transpose(1, 2).reshape(B, hk) @ P[p + "self_attn.fc.weight"].t()
            h = F.layer_norm(x, (c.d_model,), P[p + "norm2.weight"], P[p + "norm2.bias"], 1e-5)
            q = heads((h @ P[p + "enc_attn.linear_q.weight"].t())[:, None])
            w = (q @ heads(mk[i]).transpose(-1, -2) * c.d_k ** -0.5).softmax(-1)
            m = w.mean(1)[:, 0]
            x = x + (w @ heads(mv[i])).transpose(1, 2).reshape(B, hk) @ P[p + "enc_attn.fc.weight"].t()
            h = F.layer_norm(x, (c.d_model,), P[p + "norm3.weight"], P[p + "norm3.bias"], 1e-5)
            x = x + F.gelu(h @ P[p + "mlp.w_1.weight"].t() + P[p + "mlp.w_1.bias"]) @ P[p + "mlp.w_2.weight"].t() + P[p + "mlp.w_2.bias"]
        o = F.layer_norm(x, (c.d_model,), P["decoder.layer_norm.weight"], P["decoder.layer_norm.bias"], 1e-6)
        prob = F.softmax(o @ P["decoder.classifier.weight"].t() + P["decoder.classifier.bias"], dim=-1)
        tok = prob.argmax(-1)
        outs.append(prob); maps.append(m); toks.append(tok)
    return torch.stack(outs, 1), torch.stack(maps, 1), torch.stack(toks, 1)


def beam_search(P, c: DecoderConfig, memory, beam_width, eos=94):
    """TFDecoder.beam_search (decoder.py:254-370), literally.  Note what the reference does and this restatement keeps: the decoder
    input of beam slot k at step t+1 is the symbol slot k EMITTED at every earlier step (`init_target_seq[:, step + 1] =
    step_max_index`, decoder.py:307) -- the token history of a slot is never re-ordered by its predecessor, only the scores and the
    back-pointers used by the final back-tracking are.  Returns the best hypothesis per sample [B, max_seq_len] (decoder.py:369)."""
    B, N, C = memory.shape
    bw, T, nc = beam_width, c.max_seq_len, c.num_classes
    mem = memory.unsqueeze(1).repeat(1, bw, 1, 1).reshape(-1, N, C)                     # AABBCC order (decoder.py:262)
    seq = torch.zeros((B * bw, T + 1), dtype=torch.long)
    seq[:, 0] = c.start_idx
    pos_index = (torch.arange(B) * bw).view(-1, 1)
    seq_scores = torch.full((B * bw, 1), -float("inf"))
    seq_scores[torch.arange(B) * bw] = 0.0
    stored_scores, stored_pred, stored_sym = [], [], []
    for step in range(T):
        o, _ = decoder_attention(P, c, seq, torch.full((B * bw,), step + 1, dtype=torch.long), mem)
        logp = F.log_softmax(o[:, step] @ P["decoder.classifier.weight"].t() + P["decoder.classifier.bias"], dim=-1)
        cand_scores = seq_scores.repeat(1, nc) + logp
        scores, candidates = cand_scores.view(B, -1).topk(bw, dim=1)
        sym = (candidates % nc).view(B * bw)
        seq_scores = scores.view(B * bw, 1)
        pred = (candidates // nc + pos_index.expand_as(candidates)).view(B * bw, 1)
        stored_scores.append(seq_scores.clone())
        seq_scores = seq_scores.masked_fill(sym.view(-1, 1).eq(eos), -float("inf"))
        stored_pred.append(pred)
        stored_sym.append(sym)
        seq[:, step + 1] = sym
    return backtrack(torch.stack(stored_scores).squeeze(-1), torch.stack(stored_pred).squeeze(-1), torch.stack(stored_sym), B, bw, eos)


def backtrack(stored_scores, stored_pred, stored_sym, B, bw, eos):
    """The back-tracking of decoder.py:311-369 (IBM seq2seq TopKDecoder): [T, B*bw] scores / predecessors / symbols -> best
    hypothesis [B, T].  Ended hypotheses (EOS at step t) replace the worst live ones from the back, then everything is re-sorted."""
    T = stored_sym.shape[0]
    pos_index = (torch.arange(B) * bw).view(-1, 1)
    sorted_score, sorted_idx = stored_scores[-1].view(B, bw).topk(bw)
    s = sorted_score.clone()
    found = [0] * B
    t_pred = (sorted_idx + pos_index.expand_as(sorted_idx)).view(B * bw)
    p = []
    for t in range(T - 1, -1, -1):
        cur = stored_sym[t].index_select(0, t_pred)
        t_pred = stored_pred[t].index_select(0, t_pred).clone()
        eos_idx = stored_sym[t].eq(eos).nonzero()
        for i in range(eos_idx.size(0) - 1, -1, -1):
            idx = int(eos_idx[i][0])
            b = idx // bw
            res_k = bw - (found[b] % bw) - 1
            found[b] += 1
            res = b * bw + res_k
            t_pred[res] = stored_pred[t][idx]
            cur[res] = stored_sym[t][idx]
            s[b, res_k] = stored_scores[t][idx]
        p.append(cur)
    s, re_sorted = s.topk(bw)
    re_sorted = (re_sorted + pos_index.expand_as(re_sorted)).view(B * bw)
    out = torch.cat([step.index_select(0, re_sorted).view(B, bw, -1) for step in reversed(p)], -1)
    return out[:, 0, :]


# ---------------------------------------------------------------------------------------------- encoder + linear_norm
def finetune_encoder_shapes(cfg: O.DiGConfig) -> "OrderedDict[str, tuple]":
    """`encoder.*` keys of RecModel: the fine-tune factory `simmim_vit_small_patch4_32x128` is the pre-training encoder class
    (modeling_pretrain_vit.py:27-60,123-128) with its final `norm` LayerNorm active; `mask_token` stays in the state_dict."""
    o = OrderedDict(O._encoder_param_shapes(cfg, "encoder."))
    o["encoder.norm.weight"] = (cfg.embed_dim,); o["encoder.norm.bias"] = (cfg.embed_dim,)
    return o


def det_encoder_state(cfg: O.DiGConfig, seed: int):
    P = OrderedDict()
    for n, s in finetune_encoder_shapes(cfg).items():
        if "norm" in n and n.endswith("weight"):
            P[n] = O.det_tensor(n, s, seed, 0.1, 1.0)
        elif n.endswith("bias"):
            P[n] = O.det_tensor(n, s, seed, 0.05)
        else:
            P[n] = O.det_tensor(n, s, seed, 1.0 / np.sqrt(float(np.prod(s[1:]))))
    return P


def encoder_features(P, cfg: O.DiGConfig, images):
    """PretrainVisionTransformerEncoder.forward_features(x, mask=None) (modeling_pretrain_vit.py:89-106)."""
    x = O.encoder(P, "encoder.", images, None, cfg)
    return F.layer_norm(x, (cfg.embed_dim,), P["encoder.norm.weight"], P["encoder.norm.bias"], 1e-6)


def columns_1d(enc, cfg: O.DiGConfig):
    """`enc_x.view(B, *patch_shape, C).mean(1)` (model_builder.py:146-148, --use_1d_attdec): [B, gh*gw, C] -> [B, gw, C]."""
    B, N, C = enc.shape
    gh, gw = cfg.img_h // cfg.patch, cfg.img_w // cfg.patch
    return enc.view(B, gh, gw, C).mean(1)


def recognize(P, cfg: O.DiGConfig, c: DecoderConfig, images, cached=True, use_1d_attdec=False):
    """RecModel.forward in eval mode (model_builder.py:124-160): (probabilities [B,T,C], cross-attention maps, tokens)."""
    enc = encoder_features(P, cfg, images)
    if use_1d_attdec:
        enc = columns_1d(enc, cfg)
    mem = F.layer_norm(enc @ P["linear_norm.0.weight"].t() + P["linear_norm.0.bias"], (c.d_model,), P["linear_norm.1.weight"],
                       P["linear_norm.1.bias"], 1e-5)
    return (greedy_decode_cached if cached else greedy_decode)(P, c, mem)


# ---------------------------------------------------------------------------------------------- string accuracy
import string as _string


def vocabulary(voc_type="ALLCASES_SYMBOLS"):
    """dataset/dataset_image.py:60-83 `_find_classes`: the characters, then EOS, PADDING, UNKNOWN."""
    voc = {"LOWERCASE": list('0123456789abcdefghijklmnopqrstuvwxyz!"#$%&\'()*+,-./:;<=>?@[\\]^_`{|}~'),
           "ALLCASES": list(_string.digits + _string.ascii_letters), "ALLCASES_SYMBOLS": list(_string.printable[:-6])}[voc_type]
    return voc + ["EOS", "PADDING", "UNKNOWN"]


def str_list(tokens, voc):
    """evaluation_metric/metrics.py:19-62 `get_str_list` for one tensor of label sequences: cut at EOS, drop UNKNOWN, keep
    digits/letters only, lower-case."""
    eos, unk = voc.index("EOS"), voc.index("UNKNOWN")
    keep = _string.digits + _string.ascii_letters
    out = []
    for row in np.asarray(tokens):
        chars = []
        for t in row:
            if t == eos:
                break
            if t != unk:
                chars.append(voc[int(t)])
        # (_normalize_text filters the LIST of class strings, element by element, with `x in digits + letters` -- a substring test: a
        #  multi-character class such as PADDING is dropped whole, metrics.py:14-16,56-57)
        out.append("".join(c for c in chars if c in keep).lower())
    return out


def accuracy(pred_tokens, target_tokens, voc):
    """metrics.py:76-81 `Accuracy`: fraction of samples whose normalised strings are equal."""
    p, t = str_list(pred_tokens, voc), str_list(target_tokens, voc)
    return sum(a == b for a, b in zip(p, t)) / len(p)


def seq_cross_entropy(inp, target, length):
    """loss/seqCrossEntropyLoss.py:47-63 with the default sample_normalize=True.  (engine_for_finetuning.evaluate, :249, feeds it
    the soft-max PROBABILITIES forward_test returns, so in evaluation the log-softmax is taken of probabilities -- kept as is.)"""
    B, T, C = inp.shape
    mask = (torch.arange(T)[None, :] < length[:, None]).reshape(-1, 1)
    lp = F.log_softmax(inp.reshape(-1, C), dim=1)
    return torch.sum(-lp.gather(1, target.reshape(-1, 1).long()) * mask) / B


def recognition_f_measure(pred_tokens, target_tokens, voc):
    """metrics.py:83-100."""
    fs = []
    for pred, targ in zip(str_list(pred_tokens, voc), str_list(target_tokens, voc)):
        pc, tc = set(pred), set(targ)
        n = float(sum(1 for ch in pc if ch in tc))
        p, r = n / (len(pc) + 1e-5), n / (len(tc) + 1e-5)
        fs.append(2 * p * r / (p + r + 1e-5))
    return sum(fs) / len(fs)
