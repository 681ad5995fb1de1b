"""Recognition forward + greedy decode on the MI355X -- the eval path of the reference's fine-tune model
(`RecModel.forward` with `self.training == False`, models/model_builder.py:124-160; `TFDecoder.forward_test`,
models/decoder.py:224-252; encoder factory `simmim_vit_small_patch4_32x128`, modeling_pretrain_vit.py:123-128).

    model = RecModel(args)                      # args.model / decoder_name / nb_classes / max_len as in run_class_finetuning.py
    model.load_state_dict(checkpoint["model"])  # the reference's fine-tune state_dict (encoder.*, linear_norm.*, decoder.*)
    probs, _, _, attn_maps = model((images, None, None))     # probs [B, max_len, nb_classes], attn_maps [B, max_len, 256]

What differs from the reference: the decoder keeps a K/V cache (one token per step instead of re-running all 26 positions 25
times -- same result, position t only depends on tokens <= t) and the cross-attention keys/values of the encoder memory are
projected once per layer.  Inference only (the fine-tune *training* step is row N1, not built yet): there is no CPU fallback
and no autograd through this module."""
import ctypes
from collections import OrderedDict

import numpy as np
import types

import torch

from . import _lib as L
from . import ops

BF16, F32 = torch.bfloat16, torch.float32
cf = ctypes.c_float

ENCODERS = {"simmim_vit_tiny_patch4_32x128": (192, 3), "simmim_vit_small_patch4_32x128": (384, 6), "simmim_vit_base_patch4_32x128": (512, 8)}
DECODERS = {"tf_decoder": dict(n_layers=6, d_model=512, n_head=8, d_k=64, d_inner=256),
            "small_tf_decoder": dict(n_layers=2, d_model=384, n_head=6, d_k=64, d_inner=192)}


def _sinusoid(n_position, d_hid):
    """PositionalEncoding._get_sinusoid_encoding_table (models/transformer_layer.py:409-423)."""
    den = torch.Tensor([1.0 / np.power(10000, 2 * (j // 2) / d_hid) for j in range(d_hid)]).view(1, -1)
    tab = torch.arange(n_position).unsqueeze(-1).float() * den
    tab[:, 0::2] = torch.sin(tab[:, 0::2])
    tab[:, 1::2] = torch.cos(tab[:, 1::2])
    return tab


def _encoder_pos(n_pos, d):
    """get_sinusoid_encoding_table (modeling_finetune.py / modeling_pretrain_vit.py: the encoder's fixed position table)."""
    pos = np.arange(n_pos)[:, None] / np.power(10000, 2 * (np.arange(d)[None, :] // 2) / d)
    pos[:, 0::2] = np.sin(pos[:, 0::2])
    pos[:, 1::2] = np.cos(pos[:, 1::2])
    return torch.FloatTensor(pos)


class RecModel(torch.nn.Module):
    def __init__(self, args=None, *, embed_dim=None, depth=12, num_heads=None, n_layers=None, d_model=None, n_head=None, d_k=64,
                 d_inner=None, nb_classes=97, max_len=25, n_position=200, use_1d_attdec=False):
        super().__init__()
        if args is not None:
            embed_dim, num_heads = ENCODERS[args.model]
            dk = DECODERS[args.decoder_name]
            n_layers, d_model, n_head, d_k, d_inner = dk["n_layers"], dk["d_model"], dk["n_head"], dk["d_k"], dk["d_inner"]
            nb_classes, max_len = args.nb_classes, args.max_len
            use_1d_attdec = bool(getattr(args, "use_1d_attdec", False))
            if getattr(args, "text_cond_vis", False) or getattr(args, "insert_sem", False):
                raise NotImplementedError("text-conditional attention / semantic insertion are not built (tf_decoder on 2-D or 1-D features, "
                                          "greedy or beam search)")
        if d_k != 64 or embed_dim // num_heads != 64:
            raise NotImplementedError("head dimension 64 only")
        self.D, self.H, self.depth, self.F = embed_dim, num_heads, depth, 4 * embed_dim
        self.gh, self.gw, self.N = 8, 32, 256
        # --use_1d_attdec (run_class_finetuning.py:89, model_builder.py:145-148): the decoder attends over the gw column means of the
        # token grid instead of all gh*gw tokens
        self.use_1d_attdec = bool(use_1d_attdec)
        self.n_mem = self.gw if self.use_1d_attdec else self.N
        self.n_layers, self.d, self.nh, self.dk, self.d_inner = n_layers, d_model, n_head, d_k, d_inner
        self.nb_classes, self.max_len, self.n_position = nb_classes, max_len, n_position
        self.start_idx = nb_classes                                         # decoder.py:149
        self._sd = OrderedDict()
        self._ready = False
        self._graphs = {}
        self.use_hip_graph = True
        self.beam_width = int(getattr(args, "beam_width", 0) or 0) if args is not None else 0      # model_builder.py:110
        self.eos = 94                                                       # TFDecoder.beam_search's default (decoder.py:254)

    # ------------------------------------------------------------------ state
    def param_shapes(self):
        D, F, d, hk = self.D, self.F, self.d, self.nh * self.dk
        o = OrderedDict()
        e = "encoder."
        o[e + "mask_token"] = (1, 1, D)
        o[e + "patch_embed.proj.weight"] = (D, 3, 4, 4); o[e + "patch_embed.proj.bias"] = (D,)
        for i in range(self.depth):
            b = f"{e}blocks.{i}."
            o[b + "norm1.weight"] = (D,); o[b + "norm1.bias"] = (D,)
            o[b + "attn.q_bias"] = (D,); o[b + "attn.v_bias"] = (D,)
            o[b + "attn.qkv.weight"] = (3 * D, D); o[b + "attn.proj.weight"] = (D, D); o[b + "attn.proj.bias"] = (D,)
            o[b + "norm2.weight"] = (D,); o[b + "norm2.bias"] = (D,)
            o[b + "mlp.fc1.weight"] = (F, D); o[b + "mlp.fc1.bias"] = (F,); o[b + "mlp.fc2.weight"] = (D, F); o[b + "mlp.fc2.bias"] = (D,)
        o[e + "norm.weight"] = (D,); o[e + "norm.bias"] = (D,)
        o["decoder.trg_word_emb.weight"] = (self.nb_classes + 1, d)
        for i in range(self.n_layers):
            p = f"decoder.layer_stack.{i}."
            for n in ("norm1", "norm2", "norm3"):
                o[p + n + ".weight"] = (d,); o[p + n + ".bias"] = (d,)
            for a in ("self_attn", "enc_attn"):
                for w in ("linear_q", "linear_k", "linear_v"):
                    o[p + a + "." + w + ".weight"] = (hk, hk)
                o[p + a + ".fc.weight"] = (d, hk)
            o[p + "mlp.w_1.weight"] = (self.d_inner, d); o[p + "mlp.w_1.bias"] = (self.d_inner,)
            o[p + "mlp.w_2.weight"] = (d, self.d_inner); o[p + "mlp.w_2.bias"] = (d,)
        o["decoder.layer_norm.weight"] = (d,); o["decoder.layer_norm.bias"] = (d,)
        o["decoder.classifier.weight"] = (self.nb_classes, d); o["decoder.classifier.bias"] = (self.nb_classes,)
        o["linear_norm.0.weight"] = (d, D); o["linear_norm.0.bias"] = (d,)
        o["linear_norm.1.weight"] = (d,); o["linear_norm.1.bias"] = (d,)
        return o

    def load_state_dict(self, state_dict, strict=True):
        """Accepts the reference RecModel's state_dict: buffers (`decoder.position_enc.position_table`) and the aliases RecModel
        registers (`patch_embed.*` = `encoder.patch_embed.*`, model_builder.py:92-93) are ignored."""
        shapes = self.param_shapes()
        missing = [k for k in shapes if k not in state_dict]
        if missing and strict:
            raise KeyError(f"missing keys in state_dict: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        extra = [k for k in state_dict if k not in shapes and not k.endswith("position_table") and not k.startswith("patch_embed.")]
        if extra and strict:
            raise KeyError(f"unexpected keys in state_dict: {extra[:5]}")
        for k, s in shapes.items():
            if k in state_dict:
                if tuple(state_dict[k].shape) != tuple(s):
                    raise ValueError(f"{k}: shape {tuple(state_dict[k].shape)} != {s}")
                self._sd[k] = state_dict[k].detach().to(F32).clone()
        self._ready = False
        self._graphs = {}

    def state_dict(self, *a, **k):
        return OrderedDict((k_, v.cpu()) for k_, v in self._sd.items())

    def _prepare(self, dev):
        """bf16 GEMM operands (fused q|k|v and k|v weights, classifier padded to a multiple of 8 rows) + fp32 vectors."""
        sd = {k: v.to(dev) for k, v in self._sd.items()}
        D, d = self.D, self.d
        w = {}
        for i in range(self.depth):
            b = f"encoder.blocks.{i}."
            w[b] = dict(n1w=sd[b + "norm1.weight"], n1b=sd[b + "norm1.bias"], n2w=sd[b + "norm2.weight"], n2b=sd[b + "norm2.bias"],
                        qkv=sd[b + "attn.qkv.weight"].to(BF16).contiguous(), proj=sd[b + "attn.proj.weight"].to(BF16).contiguous(),
                        qkv_bias=torch.cat([sd[b + "attn.q_bias"], torch.zeros(D, device=dev), sd[b + "attn.v_bias"]]).contiguous(),
                        proj_b=sd[b + "attn.proj.bias"], fc1=sd[b + "mlp.fc1.weight"].to(BF16).contiguous(), fc1_b=sd[b + "mlp.fc1.bias"],
                        fc2=sd[b + "mlp.fc2.weight"].to(BF16).contiguous(), fc2_b=sd[b + "mlp.fc2.bias"])
        w["pe_w"] = sd["encoder.patch_embed.proj.weight"].reshape(D, 48).contiguous()
        w["pe_b"] = sd["encoder.patch_embed.proj.bias"]
        w["mask_token"] = sd["encoder.mask_token"].reshape(D).contiguous()
        w["enc_pos"] = _encoder_pos(self.N, D).to(dev).contiguous()
        w["enc_nw"], w["enc_nb"] = sd["encoder.norm.weight"], sd["encoder.norm.bias"]
        w["ln_w"] = sd["linear_norm.0.weight"].to(BF16).contiguous(); w["ln_b"] = sd["linear_norm.0.bias"]
        w["ln_nw"], w["ln_nb"] = sd["linear_norm.1.weight"], sd["linear_norm.1.bias"]
        w["emb"] = sd["decoder.trg_word_emb.weight"].contiguous()
        w["pos"] = _sinusoid(self.n_position, d).to(dev).contiguous()
        for i in range(self.n_layers):
            p = f"decoder.layer_stack.{i}."
            cat = lambda a, names: torch.cat([sd[p + a + "." + n + ".weight"] for n in names]).to(BF16).contiguous()
            w[p] = dict(n1w=sd[p + "norm1.weight"], n1b=sd[p + "norm1.bias"], n2w=sd[p + "norm2.weight"], n2b=sd[p + "norm2.bias"],
                        n3w=sd[p + "norm3.weight"], n3b=sd[p + "norm3.bias"], qkv=cat("self_attn", ("linear_q", "linear_k", "linear_v")),
                        fc=sd[p + "self_attn.fc.weight"].to(BF16).contiguous(), q2=sd[p + "enc_attn.linear_q.weight"].to(BF16).contiguous(),
                        kv2=cat("enc_attn", ("linear_k", "linear_v")), fc2=sd[p + "enc_attn.fc.weight"].to(BF16).contiguous(),
                        w1=sd[p + "mlp.w_1.weight"].to(BF16).contiguous(), b1=sd[p + "mlp.w_1.bias"],
                        w2=sd[p + "mlp.w_2.weight"].to(BF16).contiguous(), b2=sd[p + "mlp.w_2.bias"])
        w["fnw"], w["fnb"] = sd["decoder.layer_norm.weight"], sd["decoder.layer_norm.bias"]
        C, Cp = self.nb_classes, (self.nb_classes + 7) // 8 * 8
        cw = torch.zeros((Cp, d), device=dev); cw[:C] = sd["decoder.classifier.weight"]
        cb = torch.zeros(Cp, device=dev); cb[:C] = sd["decoder.classifier.bias"]
        w["cls_w"], w["cls_b"], w["Cp"] = cw.to(BF16).contiguous(), cb, Cp
        self._w, self._dev, self._ready = w, dev, True
        self._graphs = {}

    # ------------------------------------------------------------------ forward pieces
    def encoder_features(self, images):
        """PretrainVisionTransformerEncoder.forward_features(x, mask=None): bf16 [B*256, D]."""
        w, D, H = self._w, self.D, self.H
        B = images.shape[0]
        zeros = torch.zeros((B, self.N), device=images.device, dtype=torch.uint8)
        x = ops.patch_embed_fwd(images.contiguous().float(), w["pe_w"], w["pe_b"], zeros, w["mask_token"], w["enc_pos"], D, self.gh, self.gw)
        scale = (D // H) ** -0.5
        for i in range(self.depth):
            b = w[f"encoder.blocks.{i}."]
            ln1, _, _ = ops.layernorm_fwd(x, b["n1w"], b["n1b"], 1e-6)
            qkv = ops.linear_fwd(ln1, b["qkv"], bias=b["qkv_bias"], alpha=scale, alpha_cols=D)
            ctx, _ = ops.attn_fwd(qkv, B, H, D)
            x_mid = ops.linear_fwd(ctx, b["proj"], bias=b["proj_b"], resid=x)
            ln2, _, _ = ops.layernorm_fwd(x_mid, b["n2w"], b["n2b"], 1e-6)
            act = ops.linear_fwd(ln2, b["fc1"], bias=b["fc1_b"], act=1)
            x = ops.linear_fwd(act, b["fc2"], bias=b["fc2_b"], resid=x_mid)
        y, _, _ = ops.layernorm_fwd(x, w["enc_nw"], w["enc_nb"], 1e-6)
        return y

    def memory(self, enc):
        """linear_norm (model_builder.py:86-89): Linear + LayerNorm(eps 1e-5) on the feature map."""
        w = self._w
        if self.use_1d_attdec:                                          # enc_x.view(B, gh, gw, C).mean(1): [B*gw, D]
            B = enc.shape[0] // self.N
            cols = torch.empty((B * self.gw, self.D), device=enc.device, dtype=enc.dtype)
            ops.window_pool_fwd(enc, cols, B, self.gh, self.gw, self.gw, self.D)
            enc = cols
        h = ops.linear_fwd(enc, w["ln_w"], bias=w["ln_b"])
        m, _, _ = ops.layernorm_fwd(h, w["ln_nw"], w["ln_nb"], 1e-5)
        return m

    def _decode_state(self, mem, n_mem, slots_per_mem=1):
        """Buffers of a K/V-cached decode over S = B * slots_per_mem sequences (greedy: 1 slot per sample; beam search: beam_width
        slots that share their sample's projected memory)."""
        w, d, nh, dk, T = self._w, self.d, self.nh, self.dk, self.max_len
        hk = nh * dk
        dev = mem.device
        S = (mem.shape[0] // n_mem) * slots_per_mem
        st = types.SimpleNamespace(S=S, n_mem=n_mem, spm=slots_per_mem)
        st.kv_mem = [ops.linear_fwd(mem, w[f"decoder.layer_stack.{i}."]["kv2"]) for i in range(self.n_layers)]      # [B*n_mem, 2hk]
        st.cache = [torch.zeros((S, T, 3 * hk), device=dev, dtype=BF16) for _ in range(self.n_layers)]
        st.x = torch.empty((S, d), device=dev, dtype=BF16)
        st.a = torch.empty((S, hk), device=dev, dtype=BF16)
        st.wts = torch.empty((S, nh, n_mem), device=dev, dtype=F32)
        st.logits = torch.empty((S, w["Cp"]), device=dev, dtype=F32)
        return st

    def _decode_step(self, st, t, tok):
        """Feed token `tok` [S] at position t through the decoder stack; leaves the classifier logits in st.logits [S, Cp] and the
        last layer's cross-attention weights in st.wts."""
        w, d, nh, dk, T, C = self._w, self.d, self.nh, self.dk, self.max_len, self.nb_classes
        hk = nh * dk
        S = st.S
        scale = dk ** -0.5
        s_ = L.stream()
        x = st.x
        L.call("dig_decode_embed", L.ptr(tok), L.ptr(w["emb"]), L.ptr(w["pos"][t]), L.ptr(x), S, d, C + 1, s_)
        for i in range(self.n_layers):
            p = w[f"decoder.layer_stack.{i}."]
            last = i == self.n_layers - 1
            h, _, _ = ops.layernorm_fwd(x, p["n1w"], p["n1b"], 1e-5)
            row = st.cache[i][:, t]                                                # [S, 3hk] view, row stride T*3hk
            ops.gemm(h, p["qkv"], S, 3 * hk, d, out=row, ldc=T * 3 * hk)
            L.call("dig_decode_self_attn", L.ptr(st.cache[i]), L.ptr(st.a), S, T, nh, dk, t, cf(scale), s_)
            x = ops.linear_fwd(st.a, p["fc"], resid=x)
            h, _, _ = ops.layernorm_fwd(x, p["n2w"], p["n2b"], 1e-5)
            q2 = ops.linear_fwd(h, p["q2"])
            L.call("dig_decode_cross_attn", L.ptr(q2), L.ptr(st.kv_mem[i]), L.ptr(st.a), L.ptr(st.wts) if last else None, S, st.n_mem, nh, dk,
                   cf(scale), st.spm, s_)
            x = ops.linear_fwd(st.a, p["fc2"], resid=x)
            h, _, _ = ops.layernorm_fwd(x, p["n3w"], p["n3b"], 1e-5)
            u = ops.linear_fwd(h, p["w1"], bias=p["b1"], act=1)
            x = ops.linear_fwd(u, p["w2"], bias=p["b2"], resid=x)
        o, _, _ = ops.layernorm_fwd(x, w["fnw"], w["fnb"], 1e-6)
        ops.gemm(o, w["cls_w"], S, w["Cp"], d, out=st.logits, out_kind=ops.OUT_F32, bias=w["cls_b"])

    def greedy_decode(self, mem, n_mem, force_tokens=None):
        """TFDecoder.forward_test with a K/V cache.  mem: bf16 [B*n_mem, d].  force_tokens ([B, max_len] int64, optional) feeds
        the given tokens instead of the arg-max (teacher forcing, for parity tests).  Returns (probs [B,T,C] fp32,
        attn_maps [B,T,n_mem] fp32, tokens [B,T] int64)."""
        w, T, C = self._w, self.max_len, self.nb_classes
        dev = mem.device
        st = self._decode_state(mem, n_mem)
        B = st.S
        tok = torch.full((B,), self.start_idx, device=dev, dtype=torch.int64)
        probs = torch.empty((B, T, C), device=dev, dtype=F32)
        maps = torch.empty((B, T, n_mem), device=dev, dtype=F32)
        toks = torch.empty((B, T), device=dev, dtype=torch.int64)
        step_probs = torch.empty((B, C), device=dev, dtype=F32)
        for t in range(T):
            self._decode_step(st, t, tok)
            L.call("dig_softmax_argmax", L.ptr(st.logits), w["Cp"], L.ptr(step_probs), L.ptr(tok), B, C, L.stream())
            probs[:, t] = step_probs
            maps[:, t] = st.wts.mean(1)
            toks[:, t] = tok
            if force_tokens is not None:
                tok = force_tokens[:, t].contiguous()
        return probs, maps, toks

    def beam_search(self, mem, n_mem, beam_width, eos=None, force_logits=None, return_logits=False):
        """TFDecoder.beam_search (models/decoder.py:254-370) on the K/V-cached decode kernels.  As in the reference the token history
        of beam slot k is what slot k emitted (decoder.py:307 never re-orders it by predecessor), so the per-slot K/V cache is
        exactly the reference's recomputation; only scores and back-pointers are re-ranked (`dig_beam_step`, one launch per step),
        and the final back-tracking runs on the host over the [T, B*beam_width] decisions (one device -> host copy).
        Returns token ids [B, max_len] int64 (the best hypothesis per sample, decoder.py:369).  force_logits ([T, S, C] fp32,
        optional): use these classifier outputs instead of the decoder's (parity tests of the bookkeeping); return_logits: also
        return the decoder's classifier outputs [T, S, C] of every step."""
        w, T, C = self._w, self.max_len, self.nb_classes
        eos = self.eos if eos is None else eos
        dev = mem.device
        bw = int(beam_width)
        st = self._decode_state(mem, n_mem, slots_per_mem=bw)
        S = st.S
        B = S // bw
        tok = torch.full((S,), self.start_idx, device=dev, dtype=torch.int64)
        seq_scores = torch.full((S,), float("-inf"), device=dev, dtype=F32)
        seq_scores[::bw] = 0.0                                              # only slot 0 of a sample is live at step 0 (:272-274)
        syms = torch.empty((T, S), device=dev, dtype=torch.int64)
        preds = torch.empty((T, S), device=dev, dtype=torch.int64)
        scores = torch.empty((T, S), device=dev, dtype=F32)
        kept = []
        for t in range(T):
            if force_logits is None:
                self._decode_step(st, t, tok)
                lg, ld = st.logits, w["Cp"]
                if return_logits:
                    kept.append(st.logits[:, :C].clone())
            else:
                lg, ld = force_logits[t].contiguous(), force_logits.shape[-1]
            L.call("dig_beam_step", L.ptr(lg), ld, L.ptr(seq_scores), B, bw, C, eos, L.ptr(syms[t]), L.ptr(preds[t]), L.ptr(scores[t]), L.stream())
            tok = syms[t]
        ids = beam_backtrack(scores.cpu().numpy(), preds.cpu().numpy(), syms.cpu().numpy(), B, bw, eos).to(dev)
        return (ids, torch.stack(kept)) if return_logits else ids

    def forward(self, x):
        if self.training:
            raise NotImplementedError("the fine-tune training step (SURVEY.md 8f row N1) is not built; call .eval()")
        images = x[0] if isinstance(x, (tuple, list)) else x
        if not images.is_cuda:
            raise RuntimeError("dig_amd.RecModel runs on an MI355X (cuda device) only; there is no CPU fallback")
        if not self._sd:
            raise RuntimeError("load_state_dict() first")
        if not self._ready or self._dev != images.device:
            self._prepare(images.device)
        with torch.no_grad():
            if self.beam_width > 0:
                # RecModel.forward -> TFDecoder.forward(..., beam_width) (model_builder.py:151-158, decoder.py:101-102): token ids
                # [B, max_len] and an all-ones tensor in place of (probabilities, attention maps)
                ids = self.beam_search(self.memory(self.encoder_features(images)), self.n_mem, self.beam_width)
                return ids, None, None, torch.ones_like(ids)
            if not self.use_hip_graph:
                probs, maps, _ = self._recognize(images)
                return probs, None, None, maps
            # The decode loop is ~2 200 launches of microsecond kernels per batch: launch-bound.  It is captured once per batch
            # shape into a HIP graph (inputs / outputs live in static buffers) and replayed.
            key = (tuple(images.shape), images.device)
            ent = self._graphs.get(key)
            if ent is None:
                static_in = images.detach().clone().float().contiguous()
                side = torch.cuda.Stream(device=images.device)
                side.wait_stream(torch.cuda.current_stream(images.device))
                with torch.cuda.stream(side):
                    self._recognize(static_in)                               # warm-up outside capture (lazy init, attributes)
                torch.cuda.current_stream(images.device).wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    probs, maps, toks = self._recognize(static_in)
                ent = self._graphs[key] = (graph, static_in, probs, maps, toks)
            graph, static_in, probs, maps, toks = ent
            static_in.copy_(images)
            graph.replay()
            return probs.clone(), None, None, maps.clone()

    def _recognize(self, images):
        enc = self.encoder_features(images)
        mem = self.memory(enc)
        return self.greedy_decode(mem, self.n_mem)


def beam_backtrack(scores, preds, syms, B, bw, eos):
    """The back-tracking that ends TFDecoder.beam_search (models/decoder.py:311-369): numpy [T, B*bw] decisions -> int64 [B, T].
    Walk the back-pointers from the best final beams; a hypothesis that emitted EOS at step t takes over a slot of its sample from
    the back (worst live beam first) with the score it had when it ended; finally re-sort by score and keep beam 0."""
    import numpy as np
    T = syms.shape[0]
    base = (np.arange(B) * bw)[:, None]
    last = scores[-1].reshape(B, bw)
    order = np.argsort(-last, axis=1, kind="stable")
    s = np.take_along_axis(last, order, 1).copy()
    t_pred = (order + base).reshape(-1)
    found = [0] * B
    rows = []
    for t in range(T - 1, -1, -1):
        cur = syms[t][t_pred].copy()
        t_pred = preds[t][t_pred].copy()
        ended = np.nonzero(syms[t] == eos)[0]
        for idx in ended[::-1]:
            b = int(idx) // bw
            k = bw - (found[b] % bw) - 1
            found[b] += 1
            t_pred[b * bw + k] = preds[t][idx]
            cur[b * bw + k] = syms[t][idx]
            s[b, k] = scores[t][idx]
        rows.append(cur)
    best = np.argsort(-s, axis=1, kind="stable")[:, 0] + base[:, 0]
    out = np.stack([r[best] for r in reversed(rows)], axis=1)
    return torch.from_numpy(out.astype(np.int64))


def class_canon(voc):
    """canonical code per class for `Accuracy` (evaluation_metric/metrics.py:14-16,19-62): digits / letters -> 1 + index in
    '0-9a-z' (case-folded); everything else (punctuation, EOS, PADDING, UNKNOWN) -> 0 = dropped."""
    import string
    keep = string.digits + string.ascii_lowercase
    return torch.tensor([1 + keep.index(c.lower()) if (len(c) == 1 and c.lower() in keep) else 0 for c in voc], dtype=torch.uint8)


def accuracy(pred_tokens, target_tokens, voc):
    """`Accuracy(output, target, dataset)` of the reference on device tensors [B, T] of class ids; returns a 0-dim tensor."""
    B, T = pred_tokens.shape
    dev = pred_tokens.device
    canon = class_canon(voc).to(dev)
    match = torch.empty(B, device=dev, dtype=torch.uint8)
    pred, targ = pred_tokens.contiguous(), target_tokens.to(dev).contiguous()
    L.call("dig_string_match", L.ptr(pred), L.ptr(targ), L.ptr(canon), len(voc), voc.index("EOS"), B, T, L.ptr(match), L.stream())
    return match.float().mean()


def recognition_f_measure(pred_tokens, target_tokens, voc):
    """`recognition_f_measure` (evaluation_metric/metrics.py:83-100) on device tensors; returns a 0-dim float64 tensor."""
    B, T = pred_tokens.shape
    dev = pred_tokens.device
    f = torch.empty(B, device=dev, dtype=torch.float64)
    pred, targ, canon = pred_tokens.contiguous(), target_tokens.to(dev).contiguous(), class_canon(voc).to(dev)   # (named: alive across the call)
    L.call("dig_char_fmeasure", L.ptr(pred), L.ptr(targ), L.ptr(canon), len(voc), voc.index("EOS"), B, T, L.ptr(f), L.stream())
    return f.mean()


class SeqCrossEntropyLoss(torch.nn.Module):
    """loss/seqCrossEntropyLoss.py (sample_normalize): forward(input [B,T,C] fp32, target [B,T], length [B]) -> 0-dim loss.
    Forward only (evaluation); the training loss with its gradient belongs to row N1."""

    def forward(self, input, target, length):
        B, T, C = input.shape
        dev = input.device
        inp = input.detach().float().contiguous()
        rows = torch.empty(B * T, device=dev, dtype=F32)
        loss = torch.empty(1, device=dev, dtype=F32)
        tgt, lens = target.to(dev).long().contiguous(), length.to(dev).long().contiguous()
        L.call("dig_seq_cross_entropy", L.ptr(inp), L.ptr(tgt), L.ptr(lens), B, T, C, L.ptr(rows), L.ptr(loss), L.stream())
        return loss[0]
