"""`AttnRecModel` on the MI355X -- SURVEY.md 8(f) row N1, the second decoder (`--decoder_type attention`,
run_class_finetuning.py:352-353): the fine-tune encoder followed by the GRU attention recognition head.

Mirrors models/model_builder.py:40-72 (`AttnRecModel`: encoder tokens [B, 256, D] straight into the head, no `linear_norm`) and
models/attn_decoder.py:11-78,197-272 (`AttentionRecognitionHead.forward_train` -- max(lengths) teacher-forced steps, outputs
zero-padded to max_len -- and the greedy `sample`; `DecoderUnit` = additive attention + target embedding + one nn.GRU cell + classifier).
Same state-dict keys as the reference (`decoder.decoder.attention_unit.{sEmbed,xEmbed,wEmbed}.*`, `decoder.decoder.tgt_embedding.weight`,
`decoder.decoder.gru.{weight,bias}_{ih,hh}_l0`, `decoder.decoder.fc.*`).  Flat arenas, one autograd node, the encoder on the
pre-training hot-path kernels (dig_amd.finetune._TrainStep); the head's steps run on `dig_gemm_bf16` + `dig_addattn_*` +
`dig_gru_cell_*` (csrc/gru_attn.hip), weight gradients as ONE GEMM per weight over the stacked steps.  Beam search is not built."""
import ctypes
import math
from collections import OrderedDict

import torch

from . import _lib as L
from . import ops
from .finetune import RecModelTrain, _TrainStep, CLS_PAD
from .recognizer import ENCODERS

BF16, F32 = torch.bfloat16, torch.float32
PRE = "decoder.decoder."


class _AttnTrainStep(_TrainStep):
    """Forward / backward of one AttnRecModel step (teacher forcing)."""

    def _w(self):
        M = self.m
        A, S, X, E, C = M.attDim, M.sDim, M.D, M.attDim, M.nb_classes
        return A, S, X, E, C

    def forward(self, images, targets, lens):
        M = self.m
        dev = images.device
        x = self.encoder_forward(images)                                      # [B*N, X] bf16
        B, N = self.B, M.N
        A, S, X, E, C = self._w()
        hint = getattr(M, "_steps_hint", None)                                # max(lengths) read from the host copy of tgt_lens (no device sync)
        steps = hint if hint is not None else (int(lens.max().item()) if lens.numel() else 0)
        steps = max(0, min(steps, M.max_len))
        self.steps, self.x = steps, x
        self.targets, self.lens = targets.long().contiguous(), lens.long().contiguous()
        logits = torch.zeros((B, M.max_len, C), device=dev, dtype=F32)
        if steps == 0:
            return logits
        p, w = self.p, self.w
        xproj = ops.linear_fwd(x, w(PRE + "attention_unit.xEmbed.weight"), bias=p(PRE + "attention_unit.xEmbed.bias"))   # [B*N, A]
        wv = p(PRE + "attention_unit.wEmbed.weight").reshape(A).contiguous()
        # teacher forcing: every step's previous token is known up front (attn_decoder.py:46-50): <BOS> = num_classes, then targets[:, i-1]
        yprev = torch.cat([torch.full((B, 1), C, device=dev, dtype=torch.int64), self.targets[:, :steps - 1]], 1).t().contiguous()   # [steps, B]
        inp = torch.empty((steps, B, E + X), device=dev, dtype=BF16)          # GRU inputs [yProj | context] of every step
        L.call("dig_embed_rows", L.ptr(yprev), L.ptr(p(PRE + "tgt_embedding.weight")), L.ptr(inp), E + X, steps * B, E, C + 1, L.stream())
        s_all = torch.empty((steps, B, S), device=dev, dtype=F32)
        sbf_all = torch.empty((steps + 1, B, S), device=dev, dtype=BF16)      # sbf_all[t] = state BEFORE step t (row 0 = zeros)
        sbf_all[0].zero_()
        sproj_all = torch.empty((steps, B, A), device=dev, dtype=BF16)
        alpha_all = torch.empty((steps, B, N), device=dev, dtype=F32)
        gates_all = torch.empty((steps, B, 4 * S), device=dev, dtype=F32)
        wih, whh = w(PRE + "gru.weight_ih_l0"), w(PRE + "gru.weight_hh_l0")
        bih, bhh = p(PRE + "gru.bias_ih_l0"), p(PRE + "gru.bias_hh_l0")
        ws_, bs_ = w(PRE + "attention_unit.sEmbed.weight"), p(PRE + "attention_unit.sEmbed.bias")
        for t in range(steps):
            ops.linear_fwd(sbf_all[t], ws_, bias=bs_, out=sproj_all[t])
            L.call("dig_addattn_fwd", L.ptr(xproj), L.ptr(sproj_all[t]), L.ptr(wv), L.ptr(x), L.ptr(alpha_all[t]), L.ptr(inp[t][:, E:]), E + X,
                   B, N, A, X, L.stream())
            gi = ops.linear_fwd(inp[t], wih, bias=bih)
            gh = ops.linear_fwd(sbf_all[t], whh, bias=bhh)
            L.call("dig_gru_cell_fwd", L.ptr(gi), L.ptr(gh), L.ptr(s_all[t - 1]) if t else None, L.ptr(s_all[t]), L.ptr(sbf_all[t + 1]),
                   L.ptr(gates_all[t]), B, S, L.stream())
        # classifier for all steps at once (rows (t, b))
        self.cls_w = torch.zeros((CLS_PAD, S), device=dev, dtype=BF16)
        self.cls_w[:C] = w(PRE + "fc.weight")
        cb = torch.zeros(CLS_PAD, device=dev, dtype=F32)
        cb[:C] = p(PRE + "fc.bias")
        out = torch.empty((steps * B, CLS_PAD), device=dev, dtype=F32)
        ops.gemm(sbf_all[1:].reshape(steps * B, S), self.cls_w, steps * B, CLS_PAD, S, out=out, out_kind=ops.OUT_F32, bias=cb)
        logits[:, :steps] = out.view(steps, B, CLS_PAD)[:, :, :C].transpose(0, 1)
        self.saved = (xproj, wv, yprev, inp, s_all, sbf_all, sproj_all, alpha_all, gates_all)
        return logits

    def backward(self, dlogits_btc):
        M = self.m
        dev = dlogits_btc.device
        main, sd = self.begin_backward(dev)
        side = self.side
        B, N, steps = self.B, M.N, self.steps
        A, S, X, E, C = self._w()
        x = self.x
        if steps == 0:
            self.encoder_backward(torch.zeros_like(x))
            main.wait_stream(sd)
            return
        xproj, wv, yprev, inp, s_all, sbf_all, sproj_all, alpha_all, gates_all = self.saved
        p, w, g = self.p, self.w, self.g
        rows = steps * B
        # classifier (all steps): dout rows (t, b)
        dl = torch.zeros((rows, CLS_PAD), device=dev, dtype=BF16)
        dl.view(steps, B, CLS_PAD)[:, :, :C] = dlogits_btc[:, :steps].transpose(0, 1).to(BF16)
        snew = sbf_all[1:].reshape(rows, S)
        side(lambda: ops.wgrad(dl, snew, g(PRE + "fc.weight"), C, S, rows), dl, snew)
        cs = torch.zeros(CLS_PAD, device=dev, dtype=F32)
        side(lambda: (ops.colsum(dl, cs, cols=CLS_PAD), g(PRE + "fc.bias").add_(cs[:C])), dl, cs)
        ds_fc = torch.empty((rows, S), device=dev, dtype=F32)                 # classifier path into every step's new state
        ops.gemm(dl, self.cls_w, rows, S, CLS_PAD, tb=True, out=ds_fc, out_kind=ops.OUT_F32)
        ds_fc = ds_fc.view(steps, B, S)
        wih, whh, ws_ = w(PRE + "gru.weight_ih_l0"), w(PRE + "gru.weight_hh_l0"), w(PRE + "attention_unit.sEmbed.weight")
        dgi_all = torch.empty((steps, B, 3 * S), device=dev, dtype=BF16)
        dgh_all = torch.empty((steps, B, 3 * S), device=dev, dtype=BF16)
        dinp_all = torch.empty((steps, B, E + X), device=dev, dtype=BF16)
        dsproj_all = torch.empty((steps, B, A), device=dev, dtype=BF16)
        dv_all = torch.empty((steps, B, N), device=dev, dtype=F32)
        dw_acc = torch.zeros((B, A), device=dev, dtype=F32)
        dz = torch.empty((B, S), device=dev, dtype=F32)                       # z * ds of the later step
        d1 = torch.empty((B, S), device=dev, dtype=F32)                       # dgh @ W_hh of the later step
        d2 = torch.empty((B, S), device=dev, dtype=F32)                       # dsproj @ W_s of the later step
        for t in reversed(range(steps)):
            last = t == steps - 1
            L.call("dig_gru_cell_bwd", L.ptr(ds_fc[t]), None if last else L.ptr(dz), None if last else L.ptr(d1), None if last else L.ptr(d2),
                   L.ptr(gates_all[t]), L.ptr(s_all[t - 1]) if t else None, L.ptr(dgi_all[t]), L.ptr(dgh_all[t]), L.ptr(dz), B, S, L.stream())
            ops.gemm(dgi_all[t], wih, B, E + X, 3 * S, tb=True, out=dinp_all[t])            # d[yProj | context]
            L.call("dig_addattn_bwd", L.ptr(xproj), L.ptr(sproj_all[t]), L.ptr(wv), L.ptr(x), L.ptr(alpha_all[t]), L.ptr(dinp_all[t][:, E:]), E + X,
                   L.ptr(dv_all[t]), L.ptr(dsproj_all[t]), L.ptr(dw_acc), B, N, A, X, L.stream())
            if t:
                ops.gemm(dgh_all[t], whh, B, S, 3 * S, tb=True, out=d1, out_kind=ops.OUT_F32)
                ops.gemm(dsproj_all[t], ws_, B, S, A, tb=True, out=d2, out_kind=ops.OUT_F32)
        # ---- weight gradients: one GEMM per weight over the stacked steps (rows (t, b)); state before step t = sbf_all[t]
        sprev = sbf_all[:steps].reshape(rows, S)
        dgi2, dgh2, dsp2, inp2 = dgi_all.view(rows, 3 * S), dgh_all.view(rows, 3 * S), dsproj_all.view(rows, A), inp.view(rows, E + X)
        side(lambda: ops.linear_wgrad(dgi2, inp2, g(PRE + "gru.weight_ih_l0")), dgi2, inp2)
        side(lambda: ops.colsum(dgi2, g(PRE + "gru.bias_ih_l0")), dgi2)
        side(lambda: ops.linear_wgrad(dgh2, sprev, g(PRE + "gru.weight_hh_l0")), dgh2, sprev)
        side(lambda: ops.colsum(dgh2, g(PRE + "gru.bias_hh_l0")), dgh2)
        side(lambda: ops.linear_wgrad(dsp2, sprev, g(PRE + "attention_unit.sEmbed.weight")), dsp2, sprev)
        side(lambda: ops.colsum(dsp2, g(PRE + "attention_unit.sEmbed.bias")), dsp2)
        side(lambda: ops.colsum_partials(dw_acc, g(PRE + "attention_unit.wEmbed.weight").view(A)), dw_acc)   # (wEmbed.bias: softmax-invariant, gradient 0)
        dyp = dinp_all.view(rows, E + X)[:, :E].contiguous()
        side(lambda: L.call("dig_seq_embed_bwd", L.ptr(yprev), L.ptr(dyp), L.ptr(g(PRE + "tgt_embedding.weight")), rows, E, C + 1, L.stream()), dyp)
        # ---- token gradients: sums over the steps, then the xEmbed Linear
        dxproj = torch.empty_like(xproj)
        dx = torch.empty_like(x)
        L.call("dig_addattn_bwd_tokens", L.ptr(xproj), L.ptr(sproj_all), L.ptr(wv), L.ptr(dv_all), L.ptr(alpha_all), L.ptr(dinp_all.view(rows, E + X)[:, E:]),
               E + X, L.ptr(dxproj), L.ptr(dx), steps, B, N, A, X, L.stream())
        side(lambda: ops.linear_wgrad(dxproj, x, g(PRE + "attention_unit.xEmbed.weight")), dxproj, x)
        side(lambda: ops.colsum(dxproj, g(PRE + "attention_unit.xEmbed.bias")), dxproj)
        dx2 = ops.linear_dgrad(dxproj, w(PRE + "attention_unit.xEmbed.weight"))
        ops.add_bf16(dx, dx2, dx)
        self.encoder_backward(dx)
        main.wait_stream(sd)


class AttnRecModelTrain(RecModelTrain):
    """models.model_builder.AttnRecModel: `.train()` forward = teacher-forced logits [B, max_len, nb_classes] (+ three Nones),
    `.eval()` forward = greedy `sample` probabilities, both as the reference returns them."""
    _step_cls = _AttnTrainStep

    def __init__(self, args=None, *, embed_dim=None, depth=12, num_heads=None, nb_classes=97, max_len=25, sDim=512, attDim=512, drop_rate=None,
                 attn_drop_rate=None, drop_path_rate=None, drop_seed=None):
        if args is not None:
            embed_dim, num_heads = ENCODERS[args.model]
            nb_classes, max_len = args.nb_classes, args.max_len
            if getattr(args, "beam_width", 0):
                # the reference never reaches AttentionRecognitionHead.beam_search: AttnRecModel.forward (model_builder.py:66-72) calls
                # the head's __call__ (forward_train / sample) whatever args.beam_width says, and the method itself indexes with the float
                # result of `candidates / num_classes` (attn_decoder.py:126-127).  Refused here rather than silently decoded greedily.
                raise NotImplementedError("--beam_width with --decoder_type attention: the reference's AttnRecModel ignores it (greedy sample()); "
                                          "beam search is built for tf_decoder")
            drop_rate = float(getattr(args, "drop", 0.0)) if drop_rate is None else drop_rate
            attn_drop_rate = float(getattr(args, "attn_drop_rate", 0.0)) if attn_drop_rate is None else attn_drop_rate
            drop_path_rate = float(getattr(args, "drop_path", 0.0)) if drop_path_rate is None else drop_path_rate
        if attDim % 8 or attDim > 1024 or sDim % 64 or (attDim + embed_dim) % 64:
            raise NotImplementedError("attDim must be a multiple of 8 (<= 1024), sDim and attDim + embed_dim multiples of 64")
        self.sDim, self.attDim = sDim, attDim
        super().__init__(None, embed_dim=embed_dim, depth=depth, num_heads=num_heads, n_layers=0, d_model=attDim, n_head=1, d_k=64, d_inner=0,
                         nb_classes=nb_classes, max_len=max_len, drop_rate=drop_rate or 0.0, attn_drop_rate=attn_drop_rate or 0.0,
                         drop_path_rate=drop_path_rate or 0.0, decoder_dropout=0.0, drop_seed=drop_seed)

    def param_shapes(self):
        D, F, A, S, C = self.D, self.F, self.attDim, self.sDim, self.nb_classes
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
        o[PRE + "attention_unit.sEmbed.weight"] = (A, S); o[PRE + "attention_unit.sEmbed.bias"] = (A,)
        o[PRE + "attention_unit.xEmbed.weight"] = (A, D); o[PRE + "attention_unit.xEmbed.bias"] = (A,)
        o[PRE + "attention_unit.wEmbed.weight"] = (1, A); o[PRE + "attention_unit.wEmbed.bias"] = (1,)
        o[PRE + "tgt_embedding.weight"] = (C + 1, A)
        o[PRE + "gru.weight_ih_l0"] = (3 * S, D + A); o[PRE + "gru.weight_hh_l0"] = (3 * S, S)
        o[PRE + "gru.bias_ih_l0"] = (3 * S,); o[PRE + "gru.bias_hh_l0"] = (3 * S,)
        o[PRE + "fc.weight"] = (C, S); o[PRE + "fc.bias"] = (C,)
        return o

    def _init_tensor(self, k, shp):
        """Head: PyTorch defaults, which AttnRecModel keeps (the `init_weights` methods of attn_decoder.py are commented out,
        :206,249): nn.Linear U(+-1/sqrt(fan_in)), nn.Embedding N(0,1), nn.GRU U(+-1/sqrt(sDim)) for all four tensors."""
        if not k.startswith(PRE):
            return super()._init_tensor(k, shp)
        if "tgt_embedding" in k:
            return torch.randn(shp)
        if ".gru." in k:
            return (torch.rand(shp) * 2 - 1) / math.sqrt(self.sDim)
        fan_in = self.param_shapes()[k.rsplit(".", 1)[0] + ".weight"][1]
        return (torch.rand(shp) * 2 - 1) / math.sqrt(fan_in)

    def no_weight_decay(self):
        return {"encoder.pos_embed", "encoder.cls_token"}

    # ------------------------------------------------------------------ eval: greedy sample (attn_decoder.py:58-78)
    @torch.no_grad()
    def sample(self, images):
        dev = images.device
        if self._dev != dev or self._shadow is None:
            self._prepare_train(dev)
        step = _AttnTrainStep(self)
        saved = (self.drop_rate, self.attn_drop_rate, self.dpr, self.drop_step)
        self.drop_rate, self.attn_drop_rate, self.dpr = 0.0, 0.0, [0.0] * self.depth    # eval mode: no dropout
        try:
            x = step.encoder_forward(images)
        finally:
            self.drop_rate, self.attn_drop_rate, self.dpr, self.drop_step = saved
        B, N = step.B, self.N
        A, S, X, E, C = step._w()
        p, w = step.p, step.w
        xproj = ops.linear_fwd(x, w(PRE + "attention_unit.xEmbed.weight"), bias=p(PRE + "attention_unit.xEmbed.bias"))
        wv = p(PRE + "attention_unit.wEmbed.weight").reshape(A).contiguous()
        cls_w = torch.zeros((CLS_PAD, S), device=dev, dtype=BF16)
        cls_w[:C] = w(PRE + "fc.weight")
        cb = torch.zeros(CLS_PAD, device=dev, dtype=F32)
        cb[:C] = p(PRE + "fc.bias")
        probs = torch.empty((B, self.max_len, C), device=dev, dtype=F32)
        s = torch.zeros((B, S), device=dev, dtype=F32)
        sbf = torch.zeros((B, S), device=dev, dtype=BF16)
        y = torch.full((B,), C, device=dev, dtype=torch.int64)
        inp = torch.empty((B, E + X), device=dev, dtype=BF16)
        alpha = torch.empty((B, N), device=dev, dtype=F32)
        gates = torch.empty((B, 4 * S), device=dev, dtype=F32)
        logit = torch.empty((B, CLS_PAD), device=dev, dtype=F32)
        tok = torch.empty((B,), device=dev, dtype=torch.int64)
        pbuf = torch.empty((B, C), device=dev, dtype=F32)
        for t in range(self.max_len):
            sproj = ops.linear_fwd(sbf, w(PRE + "attention_unit.sEmbed.weight"), bias=p(PRE + "attention_unit.sEmbed.bias"))
            L.call("dig_embed_rows", L.ptr(y), L.ptr(p(PRE + "tgt_embedding.weight")), L.ptr(inp), E + X, B, E, C + 1, L.stream())
            L.call("dig_addattn_fwd", L.ptr(xproj), L.ptr(sproj), L.ptr(wv), L.ptr(x), L.ptr(alpha), L.ptr(inp[:, E:]), E + X, B, N, A, X, L.stream())
            gi = ops.linear_fwd(inp, w(PRE + "gru.weight_ih_l0"), bias=p(PRE + "gru.bias_ih_l0"))
            gh = ops.linear_fwd(sbf, w(PRE + "gru.weight_hh_l0"), bias=p(PRE + "gru.bias_hh_l0"))
            s_new = torch.empty_like(s)
            sbf_new = torch.empty_like(sbf)
            L.call("dig_gru_cell_fwd", L.ptr(gi), L.ptr(gh), L.ptr(s), L.ptr(s_new), L.ptr(sbf_new), L.ptr(gates), B, S, L.stream())
            s, sbf = s_new, sbf_new
            ops.gemm(sbf, cls_w, B, CLS_PAD, S, out=logit, out_kind=ops.OUT_F32, bias=cb)
            L.call("dig_softmax_argmax", L.ptr(logit), CLS_PAD, L.ptr(pbuf), L.ptr(tok), B, C, L.stream())
            probs[:, t] = pbuf
            y = tok.clone()
        return probs

    def forward(self, x):
        if self.training:
            lens = x[2]
            self._steps_hint = int(lens.max()) if (torch.is_tensor(lens) and not lens.is_cuda and lens.numel()) else None
            return super().forward(x)
        images = x[0] if isinstance(x, (tuple, list)) else x
        if not images.is_cuda:
            raise RuntimeError("dig_amd.AttnRecModelTrain runs on an MI355X (cuda device) only; there is no CPU fallback")
        return self.sample(images), None, None, None
