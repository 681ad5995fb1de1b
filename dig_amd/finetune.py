"""Fine-tune TRAINING step of the recognition model on the MI355X -- SURVEY.md 8(f) row N1.

Mirrors (label smoothing 0, as in README.md:92-118), including the stochastic regularisers `--drop`, `--attn_drop_rate`,
`--drop_path` of the encoder and the recognition decoder's own dropout (0.1, hard-wired by models/decoder.py:13-18,141) -- masks
come from a keyed counter hash instead of torch's generator, see dig_amd/dropout.py:
  * `RecModel.forward` in train mode (models/model_builder.py:124-160) -> `TFDecoder.forward_train` (models/decoder.py:196-222):
    `model((images, targets, tgt_lens))` returns the logits [B, max_len, nb_classes] (+ three Nones, as the reference does);
  * `SeqCrossEntropyLoss` (loss/seqCrossEntropyLoss.py) with its gradient;
  * `create_optimizer(args, model, get_num_layer=..., get_layer_scale=...)` (optim_factory.py:33-100, layer-wise lr decay as in
    run_class_finetuning.py:471-520) as one fused AdamW launch over a flat parameter arena.
Parameters, gradients, Adam moments and the bf16 GEMM operands live in flat arenas (each tensor padded to 256 elements) whose
layout keeps q|k|v (and k|v) projection weights adjacent, so the fused projections are views.  The whole model is ONE autograd
node with a hand-written backward on the hot-path kernels (encoder: the pre-training kernels; decoder: `dig_seq_attn_*`,
`dig_seq_embed_*`, `dig_gemm_bf16`, `dig_layernorm_*`).  The CPU checker of this step lives with the tests."""
import ctypes
import math
import os
from collections import OrderedDict

import torch

from . import _lib as L
from . import dropout as DR
from . import ops
from .recognizer import RecModel, _encoder_pos, _sinusoid

BF16, F32 = torch.bfloat16, torch.float32
cf = ctypes.c_float
FT_FUSED_QV = os.environ.get("DIG_FT_FUSED_QV", "1") != "0"
# the encoder's MLP half as the pre-training step's fused launches: norm2 -> fc1 -> GELU -> fc2 -> Mlp.drop / drop_path -> + residual -> next norm1 in
# one forward launch (dig_mlp_chain_fwd_ln_dropout), both data gradients in one backward launch (dig_mlp_chain_bwd on the masked gradient)
FT_MLP_CHAIN = os.environ.get("DIG_FT_MLP_CHAIN", "1") != "0"
FT_BATCH_REDUCE = os.environ.get("DIG_FT_BATCH_REDUCE", "0") == "1"      # opt-in: fewer launches, ~0.3 ms slower per step
CLS_PAD = 128                     # classifier rows padded to a multiple of 64 (it is a non-transposed GEMM operand in backward)


def _ref(spec):
    return ctypes.byref(spec) if spec is not None else None


def _pad256(n):
    return (n + 255) // 256 * 256


class RecModelTrain(RecModel):
    """`RecModel` with trainable flat arenas.  `.train()` forward = teacher-forced logits with autograd; `.eval()` = greedy decode."""
    _step_cls = None                                    # set below (the forward / backward of one step)

    def __init__(self, args=None, *, drop_rate=None, attn_drop_rate=None, drop_path_rate=None, decoder_dropout=0.1, drop_seed=None, **kw):
        """Drop rates: from `args` (--drop / --attn_drop_rate / --drop_path, run_class_finetuning.py:69-74 -> create_model,
        :300-313) unless given; `decoder_dropout` = 0.1 is what `create_decoder` builds (models/decoder.py:13-18: TFDecoder's
        default, not configurable in the reference).  drop_seed: base seed of the mask keys (default: torch.initial_seed())."""
        super().__init__(args, **kw)
        self.drop_rate = float(getattr(args, "drop", 0.0) if drop_rate is None else drop_rate)
        self.attn_drop_rate = float(getattr(args, "attn_drop_rate", 0.0) if attn_drop_rate is None else attn_drop_rate)
        self.drop_path_rate = float(getattr(args, "drop_path", 0.0) if drop_path_rate is None else drop_path_rate)
        self.decoder_dropout = float(decoder_dropout)
        for name in ("drop_rate", "attn_drop_rate", "drop_path_rate", "decoder_dropout"):
            if not 0.0 <= getattr(self, name) < 1.0:
                raise ValueError(f"{name} must be in [0, 1), got {getattr(self, name)}")
        # stochastic depth decay rule (modeling_finetune.py:273)
        self.dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.depth)]
        self.drop_seed = torch.initial_seed() if drop_seed is None else int(drop_seed)
        self.drop_step = 0                              # training forwards so far: every step draws fresh keys
        self.frozen = set()                             # names with requires_grad = False (`fix_encoder_layers`)
        self.frozen_blocks = 0                          # encoder blocks 0 .. frozen_blocks-1 (and patch_embed) take no gradient
        self.comm = None
        self._dev = None
        self._offsets = OrderedDict()
        off = 0
        shapes = self.param_shapes()
        for k, s in shapes.items():
            n = 1
            for d_ in s:
                n *= d_
            if k.endswith("attn.q_bias") and k[:-6] + "v_bias" in shapes:
                # q_bias | zeros (K has no bias, modeling_finetune.py:91) | v_bias laid out as ONE [3D] vector: the fused qkv GEMM takes it as
                # its bias without a per-step concatenation (the gap belongs to no optimizer granule and stays zero)
                self._offsets[k] = (off, n, tuple(s))
                self._offsets[k[:-6] + "v_bias"] = (off + 2 * n, n, tuple(s))
                off += _pad256(3 * n)
                continue
            if k in self._offsets:
                continue                                                    # (v_bias: placed with its q_bias)
            self._offsets[k] = (off, n, tuple(s))
            off += _pad256(n)
        self._offsets = OrderedDict((k, self._offsets[k]) for k in shapes)      # registration order (state_dict / optimizer indices)
        self.n_flat = off
        self.flat_params = torch.zeros(off, dtype=F32)
        self.flat_grads = torch.zeros(off, dtype=F32)
        self._shadow = None
        self.init_weights()

    def init_weights(self):
        """The reference's initialisation, drawn from torch's CPU generator: encoder Linear weights xavier-uniform with zero
        biases and LayerNorm (1, 0) (`PretrainVisionTransformerEncoder._init_weights`, modeling_pretrain_vit.py:66-74); the
        patch-embed convolution, the decoder and `linear_norm` keep PyTorch's defaults (nn.Conv2d / nn.Linear: U(-1/sqrt(fan_in),
        1/sqrt(fan_in)) for weight and bias; nn.Embedding: N(0, 1); models/model_builder.py:78-89 adds nothing); mask_token,
        q_bias, v_bias zero.  Fine-tuning then overwrites the encoder from the pre-training checkpoint (`load_pretrained`)."""
        sd = OrderedDict((k, self._init_tensor(k, shp)) for k, shp in self.param_shapes().items())
        self.load_state_dict(sd)

    def _init_tensor(self, k, shp):
        enc = k.startswith("encoder.")
        if k.endswith("mask_token") or k.endswith("q_bias") or k.endswith("v_bias"):
            return torch.zeros(shp)
        if "norm" in k.split(".")[-2] and len(shp) == 1 or k.startswith("linear_norm.1."):
            return torch.ones(shp) if k.endswith("weight") else torch.zeros(shp)
        if k.endswith("trg_word_emb.weight"):
            return torch.randn(shp)
        if k.endswith(".weight"):
            fan_in = 1
            for d_ in shp[1:]:
                fan_in *= d_
            a = math.sqrt(6.0 / (shp[0] + shp[1])) if enc and len(shp) == 2 else 1.0 / math.sqrt(fan_in)   # xavier_uniform_ / kaiming_uniform_(a=sqrt(5))
            return (torch.rand(shp) * 2 - 1) * a
        if enc and "patch_embed" not in k:                                     # biases
            return torch.zeros(shp)
        w = self.param_shapes()[k[:-4] + "weight"]
        fan_in = 1
        for d_ in w[1:]:
            fan_in *= d_
        return (torch.rand(shp) * 2 - 1) / math.sqrt(fan_in)

    def load_pretrained(self, checkpoint, model_key="model|module", prefix=""):
        """run_class_finetuning.py:362-440 for the simmim_vit encoders: pick `checkpoint[model_key]`, strip a `backbone.` prefix,
        load what matches (`encoder.*` of the pre-training model IS this model's `encoder.*`), report the rest.  Returns
        (missing, unexpected) as the reference's `utils.load_state_dict` prints them."""
        from .utils import load_state_dict
        ck = None
        for key in model_key.split("|"):
            if key in checkpoint:
                ck = checkpoint[key]
                break
        if ck is None:
            ck = checkpoint
        new = OrderedDict((k[9:] if k.startswith("backbone.") else k, v) for k, v in ck.items())
        return load_state_dict(self, new, prefix=prefix)

    # ------------------------------------------------------------------ state
    def _view(self, flat, k, dtype_shape=True):
        o, n, s = self._offsets[k]
        return flat[o:o + n].view(s)

    def load_state_dict(self, state_dict, strict=True):
        super().load_state_dict(state_dict, strict)
        for k in self._offsets:
            if k in state_dict:
                self._view(self.flat_params, k).copy_(self._sd[k])
        self._ready = False
        self.weights_changed()

    def state_dict(self, *a, **k):
        return OrderedDict((n, self._view(self.flat_params, n).detach().cpu().clone()) for n in self._offsets)

    def fix_encoder_layers(self, fixed_encoder_layers):
        """`--fixed_encoder_layers k` (run_class_finetuning.py:500-518): k >= 1 freezes `encoder.patch_embed`, k > 1 also the encoder
        blocks with index < k - 1 (k capped at depth + 1).  Call BEFORE `create_optimizer` (frozen parameters are not listed there, as
        `get_parameter_groups` skips `requires_grad = False`).  The backward stops above the last frozen block: nothing below it is
        computed.  Returns the frozen names (what the reference prints)."""
        k = int(fixed_encoder_layers)
        self.frozen, self.frozen_blocks = set(), 0
        if k >= 1:
            self.frozen |= {n for n in self._offsets if "encoder.patch_embed" in n}
        if k > 1:
            k = min(k, self.depth + 1)
            self.frozen_blocks = k - 1
            self.frozen |= {n for n in self._offsets if n.startswith("encoder.blocks.") and int(n.split(".")[2]) < k - 1}
        return [n for n in self._offsets if n in self.frozen]

    def named_parameters(self, *a, **k):
        for n in self._offsets:
            if n == "encoder.mask_token":
                continue
            p = self._view(self.flat_params, n)
            p.grad = self._view(self.flat_grads, n)
            yield n, p

    def parameters(self, *a, **k):
        for _, p in self.named_parameters():
            yield p

    def get_num_layers(self):
        return self.depth

    def no_weight_decay(self):
        return {"encoder.pos_embed", "encoder.cls_token"}

    def to(self, device=None, *a, **k):
        if device is not None:
            dev = torch.device(device)
            self.flat_params = self.flat_params.to(dev)
            self.flat_grads = self.flat_grads.to(dev)
            self._ready = False
            self.weights_changed()
        return self

    def _prepare_train(self, dev):
        if self.flat_params.device != dev:
            self.to(dev)
        if self._shadow is None or self._shadow.device != dev:
            self._shadow = torch.empty(self.n_flat, device=dev, dtype=BF16)
        self._enc_pos = _encoder_pos(self.N, self.D).to(dev).contiguous()
        self._pos = _sinusoid(self.n_position, self.d).to(dev).contiguous()
        self._dev = dev

    def _side_stream(self, dev):
        st = getattr(self, "_side", None)
        if st is None or st.device != dev:
            st = self._side = torch.cuda.Stream(device=dev, priority=-1)   # (as the pre-training side stream)
        return st

    def refresh_shadow(self):
        ops.cast_f32_to_bf16(self.flat_params, self._shadow)

    # fused views
    def _fused(self, flat, first, count):
        o, n, s = self._offsets[first]
        return flat[o:o + count * n].view(count * s[0], s[1])

    def weights_changed(self):
        """Whoever writes the flat parameter arena (FineTuneAdamW.step, load_state_dict, ModelEma.update, .to) bumps this counter;
        the eval forward re-syncs its tensors, re-packs the bf16 operands and re-captures the decode HIP graph only when it moved."""
        self._weights_version = getattr(self, "_weights_version", 0) + 1

    def sync_eval_weights(self, force=False):
        """Copy the arena back into the inference path's tensors (used by `.eval()` forward / checkpoints) -- once per weight
        version: `evaluate()` calls the eval forward for every batch, and a sync per call meant a full weight copy, a repack and a
        HIP-graph capture per batch."""
        # writers that go through the parameter views (an in-place op on model.parameters()) do not call weights_changed(): torch's own
        # version counter of the arena catches those
        ver = (getattr(self, "_weights_version", 0), self.flat_params._version)
        if not force and getattr(self, "_synced_version", None) == ver:
            return
        for k in self._offsets:
            self._sd[k] = self._view(self.flat_params, k).detach().clone()
        self._ready = False
        self._synced_version = ver

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        if not self.training:
            self.sync_eval_weights()
            return super().forward(x)
        images, targets, lens = x
        if not images.is_cuda:
            raise RuntimeError("dig_amd.RecModelTrain runs on an MI355X (cuda device) only; there is no CPU fallback")
        if self._dev != images.device or self._shadow is None:
            self._prepare_train(images.device)
        anchor = getattr(self, "_anchor", None)
        if anchor is None or anchor.device != images.device:
            anchor = self._anchor = torch.zeros(1, device=images.device, requires_grad=True)
        logits = _RecTrainFn.apply(anchor, self, images, targets.to(images.device), lens.to(images.device))
        return logits, None, None, None


class ModelEma:
    """timm.utils.ModelEma as run_class_finetuning.py:456-462 / engine_for_finetuning.py:136-139 use it (`--model_ema`): an exponential
    moving average of every state-dict tensor, `ema = decay * ema + (1 - decay) * model` after each optimizer step -- one `dig_ema_update`
    launch over the flat parameter arena.  `.ema` is a model object of the same class whose parameters are the averaged ones (for
    `evaluate(data_loader, model_ema.ema, ...)` and for the `model_ema` entry of checkpoints)."""

    def __init__(self, model, decay=0.9999, device='', resume=''):
        import copy
        if resume:
            raise NotImplementedError("ModelEma(resume=...) is not built: load the checkpoint's 'model_ema' into .ema.load_state_dict")
        self.decay = float(decay)
        self.ema = copy.copy(model)                                            # shares configuration, owns its arenas
        self.ema.flat_params = model.flat_params.detach().clone()
        self.ema.flat_grads = torch.zeros_like(model.flat_grads)
        self.ema._shadow = None
        self.ema._side = None
        self.ema.comm = None
        self.ema._ready = False
        self.ema._graphs = {}                                                  # (its own HIP-graph cache: the copy above is shallow)
        self.ema._sd = OrderedDict((k, v.clone()) for k, v in model._sd.items())
        self.ema._weights_version, self.ema._synced_version = 0, None
        self.ema.train(False)

    def update(self, model):
        e = self.ema
        if e.flat_params.device != model.flat_params.device:
            e.flat_params = e.flat_params.to(model.flat_params.device)
        if e.flat_params.is_cuda:
            ops.ema_update(e.flat_params, model.flat_params, None, e.flat_params.numel(), self.decay)
        else:
            e.flat_params.mul_(self.decay).add_(model.flat_params, alpha=1.0 - self.decay)
        e.weights_changed()


class FlatGradComm:
    """Data parallelism for the fine-tune step: one all-reduce of the flat gradient arena after the backward (the hook
    `NativeScalerWithGradNormCount` calls), parameters broadcast from rank 0 at construction -- what DistributedDataParallel does
    for the reference (run_class_finetuning.py:522-526)."""

    def __init__(self, model, process_group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, process_group
        self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        dist.broadcast(model.flat_params, src=0, group=process_group)
        model.weights_changed()              # the arena was just overwritten: an eval forward must re-sync (and re-capture its HIP graph)
        # every rank draws its own dropout masks (the reference seeds each rank with args.seed + rank, run_class_finetuning.py:262-264)
        model.drop_seed = (int(model.drop_seed) + 0x9E3779B97F4A7C15 * self.rank) & ((1 << 64) - 1)

    def finish_grad_sync(self, model):
        # (the mean: the scaler seeds every backward with loss / world, see parallel.DistComm.finish_grad_sync)
        self.dist.all_reduce(model.flat_grads, group=self.group)


class _TrainStep:
    """One forward / backward of the model; all activations needed by the backward are kept on this object."""

    def __init__(self, model):
        self.m = model

    def p(self, name):
        return self.m._view(self.m.flat_params, name)

    def w(self, name):
        return self.m._view(self.m._shadow, name)

    def g(self, name):
        return self.m._view(self.m.flat_grads, name)

    # ---------------------------------------------------------------- two streams (backward)
    def begin_backward(self, dev):
        """Two HIP streams, as in the pre-training backward (engine_core.encoder_backward): the data-gradient chain (dgrad GEMMs,
        attention / LayerNorm backward) stays on the caller's stream; weight-gradient GEMMs and bias column sums only consume
        (dy, saved activation) pairs and run on a second stream, joined before the optimizer."""
        M = self.m
        self._main = torch.cuda.current_stream(dev)
        self._sd = M._side_stream(dev) if getattr(M, "overlap_streams", True) else self._main
        return self._main, self._sd

    def side(self, fn, *tensors):
        if self._sd is self._main:
            fn()
            return
        self._sd.wait_stream(self._main)
        with torch.cuda.stream(self._sd):
            fn()
        for t in tensors:
            t.record_stream(self._sd)

    def encoder_forward(self, images):
        """PretrainVisionTransformerEncoder.forward (modeling_pretrain_vit.py:89-112, mask=None) on the pre-training hot-path kernels;
        returns the normalised tokens [B*N, D] (bf16) and keeps what `encoder_backward` needs."""
        M = self.m
        dev = images.device
        M.refresh_shadow()
        D, H, N = M.D, M.H, M.N
        B = images.shape[0]
        self.B, self.images = B, images.contiguous().float()
        self.zmask = torch.zeros((B, N), device=dev, dtype=torch.uint8)
        # ---- encoder (PretrainVisionTransformerEncoder.forward_features, mask=None) -- the pre-training hot-path kernels
        x = ops.patch_embed_fwd(self.images, self.p("encoder.patch_embed.proj.weight").view(D, 48), self.p("encoder.patch_embed.proj.bias"),
                                self.zmask, self.p("encoder.mask_token").view(D), M._enc_pos, D, M.gh, M.gw)
        # dropout / drop-path keys of this step (dig_amd/dropout.py); every spec is None when its rate is 0
        plan = self.plan = DR.DropPlan(M.drop_seed, M.drop_step)
        M.drop_step += 1
        pe, pa = M.drop_rate, M.attn_drop_rate
        self.ds_pos = None                              # PretrainVisionTransformerEncoder has no pos_drop (modeling_pretrain_vit.py:89-106)
        self.ds_enc = [dict(attn=plan.spec(DR.enc_site(i, 0), pa),
                            proj=plan.spec(DR.enc_site(i, 1), pe, DR.enc_site(i, 2), M.dpr[i], N),
                            mlp=plan.spec(DR.enc_site(i, 3), pe, DR.enc_site(i, 4), M.dpr[i], N)) for i in range(M.depth)]
        x = ops.dropout_apply(x, self.ds_pos, out=x)
        scale = (D // H) ** -0.5
        self.enc_saved = []
        self.use_chain = use_chain = FT_MLP_CHAIN and M.F <= 2048 and ops.mlp_chain_supported(D, M.F, B * N)
        ln_next = None                                                          # norm1 of the next block, when the fused MLP launch made it
        for i in range(M.depth):
            b = f"encoder.blocks.{i}."
            ds = self.ds_enc[i]
            qo = M._offsets[b + "attn.q_bias"][0]
            qkv_bias = M.flat_params[qo:qo + 3 * D]                             # q_bias | 0 | v_bias (arena layout)
            if ln_next is not None:
                ln1, mu1, rs1 = ln_next
            else:
                ln1, mu1, rs1 = ops.layernorm_fwd(x, self.p(b + "norm1.weight"), self.p(b + "norm1.bias"), 1e-6)
            qkv = ops.linear_fwd(ln1, self.w(b + "attn.qkv.weight"), bias=qkv_bias, alpha=scale, alpha_cols=D)
            ctx, lse = ops.attn_fwd(qkv, B, H, D, drop=ds["attn"])
            # x + drop_path(proj_drop(proj(.))) / x + drop_path(drop(fc2(.))) (modeling_finetune.py:120,59,156-158): GEMM epilogue
            x_mid = ops.linear_fwd(ctx, self.w(b + "attn.proj.weight"), bias=self.p(b + "attn.proj.bias"), resid=x, drop=ds["proj"])
            frozen = i < M.frozen_blocks                                        # no gradient flows into a frozen block: nothing is kept
            ln_next = None
            if use_chain:
                # (the LayerNorm behind it -- the next block's norm1, or the encoder's final norm -- rides along where its statistics are kept)
                nb = "encoder.norm." if i + 1 == M.depth else f"encoder.blocks.{i + 1}.norm1."
                r = ops.mlp_chain_fwd_ln(x_mid, self.p(b + "norm2.weight"), self.p(b + "norm2.bias"), 1e-6, self.w(b + "mlp.fc1.weight"),
                                         self.p(b + "mlp.fc1.bias"), self.w(b + "mlp.fc2.weight"), self.p(b + "mlp.fc2.bias"),
                                         nln_g=None if frozen else self.p(nb + "weight"), nln_b=None if frozen else self.p(nb + "bias"),
                                         save=not frozen, drop=ds["mlp"])
                x_out, ln2, mu2, rs2, pre, act = r["out"], r["ln"], r["ln_mean"], r["ln_rstd"], r["pre"], r["act"]
                if not frozen:
                    ln_next = (r["nln"], r["nln_mean"], r["nln_rstd"])
            else:
                ln2, mu2, rs2 = ops.layernorm_fwd(x_mid, self.p(b + "norm2.weight"), self.p(b + "norm2.bias"), 1e-6)
                pre = None if frozen else torch.empty((B * N, M.F), device=dev, dtype=BF16)
                act = ops.linear_fwd(ln2, self.w(b + "mlp.fc1.weight"), bias=self.p(b + "mlp.fc1.bias"), act=1, pre=pre)
                x_out = ops.linear_fwd(act, self.w(b + "mlp.fc2.weight"), bias=self.p(b + "mlp.fc2.bias"), resid=x_mid, drop=ds["mlp"])
            self.enc_saved.append(None if frozen else (x, ln1, mu1, rs1, qkv, ctx, lse, x_mid, ln2, mu2, rs2, pre, act))
            x = x_out
        if ln_next is not None:
            enc, emu, ers = ln_next
        else:
            enc, emu, ers = ops.layernorm_fwd(x, self.p("encoder.norm.weight"), self.p("encoder.norm.bias"), 1e-6)
        self.enc_last = (x, emu, ers, enc)
        return enc

    # ---------------------------------------------------------------- forward
    def forward(self, images, targets, lens):
        M = self.m
        dev = images.device
        enc = self.encoder_forward(images)
        D, H, N = M.D, M.H, M.n_mem                                           # N: memory tokens per sample the decoder attends over
        B = self.B
        if M.use_1d_attdec:                                                   # model_builder.py:145-148: column means of the 8 x 32 grid
            cols = torch.empty((B * M.gw, D), device=dev, dtype=BF16)
            ops.window_pool_fwd(enc, cols, B, M.gh, M.gw, M.gw, D)
            enc = cols
        self.mem_in = enc
        T, d, nh, dk = M.max_len, M.d, M.nh, M.dk
        hk = nh * dk
        plan = self.plan
        pd = M.decoder_dropout
        self.ds_tgt = plan.spec(DR.DEC_TGT, pd)
        self.ds_dec = [dict(sattn=plan.spec(DR.dec_site(i, 0), pd), sproj=plan.spec(DR.dec_site(i, 1), pd),
                            cattn=plan.spec(DR.dec_site(i, 2), pd), cproj=plan.spec(DR.dec_site(i, 3), pd),
                            act=plan.spec(DR.dec_site(i, 4), pd), out=plan.spec(DR.dec_site(i, 5), pd)) for i in range(M.n_layers)]
        # ---- linear_norm
        h = ops.linear_fwd(enc, self.w("linear_norm.0.weight"), bias=self.p("linear_norm.0.bias"))
        mem, mmu, mrs = ops.layernorm_fwd(h, self.p("linear_norm.1.weight"), self.p("linear_norm.1.bias"), 1e-5)
        self.ln_saved = (h, mmu, mrs, mem)
        # ---- decoder, teacher forcing (decoder.py:196-222)
        bos = torch.full((B, 1), M.start_idx, device=dev, dtype=torch.int64)
        query = torch.cat([bos, targets.long()], dim=-1)[:, :-1].contiguous()
        self.query, self.targets, self.lens = query, targets.long().contiguous(), lens.long().contiguous()
        x = torch.empty((B * T, d), device=dev, dtype=BF16)
        L.call("dig_seq_embed_fwd", L.ptr(query), L.ptr(self.p("decoder.trg_word_emb.weight")), L.ptr(M._pos), L.ptr(x), B, T, d,
               M.nb_classes + 1, L.stream())
        x = ops.dropout_apply(x, self.ds_tgt, out=x)                            # decoder.py:180
        sc = dk ** -0.5
        self.dec_saved = []
        mfma_cross = N == 256 and dk == 64
        kv_ready = []
        if mfma_cross:
            # K|V projections of the encoder memory depend on no decoder state: all layers' are issued on the second stream now and
            # overlap the (small, latency-bound) self-attention kernels of the decoder chain; one event per layer
            main = torch.cuda.current_stream(dev)
            sd = M._side_stream(dev) if getattr(M, "overlap_streams", True) else main
            sd.wait_stream(main)
            with torch.cuda.stream(sd):
                for i in range(M.n_layers):
                    o2, n2, _ = M._offsets[f"decoder.layer_stack.{i}.enc_attn.linear_k.weight"]
                    fused = torch.empty((B * N, 3 * hk), device=dev, dtype=BF16)
                    ops.gemm(mem, M._shadow[o2:o2 + 2 * n2].view(2 * hk, hk), B * N, 2 * hk, d, out=fused[:, hk:], ldc=3 * hk)
                    ev = torch.cuda.Event()
                    ev.record(sd)
                    fused.record_stream(main)
                    kv_ready.append((fused, ev))
            mem.record_stream(sd)
        for i in range(M.n_layers):
            p = f"decoder.layer_stack.{i}."
            ds = self.ds_dec[i]
            h1, m1, r1 = ops.layernorm_fwd(x, self.p(p + "norm1.weight"), self.p(p + "norm1.bias"), 1e-5)
            qkv = ops.linear_fwd(h1, M._fused(M._shadow, p + "self_attn.linear_q.weight", 3))
            a = torch.empty((B * T, hk), device=dev, dtype=BF16)
            lse1 = torch.empty((B, nh, T), device=dev, dtype=F32)
            L.call("dig_seq_attn_fwd_dropout", L.ptr(qkv), 3 * hk, L.ptr(qkv[:, hk:]), 3 * hk, L.ptr(qkv[:, 2 * hk:]), 3 * hk, L.ptr(a), hk,
                   L.ptr(lse1), B, nh, T, T, cf(sc), 1, L.ptr(self.lens), _ref(ds["sattn"]), L.stream())
            x1 = ops.linear_fwd(a, self.w(p + "self_attn.fc.weight"), resid=x, drop=ds["sproj"])
            h2, m2, r2 = ops.layernorm_fwd(x1, self.p(p + "norm2.weight"), self.p(p + "norm2.bias"), 1e-5)
            q2 = ops.linear_fwd(h2, self.w(p + "enc_attn.linear_q.weight"))
            o2, n2, s2 = M._offsets[p + "enc_attn.linear_k.weight"]
            wkv = M._shadow[o2:o2 + 2 * n2].view(2 * hk, hk)
            if mfma_cross:
                # cross-attention on the MFMA kernel of the encoder (256 keys, head dim 64): the T queries of a sample sit in rows
                # [0, T) of a fused q|k|v buffer of 256 rows per sample, and the kernels are told to compute the first
                # ceil(T / 32) query blocks only (rows T..31 are zero queries with a zero output gradient: no contribution).
                fused, ev = kv_ready[i]
                torch.cuda.current_stream(dev).wait_event(ev)
                fq = fused.view(B, N, 3 * hk)[:, :, :hk]
                Tp = (T + 31) // 32 * 32                                          # the kernels work on whole 32-query blocks
                fq[:, T:Tp].zero_()
                fq[:, :T] = (q2 * sc).view(B, T, hk)                              # sc = 2^-3: exact in bf16
                ctx2, lse2 = ops.attn_fwd(fused, B, nh, hk, drop=ds["cattn"], q_rows=T)   # query blocks past T are not computed
                a2 = ctx2.view(B, N, hk)[:, :T].reshape(B * T, hk)
                kvm, lse2 = fused, (lse2, ctx2)
            else:
                kvm = ops.linear_fwd(mem, wkv)
                a2 = torch.empty((B * T, hk), device=dev, dtype=BF16)
                lse2 = torch.empty((B, nh, T), device=dev, dtype=F32)
                L.call("dig_seq_attn_fwd_dropout", L.ptr(q2), hk, L.ptr(kvm), 2 * hk, L.ptr(kvm[:, hk:]), 2 * hk, L.ptr(a2), hk, L.ptr(lse2), B, nh,
                       T, N, cf(sc), 0, None, _ref(ds["cattn"]), L.stream())
            x2 = ops.linear_fwd(a2, self.w(p + "enc_attn.fc.weight"), resid=x1, drop=ds["cproj"])
            h3, m3, r3 = ops.layernorm_fwd(x2, self.p(p + "norm3.weight"), self.p(p + "norm3.bias"), 1e-5)
            pre = torch.empty((B * T, M.d_inner), device=dev, dtype=BF16)
            u = ops.linear_fwd(h3, self.w(p + "mlp.w_1.weight"), bias=self.p(p + "mlp.w_1.bias"), act=1, pre=pre, drop=ds["act"])
            x3 = ops.linear_fwd(u, self.w(p + "mlp.w_2.weight"), bias=self.p(p + "mlp.w_2.bias"), resid=x2, drop=ds["out"])
            self.dec_saved.append((x, h1, m1, r1, qkv, a, lse1, x1, h2, m2, r2, q2, kvm, a2, lse2, x2, h3, m3, r3, pre, u))
            x = x3
        o, fm, fr = ops.layernorm_fwd(x, self.p("decoder.layer_norm.weight"), self.p("decoder.layer_norm.bias"), 1e-6)
        self.fin_saved = (x, fm, fr, o)
        C = M.nb_classes
        self.cls_w = torch.zeros((CLS_PAD, d), device=dev, dtype=BF16)
        self.cls_w[:C] = self.w("decoder.classifier.weight")
        cb = torch.zeros(CLS_PAD, device=dev, dtype=F32)
        cb[:C] = self.p("decoder.classifier.bias")
        logits = torch.empty((B * T, CLS_PAD), device=dev, dtype=F32)
        ops.gemm(o, self.cls_w, B * T, CLS_PAD, d, out=logits, out_kind=ops.OUT_F32, bias=cb)
        return logits[:, :C].reshape(B, T, C)

    def encoder_backward(self, denc):
        """denc: bf16 [B*N, D] gradient w.r.t. the tokens `encoder_forward` returned (call `begin_backward` first)."""
        M = self.m
        dev = denc.device
        D, H, N = M.D, M.H, M.N
        x_last, emu, ers, enc = self.enc_last
        dx = ops.layernorm_bwd(denc, x_last, self.p("encoder.norm.weight"), self.p("encoder.norm.bias"), emu, ers, None, self.g("encoder.norm.weight"),
                               self.g("encoder.norm.bias"))
        # ---- encoder (same chain as the pre-training backward, one view, no masking)
        scale = (D // H) ** -0.5
        # the four weight gradients of a block as ONE grouped launch (csrc/wgrad.hip), folded into the gradient arena by the next block's
        # launch / the flush below -- as in the pre-training backward (engine_core.encoder_backward)
        Rg = self.B * N
        grouped = (ops.WGRAD_GROUP and not FT_BATCH_REDUCE and
                   all(ops.wgrad_group_route(o, i_, Rg) is not None for o, i_ in ((M.F, D), (D, M.F), (3 * D, D), (D, D))))
        grp = ops.WgradGroup(dev) if grouped else None
        wT = None
        if getattr(self, "use_chain", False) and M.frozen_blocks < M.depth:
            # K-contiguous copies of the MLP weights for the fused backward (one launch per weight shape, 1.2 MB per matrix)
            blocks = [f"encoder.blocks.{i}." for i in range(M.frozen_blocks, M.depth)]
            wT = dict(zip(blocks, zip(ops.transpose_bf16_multi([self.w(b_ + "mlp.fc2.weight") for b_ in blocks]),
                                      ops.transpose_bf16_multi([self.w(b_ + "mlp.fc1.weight") for b_ in blocks]))))
        for i in reversed(range(M.frozen_blocks, M.depth)):                   # (frozen blocks are a prefix: the chain stops above them)
            b = f"encoder.blocks.{i}."
            x, ln1, mu1, rs1, qkv, ctx, lse, x_mid, ln2, mu2, rs2, pre, act = self.enc_saved[i]
            self.enc_saved[i] = None
            ds = self.ds_enc[i]
            # a dropped branch (dropout and/or drop-path) back-propagates the residual gradient under the same mask; its bias
            # gradient is then the column sum of the MASKED gradient, so the LayerNorm kernel's fused residual column sum is off
            # the block's reductions (slab sums of its four weight gradients, bias / LayerNorm-parameter column sums) go out as two
            # launches after its last weight-gradient GEMM (ops.GradReduceBatch), as in the pre-training backward
            red = ops.GradReduceBatch() if FT_BATCH_REDUCE else None
            wg = red.wgrad if red else ops.linear_wgrad
            csum = red.colsum_partials if red else ops.colsum_partials
            held = []
            if grp:
                def side_wg(dy_, x_, dw_):
                    if not grp.add(dy_, x_, dw_):                    # (never inside an assert: python -O would drop the weight gradient)
                        raise RuntimeError("grouped weight gradient: a problem of this block does not fit the group's plan")
                    held.extend((dy_, x_))
            else:
                def side_wg(dy_, x_, dw_):
                    self.side(lambda: wg(dy_, x_, dw_), dy_, x_)
            dz = ops.dropout_apply(dx, ds["mlp"])
            if ds["mlp"] is not None:
                self.side(lambda: ops.colsum(dz, self.g(b + "mlp.fc2.bias")), dz)
            side_wg(dz, act, self.g(b + "mlp.fc2.weight"))
            if wT is not None:                                                  # both data gradients of the MLP in one launch
                dln2, dact, bparts = ops.mlp_chain_bwd(dz, wT[b][0], pre, wT[b][1])
            else:
                dact, bparts = ops.linear_dgrad(dz, self.w(b + "mlp.fc2.weight"), gelu_pre=pre, colsum=True)
                dln2 = None
            self.side(lambda: csum(bparts, self.g(b + "mlp.fc1.bias")), bparts)
            side_wg(dact, ln2, self.g(b + "mlp.fc1.weight"))
            if dln2 is None:
                dln2 = ops.linear_dgrad(dact, self.w(b + "mlp.fc1.weight"))
            dx_mid, fin2, ws2 = ops.layernorm_bwd(dln2, x_mid, self.p(b + "norm2.weight"), self.p(b + "norm2.bias"), mu2, rs2, dx,
                                                  self.g(b + "norm2.weight"), self.g(b + "norm2.bias"), out=dln2,
                                                  dres_colsum=self.g(b + "mlp.fc2.bias") if ds["mlp"] is None else None, defer=True)
            if red:                                                                    # parameter-gradient reduction: off the chain
                red.layernorm_finalize(ws2, x_mid.shape[0], D, self.g(b + "norm2.weight"), self.g(b + "norm2.bias"),
                                       self.g(b + "mlp.fc2.bias") if ds["mlp"] is None else None)
            else:
                self.side(fin2, ws2)
            dz = ops.dropout_apply(dx_mid, ds["proj"])
            if ds["proj"] is not None:
                self.side(lambda: ops.colsum(dz, self.g(b + "attn.proj.bias")), dz)
            side_wg(dz, ctx, self.g(b + "attn.proj.weight"))
            dctx = ops.linear_dgrad(dz, self.w(b + "attn.proj.weight"))
            # q_bias / v_bias gradients leave the attention kernel as per-image partial sums (no pass over the 150 MB dqkv)
            if FT_FUSED_QV:
                dqkv, qs, vs = ops.attn_bwd(qkv, ctx, dctx, lse, self.B, H, D, scale, drop=ds["attn"], bias_sums=True)
                side_wg(dqkv, ln1, self.g(b + "attn.qkv.weight"))
                if grp:
                    self.side(grp.launch, *held)
                self.side(lambda: (csum(qs, self.g(b + "attn.q_bias")), csum(vs, self.g(b + "attn.v_bias"))), qs, vs)
            else:
                dqkv = ops.attn_bwd(qkv, ctx, dctx, lse, self.B, H, D, scale, drop=ds["attn"])
                side_wg(dqkv, ln1, self.g(b + "attn.qkv.weight"))
                if grp:
                    self.side(grp.launch, *held)
                self.side(lambda: ops.colsum(dqkv, self.g(b + "attn.q_bias"), cols=D), dqkv)
                self.side(lambda: ops.colsum(dqkv[:, 2 * D:], self.g(b + "attn.v_bias"), cols=D))
            dln1 = ops.linear_dgrad(dqkv, self.w(b + "attn.qkv.weight"), out=dctx)
            dx, fin1, ws1 = ops.layernorm_bwd(dln1, x, self.p(b + "norm1.weight"), self.p(b + "norm1.bias"), mu1, rs1, dx_mid,
                                              self.g(b + "norm1.weight"), self.g(b + "norm1.bias"), out=dln1,
                                              dres_colsum=self.g(b + "attn.proj.bias") if ds["proj"] is None else None, defer=True)
            if red:
                red.layernorm_finalize(ws1, x.shape[0], D, self.g(b + "norm1.weight"), self.g(b + "norm1.bias"),
                                       self.g(b + "attn.proj.bias") if ds["proj"] is None else None)
                self.side(red.flush, *red.tensors())
            else:
                self.side(fin1, ws1)
        if grp:
            self.side(grp.flush)                                                # the last block's slabs: folded by a fold-only launch
        if "encoder.patch_embed.proj.weight" in M.frozen:
            return
        dx = ops.dropout_apply(dx, self.ds_pos, out=dx)
        gtok = torch.zeros(D, device=dev, dtype=F32)                          # mask_token takes no part at fine-tune: gradient discarded
        ops.patch_embed_bwd_mfma(dx, self.images, self.zmask, self.g("encoder.patch_embed.proj.weight").view(D, 48),
                                 self.g("encoder.patch_embed.proj.bias"), gtok, D, M.gh, M.gw)

    # ---------------------------------------------------------------- backward
    def backward(self, dlogits_btc):
        """dlogits_btc: fp32 [B, T, C] gradient w.r.t. the returned logits."""
        M = self.m
        dev = dlogits_btc.device
        main, sd = self.begin_backward(dev)
        side = self.side
        B, T, d, nh, dk, C, N, D, H = self.B, M.max_len, M.d, M.nh, M.dk, M.nb_classes, M.n_mem, M.D, M.H
        hk = nh * dk
        rows = B * T
        dl = torch.zeros((rows, CLS_PAD), device=dev, dtype=BF16)
        dl[:, :C] = dlogits_btc.reshape(rows, C).to(BF16)
        x, fm, fr, o = self.fin_saved
        # classifier
        side(lambda: ops.wgrad(dl, o, self.g("decoder.classifier.weight"), C, d, rows), dl, o)
        cs = torch.zeros(CLS_PAD, device=dev, dtype=F32)
        side(lambda: (ops.colsum(dl, cs, cols=CLS_PAD), self.g("decoder.classifier.bias").add_(cs[:C])), dl, cs)
        do = ops.gemm(dl, self.cls_w, rows, d, CLS_PAD, tb=True)
        dx = ops.layernorm_bwd(do, x, self.p("decoder.layer_norm.weight"), self.p("decoder.layer_norm.bias"), fm, fr, None,
                               self.g("decoder.layer_norm.weight"), self.g("decoder.layer_norm.bias"))
        self._dmem = None
        sc = dk ** -0.5
        mem = self.ln_saved[3]
        for i in reversed(range(M.n_layers)):
            p = f"decoder.layer_stack.{i}."
            (x0, h1, m1, r1, qkv, a, lse1, x1, h2, m2, r2, q2, kvm, a2, lse2, x2, h3, m3, r3, pre, u) = self.dec_saved[i]
            self.dec_saved[i] = None
            ds = self.ds_dec[i]
            # feed-forward (every dropped branch: its gradient is the residual gradient under the same mask)
            dz = ops.dropout_apply(dx, ds["out"])
            side(lambda: ops.linear_wgrad(dz, u, self.g(p + "mlp.w_2.weight")), dz, u)
            side(lambda: ops.colsum(dz, self.g(p + "mlp.w_2.bias")), dz)
            du, bparts = ops.linear_dgrad(dz, self.w(p + "mlp.w_2.weight"), gelu_pre=pre, colsum=True, drop=ds["act"])
            side(lambda: ops.colsum_partials(bparts, self.g(p + "mlp.w_1.bias")), bparts)
            side(lambda: ops.linear_wgrad(du, h3, self.g(p + "mlp.w_1.weight")), du, h3)
            dh3 = ops.linear_dgrad(du, self.w(p + "mlp.w_1.weight"))
            dx2 = ops.layernorm_bwd(dh3, x2, self.p(p + "norm3.weight"), self.p(p + "norm3.bias"), m3, r3, dx, self.g(p + "norm3.weight"),
                                    self.g(p + "norm3.bias"))
            # cross-attention over the encoder memory
            dz = ops.dropout_apply(dx2, ds["cproj"])
            side(lambda: ops.linear_wgrad(dz, a2, self.g(p + "enc_attn.fc.weight")), dz, a2)
            da2 = ops.linear_dgrad(dz, self.w(p + "enc_attn.fc.weight"))
            o2, n2, _ = M._offsets[p + "enc_attn.linear_k.weight"]
            if isinstance(lse2, tuple):                                           # MFMA path (see forward)
                lse2, ctx2 = lse2                                                 # padded rows: finite outputs, zero dO -> delta = 0
                dctx2 = torch.empty((B * N, hk), device=dev, dtype=BF16)
                dv_ = dctx2.view(B, N, hk)
                dv_[:, T:(T + 31) // 32 * 32].zero_()
                dv_[:, :T] = da2.view(B, T, hk)
                dfused = ops.attn_bwd(kvm, ctx2, dctx2, lse2, B, nh, hk, sc, drop=ds["cattn"], q_rows=T)   # kvm = the fused q|k|v buffer
                dq2 = dfused.view(B, N, 3 * hk)[:, :T, :hk].reshape(B * T, hk)
                dkvm = dfused[:, hk:]                                             # [B*N, 2hk] view, row stride 3hk
            else:
                dq2 = torch.empty_like(q2)
                dkvm = torch.empty_like(kvm)
                L.call("dig_seq_attn_bwd_dropout", L.ptr(q2), hk, L.ptr(kvm), 2 * hk, L.ptr(kvm[:, hk:]), 2 * hk, L.ptr(da2), hk, L.ptr(lse2),
                       L.ptr(dq2), hk, L.ptr(dkvm), 2 * hk, L.ptr(dkvm[:, hk:]), 2 * hk, B, nh, T, N, cf(sc), 0, None, _ref(ds["cattn"]),
                       L.stream())
            side(lambda: ops.linear_wgrad(dq2, h2, self.g(p + "enc_attn.linear_q.weight")), dq2, h2)
            dh2 = ops.linear_dgrad(dq2, self.w(p + "enc_attn.linear_q.weight"))
            side(lambda: ops.wgrad(dkvm, mem, M.flat_grads[o2:o2 + 2 * n2].view(2 * hk, hk), 2 * hk, hk, B * N), dkvm, mem)
            # the gradient w.r.t. the encoder memory is needed only after the decoder loop: its GEMMs run on the second stream too
            def mem_grad(dkvm=dkvm, o2=o2, n2=n2):
                dm = ops.gemm(dkvm, M._shadow[o2:o2 + 2 * n2].view(2 * hk, hk), B * N, hk, 2 * hk, tb=True)
                if self._dmem is None:
                    self._dmem = dm
                else:
                    ops.add_bf16(self._dmem, dm, self._dmem)
            side(mem_grad, dkvm)
            dx1 = ops.layernorm_bwd(dh2, x1, self.p(p + "norm2.weight"), self.p(p + "norm2.bias"), m2, r2, dx2, self.g(p + "norm2.weight"),
                                    self.g(p + "norm2.bias"))
            # masked self-attention
            dz = ops.dropout_apply(dx1, ds["sproj"])
            side(lambda: ops.linear_wgrad(dz, a, self.g(p + "self_attn.fc.weight")), dz, a)
            da = ops.linear_dgrad(dz, self.w(p + "self_attn.fc.weight"))
            dqkv = torch.empty_like(qkv)
            L.call("dig_seq_attn_bwd_dropout", L.ptr(qkv), 3 * hk, L.ptr(qkv[:, hk:]), 3 * hk, L.ptr(qkv[:, 2 * hk:]), 3 * hk, L.ptr(da), hk,
                   L.ptr(lse1), L.ptr(dqkv), 3 * hk, L.ptr(dqkv[:, hk:]), 3 * hk, L.ptr(dqkv[:, 2 * hk:]), 3 * hk, B, nh, T, T, cf(sc), 1,
                   L.ptr(self.lens), _ref(ds["sattn"]), L.stream())
            side(lambda: ops.linear_wgrad(dqkv, h1, M._fused(M.flat_grads, p + "self_attn.linear_q.weight", 3)), dqkv, h1)
            dh1 = ops.linear_dgrad(dqkv, M._fused(M._shadow, p + "self_attn.linear_q.weight", 3))
            dx = ops.layernorm_bwd(dh1, x0, self.p(p + "norm1.weight"), self.p(p + "norm1.bias"), m1, r1, dx1, self.g(p + "norm1.weight"),
                                   self.g(p + "norm1.bias"))
        dx = ops.dropout_apply(dx, self.ds_tgt, out=dx)
        L.call("dig_seq_embed_bwd_lens", L.ptr(self.query), L.ptr(dx), L.ptr(self.g("decoder.trg_word_emb.weight")), rows, d, C + 1, T,
               L.ptr(self.lens), L.stream())
        # ---- linear_norm
        h, mmu, mrs, _ = self.ln_saved
        enc = self.mem_in
        main.wait_stream(sd)                                                    # the memory gradient was summed on the second stream
        dmem, self._dmem = self._dmem, None
        dmem.record_stream(main)
        dh = ops.layernorm_bwd(dmem, h, self.p("linear_norm.1.weight"), self.p("linear_norm.1.bias"), mmu, mrs, None, self.g("linear_norm.1.weight"),
                               self.g("linear_norm.1.bias"))
        side(lambda: ops.linear_wgrad(dh, enc, self.g("linear_norm.0.weight")), dh, enc)
        side(lambda: ops.colsum(dh, self.g("linear_norm.0.bias")), dh)
        denc = ops.linear_dgrad(dh, self.w("linear_norm.0.weight"))
        if M.use_1d_attdec:                                                     # every token of a column gets 1/gh of the column's gradient
            dfull = torch.empty((B * M.N, D), device=dev, dtype=BF16)
            ops.window_pool_bwd(denc, dfull, B, M.gh, M.gw, M.gw, D, False)
            denc = dfull
        self.encoder_backward(denc)
        main.wait_stream(sd)


RecModelTrain._step_cls = _TrainStep


class _RecTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, images, targets, lens):
        step = model._step_cls(model)
        logits = step.forward(images, targets, lens)
        ctx.step = step
        return logits

    @staticmethod
    def backward(ctx, g):
        step, ctx.step = ctx.step, None
        step.backward(g.contiguous().float())
        return None, None, None, None, None


class SeqCrossEntropyLoss(torch.nn.Module):
    """loss/seqCrossEntropyLoss.py (sample_normalize) with its gradient: forward(logits [B,T,C] fp32, target [B,T], length [B])."""

    def forward(self, input, target, length):
        return _SeqCEFn.apply(input, target.to(input.device).long().contiguous(), length.to(input.device).long().contiguous())


class _SeqCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, length):
        B, T, C = logits.shape
        x = logits.detach().float().contiguous()
        rows = torch.empty(B * T, device=x.device, dtype=F32)
        loss = torch.empty(1, device=x.device, dtype=F32)
        L.call("dig_seq_cross_entropy", L.ptr(x), L.ptr(target), L.ptr(length), B, T, C, L.ptr(rows), L.ptr(loss), L.stream())
        ctx.save_for_backward(x, target, length)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        x, target, length = ctx.saved_tensors
        B, T, C = x.shape
        Cp = (C + 7) // 8 * 8
        dl = torch.empty((B * T, Cp), device=x.device, dtype=BF16)
        gs = g.reshape(1).float().contiguous()                         # (named: a converted copy must outlive the call)
        L.call("dig_seq_cross_entropy_bwd", L.ptr(x), C, L.ptr(target), L.ptr(length), L.ptr(gs), B, T, C, L.ptr(dl), Cp, L.stream())
        return dl[:, :C].float().reshape(B, T, C), None, None


class SeqLabelSmoothingCrossEntropyLoss(torch.nn.Module):
    """loss/seqLabelSmoothingCrossEntropyLoss.py (sample_normalize), the criterion `--smoothing > 0` selects
    (run_class_finetuning.py:538-541), with the value the reference really computes: its smoothing term broadcasts to a [BT, BT]
    matrix (include/dig_hip.h `dig_seq_ls_cross_entropy`), which this class reproduces."""

    def __init__(self, smoothing=0.1):
        super().__init__()
        if not 0.0 <= smoothing <= 1.0:
            raise ValueError("smoothing must be in [0, 1]")
        self.smoothing = float(smoothing)

    def forward(self, input, target, length):
        return _SeqLSCEFn.apply(input, target.to(input.device).long().contiguous(), length.to(input.device).long().contiguous(), self.smoothing)


class _SeqLSCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, length, smoothing):
        B, T, C = logits.shape
        x = logits.detach().float().contiguous()
        rows = torch.empty(2 * B * T, device=x.device, dtype=F32)
        loss = torch.empty(1, device=x.device, dtype=F32)
        L.call("dig_seq_ls_cross_entropy", L.ptr(x), L.ptr(target), L.ptr(length), B, T, C, cf(smoothing), L.ptr(rows), L.ptr(loss), L.stream())
        ctx.save_for_backward(x, target, length)
        ctx.smoothing = smoothing
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        x, target, length = ctx.saved_tensors
        B, T, C = x.shape
        Cp = (C + 7) // 8 * 8
        dl = torch.empty((B * T, Cp), device=x.device, dtype=BF16)
        gs = g.reshape(1).float().contiguous()
        L.call("dig_seq_ls_cross_entropy_bwd", L.ptr(x), C, L.ptr(target), L.ptr(length), L.ptr(gs), B, T, C, cf(ctx.smoothing), L.ptr(dl), Cp,
               L.stream())
        return dl[:, :C].float().reshape(B, T, C), None, None, None


# -------------------------------------------------------------------------------------------------- optimizer
def get_num_layer_for_vit(var_name, num_max_layer):
    """optim_factory.py:33-45."""
    if var_name in ("cls_token", "mask_token", "pos_embed") or var_name.startswith("patch_embed"):
        return 0
    if var_name.startswith("rel_pos_bias"):
        return num_max_layer - 1
    if var_name.startswith("blocks"):
        return int(var_name.split('.')[1]) + 1
    return num_max_layer - 1


class LayerDecayValueAssigner:
    """optim_factory.py:48-57."""

    def __init__(self, values):
        self.values = values

    def get_scale(self, layer_id):
        return self.values[layer_id]

    def get_layer_id(self, var_name):
        return get_num_layer_for_vit(var_name, len(self.values))


class FineTuneAdamW:
    """custom_optim AdamW over `RecModelTrain`'s arena with the reference's parameter groups (get_parameter_groups,
    optim_factory.py:57-100): `param_groups` carries lr / weight_decay / lr_scale per group exactly as the reference's list does (the
    engine rewrites lr = schedule * lr_scale and weight_decay every step); the update itself is one launch."""

    def __init__(self, model, lr, weight_decay, betas=(0.9, 0.999), eps=1e-8, get_num_layer=None, get_layer_scale=None, skip_list=()):
        self.model = model
        groups = OrderedDict()
        self._name_group = {}
        # every registered parameter in registration order -- `encoder.mask_token` included: the reference's optimizer lists it
        # (requires_grad) although the fine-tune forward never gives it a gradient, so it holds an index but no state
        self._stateless = set()
        for name in model._offsets:
            if name in getattr(model, "frozen", ()):
                continue                                                    # requires_grad = False: not listed (optim_factory.py:66-67)
            p = model._view(model.flat_params, name)
            if name == "encoder.mask_token":
                self._stateless.add(name)
            if p.ndim == 1 or name.endswith(".bias") or name in skip_list:
                gname, wd = "no_decay", 0.0
            else:
                gname, wd = "decay", weight_decay
            if get_num_layer is not None:
                lid = get_num_layer(name.replace('encoder.', '') if name.startswith('encoder') else name)
                gname = "layer_%d_%s" % (lid, gname)
            else:
                lid = None
            if gname not in groups:
                groups[gname] = {"weight_decay": wd, "names": [], "params": [], "lr_scale": get_layer_scale(lid) if get_layer_scale is not None else 1.0,
                                 "lr": lr, "betas": betas, "eps": eps}
            groups[gname]["names"].append(name)
            groups[gname]["params"].append(p)
            self._name_group[name] = gname
        self.param_groups = list(groups.values())
        self._gnames = list(groups.keys())
        self._step = 0
        self.exp_avg = self.exp_avg_sq = None
        self._tab_dev = None

    def zero_grad(self, set_to_none=False):
        ops.fill_f32(self.model.flat_grads, 0.0) if self.model.flat_grads.is_cuda else self.model.flat_grads.zero_()

    def _tables(self):
        M = self.model
        dev = M.flat_params.device
        if self._tab_dev == dev:
            return
        idx = torch.full((M.n_flat // 256,), 255, dtype=torch.uint8)          # 255 = no gradient (mask_token, padding)
        for gi, g in enumerate(self.param_groups):
            for n in g["names"]:
                if n in self._stateless:
                    continue                                                    # no gradient: the update leaves it alone (index 255)
                o, cnt, _ = M._offsets[n]
                idx[o // 256:(o + cnt + 255) // 256] = gi                      # (a v_bias may start inside a granule: see the arena layout)
        self._idx = idx.to(dev)
        pin = (lambda t: t.pin_memory()) if dev.type == "cuda" else (lambda t: t)
        self._host_ring = [pin(torch.empty(2, 256, dtype=F32)) for _ in range(8)]   # the host may run steps ahead
        self._dev_tab = torch.empty(2, 256, device=dev, dtype=F32)
        self.exp_avg = torch.zeros(M.n_flat, device=dev, dtype=F32)
        self.exp_avg_sq = torch.zeros(M.n_flat, device=dev, dtype=F32)
        self._tab_dev = dev

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0, finite_gate=None):
        self._tables()
        M = self.model
        ng = len(self.param_groups)
        assert ng < 255
        host = self._host_ring[self._step % len(self._host_ring)]
        for gi, g in enumerate(self.param_groups):                              # whatever the engine wrote into the groups this step
            host[0, gi] = float(g["lr"])
            host[1, gi] = float(g["weight_decay"])
        self._dev_tab.copy_(host, non_blocking=True)
        g0 = self.param_groups[0]
        self._step += 1
        L.call("dig_adamw_step_groups", L.ptr(M.flat_params), L.ptr(M.flat_grads), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq), None,
               ctypes.c_longlong(M.n_flat), L.ptr(self._idx), L.ptr(self._dev_tab[0]), L.ptr(self._dev_tab[1]), cf(g0["betas"][0]),
               cf(g0["betas"][1]), cf(g0["eps"]), self._step, cf(grad_scale), L.ptr(finite_gate), L.stream())
        M.weights_changed()


    # ---- checkpoints: torch.optim's per-parameter layout (what utils.save_model / auto_load_model exchange, utils/utils.py:546-651)
    def _ordered_names(self):
        return [n for g in self.param_groups for n in g["names"]]

    def state_dict(self):
        M = self.model
        state = {}
        if self._step > 0:
            for i, n in enumerate(self._ordered_names()):
                if n not in self._stateless:
                    state[i] = {"step": self._step, "exp_avg": M._view(self.exp_avg, n), "exp_avg_sq": M._view(self.exp_avg_sq, n)}
        groups, k = [], 0
        for g in self.param_groups:
            d = {key: v for key, v in g.items() if key not in ("params", "names")}
            d.setdefault("amsgrad", False)
            d["params"] = list(range(k, k + len(g["names"])))
            k += len(g["names"])
            groups.append(d)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        self._tables()
        M = self.model
        names = self._ordered_names()
        if [len(g["params"]) for g in sd["param_groups"]] != [len(g["names"]) for g in self.param_groups]:
            raise ValueError("loaded state dict has different parameter groups")
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = set()
        for i, st in sd["state"].items():
            n = names[int(i)]
            if tuple(st["exp_avg"].shape) != tuple(M._offsets[n][2]):
                raise ValueError(f"optimizer state {i} ({n}): shape {tuple(st['exp_avg'].shape)} != {M._offsets[n][2]}")
            M._view(self.exp_avg, n).copy_(st["exp_avg"])
            M._view(self.exp_avg_sq, n).copy_(st["exp_avg_sq"])
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ; the fused optimizer keeps one step counter")
        self._step = steps.pop() if steps else 0
        for g, lg in zip(self.param_groups, sd["param_groups"]):
            for key in ("lr", "weight_decay", "lr_scale", "betas", "eps"):
                if key in lg:
                    g[key] = lg[key]


def create_optimizer(args, model, get_num_layer=None, get_layer_scale=None, filter_bias_and_bn=True, skip_list=None):
    """optim_factory.create_optimizer for `--opt adamw` on a RecModelTrain."""
    if args.opt.lower() != "adamw":
        raise NotImplementedError("only --opt adamw is built")
    skip = skip_list if skip_list is not None else (model.no_weight_decay() if hasattr(model, "no_weight_decay") else ())
    betas = tuple(args.opt_betas) if getattr(args, "opt_betas", None) else (0.9, 0.999)
    return FineTuneAdamW(model, args.lr, args.weight_decay, betas, getattr(args, "opt_eps", 1e-8) or 1e-8, get_num_layer, get_layer_scale, skip)
