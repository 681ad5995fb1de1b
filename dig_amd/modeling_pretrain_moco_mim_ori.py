"""MoCo_ViT for MI355X: the reference's SimMIM + MoCo-v3 pre-training model
(reference: modeling_pretrain_moco_mim_ori.py:261-577, factories :627-871; encoder modeling_pretrain_vit.py:27-111;
blocks modeling_finetune.py:43-210) re-designed around flat HBM arenas and hand-written HIP kernels.

What is kept from the reference (the drop-in surface):
  * factory names (`pretrain_simmim_moco_ori_vit_{tiny,small,base}_patch4_32x128`) and their kwargs;
  * `forward(image, aug_image, vis_mask_pos, m, only_mim_on_ori_img) -> dict` with keys
    contra_loss, q1_acc1, q1_acc5, q2_acc1, q2_acc5, vis_out;
  * `named_parameters()` / `state_dict()` key names, shapes and order (356 params + 42 BN buffers for ViT-S);
  * `.encoder.patch_embed.patch_size`, `.no_weight_decay()`, `.train()`, `.to(device)`.

What is different (MI355X-first):
  * all trainable parameters live in ONE fp32 arena (+ one fp32 grad arena, + a bf16 shadow the MFMA GEMMs read);
    the momentum (EMA) copies live in a second arena with the identical layout, so the EMA update, AdamW and the
    gradient all-reduce are single flat kernels / collectives instead of 173..183 per-tensor launches;
  * the whole forward+backward is one autograd node whose backward is written by hand (no autograd graph over ops):
    every device operation is a kernel of libdig_hip.so launched through the C-ABI (include/dig_hip.h);
  * activations are bf16, statistics / losses / optimizer state are fp32.
There is no CPU or ATen fallback: without libdig_hip.so (or without a GPU) forward raises.
"""
import math
from collections import OrderedDict
from functools import partial

import numpy as np
import os

import torch
import torch.nn as nn

from . import ops
from .registry import register_model

BF16, F32 = torch.bfloat16, torch.float32
ALIGN = 256  # arena granule (elements): every parameter starts on a 1 KiB boundary


def _round_up(x, a):
    return (x + a - 1) // a * a


def get_sinusoid_encoding_table(n_position, d_hid):
    """Same values as modeling_finetune.py:200-210 (float64 math, cast to fp32), vectorised."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    ang = pos / np.power(10000.0, 2.0 * (j // 2) / d_hid)[None, :]
    tab = np.empty_like(ang)
    tab[:, 0::2] = np.sin(ang[:, 0::2])
    tab[:, 1::2] = np.cos(ang[:, 1::2])
    return torch.from_numpy(tab).to(torch.float32).unsqueeze(0)


class _Node(nn.Module):
    """Bare container used to reproduce the reference's module tree (and therefore its state_dict keys)."""


def _node_for(root, dotted):
    mod = root
    parts = dotted.split(".")
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    return mod, parts[-1]


def _mlp_dims(n_layers, din, dmid, dout):
    return [(din if l == 0 else dmid, dout if l == n_layers - 1 else dmid) for l in range(n_layers)]


class ParamSpec:
    __slots__ = ("name", "shape", "numel", "offset", "group", "arena", "bundle_off")

    def __init__(self, name, shape, group, arena):
        self.name, self.shape, self.group, self.arena = name, tuple(shape), group, arena
        self.numel = int(np.prod(shape))
        self.offset = -1


class MoCo_ViT(nn.Module):
    def __init__(self, img_size=(32, 128), patch_size=4, in_chans=3, encoder_num_classes=0, encoder_embed_dim=384,
                 encoder_depth=12, encoder_num_heads=6, decoder_num_classes=48, decoder_embed_dim=192, decoder_depth=4,
                 decoder_num_heads=3, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer=None, init_values=0., use_learnable_pos_emb=False, num_classes=0,
                 mlp_dim=4096, dim=256, T=1.0, num_windows=5, use_pixel_target=False, use_moco_target=True,
                 encoder_type='vit', queue_size=65536, patchnet_name='regular', label_smoothing=0., use_pix_projector=True,
                 device=None, **unused):
        super().__init__()
        if not (use_pixel_target or use_moco_target):
            raise ValueError("MoCo_ViT needs at least one objective (use_pixel_target / use_moco_target)")
        if use_pixel_target and use_moco_target and not use_pix_projector:
            raise NotImplementedError("use_pix_projector=False (no factory of the reference sets it) is not built")
        if patchnet_name not in ('no_patchtrans', 'regular', 'conv') and use_moco_target:
            raise NotImplementedError(f"patchnet_name={patchnet_name!r}: 'regular', 'no_patchtrans' or 'conv' (modeling_pretrain_moco_mim_ori.py:371-382)")
        if drop_rate or attn_drop_rate or init_values or use_learnable_pos_emb or label_smoothing:
            raise NotImplementedError("dropout / layer-scale / learnable pos-emb / label smoothing are 0 in pre-training (no flag of "
                                      "run_mae_pretraining_moco.py sets them)")
        if not qkv_bias or patch_size != 4 or in_chans != 3:
            raise NotImplementedError("qkv_bias=True, patch 4, RGB only")
        D, H = encoder_embed_dim, encoder_num_heads
        if D % H or D // H != 64:
            raise NotImplementedError("attention kernel is specialised for head_dim 64")
        if tuple(img_size) != (32, 128):
            raise NotImplementedError("attention kernel is specialised for 8x32 = 256 tokens (32x128 crops)")
        self.D, self.H, self.depth = D, H, encoder_depth
        # --drop_path (run_mae_pretraining_moco.py:87,283): stochastic depth, rate linspace(0, rate, depth)[i] on both branches of block i of
        # BOTH encoders (modeling_pretrain_vit.py:50-56; the momentum encoder runs in train mode too).  Masks: keyed counter hash per (step,
        # site) as in the fine-tune step (dig_amd/dropout.py); drop_seed / drop_step travel in checkpoints (utils.save_model).
        self.drop_path_rate = float(drop_path_rate)
        self.dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, encoder_depth)]
        self.drop_seed = torch.initial_seed()
        self.drop_step = 0
        self.F = int(D * mlp_ratio)
        self.gh, self.gw = img_size[0] // patch_size, img_size[1] // patch_size
        self.N = self.gh * self.gw
        self.T = T
        self.num_windows = num_windows
        if not 1 <= num_windows <= img_size[1] // patch_size:
            raise ValueError(f"num_windows={num_windows}: 1 .. {img_size[1] // patch_size} (the token grid's columns)")
        # (uneven windows -- the argparse default of 5 on a 32-column grid -- pool with adaptive_avg_pool2d's overlapping bins: dig_window_pool_*)
        self.dec_dim, self.dec_classes = decoder_embed_dim, decoder_num_classes
        self.moco_dim, self.moco_mlp_dim = dim, mlp_dim
        self.ln_eps, self.bn_eps, self.bn_momentum = 1e-6, 1e-5, 0.1
        # which objectives the model carries (modeling_pretrain_moco_mim_ori.py:340-426): both = the `simmim_moco_ori` factories; Dis-only
        # (`pretrain_moco_ori_*`: no mask, no pix_projector, no decoder); Gen-only (`pretrain_simmim_ori_*`: the encoder keeps its final
        # LayerNorm -- the moco branch is what replaces it by nn.Identity, :362-363 -- no momentum networks, no heads)
        self.use_pixel_target, self.use_moco_target = bool(use_pixel_target), bool(use_moco_target)
        self.patchnet = patchnet_name if use_moco_target else 'no_patchtrans'
        self.patchnet_depth = 2                                          # `depth=2` where MoCo_ViT builds its patch_extractor (:383-396)
        # rows per image the patch extractor hands to the projector: the pooled windows, or ConvPatchNet's single row (`.unsqueeze(1)`, :257)
        self.n_patch = 1 if self.patchnet == 'conv' else num_windows
        self.has_pix_projector = self.use_pixel_target and self.use_moco_target
        self.has_final_norm = not self.use_moco_target
        self.comm = None            # set by dig_amd.parallel.DistributedDataParallel
        self.mlps = OrderedDict()
        if self.use_moco_target:
            self.mlps["encoder_projection_layer"] = _mlp_dims(3, D, mlp_dim, dim)
            self.mlps["momentum_projection_layer"] = _mlp_dims(3, D, mlp_dim, dim)
            self.mlps["predictor"] = _mlp_dims(2, dim, mlp_dim, dim)
        if self.has_pix_projector:
            self.mlps["pix_projector"] = _mlp_dims(3, D, 512, D)
            self.mlps["pix_projector_m"] = _mlp_dims(3, D, 512, D)
        self._build_layout()
        self._allocate(torch.device(device) if device is not None else torch.device("cpu"))
        self._init_weights()
        self._register_tree()
        # attributes the reference driver reads (run_mae_pretraining_moco.py:320)
        pe = self.encoder.patch_embed
        pe.patch_size, pe.patch_shape, pe.num_patches, pe.img_size = (patch_size, patch_size), (self.gh, self.gw), self.N, tuple(img_size)

    # ------------------------------------------------------------------ layout
    def _encoder_specs(self, pre, arena):
        D, Fh = self.D, self.F
        s = [ParamSpec(pre + "mask_token", (1, 1, D), 0, arena),
             ParamSpec(pre + "patch_embed.proj.weight", (D, 3, 4, 4), 0, arena),
             ParamSpec(pre + "patch_embed.proj.bias", (D,), 1, arena)]
        for i in range(self.depth):
            b = f"{pre}blocks.{i}."
            s += [ParamSpec(b + "norm1.weight", (D,), 1, arena), ParamSpec(b + "norm1.bias", (D,), 1, arena),
                  ParamSpec(b + "attn.q_bias", (D,), 1, arena), ParamSpec(b + "attn.v_bias", (D,), 1, arena),
                  ParamSpec(b + "attn.qkv.weight", (3 * D, D), 0, arena), ParamSpec(b + "attn.proj.weight", (D, D), 0, arena),
                  ParamSpec(b + "attn.proj.bias", (D,), 1, arena), ParamSpec(b + "norm2.weight", (D,), 1, arena),
                  ParamSpec(b + "norm2.bias", (D,), 1, arena), ParamSpec(b + "mlp.fc1.weight", (Fh, D), 0, arena),
                  ParamSpec(b + "mlp.fc1.bias", (Fh,), 1, arena), ParamSpec(b + "mlp.fc2.weight", (D, Fh), 0, arena),
                  ParamSpec(b + "mlp.fc2.bias", (D,), 1, arena)]
        if self.has_final_norm:                                          # modeling_pretrain_vit.py:59,104 (Gen-only)
            s += [ParamSpec(pre + "norm.weight", (D,), 1, arena), ParamSpec(pre + "norm.bias", (D,), 1, arena)]
        return s

    def _patchnet_specs(self, pre, arena):
        """PatchNet(use_patch_transformer=True): modeling_pretrain_moco_mim_ori.py:144-154 (blocks of :88-135, q / k / v without bias)."""
        D, Fh = self.D, self.F
        s = []
        for i in range(self.patchnet_depth):
            b = f"{pre}.blocks.{i}."
            s += [ParamSpec(b + "norm1.weight", (D,), 1, arena), ParamSpec(b + "norm1.bias", (D,), 1, arena),
                  ParamSpec(b + "attn.linear_q.weight", (D, D), 0, arena), ParamSpec(b + "attn.linear_k.weight", (D, D), 0, arena),
                  ParamSpec(b + "attn.linear_v.weight", (D, D), 0, arena), ParamSpec(b + "attn.proj.weight", (D, D), 0, arena),
                  ParamSpec(b + "attn.proj.bias", (D,), 1, arena), ParamSpec(b + "norm2.weight", (D,), 1, arena),
                  ParamSpec(b + "norm2.bias", (D,), 1, arena), ParamSpec(b + "mlp.fc1.weight", (Fh, D), 0, arena),
                  ParamSpec(b + "mlp.fc1.bias", (Fh,), 1, arena), ParamSpec(b + "mlp.fc2.weight", (D, Fh), 0, arena),
                  ParamSpec(b + "mlp.fc2.bias", (D,), 1, arena)]
        return s + [ParamSpec(pre + ".norm.weight", (D,), 1, arena), ParamSpec(pre + ".norm.bias", (D,), 1, arena)]

    def _convnet_specs(self, pre, arena):
        """ConvPatchNet: modeling_pretrain_moco_mim_ori.py:216-232 (conv_layers.{0,2,4,6} = Conv2d 3x3 with bias + BatchNorm2d; patches2global =
        Linear, BatchNorm1d, ReLU, Linear, BatchNorm1d(affine=False))."""
        from .convpatchnet import CONV_IDX, channels
        c, D = channels(self.D), self.D
        s = []
        for j, i in enumerate(CONV_IDX):
            b = f"{pre}.conv_layers.{i}."
            s += [ParamSpec(b + "0.weight", (c[j + 1], c[j], 3, 3), 0, arena), ParamSpec(b + "0.bias", (c[j + 1],), 1, arena),
                  ParamSpec(b + "1.weight", (c[j + 1],), 1, arena), ParamSpec(b + "1.bias", (c[j + 1],), 1, arena)]
        g = pre + ".patches2global."
        return s + [ParamSpec(g + "0.weight", (D, c[4] * self.num_windows), 0, arena), ParamSpec(g + "0.bias", (D,), 1, arena),
                    ParamSpec(g + "1.weight", (D,), 1, arena), ParamSpec(g + "1.bias", (D,), 1, arena),
                    ParamSpec(g + "3.weight", (D, D), 0, arena), ParamSpec(g + "3.bias", (D,), 1, arena)]

    def _bn_layers(self):
        """(state_dict prefix, channels) of every BatchNorm layer, in the reference's registration order: the BN-MLP heads, then ConvPatchNet's."""
        out = [(f"{pre}.{3 * l + 1}", d2) for pre, dims in self.mlps.items() for l, (_, d2) in enumerate(dims)]
        if self.patchnet == 'conv':
            from .convpatchnet import CONV_IDX, channels
            c = channels(self.D)
            for pre in ("patch_extractor", "momentum_patch_extractor"):
                out += [(f"{pre}.conv_layers.{i}.1", c[j + 1]) for j, i in enumerate(CONV_IDX)]
                out += [(f"{pre}.patches2global.1", self.D), (f"{pre}.patches2global.4", self.D)]
        return out

    @staticmethod
    def _mlp_specs(pre, dims, arena):
        s, n = [], len(dims)
        for l, (d1, d2) in enumerate(dims):
            s.append(ParamSpec(f"{pre}.{3 * l}.weight", (d2, d1), 0, arena))
            if l < n - 1:
                s.append(ParamSpec(f"{pre}.{3 * l + 1}.weight", (d2,), 1, arena))
                s.append(ParamSpec(f"{pre}.{3 * l + 1}.bias", (d2,), 1, arena))
        return s

    def _build_layout(self):
        """Reference order of named_parameters() (this is also the registration order), then arena offsets.
        Online arena: [encoder | encoder_projection_layer | pix_projector | predictor | pix_decoder]; the momentum
        arena mirrors the first three groups offset-for-offset so the EMA is one flat kernel.  q_bias and v_bias of
        a block share one 3*D bundle [q_bias | zeros | v_bias] = the bias vector of the fused QKV GEMM
        (modeling_finetune.py:91: K has no bias)."""
        D, Dd = self.D, self.dec_dim
        specs = []
        specs += self._encoder_specs("encoder.", "online")
        if self.use_moco_target:
            specs += self._encoder_specs("momentum_encoder.", "momentum")
            specs += self._mlp_specs("encoder_projection_layer", self.mlps["encoder_projection_layer"], "online")
            specs += self._mlp_specs("momentum_projection_layer", self.mlps["momentum_projection_layer"], "momentum")
            specs += self._mlp_specs("predictor", self.mlps["predictor"], "online")
            if self.patchnet == 'regular':
                specs += self._patchnet_specs("patch_extractor", "online")
                specs += self._patchnet_specs("momentum_patch_extractor", "momentum")
            elif self.patchnet == 'conv':
                specs += self._convnet_specs("patch_extractor", "online")
                specs += self._convnet_specs("momentum_patch_extractor", "momentum")
        if self.has_pix_projector:
            specs += self._mlp_specs("pix_projector", self.mlps["pix_projector"], "online")
            specs += self._mlp_specs("pix_projector_m", self.mlps["pix_projector_m"], "momentum")
        if self.use_pixel_target:
            specs += [ParamSpec("pix_decoder.0.weight", (Dd, D), 0, "online"), ParamSpec("pix_decoder.1.weight", (Dd, Dd), 0, "online"),
                      ParamSpec("pix_decoder.2.weight", (Dd,), 1, "online"), ParamSpec("pix_decoder.2.bias", (Dd,), 1, "online"),
                      ParamSpec("pix_decoder.4.weight", (self.dec_classes, Dd), 0, "online"),
                      ParamSpec("pix_decoder.4.bias", (self.dec_classes,), 1, "online")]
        self.specs = OrderedDict((s.name, s) for s in specs)
        if not self.use_pixel_target:
            # Dis-only: the encoder runs without a mask (`vis_mask_pos = None`, :493-494), mask_token is never read, its .grad stays None
            # and the reference's AdamW skips it altogether (custom_optim/adamw.py:78-79: no decay, no state): granule group 2 = untouched
            self.specs["encoder.mask_token"].group = 2

        def place(names, arena_groups):
            off = 0
            for n in names:
                s = self.specs[n]
                if n.endswith("attn.v_bias"):
                    continue                                            # placed with its q_bias
                s.offset = off
                if n.endswith("attn.q_bias"):
                    self.specs[n[:-len("q_bias")] + "v_bias"].offset = off + 2 * D
                    size = 3 * D
                else:
                    size = s.numel
                padded = _round_up(size, ALIGN)
                arena_groups.extend([s.group] * (padded // ALIGN))
                off += padded
            return off

        ema_src = [n for n in self.specs if n.startswith(("encoder.", "encoder_projection_layer.", "patch_extractor.", "pix_projector."))]
        rest = [n for n in self.specs if n.startswith(("predictor.", "pix_decoder."))]
        self._online_groups = []
        self.n_ema = place(ema_src, self._online_groups)                    # elements covered by the EMA
        rest_groups = []
        n_rest = place(rest, rest_groups)
        for n in rest:
            if self.specs[n].offset >= 0:
                self.specs[n].offset += self.n_ema
        self._online_groups += rest_groups
        self.n_online = self.n_ema + n_rest
        mom = [n for n in self.specs if self.specs[n].arena == "momentum"]
        tmp = []
        n_mom = place(mom, tmp)
        if not self.use_moco_target:
            self.n_ema = 0                                                  # Gen-only: no momentum arena, nothing for the EMA kernel
        assert n_mom == self.n_ema
        # the momentum arena must mirror the online one offset-for-offset
        for n in mom:
            src = (n.replace("momentum_encoder.", "encoder.").replace("momentum_projection_layer.", "encoder_projection_layer.")
                   .replace("momentum_patch_extractor.", "patch_extractor.").replace("pix_projector_m.", "pix_projector."))
            assert self.specs[src].offset == self.specs[n].offset, (n, src)
        # gradient-bucket boundaries (element ranges of the online arena), in backward-completion order
        heads = [k for k, on in (("pix_decoder", self.use_pixel_target), ("predictor", self.use_moco_target),
                                 ("encoder_projection_layer", self.use_moco_target), ("patch_extractor", self.patchnet != 'no_patchtrans'),
                                 ("pix_projector", self.has_pix_projector)) if on]
        self.bucket_names = (heads + (["encoder.norm"] if self.has_final_norm else [])
                             + [f"encoder.blocks.{i}" for i in reversed(range(self.depth))] + ["encoder.embed"])

    def bucket_groups(self):
        """Gradient buckets that travel in ONE all-reduce: neighbours in the arena whose gradients become final within a few launches of each
        other (xGMI is point-to-point: fewer, larger messages -- the patch embedding's 78 KB and the decoder's 0.5 MB are latency, not
        bandwidth).  {key: tuple of the keys of its group}; a group's range is contiguous (checked)."""
        cache = getattr(self, "_bucket_groups", None)
        if cache is None:
            names = set(self.bucket_names)
            groups = [g for g in (("predictor", "pix_decoder"), ("encoder_projection_layer", "patch_extractor", "pix_projector"),
                                  ("encoder.embed", "encoder.blocks.0")) if sum(k in names for k in g) >= 2]
            cache = {}
            for g in groups:
                g = tuple(k for k in g if k in names)
                rng = sorted(self.bucket_range(k) for k in g)
                if all(a[1] == b[0] for a, b in zip(rng, rng[1:])):      # (contiguous in the arena: always, by the layout of _build_layout)
                    cache.update({k: g for k in g})
            self._bucket_groups = cache
        return cache

    def bucket_range(self, key):
        """[begin, end) element range of the online arena that holds the parameters of one backward stage."""
        if key == "encoder.embed":
            names = ["encoder.mask_token", "encoder.patch_embed.proj.weight", "encoder.patch_embed.proj.bias"]
        else:
            names = [n for n in self.specs if n.startswith(key + ".") and self.specs[n].arena == "online"]
        lo = min(self.specs[n].offset for n in names)
        hi = max(_round_up(self.specs[n].offset + self.specs[n].numel, ALIGN) for n in names)
        return lo, hi

    # ------------------------------------------------------------------ storage
    def _allocate(self, device):
        self._flat = {
            "online": torch.zeros(self.n_online, dtype=F32, device=device),
            "momentum": torch.zeros(max(self.n_ema, ALIGN) if not self.use_moco_target else self.n_ema, dtype=F32, device=device),
            "grad": torch.zeros(self.n_online, dtype=F32, device=device),
            "groups": torch.tensor(self._online_groups, dtype=torch.uint8, device=device),
        }
        n_bn = len(self._bn_layers())
        c_tot = sum(C for _, C in self._bn_layers())
        self._flat["bn_stats"] = torch.zeros(2 * c_tot, dtype=F32, device=device)      # running_mean | running_var
        self._flat["bn_count"] = torch.zeros(n_bn, dtype=torch.int64, device=device)
        self._shadow = {}
        self._views_version = 0

    def _view(self, arena, spec):
        return self._flat[arena][spec.offset:spec.offset + spec.numel].view(spec.shape)

    def _register_tree(self):
        """Create the reference's module tree with Parameters/buffers that are views into the arenas."""
        self._param_objs = OrderedDict()
        for name, s in self.specs.items():
            mod, leaf = _node_for(self, name)
            p = nn.Parameter(self._view(s.arena, s), requires_grad=(s.arena == "online"))
            mod.register_parameter(leaf, p)
            self._param_objs[name] = p
        self._buffer_slots = []
        c_off, i_bn = 0, 0
        c_tot = self._flat["bn_stats"].numel() // 2
        for key, d2 in self._bn_layers():
            mod, _ = _node_for(self, key + ".x")
            self._buffer_slots.append((key, c_off, d2, i_bn))
            mod.register_buffer("running_mean", None)
            mod.register_buffer("running_var", None)
            mod.register_buffer("num_batches_tracked", None)
            c_off += d2
            i_bn += 1
        self.encoder.pos_embed = get_sinusoid_encoding_table(self.N, self.D)     # plain attribute (not in state_dict)
        if self.use_moco_target:
            self.momentum_encoder.pos_embed = self.encoder.pos_embed
        self._rebind()

    def _rebind(self):
        for name, s in self.specs.items():
            p = self._param_objs[name]
            p.data = self._view(s.arena, s)
            if s.arena == "online":
                p.grad = self._view("grad", s)
        c_tot = self._flat["bn_stats"].numel() // 2
        self._bn_views = {}
        for key, c_off, C, i_bn in self._buffer_slots:
            mod, _ = _node_for(self, key + ".x")
            rm = self._flat["bn_stats"][c_off:c_off + C]
            rv = self._flat["bn_stats"][c_tot + c_off:c_tot + c_off + C]
            mod._buffers["running_mean"], mod._buffers["running_var"] = rm, rv
            mod._buffers["num_batches_tracked"] = self._flat["bn_count"][i_bn]
            self._bn_views[key] = (rm, rv, i_bn)
        dev = self._flat["online"].device
        self._pos = self.encoder.pos_embed[0].to(dev).contiguous()
        self._shadow = {}
        self._f32 = {n: self._view(s.arena, s) for n, s in self.specs.items()}
        self._g32 = {n: self._view("grad", s) for n, s in self.specs.items() if s.arena == "online"}
        self._qkv_bias, self._qkv_bias_grad = {}, {}
        for n, s in self.specs.items():
            if n.endswith("attn.q_bias"):
                key = n[:-len("q_bias")]
                self._qkv_bias[key] = self._flat[s.arena][s.offset:s.offset + 3 * self.D]
                if s.arena == "online":
                    self._qkv_bias_grad[key] = self._flat["grad"][s.offset:s.offset + 3 * self.D]
        self._views_version += 1
        self._fresh = None

    def _apply(self, fn, recurse=True):
        self._flat = {k: fn(v) for k, v in self._flat.items()}
        self._rebind()
        return self

    def load_state_dict(self, state_dict, strict=True):
        missing, unexpected = [], [k for k in state_dict if k not in self.specs and not k.rsplit(".", 1)[-1].startswith(("running_", "num_batches"))]
        own = self.state_dict()
        with torch.no_grad():
            for k, v in own.items():
                if k in state_dict:
                    v.copy_(state_dict[k])
                else:
                    missing.append(k)
        self.mark_weights_changed()
        if strict and (missing or [k for k in state_dict if k not in own]):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]}, unexpected {[k for k in state_dict if k not in own][:5]}")
        return torch.nn.modules.module._IncompatibleKeys(missing, [k for k in state_dict if k not in own])

    # ------------------------------------------------------------------ init (distributions of the reference)
    def _init_weights(self):
        """modeling_pretrain_vit.py:63-73 (xavier-uniform Linear, zero bias, LN 1/0), modeling_pretrain_moco_mim_ori.py:
        353-355 (patch-embed uniform +-sqrt(6/(48+D))), nn.Linear / nn.BatchNorm1d defaults for the heads, mask_token
        zeros; momentum parameters start as copies (:396-420)."""
        with torch.no_grad():
            for n, s in self.specs.items():
                if s.arena != "online":
                    continue
                v = self._view("online", s)
                if n.endswith("mask_token"):
                    v.zero_()
                elif n.endswith("patch_embed.proj.weight"):
                    # :353-355 sits inside `if use_moco_target`: a Gen-only model keeps nn.Conv2d's own init (weight and bias U(+-1/sqrt(fan_in)))
                    a = math.sqrt(6.0 / float(3 * 16 + self.D)) if self.use_moco_target else 1.0 / math.sqrt(48.0)
                    v.uniform_(-a, a)
                elif n.endswith("patch_embed.proj.bias") and not self.use_moco_target:
                    v.uniform_(-1.0 / math.sqrt(48.0), 1.0 / math.sqrt(48.0))
                elif n.startswith("patch_extractor.") and self.patchnet == 'conv':
                    # ConvPatchNet has no _init_weights: nn.Conv2d / nn.Linear defaults (weight and bias U(+-1/sqrt(fan_in))), BatchNorm 1 / 0
                    if len(s.shape) >= 2:
                        b = 1.0 / math.sqrt(s.numel // s.shape[0])
                        v.uniform_(-b, b)
                    elif n.endswith((".0.bias", ".3.bias")):
                        ws = self.specs[n[:-4] + "weight"]
                        b = 1.0 / math.sqrt(ws.numel // ws.shape[0])
                        v.uniform_(-b, b)
                    elif n.endswith(".weight"):
                        v.fill_(1.0)
                    else:
                        v.zero_()
                elif n.startswith(("encoder.", "patch_extractor.")):    # (PatchNet._init_weights, :159-166: the encoder's rule)
                    if len(s.shape) == 2:
                        nn.init.xavier_uniform_(v)
                    elif n.endswith(("norm1.weight", "norm2.weight", "norm.weight")):
                        v.fill_(1.0)
                    else:
                        v.zero_()
                elif len(s.shape) == 2:
                    nn.init.kaiming_uniform_(v, a=math.sqrt(5))
                elif n == "pix_decoder.4.bias":
                    b = 1.0 / math.sqrt(self.dec_dim)
                    v.uniform_(-b, b)
                elif n.endswith(".weight"):
                    v.fill_(1.0)
                else:
                    v.zero_()
            if self.n_ema:
                self._flat["momentum"].copy_(self._flat["online"][:self.n_ema])
            c_tot = self._flat["bn_stats"].numel() // 2
            self._flat["bn_stats"][c_tot:].fill_(1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    # ------------------------------------------------------------------ flat accessors used by optimizer / DDP
    @property
    def flat_params(self):
        return self._flat["online"]

    @property
    def flat_grads(self):
        return self._flat["grad"]

    @property
    def flat_groups(self):
        return self._flat["groups"]

    def shadow(self, arena):
        """bf16 copy of an arena (same element offsets); (re)created lazily, refreshed by sync_shadow()/EMA kernel."""
        sh = self._shadow.get(arena)
        if sh is None or sh.device != self._flat[arena].device:
            sh = torch.empty(self._flat[arena].numel(), dtype=BF16, device=self._flat[arena].device)
            self._shadow[arena] = sh
            self._w16 = None
        return sh

    def _w(self, arena):
        """name -> bf16 2-D weight view dictionary for an arena."""
        cache = getattr(self, "_w16", None)
        if cache is None:
            cache = self._w16 = {}
        if arena not in cache:
            sh = self.shadow(arena)
            cache[arena] = {n: sh[s.offset:s.offset + s.numel].view(s.shape[0], -1) for n, s in self.specs.items()
                            if s.arena == arena and len(s.shape) >= 2 and not n.endswith("mask_token")}
        return cache[arena]

    # ------------------------------------------------------------------ what the optimizer launch leaves for the next forward
    # The fused AdamW launch (optim_factory.FusedAdamW.step -> dig_adamw_step_tr) writes, next to the fp32 parameters, the bf16 operand
    # shadow of the online arena AND the K-contiguous transposed copies W^T of the weights the fused MLP backward reads (fc2, fc1, proj of
    # every block): the forward of the next step then needs neither its cast launch nor the three transpose launches.  "Fresh" = nothing has
    # changed the online arena since that launch: load_state_dict / .to() / _rebind() clear it, torch in-place operations on the arena or
    # on a parameter bump the arena's version counter (checked); code that writes parameters through `.data` or raw pointers must call
    # mark_weights_changed().
    def transposed_weight_table(self):
        """(device table of dig_adamw_step_tr, n_mats, n_tiles, tr_out, [(w2t, w1t, projt, qkvt) views per block]) or None when a shape is not a
        multiple of 64 (the transposes are then rebuilt by the forward, as before)."""
        dev = self._flat["online"].device
        cache = getattr(self, "_tr_table", None)
        if cache is not None and cache[0] == (self._views_version, dev):
            return cache[1]
        recs, views, dst, tile0 = [], [], 0, 0
        per_block = []
        ok = dev.type == "cuda"

        def add(name):
            nonlocal dst, tile0, ok
            sp = self.specs[name]
            r, c = sp.shape
            ok = ok and r % 64 == 0 and c % 64 == 0 and sp.offset % ALIGN == 0
            recs.append((sp.offset, dst, r, c, tile0))
            ent = (dst, c, r)
            dst += r * c
            tile0 += (r // 64) * (c // 64)
            return ent
        for i in range(self.depth):
            per_block.append([add(f"encoder.blocks.{i}.{leaf}") for leaf in ("mlp.fc2.weight", "mlp.fc1.weight", "attn.proj.weight", "attn.qkv.weight")])
        # opt-in (DIG_HEAD_DGRAD_DIRECT=1: no gain in the step): the BN-MLP heads' Linear weights too (their data gradients in direct form); a
        # head whose widths are not multiples of 64 (test models) stays on the transpose-read form
        head_ents = {}
        for pre in (("predictor", "encoder_projection_layer", "pix_projector") if ops.HEAD_DGRAD_DIRECT else ()):
            for l in range(len(self.mlps.get(pre, ()))):
                name = f"{pre}.{3 * l}.weight"
                r, c = self.specs[name].shape
                if r % 64 == 0 and c % 64 == 0:
                    head_ents[name] = add(name)
        out = None
        if ok:
            import struct
            raw = b"".join(struct.pack("<qqiiii", off, d, r, c, t0, 0) for off, d, r, c, t0 in recs)
            table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
            tr_out = torch.empty(dst, device=dev, dtype=BF16)
            wT = [tuple(tr_out[d:d + a * b].view(a, b) for d, a, b in row) for row in per_block]
            flags = self._flat["groups"].clone()
            for off, _, r, c, _ in recs:
                flags[off // ALIGN:(off + r * c) // ALIGN] |= 0x80
            heads = {n_: tr_out[d:d + a * b].view(a, b) for n_, (d, a, b) in head_ents.items()}
            out = (table, len(recs), tile0, tr_out, wT, flags, heads)
        self._tr_table = ((self._views_version, dev), out)
        return out

    def mark_weights_changed(self):
        """The online parameters were written by something other than the fused optimizer launch: the next forward re-casts the bf16
        shadow and rebuilds the transposed weight copies."""
        self._fresh = None

    def _set_fresh(self):
        self._fresh = (self._views_version, self._flat["online"]._version)

    def weights_fresh(self):
        f = getattr(self, "_fresh", None)
        return f is not None and f == (self._views_version, self._flat["online"]._version)

    def _side_stream(self, dev):
        """Second HIP stream for the gradient-free momentum branch (overlaps the online forward)."""
        st = getattr(self, "_side", None)
        if st is None or st.device != dev:
            # high priority: the momentum forward and the weight-gradient GEMMs are what the main chain waits for at its joins, and
            # their workgroups are the ones that lose the CU slots to the main chain's back-to-back launches (in the step, A/B on one
            # box: 25.13 -> 24.99 ms)
            st = self._side = torch.cuda.Stream(device=dev, priority=-1)
        return st

    def _bwd_side_stream(self, dev):
        """HIP stream of the backward's parameter-gradient reductions (bias / LayerNorm column sums: a few small launches per encoder block
        that only consume what the data-gradient chain has produced).  DIG_BWD_SIDE_PRIO = high (the stream above), normal or low."""
        import os
        mode = os.environ.get("DIG_BWD_SIDE_PRIO", "high")
        if mode == "high":
            return self._side_stream(dev)
        st = getattr(self, "_bwd_side", None)
        if st is None or st.device != dev:
            lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
            st = self._bwd_side = torch.cuda.Stream(device=dev, priority=(max(lo, hi) if mode == "low" else 0))
        return st

    def _fwd_stream(self, dev):
        """HIP stream of the gradient-free momentum branch in the FORWARD.  DIG_FWD_MOM_PRIO = high (the weight-gradient stream itself),
        normal or low: with the big forward kernels owning the whole chip one at a time (persistent GEMM tiles, the fused MLP chain), the
        two branches serialise kernel by kernel; below the main stream's priority the online encoder gets the chip first and the
        momentum branch fills the time in which the online branch runs its small head kernels."""
        import os
        mode = os.environ.get("DIG_FWD_MOM_PRIO", "high")
        if mode == "high":
            return self._side_stream(dev)
        st = getattr(self, "_fwd_side", None)
        if st is None or st.device != dev:
            lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
            st = self._fwd_side = torch.cuda.Stream(device=dev, priority=(max(lo, hi) if mode == "low" else 0))
        return st

    def _mask_count(self, mask_u8, B):
        """Masked tokens per sample of view 0 (the reference reshapes to [B, -1, C], so it is constant over the
        batch).  Read back once and cached: per-step validation happens where the engine synchronises anyway."""
        c = getattr(self, "_per_sample_mask", None)
        if c is None or getattr(self, "_per_sample_mask_n", None) != mask_u8.shape[1]:
            c = self._per_sample_mask = int(mask_u8[0].sum().item())
            self._per_sample_mask_n = mask_u8.shape[1]
        return c

    def reset_mask_count(self):
        """Forget the cached number of masked tokens per sample: call when the masking ratio of the data pipeline changes (the next
        forward reads it back once).  Without it a changed ratio is caught by the engine's read-back, one step late, as an error."""
        self._per_sample_mask = None

    # ------------------------------------------------------------------ forward
    def forward(self, image, aug_image, vis_mask_pos, m, only_mim_on_ori_img=True):
        from .engine_core import dig_forward
        return dig_forward(self, image, aug_image, vis_mask_pos, m, only_mim_on_ori_img)


def _factory(embed_dim, heads, pixel=True, moco=True, **kwargs):
    kwargs.pop("pretrained", None)
    init_ckpt = kwargs.pop("init_ckpt", None)
    model = MoCo_ViT(img_size=(32, 128), patch_size=4, encoder_embed_dim=embed_dim, encoder_depth=12, encoder_num_heads=heads,
                     encoder_num_classes=0, decoder_num_classes=48, decoder_embed_dim=192, decoder_depth=4, decoder_num_heads=3,
                     mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), use_pixel_target=pixel,
                     use_moco_target=moco, **kwargs)
    model.default_cfg = {'url': '', 'num_classes': 1000, 'input_size': (3, 32, 128), 'pool_size': None, 'crop_pct': 1.0,
                         'interpolation': 'bicubic', 'mean': (0.5, 0.5, 0.5), 'std': (0.5, 0.5, 0.5)}
    if init_ckpt:
        model.load_state_dict(torch.load(init_ckpt, map_location="cpu", weights_only=False)["model"])   # reference checkpoints carry an argparse.Namespace
    return model


@register_model
def pretrain_simmim_moco_ori_vit_tiny_patch4_32x128(pretrained=False, **kwargs):
    """modeling_pretrain_moco_mim_ori.py:736-761 (D=192, 3 heads)."""
    return _factory(192, 3, init_ckpt=kwargs.pop("init_ckpt", None) if pretrained else None, **kwargs)


@register_model
def pretrain_simmim_moco_ori_vit_small_patch4_32x128(pretrained=False, **kwargs):
    """modeling_pretrain_moco_mim_ori.py:682-707 (D=384, 6 heads): the north-star model."""
    return _factory(384, 6, init_ckpt=kwargs.pop("init_ckpt", None) if pretrained else None, **kwargs)


@register_model
def pretrain_simmim_moco_ori_vit_base_patch4_32x128(pretrained=False, **kwargs):
    """modeling_pretrain_moco_mim_ori.py:792-817 (D=512, 8 heads)."""
    return _factory(512, 8, init_ckpt=kwargs.pop("init_ckpt", None) if pretrained else None, **kwargs)


# ---- the single-objective factories of the reference: the paper's ablation models, driven by the same train_one_epoch
# (engine_for_pretraining_moco.py:119-144 tests `'contra_loss' in out_dict` / `'vis_out' in out_dict`)
def _single(name, embed_dim, heads, pixel, moco, doc):
    def fn(pretrained=False, **kwargs):
        return _factory(embed_dim, heads, pixel=pixel, moco=moco, init_ckpt=kwargs.pop("init_ckpt", None) if pretrained else None, **kwargs)
    fn.__name__ = fn.__qualname__ = name
    fn.__doc__ = doc
    fn.__module__ = __name__
    globals()[name] = register_model(fn)


for _size, (_d, _h) in (("tiny", (192, 3)), ("small", (384, 6)), ("base", (512, 8))):
    _single(f"pretrain_moco_ori_vit_{_size}_patch4_32x128", _d, _h, False, True,
            "Dis-only (use_pixel_target=False): MoCo-v3 on unmasked views, no pix_projector, no decoder "
            "(modeling_pretrain_moco_mim_ori.py:627-653 small, :709-735 tiny, :818-843 base).")
    _single(f"pretrain_simmim_ori_vit_{_size}_patch4_32x128", _d, _h, True, False,
            "Gen-only (use_moco_target=False): encoder + its final LayerNorm + pix_decoder "
            "(modeling_pretrain_moco_mim_ori.py:655-681 small, :737-763 tiny, :845-871 base).")
