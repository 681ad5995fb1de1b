"""CPU oracle for the DiG pre-training hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional (parameter-dict) restatement in plain fp32 PyTorch of what one
`train_one_epoch` step of the reference computes for the `pretrain_simmim_moco_ori_vit_*_patch4_32x128`
models.  Every function cites the reference file:line it follows (paths relative to the reference
repo).  It exists to CHECK the HIP path:

  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
  * the product package `dig_amd/` never imports it and has no CPU fallback.

Pinning: the reference ships no tests or golden vectors for this path, so the oracle is
pinned against outputs of the reference itself, imported unmodified in the build container by
`oracle/ref_harness/gen_golden.py`; the resulting fixtures live in `tests/golden/` and
`tests/test_oracle_golden.py` replays them on every run (CPU).  See DESIGN.md §Oracle.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------------
# configuration  (modeling_pretrain_moco_mim_ori.py:682-707 small, :792-817 base, :736-761 tiny)
# --------------------------------------------------------------------------------------------------
@dataclass
class DiGConfig:
    img_h: int = 32
    img_w: int = 128
    patch: int = 4
    in_chans: int = 3
    embed_dim: int = 384
    depth: int = 12
    heads: int = 6
    mlp_ratio: float = 4.0
    dec_dim: int = 192          # decoder_embed_dim
    dec_classes: int = 48       # decoder_num_classes = 4*4*3
    moco_dim: int = 256         # --moco_dim
    moco_mlp_dim: int = 4096    # --moco_mlp_dim
    pix_mlp_dim: int = 512      # hard-coded in modeling_pretrain_moco_mim_ori.py:415
    T: float = 0.2              # --moco_t
    num_windows: int = 4
    ln_eps: float = 1e-6        # partial(nn.LayerNorm, eps=1e-6)
    bn_eps: float = 1e-5        # nn.BatchNorm1d default
    bn_momentum: float = 0.1
    # which objectives the model carries (MoCo_ViT(use_pixel_target, use_moco_target), modeling_pretrain_moco_mim_ori.py:340-426):
    #   "simmim_moco" both (the pretrain_simmim_moco_ori_* factories), "moco" = Dis-only (pretrain_moco_ori_*, :627-653,709-735,818-843:
    #   no mask, no pix_projector, no decoder), "simmim" = Gen-only (pretrain_simmim_ori_*, :655-681,737-763,845-871: the encoder keeps its
    #   final LayerNorm, no momentum networks, no heads; the patch embedding keeps nn.Conv2d's default init)
    kind: str = "simmim_moco"
    # --patchnet_name (run_mae_pretraining_moco.py:145): "no_patchtrans" (README) = the pooled windows themselves; "regular" (the argparse
    # default) = PatchNet with its 2-block patch transformer: the pooled windows attend over all 256 tokens (modeling_pretrain_moco_mim_ori.py:137-205)
    # "conv" = ConvPatchNet (:207-260): four 3x3-convolution / BatchNorm2d / ReLU blocks with three 2x2 max-pools on the 8 x 32 token grid,
    # adaptive_avg_pool2d to (1, num_windows), then Linear-BN-ReLU-Linear-BN(affine=False): ONE patch per image
    patchnet: str = "no_patchtrans"
    patchnet_depth: int = 2
    patchnet_eps: float = 1e-5        # PatchNet's own norm_layer default: nn.LayerNorm (eps 1e-5), not the encoder's 1e-6

    @property
    def use_moco(self) -> bool:
        return self.kind in ("simmim_moco", "moco")

    @property
    def use_pixel(self) -> bool:
        return self.kind in ("simmim_moco", "simmim")

    @property
    def n_patch(self) -> int:
        """Rows per image the patch extractor hands to the projector (ConvPatchNet.forward ends in `.unsqueeze(1)`, :257)."""
        return 1 if self.patchnet == "conv" else self.num_windows

    @property
    def conv_channels(self) -> Tuple[int, ...]:
        """ConvPatchNet's conv3x3_block widths (:217-225)."""
        D = self.embed_dim
        return (D, D, int(D * 1.5), D * 2, D * 2)

    @property
    def grid(self) -> Tuple[int, int]:
        return (self.img_h // self.patch, self.img_w // self.patch)

    @property
    def num_patches(self) -> int:
        return self.grid[0] * self.grid[1]

    @property
    def hidden(self) -> int:
        return int(self.embed_dim * self.mlp_ratio)

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.heads


CONFIGS = {
    "pretrain_simmim_moco_ori_vit_tiny_patch4_32x128": dict(embed_dim=192, depth=12, heads=3),
    "pretrain_simmim_moco_ori_vit_small_patch4_32x128": dict(embed_dim=384, depth=12, heads=6),
    "pretrain_simmim_moco_ori_vit_base_patch4_32x128": dict(embed_dim=512, depth=12, heads=8),
    "pretrain_moco_ori_vit_tiny_patch4_32x128": dict(embed_dim=192, depth=12, heads=3, kind="moco"),
    "pretrain_moco_ori_vit_small_patch4_32x128": dict(embed_dim=384, depth=12, heads=6, kind="moco"),
    "pretrain_moco_ori_vit_base_patch4_32x128": dict(embed_dim=512, depth=12, heads=8, kind="moco"),
    "pretrain_simmim_ori_vit_tiny_patch4_32x128": dict(embed_dim=192, depth=12, heads=3, kind="simmim"),
    "pretrain_simmim_ori_vit_small_patch4_32x128": dict(embed_dim=384, depth=12, heads=6, kind="simmim"),
    "pretrain_simmim_ori_vit_base_patch4_32x128": dict(embed_dim=512, depth=12, heads=8, kind="simmim"),
}


def make_config(name: str, **over) -> DiGConfig:
    kw = dict(CONFIGS[name])
    kw.update(over)
    return DiGConfig(**kw)


# --------------------------------------------------------------------------------------------------
# parameter / buffer inventory, in the reference's named_parameters()/state_dict() order
# --------------------------------------------------------------------------------------------------
def _encoder_param_shapes(cfg: DiGConfig, pre: str) -> "OrderedDict[str, tuple]":
    D, Fh = cfg.embed_dim, cfg.hidden
    o = OrderedDict()
    o[pre + "mask_token"] = (1, 1, D)                                   # modeling_pretrain_vit.py:42
    o[pre + "patch_embed.proj.weight"] = (D, cfg.in_chans, cfg.patch, cfg.patch)  # modeling_finetune.py:188
    o[pre + "patch_embed.proj.bias"] = (D,)
    for i in range(cfg.depth):                                          # modeling_finetune.py:128-148
        b = f"{pre}blocks.{i}."
        o[b + "norm1.weight"] = (D,)
        o[b + "norm1.bias"] = (D,)
        o[b + "attn.q_bias"] = (D,)
        o[b + "attn.v_bias"] = (D,)
        o[b + "attn.qkv.weight"] = (3 * D, D)
        o[b + "attn.proj.weight"] = (D, D)
        o[b + "attn.proj.bias"] = (D,)
        o[b + "norm2.weight"] = (D,)
        o[b + "norm2.bias"] = (D,)
        o[b + "mlp.fc1.weight"] = (Fh, D)
        o[b + "mlp.fc1.bias"] = (Fh,)
        o[b + "mlp.fc2.weight"] = (D, Fh)
        o[b + "mlp.fc2.bias"] = (D,)
    if not cfg.use_moco:                                                # Gen-only keeps encoder.norm (the moco branch replaces it by
        o[pre + "norm.weight"] = (D,)                                   # nn.Identity, modeling_pretrain_moco_mim_ori.py:362-363)
        o[pre + "norm.bias"] = (D,)
    return o


def _patchnet_param_shapes(cfg: DiGConfig, pre: str) -> "OrderedDict[str, tuple]":
    """PatchNet(use_patch_transformer=True): `depth` cross-attention Blocks (modeling_pretrain_moco_mim_ori.py:21-135: separate q / k / v
    Linear layers without bias -- qkv_bias defaults to False there) and a final LayerNorm."""
    D, Fh = cfg.embed_dim, cfg.hidden
    o = OrderedDict()
    for i in range(cfg.patchnet_depth):
        b = f"{pre}blocks.{i}."
        o[b + "norm1.weight"] = (D,); o[b + "norm1.bias"] = (D,)
        o[b + "attn.linear_q.weight"] = (D, D); o[b + "attn.linear_k.weight"] = (D, D); o[b + "attn.linear_v.weight"] = (D, D)
        o[b + "attn.proj.weight"] = (D, D); o[b + "attn.proj.bias"] = (D,)
        o[b + "norm2.weight"] = (D,); o[b + "norm2.bias"] = (D,)
        o[b + "mlp.fc1.weight"] = (Fh, D); o[b + "mlp.fc1.bias"] = (Fh,)
        o[b + "mlp.fc2.weight"] = (D, Fh); o[b + "mlp.fc2.bias"] = (D,)
    o[pre + "norm.weight"] = (D,); o[pre + "norm.bias"] = (D,)
    return o


CONV_IDX = (0, 2, 4, 6)                  # positions of the conv3x3_blocks in ConvPatchNet.conv_layers (max-pools sit at 1, 3, 5)


def _convnet_param_shapes(cfg: DiGConfig, pre: str) -> "OrderedDict[str, tuple]":
    """ConvPatchNet (modeling_pretrain_moco_mim_ori.py:207-260): conv_layers.{0,2,4,6} = Sequential(Conv2d 3x3 pad 1 (bias), BatchNorm2d, ReLU);
    patches2global = Linear(2 D nw -> D), BatchNorm1d, ReLU, Linear(D -> D), BatchNorm1d(affine=False)."""
    c = cfg.conv_channels
    o = OrderedDict()
    for j, i in enumerate(CONV_IDX):
        o[f"{pre}conv_layers.{i}.0.weight"] = (c[j + 1], c[j], 3, 3); o[f"{pre}conv_layers.{i}.0.bias"] = (c[j + 1],)
        o[f"{pre}conv_layers.{i}.1.weight"] = (c[j + 1],); o[f"{pre}conv_layers.{i}.1.bias"] = (c[j + 1],)
    D = cfg.embed_dim
    o[pre + "patches2global.0.weight"] = (D, c[4] * cfg.num_windows); o[pre + "patches2global.0.bias"] = (D,)
    o[pre + "patches2global.1.weight"] = (D,); o[pre + "patches2global.1.bias"] = (D,)
    o[pre + "patches2global.3.weight"] = (D, D); o[pre + "patches2global.3.bias"] = (D,)
    return o


def _convnet_buffer_shapes(cfg: DiGConfig, pre: str) -> "OrderedDict[str, tuple]":
    c = cfg.conv_channels
    o = OrderedDict()
    for key, C in [(f"conv_layers.{i}.1", c[j + 1]) for j, i in enumerate(CONV_IDX)] + [("patches2global.1", cfg.embed_dim), ("patches2global.4", cfg.embed_dim)]:
        o[f"{pre}{key}.running_mean"] = (C,); o[f"{pre}{key}.running_var"] = (C,); o[f"{pre}{key}.num_batches_tracked"] = ()
    return o


def bn_cancelled_bias(name: str, cfg: DiGConfig) -> bool:
    """A bias whose layer feeds a BatchNorm directly (ConvPatchNet's conv and Linear biases): the normalisation subtracts it again, its true
    gradient is exactly zero and what any implementation holds there is round-off of its own summation order."""
    return (cfg.patchnet == "conv" and name.startswith("patch_extractor.")
            and name.endswith((".0.bias", "patches2global.3.bias")) and not name.endswith(".1.bias"))


def _mlp_dims(n_layers: int, din: int, dmid: int, dout: int) -> List[Tuple[int, int]]:
    """_build_mlp, modeling_pretrain_moco_mim_ori.py:463-482."""
    return [(din if l == 0 else dmid, dout if l == n_layers - 1 else dmid) for l in range(n_layers)]


def _mlp_param_shapes(pre: str, dims) -> "OrderedDict[str, tuple]":
    o = OrderedDict()
    n = len(dims)
    for l, (d1, d2) in enumerate(dims):
        o[f"{pre}{3 * l}.weight"] = (d2, d1)                             # Linear(bias=False)
        if l < n - 1:                                                    # BatchNorm1d(affine) + ReLU
            o[f"{pre}{3 * l + 1}.weight"] = (d2,)
            o[f"{pre}{3 * l + 1}.bias"] = (d2,)
    return o


def _mlp_buffer_shapes(pre: str, dims) -> "OrderedDict[str, tuple]":
    o = OrderedDict()
    for l, (_, d2) in enumerate(dims):
        o[f"{pre}{3 * l + 1}.running_mean"] = (d2,)
        o[f"{pre}{3 * l + 1}.running_var"] = (d2,)
        o[f"{pre}{3 * l + 1}.num_batches_tracked"] = ()
    return o


def mlp_specs(cfg: DiGConfig) -> "OrderedDict[str, list]":
    """The five BN-MLPs of MoCo_ViT (modeling_pretrain_moco_mim_ori.py:366-369, 415-416)."""
    D = cfg.embed_dim
    o = OrderedDict()
    if cfg.use_moco:
        o["encoder_projection_layer."] = _mlp_dims(3, D, cfg.moco_mlp_dim, cfg.moco_dim)
        o["momentum_projection_layer."] = _mlp_dims(3, D, cfg.moco_mlp_dim, cfg.moco_dim)
        o["predictor."] = _mlp_dims(2, cfg.moco_dim, cfg.moco_mlp_dim, cfg.moco_dim)
    if cfg.use_moco and cfg.use_pixel:                                  # `if use_moco_target and use_pix_projector`, :412-420
        o["pix_projector."] = _mlp_dims(3, D, cfg.pix_mlp_dim, D)
        o["pix_projector_m."] = _mlp_dims(3, D, cfg.pix_mlp_dim, D)
    return o


def param_shapes(cfg: DiGConfig) -> "OrderedDict[str, tuple]":
    """All 356 (ViT-S) parameters in reference named_parameters() order."""
    o = OrderedDict()
    o.update(_encoder_param_shapes(cfg, "encoder."))
    if cfg.use_moco:
        o.update(_encoder_param_shapes(cfg, "momentum_encoder."))
    for pre, dims in mlp_specs(cfg).items():
        o.update(_mlp_param_shapes(pre, dims))
        if pre == "predictor." and cfg.patchnet == "regular":            # registration order of MoCo_ViT.__init__ (:366-394)
            o.update(_patchnet_param_shapes(cfg, "patch_extractor."))
            o.update(_patchnet_param_shapes(cfg, "momentum_patch_extractor."))
        if pre == "predictor." and cfg.patchnet == "conv":
            o.update(_convnet_param_shapes(cfg, "patch_extractor."))
            o.update(_convnet_param_shapes(cfg, "momentum_patch_extractor."))
    if cfg.use_pixel:
        Dd = cfg.dec_dim                                                  # pix_decoder, :422-426
        o["pix_decoder.0.weight"] = (Dd, cfg.embed_dim)
        o["pix_decoder.1.weight"] = (Dd, Dd)
        o["pix_decoder.2.weight"] = (Dd,)
        o["pix_decoder.2.bias"] = (Dd,)
        o["pix_decoder.4.weight"] = (cfg.dec_classes, Dd)
        o["pix_decoder.4.bias"] = (cfg.dec_classes,)
    return o


def buffer_shapes(cfg: DiGConfig) -> "OrderedDict[str, tuple]":
    o = OrderedDict()
    for pre, dims in mlp_specs(cfg).items():
        o.update(_mlp_buffer_shapes(pre, dims))
        if pre == "predictor." and cfg.patchnet == "conv":
            o.update(_convnet_buffer_shapes(cfg, "patch_extractor."))
            o.update(_convnet_buffer_shapes(cfg, "momentum_patch_extractor."))
    return o


MOMENTUM_PREFIXES = ("momentum_encoder.", "momentum_projection_layer.", "pix_projector_m.", "momentum_patch_extractor.")
EMA_PAIRS = (("encoder.", "momentum_encoder."),                         # _update_momentum_encoder, :428-442
             ("encoder_projection_layer.", "momentum_projection_layer."),
             ("patch_extractor.", "momentum_patch_extractor."),
             ("pix_projector.", "pix_projector_m."))


def is_trainable(name: str) -> bool:
    return not name.startswith(MOMENTUM_PREFIXES)                         # requires_grad=False, :396-420


def ema_pairs(names) -> List[Tuple[str, str]]:
    out = []
    for n in names:
        for src, dst in EMA_PAIRS:
            if n.startswith(src):
                out.append((n, dst + n[len(src):]))
    return out


def init_state(cfg: DiGConfig, seed: int = 0, dtype=torch.float32):
    """Initialise (P, S).  Distributions follow the reference's init (modeling_pretrain_vit.py:63-73 xavier
    Linear/zero bias/LN 1,0; modeling_pretrain_moco_mim_ori.py:353-355 patch-embed uniform; heads keep
    nn.Linear default kaiming-uniform(a=sqrt5) == U(-1/sqrt(fan_in), 1/sqrt(fan_in)); momentum = copies).
    The random stream is this file's own (the exact draws are not part of the contract; parity tests load
    explicit weights)."""
    g = torch.Generator().manual_seed(seed)
    P: Dict[str, torch.Tensor] = OrderedDict()

    def uni(shape, a):
        return (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1).mul_(a).to(dtype)

    for name, shp in param_shapes(cfg).items():
        if name.startswith(MOMENTUM_PREFIXES):
            continue
        if name.endswith("mask_token"):
            t = torch.zeros(shp, dtype=dtype)
        elif name.endswith("patch_embed.proj.weight"):
            if cfg.use_moco:                                              # :353-355 (inside `if use_moco_target`)
                t = uni(shp, math.sqrt(6.0 / float(3 * cfg.patch * cfg.patch + cfg.embed_dim)))
            else:                                                         # Gen-only: nn.Conv2d's own init, U(+-1/sqrt(fan_in))
                t = uni(shp, 1.0 / math.sqrt(cfg.in_chans * cfg.patch * cfg.patch))
        elif name.endswith("patch_embed.proj.bias") and not cfg.use_moco:
            t = uni(shp, 1.0 / math.sqrt(cfg.in_chans * cfg.patch * cfg.patch))
        elif name.startswith("patch_extractor.") and cfg.patchnet == "conv":
            # ConvPatchNet has no _init_weights: nn.Conv2d / nn.Linear defaults U(+-1/sqrt(fan_in)) for weight and bias, BatchNorm 1 / 0
            fan = shp[1] * 9 if len(shp) == 4 else shp[1] if len(shp) == 2 else None
            if fan is not None:
                t = uni(shp, 1.0 / math.sqrt(fan))
            elif name.endswith(".0.bias") or name.endswith(".3.bias"):
                w = param_shapes(cfg)[name[:-4] + "weight"]
                t = uni(shp, 1.0 / math.sqrt(w[1] * (9 if len(w) == 4 else 1)))
            elif name.endswith(".weight"):
                t = torch.ones(shp, dtype=dtype)
            else:
                t = torch.zeros(shp, dtype=dtype)
        elif name.startswith(("encoder.", "patch_extractor.")):           # (PatchNet._init_weights, :159-166: the same rule)
            if len(shp) == 2:
                t = uni(shp, math.sqrt(6.0 / (shp[0] + shp[1])))           # xavier_uniform_
            elif name.endswith(("norm1.weight", "norm2.weight", "norm.weight")):
                t = torch.ones(shp, dtype=dtype)
            else:
                t = torch.zeros(shp, dtype=dtype)
        elif name == "pix_decoder.2.weight":
            t = torch.ones(shp, dtype=dtype)
        elif name == "pix_decoder.2.bias":
            t = torch.zeros(shp, dtype=dtype)
        elif len(shp) == 2:
            t = uni(shp, 1.0 / math.sqrt(shp[1]))
        elif name == "pix_decoder.4.bias":
            t = uni(shp, 1.0 / math.sqrt(cfg.dec_dim))
        elif name.endswith(".weight"):                                    # BN gamma
            t = torch.ones(shp, dtype=dtype)
        else:                                                             # BN beta
            t = torch.zeros(shp, dtype=dtype)
        P[name] = t
    shapes = param_shapes(cfg)
    for src, dst in ema_pairs(list(P.keys())):
        if dst in shapes:
            P[dst] = P[src].clone()
    P = OrderedDict((k, P[k]) for k in shapes)                            # reference order
    S = OrderedDict()
    for name, shp in buffer_shapes(cfg).items():
        if name.endswith("running_var"):
            S[name] = torch.ones(shp, dtype=dtype)
        elif name.endswith("num_batches_tracked"):
            S[name] = torch.zeros((), dtype=torch.int64)
        else:
            S[name] = torch.zeros(shp, dtype=dtype)
    return P, S


# --------------------------------------------------------------------------------------------------
# communication shim (world_size 1, or torch.distributed for the gloo multi-rank tests)
# --------------------------------------------------------------------------------------------------
class LocalComm:
    rank = 0
    world = 1

    def all_reduce_sum(self, t):          # differentiable identity
        return t

    def all_gather_nograd(self, t):
        return t


class DistComm:
    """SyncBatchNorm statistics / key all-gather semantics over torch.distributed (gloo on CPU)."""

    def __init__(self):
        import torch.distributed as dist
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()

    def all_reduce_sum(self, t):
        import torch.distributed.nn.functional as dnf
        return dnf.all_reduce(t)

    def all_gather_nograd(self, t):
        import torch.distributed as dist
        outs = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(outs, t.detach().contiguous())
        return torch.cat(outs, 0)                                         # rank order, :586-590


# --------------------------------------------------------------------------------------------------
# model math
# --------------------------------------------------------------------------------------------------
def sinusoid_table(n_pos: int, d: int) -> torch.Tensor:
    """get_sinusoid_encoding_table, modeling_finetune.py:200-210 (float64 then cast)."""
    pos = np.arange(n_pos, dtype=np.float64)[:, None]
    j = np.arange(d)
    ang = pos / np.power(10000.0, 2.0 * (j // 2) / d)[None, :]
    tab = np.empty_like(ang)
    tab[:, 0::2] = np.sin(ang[:, 0::2])
    tab[:, 1::2] = np.cos(ang[:, 1::2])
    return torch.from_numpy(tab).to(torch.float32)


def patchify_cpp(x: torch.Tensor, p: int) -> torch.Tensor:
    """[B,C,H,W] -> [B, (h w), (c p1 p2)] : the im2col of a stride-p, kernel-p conv (modeling_finetune.py:195)."""
    B, C, H, W = x.shape
    x = x.reshape(B, C, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, (H // p) * (W // p), C * p * p)


def patchify_ppc(x: torch.Tensor, p: int) -> torch.Tensor:
    """'b c (h p1) (w p2) -> b (h w) (p1 p2 c)', engine_for_pretraining_moco.py:96."""
    B, C, H, W = x.shape
    x = x.reshape(B, C, H // p, p, W // p, p).permute(0, 2, 4, 3, 5, 1)
    return x.reshape(B, (H // p) * (W // p), p * p * C)


def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def attention(x, P, pre, cfg: DiGConfig, taps=None):
    """Attention.forward, modeling_finetune.py:87-120 (K has zero bias; q scaled before q@k^T)."""
    Bn, N, D = x.shape
    H, dh = cfg.heads, cfg.head_dim
    bias = torch.cat([P[pre + "q_bias"], torch.zeros_like(P[pre + "v_bias"]), P[pre + "v_bias"]])
    qkv = F.linear(x, P[pre + "qkv.weight"], bias).reshape(Bn, N, 3, H, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (dh ** -0.5), qkv[1], qkv[2]
    s = q @ k.transpose(-2, -1)
    a = s.softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(Bn, N, D)
    if taps is not None:
        taps[pre + "ctx"] = o
    return F.linear(o, P[pre + "proj.weight"], P[pre + "proj.bias"])


MOMENTUM_SITE_OFFSET = 64        # drop-path sites of the momentum encoder's block i: those of block i + 64 (its own, independent draws)


def block(x, P, pre, cfg: DiGConfig, taps=None, path=None):
    """Block.forward with init_values=0 (no layer scale): modeling_finetune.py:150-158.  path: None (drop_path = 0, the README recipe) or
    (attention-branch mask fn, MLP-branch mask fn) -- `x + self.drop_path(...)` with the per-sample keep / scale of timm's drop_path."""
    pa, pm = path if path is not None else ((lambda t: t), (lambda t: t))
    x = x + pa(attention(layer_norm(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"], cfg.ln_eps), P, pre + "attn.", cfg, taps))
    h = F.linear(layer_norm(x, P[pre + "norm2.weight"], P[pre + "norm2.bias"], cfg.ln_eps),
                 P[pre + "mlp.fc1.weight"], P[pre + "mlp.fc1.bias"])
    h = F.gelu(h)                                                         # nn.GELU() exact erf, :43-60
    return x + pm(F.linear(h, P[pre + "mlp.fc2.weight"], P[pre + "mlp.fc2.bias"]))


def encoder(P, pre, images, mask, cfg: DiGConfig, taps=None, drop=None):
    """PretrainVisionTransformerEncoder.forward_features, modeling_pretrain_vit.py:89-106 (norm/head = Identity,
    modeling_pretrain_moco_mim_ori.py:362-363).  mask: bool [Bn, N] (True = replaced by mask_token).
    drop: None, or a finetune_oracle.DropOracle (--drop_path > 0, run_mae_pretraining_moco.py:87: stochastic depth with rate
    linspace(0, rate, depth)[i] on both branches of block i, modeling_pretrain_vit.py:50-56; keyed masks as the device draws them --
    the momentum encoder, which the reference also runs in train mode, draws its own under the sites of block i + 64)."""
    w = P[pre + "patch_embed.proj.weight"]
    x = patchify_cpp(images, cfg.patch) @ w.reshape(w.shape[0], -1).t() + P[pre + "patch_embed.proj.bias"]
    if mask is not None:
        m = mask.unsqueeze(-1).to(x.dtype)
        x = x * (1.0 - m) + P[pre + "mask_token"].expand(x.shape[0], x.shape[1], -1) * m
    x = x + sinusoid_table(cfg.num_patches, cfg.embed_dim).to(x.dtype)[None]
    if taps is not None:
        taps[pre + "embed"] = x
    for i in range(cfg.depth):
        path = None
        if drop is not None and drop.dpr[i]:
            import finetune_oracle as FO
            j = i + (MOMENTUM_SITE_OFFSET if pre.startswith("momentum_") else 0)
            path = (lambda t, j=j, i=i: drop.path(FO.enc_site(j, 2), t, drop.dpr[i]), lambda t, j=j, i=i: drop.path(FO.enc_site(j, 4), t, drop.dpr[i]))
        x = block(x, P, f"{pre}blocks.{i}.", cfg, taps, path)
        if taps is not None:
            taps[f"{pre}blocks.{i}"] = x
    if not cfg.use_moco:                                                  # x = self.norm(x), modeling_pretrain_vit.py:104 (Gen-only)
        x = layer_norm(x, P[pre + "norm.weight"], P[pre + "norm.bias"], cfg.ln_eps)
    return x


def batch_norm_train(x, gamma, beta, S, pre, cfg: DiGConfig, comm, update_buffers=True):
    """nn.BatchNorm1d in train mode; under SyncBatchNorm the statistics are over the rows of all ranks
    (run_mae_pretraining_moco.py:390).  Biased variance normalises; the running buffer gets n/(n-1)."""
    x = x.float()                                                         # (a no-op in fp32; under the tests' bf16-autocast yardstick: batch_norm is an
    n_local = x.shape[0]                                                  #  fp32 op of torch's autocast too -- E[x^2] - mean^2 from bf16 sums goes negative)
    stats = torch.cat([x.sum(0), (x * x).sum(0), x.new_tensor([float(n_local)])])
    stats = comm.all_reduce_sum(stats)
    C = x.shape[1]
    n = stats[-1]
    mean = stats[:C] / n
    var = stats[C:2 * C] / n - mean * mean
    y = (x - mean) * torch.rsqrt(var + cfg.bn_eps)
    if gamma is not None:
        y = y * gamma + beta
    if update_buffers and S is not None:
        with torch.no_grad():
            mom = cfg.bn_momentum
            S[pre + "running_mean"].mul_(1 - mom).add_(mean.detach(), alpha=mom)
            S[pre + "running_var"].mul_(1 - mom).add_(var.detach() * (n / (n - 1)), alpha=mom)
            S[pre + "num_batches_tracked"] += 1
    return y


def bn_mlp(x, P, S, pre, dims, cfg: DiGConfig, comm, taps=None):
    """_build_mlp stack: Linear(no bias) -> BN -> ReLU ... last BN affine=False (:463-482)."""
    n = len(dims)
    for l in range(n):
        x = F.linear(x, P[f"{pre}{3 * l}.weight"])
        if l < n - 1:
            x = batch_norm_train(x, P[f"{pre}{3 * l + 1}.weight"], P[f"{pre}{3 * l + 1}.bias"], S, f"{pre}{3 * l + 1}.", cfg, comm)
            x = F.relu(x)
        else:
            x = batch_norm_train(x, None, None, S, f"{pre}{3 * l + 1}.", cfg, comm)
        if taps is not None:
            taps[f"{pre}{3 * l}"] = x
    return x


def window_pool(x, cfg: DiGConfig):
    """PatchNet.forward with use_patch_transformer=False (:189-193): adaptive_avg_pool2d of the token grid
    to (1, num_windows)."""
    Bn, N, C = x.shape
    gh, gw = cfg.grid
    if gw % cfg.num_windows:
        # uneven windows (the argparse default --num_windows 5 on 32 columns, run_mae_pretraining_moco.py:143): adaptive_avg_pool2d's
        # bins [floor(i gw / nw), ceil((i + 1) gw / nw)) overlap by a column
        y = F.adaptive_avg_pool2d(x.reshape(Bn, gh, gw, C).permute(0, 3, 1, 2), (1, cfg.num_windows))
        return y.reshape(Bn, C, cfg.num_windows).permute(0, 2, 1)
    return x.reshape(Bn, gh, cfg.num_windows, gw // cfg.num_windows, C).mean(dim=(1, 3))


def conv_patch_extractor(x, P, S, pre, cfg: DiGConfig, comm):
    """ConvPatchNet.forward (modeling_pretrain_moco_mim_ori.py:250-258): the token grid as a [C, 8, 32] map through conv3x3 / BatchNorm2d / ReLU
    (x 4) with 2x2 max-pools between them (8x32 -> 4x16 -> 2x8 -> 1x4), adaptive_avg_pool2d to (1, num_windows), flattened window-major, then
    patches2global.  BatchNorm2d in train mode = per-channel statistics over (batch, y, x): batch_norm_train on the [B H W, C] rows (under
    SyncBatchNorm over all ranks, run_mae_pretraining_moco.py:390).  x: [Bn, N, C] -> [Bn, 1, C]."""
    Bn, N, C = x.shape
    gh, gw = cfg.grid
    h = x.reshape(Bn, gh, gw, C).permute(0, 3, 1, 2)
    for j, i in enumerate(CONV_IDX):
        b = f"{pre}conv_layers.{i}."
        h = F.conv2d(h, P[b + "0.weight"], P[b + "0.bias"], stride=1, padding=1)
        _, c_, hh, ww = h.shape
        r = batch_norm_train(h.permute(0, 2, 3, 1).reshape(-1, c_), P[b + "1.weight"], P[b + "1.bias"], S, b + "1.", cfg, comm)
        h = F.relu(r).reshape(Bn, hh, ww, c_).permute(0, 3, 1, 2)
        if j < 3:
            h = F.max_pool2d(h, kernel_size=2, stride=2)
    h = F.adaptive_avg_pool2d(h, (1, cfg.num_windows)).permute(0, 2, 3, 1).reshape(Bn, -1)
    g = f"{pre}patches2global."
    z = F.linear(h, P[g + "0.weight"], P[g + "0.bias"])
    z = F.relu(batch_norm_train(z, P[g + "1.weight"], P[g + "1.bias"], S, g + "1.", cfg, comm))
    z = F.linear(z, P[g + "3.weight"], P[g + "3.bias"])
    z = batch_norm_train(z, None, None, S, g + "4.", cfg, comm)
    return z.unsqueeze(1)


def patch_extractor(x, P, pre, cfg: DiGConfig, S=None, comm=None):
    """PatchNet.forward (modeling_pretrain_moco_mim_ori.py:189-205): the pooled windows; with the patch transformer ("regular") each of them
    then attends over all tokens of its image through `depth` Blocks (:88-135) -- every Block normalises queries AND keys / values with its
    own norm1, adds the attention output to the NORMALISED queries (`x = self.norm1(x) ... x = x + attn_x`, :107-121), then the MLP; a final
    LayerNorm.  x: [Bn, N, C] -> [Bn, num_windows, C]."""
    if cfg.patchnet == "conv":
        return conv_patch_extractor(x, P, S, pre, cfg, comm or LocalComm())
    pooled = window_pool(x, cfg)
    if cfg.patchnet != "regular":
        return pooled
    Bn, N, C = x.shape
    H, dh, eps = cfg.heads, cfg.head_dim, cfg.patchnet_eps
    y = pooled
    for i in range(cfg.patchnet_depth):
        b = f"{pre}blocks.{i}."
        yn = layer_norm(y, P[b + "norm1.weight"], P[b + "norm1.bias"], eps)
        kn = layer_norm(x, P[b + "norm1.weight"], P[b + "norm1.bias"], eps)
        q = F.linear(yn, P[b + "attn.linear_q.weight"]).reshape(Bn, -1, H, dh).permute(0, 2, 1, 3) * (dh ** -0.5)
        k = F.linear(kn, P[b + "attn.linear_k.weight"]).reshape(Bn, N, H, dh).permute(0, 2, 3, 1)
        v = F.linear(kn, P[b + "attn.linear_v.weight"]).reshape(Bn, N, H, dh).permute(0, 2, 1, 3)
        a = ((q @ k).softmax(dim=-1) @ v).transpose(1, 2).reshape(Bn, -1, C)
        y = yn + F.linear(a, P[b + "attn.proj.weight"], P[b + "attn.proj.bias"])
        h = F.gelu(F.linear(layer_norm(y, P[b + "norm2.weight"], P[b + "norm2.bias"], eps), P[b + "mlp.fc1.weight"], P[b + "mlp.fc1.bias"]))
        y = y + F.linear(h, P[b + "mlp.fc2.weight"], P[b + "mlp.fc2.bias"])
    return layer_norm(y, P[pre + "norm.weight"], P[pre + "norm.bias"], eps)


def info_nce(q, k, T, comm):
    """contrastive_loss + accuracy + label_smooth_loss(smoothing=0): :444-461, :593-625."""
    q = F.normalize(q, dim=1)
    k = F.normalize(k, dim=1)
    k = comm.all_gather_nograd(k)
    logits = q @ k.t() / T
    n = logits.shape[0]
    labels = torch.arange(n, dtype=torch.long) + n * comm.rank
    logp = logits.log_softmax(dim=1)
    loss = -(logp.gather(1, labels[:, None]).squeeze(1)).mean() * (2 * T)
    with torch.no_grad():
        top = logits.topk(min(5, logits.shape[1]), 1, True, True)[1]
        hit = top.eq(labels[:, None])
        acc1 = hit[:, :1].float().sum() * (100.0 / n)
        acc5 = hit[:, :5].float().sum() * (100.0 / n)
    return loss, acc1, acc5


def ema_update(P, m: float):
    """_update_momentum_encoder, :428-442 (parameters only; BN buffers are not EMA'd)."""
    with torch.no_grad():
        for src, dst in ema_pairs([n for n in P if is_trainable(n)]):
            if dst not in P:                                              # (Gen-only models have no momentum networks)
                continue
            P[dst].copy_(P[dst] * m + P[src].detach() * (1.0 - m))


def model_forward(P, S, images, aug_images, mask, m: float, cfg: DiGConfig, comm=None,
                  only_mim_on_ori_img: bool = True, taps=None, do_ema: bool = True, drop=None):
    """MoCo_ViT.forward, modeling_pretrain_moco_mim_ori.py:488-577.  mask: bool [B, num_view, N] (ignored by the Dis-only models:
    `if not self.use_pixel_target: vis_mask_pos = None`, :493-494)."""
    comm = comm or LocalComm()
    B = images.shape[0]
    N, D = cfg.num_patches, cfg.embed_dim
    specs = mlp_specs(cfg)
    allim = torch.cat([images, aug_images], 0)
    num_view = mask.shape[1]
    mflat = mask.permute(1, 0, 2).reshape(-1, N) if cfg.use_pixel else None     # rows 0..B-1 = view 0 (:497)
    enc = encoder(P, "encoder.", allim, mflat, cfg, taps, drop)
    has_pp = "pix_projector." in specs                                    # hasattr(self, 'pix_projector'), :500
    out = {}
    if cfg.use_moco:
        if has_pp:
            masked = bn_mlp(enc[:B].reshape(B * N, D), P, S, "pix_projector.", specs["pix_projector."], cfg, comm, taps)
            feat = torch.cat([masked.reshape(B, N, D), enc[B:]], 0)
        else:
            feat = enc
        pooled = patch_extractor(feat, P, "patch_extractor.", cfg, S, comm).reshape(2 * B * cfg.n_patch, D)
        qs = bn_mlp(pooled, P, S, "encoder_projection_layer.", specs["encoder_projection_layer."], cfg, comm, taps)
        qs = bn_mlp(qs, P, S, "predictor.", specs["predictor."], cfg, comm, taps)
        half = B * cfg.n_patch
        q1, q2 = qs[:half], qs[half:]
        with torch.no_grad():
            if do_ema:
                ema_update(P, m)
            enc_m = encoder(P, "momentum_encoder.", allim, mflat, cfg, drop=drop)
            if has_pp:
                masked_m = bn_mlp(enc_m[:B].reshape(B * N, D), P, S, "pix_projector_m.", specs["pix_projector_m."], cfg, comm)
                feat_m = torch.cat([masked_m.reshape(B, N, D), enc_m[B:]], 0)
            else:
                feat_m = enc_m
            pooled_m = patch_extractor(feat_m, P, "momentum_patch_extractor.", cfg, S, comm).reshape(2 * B * cfg.n_patch, D)
            ks = bn_mlp(pooled_m, P, S, "momentum_projection_layer.", specs["momentum_projection_layer."], cfg, comm)
            k1, k2 = ks[:half], ks[half:]
        l1, a11, a15 = info_nce(q1, k2, cfg.T, comm)
        l2, a21, a25 = info_nce(q2, k1, cfg.T, comm)
        out.update({"contra_loss": l1 + l2, "q1_acc1": a11, "q1_acc5": a15, "q2_acc1": a21, "q2_acc5": a25})
        if taps is not None:
            taps.update(q1=q1, q2=q2, k1=k1, k2=k2, feat=feat, pooled=pooled)
    if cfg.use_pixel:
        h = F.linear(enc, P["pix_decoder.0.weight"])                      # :422-426, on raw encoder output (:561)
        h = F.linear(h, P["pix_decoder.1.weight"])
        h = F.gelu(layer_norm(h, P["pix_decoder.2.weight"], P["pix_decoder.2.bias"], cfg.ln_eps))
        dec = F.linear(h, P["pix_decoder.4.weight"], P["pix_decoder.4.bias"])
        C = dec.shape[-1]
        views = range(1) if only_mim_on_ori_img else range(num_view)
        out["vis_out"] = [dec[v * B:(v + 1) * B][mflat[v * B:(v + 1) * B]].reshape(B, -1, C) for v in views]
        if taps is not None:
            taps.update(dec=dec)
    if taps is not None:
        taps.update(enc=enc)
    return out


def mim_targets(images, mask_f, cfg: DiGConfig, num_view: int = 2, only_mim_on_ori_img: bool = True, normlize_target: bool = False):
    """engine_for_pretraining_moco.py:80-111 (normlize_target=False is the pretrain default, run_mae…:90; True: :88-93, every patch is
    standardised per channel over its p1*p2 pixels with the unbiased variance).
    mask_f: the loader's [B, num_view, N] float64 0/1 array.  Returns (bool mask [B,num_view,N], [labels])."""
    B = images.shape[0]
    mask = mask_f.flatten(1).to(torch.bool).view(B, num_view, -1).clone()
    if only_mim_on_ori_img:
        mask[:, 1:, :] = False
    patches = patchify_ppc(images * 0.5 + 0.5, cfg.patch)
    if normlize_target:
        sq = patches.reshape(B, patches.shape[1], cfg.patch * cfg.patch, -1)              # 'b n (p1 p2) c'
        sq = (sq - sq.mean(dim=-2, keepdim=True)) / (sq.var(dim=-2, unbiased=True, keepdim=True).sqrt() + 1e-6)
        patches = sq.reshape(B, patches.shape[1], -1)
    C = patches.shape[-1]
    labels = [patches[mask[:, v]].reshape(B, -1, C) for v in range(num_view)]
    return mask, labels


# --------------------------------------------------------------------------------------------------
# schedules / optimizer
# --------------------------------------------------------------------------------------------------
def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0,
                     warmup_steps=-1):
    """utils/utils.py:522-538, including its quirk: warm-up only when warmup_epochs > 0."""
    warm = np.array([])
    warm_iters = warmup_epochs * niter_per_ep
    if warmup_steps > 0:
        warm_iters = warmup_steps
    if warmup_epochs > 0:
        warm = np.linspace(start_warmup_value, base_value, warm_iters)
    it = np.arange(epochs * niter_per_ep - warm_iters)
    sched = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * it / len(it)))
    sched = np.concatenate((warm, sched))
    assert len(sched) == epochs * niter_per_ep
    return sched


def adjust_moco_momentum(epoch_f: float, epochs: int, moco_m: float) -> float:
    """utils/utils.py:540-543."""
    return 1.0 - 0.5 * (1.0 + math.cos(math.pi * epoch_f / epochs)) * (1.0 - moco_m)


def contrast_weights(epoch, start_epoch, warmup_steps, weight, iters):
    """engine_for_pretraining_moco.py:48-56."""
    if epoch == start_epoch:
        ws = min(warmup_steps, iters)
        w = np.linspace(0.0, weight, ws)
        if ws < iters:
            w = np.hstack([w, np.ones(iters - ws) * weight])
        return w
    if epoch > start_epoch:
        return np.ones(iters) * weight
    return np.zeros(iters)


def param_groups(P, weight_decay: float, skip=("pos_embed", "cls_token")):
    """get_parameter_groups, optim_factory.py:57-100 -> [decay names], [no_decay names] (lr_scale = 1)."""
    decay, no_decay = [], []
    for n, p in P.items():
        if not is_trainable(n):
            continue
        if p.ndim == 1 or n.endswith(".bias") or n in skip:
            no_decay.append(n)
        else:
            decay.append(n)
    return decay, no_decay


def adamw_update(p, g, m, v, step, lr, wd, beta1=0.9, beta2=0.999, eps=1e-8):
    """custom_optim/_functional.py:115-140 (in place)."""
    p.mul_(1 - lr * wd)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def random_masks(B: int, cfg: DiGConfig, mask_ratio: float, rng: np.random.RandomState, num_view: int = 2):
    """RandomMaskingGenerator.__call__, masking_generator.py:27-45 : float64 [B, num_view, N], exactly
    int(mask_ratio*N) ones per view, an independent shuffle per view."""
    N = cfg.num_patches
    k = int(mask_ratio * N)
    out = np.zeros((B, num_view, N), dtype=np.float64)
    for b in range(B):
        for v in range(num_view):
            row = np.hstack([np.zeros(N - k), np.ones(k)])
            rng.shuffle(row)
            out[b, v] = row
    return torch.from_numpy(out)


# --------------------------------------------------------------------------------------------------
# one engine step
# --------------------------------------------------------------------------------------------------
@dataclass
class StepHyper:
    lr: float = 1.5e-4
    weight_decay: float = 0.1
    moco_m: float = 0.99
    w_contrast: float = 0.1
    w_pixel: float = 1.0
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    only_mim_on_ori_img: bool = True
    clip_grad: Optional[float] = None
    drop_path: float = 0.0            # --drop_path (run_mae_pretraining_moco.py:87); masks keyed by (drop_seed, step): finetune_oracle.DropOracle
    drop_seed: int = 0


class OracleTrainer:
    """State + one `train_one_epoch` step (engine_for_pretraining_moco.py:58-186) in fp32 on CPU."""

    def __init__(self, cfg: DiGConfig, P=None, S=None, seed: int = 0, comm=None):
        self.cfg = cfg
        if P is None:
            P, S = init_state(cfg, seed)
        self.P, self.S = P, S
        self.comm = comm or LocalComm()
        self.exp_avg = {n: torch.zeros_like(p) for n, p in P.items() if is_trainable(n)}
        self.exp_avg_sq = {n: torch.zeros_like(p) for n, p in P.items() if is_trainable(n)}
        self.step_count = 0
        self.decay, self.no_decay = param_groups(P, 1.0)

    def loss_and_grads(self, images, aug_images, mask_f, hp: StepHyper, taps=None):
        cfg = self.cfg
        mask, labels = mim_targets(images, mask_f, cfg, mask_f.shape[1], hp.only_mim_on_ori_img)
        train = [n for n in self.P if is_trainable(n)]
        for n in train:
            self.P[n].requires_grad_(True)
            self.P[n].grad = None
        drop = None
        if hp.drop_path:
            import finetune_oracle as FO
            drop = FO.DropOracle(hp.drop_seed, self.step_count, drop_path=hp.drop_path, depth=cfg.depth)
        out = model_forward(self.P, self.S, images, aug_images, mask, hp.moco_m, cfg, self.comm,
                            hp.only_mim_on_ori_img, taps, drop=drop)
        nv = 1 if hp.only_mim_on_ori_img else mask_f.shape[1]
        # engine_for_pretraining_moco.py:119-144: each term only `if '<key>' in out_dict` (the single-objective models leave one out)
        loss = 0.0
        loss_pixel = None
        if "contra_loss" in out:
            loss = loss + out["contra_loss"] * hp.w_contrast
        if "vis_out" in out:
            loss_pixel = sum((1.0 / nv) * F.mse_loss(out["vis_out"][i], labels[i]) for i in range(nv))
            loss = loss + loss_pixel * hp.w_pixel
        loss.backward()
        grads = {}
        self.never_grad = set()               # `p.grad is None`: the reference's AdamW skips such a parameter altogether (custom_optim/adamw.py:78-79:
        for n in train:                       #  no decay, no state) -- Dis-only's encoder.mask_token, which its unmasked encoder never reads
            g = self.P[n].grad
            if g is None:
                self.never_grad.add(n)
            grads[n] = torch.zeros_like(self.P[n]) if g is None else g.detach()
            self.P[n].requires_grad_(False)
            self.P[n].grad = None
        if self.comm.world > 1:                                           # DDP gradient averaging (C1)
            import torch.distributed as dist
            flat = torch.cat([grads[n].reshape(-1) for n in train])
            dist.all_reduce(flat)
            flat /= self.comm.world
            o = 0
            for n in train:
                k = grads[n].numel()
                grads[n] = flat[o:o + k].view_as(grads[n])
                o += k
        metrics = {"loss": float(loss.detach())}
        if loss_pixel is not None:
            metrics["loss_pixel"] = float(loss_pixel.detach())
        if "contra_loss" in out:
            metrics.update({"loss_contrast": float(out["contra_loss"].detach()),
                            "q1_acc1": float(out["q1_acc1"]), "q1_acc5": float(out["q1_acc5"]),
                            "q2_acc1": float(out["q2_acc1"]), "q2_acc5": float(out["q2_acc5"])})
        gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values()))       # get_grad_norm_, utils.py:507-519
        metrics["grad_norm"] = float(gn)
        return metrics, grads, out, labels

    def step(self, images, aug_images, mask_f, hp: StepHyper, taps=None):
        metrics, grads, out, labels = self.loss_and_grads(images, aug_images, mask_f, hp, taps)
        if hp.clip_grad is not None:
            coef = min(1.0, hp.clip_grad / (metrics["grad_norm"] + 1e-6))
            for g in grads.values():
                g.mul_(coef)
        self.step_count += 1
        with torch.no_grad():
            for names, wd in ((self.decay, hp.weight_decay), (self.no_decay, 0.0)):
                for n in names:
                    if n in self.never_grad:
                        continue
                    adamw_update(self.P[n], grads[n], self.exp_avg[n], self.exp_avg_sq[n], self.step_count,
                                 hp.lr, wd, hp.beta1, hp.beta2, hp.eps)
        return metrics, grads, out, labels


def synthetic_batch(B: int, cfg: DiGConfig, seed: int, mask_ratio: float = 0.7):
    """SURVEY.md §8(d) synthetic inputs: U(-1,1) crops, fresh 179/256 masks."""
    g = torch.Generator().manual_seed(seed)
    images = torch.rand((B, 3, cfg.img_h, cfg.img_w), generator=g) * 2 - 1
    aug = torch.rand((B, 3, cfg.img_h, cfg.img_w), generator=g) * 2 - 1
    masks = random_masks(B, cfg, mask_ratio, np.random.RandomState(seed))
    return images, aug, masks


# --------------------------------------------------------------------------------------------------
# deterministic closed-form state for fixtures (so no weight files need to be committed)
# --------------------------------------------------------------------------------------------------
def det_tensor(name: str, shape, seed: int, scale: float, offset: float = 0.0) -> torch.Tensor:
    """Reproducible N(0,1)*scale+offset tensor keyed by (name, seed); numpy's legacy RandomState stream is
    frozen by numpy's compatibility policy, so generator and replay sides agree bit-for-bit."""
    import zlib
    rs = np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    a = rs.standard_normal(tuple(shape) if len(shape) else (1,)).astype(np.float64) * scale + offset
    return torch.from_numpy(a.astype(np.float32)).reshape(tuple(shape))


def det_state(cfg: DiGConfig, seed: int = 0, momentum_delta: float = 0.01):
    """(P, S) with every tensor a closed-form function of (name, seed).  Momentum copies are the online
    tensors plus a small deterministic perturbation so the EMA and key branch are exercised."""
    P = OrderedDict()
    for name, shp in param_shapes(cfg).items():
        if name.startswith(MOMENTUM_PREFIXES):
            continue
        if len(shp) >= 2 and not name.endswith("mask_token"):
            fan_in = int(np.prod(shp[1:]))
            t = det_tensor(name, shp, seed, 1.0 / math.sqrt(fan_in))
        elif name.endswith("mask_token"):
            t = det_tensor(name, shp, seed, 0.02)
        elif name.endswith(".weight"):                                    # LN / BN gamma
            t = det_tensor(name, shp, seed, 0.1, 1.0)
        else:                                                             # biases, betas, q_bias, v_bias
            t = det_tensor(name, shp, seed, 0.02)
        P[name] = t
    shapes = param_shapes(cfg)
    for src, dst in ema_pairs(list(P.keys())):
        if dst not in shapes:
            continue
        P[dst] = P[src] + det_tensor(dst, P[src].shape, seed, momentum_delta * float(P[src].std() if P[src].numel() > 1 else 1.0))
    P = OrderedDict((k, P[k].contiguous()) for k in shapes)
    S = OrderedDict()
    for name, shp in buffer_shapes(cfg).items():
        if name.endswith("running_var"):
            S[name] = torch.ones(shp)
        elif name.endswith("num_batches_tracked"):
            S[name] = torch.zeros((), dtype=torch.int64)
        else:
            S[name] = torch.zeros(shp)
    return P, S


TINY = dict(embed_dim=128, depth=2, heads=2, dec_dim=64, moco_dim=64, moco_mlp_dim=256)   # GPU-kernel-compatible tiny model
