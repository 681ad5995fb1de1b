"""The C-ABI kernels as PyTorch custom operators (`torch.library`), for a caller that keeps the reference's `nn.Module`s and swaps single ops.

`import dig_amd.torch_ops` registers, in the namespace `dig`:

    dig::linear(x, w, bias)                         y = x w^T + bias                       nn.Linear (modeling_finetune.py:43-60, 87-125)
    dig::layer_norm(x, weight, bias, eps)           (y, mean, rstd)                        nn.LayerNorm (modeling_finetune.py:134,140)
    dig::attention(qkv, n_img, heads)               (ctx, lse): softmax(q k^T) v per head  Attention.forward :97-118 (q pre-scaled, 256 tokens, head_dim 64)
    dig::mlp_block(x, ln_w, ln_b, eps, w1, b1, w2, b2)   x + fc2(gelu(fc1(LN(x))))         Block.forward :156-158 in ONE forward launch

each with a fake (meta) implementation -- so they trace under `torch.compile` / `make_fx` as opaque calls -- and an autograd formula whose
backward is again made of registered operators (`dig::*_bwd`), i.e. visible to the dispatcher all the way.  Tensors are bf16 activations /
weights and fp32 LayerNorm parameters and biases, as everywhere in this package; weight gradients come back in the weight's dtype.  The
pre-training step itself is dispatched as two registered operators of its own, `dig::pretrain_step_fwd` / `dig::pretrain_step_bwd`
(`dig_amd/engine_core.py`: one node over flat arenas, which is what makes its two-stream backward and bucketed collectives possible); the
operators below are the "one op at a time" integration level of INTEGRATION.md section 2.  There is no CPU implementation: the operators are registered for CUDA (= HIP) tensors only.
"""
from typing import Optional, Tuple

import torch
from torch.library import custom_op

from . import ops

BF16, F32 = torch.bfloat16, torch.float32
Tensor = torch.Tensor


def _cont(t):
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------------------------- linear
@custom_op("dig::linear", mutates_args=(), device_types="cuda")
def linear(x: Tensor, w: Tensor, bias: Optional[Tensor]) -> Tensor:
    return ops.linear_fwd(_cont(x), _cont(w), bias=bias)


@linear.register_fake
def _(x, w, bias):
    return x.new_empty((x.shape[0], w.shape[0]))


@custom_op("dig::linear_bwd", mutates_args=(), device_types="cuda")
def linear_bwd(dy: Tensor, x: Tensor, w: Tensor, need_bias: bool) -> Tuple[Tensor, Tensor, Tensor]:
    dy, x, w = _cont(dy), _cont(x), _cont(w)
    dx = ops.linear_dgrad(dy, w)
    dw = torch.zeros(w.shape, device=w.device, dtype=F32)
    ops.linear_wgrad(dy, x, dw)
    db = torch.zeros(w.shape[0], device=w.device, dtype=F32)
    if need_bias:
        ops.colsum(dy, db)
    return dx, dw.to(w.dtype), db


@linear_bwd.register_fake
def _(dy, x, w, need_bias):
    return x.new_empty(x.shape), w.new_empty(w.shape), w.new_empty((w.shape[0],), dtype=F32)


def _linear_setup(ctx, inputs, output):
    x, w, bias = inputs
    ctx.save_for_backward(x, w)
    ctx.has_bias = bias is not None


def _linear_backward(ctx, dy):
    x, w = ctx.saved_tensors
    dx, dw, db = torch.ops.dig.linear_bwd(dy, x, w, ctx.has_bias)
    return dx, dw, (db if ctx.has_bias else None)


linear.register_autograd(_linear_backward, setup_context=_linear_setup)


# ---------------------------------------------------------------------------------------------------------------- layer norm
@custom_op("dig::layer_norm", mutates_args=(), device_types="cuda")
def layer_norm(x: Tensor, weight: Tensor, bias: Tensor, eps: float) -> Tuple[Tensor, Tensor, Tensor]:
    return ops.layernorm_fwd(_cont(x), weight, bias, eps)


@layer_norm.register_fake
def _(x, weight, bias, eps):
    return x.new_empty(x.shape), x.new_empty((x.shape[0],), dtype=F32), x.new_empty((x.shape[0],), dtype=F32)


@custom_op("dig::layer_norm_bwd", mutates_args=(), device_types="cuda")
def layer_norm_bwd(dy: Tensor, x: Tensor, weight: Tensor, bias: Tensor, mean: Tensor, rstd: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    dg = torch.zeros_like(weight)
    db = torch.zeros_like(bias)
    dx = ops.layernorm_bwd(_cont(dy), _cont(x), weight, bias, mean, rstd, None, dg, db)
    return dx, dg, db


@layer_norm_bwd.register_fake
def _(dy, x, weight, bias, mean, rstd):
    return x.new_empty(x.shape), weight.new_empty(weight.shape), bias.new_empty(bias.shape)


def _ln_setup(ctx, inputs, output):
    x, weight, bias, _ = inputs
    _, mean, rstd = output
    ctx.save_for_backward(x, weight, bias, mean, rstd)
    ctx.mark_non_differentiable(mean, rstd)


def _ln_backward(ctx, dy, _dmean, _drstd):
    x, weight, bias, mean, rstd = ctx.saved_tensors
    dx, dg, db = torch.ops.dig.layer_norm_bwd(dy, x, weight, bias, mean, rstd)
    return dx, dg, db, None


layer_norm.register_autograd(_ln_backward, setup_context=_ln_setup)


# ---------------------------------------------------------------------------------------------------------------- attention
@custom_op("dig::attention", mutates_args=(), device_types="cuda")
def attention(qkv: Tensor, n_img: int, heads: int) -> Tuple[Tensor, Tensor]:
    """qkv: bf16 [n_img * 256, 3 * D] with q already scaled by head_dim ** -0.5 (the qkv GEMM epilogue does it); returns the context rows
    [n_img * 256, D] and the per-row log-sum-exp [n_img * heads, 256] (fp32) the backward recomputes the probabilities from."""
    return ops.attn_fwd(_cont(qkv), n_img, heads, qkv.shape[1] // 3)


@attention.register_fake
def _(qkv, n_img, heads):
    return qkv.new_empty((qkv.shape[0], qkv.shape[1] // 3)), qkv.new_empty((n_img * heads, 256), dtype=F32)


@custom_op("dig::attention_bwd", mutates_args=(), device_types="cuda")
def attention_bwd(dctx: Tensor, qkv: Tensor, ctx_rows: Tensor, lse: Tensor, n_img: int, heads: int) -> Tensor:
    # scale 1.0: the gradient with respect to the (already scaled) q this operator received
    return ops.attn_bwd(_cont(qkv), ctx_rows, _cont(dctx), lse, n_img, heads, qkv.shape[1] // 3, 1.0)


@attention_bwd.register_fake
def _(dctx, qkv, ctx_rows, lse, n_img, heads):
    return qkv.new_empty(qkv.shape)


def _attn_setup(ctx, inputs, output):
    qkv, n_img, heads = inputs
    ctx_rows, lse = output
    ctx.save_for_backward(qkv, ctx_rows, lse)
    ctx.dims = (n_img, heads)
    ctx.mark_non_differentiable(lse)


def _attn_backward(ctx, dctx, _dlse):
    qkv, ctx_rows, lse = ctx.saved_tensors
    return torch.ops.dig.attention_bwd(dctx, qkv, ctx_rows, lse, *ctx.dims), None, None


attention.register_autograd(_attn_backward, setup_context=_attn_setup)


# ---------------------------------------------------------------------------------------------------------------- MLP half of a block
@custom_op("dig::mlp_block", mutates_args=(), device_types="cuda")
def mlp_block(x: Tensor, ln_w: Tensor, ln_b: Tensor, eps: float, w1: Tensor, b1: Tensor, w2: Tensor,
              b2: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """out = x + b2 + gelu(LN(x) w1^T + b1) w2^T in one launch (dig_mlp_chain_fwd_ln); also returns what the backward reads: the normalised
    rows, their statistics, the pre-activation and the GELU output.  D = 384, hidden width a multiple of 128 up to 2048."""
    r = ops.mlp_chain_fwd_ln(_cont(x), ln_w, ln_b, eps, _cont(w1), b1, _cont(w2), b2, save=True)
    return r["out"], r["ln"], r["ln_mean"], r["ln_rstd"], r["pre"], r["act"]


@mlp_block.register_fake
def _(x, ln_w, ln_b, eps, w1, b1, w2, b2):
    rows, Fh = x.shape[0], w1.shape[0]
    f = lambda *s, dt=None: x.new_empty(s, dtype=dt or x.dtype)
    return f(*x.shape), f(*x.shape), f(rows, dt=F32), f(rows, dt=F32), f(rows, Fh), f(rows, Fh)


@custom_op("dig::mlp_block_bwd", mutates_args=(), device_types="cuda")
def mlp_block_bwd(dout: Tensor, x: Tensor, ln_w: Tensor, ln_b: Tensor, w1: Tensor, w2: Tensor, ln: Tensor, mean: Tensor, rstd: Tensor,
                  pre: Tensor, act: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    dout, w1, w2 = _cont(dout), _cont(w1), _cont(w2)
    dev = x.device
    dw2 = torch.zeros(w2.shape, device=dev, dtype=F32)
    ops.linear_wgrad(dout, act, dw2)
    db2 = torch.zeros(w2.shape[0], device=dev, dtype=F32)
    dpre, parts = ops.linear_dgrad(dout, w2, gelu_pre=pre, colsum=True)      # (dout w2) * gelu'(pre) + the fc1 bias partial sums
    db1 = torch.zeros(w1.shape[0], device=dev, dtype=F32)
    ops.colsum_partials(parts, db1)
    dw1 = torch.zeros(w1.shape, device=dev, dtype=F32)
    ops.linear_wgrad(dpre, ln, dw1)
    dln = ops.linear_dgrad(dpre, w1)
    dg, db = torch.zeros_like(ln_w), torch.zeros_like(ln_b)
    # dx = dout (the skip path) + LN'(dln); the column sums of dout that pass by are the fc2 bias gradient
    dx = ops.layernorm_bwd(dln, _cont(x), ln_w, ln_b, mean, rstd, dout, dg, db, dres_colsum=db2)
    return dx, dg, db, dw1.to(w1.dtype), db1, dw2.to(w2.dtype), db2


@mlp_block_bwd.register_fake
def _(dout, x, ln_w, ln_b, w1, w2, ln, mean, rstd, pre, act):
    return (x.new_empty(x.shape), ln_w.new_empty(ln_w.shape), ln_b.new_empty(ln_b.shape), w1.new_empty(w1.shape),
            ln_w.new_empty((w1.shape[0],)), w2.new_empty(w2.shape), ln_w.new_empty((w2.shape[0],)))


def _mlp_setup(ctx, inputs, output):
    x, ln_w, ln_b, _, w1, _, w2, _ = inputs
    _, ln, mean, rstd, pre, act = output
    ctx.save_for_backward(x, ln_w, ln_b, w1, w2, ln, mean, rstd, pre, act)
    ctx.mark_non_differentiable(ln, mean, rstd, pre, act)


def _mlp_backward(ctx, dout, *_unused):
    dx, dg, db, dw1, db1, dw2, db2 = torch.ops.dig.mlp_block_bwd(dout, *ctx.saved_tensors)
    return dx, dg, db, None, dw1, db1, dw2, db2


mlp_block.register_autograd(_mlp_backward, setup_context=_mlp_setup)
