"""Thin tensor-level wrappers over the C-ABI (include/dig_hip.h).  Every function launches hand-written HIP
kernels on the caller's current stream; nothing here computes on the host or falls back to ATen."""
import ctypes
import os

import torch

from . import _lib as L

cf = ctypes.c_float
cll = ctypes.c_longlong
BF16 = torch.bfloat16
F32 = torch.float32

OUT_BF16, OUT_F32, OUT_F32_PARTIAL = 0, 1, 2
# K-tile depth of the MFMA GEMM: 64 for direct/direct (forward), 32 when a transpose-read operand is involved
# (dgrad / wgrad): measured on MI355X, see profiles/r01_gemm_bk_sweep.txt
GEMM_BK_FWD, GEMM_BK_BWD = 64, 32
_ws = {}


def _stream_id(dev):
    """Raw handle of the current stream of `dev` (a dictionary key; see _lib.stream for why not torch.cuda.current_stream())."""
    if L._raw_stream is not None and dev.index is not None:
        return L._raw_stream(dev.index)
    return torch.cuda.current_stream(dev).cuda_stream


def _workspace(dev, numel):
    """Per-device fp32 scratch for split-R partial slabs (grown on demand, reused across launches on one stream)."""
    key = (dev, _stream_id(dev))        # one scratch per stream: launches on a stream are ordered
    w = _ws.get(key)
    if w is None or w.numel() < numel:
        w = _ws[key] = torch.empty(max(numel, 1 << 22), device=dev, dtype=F32)
    return w


def gemm(A, B, I, J, R, *, ta=False, tb=False, out=None, out_kind=OUT_BF16, bias=None, resid=None, pre=None, alpha=1.0,
         alpha_cols=0, act=0, splits=1, ldc=None, a_rows=0, b_rows=0, bk=0, colsum_partials=None, drop=None):
    """C[I,J] = sum_r opA(i,r) opB(j,r); see csrc/gemm.hip for the operand conventions.  drop: dropout.DropSpec applied to the
    result before the residual add (dig_gemm_bf16_dropout)."""
    if out is None:
        out = torch.empty((I, J), device=A.device, dtype=BF16 if out_kind == OUT_BF16 else F32)
    if not bk:
        bk = GEMM_BK_FWD if not (ta or tb) else GEMM_BK_BWD
    if drop is None:
        L.call("dig_gemm_bf16", L.ptr(A), L.ptr(B), L.ptr(out), I, J, R, A.stride(0), B.stride(0),
               out.stride(0) if ldc is None else ldc, int(ta), int(tb), out_kind, L.ptr(bias), L.ptr(resid),
               resid.stride(0) if resid is not None else 0, L.ptr(pre), pre.stride(0) if pre is not None else 0, cf(alpha),
               alpha_cols, act, splits, a_rows, b_rows, bk, L.ptr(colsum_partials), L.stream())
    else:
        L.call("dig_gemm_bf16_dropout", L.ptr(A), L.ptr(B), L.ptr(out), I, J, R, A.stride(0), B.stride(0),
               out.stride(0) if ldc is None else ldc, int(ta), int(tb), out_kind, L.ptr(bias), L.ptr(resid),
               resid.stride(0) if resid is not None else 0, L.ptr(pre), pre.stride(0) if pre is not None else 0, cf(alpha),
               alpha_cols, act, splits, a_rows, b_rows, bk, L.ptr(colsum_partials), ctypes.byref(drop), L.stream())
    return out


PERSISTENT_FWD = os.environ.get("DIG_PERSISTENT_FWD", "1") != "0"
FWD_192_BELOW = int(os.environ.get("DIG_FWD_192_BELOW", "512"))     # output widths = 128 mod 256 below this run on 256x192 tiles (no padded columns)
# Tile of the fc2 dgrad + GELU' + bias-sum GEMM (the models the fused MLP backward does not take: D = 512).  -1 = by shape: the 256x256 tile for
# tall problems with a 256-multiple width (round 1 kept 128x128 / BK32 everywhere: it shared the CUs better with the weight-gradient stream
# of that time, 25.9 vs 26.1 ms; with the grouped weight gradients the backward is a sum of solo kernel times -- see DGRAD_BK below)
DGRAD_GELU_BK = int(os.environ.get("DIG_DGRAD_GELU_BK", "-1"))


def dgrad_gelu_tile_code(rows, J, drop=None):
    if DGRAD_GELU_BK >= 0:
        return DGRAD_GELU_BK
    return 244 if (rows >= 8192 and drop is None and J % 256 == 0) else 32


def fwd_tile_code(rows, out_dim, K, *, act=0, has_resid=False, drop=None, out_kind=OUT_BF16):
    """DIG_GEMM_TILE_* of a forward Linear layer [rows, K] -> [rows, out_dim].
    Measured on MI355X (profiles/r01_gemm_variants.txt, tools/gpu_bk_probe.py):
      tall layers: 256x256 tiles (16 waves) halve the L2->LDS operand traffic per FLOP;  small GELU layers: BK=32,
      4 workgroups/CU hide the VALU + double-store epilogue;  everything else: the default 128x128 / BK=64."""
    if rows >= 8192 and out_dim >= 384 and rows * out_dim >= (1 << 24) and not (drop is not None and act):
        bk = 244
        if out_dim % 256 == 128 and out_dim < FWD_192_BELOW:
            # 384 outputs = 1.5 tiles of 256: a quarter of the 256x256 tile's columns would be padding.  The 256x192 tile (12 waves) covers
            # them in two exact tiles and keeps the two-fold reuse of the activation rows: proj 39.4 -> 34.5 us, fc2 114.8 -> 99.2 us alone;
            # in the step 25.98 -> 25.57 ms and the forward family 9.76 -> 9.43 ms (three A/B pairs on one box)
            bk = 264
    elif act == 1:
        bk = 32
    elif rows <= 2048 and out_dim <= 512 and K >= 2048 and drop is None:
        bk = 212                                        # few output tiles, long K: 64x128 tiles for more workgroups
    else:
        bk = 0
    if PERSISTENT_FWD and drop is None and out_kind == OUT_BF16 and ((bk == 244 and not has_resid) or bk == 264):
        bk += 300                                       # 544 / 564: the persistent form of the same tile (bit-identical results)
    return bk


NARROW_SPLIT = os.environ.get("DIG_NARROW_SPLIT", "1") != "0"


def narrow_splits(rows, out_cols, K):
    """R-slices for a few-row, narrow-output, long-K layer (the 4096 -> 256 / 384 layers of the BatchNorm-MLP heads and their data
    gradients: 1024 x 256 outputs are 32 workgroups walking 64 K-steps; eight slices of 512 make them 256 workgroups + a 1 MB combine:
    67-73 us in the step -> ~20), or 1."""
    if NARROW_SPLIT and rows <= 2048 and out_cols <= 512 and K >= 2048 and K % 512 == 0 and (rows * out_cols) % 4 == 0:
        return L.lib().dig_gemm_effective_splits(int(K), K // 512)
    return 1


def _split_gemm_bf16(A, B, I, J, R, tb, bk, sp, out):
    ws = _workspace(A.device, sp * I * J)
    gemm(A, B, I, J, R, tb=tb, out=ws, out_kind=OUT_F32_PARTIAL, splits=sp, ldc=J, bk=bk)
    if out is None:
        out = torch.empty((I, J), device=A.device, dtype=BF16)
    L.call("dig_reduce_partials_bf16", L.ptr(ws), sp, cll(I * J), L.ptr(out), L.stream())
    return out


def linear_fwd(x, w, *, bias=None, resid=None, act=0, pre=None, alpha=1.0, alpha_cols=0, out=None, out_kind=OUT_BF16, drop=None):
    """y[rows,out] = x[rows,in] @ w[out,in]^T (+bias)(gelu)(+resid)."""
    if bias is None and resid is None and not act and pre is None and alpha == 1.0 and drop is None and out_kind == OUT_BF16 and (out is None or out.is_contiguous()):
        sp = narrow_splits(x.shape[0], w.shape[0], w.shape[1])
        if sp > 1:
            return _split_gemm_bf16(x, w, x.shape[0], w.shape[0], w.shape[1], False, 212, sp, out)
    bk = fwd_tile_code(x.shape[0], w.shape[0], w.shape[1], act=act, has_resid=resid is not None, drop=drop, out_kind=out_kind)
    return gemm(x, w, x.shape[0], w.shape[0], w.shape[1], bias=bias, resid=resid, act=act, pre=pre, alpha=alpha,
                alpha_cols=alpha_cols, out=out, out_kind=out_kind, bk=bk, drop=drop)


MLP_CHAIN = os.environ.get("DIG_MLP_CHAIN", "1") != "0"      # fused fc1 -> GELU -> fc2 (csrc/mlp_chain.hip) where the widths allow it
# Where the fused kernels are used: bit 0 momentum forward, bit 1 online forward, bit 2 backward.  A chain workgroup owns its CU (8 waves x
# 256 VGPRs, up to 144 KiB of LDS), so nothing of the other HIP stream runs beside it: in the forward that is a net win (24.89 -> 24.03
# ms per step with both branches fused), in the backward the weight-gradient stream loses more than the fused data gradient gains
# (25.73 ms with bit 2 alone, 24.87 with all three; one box, 40 timed steps each) -- default 3.
# Round 4: with the weight gradients of a block as ONE grouped launch (csrc/wgrad.hip) that owns the chip as well, nothing is left for the
# fused backward to starve: 22.92 -> 22.63 ms with bit 2 (two A/B pairs on one box), 22.38 with the 256x192 data-gradient tiles on top -- default 7.
MLP_CHAIN_MASK = int(os.environ.get("DIG_MLP_CHAIN_MASK", "7"))
# the block's norm2 and the next block's norm1 inside the forward chain launch (dig_mlp_chain_fwd_ln)
MLP_CHAIN_LN = os.environ.get("DIG_MLP_CHAIN_LN", "1") != "0"


def mlp_chain_supported(D, F, rows=None):
    """rows: the token rows of the call (the fused kernels address the [rows, F] side tensors with 32-bit byte offsets: beyond 2^32 bytes
    the entry points return "unsupported", and the caller takes the two-GEMM path instead)."""
    return MLP_CHAIN and bool(L.lib().dig_mlp_chain_supported(int(D), int(F))) and (rows is None or rows * F * 2 < (1 << 32))


def mlp_chain_fwd(x, w1, b1, w2, b2, resid, save=False):
    """out = resid + b2 + gelu(x w1^T + b1) w2^T in one launch (Mlp.forward + the block's residual add).  save=True also returns the
    pre-activation and the GELU output ([rows, F] bf16) for the backward: (out, pre, act)."""
    rows, D = x.shape
    Fh = w1.shape[0]
    out = torch.empty((rows, D), device=x.device, dtype=BF16)
    pre = torch.empty((rows, Fh), device=x.device, dtype=BF16) if save else None
    act = torch.empty((rows, Fh), device=x.device, dtype=BF16) if save else None
    L.call("dig_mlp_chain_fwd", L.ptr(x), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), L.ptr(resid), L.ptr(out), L.ptr(pre), L.ptr(act),
           rows, D, Fh, L.stream())
    return (out, pre, act) if save else out


def mlp_chain_fwd_ln(x, ln_g, ln_b, eps, w1, b1, w2, b2, nln_g=None, nln_b=None, save=False, resid=None, drop=None):
    """The second half of a transformer block with its LayerNorms in one launch: out = x + b2 + gelu(LN(x; ln_g, ln_b) w1^T + b1) w2^T and,
    when nln_g is given, the next block's norm1 of `out`.  Returns a dict: out; nln / nln_mean / nln_rstd (when nln_g is given; the
    statistics only with save=True); with save=True also ln, ln_mean, ln_rstd, pre, act -- what the backward reads.
    ln_g None: x holds rows that are normalised already and `resid` the raw rows that are added back.
    drop (dropout.DropSpec or None): out = resid + drop_path(dropout(fc2(.) + b2)) -- Mlp.drop behind fc2 and the MLP branch's drop_path."""
    if resid is None:
        resid = x
    rows, D = x.shape
    Fh = w1.shape[0]
    dev = x.device
    r = {"out": torch.empty((rows, D), device=dev, dtype=BF16)}
    for k in ("ln", "pre", "act", "nln"):
        r[k] = None
    for k in ("ln_mean", "ln_rstd", "nln_mean", "nln_rstd"):
        r[k] = None
    if save:
        if ln_g is not None:
            r["ln"] = torch.empty((rows, D), device=dev, dtype=BF16)
            r["ln_mean"], r["ln_rstd"] = torch.empty(rows, device=dev, dtype=F32), torch.empty(rows, device=dev, dtype=F32)
        r["pre"], r["act"] = torch.empty((rows, Fh), device=dev, dtype=BF16), torch.empty((rows, Fh), device=dev, dtype=BF16)
    if nln_g is not None:
        r["nln"] = torch.empty((rows, D), device=dev, dtype=BF16)
        if save:
            r["nln_mean"], r["nln_rstd"] = torch.empty(rows, device=dev, dtype=F32), torch.empty(rows, device=dev, dtype=F32)
    L.call("dig_mlp_chain_fwd_ln_dropout", L.ptr(x), L.ptr(resid), L.ptr(ln_g), L.ptr(ln_b), cf(eps), L.ptr(r["ln"]), L.ptr(r["ln_mean"]),
           L.ptr(r["ln_rstd"]), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), L.ptr(r["out"]), L.ptr(r["pre"]), L.ptr(r["act"]), L.ptr(nln_g),
           L.ptr(nln_b), L.ptr(r["nln"]), L.ptr(r["nln_mean"]), L.ptr(r["nln_rstd"]), rows, D, Fh, ctypes.byref(drop) if drop is not None else None,
           L.stream())
    return r


def mlp_chain_bwd(dy, w2t, pre, w1t, colsum=True, out=None):
    """dpre = (dy w2) * gelu'(pre), dx = dpre w1 in one launch (w2t = w2^T [F, D], w1t = w1^T [D, F]: transpose_bf16).
    Returns (dx, dpre, parts): parts = [n, F] fp32 partial column sums of dpre (the fc1 bias gradient, for colsum_partials) or None."""
    rows, D = dy.shape
    Fh = w2t.shape[0]
    dx = torch.empty((rows, D), device=dy.device, dtype=BF16) if out is None else out
    dpre = torch.empty((rows, Fh), device=dy.device, dtype=BF16)
    parts = torch.empty((L.lib().dig_mlp_chain_colsum_rows(rows), Fh), device=dy.device, dtype=F32) if colsum else None
    L.call("dig_mlp_chain_bwd", L.ptr(dy), L.ptr(w2t), L.ptr(pre), L.ptr(w1t), L.ptr(dpre), L.ptr(dx), L.ptr(parts), rows, D, Fh, L.stream())
    return dx, dpre, parts


MLP_CHAIN_LNB = os.environ.get("DIG_CHAIN_LNB", "1") != "0"     # norm2's backward inside the fused MLP backward launch
# ... and the attention projection's data gradient behind it (dig_mlp_chain_bwd_ln_proj).  Opt-in: alone the launch beats the two it replaces by
# 12 us (245 against 258), in the step it loses 0.05 ms -- the phase is LDS-bandwidth bound (every weight fragment feeds one MFMA) and the GEMM
# launch it replaces overlaps the side stream's work better (profiles/r05_step_ab.txt)
MLP_CHAIN_PROJ = os.environ.get("DIG_CHAIN_PROJ", "0") == "1"


def mlp_chain_bwd_ln(dy, w2t, pre, w1t, x_mid, ln_g, ln_mean, ln_rstd, colsum=True, out=None, projt=None, dctx=None):
    """mlp_chain_bwd with norm2's backward behind it in the same launch: dx_mid = dy + LN'(dpre w1).  Returns (dx_mid, dpre, parts, ln_parts):
    ln_parts = [n, 3, D] fp32 partial sums (d gamma, d beta, column sums of dy) for layernorm_finalize_parts.
    projt (= proj.weight^T, transpose_bf16): the attention projection's data gradient dctx = dx_mid proj.weight rides in the launch as
    well and is returned as a fifth value."""
    rows, D = dy.shape
    Fh = w2t.shape[0]
    dx = torch.empty((rows, D), device=dy.device, dtype=BF16) if out is None else out
    dpre = torch.empty((rows, Fh), device=dy.device, dtype=BF16)
    n = L.lib().dig_mlp_chain_colsum_rows(rows)
    parts = torch.empty((n, Fh), device=dy.device, dtype=F32) if colsum else None
    ln_parts = torch.empty((L.lib().dig_mlp_chain_ln_parts(rows), 3, D), device=dy.device, dtype=F32)
    if projt is not None:
        dctx = torch.empty((rows, D), device=dy.device, dtype=BF16) if dctx is None else dctx
        L.call("dig_mlp_chain_bwd_ln_proj", L.ptr(dy), L.ptr(w2t), L.ptr(pre), L.ptr(w1t), L.ptr(dpre), L.ptr(x_mid), L.ptr(ln_g), L.ptr(ln_mean),
               L.ptr(ln_rstd), L.ptr(dx), L.ptr(parts), L.ptr(ln_parts), L.ptr(projt), L.ptr(dctx), rows, D, Fh, L.stream())
        return dx, dpre, parts, ln_parts, dctx
    L.call("dig_mlp_chain_bwd_ln", L.ptr(dy), L.ptr(w2t), L.ptr(pre), L.ptr(w1t), L.ptr(dpre), L.ptr(x_mid), L.ptr(ln_g), L.ptr(ln_mean),
           L.ptr(ln_rstd), L.ptr(dx), L.ptr(parts), L.ptr(ln_parts), rows, D, Fh, L.stream())
    return dx, dpre, parts, ln_parts


def layernorm_finalize_parts(ln_parts, dgamma, dbeta, dcolsum=None):
    """dgamma / dbeta / dcolsum += the column sums of ln_parts [n, 3, D] (dig_layernorm_bwd_finalize_parts)."""
    n, _, D = ln_parts.shape
    L.call("dig_layernorm_bwd_finalize_parts", L.ptr(ln_parts), n, D, L.ptr(dgamma), L.ptr(dbeta), L.ptr(dcolsum), L.stream())


def transpose_bf16(src, out=None):
    rows, cols = src.shape
    out = torch.empty((cols, rows), device=src.device, dtype=BF16) if out is None else out
    L.call("dig_transpose_bf16", L.ptr(src), L.ptr(out), rows, cols, L.stream())
    return out


def transpose_bf16_multi(srcs, outs=None):
    """[src^T for src in srcs] for equally shaped bf16 matrices, 32 per launch (dig_transpose_bf16_multi); into `outs` when given."""
    rows, cols = srcs[0].shape
    if outs is None:
        outs = [torch.empty((cols, rows), device=s_.device, dtype=BF16) for s_ in srcs]
    for i in range(0, len(srcs), 32):
        n = min(32, len(srcs) - i)
        sp = (ctypes.c_void_p * n)(*[s_.data_ptr() for s_ in srcs[i:i + n]])
        dp = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs[i:i + n]])
        L.call("dig_transpose_bf16_multi", sp, dp, n, rows, cols, L.stream())
    return outs


def dropout_apply(x, drop, out=None):
    """out = dropout / drop-path (x) for a [rows, cols] bf16 tensor under `drop` (dropout.DropSpec); None -> x itself."""
    if drop is None:
        return x
    out = torch.empty_like(x) if out is None else out
    L.call("dig_dropout_apply", L.ptr(x), L.ptr(out), x.shape[0], x.shape[1], ctypes.byref(drop), L.stream())
    return out


def linear_dgrad(dy, w, out=None, gelu_pre=None, colsum=False, drop=None):
    """dx[rows,in] = dy[rows,out] @ w[out,in]  (* gelu'(gelu_pre) when the input of this layer was a GELU output).
    colsum=True (GELU' form only) also returns the [ceil(rows/64), in] fp32 column sums of dx per 64-row group, i.e. the
    bias gradient of the layer that produced gelu_pre, for colsum_partials()."""
    if gelu_pre is not None:
        parts = torch.empty(((dy.shape[0] + 63) // 64, w.shape[1]), device=dy.device, dtype=F32) if colsum else None
        dx = gemm(dy, w, dy.shape[0], w.shape[1], w.shape[0], tb=True, out=out, act=2, resid=gelu_pre, bk=dgrad_gelu_tile_code(dy.shape[0], w.shape[1], drop), colsum_partials=parts,
                  drop=drop)
        return (dx, parts) if colsum else dx
    # few rows (the BN-MLP heads on pooled features): 128x64 tiles double the workgroup count (tools/gpu_head_gemm_probe.py).
    # (The 256x192 tile is 8-20 % faster for the tall 384-wide dgrads alone -- tools/experiments/gpu_dgrad_tile_probe.py -- but not in
    #  the step: 25.69 vs 25.77 ms over three A/B pairs; the small 128x128 workgroups share the CUs better with the weight-gradient stream.)
    rows, J = dy.shape[0], w.shape[1]
    if drop is None and (out is None or out.is_contiguous()):
        sp = narrow_splits(rows, J, w.shape[0])
        if sp > 1:
            return _split_gemm_bf16(dy, w, rows, J, w.shape[0], True, 221, sp, out)
    return gemm(dy, w, rows, J, w.shape[0], tb=True, out=out, bk=dgrad_tile_code(rows, J, drop))


# The projection's data gradient inside the attention backward launch (dig_attn_bwd_proj): every (image, head) workgroup computes its own d(ctx)
# tile from dx_mid and proj.weight^T; no GEMM launch, d(ctx) never written.  Opt-in ("1"): measured in the step, three A/B pairs on one box,
# 18.66 against 18.65 ms and 19.15 against 19.18 -- the launch grows by what the GEMM took (129 -> 169 us in the step, 127 -> 147 alone + 28.5 for
# the GEMM): the six head-workgroups of an image each pull the image's 196 KB of dy rows and 48 KB of weight through L2 -> CU, 244 KB on top of
# the 288 KB the backward proper moves per workgroup (profiles/r06_attn_bwd_lab.txt).
ATTN_BWD_PROJ = os.environ.get("DIG_ATTN_BWD_PROJ", "0") == "1"
DGRAD_DIRECT = os.environ.get("DIG_DGRAD_DIRECT", "1") != "0"     # the proj / qkv data gradients as direct-form GEMMs on the transposed weight copies
HEAD_DGRAD_DIRECT = os.environ.get("DIG_HEAD_DGRAD_DIRECT", "0") == "1"   # ... and those of the BN-MLP heads: opt-in.  Measured in the step, two A/B pairs
#                                                                           on one box: 19.18 / 19.20 against 19.17 / 19.19 ms -- the 4096 x 4096 layer's
#                                                                           105 -> 64 us is eaten by the 17 M more elements the optimizer launch turns


def dgrad_direct_tile_code(rows, J):
    """DIG_GEMM_TILE_* of a data-gradient GEMM in its DIRECT form dx[rows, J] = dy wt^T on the K-contiguous copy wt = w^T [J, K], or 0: keep
    the transpose-read form.  Measured for the 384-wide gradients of ViT-S on the 256 x 192 persistent tiles (tools/gpu_dgrad_form_probe.py)."""
    return 264 if (DGRAD_DIRECT and rows >= 8192 and J % 192 == 0 and J % 256 != 0) else 0


def dgrad_direct(dy, wt, out=None):
    """dx[rows, J] = dy[rows, K] wt[J, K]^T with wt = w^T, the K-contiguous copy of a Linear weight w [K out, J in] (what dig_adamw_step_tr
    leaves): the data gradient as a direct-form GEMM -- bit-identical to linear_dgrad(dy, w), faster on the tall shapes of the encoder
    (tools/gpu_dgrad_form_probe.py: 384-wide 28.5 / 57.9 us against 31.9 / 69.0; 512-wide 83.8 / 197 / 258 us against 99 / 230 / 301).
    Returns None where the transpose-read form stays (small row counts, DIG_DGRAD_DIRECT=0)."""
    rows, K = dy.shape
    J = wt.shape[0]
    if not DGRAD_DIRECT or rows < 8192:
        return None
    if J % 192 == 0 and J % 256 != 0:
        return gemm(dy, wt, rows, J, K, out=out, bk=264)
    if J % 256 == 0:
        return gemm(dy, wt, rows, J, K, out=out, bk=244) if K >= 1024 else linear_fwd(dy, wt, out=out)
    return None


def dgrad_tile_code(rows, J, drop=None):
    """DIG_GEMM_TILE_* of a data-gradient GEMM dx[rows, J] = dy w."""
    if rows <= 2048:
        return 221
    if DGRAD_BK >= 0:
        return DGRAD_BK
    if rows >= 8192 and drop is None and J % 192 == 0 and J % 256 != 0:
        return 264                                      # 256x192 tiles: exact for the 384-wide data gradients of ViT-S
    if rows >= 8192 and drop is None and J % 256 == 0:
        return 244                                      # 256x256 (D = 512)
    return 0


# Tile code of the tall data-gradient GEMMs: -1 (default) = by width: 256x192 / 256x256 tiles.  Rounds 1-3 kept 128x128 / BK 32 (code 0) here
# because the small workgroups shared the CUs with the weight-gradient stream's 128x128 tiles; with the grouped weight-gradient launch the
# backward is a sum of solo kernel times and the big tile's 8-20 % solo advantage shows in the step (22.63 -> 22.38 ms, two A/B pairs).
DGRAD_BK = int(os.environ.get("DIG_DGRAD_BK", "-1"))
WGRAD_TALL = tuple(int(v) for v in os.environ.get("DIG_WGRAD_TALL", "16,32").split(","))   # (R-splits, tile code) of the 36-tile weight gradients


def colsum_partials(parts, out):
    """out[c] += sum_b parts[b, c]."""
    L.call("dig_colsum_partials", L.ptr(parts), parts.shape[0], parts.shape[1], L.ptr(out), L.stream())


def wgrad_splits(rows, tiles):
    """(R-splits, BK) of a weight-gradient GEMM.  Splits come in multiples of 8 so that every split is pinned to one XCD
    (csrc/gemm.hip: its tiles then share the operand rows through that XCD's L2 instead of re-fetching them from HBM);
    measured on MI355X: 36 tiles -> 16 splits / BK 32 (in the step), 9 tiles -> 40 / BK 64 (tools/experiments/gpu_gemm_shapes.py)."""
    if tiles >= 24:
        # 16 splits x 36 tiles = 576 workgroups: ONE round at three workgroups per CU (24 splits = 864 left a 96-workgroup second round);
        # in the step, same box: 25.22 -> 24.98 ms (tools/experiments/sweep_wgrad_splits.py; 8 / 12 / 32 / 40 splits are slower)
        want, bk = WGRAD_TALL
    else:
        want, bk = min(40, 8 * max(1, round(360 / tiles / 8))), 64
    cap = max(1, rows // 512)
    if cap < want:
        want = max(1, (cap // 8) * 8 or cap)
    return L.lib().dig_gemm_effective_splits(rows, want), bk


def wgrad(dy, x, dw, I, J, rows):
    """dw[I,J] += dy[rows, :I]^T @ x[rows, :J]: split over rows into fp32 slabs, then one deterministic reduce that
    also performs the += into the gradient arena."""
    tiles = ((I + 127) // 128) * ((J + 127) // 128)
    sp, bk = wgrad_splits(rows, tiles)
    ws = _workspace(dy.device, sp * I * J)
    gemm(dy, x, I, J, rows, ta=True, tb=True, out=ws, out_kind=OUT_F32_PARTIAL, splits=sp, ldc=J, bk=bk)
    L.call("dig_reduce_partials", L.ptr(ws), sp, cll(I * J), L.ptr(dw), 1, L.stream())


BATCH_SLABS = os.environ.get("DIG_BATCH_SLABS", "1") != "0"
_ws3 = {}


def _batch_workspace(dev, numel):
    key = (dev, _stream_id(dev) if dev.type == "cuda" else 0)
    w = _ws3.get(key)
    if w is None or w.numel() < numel:
        w = _ws3[key] = torch.empty(numel, device=dev, dtype=F32)
    return w


COLSUM_MAX_SEGS = 112                                               # include/dig_hip.h DIG_COLSUM_MAX_SEGS


class _ReduceSeg(ctypes.Structure):
    """include/dig_hip.h `dig_reduce_seg_t`."""
    _fields_ = [("partials", ctypes.c_void_p), ("out", ctypes.c_void_p), ("n", ctypes.c_longlong), ("splits", ctypes.c_int), ("reserved", ctypes.c_int)]


class _ColsumSeg(ctypes.Structure):
    """include/dig_hip.h `dig_colsum_seg_t`."""
    _fields_ = [("partials", ctypes.c_void_p), ("out", ctypes.c_void_p), ("stride", ctypes.c_longlong), ("n_parts", ctypes.c_int), ("C", ctypes.c_int)]


class GradReduceBatch:
    """The reductions an encoder block's backward leaves behind -- split-R slabs of its four weight gradients, the fc1 / q / v bias
    column sums, the parameter-gradient partials of its two LayerNorms -- collected and issued as TWO launches at flush()
    (dig_reduce_partials_multi, dig_colsum_partials_multi) instead of eleven; same per-element summation order as the single forms."""
    SLAB_FLOATS = 1 << 25                                       # 128 MiB: the four ViT-S weight gradients at 16 splits need 113 MiB

    def __init__(self):
        self.slabs, self.vecs, self.keep, self.off = [], [], [], 0

    def wgrad(self, dy, x, dw, rows=None):
        """dw[I,J] += dy[rows,:I]^T x[rows,:J]: the split-R GEMM now, the slab sum at flush()."""
        I, J = dw.shape
        rows = dy.shape[0] if rows is None else rows
        if not BATCH_SLABS:
            wgrad(dy, x, dw, I, J, rows)
            return
        sp, bk = wgrad_splits(rows, ((I + 127) // 128) * ((J + 127) // 128))
        need = sp * I * J
        if need > self.SLAB_FLOATS or len(self.slabs) == 8:
            wgrad(dy, x, dw, I, J, rows)
            return
        if self.off + need > self.SLAB_FLOATS:
            self.flush()
        ws = _batch_workspace(dy.device, self.SLAB_FLOATS)[self.off:self.off + need]      # its own scratch: slabs stay pending until flush()
        self.off += need
        gemm(dy, x, I, J, rows, ta=True, tb=True, out=ws, out_kind=OUT_F32_PARTIAL, splits=sp, ldc=J, bk=bk)
        self.slabs.append((ws, dw, I * J, sp))

    def colsum_partials(self, parts, out):
        """out[c] += sum_b parts[b, c] at flush()."""
        self._vec(parts, out, parts.shape[1], parts.shape[0], parts.shape[1])

    def layernorm_finalize(self, ws, rows, D, dgamma, dbeta, dcolsum):
        """dig_layernorm_bwd_finalize of a deferred layernorm_bwd(..., defer=True) workspace at flush()."""
        n = L.lib().dig_layernorm_bwd_parts(rows)
        for k, out in enumerate((dgamma, dbeta, dcolsum)):
            if out is not None:
                self._vec(ws[k * D:], out, 3 * D, n, D)

    def layernorm_finalize_parts(self, ln_parts, dgamma, dbeta, dcolsum):
        """dig_layernorm_bwd_finalize_parts of a [parts, 3, D] partial tensor (the fused MLP backward's norm2 sums) at flush()."""
        n, _, D = ln_parts.shape
        flat = ln_parts.view(-1)
        for k, out in enumerate((dgamma, dbeta, dcolsum)):
            if out is not None:
                self._vec(flat[k * D:], out, 3 * D, n, D)

    def _vec(self, parts, out, stride, n_parts, C):
        if len(self.vecs) == COLSUM_MAX_SEGS:
            self._flush_vecs()
        self.vecs.append((parts, out, stride, n_parts, C))

    def tensors(self):
        """Everything flush() reads that another stream may have produced (for record_stream)."""
        return [v[0] for v in self.vecs]

    def _flush_vecs(self):
        if self.vecs:
            segs = (_ColsumSeg * len(self.vecs))(*[_ColsumSeg(p.data_ptr(), o.data_ptr(), st, n, C) for p, o, st, n, C in self.vecs])
            L.call("dig_colsum_partials_multi", segs, len(self.vecs), L.stream())
            self.vecs = []

    def flush(self):
        if self.slabs:
            segs = (_ReduceSeg * len(self.slabs))(*[_ReduceSeg(w.data_ptr(), d.data_ptr(), n, sp, 0) for w, d, n, sp in self.slabs])
            L.call("dig_reduce_partials_multi", segs, len(self.slabs), L.stream())
            self.slabs, self.off = [], 0
        self._flush_vecs()


WGRAD_ASSIGN = os.environ.get("DIG_WGRAD_ASSIGN", "1") != "0"


def linear_wgrad(dy, x, dw, rows=None, assign=False):
    """dw[out,in] += dy[rows,out]^T @ x[rows,in].
    assign=True: the caller states that dw is still zero (its only contribution of this backward pass).  For a SHORT reduction with many
    output tiles (the BatchNorm-MLP heads: <= 2048 pooled rows, up to 4096 x 4096 outputs) the product is then written straight into dw as
    the one "slab" of a one-split launch -- no fp32 slab round trip, no dig_reduce_partials (4096 x 4096: 79 + 70 us -> one launch)."""
    rows = dy.shape[0] if rows is None else rows
    I, J = dw.shape
    if assign and WGRAD_ASSIGN and rows <= 2048 and dw.is_contiguous() and ((I + 127) // 128) * ((J + 127) // 128) >= 64:
        gemm(dy, x, I, J, rows, ta=True, tb=True, out=dw, out_kind=OUT_F32_PARTIAL, splits=1, ldc=J, bk=32)
        return
    wgrad(dy, x, dw, I, J, rows)


# ---- grouped weight gradients (csrc/wgrad.hip): the Linear layers of a transformer block in ONE launch, slabs folded by the next launch
WGRAD_GROUP = os.environ.get("DIG_WGRAD_GROUP", "1") != "0"
WGRAD_GROUP_WA = int(os.environ.get("DIG_WGRAD_WA", "2"))            # 1: 128-row tiles, 4 waves, two workgroups per CU;  2: 256-row tiles, 8 waves, one per CU
WGRAD_GROUP_SLOTS = int(os.environ.get("DIG_WGRAD_SLOTS", "0")) or 512 // WGRAD_GROUP_WA     # workgroups per launch: one round on 256 CUs
_wg_plans, _wg_slabs = {}, {}


class _WgProb(ctypes.Structure):
    """include/dig_hip.h `dig_wgrad_prob_t`."""
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("out", ctypes.c_void_p), ("lda", ctypes.c_int), ("ldb", ctypes.c_int),
                ("ldo", ctypes.c_int), ("I", ctypes.c_int), ("J", ctypes.c_int), ("trans_out", ctypes.c_int)]


def wgrad_group_route(out_dim, in_dim, rows, fn=None):
    """How dW[out_dim, in_dim] = dy^T x joins a group: (trans_out, fn) -- the narrow operand is x (fc1, qkv, proj) or dy (fc2) -- or None.
    fn: the tile width code of the group it has to match (None: any)."""
    lib = L.lib()
    cands = ((0, out_dim, in_dim), (1, in_dim, out_dim))
    if out_dim < in_dim:                                             # the narrow operand is the model width: the smaller dimension first
        cands = cands[::-1]
    for trans, wide, narrow in cands:
        if narrow <= 512 and lib.dig_wgrad_group_supported(int(wide), int(narrow), int(rows)):
            f = lib.dig_wgrad_group_fn(int(narrow))
            if fn is None or f == fn:
                return trans, f
    return None


class WgradGroup:
    """Weight gradients dW += dy^T x collected with add() and computed by launch() in one dig_wgrad_group call; each launch leaves fp32
    partial slabs that the NEXT launch (or flush()) sums -- in split order, deterministic -- into the gradient tensors.  One object per
    backward pass and stream: add / launch / flush must be issued on the same stream (the kernel boundary orders slabs and fold)."""

    def __init__(self, dev):
        self.dev, self.cur, self.pending, self.set, self.fn, self.rows = dev, [], None, 0, None, None

    def add(self, dy, x, dw, rows=None):
        """Queue dw[out, in] += dy[rows, out]^T x[rows, in] for the next launch(); False (nothing queued) when the shape cannot join."""
        rows = dy.shape[0] if rows is None else rows
        r = wgrad_group_route(dw.shape[0], dw.shape[1], rows, self.fn) if WGRAD_GROUP else None
        if r is None or len(self.cur) == 6 or (self.rows not in (None, rows)) or dw.stride(1) != 1:
            return False
        if rows * max(dy.stride(0), x.stride(0)) * 2 >= 1 << 32:      # 32-bit buffer offsets in the kernel: the caller's tiled path takes it
            return False
        trans, self.fn = r
        self.rows = rows
        A, B = (x, dy) if trans else (dy, x)
        self.cur.append((A, B, dw, trans))
        return True

    def _plan(self, tiles):
        key = (tuple(tiles), self.rows, WGRAD_GROUP_SLOTS, WGRAD_GROUP_WA, str(self.dev))
        pl = _wg_plans.get(key)
        if pl is None:
            tp = (ctypes.c_int * len(tiles))(*tiles)
            cap = 8 * max(WGRAD_GROUP_SLOTS, sum(tiles))
            buf = (ctypes.c_uint * cap)()
            sp = ctypes.c_int(0)
            n = L.lib().dig_wgrad_group_plan(tp, len(tiles), int(self.rows), WGRAD_GROUP_SLOTS, ctypes.byref(sp), buf, cap)
            if n <= 0:
                L.check(n or -1, "dig_wgrad_group_plan")
            import numpy as np
            m = torch.from_numpy(np.frombuffer(buf, dtype=np.uint32, count=n).view(np.int32).copy())
            pl = _wg_plans[key] = (sp.value, n, m.to(self.dev))
        return pl

    def _slabs(self, nbytes):
        key = (str(self.dev), _stream_id(self.dev) if self.dev.type == "cuda" else 0, self.set)
        w = _wg_slabs.get(key)
        if w is None or w.numel() * 4 < nbytes:
            w = _wg_slabs[key] = torch.empty((max(nbytes, 1 << 26) + 3) // 4, device=self.dev, dtype=F32)
        return w

    def _structs(self, items):
        arr = (_WgProb * max(len(items), 1))()
        for k, (A, B, dw, trans) in enumerate(items):
            J = dw.shape[0] if trans else dw.shape[1]
            I = dw.shape[1] if trans else dw.shape[0]
            arr[k] = _WgProb(A.data_ptr() if A is not None else None, B.data_ptr() if B is not None else None, dw.data_ptr(),
                             A.stride(0) if A is not None else 0, B.stride(0) if B is not None else 0, dw.stride(0), I, J, trans)
        return arr

    def launch(self):
        """Partial products of everything add()ed since the last launch + the fold of the previous launch's slabs."""
        if not self.cur and self.pending is None:
            return
        fn = self.fn
        tj = 128 * fn
        wa = WGRAD_GROUP_WA
        tiles = [L.lib().dig_wgrad_group_tiles(int(dw.shape[1] if t else dw.shape[0]), int(dw.shape[0] if t else dw.shape[1]), fn, wa)
                 for _, _, dw, t in self.cur]
        if self.cur:
            splits, n_wg, wmap = self._plan(tiles)
            slabs = self._slabs(sum(tiles) * splits * 128 * wa * tj * 4)
        else:
            splits, n_wg, wmap, slabs = 1, WGRAD_GROUP_SLOTS, None, None
        pend = self.pending
        L.call("dig_wgrad_group", self._structs(self.cur), len(self.cur), self._structs(pend[0]) if pend else None, len(pend[0]) if pend else 0,
               int(self.rows), splits, L.ptr(wmap), n_wg, L.ptr(slabs), L.ptr(pend[1]) if pend else None, pend[2] if pend else 1, fn, wa, L.stream())
        # the fold list keeps the gradient tensors only (operands are dead once this launch has run)
        self.pending = ([(None, None, dw, t) for _, _, dw, t in self.cur], slabs, splits) if self.cur else None
        self.cur = []
        self.set ^= 1

    def flush(self):
        """Fold what the last launch left (a fold-only launch); afterwards every gradient add()ed so far is final on this stream."""
        assert not self.cur, "launch() the queued problems first"
        self.launch()


# ---- one FFI crossing per encoder block (include/dig_block_types.h, csrc/encoder_block.hip) -----------------------------------------------
BLOCK_CALLS = os.environ.get("DIG_BLOCK_CALLS", "1") != "0"
_VP, _FP, _I, _F = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float


class BlockFwd(ctypes.Structure):
    """include/dig_block_types.h `dig_block_fwd_t`."""
    _fields_ = ([(k, _I) for k in ("n_img", "heads", "D", "F", "rows", "save", "tile_qkv", "tile_proj", "fuse_attn", "reserved0")] + [("eps", _F), ("scale", _F)] +
                [(k, _VP) for k in ("qkv_w", "qkv_b", "proj_w", "proj_b", "n2_g", "n2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "next_n1_g", "next_n1_b",
                                    "x", "ln1", "qkv", "ctx", "lse", "x_mid", "ln2", "mu2", "rs2", "pre", "act", "out", "nln", "nmu", "nrs")])


class BlockBwd(ctypes.Structure):
    """include/dig_block_types.h `dig_block_bwd_t`."""
    _fields_ = ([(k, _I) for k in ("n_img", "heads", "D", "F", "rows", "tile_dgrad", "tile_direct")] + [("scale", _F)] +
                [(k, _VP) for k in ("qkv_w", "proj_w", "w2t", "w1t", "projt", "proj_wt", "qkv_wt", "n1_g", "n1_b", "n2_g", "n2_b",
                                    "g_n1_g", "g_n1_b", "g_qkv_w", "g_q_b", "g_v_b", "g_proj_w", "g_proj_b", "g_n2_g", "g_n2_b", "g_fc1_w", "g_fc1_b",
                                    "g_fc2_w", "g_fc2_b",
                                    "x", "ln1", "mu1", "rs1", "qkv", "ctx", "lse", "x_mid", "ln2", "mu2", "rs2", "pre", "act", "dy",
                                    "dln2", "dpre", "dctx", "dqkv", "bparts", "ws1", "ws2", "qs", "vs")] +
                [(k, _I) for k in ("wg_fn", "wg_wa", "wg_splits", "wg_n_wg", "wg_fold_n", "wg_fold_splits")] + [("wg_trans", _I * 4)] +
                [("wg_defer", _I), ("fuse_ln2", _I), ("attn_proj", _I), ("defer_red", _I)] +
                [(k, _VP) for k in ("wg_map", "wg_slabs", "wg_fold_slabs", "wg_probs", "wg_fold_probs", "side")])


def _round_up(n, m):
    return (n + m - 1) // m * m


_block_layouts = {}


def block_fwd_layout(rows, D, Fh, n_img, heads, save):
    """Byte offsets of one block's forward outputs inside its two buffers (bf16 tensors; fp32 statistics): ({name: offset}, bytes16, bytes32).
    Every piece starts on a 256-byte boundary."""
    key = ("f", rows, D, Fh, n_img, heads, bool(save))
    lay = _block_layouts.get(key)
    if lay is None:
        off, n16, n32 = {}, 0, 0
        for name, cols in (("qkv", 3 * D), ("ctx", D), ("x_mid", D), ("out", D), ("nln", D)) + ((("ln2", D), ("pre", Fh), ("act", Fh)) if save else ()):
            off[name] = n16
            n16 += _round_up(rows * cols * 2, 256)
        for name, n in (("lse", n_img * heads * 256),) + ((("mu2", rows), ("rs2", rows), ("nmu", rows), ("nrs", rows)) if save else ()):
            off[name] = n32
            n32 += _round_up(n * 4, 256)
        lay = _block_layouts[key] = (off, n16, n32)
    return lay


def block_bwd_layout(rows, D, Fh, n_img):
    """The same for one block's backward temporaries: dln2, dpre, dctx, dqkv (bf16) and bparts, ws1, ws2, qs, vs (fp32)."""
    key = ("b", rows, D, Fh, n_img)
    lay = _block_layouts.get(key)
    if lay is None:
        off, n16, n32 = {}, 0, 0
        for name, cols in (("dln2", D), ("dpre", Fh), ("dctx", D), ("dqkv", 3 * D)):
            off[name] = n16
            n16 += _round_up(rows * cols * 2, 256)
        lib = L.lib()
        parts = lib.dig_layernorm_bwd_parts(rows) * 3 * D
        parts2 = max(parts, lib.dig_mlp_chain_ln_parts(rows) * 3 * D)                 # (ws2 with fuse_ln2: one partial row per 128 tokens)
        for name, n in (("bparts", lib.dig_mlp_chain_colsum_rows(rows) * Fh), ("ws1", parts), ("ws2", parts2), ("qs", n_img * D), ("vs", n_img * D)):
            off[name] = n32
            n32 += _round_up(n * 4, 256)
        lay = _block_layouts[key] = (off, n16, n32)
    return lay


def wgrad_block_plan(dev, rows, D, Fh):
    """The grouped weight-gradient launch of one encoder block (fc2, fc1, proj, qkv) for dig_encoder_block_bwd: None when a shape cannot
    join, else a dict with fn, wa, splits, n_wg, wmap (device tensor), slab_bytes, trans (one flag per problem) and `group` (the WgradGroup
    whose slab sets the launches alternate between)."""
    grp = WgradGroup(dev)
    grp.rows = rows
    trans, tiles = [], []
    for out_dim, in_dim in ((D, Fh), (Fh, D), (D, D), (3 * D, D)):
        r = wgrad_group_route(out_dim, in_dim, rows, grp.fn) if WGRAD_GROUP else None
        if r is None:
            return None
        t, grp.fn = r
        trans.append(t)
        tiles.append(L.lib().dig_wgrad_group_tiles(int(in_dim if t else out_dim), int(out_dim if t else in_dim), grp.fn, WGRAD_GROUP_WA))
    splits, n_wg, wmap = grp._plan(tiles)
    return {"fn": grp.fn, "wa": WGRAD_GROUP_WA, "splits": splits, "n_wg": n_wg, "wmap": wmap, "trans": trans, "group": grp,
            "slab_bytes": sum(tiles) * splits * 128 * WGRAD_GROUP_WA * 128 * grp.fn * 4}


_ws2 = {}


def _workspace2(dev, numel):
    """Second fp32 scratch (column-sum / LayerNorm-backward partials)."""
    key = (dev, _stream_id(dev))
    w = _ws2.get(key)
    if w is None or w.numel() < numel:
        w = _ws2[key] = torch.empty(max(numel, 1 << 20), device=dev, dtype=F32)
    return w


def colsum(x, out, rows=None, cols=None):
    """out[c] += sum_r x[r, c]  (two-stage, deterministic)."""
    rows = x.shape[0] if rows is None else rows
    cols = x.shape[1] if cols is None else cols
    ws = _workspace2(x.device, 1024 * cols)
    L.call("dig_colsum", L.ptr(x), L.ptr(out), L.ptr(ws), rows, cols, x.stride(0), L.stream())


def layernorm_fwd(x, gamma, beta, eps, gelu=False):
    rows, D = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=x.device, dtype=F32)
    rstd = torch.empty(rows, device=x.device, dtype=F32)
    L.call("dig_layernorm_fwd", L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(mean), L.ptr(rstd), rows, D, cf(eps),
           int(gelu), L.stream())
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, beta, mean, rstd, dres, dgamma, dbeta, gelu=False, out=None, dres_colsum=None, defer=False):
    """dx = [dres +] LN'(dy); dgamma/dbeta accumulated; dres_colsum (optional) += column sums of dres, i.e. the bias
    gradient of the layer whose output fed the residual sum -- read for free while dres streams through.
    defer=True returns (dx, finish, workspace): `finish()` launches the parameter-gradient reduction (callable on another
    stream; keep `workspace` alive / record_stream it there)."""
    rows, D = x.shape
    dx = torch.empty_like(x) if out is None else out
    if not defer:
        ws = _workspace2(x.device, 1024 * 3 * D)
        L.call("dig_layernorm_bwd", L.ptr(dy), L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(rstd), L.ptr(dres), L.ptr(dx),
               L.ptr(dgamma), L.ptr(dbeta), L.ptr(dres_colsum), L.ptr(ws), rows, D, int(gelu), L.stream())
        return dx
    ws = torch.empty(L.lib().dig_layernorm_bwd_parts(rows) * 3 * D, device=x.device, dtype=F32)
    L.call("dig_layernorm_bwd_partials", L.ptr(dy), L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(rstd), L.ptr(dres),
           L.ptr(dx), L.ptr(ws), rows, D, int(gelu), L.stream())

    def finish():
        L.call("dig_layernorm_bwd_finalize", L.ptr(ws), rows, D, L.ptr(dgamma), L.ptr(dbeta), L.ptr(dres_colsum), L.stream())
    return dx, finish, ws


def attn_fwd(qkv, n_img, heads, D, drop=None, q_rows=256):
    """q_rows < 256: only the first q_rows query rows per image are computed (the other ctx / lse rows stay uninitialised)."""
    ctx = torch.empty((qkv.shape[0], D), device=qkv.device, dtype=BF16)
    lse = torch.empty((n_img * heads, 256), device=qkv.device, dtype=F32)
    L.call("dig_attn_fwd_dropout", L.ptr(qkv), L.ptr(ctx), L.ptr(lse), n_img, heads, D, ctypes.byref(drop) if drop is not None else None,
           q_rows, L.stream())
    return ctx, lse


ATTN_BLOCK = os.environ.get("DIG_ATTN_BLOCK", "1") != "0"    # the fused attention sub-block (csrc/attn_block.hip) where the widths allow it


# one workgroup per image: the single-view encoder of a Gen-only model at B = 128 is 128 images on 256 CUs -- the launch would leave half
# the chip idle for its whole length, and the three launches it replaces (which tile over rows / (image, head) pairs) win.  Applied to the
# single-view encoder only (n_img given); the two-view step keeps the fused launch at every batch size.
ATTN_BLOCK_MIN_IMG = int(os.environ.get("DIG_ATTN_BLOCK_MIN_IMG", "160"))


def attn_block_supported(heads, D, n_img=None):
    return (ATTN_BLOCK and (n_img is None or n_img >= ATTN_BLOCK_MIN_IMG or n_img < 32)
            and bool(L.lib().dig_attn_block_supported(int(heads), int(D))))


def attn_block_fwd(ln1, x, qkv_w, qkv_b, proj_w, proj_b, n_img, heads, D, scale, save=False):
    """x_mid = x + proj(attention(ln1)) in one launch.  Returns (x_mid, ctx, qkv, lse); qkv / lse are None unless save."""
    rows = ln1.shape[0]
    x_mid = torch.empty((rows, D), device=ln1.device, dtype=BF16)
    ctx = torch.empty((rows, D), device=ln1.device, dtype=BF16)
    qkv = torch.empty((rows, 3 * D), device=ln1.device, dtype=BF16) if save else None
    lse = torch.empty((n_img * heads, 256), device=ln1.device, dtype=F32) if save else None
    L.call("dig_attn_block_fwd", L.ptr(ln1), L.ptr(x), L.ptr(qkv_w), L.ptr(qkv_b), L.ptr(proj_w), L.ptr(proj_b), L.ptr(qkv), L.ptr(ctx),
           L.ptr(lse), L.ptr(x_mid), n_img, heads, D, cf(scale), L.stream())
    return x_mid, ctx, qkv, lse


def attn_bwd_mode(single_pass=None):
    """Select (True / False) or query (None) the kernel behind attn_bwd for full self-attention: single pass or two phases.  Returns the previous setting."""
    return bool(L.lib().dig_attn_bwd_mode(-1 if single_pass is None else int(bool(single_pass))))


if os.environ.get("DIG_ATTN_BWD_SP") in ("0", "1"):
    attn_bwd_mode(os.environ["DIG_ATTN_BWD_SP"] == "1")


def attn_bwd_store(mode=None):
    """Select (0 / 1 / 3) or query (None) how the two-phase attention backward writes dqkv: 16-byte row stores, the same non-temporal, or full
    128-byte lines through LDS, non-temporal (the default).  Bit-identical results.  Returns the previous setting."""
    if mode is not None and mode not in (0, 1, 3):
        raise ValueError(f"attn_bwd_store: mode {mode!r} (one of 0, 1, 3)")
    return int(L.lib().dig_attn_bwd_store(-1 if mode is None else int(mode)))


if os.environ.get("DIG_ATTN_BWD_STORE") is not None:
    attn_bwd_store(int(os.environ["DIG_ATTN_BWD_STORE"]))


def attn_bwd(qkv, ctx, dctx, lse, n_img, heads, D, scale, bias_sums=False, drop=None, q_rows=256):
    """dqkv (dq pre-multiplied by `scale`).  bias_sums=True also returns the per-image column sums of the dq and dv parts
    ([n_img, D] fp32 each): the q_bias / v_bias gradient partials for colsum_partials()."""
    dqkv = torch.empty_like(qkv)
    qs = torch.empty((n_img, D), device=qkv.device, dtype=F32) if bias_sums else None
    vs = torch.empty((n_img, D), device=qkv.device, dtype=F32) if bias_sums else None
    L.call("dig_attn_bwd_dropout", L.ptr(qkv), L.ptr(ctx), L.ptr(dctx), L.ptr(lse), L.ptr(dqkv), n_img, heads, D, cf(scale), L.ptr(qs), L.ptr(vs),
           ctypes.byref(drop) if drop is not None else None, q_rows, L.stream())
    return (dqkv, qs, vs) if bias_sums else dqkv


def attn_bwd_proj(qkv, ctx, dy, projt, lse, n_img, heads, D, scale, bias_sums=False):
    """attn_bwd with the projection's data gradient inside the launch: dy = gradient of the projection's output rows [R, D], projt = proj.weight^T
    [in][out] (bf16); d(ctx) = dy @ proj.weight is computed per (image, head) workgroup and never written."""
    dqkv = torch.empty_like(qkv)
    qs = torch.empty((n_img, D), device=qkv.device, dtype=F32) if bias_sums else None
    vs = torch.empty((n_img, D), device=qkv.device, dtype=F32) if bias_sums else None
    L.call("dig_attn_bwd_proj", L.ptr(qkv), L.ptr(ctx), L.ptr(dy), L.ptr(projt), L.ptr(lse), L.ptr(dqkv), n_img, heads, D, cf(scale), L.ptr(qs), L.ptr(vs),
           L.stream())
    return (dqkv, qs, vs) if bias_sums else dqkv


def attn_bwd_proj_supported(D):
    return D % 128 == 0 and D <= 512


def patch_embed_fwd(img, W, bias, mask_u8, mask_token, pos, D, gh, gw):
    n_img = img.shape[0]
    out = torch.empty((n_img * gh * gw, D), device=img.device, dtype=BF16)
    L.call("dig_patch_embed_fwd", L.ptr(img), L.ptr(W), L.ptr(bias), L.ptr(mask_u8), L.ptr(mask_token), L.ptr(pos), L.ptr(out),
           n_img, gh, gw, D, L.stream())
    return out


def patch_embed_bwd(dy, img, mask_u8, dW, dbias, dmask_token, D, gh, gw):
    L.call("dig_patch_embed_bwd", L.ptr(dy), L.ptr(img), L.ptr(mask_u8), L.ptr(dW), L.ptr(dbias), L.ptr(dmask_token),
           img.shape[0], gh, gw, D, L.stream())


def patch_embed_bwd_mfma(dy, img, mask_u8, dW, dbias, dmask_token, D, gh, gw):
    """Patch-embed gradients with the weight gradient on the matrix cores: P = bf16 patch matrix (masked rows zero),
    dW[D,48] += dy^T P (wgrad GEMM), bias / mask_token gradients by one masked column-sum pass over dy."""
    n_tok = img.shape[0] * gh * gw
    P = torch.empty((n_tok, 64), device=img.device, dtype=BF16)
    L.call("dig_patchify_bf16", L.ptr(img), L.ptr(mask_u8), L.ptr(P), img.shape[0], gh, gw, L.stream())
    wgrad(dy, P, dW, D, 48, n_tok)
    ws = _workspace2(dy.device, 2 * 1024 * D)
    L.call("dig_colsum_masked", L.ptr(dy), L.ptr(mask_u8), L.ptr(dbias), L.ptr(dmask_token), L.ptr(ws), n_tok, D, L.stream())


def window_pool_fwd(x, out, n_img, gh, gw, nwin, D):
    L.call("dig_window_pool_fwd", L.ptr(x), L.ptr(out), int(out.dtype == F32), n_img, gh, gw, nwin, D, L.stream())


def window_pool_bwd(dpool, dx, n_img, gh, gw, nwin, D, accumulate):
    L.call("dig_window_pool_bwd", L.ptr(dpool), L.ptr(dx), n_img, gh, gw, nwin, D, int(accumulate), L.stream())


# ---- ConvPatchNet data movement (csrc/conv_patch.hip): NHWC bf16 maps [n_img * H * W, C]
def im2col3x3(x, n_img, H, W, C):
    """[n_img H W, ldc] matrix of a 3x3 / pad 1 convolution over x, columns c * 9 + ky * 3 + kx (conv.weight.view(C_out, -1)'s order), the row
    pitch rounded up to the GEMM's 64-element reduction granule (pad columns zero)."""
    ldc = _round_up(9 * C, 64)
    col = torch.empty((n_img * H * W, ldc), device=x.device, dtype=BF16)
    L.call("dig_im2col3x3", L.ptr(x), L.ptr(col), n_img, H, W, C, ldc, L.stream())
    return col


def conv3x3_weight_flip(w, c_out, c_in):
    """[c_in, c_out * 9] bf16: the taps of w [c_out, c_in * 9] flipped and transposed (the data gradient's convolution weights)."""
    wt = torch.empty((c_in, c_out * 9), device=w.device, dtype=BF16)
    L.call("dig_conv3x3_weight_flip", L.ptr(w), L.ptr(wt), c_out, c_in, L.stream())
    return wt


def maxpool2x2_fwd(x, n_img, H, W, C):
    y = torch.empty((n_img * (H // 2) * (W // 2), C), device=x.device, dtype=BF16)
    idx = torch.empty((n_img * (H // 2) * (W // 2), C), device=x.device, dtype=torch.uint8)
    L.call("dig_maxpool2x2_fwd", L.ptr(x), L.ptr(y), L.ptr(idx), n_img, H, W, C, L.stream())
    return y, idx


def maxpool2x2_bwd(dy, idx, n_img, H, W, C):
    dx = torch.empty((n_img * H * W, C), device=dy.device, dtype=BF16)
    L.call("dig_maxpool2x2_bwd", L.ptr(dy), L.ptr(idx), L.ptr(dx), n_img, H, W, C, L.stream())
    return dx


_MASK_KINDS = {torch.bool: 0, torch.uint8: 0, torch.float32: 1, torch.float64: 2, torch.int32: 3, torch.int64: 4}


def mask_views_u8(mask_bvn, keep_views):
    """[B, V, N] mask of any of those element types -> uint8 [V * B, N], view major, views >= keep_views zeroed; None for other inputs."""
    kind = _MASK_KINDS.get(mask_bvn.dtype)
    if kind is None or mask_bvn.dim() != 3 or not mask_bvn.is_contiguous():
        return None
    B, V, N = mask_bvn.shape
    out = torch.empty((V * B, N), device=mask_bvn.device, dtype=torch.uint8)
    L.call("dig_mask_views_u8", L.ptr(mask_bvn), kind, B, V, N, int(keep_views), L.ptr(out), L.stream())
    return out


def mask_to_index(mask_u8, max_per_sample):
    B, N = mask_u8.shape
    idx = torch.zeros((B, max_per_sample), device=mask_u8.device, dtype=torch.int32)   # ragged masks are reported one step late
    cnt = torch.empty((B,), device=mask_u8.device, dtype=torch.int32)
    L.call("dig_mask_to_index", L.ptr(mask_u8), L.ptr(idx), L.ptr(cnt), B, N, max_per_sample, L.stream())
    return idx, cnt


def gather_rows(src, idx, M, M_pad):
    D = src.shape[1]
    dst = torch.empty((M_pad, D), device=src.device, dtype=BF16)
    L.call("dig_gather_rows", L.ptr(src), L.ptr(idx), L.ptr(dst), M, M_pad, D, L.stream())
    return dst


def scatter_rows_add(src, idx, dst, M):
    L.call("dig_scatter_rows_add", L.ptr(src), L.ptr(idx), L.ptr(dst), M, src.shape[1], L.stream())


def mim_target(img, idx, M, gh, gw, normalize=False):
    tgt = torch.empty((M, 48), device=img.device, dtype=F32)
    L.call("dig_mim_target", L.ptr(img), L.ptr(idx), L.ptr(tgt), M, gh, gw, int(normalize), L.stream())
    return tgt


_loss_ws = {}


def _loss_workspace(dev, tag, numel):
    """Zero-initialised fp32 scratch of the deterministic loss reductions (the kernels leave its ticket word at zero)."""
    key = (str(dev), tag, _stream_id(dev) if dev.type == "cuda" else 0)
    w = _loss_ws.get(key)
    if w is None or w.numel() < numel:
        w = _loss_ws[key] = torch.zeros(max(numel, 4096), device=dev, dtype=F32)
    return w


def mse_fwd_bwd(pred, ld_pred, target, M, C, gscale, loss, dpred, ld_dpred):
    """loss += mean((pred - target)^2) with the block sums added in block order (no floating-point atomics: the logged value is bit-stable)."""
    L.call("dig_mse_fwd_bwd_ws", L.ptr(pred), ld_pred, L.ptr(target), M, C, cf(gscale), L.ptr(loss), L.ptr(dpred), ld_dpred,
           L.ptr(_loss_workspace(pred.device, "mse", 257)), L.stream())


def add_bf16(a, b, out):
    L.call("dig_add_bf16", L.ptr(a), L.ptr(b), L.ptr(out), cll(a.numel()), L.stream())


def gelu_bwd(dact, pre, out):
    L.call("dig_gelu_bwd", L.ptr(dact), L.ptr(pre), L.ptr(out), cll(dact.numel()), L.stream())


def _bn_ws_floats(rows, C):
    f = L.lib().dig_bn_stats_workspace_bytes
    f.restype = ctypes.c_longlong
    return int(f(int(rows), int(C))) // 4


def bn_stats(x, sums):
    """sums[2,C] = (sum_r x, sum_r x^2) -- overwritten, deterministic."""
    ws = _workspace2(x.device, _bn_ws_floats(x.shape[0], x.shape[1]))
    L.call("dig_bn_stats", L.ptr(x), L.ptr(sums), L.ptr(ws), x.shape[0], x.shape[1], L.stream())


def bn_fwd_apply(x, sums, n_total, eps, gamma, beta, relu, running=None):
    """running = (running_mean, running_var, momentum): the running statistics are updated by the same launch (dig_bn_update_running's math)."""
    rows, C = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(C, device=x.device, dtype=F32)
    rstd = torch.empty(C, device=x.device, dtype=F32)
    rm, rv, mom = running if running is not None else (None, None, 0.0)
    L.call("dig_bn_fwd_apply_running", L.ptr(x), L.ptr(sums), cf(n_total), cf(eps), L.ptr(gamma), L.ptr(beta), int(relu), L.ptr(y),
           L.ptr(mean), L.ptr(rstd), cf(mom), L.ptr(rm), L.ptr(rv), rows, C, L.stream())
    return y, mean, rstd


BN_FUSED = os.environ.get("DIG_BN_FUSED", "1") != "0"      # few-row BatchNorm layers of a single rank in one launch each (csrc/norm.hip)


def bn_fused_supported(rows, C):
    return BN_FUSED and bool(L.lib().dig_bn_fused_supported(int(rows), int(C)))


def bn_fwd_fused(x, eps, gamma, beta, relu, running=None):
    """BatchNorm (train mode, statistics over these rows) of a few-row layer in ONE launch: (y, mean, rstd); running = (running_mean,
    running_var, momentum) is updated by the same launch."""
    rows, C = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(C, device=x.device, dtype=F32)
    rstd = torch.empty(C, device=x.device, dtype=F32)
    rm, rv, mom = running if running is not None else (None, None, 0.0)
    L.call("dig_bn_fwd_fused", L.ptr(x), cf(eps), L.ptr(gamma), L.ptr(beta), int(relu), L.ptr(y), L.ptr(mean), L.ptr(rstd), cf(mom), L.ptr(rm),
           L.ptr(rv), rows, C, L.stream())
    return y, mean, rstd


def bn_bwd_fused(dy, x, mean, rstd, gamma, beta, relu, dbeta=None, dgamma=None):
    """Its gradient in ONE launch: dx; dbeta / dgamma (both or neither) += the layer's affine gradients."""
    dx = torch.empty_like(x)
    L.call("dig_bn_bwd_fused", L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(beta), int(relu), L.ptr(dbeta), L.ptr(dgamma),
           L.ptr(dx), x.shape[0], x.shape[1], L.stream())
    return dx


def bn_update_running(sums, n_total, momentum, rm, rv):
    L.call("dig_bn_update_running", L.ptr(sums), cf(n_total), cf(momentum), L.ptr(rm), L.ptr(rv), rm.numel(), L.stream())


def bn_bwd_stats(dy, x, mean, rstd, gamma, beta, relu, sums, dbeta=None, dgamma=None):
    """sums[0] = sum g, sums[1] = sum g * x_hat over this rank's rows; dbeta / dgamma (both or neither): the layer's affine gradients,
    += those LOCAL sums in the same launch."""
    ws = _workspace2(x.device, _bn_ws_floats(x.shape[0], x.shape[1]))
    L.call("dig_bn_bwd_stats_acc", L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(beta), int(relu), L.ptr(sums),
           L.ptr(dbeta), L.ptr(dgamma), L.ptr(ws), x.shape[0], x.shape[1], L.stream())


def bn_bwd_apply(dy, x, mean, rstd, gamma, beta, relu, sums, n_total):
    dx = torch.empty_like(x)
    L.call("dig_bn_bwd_apply", L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(beta), int(relu), L.ptr(sums),
           cf(n_total), L.ptr(dx), x.shape[0], x.shape[1], L.stream())
    return dx


def l2norm_fwd(x, eps=1e-12):
    y = torch.empty_like(x)
    inv = torch.empty(x.shape[0], device=x.device, dtype=F32)
    L.call("dig_l2norm_fwd", L.ptr(x), L.ptr(y), L.ptr(inv), x.shape[0], x.shape[1], cf(eps), L.stream())
    return y, inv


def l2norm_bwd(dy, y, inv):
    dx = torch.empty_like(y)
    L.call("dig_l2norm_bwd", L.ptr(dy), L.ptr(y), L.ptr(inv), L.ptr(dx), y.shape[0], y.shape[1], L.stream())
    return dx


SGEMM_SPLIT_SMALL = os.environ.get("DIG_SGEMM_SPLIT_SMALL", "1") != "0"
SGEMM_SPLIT_MIN_R = 256          # (the 512 x 512 x 256 logits: 16.5 -> 13.0 us; the 512 x 256 x 512 logit gradient: 36.3 -> 14.8 us)


def sgemm(A, B, C, I, J, R, trans_b, alpha):
    """fp32 C = alpha * A[I,R] * (B[R,J] if trans_b else B[J,R]^T).  A long reduction (the logit gradient against keys gathered
    from many ranks) is split into fp32 slabs and summed in a fixed order (deterministic)."""
    sp = 1
    if R >= 2048 and C.stride(0) == J:
        sp = R // 512
        while sp > 1 and R % (64 * sp):
            sp -= 1
    elif SGEMM_SPLIT_SMALL and R >= SGEMM_SPLIT_MIN_R and I * J <= (1 << 18) and C.stride(0) == J and R % 256 == 0:
        # few output tiles (the logit gradient against this rank's own keys: 512 x 256 outputs = 128 workgroups walking R = 512): four
        # R-slices of 128 make it one round of 512 workgroups; the slab sum is a 2 MB pass (50.8 -> ~25 us per launch, twice per step)
        sp = R // 128
    if sp > 1:
        ws = _workspace(A.device, sp * I * J)
        L.call("dig_sgemm", L.ptr(A), L.ptr(B), L.ptr(ws), I, J, R, A.stride(0), B.stride(0), J, int(trans_b), cf(alpha), sp, L.stream())
        L.call("dig_reduce_partials", L.ptr(ws), sp, cll(I * J), L.ptr(C), 0, L.stream())
    else:
        L.call("dig_sgemm", L.ptr(A), L.ptr(B), L.ptr(C), I, J, R, A.stride(0), B.stride(0), C.stride(0), int(trans_b), cf(alpha), 1, L.stream())


def ce_rows(logits, label_offset, gscale, out3):
    """The rows' loss / hit counts are added in row order (no floating-point atomics)."""
    n = logits.shape[0]
    L.call("dig_ce_rows_ws", L.ptr(logits), n, logits.shape[1], label_offset, cf(gscale), L.ptr(out3),
           L.ptr(_loss_workspace(logits.device, "ce", 1 + 3 * n)), L.stream())


def infonce_finish(stats, loss_scale, acc_scale):
    """(contra [1], accs [4]) from the two dig_ce_rows triples `stats` [2, 3] in one launch."""
    contra = torch.empty((), device=stats.device, dtype=F32)
    accs = torch.empty(4, device=stats.device, dtype=F32)
    L.call("dig_infonce_finish", L.ptr(stats), cf(loss_scale), cf(acc_scale), L.ptr(contra), L.ptr(accs), L.stream())
    return contra, accs


def step_meters(loss, contra, pixel, accs4, counts, grad_norm):
    """The ten logged values of a step as one fp32 vector (one launch): loss, contra, pixel, accs4, min / max of counts, grad_norm (None -> NaN)."""
    out = torch.empty(10, device=loss.device, dtype=F32)
    L.call("dig_step_meters", L.ptr(loss), L.ptr(contra), L.ptr(pixel), L.ptr(accs4), L.ptr(counts), counts.numel(), L.ptr(grad_norm), L.ptr(out),
           L.stream())
    return out


def cast_f32_to_bf16(x, y, n=None):
    L.call("dig_cast_f32_to_bf16", L.ptr(x), L.ptr(y), cll(x.numel() if n is None else n), L.stream())


def cast_bf16_to_f32(x, y, n=None):
    L.call("dig_cast_bf16_to_f32", L.ptr(x), L.ptr(y), cll(x.numel() if n is None else n), L.stream())


def fill_f32(x, value=0.0):
    L.call("dig_fill_f32", L.ptr(x), cll(x.numel()), cf(value), L.stream())


def scale_f32(x, s):
    L.call("dig_scale_f32", L.ptr(x), cll(x.numel()), cf(s), L.stream())


def scale_by_device_scalar(x, scalar, extra=1.0):
    L.call("dig_scale_by_device_scalar", L.ptr(x), cll(x.numel()), L.ptr(scalar), cf(extra), L.stream())


def pad_cast_rows(src, dst, M, C):
    L.call("dig_pad_cast_rows", L.ptr(src), L.ptr(dst), M, C, dst.shape[0], dst.stride(0), L.stream())


def axpy_f32(y, x, a=1.0):
    L.call("dig_axpy_f32", L.ptr(y), L.ptr(x), cll(x.numel()), cf(a), L.stream())


def ema_update(pm, p, shadow, n, m):
    """pm = pm*m + p*(1-m).  m: a Python float, or a device fp32 tensor [m, 1-m] (captured-graph form: the value is read at run time)."""
    if isinstance(m, torch.Tensor):
        L.call("dig_ema_update_dev", L.ptr(pm), L.ptr(p), L.ptr(shadow), cll(n), L.ptr(m), L.stream())
    else:
        L.call("dig_ema_update", L.ptr(pm), L.ptr(p), L.ptr(shadow), cll(n), cf(m), L.stream())


def sumsq(x, workspace, out):
    L.call("dig_sumsq", L.ptr(x), cll(x.numel()), L.ptr(workspace), L.ptr(out), L.stream())


def adamw_step(p, g, m, v, shadow, group_flags, lr0, wd0, lr1, wd1, beta1, beta2, eps, step, grad_scale=1.0, finite_gate=None):
    L.call("dig_adamw_step", L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.ptr(shadow), cll(p.numel()), L.ptr(group_flags), cf(lr0),
           cf(wd0), cf(lr1), cf(wd1), cf(beta1), cf(beta2), cf(eps), int(step), cf(grad_scale), L.ptr(finite_gate), L.stream())


def adamw_step_tr(p, g, m, v, shadow, group_flags, lr0, wd0, lr1, wd1, beta1, beta2, eps, step, mats, n_mats, n_tiles, tr_out, grad_scale=1.0,
                  finite_gate=None):
    """dig_adamw_step + the transposed bf16 copies of the weights listed in the device table `mats` (written into tr_out)."""
    L.call("dig_adamw_step_tr", L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.ptr(shadow), cll(p.numel()), L.ptr(group_flags), cf(lr0),
           cf(wd0), cf(lr1), cf(wd1), cf(beta1), cf(beta2), cf(eps), int(step), cf(grad_scale), L.ptr(finite_gate), L.ptr(mats), int(n_mats),
           int(n_tiles), L.ptr(tr_out), L.stream())


def adamw_step_dev(p, g, m, v, shadow, group_flags, scalars6, beta1, beta2, eps, grad_scale=1.0, finite_gate=None):
    """dig_adamw_step with (lr0, wd0, lr1, wd1, 1/bc1, 1/sqrt(bc2)) read from the device tensor scalars6 at run time."""
    L.call("dig_adamw_step_dev", L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.ptr(shadow), cll(p.numel()), L.ptr(group_flags),
           L.ptr(scalars6), cf(beta1), cf(beta2), cf(eps), cf(grad_scale), L.ptr(finite_gate), L.stream())


def adamw_bias_corrections(beta1, beta2, step):
    """(1/(1-beta1^t), 1/sqrt(1-beta2^t)) as fp32, computed by the library exactly as dig_adamw_step does."""
    import ctypes
    out = (ctypes.c_float * 2)()
    L.call("dig_adamw_bias_corrections", cf(beta1), cf(beta2), int(step), ctypes.cast(out, ctypes.c_void_p))
    return float(out[0]), float(out[1])
