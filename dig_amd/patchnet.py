"""PatchNet with its patch transformer (`--patchnet_name regular`, the reference CLI's default: run_mae_pretraining_moco.py:145;
modeling_pretrain_moco_mim_ori.py:137-205 PatchNet, :21-135 its cross-attention Attention / Block) on the hot-path kernels.

forward:  the pooled windows [n_img * nw, D] (dig_window_pool_fwd) attend over ALL tokens of their image through `depth` (2) blocks
    yn = LN1(y); kn = LN1(tokens)                       (the block's own norm1 on queries AND keys / values, eps 1e-5)
    y  = yn + proj(softmax(q k^T / sqrt(64)) v)         q = Wq yn, k = Wk kn, v = Wv kn (no bias)  -- residual on the NORMALISED queries (:107-121)
    y  = y + fc2(gelu(fc1(LN2(y))))
then a final LayerNorm.  Launches per block: two LayerNorms, the q GEMM, ONE k|v GEMM over the image tokens (linear_k / linear_v weights are
neighbours in the arena: a [2 D, D] operand), dig_seq_attn_fwd (nw <= 32 queries x 256 keys per (image, head)), proj + residual, LN2, fc1 + GELU,
fc2 + residual -- the big ones (LN over 2 B x 256 rows, the k|v GEMM: 2 x 19.3 GFLOP at ViT-S B = 128) are the encoder's own kernels.
backward: the explicit reverse; the gradient w.r.t. the image tokens (through both blocks' keys / values) is accumulated by the LayerNorm
backward's residual input and handed back to the caller, which adds the pooling gradient and continues into the encoder."""
import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32
L, cf = ops.L, ops.cf
EPS = 1e-5


def _names(pre, i):
    b = f"{pre}.blocks.{i}."
    return {k: b + k for k in ("norm1.weight", "norm1.bias", "attn.linear_q.weight", "attn.linear_k.weight", "attn.linear_v.weight",
                               "attn.proj.weight", "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
                               "mlp.fc2.weight", "mlp.fc2.bias")}


def _kv_view(model, flat, name_k, dtype_numel=1):
    """[2 D, D] view over linear_k.weight | linear_v.weight (consecutive parameters of the arena, D * D a multiple of the 256-element granule)."""
    sp = model.specs[name_k]
    D = sp.shape[0]
    assert model.specs[name_k.replace("linear_k", "linear_v")].offset == sp.offset + D * D
    return flat[sp.offset:sp.offset + 2 * D * D].view(2 * D, D)


def forward(step, feat, pooled, pre, arena, n_img, save):
    """feat: bf16 [n_img * N, D] image tokens; pooled: bf16 [n_img * nw, D].  Returns (out [n_img * nw, D], saved or None)."""
    M = step.m
    D, H, N, nw = M.D, M.H, M.N, M.num_windows
    w16, f32 = M._w(arena), M._f32
    sh = M.shadow(arena)
    dev = feat.device
    scale = (D // H) ** -0.5
    y = pooled
    blocks = []
    for i in range(M.patchnet_depth):
        n = _names(pre, i)
        yn, mux, rsx = ops.layernorm_fwd(y, f32[n["norm1.weight"]], f32[n["norm1.bias"]], EPS)
        kn, muk, rsk = ops.layernorm_fwd(feat, f32[n["norm1.weight"]], f32[n["norm1.bias"]], EPS)
        q = ops.linear_fwd(yn, w16[n["attn.linear_q.weight"]])
        if N == 256 and D == H * 64:
            # on the encoder's MFMA attention kernels (as the recognition decoder's cross-attention, dig_amd/finetune.py): the nw queries of
            # an image sit in rows [0, nw) of a fused q | k | v buffer of 256 rows per image, the k | v GEMM writes its columns [D, 3D), and the
            # kernels compute the first query block only (q_rows); rows nw..31 are zero queries with a zero output gradient.
            # (dig_seq_attn_fwd / _bwd -- a thread per key -- took 257 / 426 us per launch here: 1.9 ms per step for 5 x 256 scores per head)
            kv = torch.empty((n_img * N, 3 * D), device=dev, dtype=BF16)
            ops.gemm(kn, _kv_view(M, sh, n["attn.linear_k.weight"]), n_img * N, 2 * D, D, out=kv[:, D:], ldc=3 * D)
            fq = kv.view(n_img, N, 3 * D)[:, :, :D]
            fq[:, nw:32].zero_()
            fq[:, :nw] = (q * scale).view(n_img, nw, D)                 # (scale = 2^-3: exact in bf16)
            ctx, lse = ops.attn_fwd(kv, n_img, H, D, q_rows=nw)
            a = ctx.view(n_img, N, D)[:, :nw].reshape(n_img * nw, D)
            lse = (lse, ctx)
        else:
            kv = ops.linear_fwd(kn, _kv_view(M, sh, n["attn.linear_k.weight"]))
            a = torch.empty((n_img * nw, D), device=dev, dtype=BF16)
            lse = torch.empty((n_img, H, nw), device=dev, dtype=F32)
            L.call("dig_seq_attn_fwd", L.ptr(q), D, L.ptr(kv), 2 * D, L.ptr(kv[:, D:]), 2 * D, L.ptr(a), D, L.ptr(lse), n_img, H, nw, N, cf(scale), 0,
                   None, L.stream())
        y1 = ops.linear_fwd(a, w16[n["attn.proj.weight"]], bias=f32[n["attn.proj.bias"]], resid=yn)
        h2, mu2, rs2 = ops.layernorm_fwd(y1, f32[n["norm2.weight"]], f32[n["norm2.bias"]], EPS)
        pre_act = torch.empty((n_img * nw, M.F), device=dev, dtype=BF16) if save else None
        u = ops.linear_fwd(h2, w16[n["mlp.fc1.weight"]], bias=f32[n["mlp.fc1.bias"]], act=1, pre=pre_act)
        y2 = ops.linear_fwd(u, w16[n["mlp.fc2.weight"]], bias=f32[n["mlp.fc2.bias"]], resid=y1)
        if save:
            blocks.append((y, mux, rsx, yn, muk, rsk, kn, q, kv, a, lse, y1, h2, mu2, rs2, pre_act, u))
        y = y2
    out, muo, rso = ops.layernorm_fwd(y, f32[pre + ".norm.weight"], f32[pre + ".norm.bias"], EPS)
    return out, ((feat, blocks, y, muo, rso) if save else None)


def backward(step, dout, pre, saved, n_img):
    """dout: bf16 [n_img * nw, D].  Accumulates every parameter gradient of `pre` (online arena) and returns (d pooled [n_img * nw, D],
    d feat [n_img * N, D]: the gradient w.r.t. the image tokens through both blocks' keys / values)."""
    M = step.m
    D, H, N, nw = M.D, M.H, M.N, M.num_windows
    w16, f32, g32 = M._w("online"), M._f32, M._g32
    sh, gflat = M.shadow("online"), M._flat["grad"]
    dev = dout.device
    scale = (D // H) ** -0.5
    feat, blocks, y_last, muo, rso = saved
    side = lambda fn, *t: step._on_side(dev, fn, *t)                     # noqa: E731  (weight gradients / column sums: consumers only)
    dy = ops.layernorm_bwd(dout, y_last, f32[pre + ".norm.weight"], f32[pre + ".norm.bias"], muo, rso, None, g32[pre + ".norm.weight"],
                           g32[pre + ".norm.bias"])
    dfeat = None
    for i in reversed(range(M.patchnet_depth)):
        n = _names(pre, i)
        y, mux, rsx, yn, muk, rsk, kn, q, kv, a, lse, y1, h2, mu2, rs2, pre_act, u = blocks[i]
        blocks[i] = None
        # y2 = y1 + fc2(gelu(fc1(LN2(y1))))
        side(lambda dy=dy, u=u, n=n: (ops.linear_wgrad(dy, u, g32[n["mlp.fc2.weight"]]), ops.colsum(dy, g32[n["mlp.fc2.bias"]])), dy, u)
        dact, bparts = ops.linear_dgrad(dy, w16[n["mlp.fc2.weight"]], gelu_pre=pre_act, colsum=True)
        side(lambda dact=dact, h2=h2, bparts=bparts, n=n: (ops.linear_wgrad(dact, h2, g32[n["mlp.fc1.weight"]]),
                                                           ops.colsum_partials(bparts, g32[n["mlp.fc1.bias"]])), dact, h2, bparts)
        dh2 = ops.linear_dgrad(dact, w16[n["mlp.fc1.weight"]])
        dy1 = ops.layernorm_bwd(dh2, y1, f32[n["norm2.weight"]], f32[n["norm2.bias"]], mu2, rs2, dy, g32[n["norm2.weight"]], g32[n["norm2.bias"]])
        # y1 = yn + proj(a)
        side(lambda dy1=dy1, a=a, n=n: (ops.linear_wgrad(dy1, a, g32[n["attn.proj.weight"]]), ops.colsum(dy1, g32[n["attn.proj.bias"]])), dy1, a)
        da = ops.linear_dgrad(dy1, w16[n["attn.proj.weight"]])
        if isinstance(lse, tuple):                                       # MFMA path (see forward): kv = the fused q | k | v buffer
            lse_, ctx = lse
            dctx = torch.empty((n_img * N, D), device=dev, dtype=BF16)
            dv_ = dctx.view(n_img, N, D)
            dv_[:, nw:32].zero_()
            dv_[:, :nw] = da.view(n_img, nw, D)
            dfused = ops.attn_bwd(kv, ctx, dctx, lse_, n_img, H, D, scale, q_rows=nw)
            dq = dfused.view(n_img, N, 3 * D)[:, :nw, :D].reshape(n_img * nw, D)
            dkv = dfused[:, D:]                                           # [n_img * N, 2 D] view, row stride 3 D
        else:
            dq = torch.empty_like(q)
            dkv = torch.empty_like(kv)
            L.call("dig_seq_attn_bwd", L.ptr(q), D, L.ptr(kv), 2 * D, L.ptr(kv[:, D:]), 2 * D, L.ptr(da), D, L.ptr(lse), L.ptr(dq), D, L.ptr(dkv), 2 * D,
                   L.ptr(dkv[:, D:]), 2 * D, n_img, H, nw, N, cf(scale), 0, None, L.stream())
        side(lambda dq=dq, yn=yn, n=n: ops.linear_wgrad(dq, yn, g32[n["attn.linear_q.weight"]]), dq, yn)
        gkv = _kv_view(M, gflat, n["attn.linear_k.weight"])
        side(lambda dkv=dkv, kn=kn, gkv=gkv: ops.wgrad(dkv, kn, gkv, 2 * D, D, kn.shape[0]), dkv, kn)
        # d yn = dy1 (the residual runs on the NORMALISED queries) + dq Wq: the GEMM's residual epilogue adds them
        rows = dq.shape[0]
        dyn = ops.gemm(dq, w16[n["attn.linear_q.weight"]], rows, D, D, tb=True, resid=dy1, bk=ops.dgrad_tile_code(rows, D))
        dkn = ops.linear_dgrad(dkv, _kv_view(M, sh, n["attn.linear_k.weight"]))
        # norm1 of this block normalised the queries and the image tokens: both gradients land in its parameters' sums
        dy = ops.layernorm_bwd(dyn, y, f32[n["norm1.weight"]], f32[n["norm1.bias"]], mux, rsx, None, g32[n["norm1.weight"]], g32[n["norm1.bias"]])
        dfeat = ops.layernorm_bwd(dkn, feat, f32[n["norm1.weight"]], f32[n["norm1.bias"]], muk, rsk, dfeat, g32[n["norm1.weight"]],
                                  g32[n["norm1.bias"]], out=dkn)
    return dy, dfeat
