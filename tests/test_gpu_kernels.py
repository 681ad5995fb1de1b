"""Operator-level parity: every C-ABI op of the pre-training step against a plain PyTorch fp32 reference of the same op
(bf16 I/O => tolerance 1e-2 relative Frobenius; fp32 / integer paths tight or exact).  Every test runs twice: on the MI355X
through libdig_hip.so (`hip`, marked gpu) and in the GPU-less container through cpu_abi/libdig_cpu.so, the plain-C++ build of the
same entry points (`cpu_abi`; same dig_amd/ops.py host code, large shapes skipped: the loops are not meant to be fast)."""
import ctypes
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import rel

cf = ctypes.c_float


@pytest.fixture(scope="module", params=[pytest.param("hip", marks=pytest.mark.gpu), "cpu_abi"])
def dev(request):
    torch.manual_seed(0)
    if request.param == "hip":
        assert torch.cuda.is_available(), "GPU tests need an MI355X"
        from dig_amd import _lib
        _lib.lib()
        yield torch.device("cuda:0")
    else:
        from cpu_abi_util import cpu_abi_backend
        with cpu_abi_backend() as d:
            yield d


def cpu_limit(dev, work, limit=6e9):
    """The plain-loop build covers the small and ragged shapes; the big ones stay on the GPU."""
    if dev.type == "cpu" and work > limit:
        pytest.skip("cpu_abi: large shape, covered by the hip run")


@pytest.mark.parametrize("I,J,R", [(256, 256, 128), (2048, 1152, 384), (716, 48, 192), (8192, 1536, 384), (1024, 4096, 4096),
                                   (300, 200, 64), (32, 64, 256), (129, 136, 192)])
@pytest.mark.parametrize("bk", [32, 64])
def test_gemm_fwd_dgrad_wgrad(dev, I, J, R, bk):
    from dig_amd import ops
    cpu_limit(dev, I * J * R * (1 if bk == 64 else 1e9))            # (the tile code means nothing to the CPU build: one pass)
    ops.GEMM_BK_FWD = ops.GEMM_BK_BWD = bk
    try:
        x = torch.randn(I, R, device=dev).bfloat16()
        w = (torch.randn(J, R, device=dev) * 0.05).bfloat16()
        bias = torch.randn(J, device=dev)
        res = torch.randn(I, J, device=dev).bfloat16()
        h = x.float() @ w.float().t() + bias
        assert rel(ops.linear_fwd(x, w, bias=bias, resid=res), h + res.float()) < 1e-2
        pre = torch.empty(I, J, device=dev, dtype=torch.bfloat16)
        y = ops.linear_fwd(x, w, bias=bias, pre=pre, act=1)
        assert rel(y, F.gelu(h)) < 1e-2 and rel(pre, h) < 1e-2
        ac = (J // 16) * 8
        ref = h.clone(); ref[:, :ac] *= 0.125
        assert rel(ops.linear_fwd(x, w, bias=bias, alpha=0.125, alpha_cols=ac, out_kind=ops.OUT_F32), ref) < 1e-5
        dy = torch.randn(I, J, device=dev).bfloat16()
        if J % 64 == 0:
            assert rel(ops.linear_dgrad(dy, w), dy.float() @ w.float()) < 1e-2
            prex = torch.randn(I, R, device=dev).bfloat16()
            pp = prex.float().requires_grad_(True)
            F.gelu(pp).backward(dy.float() @ w.float())
            assert rel(ops.linear_dgrad(dy, w, gelu_pre=prex), pp.grad) < 1e-2
            for tile in (32, 244):                                                  # 128x128 kernel and the 256x256 wide kernel
                saved, ops.DGRAD_GELU_BK = ops.DGRAD_GELU_BK, tile
                try:
                    dxg, parts = ops.linear_dgrad(dy, w, gelu_pre=prex, colsum=True)   # fused bias gradient of the layer below
                finally:
                    ops.DGRAD_GELU_BK = saved
                bsum = torch.randn(R, device=dev); b0 = bsum.clone()
                ops.colsum_partials(parts, bsum)
                assert rel(dxg, pp.grad) < 1e-2 and rel(bsum - b0, pp.grad.sum(0)) < 1e-3
        dW = torch.randn(J, R, device=dev); dW0 = dW.clone()
        ops.linear_wgrad(dy, x, dW)
        assert rel(dW, dW0 + dy.float().t() @ x.float()) < 2e-5        # fp32 accumulate, deterministic
        dW2 = dW0.clone()
        ops.linear_wgrad(dy, x, dW2)
        assert torch.equal(dW, dW2)                                     # bit-reproducible (no atomics)
    finally:
        ops.GEMM_BK_FWD, ops.GEMM_BK_BWD = 64, 32


@pytest.mark.parametrize("I,J,R", [(2048, 1152, 384), (716, 200, 192), (8192, 1536, 384), (300, 520, 64), (8192, 384, 1536)])
@pytest.mark.parametrize("bk", [244, 264, 212, 221])
def test_gemm_wide_tile_variants(dev, I, J, R, bk):
    """Every multi-wave tile shape of gemm_wide_kernel (WM x WN waves of FM x FN MFMA blocks) on full and ragged tiles:
    forward epilogues, transposed-B (dgrad) and split-R partial slabs (wgrad)."""
    from dig_amd import ops
    cpu_limit(dev, I * J * R * (1 if bk == 244 else 1e9))
    x = torch.randn(I, R, device=dev).bfloat16()
    w = (torch.randn(J, R, device=dev) * 0.05).bfloat16()
    bias = torch.randn(J, device=dev)
    res = torch.randn(I, J, device=dev).bfloat16()
    h = x.float() @ w.float().t() + bias
    assert rel(ops.gemm(x, w, I, J, R, bias=bias, resid=res, bk=bk), h + res.float()) < 1e-2
    pre = torch.empty(I, J, device=dev, dtype=torch.bfloat16)
    y = ops.gemm(x, w, I, J, R, bias=bias, pre=pre, act=1, bk=bk)
    assert rel(y, F.gelu(h)) < 1e-2 and rel(pre, h) < 1e-2
    assert rel(ops.gemm(x, w, I, J, R, bias=bias, out_kind=ops.OUT_F32, bk=bk), h) < 1e-5
    dy = torch.randn(I, J, device=dev).bfloat16()
    if J % 64 == 0:
        assert rel(ops.gemm(dy, w, I, R, J, tb=True, bk=bk), dy.float() @ w.float()) < 1e-2
    sp = ops.L.lib().dig_gemm_effective_splits(I, 3)
    ws = torch.empty(sp, J, R, device=dev)
    ops.gemm(dy, x, J, R, I, ta=True, tb=True, out=ws, out_kind=ops.OUT_F32_PARTIAL, splits=sp, ldc=R, bk=bk)
    assert rel(ws.sum(0), dy.float().t() @ x.float()) < 2e-5


@pytest.mark.parametrize("single_pass", [False, True])
@pytest.mark.parametrize("Bn,H,scale,spike", [(2, 2, 1.0, False), (4, 6, 0.125, False), (3, 8, 0.125, True)])
def test_attention_fwd_bwd(dev, Bn, H, scale, spike, single_pass):
    """single_pass: the backward as one 8-wave workgroup per (image, head) (dig_attn_bwd_mode(1): five matrix products per tile pair, dQ summed
    in LDS in a fixed order) instead of the default two-phase kernel -- same bands."""
    from dig_amd import ops
    if single_pass and dev.type == "cpu":
        pytest.skip("one backward form in the CPU build")
    prev = ops.attn_bwd_mode(single_pass) if dev.type != "cpu" else False
    try:
        _attention_fwd_bwd(dev, Bn, H, scale, spike)
    finally:
        if dev.type != "cpu":
            ops.attn_bwd_mode(prev)


def _attention_fwd_bwd(dev, Bn, H, scale, spike):
    from dig_amd import ops
    D = H * 64
    qkv = torch.randn(Bn * 256, 3 * D, device=dev).bfloat16()
    if spike:                                                           # force a peaked softmax row (dominant key)
        qkv[5, :64] *= 6.0
        qkv[77, D:D + 64] *= 6.0
    ctx, lse = ops.attn_fwd(qkv, Bn, H, D)
    x = qkv.float().requires_grad_(True)
    t = x.reshape(Bn, 256, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = t[0] @ t[1].transpose(-2, -1)
    o = (s.softmax(-1) @ t[2]).transpose(1, 2).reshape(Bn * 256, D)
    assert rel(ctx, o) < 1e-2
    assert (lse - torch.logsumexp(s, -1).reshape(Bn * H, 256)).abs().max().item() < 2e-2
    dctx = torch.randn(Bn * 256, D, device=dev).bfloat16()
    o.backward(dctx.float())
    g = x.grad.clone(); g[:, :D] *= scale
    dqkv = ops.attn_bwd(qkv, ctx, dctx, lse, Bn, H, D, scale)
    for lo in (0, D, 2 * D):
        assert rel(dqkv[:, lo:lo + D], g[:, lo:lo + D]) < 2e-2
    dqkv2, qs, vs = ops.attn_bwd(qkv, ctx, dctx, lse, Bn, H, D, scale, bias_sums=True)      # fused q_bias / v_bias gradients
    assert torch.equal(dqkv2, dqkv)
    assert torch.equal(ops.attn_bwd(qkv, ctx, dctx, lse, Bn, H, D, scale), dqkv)            # bit-reproducible run to run
    if dev.type != "cpu":                                                                   # the three store forms of the two-phase kernel: the same bits
        prev = ops.attn_bwd_store()
        try:
            for mode in (0, 1, 3):
                ops.attn_bwd_store(mode)
                d3, q3, v3 = ops.attn_bwd(qkv, ctx, dctx, lse, Bn, H, D, scale, bias_sums=True)
                # (the q sums of the full-line form add up the rows as stored -- bf16 -- the row-store forms the fp32 accumulators: one bias gradient,
                #  two roundings; the v sums are the column sums of d(ctx) in every form)
                assert torch.equal(d3, dqkv) and torch.equal(v3, vs) and (torch.equal(q3, qs) if mode == 3 else rel(q3, qs) < 2e-3), mode
        finally:
            ops.attn_bwd_store(prev)
    bq, bv = torch.randn(D, device=dev), torch.randn(D, device=dev)
    bq0, bv0 = bq.clone(), bv.clone()
    ops.colsum_partials(qs, bq); ops.colsum_partials(vs, bv)
    assert rel(bq - bq0, g[:, :D].sum(0)) < 2e-2 and rel(bv - bv0, g[:, 2 * D:].sum(0)) < 2e-2     # same tolerance as dq, dv


@pytest.mark.parametrize("Bn,H", [(2, 2), (8, 6), (3, 8)])
def test_attention_bwd_with_projection_gradient(dev, Bn, H):
    """dig_attn_bwd_proj: d(ctx) = dy @ proj.weight computed inside the attention backward (per (image, head) workgroup, fp32 accumulation, rounded to
    bf16) against the two-launch form (data-gradient GEMM -> dig_attn_bwd) and against fp32 autograd of the whole sub-block."""
    from dig_amd import ops
    D, scale = H * 64, 0.125
    R = Bn * 256
    qkv = torch.randn(R, 3 * D, device=dev).bfloat16()
    ctx, lse = ops.attn_fwd(qkv, Bn, H, D)
    Wp = (torch.randn(D, D, device=dev) * D ** -0.5).bfloat16()          # proj.weight [out, in]
    dy = torch.randn(R, D, device=dev).bfloat16()
    projt = Wp.t().contiguous()                                          # [in][out]
    dctx = (dy.float() @ Wp.float()).bfloat16()                          # what the projection's data-gradient GEMM writes
    ref, qs0, vs0 = ops.attn_bwd(qkv, ctx, dctx, lse, Bn, H, D, scale, bias_sums=True)
    got, qs, vs = ops.attn_bwd_proj(qkv, ctx, dy, projt, lse, Bn, H, D, scale, bias_sums=True)
    # the same arithmetic up to the summation order inside d(ctx) (a few bf16 roundings of d(ctx) differ by one ulp)
    for lo in (0, D, 2 * D):
        assert rel(got[:, lo:lo + D], ref[:, lo:lo + D]) < 4e-3
    assert rel(qs, qs0) < 4e-3 and rel(vs, vs0) < 4e-3
    assert torch.equal(ops.attn_bwd_proj(qkv, ctx, dy, projt, lse, Bn, H, D, scale), got)          # bit-reproducible run to run
    # fp32 autograd of y = softmax(q k^T) v Wp^T
    x = qkv.float().requires_grad_(True)
    t = x.reshape(Bn, 256, 3, H, 64).permute(2, 0, 3, 1, 4)
    o = ((t[0] @ t[1].transpose(-2, -1)).softmax(-1) @ t[2]).transpose(1, 2).reshape(R, D)
    (o @ Wp.float().t()).backward(dy.float())
    g = x.grad.clone(); g[:, :D] *= scale
    for lo in (0, D, 2 * D):
        assert rel(got[:, lo:lo + D], g[:, lo:lo + D]) < 2e-2
    if dev.type != "cpu":
        bad = torch.randn(256, 3 * 192, device=dev).bfloat16()           # D = 192: not a multiple of 128 -> refused, nothing launched
        with pytest.raises(RuntimeError):
            ops.attn_bwd_proj(bad, bad[:, :192].contiguous(), bad[:, :192].contiguous(), torch.zeros(192, 192, device=dev).bfloat16(),
                              torch.zeros(3, 256, device=dev), 1, 3, 192, scale)


@pytest.mark.parametrize("D,gelu", [(384, 0), (512, 0), (128, 0), (192, 1), (64, 1), (256, 0)])
def test_layernorm(dev, D, gelu):
    from dig_amd import ops
    rows = 1000 + (3 if D in (384, 192) else 0)                         # (1003: a last group of rows shorter than the 4 / 2 rows a wave keeps in flight)
    x = torch.randn(rows, D, device=dev).bfloat16()
    g = torch.randn(D, device=dev) * 0.2 + 1
    b = torch.randn(D, device=dev) * 0.1
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6, gelu=bool(gelu))
    xf, gf, bf = x.float().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xf, (D,), gf, bf, 1e-6)
    ref = F.gelu(ref) if gelu else ref
    dy, dres = torch.randn(rows, D, device=dev).bfloat16(), torch.randn(rows, D, device=dev).bfloat16()
    ref.backward(dy.float())
    dg, db, dc = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    dx = ops.layernorm_bwd(dy, x, g, b, mean, rstd, dres, dg, db, gelu=bool(gelu), dres_colsum=dc)
    assert rel(y, ref) < 1e-2 and rel(dx, xf.grad + dres.float()) < 1e-2
    assert rel(dg, gf.grad) < 1e-4 and rel(db, bf.grad) < 1e-4 and rel(dc, dres.float().sum(0)) < 1e-4


@pytest.mark.parametrize("rows,C,affine,relu", [(1024, 4096, 1, 1), (4096, 512, 1, 1), (333, 256, 0, 0), (32, 64, 0, 0)])
def test_batchnorm(dev, rows, C, affine, relu):
    from dig_amd import ops
    x = (torch.randn(rows, C, device=dev) * 2 + 0.5).bfloat16()
    gamma = (torch.randn(C, device=dev) * 0.2 + 1) if affine else None
    beta = (torch.randn(C, device=dev) * 0.1) if affine else None
    sums = torch.empty(2, C, device=dev)
    ops.bn_stats(x, sums)
    y, mean, rstd = ops.bn_fwd_apply(x, sums, float(rows), 1e-5, gamma, beta, relu)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ops.bn_update_running(sums, float(rows), 0.1, rm, rv)
    xf = x.float().requires_grad_(True)
    gf = gamma.clone().requires_grad_(True) if affine else None
    bf = beta.clone().requires_grad_(True) if affine else None
    trm, trv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ref = F.batch_norm(xf, trm, trv, gf, bf, True, 0.1, 1e-5)
    ref = F.relu(ref) if relu else ref
    dy = torch.randn(rows, C, device=dev).bfloat16()
    ref.backward(dy.float())
    s2 = torch.empty(2, C, device=dev)
    ops.bn_bwd_stats(dy, x, mean, rstd, gamma, beta, relu, s2)
    dx = ops.bn_bwd_apply(dy, x, mean, rstd, gamma, beta, relu, s2, float(rows))
    assert rel(y, ref) < 1e-2 and rel(dx, xf.grad) < 1.5e-2
    assert rel(rm, trm) < 1e-4 and rel(rv, trv) < 1e-4                 # running stats incl. the n/(n-1) factor
    if affine:
        assert rel(s2[1], gf.grad) < 1e-4 and rel(s2[0], bf.grad) < 1e-4
    # the forward apply updating the running statistics itself (dig_bn_fwd_apply_running): same outputs, same running statistics
    rm2, rv2 = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    y2, mean2, rstd2 = ops.bn_fwd_apply(x, sums, float(rows), 1e-5, gamma, beta, relu, running=(rm2, rv2, 0.1))
    assert torch.equal(y2, y) and torch.equal(mean2, mean) and torch.equal(rstd2, rstd)
    assert rel(rm2, rm) < 1e-6 and rel(rv2, rv) < 1e-6
    # the same launch accumulating the affine gradients from the local sums (dig_bn_bwd_stats_acc): sums unchanged, acc += sums, bit for bit
    s3, db, dg = torch.empty(2, C, device=dev), torch.full((C,), 0.5, device=dev), torch.full((C,), -0.25, device=dev)
    ops.bn_bwd_stats(dy, x, mean, rstd, gamma, beta, relu, s3, db, dg)
    assert torch.equal(s3, s2) and torch.equal(db, 0.5 + s2[0]) and torch.equal(dg, -0.25 + s2[1])


@pytest.mark.parametrize("rows,C,affine,relu", [(1024, 4096, True, True), (1024, 256, False, False), (2048, 512, True, True), (40, 256, True, False),
                                                (333, 288, True, True)])
def test_batchnorm_fused_few_rows(dev, rows, C, affine, relu):
    """dig_bn_fwd_fused / dig_bn_bwd_fused (a few-row BatchNorm layer of a single rank in ONE launch each) against the three-launch path of
    the same library -- the same expressions, another summation order over rows: statistics to fp32 round-off, outputs within one bf16
    rounding -- and against torch."""
    from dig_amd import ops
    assert ops.bn_fused_supported(rows, C) and not ops.bn_fused_supported(32768, 512) and not ops.bn_fused_supported(1024, 64)
    x = (torch.randn(rows, C, device=dev) * 2 + 0.5).bfloat16()
    gamma = (torch.randn(C, device=dev) * 0.2 + 1) if affine else None
    beta = (torch.randn(C, device=dev) * 0.1) if affine else None
    sums = torch.empty(2, C, device=dev)
    ops.bn_stats(x, sums)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    y, mean, rstd = ops.bn_fwd_apply(x, sums, float(rows), 1e-5, gamma, beta, relu, running=(rm, rv, 0.1))
    rm2, rv2 = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    y2, mean2, rstd2 = ops.bn_fwd_fused(x, 1e-5, gamma, beta, relu, running=(rm2, rv2, 0.1))
    assert rel(mean2, mean) < 1e-5 and rel(rstd2, rstd) < 1e-5 and rel(rm2, rm) < 1e-5 and rel(rv2, rv) < 1e-5
    assert (y2.float() - y.float()).abs().max().item() <= 2 ** -7 * y.float().abs().max().item() and rel(y2, y) < 1e-3
    dy = torch.randn(rows, C, device=dev).bfloat16()
    s2, db, dg = torch.empty(2, C, device=dev), torch.full((C,), 0.5, device=dev), torch.full((C,), -0.25, device=dev)
    ops.bn_bwd_stats(dy, x, mean, rstd, gamma, beta, relu, s2, db, dg)
    dx = ops.bn_bwd_apply(dy, x, mean, rstd, gamma, beta, relu, s2, float(rows))
    db2, dg2 = torch.full((C,), 0.5, device=dev), torch.full((C,), -0.25, device=dev)
    dx2 = ops.bn_bwd_fused(dy, x, mean, rstd, gamma, beta, relu, db2, dg2)
    assert rel(db2, db) < 1e-5 and rel(dg2, dg) < 1e-5 and rel(dx2, dx) < 2e-3
    dx3 = ops.bn_bwd_fused(dy, x, mean, rstd, gamma, beta, relu)            # (the last layer of a stack: no affine gradients)
    assert torch.equal(dx3, dx2)
    xf = x.float().requires_grad_(True)
    ref = F.batch_norm(xf, torch.zeros(C, device=dev), torch.ones(C, device=dev), gamma, beta, True, 0.1, 1e-5)
    ref = F.relu(ref) if relu else ref
    ref.backward(dy.float())
    assert rel(y2, ref) < 1e-2 and rel(dx2, xf.grad) < 1.5e-2


@pytest.mark.parametrize("Bn,D", [(6, 384), (3, 128)])
def test_patch_embed(dev, Bn, D):
    from dig_amd import ops
    img = torch.rand(Bn, 3, 32, 128, device=dev) * 2 - 1
    W = torch.randn(D, 3, 4, 4, device=dev) * 0.1
    bias, mt, pos = torch.randn(D, device=dev) * 0.1, torch.randn(D, device=dev) * 0.1, torch.randn(256, D, device=dev)
    mask = torch.rand(Bn, 256, device=dev) < 0.5
    m8 = mask.to(torch.uint8)
    out = ops.patch_embed_fwd(img, W.view(D, 48), bias, m8, mt, pos, D, 8, 32)
    Wf, bf, mtf = W.clone().requires_grad_(True), bias.clone().requires_grad_(True), mt.clone().requires_grad_(True)
    mm = mask.unsqueeze(-1).float()
    ref = F.conv2d(img, Wf, bf, stride=4).flatten(2).transpose(1, 2) * (1 - mm) + mtf * mm + pos
    dy = torch.randn(Bn * 256, D, device=dev).bfloat16()
    ref.reshape(-1, D).backward(dy.float())
    assert rel(out, ref.reshape(-1, D)) < 1e-2
    for fn in (ops.patch_embed_bwd, ops.patch_embed_bwd_mfma):
        dW, dbias, dmt = torch.zeros(D, 48, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        fn(dy, img, m8, dW, dbias, dmt, D, 8, 32)
        assert rel(dW, Wf.grad.reshape(D, 48)) < 5e-3 and rel(dbias, bf.grad) < 1e-4 and rel(dmt, mtf.grad) < 1e-4


def test_window_pool(dev):
    from dig_amd import ops
    Bn, D = 6, 384
    x = torch.randn(Bn, 256, D, device=dev).bfloat16()
    out = torch.empty(Bn * 4, D, device=dev, dtype=torch.bfloat16)
    ops.window_pool_fwd(x, out, Bn, 8, 32, 4, D)
    ref = x.float().reshape(Bn, 8, 4, 8, D).mean(dim=(1, 3)).reshape(Bn * 4, D)       # == adaptive_avg_pool2d(grid,(1,4))
    ref2 = F.adaptive_avg_pool2d(x.float().reshape(Bn, 8, 32, D).permute(0, 3, 1, 2), (1, 4)).permute(0, 2, 3, 1).reshape(Bn * 4, D)
    assert rel(ref, ref2) < 1e-6 and rel(out, ref) < 1e-2
    dp = torch.randn(Bn * 4, D, device=dev).bfloat16()
    dx = torch.randn(Bn, 256, D, device=dev).bfloat16(); dx0 = dx.clone()
    ops.window_pool_bwd(dp, dx, Bn, 8, 32, 4, D, True)
    refdx = dx0.float() + (dp.float().reshape(Bn, 1, 4, 1, D) / 64).expand(Bn, 8, 4, 8, D).reshape(Bn, 256, D)
    assert rel(dx, refdx) < 1e-2


@pytest.mark.parametrize("nwin", [5, 3, 7, 32, 1])
def test_window_pool_uneven_windows(dev, nwin):
    """--num_windows 5, the reference CLI's default (run_mae_pretraining_moco.py:143): adaptive_avg_pool2d with overlapping bins, forward and
    gradient against torch (PatchNet.forward, modeling_pretrain_moco_mim_ori.py:189-193)."""
    from dig_amd import ops
    Bn, D = 5, 128
    x = torch.randn(Bn, 256, D, device=dev).bfloat16()
    out = torch.empty(Bn * nwin, D, device=dev, dtype=torch.float32)
    ops.window_pool_fwd(x, out, Bn, 8, 32, nwin, D)
    xr = x.float().cpu().reshape(Bn, 8, 32, D).permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.adaptive_avg_pool2d(xr, (1, nwin))
    assert (out.cpu() - ref.permute(0, 2, 3, 1).reshape(Bn * nwin, D)).abs().max().item() < 2e-6 * 8 * 32       # fp32 sums of the same bf16 values
    dp = torch.randn(Bn * nwin, D, device=dev).bfloat16()
    ref.backward(dp.float().cpu().reshape(Bn, 1, nwin, D).permute(0, 3, 1, 2))
    refdx = xr.grad.permute(0, 2, 3, 1).reshape(Bn, 256, D)
    for acc in (False, True):
        dx = torch.randn(Bn, 256, D, device=dev).bfloat16()
        dx0 = dx.clone()
        ops.window_pool_bwd(dp, dx, Bn, 8, 32, nwin, D, acc)
        want = (refdx + (dx0.float().cpu() if acc else 0)).bfloat16()
        assert (dx.cpu().float() - want.float()).abs().max().item() <= 2 ** -7 * want.float().abs().max().item()     # one bf16 rounding


@pytest.mark.parametrize("nwin", [4, 5, 7, 32])
def test_window_pool_more_windows_than_columns(dev, nwin):
    """ConvPatchNet pools its 1 x 4 map to (1, num_windows) (modeling_pretrain_moco_mim_ori.py:254): with 5 (the CLI default) .. 32 windows on 4
    columns a column feeds several windows -- forward and gradient against torch."""
    from dig_amd import ops
    Bn, D = 3, 256
    x = torch.randn(Bn, 4, D, device=dev).bfloat16()
    out = torch.empty(Bn * nwin, D, device=dev, dtype=torch.bfloat16)
    ops.window_pool_fwd(x, out, Bn, 1, 4, nwin, D)
    xr = x.float().cpu().reshape(Bn, 1, 4, D).permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.adaptive_avg_pool2d(xr, (1, nwin))
    assert torch.equal(out.cpu(), ref.permute(0, 2, 3, 1).reshape(Bn * nwin, D).bfloat16())
    dp = torch.randn(Bn * nwin, D, device=dev).bfloat16()
    ref.backward(dp.float().cpu().reshape(Bn, 1, nwin, D).permute(0, 3, 1, 2))
    dx = torch.empty(Bn, 4, D, device=dev, dtype=torch.bfloat16)
    ops.window_pool_bwd(dp, dx, Bn, 1, 4, nwin, D, False)
    want = xr.grad.permute(0, 2, 3, 1).reshape(Bn, 4, D)
    assert (dx.cpu().float() - want).abs().max().item() <= 2 ** -7 * want.abs().max().item()


@pytest.mark.parametrize("n_img,H,W,C", [(3, 8, 32, 64), (2, 4, 16, 96), (5, 2, 8, 288), (4, 1, 4, 128)])
def test_im2col_and_weight_flip_are_bit_exact(dev, n_img, H, W, C):
    """Byte work of the convolution-as-GEMM (csrc/conv_patch.hip): the im2col matrix against F.unfold (the column order of conv.weight.view(C_out,
    -1)), its zero pad columns, and the flipped / transposed taps of the data gradient."""
    from dig_amd import ops
    x = torch.randn(n_img * H * W, C, device=dev).bfloat16()
    col = ops.im2col3x3(x, n_img, H, W, C)
    assert col.shape == (n_img * H * W, -(-9 * C // 64) * 64)
    ref = F.unfold(x.cpu().float().reshape(n_img, H, W, C).permute(0, 3, 1, 2), 3, padding=1).transpose(1, 2).reshape(n_img * H * W, 9 * C)
    assert torch.equal(col[:, :9 * C].cpu().float(), ref)
    assert not col[:, 9 * C:].any()
    co = 40
    w = torch.randn(co, C * 9, device=dev).bfloat16()
    wt = ops.conv3x3_weight_flip(w, co, C)
    want = w.cpu().reshape(co, C, 9).flip(2).permute(1, 0, 2).reshape(C, co * 9)
    assert torch.equal(wt.cpu(), want)


@pytest.mark.parametrize("n_img,H,W,C", [(3, 8, 32, 64), (2, 4, 16, 96), (5, 2, 8, 288)])
def test_maxpool2x2_matches_torch_including_ties(dev, n_img, H, W, C):
    """nn.MaxPool2d(2, 2) (modeling_pretrain_moco_mim_ori.py:219-223) after a ReLU: values, arg-max bytes (first maximum wins: whole windows of
    zeros are common) and the gradient routing, bit-exact against torch."""
    from dig_amd import ops
    x = torch.relu(torch.randn(n_img * H * W, C, device=dev)).bfloat16()
    x[::3] = torch.round(x[::3].float() * 2).bfloat16() / 2                   # coarse values: ties between positive entries too
    y, idx = ops.maxpool2x2_fwd(x, n_img, H, W, C)
    xr = x.cpu().float().reshape(n_img, H, W, C).permute(0, 3, 1, 2).requires_grad_(True)
    yr, ir = F.max_pool2d(xr, 2, 2, return_indices=True)
    assert torch.equal(y.cpu().float(), yr.permute(0, 2, 3, 1).reshape(-1, C))
    ir = ir.permute(0, 2, 3, 1)                                              # flat index h * W + w of the plane
    hh, ww = ir // W, ir % W
    pos = ((hh % 2) * 2 + (ww % 2)).reshape(-1, C).to(torch.uint8)
    assert torch.equal(idx.cpu(), pos)
    dy = torch.randn(n_img * (H // 2) * (W // 2), C, device=dev).bfloat16()
    dx = ops.maxpool2x2_bwd(dy, idx, n_img, H, W, C)
    yr.backward(dy.cpu().float().reshape(n_img, H // 2, W // 2, C).permute(0, 3, 1, 2))
    assert torch.equal(dx.cpu().float(), xr.grad.permute(0, 2, 3, 1).reshape(-1, C))


@pytest.mark.parametrize("n_img,H,W,ci,co", [(4, 8, 32, 128, 128), (4, 4, 16, 128, 192), (6, 2, 8, 288, 384), (8, 1, 4, 256, 256)])
def test_conv3x3_as_gemm_forward_dgrad_wgrad(dev, n_img, H, W, ci, co):
    """conv3x3_block's convolution (modeling_pretrain_moco_mim_ori.py:239-248) on the GEMM kernels: forward = im2col @ W^T + bias, data gradient =
    im2col(dy) @ flip(W)^T, weight gradient = dy^T @ im2col -- against F.conv2d and autograd in fp32 on the same bf16 values (incl. the
    288-channel map of ViT-Tiny whose 2592 columns are padded to 2624)."""
    from dig_amd import ops
    cpu_limit(dev, n_img * H * W * 9 * ci * co * 3.0)
    rows = n_img * H * W
    x = torch.randn(rows, ci, device=dev).bfloat16()
    w = (torch.randn(co, ci * 9, device=dev) / (3 * ci ** 0.5)).bfloat16()
    b = torch.randn(co, device=dev)
    col = ops.im2col3x3(x, n_img, H, W, ci)
    y = ops.gemm(col, w, rows, co, col.shape[1], bias=b)
    xr = x.cpu().float().reshape(n_img, H, W, ci).permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.cpu().float().reshape(co, ci, 3, 3).requires_grad_(True)
    yr = F.conv2d(xr, wr, b.cpu(), padding=1)
    assert rel(y.cpu(), yr.permute(0, 2, 3, 1).reshape(rows, co)) < 1e-2
    dy = torch.randn(rows, co, device=dev).bfloat16()
    yr.backward(dy.cpu().float().reshape(n_img, H, W, co).permute(0, 3, 1, 2))
    cold = ops.im2col3x3(dy, n_img, H, W, co)
    dx = ops.gemm(cold, ops.conv3x3_weight_flip(w, co, ci), rows, ci, cold.shape[1])
    assert rel(dx.cpu(), xr.grad.permute(0, 2, 3, 1).reshape(rows, ci)) < 1e-2
    dw = torch.zeros(co, ci * 9, device=dev)
    ops.wgrad(dy, col, dw, co, ci * 9, rows)
    assert rel(dw.cpu(), wr.grad.reshape(co, ci * 9)) < 1e-2


def test_conv_patchnet_module_vs_oracle(dev):
    """dig_amd/convpatchnet.py (ConvPatchNet, --patchnet_name conv) in isolation: output, the gradient w.r.t. the image tokens and every
    parameter gradient against fp32 autograd through oracle.conv_patch_extractor (pinned to the reference by tests/golden/tiny_w1_conv.npz) on
    the same bf16 inputs and weights.  The map is piecewise linear with a kink per ReLU sign and per max-pool arg-max: bf16 rounding flips
    some, each flip moves its own gradient elements by their full size -- the yardstick is the oracle itself under CPU bf16 autocast (the
    device may be twice as far from fp32 as that run is, + 3 %).  Running statistics and batch counters: against the oracle's buffers."""
    import dataclasses
    import dig_oracle as O
    from gpu_util import build_model
    from dig_amd import engine_core, convpatchnet, ops
    cfg = dataclasses.replace(O.DiGConfig(**O.TINY), patchnet="conv", num_windows=5)
    n_img, N, D = 48, 256, cfg.embed_dim
    P, S = O.det_state(cfg, 61)
    model = build_model(cfg, P, S, device=str(dev))
    torch.manual_seed(5)
    feat = torch.randn(n_img * N, D).bfloat16().to(dev)
    dout = (torch.randn(n_img, D) * 0.1).bfloat16().to(dev)
    ops.cast_f32_to_bf16(model._flat["online"], model.shadow("online"))
    model.flat_grads.zero_()
    if dev.type == "cuda":
        step = engine_core._Step(model)
    else:
        class step:                                                      # the CPU build: one "stream"
            m, comm = model, engine_core.LOCAL

            @staticmethod
            def _on_side(dev_, fn, *t):
                fn()
    step._bn_touched = []
    out, saved = convpatchnet.forward(step, feat, "patch_extractor", "online", n_img, True)
    dfeat = convpatchnet.backward(step, dout, "patch_extractor", saved, n_img)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    assert out.shape == (n_img, D) and dfeat.shape == (n_img * N, D) and len(step._bn_touched) == 6

    def run(autocast):
        # the weights as the kernels read them (bf16 shadow values), in fp32
        Pf = {k: (v.bfloat16().float() if v.dim() >= 2 else v.clone()).requires_grad_(True) for k, v in P.items() if k.startswith("patch_extractor.")}
        Sf = {k: v.clone() for k, v in S.items()}
        xf = feat.float().cpu().view(n_img, N, D).requires_grad_(True)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            y = O.conv_patch_extractor(xf, Pf, Sf, "patch_extractor.", cfg, O.LocalComm())
        y.float().backward(dout.float().cpu().view(n_img, 1, D))
        return y.detach().float().reshape(n_img, D), xf.grad.reshape(n_img * N, D), {k: v.grad for k, v in Pf.items()}, Sf
    y32, dx32, g32, S32 = run(False)
    y16, dx16, g16, _ = run(True)

    def far(a, b):
        return ((a.detach().float().cpu().reshape(-1) - b.float().reshape(-1)).norm() / b.float().norm()).item()
    checks = [("out", out, y32, y16), ("d tokens", dfeat, dx32, dx16)]
    for n_, sp in model.specs.items():
        if n_.startswith("patch_extractor.") and not O.bn_cancelled_bias(n_, cfg):
            checks.append((n_, model.flat_grads[sp.offset:sp.offset + sp.numel], g32[n_], g16[n_]))
    bad = [(n_, far(a, r), far(y, r)) for n_, a, r, y in checks if far(a, r) > 2 * far(y, r) + 3e-2]
    assert not bad, bad
    for n_, sp in model.specs.items():                                   # biases in front of a BatchNorm: zero gradient, round-off on every side
        if n_.startswith("patch_extractor.") and O.bn_cancelled_bias(n_, cfg):
            w = n_[:-4] + "weight"
            assert model._g32[n_].norm().item() <= 2e-2 * model._g32[w].norm().item() + 1e-6, n_
    sd = model.state_dict()
    for k in S32:
        if k.startswith("patch_extractor."):
            if k.endswith("num_batches_tracked"):
                continue                                                  # (counted once per step by the engine: dig_amd/engine_core.py)
            assert far(sd[k], S32[k]) < 2e-2, k


def test_mask_index_gather_target_are_bit_exact(dev):
    """Integer / byte work: bit-exact against the reference's boolean indexing (engine_for_pretraining_moco.py:96-107)."""
    from dig_amd import ops
    import dig_oracle as O
    B = 7
    cfg = O.DiGConfig()
    im, _, mk = O.synthetic_batch(B, cfg, 21)
    mask_b, labels = O.mim_targets(im, mk, cfg)
    m8 = mask_b[:, 0].to(torch.uint8).to(dev)
    idx, cnt = ops.mask_to_index(m8, 179)
    ref_idx = torch.nonzero(mask_b[:, 0].reshape(-1)).squeeze(1).to(torch.int32).reshape(B, 179)
    assert torch.equal(idx.cpu(), ref_idx) and bool((cnt.cpu() == 179).all())
    tgt = ops.mim_target(im.to(dev), idx, B * 179, 8, 32)
    assert torch.equal(tgt.cpu().reshape(B, 179, 48), labels[0])                          # bit-exact floats too (x*0.5+0.5)
    # normlize_target=True (engine_for_pretraining_moco.py:88-93): per-patch, per-channel standardisation with the unbiased variance
    _, nlabels = O.mim_targets(im, mk, cfg, normlize_target=True)
    ntgt = ops.mim_target(im.to(dev), idx, B * 179, 8, 32, normalize=True)
    assert (ntgt.cpu().reshape(B, 179, 48) - nlabels[0]).abs().max().item() < 2e-5 * nlabels[0].abs().max().item()
    src = torch.randn(B * 256, 384, device=dev).bfloat16()
    M = B * 179
    Mp = (M + 63) // 64 * 64
    dst = ops.gather_rows(src, idx, M, Mp)
    assert torch.equal(dst[:M], src[idx.reshape(-1).long()]) and bool((dst[M:] == 0).all())
    acc = torch.randn(B * 256, 384, device=dev).bfloat16(); acc0 = acc.clone()
    ops.scatter_rows_add(dst, idx, acc, M)
    refacc = acc0.float(); refacc[idx.reshape(-1).long()] += dst[:M].float()
    assert rel(acc, refacc) < 5e-3
    # ragged / empty masks: counts are reported, nothing is written past max_per_sample
    z = torch.zeros(2, 256, dtype=torch.uint8, device=dev); z[1, ::2] = 1
    idx2, cnt2 = ops.mask_to_index(z, 179)
    assert cnt2.cpu().tolist() == [0, 128] and idx2[1, :128].cpu().tolist() == list(range(256, 512, 2))


def test_mse(dev):
    from dig_amd import ops
    M, Mp = 895, 896
    pred = torch.randn(Mp, 64, device=dev); tgt = torch.rand(M, 48, device=dev)
    loss = torch.zeros(1, device=dev); dpred = torch.empty(Mp, 64, device=dev, dtype=torch.bfloat16)
    ops.mse_fwd_bwd(pred, 64, tgt, M, 48, 0.7, loss, dpred, 64)
    pf = pred[:M, :48].clone().requires_grad_(True)
    rl = F.mse_loss(pf, tgt); (rl * 0.7).backward()
    assert abs(loss.item() - rl.item()) < 1e-5 * rl.item() and rel(dpred[:M, :48], pf.grad) < 1e-2
    assert float(dpred[:M, 48:].abs().max()) == 0.0                     # pad columns of the written rows are zeroed


@pytest.mark.parametrize("mk,off", [(512, 0), (2048, 512), (4096, 3584), (520, 8)])
def test_infonce(dev, mk, off):
    """1, 4 and 8 ranks' worth of gathered keys (the long logit-gradient reduction runs as deterministic fp32 slabs)."""
    from dig_amd import ops
    nq, C, T = 512, 256, 0.2
    q, k = torch.randn(nq, C, device=dev), torch.randn(mk, C, device=dev)
    qn, qi = ops.l2norm_fwd(q); kn, _ = ops.l2norm_fwd(k)
    logits = torch.empty(nq, mk, device=dev)
    ops.sgemm(qn, kn, logits, nq, mk, C, False, 1.0 / T)
    qf = q.clone().requires_grad_(True)
    rlog = F.normalize(qf, dim=1) @ F.normalize(k, dim=1).t() / T
    labels = torch.arange(nq, device=dev) + off
    rloss = F.cross_entropy(rlog, labels) * 2 * T
    rloss.backward()
    assert rel(logits, rlog) < 1e-5
    out3 = torch.zeros(3, device=dev)
    ops.ce_rows(logits, off, 2 * T / nq, out3)
    dqn = torch.empty(nq, C, device=dev)
    ops.sgemm(logits, kn, dqn, nq, C, mk, True, 1.0 / T)
    dq = ops.l2norm_bwd(dqn, qn, qi)
    hit = rlog.topk(5, 1)[1].eq(labels[:, None])
    assert abs(out3[0].item() * 2 * T / nq - rloss.item()) < 1e-5 * rloss.item() and rel(dq, qf.grad) < 1e-4
    assert out3[1].item() == hit[:, :1].sum().item() and out3[2].item() == hit.sum().item()


def test_adamw_ema_sumsq_colsum(dev):
    from dig_amd import ops
    n = 1 << 20
    p, g = torch.randn(n, device=dev), torch.randn(n, device=dev) * 1e-2
    m, v = torch.randn(n, device=dev) * 1e-3, torch.rand(n, device=dev) * 1e-5
    p0, m0, v0 = p.clone(), m.clone(), v.clone()
    sh = torch.empty(n, device=dev, dtype=torch.bfloat16)
    flags = torch.zeros(n // 256, dtype=torch.uint8, device=dev); flags[n // 512:] = 1
    ops.adamw_step(p, g, m, v, sh, flags, 1e-3, 0.1, 2e-3, 0.0, 0.9, 0.999, 1e-8, 3, 0.5)
    ge = g * 0.5
    rp = p0.clone(); rp[: n // 2] *= (1 - 1e-3 * 0.1)
    rm, rv = m0 * 0.9 + ge * 0.1, v0 * 0.999 + ge * ge * 0.001
    den = rv.sqrt() / math.sqrt(1 - 0.999 ** 3) + 1e-8
    lr = torch.full((n,), 1e-3, device=dev); lr[n // 2:] = 2e-3
    rp -= (lr / (1 - 0.9 ** 3)) * rm / den
    assert float((p - rp).abs().max()) < 2e-6 and rel(m, rm) < 1e-6 and rel(v, rv) < 1e-6 and rel(sh, rp) < 5e-3
    pm = torch.randn(n, device=dev); pm0 = pm.clone()
    ops.ema_update(pm, p, sh, n, 0.99)
    assert float((pm - (pm0 * 0.99 + p * (1 - 0.99))).abs().max()) < 1e-6
    ws, o1 = torch.empty(1024, device=dev), torch.empty(1, device=dev)
    ops.sumsq(g, ws, o1)
    assert abs(o1.item() - (g.double() ** 2).sum().item()) < 1e-5 * (g.double() ** 2).sum().item()
    xx = torch.randn(3000, 48, device=dev).bfloat16(); cs = torch.ones(48, device=dev)
    ops.colsum(xx, cs)
    assert rel(cs, 1 + xx.float().sum(0)) < 1e-5


def test_adamw_step_tr_equals_plain_step_plus_transposes(dev):
    """dig_adamw_step_tr: the plain launch's p / m / v / shadow bit for bit (flat granules and the 64 x 64 tiles of the listed weights),
    W^T of every listed weight == the transposed shadow, group 2 (a parameter without a gradient) untouched, the non-finite gate a no-op."""
    import struct
    from dig_amd import ops
    mats = [(128, 192), (192, 64), (64, 64)]                      # (rows, cols) of the listed weights, with other parameters between them
    layout, off = [], 0
    for k, (r, c) in enumerate(mats):
        off += 256 * (k + 1)                                      # a few 1-D parameters in front
        layout.append((off, r, c))
        off += r * c
    n = off + 512
    torch.manual_seed(5)
    p0, g = torch.randn(n, device=dev), torch.randn(n, device=dev) * 1e-2
    m0, v0 = torch.randn(n, device=dev) * 1e-3, torch.rand(n, device=dev) * 1e-5
    flags = torch.zeros(n // 256, dtype=torch.uint8, device=dev)
    flags[0] = 1
    flags[-1] = 2                                                 # the last granule: no gradient, untouched
    hyp = (1e-3, 0.1, 2e-3, 0.0, 0.9, 0.999, 1e-8, 3)
    p1, m1, v1 = p0.clone(), m0.clone(), v0.clone()
    sh1 = torch.zeros(n, device=dev, dtype=torch.bfloat16)
    ops.adamw_step(p1, g, m1, v1, sh1, flags, *hyp, 0.5)
    assert torch.equal(p1[-256:], p0[-256:]) and torch.equal(m1[-256:], m0[-256:]) and torch.equal(sh1[-256:], p0[-256:].bfloat16())
    assert not torch.equal(p1[:256], p0[:256])
    recs, dst, t0, fl = [], 0, 0, flags.clone()
    for o, r, c in layout:
        recs.append(struct.pack("<qqiiii", o, dst, r, c, t0, 0))
        fl[o // 256:(o + r * c) // 256] |= 0x80
        dst += r * c
        t0 += (r // 64) * (c // 64)
    table = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(dev)
    p2, m2, v2 = p0.clone(), m0.clone(), v0.clone()
    sh2 = torch.zeros(n, device=dev, dtype=torch.bfloat16)
    tr = torch.zeros(dst, device=dev, dtype=torch.bfloat16)
    ops.adamw_step_tr(p2, g, m2, v2, sh2, fl, *hyp, table, len(layout), t0, tr, 0.5)
    assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2) and torch.equal(sh1, sh2)
    d = 0
    for o, r, c in layout:
        assert torch.equal(tr[d:d + r * c].view(c, r), sh1[o:o + r * c].view(r, c).t())
        d += r * c
    gate = torch.full((1,), float("inf"), device=dev)
    p3, tr3 = p0.clone(), torch.zeros_like(tr)
    ops.adamw_step_tr(p3, g, m0.clone(), v0.clone(), sh2, fl, *hyp, table, len(layout), t0, tr3, 0.5, gate)
    assert torch.equal(p3, p0) and not bool(tr3.any())


def test_grad_reduce_batch_equals_single_launches(dev):
    """ops.GradReduceBatch (dig_reduce_partials_multi + dig_colsum_partials_multi: an encoder block's eleven reduction launches as two)
    against the single-launch forms on the same inputs."""
    from dig_amd import ops
    rows, D, F4 = 4096, 128, 512
    dy1, x1 = torch.randn(rows, F4, device=dev).bfloat16(), torch.randn(rows, D, device=dev).bfloat16()
    dy2, x2 = torch.randn(rows, D, device=dev).bfloat16(), torch.randn(rows, F4, device=dev).bfloat16()
    parts = torch.randn(rows // 64, F4, device=dev)
    qs = torch.randn(16, D, device=dev)
    xx, gy, dres = (torch.randn(rows, D, device=dev).bfloat16() for _ in range(3))
    gam, bet = torch.randn(D, device=dev), torch.randn(D, device=dev)
    _, mean, rstd = ops.layernorm_fwd(xx, gam, bet, 1e-6)

    def run(batched):
        dw1, dw2 = torch.randn(F4, D, device=dev), torch.randn(D, F4, device=dev)
        torch.manual_seed(5)
        b1, bq, dg, db, dc = (torch.randn(n, device="cpu").to(dev) for n in (F4, D, D, D, D))
        dx, fin, ws = ops.layernorm_bwd(gy, xx, gam, bet, mean, rstd, dres, dg, db, dres_colsum=dc, defer=True)
        if batched:
            red = ops.GradReduceBatch()
            red.wgrad(dy1, x1, dw1); red.wgrad(dy2, x2, dw2)
            red.colsum_partials(parts, b1); red.colsum_partials(qs, bq)
            red.layernorm_finalize(ws, rows, D, dg, db, dc)
            assert len(red.slabs) == 2 and len(red.vecs) == 5
            red.flush()
            assert not red.slabs and not red.vecs
        else:
            ops.linear_wgrad(dy1, x1, dw1); ops.linear_wgrad(dy2, x2, dw2)
            ops.colsum_partials(parts, b1); ops.colsum_partials(qs, bq)
            fin()
        return dw1, dw2, b1, bq, dg, db, dc
    torch.manual_seed(9)
    a = run(False)
    torch.manual_seed(9)
    b = run(True)
    for i, (u, v) in enumerate(zip(a, b)):
        if i < 4:                                   # slab sums and column sums: the same order per element -> the same bits
            assert torch.equal(u, v), i
        else:                                       # LayerNorm vectors: 128 x 2 instead of 64 x 4 row groups in the tree (both fixed orders)
            assert rel(u, v) < 1e-6, i


def test_colsum_partials_multi_many_segments(dev):
    """dig_colsum_partials_multi with more than one encoder block's worth of segments (the deferred reductions of the single-process backward: every
    block's bias and LayerNorm partial rows in one launch; widths that are multiples of 32 take the full-line kernel, one width that is not sends the
    whole launch to the 8-column kernel): out += column sums, against fp64 torch, += semantics, interleaved strides, fixed order."""
    import ctypes
    from dig_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    for odd in (False, True):
        segs, want, outs, keep = [], [], [], []
        for k in range(40):
            C = (384, 1536, 64, 512)[k % 4] if not (odd and k == 7) else 40
            n_parts = (1, 5, 33, 512, 2048)[k % 5]
            vecs = 3 if k % 3 == 0 else 1                                  # (a LayerNorm workspace interleaves three vectors: stride 3 C)
            ws = torch.randn(n_parts, vecs, C, generator=g).to(dev)
            keep.append(ws)
            for v in range(vecs):
                out = torch.randn(C, generator=g).to(dev)
                want.append(out.double() + ws[:, v].double().sum(0))
                outs.append(out)
                segs.append(ops._ColsumSeg(ws.data_ptr() + 4 * v * C, out.data_ptr(), vecs * C, n_parts, C))
        assert 12 < len(segs) <= ops.COLSUM_MAX_SEGS
        arr = (ops._ColsumSeg * len(segs))(*segs)
        ops.L.call("dig_colsum_partials_multi", arr, len(segs), ops.L.stream())
        for o, w in zip(outs, want):
            assert float((o.double() - w).abs().max() / w.abs().max()) < (2e-6 if dev.type != "cpu" else 2e-5)     # (the CPU build sums each column in one fp32 chain)
    with pytest.raises(RuntimeError):                                      # more segments than one launch takes
        big = (ops._ColsumSeg * (ops.COLSUM_MAX_SEGS + 1))(*([segs[0]] * (ops.COLSUM_MAX_SEGS + 1)))
        ops.L.call("dig_colsum_partials_multi", big, ops.COLSUM_MAX_SEGS + 1, ops.L.stream())


@pytest.mark.parametrize("R,D,Fh", [(64, 384, 512), (1024, 384, 1536), (65536, 384, 1536), (32768, 384, 1536), (256, 256, 1024), (8192, 512, 2048)])
def test_wgrad_group(dev, R, D, Fh):
    """The grouped weight-gradient kernel (csrc/wgrad.hip) on the four Linear layers of a transformer block, in every launch grouping:
    each gradient against fp32 torch (fp32 accumulation of bf16 products: tight), the fold of the NEXT launch / the flush as the only
    place where gradients change, += semantics, transposed output (fc2), and bit-reproducibility (split-ordered fold, no atomics)."""
    from dig_amd import ops
    cpu_limit(dev, 2.0 * R * D * (2 * Fh + 4 * D), 5e9)
    g = torch.Generator(device="cpu").manual_seed(R + D)
    mk = lambda *sh: (torch.randn(*sh, generator=g) * 0.5).bfloat16().to(dev)
    dact, act, ln2, dx = mk(R, Fh), mk(R, Fh), mk(R, D), mk(R, D)
    dqkv, ln1, ctx, dxm = mk(R, 3 * D), mk(R, D), mk(R, D), mk(R, D)
    layers = [(dx, act, (D, Fh)), (dact, ln2, (Fh, D)), (dxm, ctx, (D, D)), (dqkv, ln1, (3 * D, D))]      # (dy, x, dW shape)
    refs = [dy.float().t() @ x.float() for dy, x, _ in layers]
    outs = {}
    for grouping in ((4,), (2, 2), (1, 1, 1, 1)):
        dws = [torch.full(sh, 0.25, device=dev) for _, _, sh in layers]
        grp = ops.WgradGroup(dev)
        k = 0
        for n in grouping:
            for li in range(k, k + n):
                assert grp.add(layers[li][0], layers[li][1], dws[li])
            grp.launch()
            k += n
        if grouping == (4,):
            assert all(torch.equal(dw, torch.full_like(dw, 0.25)) for dw in dws)       # nothing is final before the fold
        grp.flush()
        for dw, ref in zip(dws, refs):
            assert rel(dw - 0.25, ref) < 2e-5
        outs[grouping] = dws
    # second pass of one grouping: same bits
    dws = [torch.full(sh, 0.25, device=dev) for _, _, sh in layers]
    grp = ops.WgradGroup(dev)
    for (dy, x, _), dw in zip(layers, dws):
        assert grp.add(dy, x, dw)
    grp.launch(); grp.flush()
    assert all(torch.equal(a, b) for a, b in zip(dws, outs[(4,)]))
    # a shape the kernel does not take is refused, not mangled
    assert not ops.WgradGroup(dev).add(mk(R, 100), ln2, torch.zeros(100, D, device=dev))


@pytest.mark.parametrize("R,Fh", [(128, 128), (4096, 1536), (333, 256), (65536, 1536), (1000, 2048)])
def test_mlp_chain_fwd_with_layernorms(dev, R, Fh):
    """dig_mlp_chain_fwd_ln: norm2 on the way in, the next block's norm1 on the way out (modeling_finetune.py:151,156-158), against fp32 torch
    and against the three launches it replaces (layernorm_fwd + mlp_chain_fwd + layernorm_fwd); both forms (with / without what the backward
    keeps), with and without the trailing LayerNorm, ragged row counts."""
    from dig_amd import ops
    D, eps = 384, 1e-6
    cpu_limit(dev, 4.0 * R * D * Fh, limit=3e9)
    g = torch.Generator(device="cpu").manual_seed(7 * R + Fh)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    x = (rn(R, D) * 1.7 + 0.3 * rn(R, 1)).bfloat16()                           # rows with different means
    w1 = (rn(Fh, D) * 0.06).bfloat16(); b1 = rn(Fh) * 0.5
    w2 = (rn(D, Fh) * 0.04).bfloat16(); b2 = rn(D) * 0.5
    g1, be1, g2, be2 = 1 + 0.2 * rn(D), 0.3 * rn(D), 1 + 0.2 * rn(D), 0.3 * rn(D)
    ln_ref = F.layer_norm(x.float(), (D,), g1, be1, eps)
    pre_ref = ln_ref.bfloat16().float() @ w1.float().t() + b1
    out_ref = F.gelu(pre_ref).bfloat16().float() @ w2.float().t() + b2 + x.float()
    nln_ref = F.layer_norm(out_ref.bfloat16().float(), (D,), g2, be2, eps)
    r = ops.mlp_chain_fwd_ln(x, g1, be1, eps, w1, b1, w2, b2, g2, be2, save=True)
    assert rel(r["ln"], ln_ref) < 6e-3 and rel(r["pre"], pre_ref) < 1e-2 and rel(r["act"], F.gelu(pre_ref)) < 1e-2
    assert rel(r["out"], out_ref) < 1e-2 and rel(r["nln"], nln_ref) < 1.5e-2
    mu = x.float().mean(1); var = x.float().var(1, unbiased=False)
    assert (r["ln_mean"] - mu).abs().max().item() < 1e-4 * (1 + mu.abs().max().item()) and rel(r["ln_rstd"], (var + eps).rsqrt()) < 1e-4
    o = r["out"].float()
    assert (r["nln_mean"] - o.mean(1)).abs().max().item() < 2e-4 * (1 + o.mean(1).abs().max().item())
    assert rel(r["nln_rstd"], (o.var(1, unbiased=False) + eps).rsqrt()) < 1e-4
    # the three launches it replaces
    ln3, mu3, rs3 = ops.layernorm_fwd(x, g1, be1, eps)
    out3 = ops.mlp_chain_fwd(ln3, w1, b1, w2, b2, x)
    nln3, _, _ = ops.layernorm_fwd(out3, g2, be2, eps)
    assert rel(r["ln"], ln3) < 3e-3 and rel(r["out"], out3) < 4e-3 and rel(r["nln"], nln3) < 6e-3
    assert rel(r["ln_rstd"], rs3) < 1e-5
    # the gradient-free form: same arithmetic, nothing kept; and without the trailing LayerNorm
    q = ops.mlp_chain_fwd_ln(x, g1, be1, eps, w1, b1, w2, b2, g2, be2)
    assert torch.equal(q["out"], r["out"]) and torch.equal(q["nln"], r["nln"]) and q["ln"] is None and q["pre"] is None
    q = ops.mlp_chain_fwd_ln(x, g1, be1, eps, w1, b1, w2, b2, save=True)
    assert torch.equal(q["out"], r["out"]) and torch.equal(q["ln"], r["ln"]) and torch.equal(q["pre"], r["pre"]) and q["nln"] is None
    q2 = ops.mlp_chain_fwd_ln(x, g1, be1, eps, w1, b1, w2, b2, g2, be2, save=True)
    assert all(torch.equal(q2[k], r[k]) for k in r if r[k] is not None)               # bit-reproducible
    # rows normalised elsewhere (the row-panel GEMM's epilogue): x = LN rows, resid = the raw rows
    q3 = ops.mlp_chain_fwd_ln(ln3, None, None, eps, w1, b1, w2, b2, g2, be2, save=True, resid=x)
    assert q3["ln"] is None and torch.equal(q3["out"], out3) and rel(q3["nln"], nln3) < 1e-3


@pytest.mark.parametrize("R,Fh", [(128, 128), (256, 1536), (4096, 1536), (333, 256), (65536, 1536), (1000, 2048)])
def test_mlp_chain_fwd_bwd(dev, R, Fh):
    """dig_mlp_chain_fwd / _bwd (fc1 -> GELU -> fc2 and its data gradient in one launch each) against fp32 torch and against the
    two-GEMM path they replace; ragged row counts, both forward forms (with and without the saved pre-activation / GELU output)."""
    from dig_amd import ops
    D = 384
    cpu_limit(dev, 4.0 * R * D * Fh, limit=3e9)
    assert ops.mlp_chain_supported(D, Fh) and not ops.mlp_chain_supported(512, 2048) and not ops.mlp_chain_supported(D, 192)
    g = torch.Generator(device="cpu").manual_seed(R + Fh)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    x = rn(R, D).bfloat16()
    w1 = (rn(Fh, D) * 0.06).bfloat16(); b1 = rn(Fh) * 0.5
    w2 = (rn(D, Fh) * 0.04).bfloat16(); b2 = rn(D) * 0.5
    res = rn(R, D).bfloat16()
    pre_ref = x.float() @ w1.float().t() + b1
    act_ref = F.gelu(pre_ref)
    out_ref = act_ref.bfloat16().float() @ w2.float().t() + b2 + res.float()
    out = ops.mlp_chain_fwd(x, w1, b1, w2, b2, res)
    assert rel(out, out_ref) < 1e-2
    out2, pre, act = ops.mlp_chain_fwd(x, w1, b1, w2, b2, res, save=True)
    assert torch.equal(out, out2)                                     # the side outputs do not change the arithmetic
    assert rel(pre, pre_ref) < 1e-2 and rel(act, act_ref) < 1e-2
    assert rel(ops.mlp_chain_fwd(x, w1, None, w2, None, None), F.gelu(x.float() @ w1.float().t()).bfloat16().float() @ w2.float().t()) < 1e-2
    # against the two launches it replaces: same operand rounding, same fp32 accumulation -> agreement far inside the bf16 yardstick
    pre2 = torch.empty_like(pre)
    act2 = ops.linear_fwd(x, w1, bias=b1, act=1, pre=pre2)
    out3 = ops.linear_fwd(act2, w2, bias=b2, resid=res)
    assert rel(pre, pre2) < 2e-3 and rel(act, act2) < 2e-3 and rel(out, out3) < 4e-3
    # ---- backward: dpre = (dy w2) * gelu'(pre), dx = dpre w1, fc1 bias gradient = column sums of dpre
    dy = rn(R, D).bfloat16()
    w2t, w1t = ops.transpose_bf16(w2), ops.transpose_bf16(w1)
    assert torch.equal(w2t, w2.t().contiguous()) and torch.equal(w1t, w1.t().contiguous())
    dx, dpre, parts = ops.mlp_chain_bwd(dy, w2t, pre, w1t)
    pp = pre.float().requires_grad_(True)
    F.gelu(pp).backward(dy.float() @ w2.float())
    assert rel(dpre, pp.grad) < 1e-2
    assert rel(dx, dpre.float() @ w1.float()) < 1e-2 and rel(dx, pp.grad @ w1.float()) < 1.5e-2
    db = torch.randn(Fh, generator=g).to(dev); db0 = db.clone()
    ops.colsum_partials(parts, db)
    assert rel(db - db0, dpre.float().sum(0)) < 1e-4 and rel(db - db0, pp.grad.sum(0)) < 3e-3
    # the polynomial GELU' (common.h::dgelu_f, |abs err| <= 2.8e-4, systematic in the tails) in aggregate: the fc1 weight gradient built
    # from the device's d(pre-activation) against the one built from the exact erf-GELU gradient, on N(0, ~1.2) pre-activations
    if R <= 4096:
        assert rel(dpre.float().t() @ x.float(), pp.grad.t() @ x.float()) < 3e-3
    dact, bparts = ops.linear_dgrad(dy, w2, gelu_pre=pre, colsum=True)
    assert rel(dpre, dact) < 2e-3 and rel(dx, ops.linear_dgrad(dact, w1)) < 4e-3
    dx2, dpre2, parts2 = ops.mlp_chain_bwd(dy, w2t, pre, w1t)
    assert torch.equal(dx, dx2) and torch.equal(dpre, dpre2) and torch.equal(parts, parts2)     # bit-reproducible


@pytest.mark.parametrize("R,Fh,p_el,p_path,save", [(512, 256, 0.1, 0.0, True), (768, 1536, 0.1, 0.2, True), (333, 128, 0.0, 0.3, False),
                                                   (4096, 1536, 0.25, 0.1, False)])
def test_mlp_chain_fwd_ln_dropout(dev, R, Fh, p_el, p_path, save):
    """dig_mlp_chain_fwd_ln_dropout (the fine-tune encoder's MLP half: Mlp.drop behind fc2 + drop_path of the branch, modeling_finetune.py:59,158)
    against the launches it replaces -- LayerNorm, fc1 + GELU, fc2 with the GEMM dropout epilogue + residual, the next LayerNorm: the same keyed
    mask (dropped elements return the residual value exactly, in the same places), kept values to the fused-vs-unfused yardstick."""
    from dig_amd import ops, dropout as DR
    D = 384
    cpu_limit(dev, 4.0 * R * D * Fh, limit=3e9)
    g = torch.Generator(device="cpu").manual_seed(R + Fh)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    x = (rn(R, D) * 1.2).bfloat16()
    gam, bet, ngam, nbet = 1.0 + 0.2 * rn(D), 0.1 * rn(D), 1.0 + 0.2 * rn(D), 0.1 * rn(D)
    w1 = (rn(Fh, D) * 0.06).bfloat16(); b1 = rn(Fh) * 0.5
    w2 = (rn(D, Fh) * 0.04).bfloat16(); b2 = rn(D) * 0.5
    spec = DR.DropPlan(1234, 5).spec(DR.enc_site(3, 3), p_el, DR.enc_site(3, 4), p_path, 64)
    r = ops.mlp_chain_fwd_ln(x, gam, bet, 1e-6, w1, b1, w2, b2, nln_g=ngam, nln_b=nbet, save=save, drop=spec)
    ln2, mu, rs = ops.layernorm_fwd(x, gam, bet, 1e-6)
    pre = torch.empty((R, Fh), device=dev, dtype=torch.bfloat16)
    act = ops.linear_fwd(ln2, w1, bias=b1, act=1, pre=pre)
    out = ops.linear_fwd(act, w2, bias=b2, resid=x, drop=spec)
    nln, nmu, nrs = ops.layernorm_fwd(out, ngam, nbet, 1e-6)
    assert rel(r["out"], out) < 4e-3 and rel(r["nln"], nln) < 1e-2
    same_a, same_b = r["out"] == x, out == x                           # dropped elements (and drop-path rows): the residual value itself
    assert float((same_a != same_b).float().mean()) < 1e-4
    if p_path == 0.0:                                                  # (drop-path acts on R / 64 samples: too few for a rate at these sizes)
        assert abs(float(same_b.float().mean()) - p_el) < 0.02
    else:
        rows_dropped = same_b.all(dim=1).view(-1)                      # a dropped sample: 64 whole rows return the residual
        assert bool(rows_dropped.any()) and not bool(rows_dropped.all())
        assert torch.equal(rows_dropped, same_a.all(dim=1).view(-1))
    if save:
        assert rel(r["pre"], pre) < 2e-3 and rel(r["act"], act) < 2e-3 and rel(r["ln"], ln2) < 1e-2
        assert rel(r["nln_mean"], nmu) < 1e-2
    plain = ops.mlp_chain_fwd_ln(x, gam, bet, 1e-6, w1, b1, w2, b2, nln_g=ngam, nln_b=nbet, save=save)
    none = ops.mlp_chain_fwd_ln(x, gam, bet, 1e-6, w1, b1, w2, b2, nln_g=ngam, nln_b=nbet, save=save, drop=DR.DropPlan(1, 1).spec(7, 0.0))
    assert torch.equal(plain["out"], none["out"])                      # no rate: the plain launch
    again = ops.mlp_chain_fwd_ln(x, gam, bet, 1e-6, w1, b1, w2, b2, nln_g=ngam, nln_b=nbet, save=save, drop=spec)
    assert torch.equal(again["out"], r["out"]) and torch.equal(again["nln"], r["nln"])


@pytest.mark.parametrize("R,Fh", [(128, 128), (333, 256), (4096, 1536), (1000, 2048), (65536, 1536)])
def test_mlp_chain_bwd_ln(dev, R, Fh):
    """dig_mlp_chain_bwd_ln (the MLP's data gradient AND norm2's backward in one launch) against the two launches it replaces
    (dig_mlp_chain_bwd -> dig_layernorm_bwd_partials / _finalize) and against fp32 torch autograd of the same half block; ragged rows."""
    from dig_amd import ops
    D = 384
    cpu_limit(dev, 4.0 * R * D * Fh, limit=3e9)
    g = torch.Generator(device="cpu").manual_seed(7 * R + Fh)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    x_mid = (rn(R, D) * 1.3 + 0.2).bfloat16()
    gam, bet = 1.0 + 0.3 * rn(D), 0.1 * rn(D)
    w1 = (rn(Fh, D) * 0.06).bfloat16(); b1 = rn(Fh) * 0.5
    w2 = (rn(D, Fh) * 0.04).bfloat16()
    dy = rn(R, D).bfloat16()
    ln2, mu, rs = ops.layernorm_fwd(x_mid, gam, bet, 1e-6)
    pre = torch.empty((R, Fh), device=dev, dtype=torch.bfloat16)
    ops.linear_fwd(ln2, w1, bias=b1, act=1, pre=pre)
    w2t, w1t = ops.transpose_bf16(w2), ops.transpose_bf16(w1)
    # the two launches
    dln2, dpre0, parts0 = ops.mlp_chain_bwd(dy, w2t, pre, w1t)
    dgam0, dbet0, dcol0 = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    dxm0 = ops.layernorm_bwd(dln2, x_mid, gam, bet, mu, rs, dy, dgam0, dbet0, dres_colsum=dcol0)
    # one launch
    dxm, dpre, parts, lnp = ops.mlp_chain_bwd_ln(dy, w2t, pre, w1t, x_mid, gam, mu, rs)
    assert torch.equal(dpre, dpre0) and torch.equal(parts, parts0)                 # the MLP half is the same arithmetic
    seed = rn(3, D)
    dgam, dbet, dcol = seed[0].clone(), seed[1].clone(), seed[2].clone()
    ops.layernorm_finalize_parts(lnp, dgam, dbet, dcol)
    dgam, dbet, dcol = dgam - seed[0], dbet - seed[1], dcol - seed[2]              # (finalize accumulates)
    assert rel(dxm, dxm0) < 4e-3
    assert rel(dgam, dgam0) < 2e-3 and rel(dbet, dbet0) < 1e-4 and rel(dcol, dcol0) < 1e-4
    assert rel(dcol, dy.float().sum(0)) < 1e-4 and rel(dbet, dln2.float().sum(0)) < 1e-4
    # fp32 autograd of y = x_mid + (dln2 flowing into LN): d x_mid = dy + LN'(dln2)
    xm = x_mid.float().requires_grad_(True)
    gp = gam.clone().requires_grad_(True)
    F.layer_norm(xm, (D,), gp, bet, 1e-6).backward(dln2.float())
    assert rel(dxm, xm.grad + dy.float()) < 1e-2 and rel(dgam, gp.grad) < 3e-3
    again = ops.mlp_chain_bwd_ln(dy, w2t, pre, w1t, x_mid, gam, mu, rs)
    assert all(torch.equal(a, b) for a, b in zip(again, (dxm, dpre, parts, lnp)))  # bit-reproducible
    # ... and with the attention projection's data gradient behind it: everything else unchanged, dctx = dx_mid Wproj against the GEMM launch
    # on the same bf16 rows (same products, another summation order) and against fp32 torch
    wp = (rn(D, D) * 0.05).bfloat16()
    out5 = ops.mlp_chain_bwd_ln(dy, w2t, pre, w1t, x_mid, gam, mu, rs, projt=ops.transpose_bf16(wp))
    assert all(torch.equal(a, b) for a, b in zip(out5[:4], (dxm, dpre, parts, lnp)))
    assert rel(out5[4], ops.linear_dgrad(dxm, wp)) < 2e-3 and rel(out5[4], dxm.float() @ wp.float()) < 5e-3
    assert torch.equal(out5[4], ops.mlp_chain_bwd_ln(dy, w2t, pre, w1t, x_mid, gam, mu, rs, projt=ops.transpose_bf16(wp))[4])


# ------------------------------------------------------------------------------------------------------------------------------------
# Exact-arithmetic parity: operands drawn from small integers / powers of two, so that every product and every fp32 partial sum is
# exact WHATEVER the summation order, and the only rounding left is the final fp32 -> bf16 round-to-nearest-even -- which torch's
# `.bfloat16()` of the exact fp32 result performs identically.  These tests have no tolerance to tune: `torch.equal` or fail.
def _ints(gen, shape, lo, hi, dev, scale=1.0):
    return (torch.randint(lo, hi + 1, shape, generator=gen).float() * scale).bfloat16().to(dev)


@pytest.mark.parametrize("bk", [0, 32, 64, 244, 264, 544, 564, 212, 221])
@pytest.mark.parametrize("I,J,R", [(512, 384, 256), (8192, 1152, 384), (1000, 520, 192)])
def test_gemm_exact_arithmetic(dev, bk, I, J, R):
    from dig_amd import ops
    cpu_limit(dev, I * J * R * (1 if bk == 0 else 1e9), 4e9)
    g = torch.Generator().manual_seed(I + J + R + bk)
    x, w = _ints(g, (I, R), -2, 2, dev), _ints(g, (J, R), -2, 2, dev, 0.5)
    bias = torch.randint(-8, 9, (J,), generator=g).float().to(dev)
    res = _ints(g, (I, J), -16, 16, dev)
    h = x.float() @ w.float().t()                                      # exact: |h| <= 2 * 1 * R, multiples of 0.5
    # forward epilogues: bias, power-of-two alpha on a column range, residual; bf16 and fp32 outputs; the pre-activation side output
    assert torch.equal(ops.gemm(x, w, I, J, R, bias=bias, resid=res, bk=bk), (h + bias + res.float()).bfloat16())
    ac = (J // 16) * 8
    ref = h + bias
    ref[:, :ac] *= 0.125
    assert torch.equal(ops.gemm(x, w, I, J, R, bias=bias, alpha=0.125, alpha_cols=ac, out_kind=ops.OUT_F32, bk=bk), ref)
    pre = torch.empty(I, J, device=dev, dtype=torch.bfloat16)
    ops.gemm(x, w, I, J, R, bias=bias, pre=pre, act=1, bk=bk)
    assert torch.equal(pre, (h + bias).bfloat16())
    # data gradient (transpose-read weight operand)
    if J % 64 == 0 and bk not in (212, 544, 564):
        dy = _ints(g, (I, J), -2, 2, dev)
        assert torch.equal(ops.gemm(dy, w, I, R, J, tb=True, bk=bk), (dy.float() @ w.float()).bfloat16())
    # weight gradient: split-R slabs + ordered sum, fp32 -- exact and therefore equal to torch bit for bit
    if bk in (0, 32, 64, 244, 264):
        dy = _ints(g, (I, J), -2, 2, dev)
        sp = ops.L.lib().dig_gemm_effective_splits(I, 4)
        ws = torch.empty(sp, J, R, device=dev)
        ops.gemm(dy, x, J, R, I, ta=True, tb=True, out=ws, out_kind=ops.OUT_F32_PARTIAL, splits=sp, ldc=R, bk=bk or 32)
        assert torch.equal(ws.sum(0), dy.float().t() @ x.float())


@pytest.mark.parametrize("wa", [1, 2])
@pytest.mark.parametrize("R,D,Fh", [(256, 384, 1536), (16384, 384, 1536), (4096, 512, 2048)])
def test_wgrad_group_exact_arithmetic(dev, wa, R, D, Fh):
    """Integer operands: every split count, tile form and fold order of the grouped weight-gradient kernel must give THE fp32 result."""
    from dig_amd import ops
    cpu_limit(dev, 2.0 * R * D * (2 * Fh + 4 * D), 5e9)
    g = torch.Generator().manual_seed(R + D + wa)
    dact, act, ln2, dx = _ints(g, (R, Fh), -2, 2, dev), _ints(g, (R, Fh), -2, 2, dev), _ints(g, (R, D), -2, 2, dev, 0.5), _ints(g, (R, D), -2, 2, dev, 0.5)
    dqkv, ln1, ctx, dxm = _ints(g, (R, 3 * D), -2, 2, dev), _ints(g, (R, D), -2, 2, dev, 0.5), _ints(g, (R, D), -2, 2, dev), _ints(g, (R, D), -2, 2, dev, 0.5)
    layers = [(dx, act, (D, Fh)), (dact, ln2, (Fh, D)), (dxm, ctx, (D, D)), (dqkv, ln1, (3 * D, D))]
    saved = ops.WGRAD_GROUP_WA, ops.WGRAD_GROUP_SLOTS
    ops.WGRAD_GROUP_WA, ops.WGRAD_GROUP_SLOTS = wa, 512 // wa
    try:
        dws = [torch.full(sh, 3.0, device=dev) for _, _, sh in layers]
        grp = ops.WgradGroup(dev)
        for (dy, x, _), dw in zip(layers, dws):
            assert grp.add(dy, x, dw)
        grp.launch(); grp.flush()
        for (dy, x, _), dw in zip(layers, dws):
            assert torch.equal(dw, 3.0 + dy.float().t() @ x.float())
    finally:
        ops.WGRAD_GROUP_WA, ops.WGRAD_GROUP_SLOTS = saved


@pytest.mark.parametrize("R", [128, 4096, 1000])
def test_mlp_chain_fwd_exact_arithmetic(dev, R):
    """Fused fc1 -> GELU -> fc2 with every pre-activation an integer in GELU's linear tail (x >= 8: gelu(x) rounds to x in bf16, in the
    kernel's polynomial and in torch alike): hidden values, both GEMMs, both biases and the residual are exact; the output is the
    round-to-nearest-even of the exact fp32 value."""
    from dig_amd import ops
    D, Fh = 384, 1536
    if not ops.mlp_chain_supported(D, Fh):
        pytest.skip("fused MLP kernel not built for this width")
    cpu_limit(dev, 4.0 * R * D * Fh, 4e9)
    g = torch.Generator().manual_seed(R)
    # x has 8 nonzeros of +-1 per row, w1 in {-1, 0, 1}: |x w1^T| <= 8; b1 = 24 -> pre-activation in [16, 32]
    x = torch.zeros(R, D)
    idx = torch.stack([torch.randperm(D, generator=g)[:8] for _ in range(R)])
    x.scatter_(1, idx, torch.randint(0, 2, (R, 8), generator=g).float() * 2 - 1)
    x = x.bfloat16().to(dev)
    w1 = _ints(g, (Fh, D), -1, 1, dev)
    b1 = torch.full((Fh,), 24.0, device=dev)
    # w2 sparse in {-1/8, 0, 1/8}: |hidden w2^T| <= 32 * 1536 / 8, partial sums multiples of 1/8: exact in fp32
    w2 = (_ints(g, (D, Fh), -1, 1, dev).float() * (torch.rand(D, Fh, generator=g) < 0.05).to(dev) * 0.125).bfloat16()
    b2 = torch.randint(-4, 5, (D,), generator=g).float().to(dev)
    res = _ints(g, (R, D), -8, 8, dev)
    pre_ref = x.float() @ w1.float().t() + b1
    assert pre_ref.min().item() >= 16 and pre_ref.max().item() <= 32
    ref = (pre_ref @ w2.float().t() + b2 + res.float()).bfloat16()
    assert torch.equal(ops.mlp_chain_fwd(x, w1, b1, w2, b2, res), ref)
    out, pre, act = ops.mlp_chain_fwd(x, w1, b1, w2, b2, res, save=True)
    assert torch.equal(out, ref) and torch.equal(pre, pre_ref.bfloat16()) and torch.equal(act, pre_ref.bfloat16())


def test_attention_exact_arithmetic(dev):
    """One-hot queries / keys: a score is 256 where query and key share their hot channel and 0 elsewhere, so the softmax is exactly
    uniform over the matching keys (exp(-256) underflows to 0 in fp32) and the context is the exact mean of their integer value rows."""
    from dig_amd import ops
    Bn, H = 2, 6
    D = H * 64
    g = torch.Generator().manual_seed(7)
    qh = torch.randint(0, 64, (Bn, H, 256), generator=g)                  # hot channel of every query
    kh = torch.arange(256).remainder(64).expand(Bn, H, 256).clone()      # key j is hot in channel j % 64: four keys per channel
    for b in range(Bn):
        for h in range(H):
            kh[b, h] = kh[b, h][torch.randperm(256, generator=g)]
    v = torch.randint(-8, 9, (Bn, H, 256, 64), generator=g).float() * 4.0  # multiples of 4: the mean of four rows is an integer
    q = torch.zeros(Bn, H, 256, 64).scatter_(3, qh.unsqueeze(-1), 16.0)
    k = torch.zeros(Bn, H, 256, 64).scatter_(3, kh.unsqueeze(-1), 16.0)
    qkv = torch.stack([q, k, v], 0).permute(1, 3, 0, 2, 4).reshape(Bn * 256, 3 * D).bfloat16().to(dev)      # [image, token, (q|k|v), head, 64]
    ctx, lse = ops.attn_fwd(qkv, Bn, H, D)
    match = (qh.unsqueeze(-1) == kh.unsqueeze(-2)).float()                # [Bn, H, query, key]
    ref = (match @ v) / match.sum(-1, keepdim=True)
    assert torch.equal(ctx.float().cpu().view(Bn, 256, H, 64).permute(0, 2, 1, 3), ref)
    assert (lse.cpu().view(Bn, H, 256) - (256.0 + math.log(4.0))).abs().max().item() < 1e-3


def _attn_block_reference(ln1, x, wq, bq, wp, bp, n_img, H, D, scale):
    """Attention.forward + the residual add (modeling_finetune.py:87-120, :156) in fp32 torch on the bf16 operands."""
    qkv = ln1.float() @ wq.float().t() + bq
    qkv[:, :D] *= scale
    q, k, v = qkv.bfloat16().float().view(n_img, 256, 3, H, D // H).permute(2, 0, 3, 1, 4)      # (the q | k | v rows are bf16 tensors in the reference's autocast step too)
    s = q @ k.transpose(-1, -2)
    ctx = (torch.softmax(s, dim=-1) @ v).permute(0, 2, 1, 3).reshape(n_img * 256, D)
    return x.float() + ctx @ wp.float().t() + bp, ctx, qkv, torch.logsumexp(s, dim=-1).reshape(n_img * H, 256)


@pytest.mark.parametrize("n_img,spike", [(1, False), (3, True), (64, False)])
def test_attn_block_fwd(dev, n_img, spike):
    """dig_attn_block_fwd (qkv Linear -> attention -> proj Linear + residual in one launch, csrc/attn_block.hip) against fp32 torch and
    against the three launches it replaces: the q | k | v rows bit for bit, lse to fp32 round-off, ctx / x_mid to bf16 rounding; the form
    that keeps nothing (momentum branch) writes the same ctx / x_mid as the form that keeps qkv and lse."""
    from dig_amd import ops
    D, H = 384, 6
    if not ops.attn_block_supported(H, D):
        pytest.skip("DIG_ATTN_BLOCK=0")
    R = n_img * 256
    cpu_limit(dev, 2.0 * R * D * 4 * D + 4.0 * R * 256 * D, 3e9)
    scale = (D // H) ** -0.5
    ln1, x = torch.randn(R, D, device=dev).bfloat16(), torch.randn(R, D, device=dev).bfloat16()
    wq, wp = (torch.randn(3 * D, D, device=dev) * 0.05).bfloat16(), (torch.randn(D, D, device=dev) * 0.05).bfloat16()
    bq, bp = torch.randn(3 * D, device=dev) * 0.3, torch.randn(D, device=dev) * 0.3
    bq[D:2 * D] = 0
    if spike:                                                          # one dominant key in the SECOND half of the keys: the running maximum moves
        ln1[200] *= 4.0
        ln1[256 + 7] *= 5.0
    x_mid, ctx, qkv, lse = ops.attn_block_fwd(ln1, x, wq, bq, wp, bp, n_img, H, D, scale, save=True)
    rx, rctx, rqkv, rlse = _attn_block_reference(ln1, x, wq, bq, wp, bp, n_img, H, D, scale)
    assert rel(qkv, rqkv) < 1e-2 and rel(ctx, rctx) < 1e-2 and rel(x_mid, rx) < 1e-2
    assert (lse - rlse).abs().max().item() < 2e-2
    x_mid0, ctx0, qkv0, lse0 = ops.attn_block_fwd(ln1, x, wq, bq, wp, bp, n_img, H, D, scale, save=False)
    assert qkv0 is None and lse0 is None and torch.equal(ctx0, ctx) and torch.equal(x_mid0, x_mid)
    # the three launches
    qkv3 = ops.linear_fwd(ln1, wq, bias=bq, alpha=scale, alpha_cols=D)
    ctx3, lse3 = ops.attn_fwd(qkv3, n_img, H, D)
    x_mid3 = ops.linear_fwd(ctx3, wp, bias=bp, resid=x)
    assert torch.equal(qkv, qkv3)
    assert (lse - lse3).abs().max().item() < 1e-4
    assert rel(ctx, ctx3.float()) < 2e-3 and rel(x_mid, x_mid3.float()) < 2e-3
    # fed with the fused launch's qkv / ctx / lse, the backward kernels give what they give for the three launches' (same values up to rounding)
    dctx = torch.randn(R, D, device=dev).bfloat16()
    assert rel(ops.attn_bwd(qkv, ctx, dctx, lse, n_img, H, D, scale), ops.attn_bwd(qkv3, ctx3, dctx, lse3, n_img, H, D, scale).float()) < 2e-3


def test_attn_block_fwd_exact_arithmetic(dev):
    """Every product and every fp32 sum of the fused launch is exact here, so the result must EQUAL the definition.  ln1 carries three 64-channel
    patterns (a one-hot query pattern, a one-hot key pattern with every channel hot in four tokens, integer values); the qkv weight routes them
    into every head through head-specific channel permutations (one 1 per row), so a score is 256 where query and key meet and 0 elsewhere:
    exp(-256) is 0 in fp32, the softmax is exactly uniform over the four matching keys wherever they sit in the two halves of the key loop (a
    first half without a match is multiplied by exp(0 - 256) = 0 when the second one raises the maximum), the context is the integer mean of
    their value rows, and the projection (one +-1/2 per row) + bias + residual stays in bf16's exact range."""
    from dig_amd import ops
    D, H, n_img = 384, 6, 2
    if not ops.attn_block_supported(H, D):
        pytest.skip("DIG_ATTN_BLOCK=0")
    R = n_img * 256
    g = torch.Generator().manual_seed(11)
    a = torch.randint(0, 64, (R,), generator=g)                                    # hot query channel of a token
    b = torch.stack([torch.arange(256).remainder(64)[torch.randperm(256, generator=g)] for _ in range(n_img)]).reshape(R)
    vals = torch.randint(-8, 9, (R, 64), generator=g).float() * 4.0
    ln1 = torch.zeros(R, D)
    ln1[torch.arange(R), a] = 128.0                                                # q = 128 * scale = 16
    ln1[torch.arange(R), 64 + b] = 16.0
    ln1[:, 128:192] = vals
    pq = [torch.randperm(64, generator=g) for _ in range(H)]
    pk = [torch.randperm(64, generator=g) for _ in range(H)]
    pv = [torch.randperm(64, generator=g) for _ in range(H)]
    wq = torch.zeros(3 * D, D)
    for h in range(H):
        d = torch.arange(64)
        wq[h * 64 + d, pq[h]] = 1.0
        wq[D + h * 64 + d, 64 + pk[h]] = 1.0
        wq[2 * D + h * 64 + d, 128 + pv[h]] = 1.0
    bq = torch.zeros(3 * D)
    bq[2 * D:] = torch.randint(-2, 3, (D,), generator=g).float() * 4.0             # v_bias: multiples of 4
    wp = torch.zeros(D, D)
    wp[torch.arange(D), torch.randperm(D, generator=g)] = torch.randint(0, 2, (D,), generator=g).float() - 0.5
    bp = torch.randint(-4, 5, (D,), generator=g).float()
    x = torch.randint(-8, 9, (R, D), generator=g).float()
    scale = 0.125
    # the definition, in exact arithmetic
    qkv = ln1 @ wq.t() + bq
    q, k, v = qkv.view(n_img, 256, 3, H, 64).permute(2, 0, 3, 1, 4)
    match = ((q * scale) @ k.transpose(-1, -2) == 256.0).float()
    assert bool((match.sum(-1) == 4).all())
    rctx = ((match @ v) / 4.0).permute(0, 2, 1, 3).reshape(R, D)
    rx = x + rctx @ wp.t() + bp
    qkv[:, :D] *= scale
    to = lambda t: t.to(dev)
    x_mid, ctx, qkv_d, lse = ops.attn_block_fwd(to(ln1.bfloat16()), to(x.bfloat16()), to(wq.bfloat16()), to(bq), to(wp.bfloat16()), to(bp),
                                                n_img, H, D, scale, save=True)
    assert torch.equal(qkv_d.float().cpu(), qkv) and torch.equal(ctx.float().cpu(), rctx) and torch.equal(x_mid.float().cpu(), rx)
    assert (lse.cpu() - (256.0 + math.log(4.0))).abs().max().item() < 1e-4


@pytest.mark.parametrize("n_img", [2, 64])
def test_encoder_block_calls_equal_the_entry_point_sequence(dev, n_img):
    """dig_encoder_block_fwd / dig_encoder_block_bwd (one FFI crossing per encoder block; include/dig_block_types.h) against the sequence of
    entry points they stand for (dig_amd/engine_core.py's per-entry-point plan): every output, every gradient and the fold of the grouped
    weight gradients across two consecutive blocks, bit for bit; both forward forms (online: everything kept; momentum: nothing kept)."""
    from dig_amd import ops
    D, Fh, H, eps = 384, 1536, 6, 1e-6
    R = n_img * 256
    cpu_limit(dev, 40.0 * R * D * D, 2e10)
    scale = (D // H) ** -0.5
    g = torch.Generator(device="cpu").manual_seed(n_img)
    bf = lambda *sh, s=0.5: (torch.randn(*sh, generator=g) * s).bfloat16().to(dev)
    f32 = lambda *sh, s=0.1, b=0.0: (torch.randn(*sh, generator=g) * s + b).to(dev)
    P = {"qkv_w": bf(3 * D, D, s=0.05), "qkv_b": f32(3 * D), "proj_w": bf(D, D, s=0.05), "proj_b": f32(D), "n1_g": f32(D, b=1.0), "n1_b": f32(D),
         "n2_g": f32(D, b=1.0), "n2_b": f32(D), "fc1_w": bf(Fh, D, s=0.05), "fc1_b": f32(Fh), "fc2_w": bf(D, Fh, s=0.03), "fc2_b": f32(D),
         "nn1_g": f32(D, b=1.0), "nn1_b": f32(D)}
    P["qkv_b"][D:2 * D] = 0                                           # K has no bias
    x = bf(R, D, s=1.0)
    ln1, mu1, rs1 = ops.layernorm_fwd(x, P["n1_g"], P["n1_b"], eps)

    def fwd_entry_points(save, last, fuse=True):
        if fuse and ops.attn_block_supported(H, D):
            x_mid, ctx, qkv, lse = ops.attn_block_fwd(ln1, x, P["qkv_w"], P["qkv_b"], P["proj_w"], P["proj_b"], n_img, H, D, scale, save=save)
        else:
            qkv = ops.linear_fwd(ln1, P["qkv_w"], bias=P["qkv_b"], alpha=scale, alpha_cols=D)
            ctx, lse = ops.attn_fwd(qkv, n_img, H, D)
            x_mid = ops.linear_fwd(ctx, P["proj_w"], bias=P["proj_b"], resid=x)
        r = ops.mlp_chain_fwd_ln(x_mid, P["n2_g"], P["n2_b"], eps, P["fc1_w"], P["fc1_b"], P["fc2_w"], P["fc2_b"],
                                 None if last else P["nn1_g"], None if last else P["nn1_b"], save=save)
        return dict(r, qkv=qkv, ctx=ctx, lse=lse, x_mid=x_mid)

    def fwd_block_call(save, last, fuse=True):
        off, n16, n32 = ops.block_fwd_layout(R, D, Fh, n_img, H, save)
        b16, b32 = torch.full((n16 // 2,), 7.0, device=dev, dtype=torch.bfloat16), torch.full((n32 // 4,), 7.0, device=dev)
        p16, p32 = b16.data_ptr(), b32.data_ptr()
        st = ops.BlockFwd(n_img=n_img, heads=H, D=D, F=Fh, rows=R, save=int(save), tile_qkv=ops.fwd_tile_code(R, 3 * D, D) or ops.GEMM_BK_FWD,
                          tile_proj=ops.fwd_tile_code(R, D, D, has_resid=True) or ops.GEMM_BK_FWD, fuse_attn=int(fuse), eps=eps, scale=scale,
                          qkv_w=P["qkv_w"].data_ptr(), qkv_b=P["qkv_b"].data_ptr(), proj_w=P["proj_w"].data_ptr(), proj_b=P["proj_b"].data_ptr(),
                          n2_g=P["n2_g"].data_ptr(), n2_b=P["n2_b"].data_ptr(), fc1_w=P["fc1_w"].data_ptr(), fc1_b=P["fc1_b"].data_ptr(),
                          fc2_w=P["fc2_w"].data_ptr(), fc2_b=P["fc2_b"].data_ptr(), next_n1_g=None if last else P["nn1_g"].data_ptr(),
                          next_n1_b=None if last else P["nn1_b"].data_ptr(), x=x.data_ptr(), ln1=ln1.data_ptr())
        for k in off:
            setattr(st, k, (p32 if k in ("lse", "mu2", "rs2", "nmu", "nrs") else p16) + off[k])
        ops.L.call("dig_encoder_block_fwd", ctypes.byref(st), ops.L.stream())

        def v(name, cols=D):
            if name == "lse":
                return b32[off[name] // 4:][:n_img * H * 256].view(n_img * H, 256)
            if name in ("mu2", "rs2", "nmu", "nrs"):
                return b32[off[name] // 4:][:R]
            return b16[off[name] // 2:][:R * cols].view(R, cols)
        return v

    names = {"qkv": ("qkv", 3 * D), "ctx": ("ctx", D), "lse": ("lse", 0), "x_mid": ("x_mid", D), "out": ("out", D), "ln": ("ln2", D),
             "ln_mean": ("mu2", 0), "ln_rstd": ("rs2", 0), "pre": ("pre", Fh), "act": ("act", Fh), "nln": ("nln", D), "nln_mean": ("nmu", 0),
             "nln_rstd": ("nrs", 0)}
    for save, last, fuse in ((True, False, True), (False, False, True), (True, True, True), (False, True, True), (True, False, False), (False, True, False)):
        ref, v = fwd_entry_points(save, last, fuse), fwd_block_call(save, last, fuse)
        for k, (nm, cols) in names.items():
            if ref.get(k) is not None:
                assert torch.equal(ref[k], v(nm, cols)), (k, save, last, fuse)
    # ---- backward over two "blocks" (the same saved tensors twice, two incoming gradients): the second call folds the first one's slabs
    sv = fwd_entry_points(True, False)
    w2t, w1t = ops.transpose_bf16(P["fc2_w"]), ops.transpose_bf16(P["fc1_w"])
    dys = [bf(R, D, s=0.02), bf(R, D, s=0.02)]
    gnames = ("n1_g", "n1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "n2_g", "n2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")
    mkgrads = lambda: [{k: torch.full(P[k].shape, 0.125, device=dev, dtype=torch.float32) for k in gnames} for _ in range(2)]

    def bwd_entry_points(G, fuse_ln2):
        grp, outs = ops.WgradGroup(dev), []
        for dy, gk in zip(dys, G):
            assert grp.add(dy, sv["act"], gk["fc2_w"])
            dctx = None
            if fuse_ln2:
                r5 = ops.mlp_chain_bwd_ln(dy, w2t, sv["pre"], w1t, sv["x_mid"], P["n2_g"], sv["ln_mean"], sv["ln_rstd"],
                                          projt=projt if fuse_ln2 == 2 else None)
                dx_mid, dpre, bparts, lnp = r5[:4]
                dctx = r5[4] if fuse_ln2 == 2 else None
                fin2 = lambda lnp=lnp, gk=gk: ops.layernorm_finalize_parts(lnp, gk["n2_g"], gk["n2_b"], gk["fc2_b"])
            else:
                dln2, dpre, bparts = ops.mlp_chain_bwd(dy, w2t, sv["pre"], w1t)
                dx_mid, fin2, ws2 = ops.layernorm_bwd(dln2, sv["x_mid"], P["n2_g"], P["n2_b"], sv["ln_mean"], sv["ln_rstd"], dy, gk["n2_g"],
                                                      gk["n2_b"], out=dln2, dres_colsum=gk["fc2_b"], defer=True)
            assert grp.add(dpre, sv["ln"], gk["fc1_w"])
            assert grp.add(dx_mid, sv["ctx"], gk["proj_w"])
            if dctx is None:
                dctx = ops.linear_dgrad(dx_mid, P["proj_w"])
            dqkv, qs, vs = ops.attn_bwd(sv["qkv"], sv["ctx"], dctx, sv["lse"], n_img, H, D, scale, bias_sums=True)
            assert grp.add(dqkv, ln1, gk["qkv_w"])
            grp.launch()
            dln1 = ops.linear_dgrad(dqkv, P["qkv_w"], out=dctx)
            dx, fin1, ws1 = ops.layernorm_bwd(dln1, x, P["n1_g"], P["n1_b"], mu1, rs1, dx_mid, gk["n1_g"], gk["n1_b"], out=dln1,
                                              dres_colsum=gk["proj_b"], defer=True)
            ops.colsum_partials(bparts, gk["fc1_b"]); fin2()
            ops.colsum_partials(qs, gk["qkv_b"][:D]); ops.colsum_partials(vs, gk["qkv_b"][2 * D:]); fin1()
            outs.append(dx.clone())
        grp.flush()
        return outs

    def bwd_block_calls(G, fuse_ln2):
        plan = ops.wgrad_block_plan(dev, R, D, Fh)
        assert plan is not None
        off, n16, n32 = ops.block_bwd_layout(R, D, Fh, n_img)
        probs, outs, keep, slabs_prev = ((ops._WgProb * 4)(), (ops._WgProb * 4)()), [], [], None
        for n, (dy, gk) in enumerate(zip(dys, G)):
            t16, t32 = torch.empty(n16 // 2, device=dev, dtype=torch.bfloat16), torch.empty(n32 // 4, device=dev)
            keep += [t16, t32]
            p16, p32 = t16.data_ptr(), t32.data_ptr()
            slabs = plan["group"]._slabs(plan["slab_bytes"])
            plan["group"].set ^= 1
            st = ops.BlockBwd(n_img=n_img, heads=H, D=D, F=Fh, rows=R, tile_dgrad=ops.dgrad_tile_code(R, D) or ops.GEMM_BK_BWD, scale=scale,
                              qkv_w=P["qkv_w"].data_ptr(), proj_w=P["proj_w"].data_ptr(), w2t=w2t.data_ptr(), w1t=w1t.data_ptr(),
                              n1_g=P["n1_g"].data_ptr(), n1_b=P["n1_b"].data_ptr(), n2_g=P["n2_g"].data_ptr(), n2_b=P["n2_b"].data_ptr(),
                              g_n1_g=gk["n1_g"].data_ptr(), g_n1_b=gk["n1_b"].data_ptr(), g_qkv_w=gk["qkv_w"].data_ptr(), g_q_b=gk["qkv_b"].data_ptr(),
                              g_v_b=gk["qkv_b"][2 * D:].data_ptr(), g_proj_w=gk["proj_w"].data_ptr(), g_proj_b=gk["proj_b"].data_ptr(),
                              g_n2_g=gk["n2_g"].data_ptr(), g_n2_b=gk["n2_b"].data_ptr(), g_fc1_w=gk["fc1_w"].data_ptr(), g_fc1_b=gk["fc1_b"].data_ptr(),
                              g_fc2_w=gk["fc2_w"].data_ptr(), g_fc2_b=gk["fc2_b"].data_ptr(),
                              x=x.data_ptr(), ln1=ln1.data_ptr(), mu1=mu1.data_ptr(), rs1=rs1.data_ptr(), qkv=sv["qkv"].data_ptr(), ctx=sv["ctx"].data_ptr(),
                              lse=sv["lse"].data_ptr(), x_mid=sv["x_mid"].data_ptr(), ln2=sv["ln"].data_ptr(), mu2=sv["ln_mean"].data_ptr(),
                              rs2=sv["ln_rstd"].data_ptr(), pre=sv["pre"].data_ptr(), act=sv["act"].data_ptr(), dy=dy.data_ptr(),
                              wg_fn=plan["fn"], wg_wa=plan["wa"], wg_splits=plan["splits"], wg_n_wg=plan["n_wg"], wg_fold_n=4 if n else 0,
                              wg_fold_splits=plan["splits"], wg_trans=(ctypes.c_int * 4)(*plan["trans"]), wg_map=plan["wmap"].data_ptr(),
                              wg_slabs=slabs.data_ptr(), wg_fold_slabs=slabs_prev.data_ptr() if n else None,
                              wg_probs=ctypes.addressof(probs[n & 1]), wg_fold_probs=ctypes.addressof(probs[(n & 1) ^ 1]) if n else None,
                              side=ops.L.stream(), fuse_ln2=int(bool(fuse_ln2)), projt=projt.data_ptr() if fuse_ln2 == 2 else None)
            for k in ("dln2", "dpre", "dctx", "dqkv"):
                setattr(st, k, p16 + off[k])
            for k in ("bparts", "ws1", "ws2", "qs", "vs"):
                setattr(st, k, p32 + off[k])
            ops.L.call("dig_encoder_block_bwd", ctypes.byref(st), ops.L.stream())
            slabs_prev = slabs
            outs.append(t16[off["dctx"] // 2:][:R * D].view(R, D).clone())
        ops.L.call("dig_wgrad_group", None, 0, ctypes.addressof(probs[(len(dys) & 1) ^ 1]), 4, R, 1, None, ops.WGRAD_GROUP_SLOTS, None,
                   ops.L.ptr(slabs_prev), plan["splits"], plan["fn"], plan["wa"], ops.L.stream())
        return outs

    projt = ops.transpose_bf16(P["proj_w"])
    for fuse_ln2 in (2, 1, 0):                # norm2's backward (2: and the projection's data gradient) inside the fused MLP launch / as its own launch
        Ga, Gb = mkgrads(), mkgrads()
        dxa, dxb = bwd_entry_points(Ga, fuse_ln2), bwd_block_calls(Gb, fuse_ln2)
        for a, b in zip(dxa, dxb):
            assert torch.equal(a, b)
        for ga, gb_ in zip(Ga, Gb):
            for k in gnames:
                assert torch.equal(ga[k], gb_[k]), (k, fuse_ln2)
                assert not torch.equal(ga[k], torch.full_like(ga[k], 0.125)) or k == "qkv_b", k
    # a table with a missing pointer is refused
    assert ops.L.lib().dig_encoder_block_fwd(ctypes.byref(ops.BlockFwd(n_img=n_img, heads=H, D=D, F=Fh, rows=R)), None) == -1


@pytest.mark.parametrize("rows,I,J", [(1024, 4096, 4096), (512, 1024, 1024), (1024, 4096, 384)])
def test_linear_wgrad_assign_writes_what_the_accumulating_form_adds(dev, rows, I, J):
    """ops.linear_wgrad(assign=True) -- the heads' weight gradients right after zero_grad(): one launch that WRITES dw -- against the slab
    form that adds into a zeroed dw (same fp32 products, another split of the token sum) and against fp32 torch; stale contents of dw are
    overwritten, not added to; shapes below the tile threshold keep the accumulating form."""
    from dig_amd import ops
    cpu_limit(dev, 2.0 * rows * I * J, 3e9)
    g = torch.Generator(device="cpu").manual_seed(rows + I)
    dy = (torch.randn(rows, I, generator=g) * 0.5).bfloat16().to(dev)
    x = (torch.randn(rows, J, generator=g) * 0.5).bfloat16().to(dev)
    ref = dy.float().t() @ x.float()
    a = torch.zeros(I, J, device=dev)
    ops.linear_wgrad(dy, x, a)
    b = torch.full((I, J), 3.0, device=dev)
    ops.linear_wgrad(dy, x, b, assign=True)
    tiles = ((I + 127) // 128) * ((J + 127) // 128)
    if tiles >= 64:
        assert rel(b, ref) < 2e-5 and rel(b, a) < 2e-5
    else:
        assert rel(b - 3.0, ref) < 2e-5                     # too few tiles for a one-split launch: the accumulating form ran


def test_scalar_tails_of_infonce_and_the_log_line(dev):
    """dig_infonce_finish / dig_step_meters: the handful of scalars behind `contra_loss`, the four accuracies and a step's log line, against
    the framework expressions they replace (exact: one multiply / one copy per value)."""
    from dig_amd import ops
    stats = torch.tensor([[3.5, 7.0, 40.0], [2.25, 9.0, 61.0]], device=dev)
    contra, accs = ops.infonce_finish(stats, 2.0 * 0.2 / 512, 100.0 / 512)
    assert torch.equal(contra, (stats[0, 0] + stats[1, 0]) * (2.0 * 0.2 / 512)) and contra.dim() == 0
    assert torch.equal(accs, stats[:, 1:].reshape(4) * (100.0 / 512))
    counts = torch.tensor([179, 180, 12, 179, 255, 179, 178], device=dev, dtype=torch.int32).repeat(37)
    loss, pix, gn = torch.tensor(0.4772, device=dev), torch.tensor(0.0845, device=dev), torch.tensor(0.1073, device=dev)
    v = ops.step_meters(loss, contra, pix, accs, counts, gn)
    ref = torch.stack([loss, contra, pix, accs[0], accs[1], accs[2], accs[3], counts.min().float(), counts.max().float(), gn])
    assert torch.equal(v, ref)
    v = ops.step_meters(loss, contra, pix, accs, counts[:1], None)
    assert torch.equal(v[:9], torch.stack([loss, contra, pix, accs[0], accs[1], accs[2], accs[3], counts[0].float(), counts[0].float()])) and bool(torch.isnan(v[9]))


@pytest.mark.parametrize("dtype", [torch.bool, torch.uint8, torch.float32, torch.float64, torch.int32, torch.int64])
@pytest.mark.parametrize("keep", [1, 2])
def test_mask_views_u8(dev, dtype, keep):
    """dig_mask_views_u8 against the expressions it replaces (engine_for_pretraining_moco.py:99-104: bool cast, view-1 fill;
    modeling_pretrain_moco_mim_ori.py:497: view-major rows), bit-exact for every mask element type the loaders produce."""
    from dig_amd import ops
    B, V, N = 5, 2, 256
    g = torch.Generator(device="cpu").manual_seed(3)
    m = (torch.rand(B, V, N, generator=g) < 0.7)
    src = (m.to(dtype) * (3 if dtype not in (torch.bool,) else 1)).to(dtype).to(dev)              # non-zero, not just one
    ref = src.flatten(1).to(torch.bool).view(B, V, -1).clone()
    if keep == 1:
        ref[:, 1, :].fill_(0)
    ref = ref.permute(1, 0, 2).reshape(V * B, N).to(torch.uint8).contiguous()
    out = ops.mask_views_u8(src, keep)
    assert out is not None and out.dtype == torch.uint8 and torch.equal(out, ref)
    assert ops.mask_views_u8(src.to(torch.float16), keep) is None                                 # other element types: the caller's torch path


def test_loss_reductions_with_a_workspace_are_fixed_order(dev):
    """dig_mse_fwd_bwd_ws / dig_ce_rows_ws: the same values as the atomic forms up to summation order, the same BITS from launch to launch,
    the ticket word left at zero (a second launch on the same workspace gives the same result), += into the output."""
    from dig_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    M, C = 4099, 48
    pred = torch.randn(M, 64, generator=g).to(dev); target = torch.randn(M, C, generator=g).to(dev)
    outs = []
    for _ in range(3):
        loss = torch.full((1,), 0.5, device=dev)
        dpred = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
        ops.mse_fwd_bwd(pred, 64, target, M, C, 1.0, loss, dpred, C)
        outs.append(loss.clone())
    ref = ((pred[:, :C] - target) ** 2).mean()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    assert abs(float(outs[0]) - 0.5 - float(ref)) < 1e-5 * float(ref)
    n, m = 512, 1024
    logits0 = (torch.randn(n, m, generator=g) * 3).to(dev)
    res = []
    for _ in range(3):
        lg = logits0.clone(); out3 = torch.zeros(3, device=dev)
        ops.ce_rows(lg, 128, 0.25, out3)
        res.append((out3.clone(), lg))
    assert all(torch.equal(res[0][0], r[0]) and torch.equal(res[0][1], r[1]) for r in res[1:])
    lbl = torch.arange(n, device=dev) + 128
    ref_loss = torch.nn.functional.cross_entropy(logits0, lbl, reduction="sum")
    assert abs(float(res[0][0][0]) - float(ref_loss)) < 1e-5 * float(ref_loss)
    top5 = logits0.topk(5, dim=1).indices
    assert float(res[0][0][1]) == float((top5[:, 0] == lbl).sum()) and float(res[0][0][2]) == float((top5 == lbl[:, None]).any(1).sum())


@pytest.mark.parametrize("rows,out_dim,K", [(1024, 256, 4096), (1024, 384, 4096), (512, 256, 2048)])
def test_narrow_long_k_layers_in_r_slices(dev, rows, out_dim, K):
    """The R-sliced form of the heads' few-row, narrow-output, long-K layers (dig_gemm_bf16 out_kind 2 with non-transposed A +
    dig_reduce_partials_bf16) against the one-launch form and fp32 torch: forward y = x W^T and data gradient dx = dy W; exact on
    small-integer operands."""
    from dig_amd import ops
    cpu_limit(dev, 4.0 * rows * out_dim * K, 4e10)
    assert ops.narrow_splits(rows, out_dim, K) == K // 512 and ops.narrow_splits(65536, out_dim, K) == 1 and ops.narrow_splits(rows, 4096, K) == 1
    g = torch.Generator(device="cpu").manual_seed(rows + K)
    x = (torch.randn(rows, K, generator=g) * 0.5).bfloat16().to(dev)
    w = (torch.randn(out_dim, K, generator=g) * 0.05).bfloat16().to(dev)
    y = ops.linear_fwd(x, w)
    ops.NARROW_SPLIT = False
    try:
        y1 = ops.linear_fwd(x, w)
    finally:
        ops.NARROW_SPLIT = True
    ref = x.float() @ w.float().t()
    assert rel(y, ref) < 5e-3 and rel(y, y1) < 5e-3
    dy = (torch.randn(rows, K, generator=g) * 0.5).bfloat16().to(dev)
    w2 = (torch.randn(K, out_dim, generator=g) * 0.05).bfloat16().to(dev)          # a layer out_dim -> K: its data gradient is [rows, K] x [K, out_dim]
    dx = ops.linear_dgrad(dy, w2)
    assert rel(dx, dy.float() @ w2.float()) < 5e-3
    xi = torch.randint(-2, 3, (rows, K), generator=g).bfloat16().to(dev)
    wi = torch.randint(-1, 2, (out_dim, K), generator=g).bfloat16().to(dev)
    assert torch.equal(ops.linear_fwd(xi, wi).float(), (xi.float() @ wi.float().t()).bfloat16().float())
