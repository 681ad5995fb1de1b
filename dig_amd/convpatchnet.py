"""ConvPatchNet (`--patchnet_name conv`: run_mae_pretraining_moco.py:145; modeling_pretrain_moco_mim_ori.py:207-260) on the hot-path kernels.

forward (ConvPatchNet.forward, :250-258): the image tokens [n_img, 8 * 32, D] ARE the NHWC map of `seq_x.reshape(B, 8, 32, C).permute(0, 3, 1, 2)`
    four conv3x3_blocks (:239-248: Conv2d(k 3, pad 1, bias) -> BatchNorm2d -> ReLU), widths D -> D -> 1.5 D -> 2 D -> 2 D, a 2x2 max-pool after
    the first three (8x32 -> 4x16 -> 2x8 -> 1x4); adaptive_avg_pool2d to (1, num_windows), flattened window-major; patches2global =
    Linear(2 D nw -> D) -> BatchNorm1d -> ReLU -> Linear(D -> D) -> BatchNorm1d(affine=False).  ONE row per image.
Each convolution is a GEMM of csrc/gemm.hip over the im2col matrix (dig_im2col3x3: columns in conv.weight.view(C_out, -1)'s order, so the arena's
own weight / gradient views are the operands); BatchNorm2d in train mode = per-channel statistics over (batch, y, x) = dig_bn_* over the
[n_img H W, C] rows, cross-rank sums under a process group (SyncBatchNorm, run_mae_pretraining_moco.py:390) exactly like the BN-MLP heads.
backward: the explicit reverse.  The data gradient of a convolution is the convolution of dy with the flipped, transposed taps
(dig_conv3x3_weight_flip of this step's bf16 weights): im2col(dy) @ Wt^T, a direct-form GEMM again; the weight gradient is dy^T @ im2col(x)
on the saved matrix (side stream); the conv / Linear biases sit in front of a BatchNorm: their gradient is the column sum of a centred
matrix -- round-off, as in the reference."""
import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32
CONV_IDX = (0, 2, 4, 6)                 # positions of the conv3x3_blocks in ConvPatchNet.conv_layers (:217-225; max-pools at 1, 3, 5)


def channels(D):
    return (D, D, int(D * 1.5), D * 2, D * 2)


def _bn_fwd(step, h, key, arena_pre, relu, affine=True):
    """BatchNorm (train) over the rows of h with this step's cross-rank statistics; running buffers updated.  Returns (y, mean, rstd)."""
    M = step.m
    f32 = M._f32
    gamma = f32[f"{arena_pre}.{key}.weight"] if affine else None
    beta = f32[f"{arena_pre}.{key}.bias"] if affine else None
    rm, rv, i_bn = M._bn_views[f"{arena_pre}.{key}"]
    rows, C = h.shape
    from .engine_core import LOCAL
    if step.comm is LOCAL and ops.bn_fused_supported(rows, C):
        y, mean, rstd = ops.bn_fwd_fused(h, M.bn_eps, gamma, beta, relu=relu, running=(rm, rv, M.bn_momentum))
    else:
        sums = torch.empty((2, C), device=h.device, dtype=F32)
        ops.bn_stats(h, sums)
        step.comm.all_reduce_(sums)
        y, mean, rstd = ops.bn_fwd_apply(h, sums, float(rows * step.comm.world), M.bn_eps, gamma, beta, relu=relu, running=(rm, rv, M.bn_momentum))
    step._bn_touched.append(i_bn)
    return y, mean, rstd


def _bn_bwd(step, dy, h, mean, rstd, key, relu, affine=True):
    M = step.m
    f32, g32 = M._f32, M._g32
    pre = "patch_extractor"
    gamma = f32[f"{pre}.{key}.weight"] if affine else None
    beta = f32[f"{pre}.{key}.bias"] if affine else None
    dbeta = g32[f"{pre}.{key}.bias"] if affine else None
    dgamma = g32[f"{pre}.{key}.weight"] if affine else None
    rows, C = h.shape
    from .engine_core import LOCAL
    if step.comm is LOCAL and ops.bn_fused_supported(rows, C):
        return ops.bn_bwd_fused(dy, h, mean, rstd, gamma, beta, relu, dbeta, dgamma)
    sums = torch.empty((2, C), device=dy.device, dtype=F32)
    ops.bn_bwd_stats(dy, h, mean, rstd, gamma, beta, relu, sums, dbeta, dgamma)
    step.comm.all_reduce_(sums)
    return ops.bn_bwd_apply(dy, h, mean, rstd, gamma, beta, relu, sums, float(rows * step.comm.world))


def _conv_gemm(a, w, rows, out_cols, bias=None):
    """[rows, out_cols] = a[rows, Kp] @ w[out_cols, K]^T: K <= Kp = a's row pitch; where the im2col matrix carries pad columns (ViT-Tiny's
    288-channel map: 2592 -> 2624) the weight rows wrap into their successors under zero columns of a -- the 128x128 kernel, as the kernel
    test covers it."""
    Kp = a.shape[1]
    bk = ops.fwd_tile_code(rows, out_cols, Kp) if w.shape[1] == Kp else 0
    return ops.gemm(a, w, rows, out_cols, Kp, bias=bias, bk=bk)


def forward(step, feat, pre, arena, n_img, save):
    """feat: bf16 [n_img * 256, D] image tokens of [masked view | augmented view].  Returns (out bf16 [n_img, D], saved or None)."""
    M = step.m
    D, nw = M.D, M.num_windows
    w16, f32 = M._w(arena), M._f32
    c = channels(D)
    H, W = M.gh, M.gw
    x = feat
    layers = []
    for j, i in enumerate(CONV_IDX):
        key = f"conv_layers.{i}"
        col = ops.im2col3x3(x, n_img, H, W, c[j])
        h = _conv_gemm(col, w16[f"{pre}.{key}.0.weight"], n_img * H * W, c[j + 1], bias=f32[f"{pre}.{key}.0.bias"])
        a, mean, rstd = _bn_fwd(step, h, key + ".1", pre, relu=True)
        idx = None
        if j < 3:
            a, idx = ops.maxpool2x2_fwd(a, n_img, H, W, c[j + 1])
        if save:
            layers.append((col, h, mean, rstd, idx, H, W))
        if j < 3:
            H, W = H // 2, W // 2
        x = a
    g = torch.empty((n_img * nw, c[4]), device=feat.device, dtype=BF16)
    ops.window_pool_fwd(x, g, n_img, H, W, nw, c[4])                       # adaptive_avg_pool2d((1, nw)) of the [1, 4] map, rows = (image, window)
    g = g.view(n_img, nw * c[4])
    z = ops.linear_fwd(g, w16[f"{pre}.patches2global.0.weight"], bias=f32[f"{pre}.patches2global.0.bias"])
    a1, mean1, rstd1 = _bn_fwd(step, z, "patches2global.1", pre, relu=True)
    z2 = ops.linear_fwd(a1, w16[f"{pre}.patches2global.3.weight"], bias=f32[f"{pre}.patches2global.3.bias"])
    out, mean2, rstd2 = _bn_fwd(step, z2, "patches2global.4", pre, relu=False, affine=False)
    return out, ((layers, (H, W), g, z, mean1, rstd1, a1, z2, mean2, rstd2) if save else None)


def backward(step, dout, pre, saved, n_img):
    """dout: bf16 [n_img, D].  Accumulates every parameter gradient of `pre` (online arena); returns d feat [n_img * 256, D]."""
    M = step.m
    D, nw = M.D, M.num_windows
    w16, g32 = M._w("online"), M._g32
    c = channels(D)
    dev = dout.device
    layers, (H, W), g, z, mean1, rstd1, a1, z2, mean2, rstd2 = saved
    side = lambda fn, *t: step._on_side(dev, fn, *t)                     # noqa: E731  (weight gradients / column sums: consumers only)
    p2g = f"{pre}.patches2global"
    dz2 = _bn_bwd(step, dout, z2, mean2, rstd2, "patches2global.4", relu=False, affine=False)
    side(lambda: (ops.linear_wgrad(dz2, a1, g32[p2g + ".3.weight"]), ops.colsum(dz2, g32[p2g + ".3.bias"])), dz2, a1)
    da1 = ops.linear_dgrad(dz2, w16[p2g + ".3.weight"])
    dz = _bn_bwd(step, da1, z, mean1, rstd1, "patches2global.1", relu=True)
    side(lambda: (ops.linear_wgrad(dz, g, g32[p2g + ".0.weight"]), ops.colsum(dz, g32[p2g + ".0.bias"])), dz, g)
    dg = ops.linear_dgrad(dz, w16[p2g + ".0.weight"])                    # [n_img, nw * 2 D]
    da = torch.empty((n_img * H * W, c[4]), device=dev, dtype=BF16)
    ops.window_pool_bwd(dg.view(n_img * nw, c[4]), da, n_img, H, W, nw, c[4], False)
    for j in reversed(range(4)):
        key = f"{pre}.conv_layers.{CONV_IDX[j]}"
        col, h, mean, rstd, idx, H, W = layers[j]
        layers[j] = None
        if idx is not None:
            da = ops.maxpool2x2_bwd(da, idx, n_img, H, W, c[j + 1])
        dh = _bn_bwd(step, da, h, mean, rstd, f"conv_layers.{CONV_IDX[j]}.1", relu=True)
        gw_ = g32[key + ".0.weight"].view(c[j + 1], c[j] * 9)
        side(lambda dh=dh, col=col, gw_=gw_, key=key, j=j: (ops.wgrad(dh, col, gw_, c[j + 1], c[j] * 9, dh.shape[0]),
                                                             ops.colsum(dh, g32[key + ".0.bias"])), dh, col)
        cold = ops.im2col3x3(dh, n_img, H, W, c[j + 1])
        wt = ops.conv3x3_weight_flip(w16[key + ".0.weight"], c[j + 1], c[j])
        da = _conv_gemm(cold, wt, n_img * H * W, c[j])
    return da
