"""CPU oracle (test infrastructure only) for the input transform either side of the hot path -- SURVEY.md 8(f) row N3.

Reference call chain: dataset/datasets.py:27-42 `DataAugmentationForMAE`: `transforms.Resize((input_h, input_w),
interpolation=3)` on the RGB PIL crop that dataset/dataset_image.py:128-160 opens, `ToTensor`, `Normalize(0.5, 0.5)`, and
`RandomMaskingGenerator` (masking_generator.py:12-49).  torchvision's Resize on a PIL image is `Image.resize(size,
BICUBIC)`, i.e. Pillow's C routine `ImagingResample` (third-party, not vendored in /root/reference; Pillow 12.2.0 in this
image): two separable passes (horizontal, then vertical) over 8-bit pixels with coefficients computed in double precision,
normalised, rounded to 22-bit fixed point, accumulated in int32 from 1 << 21 and clipped to [0, 255] after >> 22.  This
file restates that published algorithm in numpy (integer arithmetic, so bit-exact) and is pinned against Pillow itself by
tests/golden/input_pipeline.npz (oracle/ref_harness/gen_input_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size):
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter (support 2) over the full axis.
    Returns (ksize, bounds[out,2] = (xmin, count), kk[out, ksize] int32)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass(img, out_size, axis):
    """One 8-bit resampling pass along `axis` (0 = vertical, 1 = horizontal) of an HxWxC uint8 array."""
    in_size = img.shape[axis]
    _, bounds, kk = precompute_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)                      # [in, other, C]
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        lo, n = bounds[xx]
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[xx, :n].astype(np.int64), src[lo:lo + n], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(img, out_h, out_w):
    """`PIL.Image.resize((out_w, out_h), Image.BICUBIC)` of an RGB uint8 HxWx3 array: horizontal pass, then vertical
    (ImagingResample skips a pass whose size is unchanged)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.shape[1] != out_w:
        img = _pass(img, out_w, 1)
    if img.shape[0] != out_h:
        img = _pass(img, out_h, 0)
    return img


def to_tensor_normalize(img_u8, mean=0.5, std=0.5):
    """torchvision ToTensor (uint8 HWC -> float32 CHW / 255) followed by Normalize(mean, std) (datasets.py:31-37)."""
    x = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    return (x - np.float32(mean)) / np.float32(std)


def transform(img_u8, out_h=32, out_w=128):
    return to_tensor_normalize(resize_bicubic_u8(img_u8, out_h, out_w))


# ---------------------------------------------------------------------------------------------- masks on device
# The reference draws masks with numpy's global Mersenne Twister inside dataloader workers (masking_generator.py:32-48), a
# stream that depends on worker count and seeding order; no GPU generator can (or should) reproduce it.  The device
# generator keeps the distribution -- every view's mask is a uniformly random subset of exactly `num_mask` of the N
# patches -- and is itself deterministic: Philox4x32-10 keyed by (seed), counter (row, patch, step, 0); the num_mask
# patches with the smallest 32-bit key (ties by patch index) are masked.  This is the bit-exact statement of that rule.
_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(counter, key):
    c = [int(v) & 0xFFFFFFFF for v in counter]
    k = [int(v) & 0xFFFFFFFF for v in key]
    for _ in range(10):
        p0, p1 = _M0 * c[0], _M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + _W0) & 0xFFFFFFFF, (k[1] + _W1) & 0xFFFFFFFF]
    return c


def random_masks(n_rows, n_patches, num_mask, seed, step):
    """[n_rows, n_patches] uint8; row r = (sample * num_view + view)."""
    out = np.zeros((n_rows, n_patches), dtype=np.uint8)
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    for r in range(n_rows):
        keys = np.array([philox4x32_10((r, p, step & 0xFFFFFFFF, 0), key)[0] for p in range(n_patches)], dtype=np.uint64)
        order = np.lexsort((np.arange(n_patches), keys))
        out[r, order[:num_mask]] = 1
    return out
