"""Device-side input transform: the reference's `DataAugmentationForMAE` (dataset/datasets.py:27-52) with the
resize / ToTensor / Normalize and the mask generator running on the MI355X instead of in dataloader workers.

The reference pipeline per sample (dataset/dataset_image.py:128-160): decode the crop to an RGB PIL image, optionally run
the imgaug augmentor on it (CPU, stays on the host), then `transform(img) -> (tensor [3,32,128], mask [num_view, 256])`.
Here the dataloader hands over the *decoded uint8 crops* (any size); `GpuBatchTransform` packs them into one pinned
buffer, uploads it once (H x W x 3 bytes per crop instead of 3 x 32 x 128 x 4 after resize -- usually less), and one kernel
per view produces the normalised fp32 batch bit-exactly as Pillow + torchvision would; masks are drawn on the device.

    tf = GpuBatchTransform(args)                       # args.input_h/input_w/window_size/mask_ratio/num_view as in the reference
    images, aug_images, masks = tf(crops, aug_crops)   # lists of HxWx3 uint8 arrays -> the `batch` triple of train_one_epoch
"""
import ctypes

import numpy as np
import torch

from . import _lib as L


class RandomMaskingGenerator:
    """masking_generator.py:12-49 on the device: `__call__(n)` returns [n, num_view, num_patches] uint8 with exactly
    `num_mask` ones per view.  The reference draws from numpy's global Mersenne Twister inside dataloader workers; this
    generator keeps the distribution (uniform over subsets of that size) with its own counter-based stream (Philox4x32-10,
    key = seed, counter = (row, patch, call index)), so a run is reproducible from (seed, step) on any number of ranks."""

    def __init__(self, input_size, mask_ratio, aug_ratio=0., num_view=1, seed=0, device="cuda"):
        if not isinstance(input_size, tuple):
            input_size = (input_size,) * 2
        self.height, self.width = input_size
        self.num_patches = self.height * self.width
        self.num_mask = int(mask_ratio * self.num_patches)
        self.num_aug = int(self.num_mask * aug_ratio)
        self.num_view = num_view
        self.seed, self.step, self.device = int(seed), 0, torch.device(device)

    def __repr__(self):
        return "Mask: total patches {}, mask patches {}".format(self.num_patches, self.num_mask)

    def __call__(self, n=1):
        rows = n * self.num_view
        mask = torch.empty((rows, self.num_patches), device=self.device, dtype=torch.uint8)
        L.call("dig_random_masks", L.ptr(mask), rows, self.num_patches, self.num_mask, ctypes.c_ulonglong(self.seed),
               ctypes.c_uint(self.step & 0xFFFFFFFF), L.stream())
        self.step += 1
        return mask.view(n, self.num_view, self.num_patches)


def resize_normalize(crops, out_h=32, out_w=128, mean=0.5, std=0.5, device="cuda"):
    """List of HxWx3 uint8 numpy arrays -> fp32 [n, 3, out_h, out_w] on `device`:
    Normalize(mean, std)(ToTensor(Resize((out_h, out_w), interpolation=BICUBIC)(PIL crop)))  (datasets.py:31-37)."""
    n = len(crops)
    hs = np.array([c.shape[0] for c in crops], dtype=np.int32)
    ws = np.array([c.shape[1] for c in crops], dtype=np.int32)
    sizes = hs.astype(np.int64) * ws.astype(np.int64) * 3
    offs = np.zeros(n, dtype=np.int64)
    np.cumsum(sizes[:-1], out=offs[1:])
    dev = torch.device(device)
    pin = (lambda t: t.pin_memory()) if dev.type == "cuda" else (lambda t: t)       # (pinned staging only where there is a device to copy to)
    packed = pin(torch.empty(int(sizes.sum()), dtype=torch.uint8))
    flat = packed.numpy()
    for c, o, s in zip(crops, offs, sizes):
        if c.dtype != np.uint8 or c.ndim != 3 or c.shape[2] != 3:
            raise ValueError("crops must be HxWx3 uint8 (the RGB image PIL decodes)")
        flat[o:o + s] = np.ascontiguousarray(c).reshape(-1)
    d_packed = packed.to(dev, non_blocking=True)
    d_meta = pin(torch.from_numpy(np.concatenate([offs, hs.astype(np.int64), ws.astype(np.int64)]))).to(dev, non_blocking=True)
    d_off, d_h, d_w = d_meta[:n], d_meta[n:2 * n].to(torch.int32), d_meta[2 * n:].to(torch.int32)
    out = torch.empty((n, 3, out_h, out_w), device=dev, dtype=torch.float32)
    L.call("dig_resize_bicubic_normalize_u8", L.ptr(d_packed), L.ptr(d_off), L.ptr(d_h), L.ptr(d_w), n, L.ptr(out), out_h, out_w,
           ctypes.c_float(mean), ctypes.c_float(std), int(hs.max()), int(ws.max()), L.stream())
    return out


class GpuBatchTransform:
    """Batch form of DataAugmentationForMAE.__call__ (datasets.py:41-42): (crops, aug_crops) -> (images, aug_images, masks)."""

    def __init__(self, args, seed=0, device="cuda"):
        self.h, self.w = args.input_h, args.input_w
        self.device = device
        self.masked_position_generator = RandomMaskingGenerator(args.window_size, args.mask_ratio, num_view=args.num_view,
                                                                seed=seed, device=device)

    def __call__(self, crops, aug_crops=None):
        images = resize_normalize(crops, self.h, self.w, device=self.device)
        aug = resize_normalize(aug_crops, self.h, self.w, device=self.device) if aug_crops is not None else None
        masks = self.masked_position_generator(len(crops))
        return images, aug, masks
