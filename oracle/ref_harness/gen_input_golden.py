"""Golden vectors for the input transform (row N3): Pillow's own bicubic resize (what torchvision's
transforms.Resize(..., interpolation=3) calls on the PIL crops of dataset/dataset_image.py) on seeded uint8 crops of the
sizes scene-text crops come in, the ToTensor/Normalize result, and the reference's RandomMaskingGenerator invariants.
Asserts the numpy restatement (oracle/input_oracle.py) equals Pillow bit for bit, then writes tests/golden/input_pipeline.npz.

    python oracle/ref_harness/gen_input_golden.py          # needs Pillow and /root/reference (this container only)"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import input_oracle as IO

SIZES = [(32, 128), (31, 100), (48, 160), (64, 256), (17, 40), (100, 300), (25, 333), (8, 16), (200, 64), (37, 128), (32, 77),
         (3, 5), (150, 600), (33, 129)]


def main():
    rng = np.random.RandomState(20240607)
    out = {}
    for n, (h, w) in enumerate(SIZES):
        kind = n % 3
        if kind == 0:
            img = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)                     # noise: every clip path
        elif kind == 1:
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) % 2) * 255], -1).astype(np.uint8)
        else:
            img = np.zeros((h, w, 3), np.uint8)
            img[h // 4:h - h // 4, w // 5:w - w // 5] = rng.randint(0, 256, size=3)           # flat text-like block: over/undershoot
        ref = np.asarray(Image.fromarray(img, "RGB").resize((128, 32), Image.BICUBIC))
        mine = IO.resize_bicubic_u8(img, 32, 128)
        assert np.array_equal(ref, mine), (h, w, np.abs(ref.astype(int) - mine.astype(int)).max())
        out[f"in_{n}"] = img
        out[f"out_{n}"] = ref
    # ToTensor + Normalize(0.5, 0.5) semantics, with torch doing the float arithmetic exactly as torchvision does
    import torch
    x = torch.from_numpy(out["out_0"].transpose(2, 0, 1).copy()).to(torch.float32).div(255)
    x = (x - 0.5) / 0.5
    assert np.array_equal(x.numpy(), IO.to_tensor_normalize(out["out_0"]))
    out["norm_0"] = x.numpy()
    # the reference's mask generator: shape and per-view count (the only properties a device generator can share with it)
    sys.path.insert(0, "/root/reference")
    from masking_generator import RandomMaskingGenerator
    g = RandomMaskingGenerator((8, 32), 0.7, num_view=2)
    np.random.seed(0)
    m = g()
    assert m.shape == (2, 256) and (m.sum(1) == 179).all() and set(np.unique(m)) == {0.0, 1.0}
    out["mask_shape"] = np.array(m.shape)
    out["mask_count"] = np.array([g.num_mask])
    out["n_cases"] = np.array([len(SIZES)])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "input_pipeline.npz"), **out)
    print("wrote tests/golden/input_pipeline.npz:", len(SIZES), "resize cases; oracle == Pillow", Image.__version__ if hasattr(Image, "__version__") else "")


if __name__ == "__main__":
    main()
