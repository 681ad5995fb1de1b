"""Row N3 (input transform either side of the hot path): the numpy oracle against the Pillow-generated golden vectors
(CPU), and the kernels against the oracle bit for bit -- the HIP build on the GPU, the plain-C++ build of the same entry points
(cpu_abi/dig_cpu_rec.cpp) in the GPU-less container (`abi_dev`, tests/conftest.py)."""
import os

import numpy as np
import pytest

import input_oracle as IO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "input_pipeline.npz")


def _cases():
    z = np.load(GOLD)
    return z, int(z["n_cases"][0])


def test_oracle_resize_equals_pillow_fixture():
    z, n = _cases()
    for i in range(n):
        assert np.array_equal(IO.resize_bicubic_u8(z[f"in_{i}"], 32, 128), z[f"out_{i}"]), i
    assert np.array_equal(IO.to_tensor_normalize(z["out_0"]), z["norm_0"])


def test_oracle_resize_identity_and_constant():
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, size=(32, 128, 3)).astype(np.uint8)
    assert np.array_equal(IO.resize_bicubic_u8(img, 32, 128), img)             # both passes skipped
    flat = np.full((57, 211, 3), 137, np.uint8)
    assert (IO.resize_bicubic_u8(flat, 32, 128) == 137).all()                  # normalised coefficients sum to 1 << 22


def test_oracle_masks_are_fixed_size_subsets():
    z, _ = _cases()
    m = IO.random_masks(12, 256, int(z["mask_count"][0]), seed=1234, step=3)
    assert m.shape == (12, 256) and (m.sum(1) == 179).all()
    assert not np.array_equal(m[0], m[1])
    assert np.array_equal(m, IO.random_masks(12, 256, 179, seed=1234, step=3))          # counter-based: reproducible
    assert not np.array_equal(m, IO.random_masks(12, 256, 179, seed=1234, step=4))
    assert IO.philox4x32_10((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]   # Random123 known answer


def test_device_resize_normalize_bit_exact(abi_dev):
    import torch
    from dig_amd.datasets import resize_normalize as _rn
    resize_normalize = lambda crops, h, w: _rn(crops, h, w, device=abi_dev)
    z, n = _cases()
    crops = [z[f"in_{i}"] for i in range(n)]
    out = resize_normalize(crops, 32, 128).cpu().numpy()
    for i in range(n):
        want = IO.to_tensor_normalize(z[f"out_{i}"])
        assert np.array_equal(out[i], want), (i, np.abs(out[i] - want).max())
    rng = np.random.RandomState(7)                                              # ragged random batch, sizes 1..max
    crops = [rng.randint(0, 256, size=(rng.randint(1, 180), rng.randint(1, 700), 3)).astype(np.uint8) for _ in range(40)]
    out = resize_normalize(crops, 32, 128).cpu().numpy()
    for c, o in zip(crops, out):
        assert np.array_equal(o, IO.transform(c))
    out = resize_normalize(crops[:5], 48, 160).cpu().numpy()                    # another target size (args.input_h / input_w)
    for c, o in zip(crops, out):
        assert np.array_equal(o, IO.transform(c, 48, 160))


def test_device_masks_match_oracle_and_are_uniform(abi_dev):
    import torch
    from dig_amd.datasets import RandomMaskingGenerator
    g = RandomMaskingGenerator((8, 32), 0.7, num_view=2, seed=0x1234567890AB, device=abi_dev)
    m0 = g(6).cpu().numpy()
    m1 = g(6).cpu().numpy()
    assert m0.shape == (6, 2, 256) and (m0.sum(-1) == 179).all() and (m1.sum(-1) == 179).all()
    assert np.array_equal(m0.reshape(12, 256), IO.random_masks(12, 256, 179, 0x1234567890AB, 0))
    assert np.array_equal(m1.reshape(12, 256), IO.random_masks(12, 256, 179, 0x1234567890AB, 1))
    big = RandomMaskingGenerator((8, 32), 0.7, num_view=2, seed=5, device=abi_dev)(4096).float()       # 8192 rows
    freq = big.mean((0, 1)).cpu().numpy()                                                                # per-patch masking rate
    assert abs(freq.mean() - 179 / 256) < 1e-6
    assert np.abs(freq - 179 / 256).max() < 5 * np.sqrt(0.7 * 0.3 / 8192)                             # 5 sigma per position


@pytest.mark.gpu
def test_gpu_batch_transform_feeds_the_engine_shapes():
    import types
    import torch
    from dig_amd.datasets import GpuBatchTransform
    args = types.SimpleNamespace(input_h=32, input_w=128, window_size=(8, 32), mask_ratio=0.7, num_view=2)
    rng = np.random.RandomState(3)
    crops = [rng.randint(0, 256, size=(rng.randint(20, 60), rng.randint(60, 300), 3)).astype(np.uint8) for _ in range(8)]
    images, aug, masks = GpuBatchTransform(args, seed=1, device="cuda:0")(crops, crops[::-1])
    assert images.shape == (8, 3, 32, 128) and aug.shape == (8, 3, 32, 128) and masks.shape == (8, 2, 256)
    assert images.dtype == torch.float32 and float(images.min()) >= -1.0 and float(images.max()) <= 1.0
    assert np.array_equal(aug[0].cpu().numpy(), IO.transform(crops[-1]))
