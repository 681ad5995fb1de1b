"""Throughput of the device input transform (row N3) on batches of scene-text-sized crops."""
import os, sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd import _lib as L
from dig_amd.datasets import resize_normalize, RandomMaskingGenerator
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
for name, hh, ww in (("32x100..64x320", (32, 64), (100, 320)), ("100x300..200x700", (100, 200), (300, 700))):
    crops = [rng.randint(0, 256, size=(rng.randint(*hh), rng.randint(*ww), 3)).astype(np.uint8) for _ in range(256)]
    t = time.perf_counter(); out = resize_normalize(crops); torch.cuda.synchronize(); t_all = time.perf_counter() - t
    # kernel alone (buffers already on the device)
    n = len(crops)
    hs = np.array([c.shape[0] for c in crops], np.int32); ws = np.array([c.shape[1] for c in crops], np.int32)
    sizes = hs.astype(np.int64) * ws * 3; offs = np.zeros(n, np.int64); np.cumsum(sizes[:-1], out=offs[1:])
    packed = torch.from_numpy(np.concatenate([c.reshape(-1) for c in crops])).to(dev)
    d_off, d_h, d_w = torch.from_numpy(offs).to(dev), torch.from_numpy(hs).to(dev), torch.from_numpy(ws).to(dev)
    o = torch.empty((n, 3, 32, 128), device=dev)
    def k():
        L.call("dig_resize_bicubic_normalize_u8", L.ptr(packed), L.ptr(d_off), L.ptr(d_h), L.ptr(d_w), n, L.ptr(o), 32, 128,
               ctypes.c_float(0.5), ctypes.c_float(0.5), int(hs.max()), int(ws.max()), L.stream())
    for _ in range(3): k()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): k()
    torch.cuda.synchronize(); tk = (time.perf_counter() - t) / 20
    print(f"{name}: {n} crops, {sizes.sum()/1e6:.1f} MB uint8 in, kernel {tk*1e6:.0f} us ({n/tk:.0f} crops/s), pack+upload+kernel {t_all*1e3:.1f} ms")
g = RandomMaskingGenerator((8, 32), 0.7, num_view=2, seed=1, device="cuda:0")
for _ in range(3): g(128)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(20): g(128)
torch.cuda.synchronize(); print(f"masks for 128 samples x 2 views: {(time.perf_counter()-t)/20*1e6:.0f} us")
