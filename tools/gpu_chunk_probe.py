"""Does producer -> consumer locality in the 256 MiB Infinity Cache matter for the encoder forward?  Time the gradient-free
(momentum-style) encoder forward on the full 2B-image batch against the same work issued as 2 / 4 / 8 image chunks, one stream."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dig_amd.registry import create_model
from dig_amd import engine_core

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = create_model("pretrain_simmim_moco_ori_vit_small_patch4_32x128", pretrained=False, drop_path_rate=0.0, drop_block_rate=None, mlp_dim=4096,
                     dim=256, T=0.2, num_windows=4, encoder_type='vit', queue_size=65536, patchnet_name='no_patchtrans')
model.to(dev)
B = 128
im = torch.rand((B, 3, 32, 128), device=dev) * 2 - 1
au = torch.rand((B, 3, 32, 128), device=dev) * 2 - 1
mask = torch.zeros((2 * B, 256), device=dev, dtype=torch.uint8)
step = engine_core._Step(model)
ew_on, ew_mo = engine_core._weights(model)


def run(chunks, save):
    n = B // chunks
    for c in range(chunks):
        s = slice(c * n, (c + 1) * n)
        m = torch.cat([mask[:B][s], mask[B:][s]])
        step.encoder_forward(ew_on, im[s], au[s], m, save)


for save in (False, True):
    for chunks in (1, 2, 4, 8):
        for _ in range(3):
            run(chunks, save)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            run(chunks, save)
        torch.cuda.synchronize()
        print(f"encoder forward save={save} chunks={chunks}: {(time.perf_counter() - t) / 10 * 1e3:.3f} ms", flush=True)
