"""Throughput of the recognition forward + greedy decode (row N4) at the fine-tune configuration: ViT-S encoder, tf_decoder
(6 layers, d 512), 97 classes, 25 steps, batch 256, random weights."""
import os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd.recognizer import RecModel
dev = torch.device("cuda:0")
args = types.SimpleNamespace(model="simmim_vit_small_patch4_32x128", decoder_name="tf_decoder", nb_classes=97, max_len=25)
m = RecModel(args).eval()
g = torch.Generator().manual_seed(0)
sd = {}
for k, s in m.param_shapes().items():
    if k.endswith("norm.weight") or ".norm1.weight" in k or ".norm2.weight" in k or ".norm3.weight" in k or k == "linear_norm.1.weight":
        sd[k] = torch.ones(s)
    elif k.endswith("bias"):
        sd[k] = torch.zeros(s)
    else:
        sd[k] = torch.randn(s, generator=g) * (0.5 if "emb" in k else 1.0 / (s[-1] ** 0.5))
m.load_state_dict(sd)
B = 256
images = (torch.rand(B, 3, 32, 128, generator=g) * 2 - 1).to(dev)
m.use_hip_graph = "--eager" not in sys.argv
for _ in range(2): out = m((images, None, None))
torch.cuda.synchronize(); t = time.perf_counter()
n = 5
for _ in range(n): out = m((images, None, None))
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
enc = m.encoder_features(images); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(n): enc = m.encoder_features(images)
torch.cuda.synchronize(); de = (time.perf_counter() - t) / n
print(f"recognize B={B} ({'HIP graph' if m.use_hip_graph else 'eager'}): {dt*1e3:.1f} ms per batch = {B/dt:.0f} images/s (encoder {de*1e3:.1f} ms, decode {1e3*(dt-de):.1f} ms for 25 steps x 6 layers)")
