"""Throughput of the AttnRecModel training step (GRU attention head, `--decoder_type attention`): simmim_vit_small_patch4_32x128 encoder,
sDim = attDim = 512, 97 classes, max_len 25, batch 256, README drop rates for the encoder, AdamW with layer decay 0.75; random weights
and labels.  Also the greedy sample (evaluation) rate."""
import os, sys, time, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd.attn_recognizer import AttnRecModelTrain
from dig_amd.finetune import SeqCrossEntropyLoss, LayerDecayValueAssigner, create_optimizer
from dig_amd.utils import NativeScalerWithGradNormCount
dev = torch.device("cuda:0")
args = types.SimpleNamespace(model="simmim_vit_small_patch4_32x128", nb_classes=97, max_len=25, drop=0.1, attn_drop_rate=0.1, drop_path=0.1,
                             opt="adamw", lr=1e-4, weight_decay=0.05, opt_eps=1e-8, opt_betas=[0.9, 0.999])
torch.manual_seed(0)
m = AttnRecModelTrain(args)
m.to(dev); m.train()
nl = m.get_num_layers()
asg = LayerDecayValueAssigner([0.75 ** (nl + 1 - i) for i in range(nl + 2)])
opt = create_optimizer(args, m, get_num_layer=asg.get_layer_id, get_layer_scale=asg.get_scale)
for grp in opt.param_groups: grp["lr"] = args.lr * grp["lr_scale"]
B = 256
g = torch.Generator().manual_seed(0)
images = (torch.rand(B, 3, 32, 128, generator=g) * 2 - 1).to(dev)
rng = np.random.RandomState(0)
lens = torch.from_numpy(rng.randint(3, 26, size=B)); tg = torch.from_numpy(rng.randint(0, 94, size=(B, 25)))
for b in range(B): tg[b, int(lens[b]) - 1] = 94; tg[b, int(lens[b]):] = 95
crit, scaler = SeqCrossEntropyLoss(), NativeScalerWithGradNormCount()
def step():
    opt.zero_grad()
    loss = crit(m((images, tg, lens))[0], tg, lens)
    return loss, scaler(loss, opt, clip_grad=None, parameters=None)
for _ in range(3): loss, gn = step()
torch.cuda.synchronize(); t = time.perf_counter(); n = 10
for _ in range(n): loss, gn = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
print(f"AttnRecModel training step B={B}: {dt*1e3:.1f} ms = {B/dt:.0f} images/s  (loss {loss.item():.3f}, grad norm {gn.item():.3f})")
m.eval()
for _ in range(2): p = m((images, None, None))[0]
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): p = m((images, None, None))[0]
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print(f"AttnRecModel greedy sample B={B}: {dt*1e3:.1f} ms = {B/dt:.0f} images/s")
