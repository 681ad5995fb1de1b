"""Throughput of the fine-tune training step (row N1) at the README configuration (README.md:92-118): simmim_vit_small_patch4_32x128 +
tf_decoder, 97 classes, max_len 25, batch 256, --drop 0.1 --attn_drop_rate 0.1 --drop_path 0.1 (decoder dropout 0.1), AdamW with layer
decay 0.75; random weights and labels.  `--no-drop`: every rate 0 (the deterministic step)."""
import os, sys, time, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd.finetune import RecModelTrain, SeqCrossEntropyLoss, LayerDecayValueAssigner, create_optimizer
from dig_amd.utils import NativeScalerWithGradNormCount
dev = torch.device("cuda:0")
args = types.SimpleNamespace(model="simmim_vit_small_patch4_32x128", decoder_name="tf_decoder", nb_classes=97, max_len=25, drop=0.0 if "--no-drop" in sys.argv else 0.1,
                             attn_drop_rate=0.0 if "--no-drop" in sys.argv else 0.1, drop_path=0.0 if "--no-drop" in sys.argv else 0.1, opt="adamw", lr=1e-4, weight_decay=0.05, opt_eps=1e-8, opt_betas=[0.9, 0.999])
m = RecModelTrain(args, decoder_dropout=0.0 if "--no-drop" in sys.argv else 0.1)
g = torch.Generator().manual_seed(0)
sd = {}
for k, s in m.param_shapes().items():
    if "norm" in k and k.endswith("weight") and len(s) == 1:
        sd[k] = torch.ones(s)
    elif k.endswith("bias") or len(s) == 1:
        sd[k] = torch.zeros(s)
    else:
        sd[k] = torch.randn(s, generator=g) * (0.5 if "emb" in k else 1.0 / (s[-1] ** 0.5))
sd["encoder.mask_token"] = torch.zeros(1, 1, 384)
m.load_state_dict(sd); m.to(dev); m.train()
nl = m.get_num_layers()
asg = LayerDecayValueAssigner([0.75 ** (nl + 1 - i) for i in range(nl + 2)])
opt = create_optimizer(args, m, get_num_layer=asg.get_layer_id, get_layer_scale=asg.get_scale)
for grp in opt.param_groups: grp["lr"] = args.lr * grp["lr_scale"]
B = 256
images = (torch.rand(B, 3, 32, 128, generator=g) * 2 - 1).to(dev)
rng = np.random.RandomState(0)
lens = torch.from_numpy(rng.randint(3, 26, size=B)); tg = torch.from_numpy(rng.randint(0, 94, size=(B, 25)))
for b in range(B): tg[b, int(lens[b]) - 1] = 94; tg[b, int(lens[b]):] = 95
tg, lens = tg.to(dev), lens.to(dev)
crit, scaler = SeqCrossEntropyLoss(), NativeScalerWithGradNormCount()
def step():
    opt.zero_grad()
    loss = crit(m((images, tg, lens))[0], tg, lens)
    gn = scaler(loss, opt, clip_grad=None, parameters=None)
    return loss, gn
for _ in range(3): loss, gn = step()
torch.cuda.synchronize(); t = time.perf_counter(); n = 10
for _ in range(n): loss, gn = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
flops = B * (3 * 12.089e9)
print(f"fine-tune step ({'no dropout' if '--no-drop' in sys.argv else 'README drop rates'}) B={B}: {dt*1e3:.1f} ms = {B/dt:.0f} images/s  (loss {loss.item():.3f}, grad norm {gn.item():.3f}; encoder fwd+bwd alone = {flops/1e12:.1f} TFLOP -> {flops/dt/1e12:.0f} TFLOP/s)")
