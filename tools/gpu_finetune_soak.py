"""End-to-end sanity of the fine-tune engines: overfit one fixed synthetic batch (B = 64) for N steps with the README regularisers on --
the loss must fall, stay finite, and the memory footprint must stay flat.  Both decoders."""
import os, sys, time, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd.finetune import RecModelTrain, SeqCrossEntropyLoss, LayerDecayValueAssigner, create_optimizer
from dig_amd.attn_recognizer import AttnRecModelTrain
from dig_amd.utils import NativeScalerWithGradNormCount
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
B = 64
g = torch.Generator().manual_seed(1)
images = (torch.rand(B, 3, 32, 128, generator=g) * 2 - 1).to(dev)
rng = np.random.RandomState(1)
lens = torch.from_numpy(rng.randint(3, 12, size=B)); tg = torch.from_numpy(rng.randint(0, 94, size=(B, 25)))
for b in range(B): tg[b, int(lens[b]) - 1] = 94; tg[b, int(lens[b]):] = 95
for name, cls, kw in (("tf_decoder", RecModelTrain, dict(decoder_name="tf_decoder")), ("gru attention head", AttnRecModelTrain, {})):
    torch.manual_seed(0)
    args = types.SimpleNamespace(model="simmim_vit_small_patch4_32x128", nb_classes=97, max_len=25, drop=0.1, attn_drop_rate=0.1, drop_path=0.1,
                                 opt="adamw", lr=1e-3, weight_decay=0.05, opt_eps=1e-8, opt_betas=[0.9, 0.999], **kw)
    m = cls(args); m.to(dev); m.train()
    nl = m.get_num_layers(); asg = LayerDecayValueAssigner([0.75 ** (nl + 1 - i) for i in range(nl + 2)])
    opt = create_optimizer(args, m, get_num_layer=asg.get_layer_id, get_layer_scale=asg.get_scale)
    crit, scaler = SeqCrossEntropyLoss(), NativeScalerWithGradNormCount()
    losses, mem = [], []
    for it in range(steps):
        lr = args.lr * min(1.0, (it + 1) / 20)
        for grp in opt.param_groups: grp["lr"] = lr * grp["lr_scale"]
        opt.zero_grad()
        loss = crit(m((images, tg, lens))[0], tg, lens)
        scaler(loss, opt, clip_grad=5.0, parameters=None)
        if it % 10 == 0 or it == steps - 1:
            losses.append(loss.item()); mem.append(torch.cuda.max_memory_allocated() / 2 ** 20)
    m.drop_rate = m.attn_drop_rate = 0.0; m.dpr = [0.0] * m.depth; m.decoder_dropout = 0.0
    tf_pred = m((images, tg, lens))[0].detach().argmax(-1).cpu()                # teacher-forced, regularisers off
    m.eval()
    pred = m((images, None, None))[0].argmax(-1).cpu()
    valid = torch.arange(25)[None, :] < lens[:, None]
    acc = (pred[valid] == tg[valid]).float().mean().item()
    tf_acc = (tf_pred[valid] == tg[valid]).float().mean().item()
    ok = all(np.isfinite(losses)) and losses[-1] < 0.35 * losses[0] and mem[-1] <= mem[2] * 1.02
    print(f"{name}: loss {losses[0]:.3f} -> {losses[len(losses)//2]:.3f} -> {losses[-1]:.3f} over {steps} steps; token accuracy on the trained batch: teacher-forced "
          f"{tf_acc:.3f}, greedy {acc:.3f}; peak memory {mem[2]:.0f} -> {mem[-1]:.0f} MiB; {'OK' if ok else 'FAIL'}")
