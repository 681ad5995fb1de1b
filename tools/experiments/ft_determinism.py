"""Is the fine-tune step bit-reproducible run to run (same state, same mask keys)?  With / without the side stream."""
import os, sys, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dig_oracle as O, decode_oracle as D
from dig_amd.finetune import RecModelTrain, SeqCrossEntropyLoss, LayerDecayValueAssigner, create_optimizer
from dig_amd.utils import NativeScalerWithGradNormCount
c, ecfg = D.DecoderConfig(**D.TINY), O.DiGConfig(**O.TINY)
P = {**D.det_encoder_state(ecfg, 32), **D.det_decoder_state(c, 31)}
B = 6
images = O.synthetic_batch(B, ecfg, 555)[0].to("cuda:0")
rng = np.random.RandomState(9)
lens = torch.from_numpy(rng.randint(1, c.max_seq_len + 1, size=B)); targets = torch.from_numpy(rng.randint(0, 94, size=(B, c.max_seq_len)))
for b in range(B): targets[b, int(lens[b]) - 1] = 94; targets[b, int(lens[b]):] = 95
def run(overlap, drop, steps=3):
    kw = dict(decoder_dropout=0.1, drop_rate=0.1, attn_drop_rate=0.1, drop_path_rate=0.1) if drop else dict(decoder_dropout=0.0)
    m = RecModelTrain(embed_dim=ecfg.embed_dim, depth=ecfg.depth, num_heads=ecfg.heads, n_layers=c.n_layers, d_model=c.d_model, n_head=c.n_head,
                      d_k=c.d_k, d_inner=c.d_inner, nb_classes=c.num_classes, max_len=c.max_seq_len, drop_seed=17, **kw)
    m.load_state_dict(P); m.to("cuda:0"); m.train(); m.overlap_streams = overlap
    nl = m.get_num_layers(); asg = LayerDecayValueAssigner([0.75 ** (nl + 1 - i) for i in range(nl + 2)])
    args = types.SimpleNamespace(opt="adamw", lr=1e-3, weight_decay=0.05, opt_eps=1e-8, opt_betas=None)
    opt = create_optimizer(args, m, get_num_layer=asg.get_layer_id, get_layer_scale=asg.get_scale)
    for g in opt.param_groups: g["lr"] = args.lr * g["lr_scale"]
    out = []
    for _ in range(steps):
        opt.zero_grad()
        loss = SeqCrossEntropyLoss()(m((images, targets, lens))[0], targets, lens)
        NativeScalerWithGradNormCount()(loss, opt, clip_grad=1.0, parameters=None)
        out.append((loss.item(), m.flat_grads.clone(), m.flat_params.clone()))
    return out, m
for overlap in (False, True):
    for drop in (False, True):
        a, m = run(overlap, drop); b, _ = run(overlap, drop)
        for s, (x, y) in enumerate(zip(a, b)):
            eg, ep = torch.equal(x[1], y[1]), torch.equal(x[2], y[2])
            msg = ""
            if not eg:
                d = (x[1] - y[1]).abs()
                bad = [n for n in m._offsets if not torch.equal(m._view(x[1], n), m._view(y[1], n))]
                msg = f" max|dg| {d.max().item():.3e}; differing: {bad[:6]} ({len(bad)})"
            print(f"overlap={overlap} drop={drop} step {s}: loss equal {x[0] == y[0]}, grads equal {eg}, params equal {ep}{msg}")
