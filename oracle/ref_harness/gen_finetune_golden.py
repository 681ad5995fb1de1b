"""Golden vectors for the fine-tune training step (row N1, drop rates 0) from the UNMODIFIED reference: the reference encoder /
TFDecoder classes wired as RecModel.forward does in train mode, the two lines of engine_for_finetuning.train_class_batch, the
reference SeqCrossEntropyLoss, and
optim_factory.create_optimizer with LayerDecayValueAssigner (run_class_finetuning.py:471-520).  Asserts oracle/finetune_oracle.py
== reference (loss, logits, every gradient, parameters after one AdamW step given the same gradients), writes
tests/golden/finetune_tiny.npz.

    python oracle/ref_harness/gen_finetune_golden.py        # needs /root/reference (build container only)
    python oracle/ref_harness/gen_finetune_golden.py --drop # the same step with dropout / drop-path -> finetune_tiny_drop.npz

--drop: the reference modules are built with drop_rate / attn_drop_rate / drop_path_rate > 0 and the decoder's default
dropout 0.1, in train mode; every nn.Dropout / DropPath INSTANCE of the unmodified model has its forward replaced by the keyed
mask of its site (finetune_oracle.DropOracle), looked up by module name -- the rest of the reference code (where each mask acts,
its scale, the order of operations) runs as is.  Asserts oracle == reference and writes the fixture."""
import os
import sys
import types
from functools import partial

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)
import dig_oracle as O
import decode_oracle as D
import finetune_oracle as F
import refenv

GOLD = os.path.join(ROOT, "tests", "golden")


def sample_index(numel, k=8):
    if numel <= k:
        return np.arange(numel)
    return (np.arange(k, dtype=np.int64) * 2654435761 + 12345) % numel


class TinyRec(nn.Module):
    """RecModel (models/model_builder.py:74-160) with the reference's own sub-modules at test widths."""

    def __init__(self, enc, ln, dec, use_1d_attdec=False):
        super().__init__()
        self.encoder, self.decoder, self.linear_norm = enc, dec, ln
        self.use_1d_attdec = use_1d_attdec

    def forward(self, x):
        x, tgt, tgt_lens = x
        enc_x = self.encoder(x)
        if not self.training:
            tgt = tgt_lens = None                                               # model_builder.py:138-140
        if self.use_1d_attdec:                                                  # model_builder.py:145-148
            B, N, C = enc_x.shape
            enc_x = enc_x.view(B, *self.encoder.patch_embed.patch_shape, C).mean(1)
        dec_in = self.linear_norm(enc_x)
        out, maps = self.decoder(dec_in, dec_in, targets=tgt, tgt_lens=tgt_lens, train_mode=self.training, cls_query_attn_maps=None,
                                 trg_word_emb=None, beam_width=0)
        return out, None, None, maps

    def get_num_layers(self):
        return self.encoder.get_num_layers()

    def no_weight_decay(self):
        return {'encoder.' + i for i in self.encoder.no_weight_decay()}


def main():
    refenv.setup()
    torch.manual_seed(0)
    from models.decoder import TFDecoder
    import modeling_pretrain_vit as V
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_seq_ce", os.path.join(refenv.REF, "loss", "seqCrossEntropyLoss.py"))
    ref_ce = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref_ce)      # (the package __init__ pulls unrelated deps)
    SeqCrossEntropyLoss = ref_ce.SeqCrossEntropyLoss
    import optim_factory
    c = D.DecoderConfig(**D.TINY)
    ecfg = O.DiGConfig(**O.TINY)
    enc = V.PretrainVisionTransformerEncoder(img_size=(32, 128), patch_size=4, embed_dim=ecfg.embed_dim, depth=ecfg.depth,
                                             num_heads=ecfg.heads, mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                             num_classes=0, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0)
    dec = TFDecoder(n_layers=c.n_layers, d_embedding=c.d_model, n_head=c.n_head, d_k=c.d_k, d_v=c.d_k, d_model=c.d_model, d_inner=c.d_inner,
                    num_classes=c.num_classes, max_seq_len=c.max_seq_len, dropout=0.0)
    ln = nn.Sequential(nn.Linear(ecfg.embed_dim, c.d_model), nn.LayerNorm(c.d_model))
    model = TinyRec(enc, ln, dec).train()
    P = {**D.det_encoder_state(ecfg, 32), **D.det_decoder_state(c, 31)}
    sd = model.state_dict()
    for k, v in P.items():
        sd[k].copy_(v)
    B = 6
    images = O.synthetic_batch(B, ecfg, 555)[0]
    rng = np.random.RandomState(9)
    lens = torch.from_numpy(rng.randint(1, c.max_seq_len + 1, size=B))
    targets = torch.from_numpy(rng.randint(0, 94, size=(B, c.max_seq_len)))
    for b in range(B):
        targets[b, int(lens[b]) - 1] = 94
        targets[b, int(lens[b]):] = 95
    args = types.SimpleNamespace(use_seq_cls_token=False, opt='adamw', lr=1e-3, weight_decay=0.05, opt_eps=1e-8, opt_betas=None, momentum=0.9)
    layer_decay = 0.75
    num_layers = model.get_num_layers()
    assigner = optim_factory.LayerDecayValueAssigner(list(layer_decay ** (num_layers + 1 - i) for i in range(num_layers + 2)))
    opt = optim_factory.create_optimizer(args, model, skip_list=model.no_weight_decay(), get_num_layer=assigner.get_layer_id,
                                         get_layer_scale=assigner.get_scale)
    for g in opt.param_groups:
        g["lr"] = args.lr * g["lr_scale"]                                       # what train_one_epoch does each step (:73-78)
    # engine_for_finetuning.train_class_batch (:26-46; the module itself needs torchvision / editdistance, absent here):
    outputs, _, _, _ = model((images, targets, lens))
    loss = SeqCrossEntropyLoss()(outputs, targets, lens)
    opt.zero_grad()
    loss.backward()
    ref_grads = {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in model.named_parameters()}
    opt.step()
    # ---- oracle
    o_loss, o_grads, o_logits = F.loss_and_grads(P, ecfg, c, images, targets, lens)
    assert abs(o_loss - loss.item()) < 1e-5 * abs(loss.item()), (o_loss, loss.item())
    assert (o_logits - outputs.detach()).abs().max() < 2e-5
    worst = 0.0
    for n, g in ref_grads.items():
        if g is None:
            assert n == "encoder.mask_token", n
            continue
        e = (o_grads[n] - g).abs().max().item() / (g.abs().max().item() + 1e-12)
        worst = max(worst, e)
        assert e < 2e-3, (n, e)
    groups = F.param_groups(P, num_layers, layer_decay, args.weight_decay)
    Pn = {k: v.clone() for k, v in P.items()}
    F.adamw_step(Pn, {k: (v if v is not None else torch.zeros_like(P[k])) for k, v in ref_grads.items()}, {}, 1, args.lr, groups)
    sd2 = model.state_dict()
    for n in Pn:
        assert (Pn[n] - sd2[n]).abs().max() < 1e-6, n
    print(f"fine-tune step: oracle == reference (loss {o_loss:.6f}, worst gradient rel-to-max err {worst:.2e}); AdamW + layer decay restatement exact")
    # optimizer checkpoint layout (torch per-parameter indices in group order): which parameter name sits at which state index
    id2name = {id(p): n for n, p in model.named_parameters()}
    osd = opt.state_dict()
    index_names = [id2name[id(p)] for g_ in opt.param_groups for p in g_["params"]]
    stateless = [index_names[i] for i in range(len(index_names)) if i not in osd["state"]]
    assert stateless == ["encoder.mask_token"], stateless                        # in the list (requires_grad) but never given a gradient
    group_sizes = [len(g_["params"]) for g_ in osd["param_groups"]]
    group_lr_scale = [float(g_["lr_scale"]) for g_ in osd["param_groups"]]
    group_wd_list = [float(g_["weight_decay"]) for g_ in osd["param_groups"]]
    names = [n for n in P if ref_grads.get(n) is not None]
    np.savez_compressed(os.path.join(GOLD, "finetune_tiny.npz"), opt_index_names=np.array(index_names), opt_group_sizes=np.array(group_sizes),
                        opt_group_lr_scale=np.array(group_lr_scale), opt_group_wd=np.array(group_wd_list), seed_enc=32, seed_dec=31, B=B, batch_seed=555, targets=targets.numpy(),
                        lens=lens.numpy(), loss=np.float64(loss.item()), logits=outputs.detach().numpy(), lr=args.lr, weight_decay=args.weight_decay,
                        layer_decay=layer_decay, grad_names=np.array(names),
                        grad_norms=np.array([ref_grads[n].double().norm().item() for n in names]),
                        grad_samples=np.stack([np.resize(ref_grads[n].reshape(-1)[sample_index(ref_grads[n].numel())].numpy(), 8) for n in names]),
                        param_norms=np.array([sd2[n].double().norm().item() for n in names]),
                        group_scale=np.array([groups[n][0] for n in names]), group_wd=np.array([groups[n][1] for n in names]))
    print("wrote tests/golden/finetune_tiny.npz")


def main_1d():
    """--use_1d_attdec (run_class_finetuning.py:89, model_builder.py:145-148): train step and greedy evaluation with the decoder on the
    32 column means of the token grid."""
    refenv.setup()
    torch.manual_seed(0)
    from models.decoder import TFDecoder
    import modeling_pretrain_vit as V
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_seq_ce", os.path.join(refenv.REF, "loss", "seqCrossEntropyLoss.py"))
    ref_ce = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref_ce)
    c = D.DecoderConfig(**D.TINY)
    ecfg = O.DiGConfig(**O.TINY)
    enc = V.PretrainVisionTransformerEncoder(img_size=(32, 128), patch_size=4, embed_dim=ecfg.embed_dim, depth=ecfg.depth,
                                             num_heads=ecfg.heads, mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                             num_classes=0, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0)
    dec = TFDecoder(n_layers=c.n_layers, d_embedding=c.d_model, n_head=c.n_head, d_k=c.d_k, d_v=c.d_k, d_model=c.d_model, d_inner=c.d_inner,
                    num_classes=c.num_classes, max_seq_len=c.max_seq_len, dropout=0.0)
    ln = nn.Sequential(nn.Linear(ecfg.embed_dim, c.d_model), nn.LayerNorm(c.d_model))
    model = TinyRec(enc, ln, dec, use_1d_attdec=True).train()
    P = {**D.det_encoder_state(ecfg, 42), **D.det_decoder_state(c, 41)}
    sd = model.state_dict()
    for k, v in P.items():
        sd[k].copy_(v)
    B = 5
    images = O.synthetic_batch(B, ecfg, 565)[0]
    rng = np.random.RandomState(19)
    lens = torch.from_numpy(rng.randint(1, c.max_seq_len + 1, size=B))
    targets = torch.from_numpy(rng.randint(0, 94, size=(B, c.max_seq_len)))
    for b in range(B):
        targets[b, int(lens[b]) - 1] = 94
        targets[b, int(lens[b]):] = 95
    outputs, _, _, _ = model((images, targets, lens))
    loss = ref_ce.SeqCrossEntropyLoss()(outputs, targets, lens)
    loss.backward()
    ref_grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    o_loss, o_grads, o_logits = F.loss_and_grads(P, ecfg, c, images, targets, lens, use_1d_attdec=True)
    assert abs(o_loss - loss.item()) < 1e-5 * abs(loss.item()), (o_loss, loss.item())
    assert (o_logits - outputs.detach()).abs().max() < 2e-5
    worst = max((o_grads[n] - g).abs().max().item() / (g.abs().max().item() + 1e-12) for n, g in ref_grads.items())
    assert worst < 2e-3, worst
    model.eval()
    with torch.no_grad():
        probs, _, _, maps = model((images, None, None))
    o_probs, o_maps, o_tok = D.recognize(P, ecfg, c, images, cached=True, use_1d_attdec=True)
    assert (o_probs - probs).abs().max() < 2e-5 and torch.equal(o_tok, probs.argmax(-1))
    print(f"1-D decoder: oracle == reference (train loss {o_loss:.6f}, worst gradient rel-to-max err {worst:.2e}; greedy eval probabilities equal)")
    names = sorted(ref_grads)
    np.savez_compressed(os.path.join(GOLD, "finetune_tiny_1d.npz"), seed_enc=42, seed_dec=41, B=B, batch_seed=565, targets=targets.numpy(),
                        lens=lens.numpy(), loss=np.float64(loss.item()), logits=outputs.detach().numpy(), grad_names=np.array(names),
                        grad_norms=np.array([ref_grads[n].double().norm().item() for n in names]),
                        eval_probs=probs.numpy(), eval_tokens=probs.argmax(-1).numpy())
    print("wrote tests/golden/finetune_tiny_1d.npz")


def patch_dropouts(model, dr, depth, n_layers):
    """Replace the forward of every stochastic module instance by the site's keyed mask."""
    import modeling_finetune as MF
    counts = {}

    def elem(site, p):
        return lambda x: dr.elem(site, x, p)

    def attn(site, p):
        return lambda x: dr.attn(site, x, p)

    def twice(name, first, second):
        def f(x):
            n = counts.get(name, 0)
            counts[name] = n + 1
            return (first if n % 2 == 0 else second)(x)
        return f

    done = []
    for name, m in model.named_modules():
        parts = name.split(".")
        if isinstance(m, MF.DropPath):                                      # encoder.blocks.i.drop_path: attention branch, then MLP branch
            i = int(parts[2])
            m.forward = twice(name, lambda x, i=i: dr.path(F.enc_site(i, 2), x, dr.dpr[i]), lambda x, i=i: dr.path(F.enc_site(i, 4), x, dr.dpr[i]))
        elif isinstance(m, nn.Dropout):
            if name.startswith("encoder.blocks."):
                i = int(parts[2])
                kind = {"attn.attn_drop": 0, "attn.proj_drop": 1, "mlp.drop": 3}[".".join(parts[3:])]
                m.forward = (attn if kind == 0 else elem)(F.enc_site(i, kind), dr.attn_drop if kind == 0 else dr.drop)
            elif name == "decoder.dropout":
                m.forward = elem(F.DEC_TGT, dr.decoder_dropout)
            elif name.startswith("decoder.layer_stack."):
                i = int(parts[2])
                tail = ".".join(parts[3:])
                if tail == "mlp.dropout":                                   # used twice: after the activation, after w_2
                    m.forward = twice(name, elem(F.dec_site(i, 4), dr.decoder_dropout), elem(F.dec_site(i, 5), dr.decoder_dropout))
                elif tail in ("self_attn.attn_drop", "self_attn.proj_drop", "enc_attn.attn_drop", "enc_attn.proj_drop"):
                    kind = {"self_attn.attn_drop": 0, "self_attn.proj_drop": 1, "enc_attn.attn_drop": 2, "enc_attn.proj_drop": 3}[tail]
                    m.forward = (attn if kind in (0, 2) else elem)(F.dec_site(i, kind), dr.decoder_dropout)
                else:
                    continue                                                # (mlp_order2cls_attn etc.: not on the tf_decoder path)
            else:
                raise AssertionError("unmapped dropout module " + name)
        else:
            continue
        done.append(name)
    return done


def main_drop():
    refenv.setup()
    torch.manual_seed(0)
    from models.decoder import TFDecoder
    import modeling_pretrain_vit as V
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_seq_ce", os.path.join(refenv.REF, "loss", "seqCrossEntropyLoss.py"))
    ref_ce = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref_ce)
    c = D.DecoderConfig(**D.TINY)
    ecfg = O.DiGConfig(**O.TINY)
    rates = dict(drop=0.1, attn_drop=0.1, drop_path=0.2, decoder_dropout=0.1)
    seed, step = 1234, 5
    enc = V.PretrainVisionTransformerEncoder(img_size=(32, 128), patch_size=4, embed_dim=ecfg.embed_dim, depth=ecfg.depth,
                                             num_heads=ecfg.heads, mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                             num_classes=0, drop_rate=rates["drop"], attn_drop_rate=rates["attn_drop"],
                                             drop_path_rate=rates["drop_path"])
    dec = TFDecoder(n_layers=c.n_layers, d_embedding=c.d_model, n_head=c.n_head, d_k=c.d_k, d_v=c.d_k, d_model=c.d_model, d_inner=c.d_inner,
                    num_classes=c.num_classes, max_seq_len=c.max_seq_len)                  # dropout: the class default (0.1), as create_decoder
    ln = nn.Sequential(nn.Linear(ecfg.embed_dim, c.d_model), nn.LayerNorm(c.d_model))
    model = TinyRec(enc, ln, dec).train()
    dr = F.DropOracle(seed, step, depth=ecfg.depth, **rates)
    patched = patch_dropouts(model, dr, ecfg.depth, c.n_layers)
    P = {**D.det_encoder_state(ecfg, 32), **D.det_decoder_state(c, 31)}
    sd = model.state_dict()
    for k, v in P.items():
        sd[k].copy_(v)
    B = 6
    images = O.synthetic_batch(B, ecfg, 556)[0]
    rng = np.random.RandomState(10)
    lens = torch.from_numpy(rng.randint(1, c.max_seq_len + 1, size=B))
    targets = torch.from_numpy(rng.randint(0, 94, size=(B, c.max_seq_len)))
    for b in range(B):
        targets[b, int(lens[b]) - 1] = 94
        targets[b, int(lens[b]):] = 95
    outputs, _, _, _ = model((images, targets, lens))
    loss = ref_ce.SeqCrossEntropyLoss()(outputs, targets, lens)
    loss.backward()
    ref_grads = {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in model.named_parameters()}
    o_loss, o_grads, o_logits = F.loss_and_grads(P, ecfg, c, images, targets, lens, drop=dr)
    assert abs(o_loss - loss.item()) < 1e-5 * abs(loss.item()), (o_loss, loss.item())
    assert (o_logits - outputs.detach()).abs().max() < 2e-5
    worst = 0.0
    for n, g in ref_grads.items():
        if g is None:
            assert n == "encoder.mask_token", n
            continue
        e = (o_grads[n] - g).abs().max().item() / (g.abs().max().item() + 1e-12)
        worst = max(worst, e)
        assert e < 2e-3, (n, e)
    # the masks matter: the deterministic forward gives a different loss
    d_loss = F.loss_and_grads(P, ecfg, c, images, targets, lens)[0]
    assert abs(d_loss - o_loss) > 1e-3 * abs(o_loss)
    print(f"fine-tune step with dropout: oracle == reference under the keyed masks ({len(patched)} stochastic modules patched; loss {o_loss:.6f} "
          f"vs {d_loss:.6f} without dropout; worst gradient rel-to-max err {worst:.2e})")
    names = [n for n in P if ref_grads.get(n) is not None]
    np.savez_compressed(os.path.join(GOLD, "finetune_tiny_drop.npz"), seed_enc=32, seed_dec=31, B=B, batch_seed=556, targets=targets.numpy(),
                        lens=lens.numpy(), loss=np.float64(loss.item()), logits=outputs.detach().numpy(), drop_seed=seed, drop_step=step,
                        rates=np.array([rates["drop"], rates["attn_drop"], rates["drop_path"], rates["decoder_dropout"]]),
                        grad_names=np.array(names), grad_norms=np.array([ref_grads[n].double().norm().item() for n in names]),
                        grad_samples=np.stack([np.resize(ref_grads[n].reshape(-1)[sample_index(ref_grads[n].numel())].numpy(), 8) for n in names]))
    print("wrote tests/golden/finetune_tiny_drop.npz")


def main_smoothing():
    """The reference SeqLabelSmoothingCrossEntropyLoss (run as is) on seeded logits -> tests/golden/seq_ls_loss.npz."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_seq_ls", os.path.join(refenv.REF, "loss", "seqLabelSmoothingCrossEntropyLoss.py"))
    ref = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref)
    out = {}
    for case, (B, T, C, sm, lens) in enumerate([(3, 5, 7, 0.1, [5, 2, 0]), (6, 25, 97, 0.1, [25, 1, 7, 13, 25, 3]), (2, 8, 97, 0.3, [8, 4])]):
        g = torch.Generator().manual_seed(100 + case)
        x = (torch.randn(B, T, C, generator=g) * 2).requires_grad_(True)
        t = torch.randint(0, C, (B, T), generator=g)
        l = torch.tensor(lens)
        loss = ref.SeqLabelSmoothingCrossEntropyLoss(smoothing=sm)(x, t, l)
        loss.backward()
        mine = F.seq_label_smoothing_cross_entropy(x.detach(), t, l, sm)
        assert abs(mine.item() - loss.item()) <= 1e-6 * abs(loss.item()), (mine.item(), loss.item())
        out.update({f"c{case}/shape": np.array([B, T, C]), f"c{case}/smoothing": sm, f"c{case}/lens": l.numpy(), f"c{case}/target": t.numpy(),
                    f"c{case}/logits": x.detach().numpy(), f"c{case}/loss": np.float64(loss.item()), f"c{case}/grad": x.grad.numpy()})
        print(f"case {case}: reference loss {loss.item():.6f} == restatement")
    np.savez_compressed(os.path.join(GOLD, "seq_ls_loss.npz"), **out)
    print("wrote tests/golden/seq_ls_loss.npz")


if __name__ == "__main__":
    if "--smoothing" in sys.argv:
        main_smoothing()
    elif "--1d" in sys.argv:
        main_1d()
    else:
        main_drop() if "--drop" in sys.argv else main()
