"""Golden vectors for the GRU attention recognition head from the UNMODIFIED reference: the reference encoder and
`AttentionRecognitionHead` wired as `AttnRecModel.forward` does (models/model_builder.py:67-72), train mode (teacher forcing) with the
reference SeqCrossEntropyLoss, and eval mode (greedy `sample`).  Asserts oracle/attn_decoder_oracle.py == reference, writes
tests/golden/attn_decoder_tiny.npz.

    python oracle/ref_harness/gen_attn_decoder_golden.py        # needs /root/reference (build container only)"""
import os
import sys
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
import attn_decoder_oracle as A
import refenv

GOLD = os.path.join(ROOT, "tests", "golden")


def sample_index(numel, k=8):
    if numel <= k:
        return np.arange(numel)
    return (np.arange(k, dtype=np.int64) * 2654435761 + 12345) % numel


class TinyAttnRec(nn.Module):
    """AttnRecModel (models/model_builder.py:40-72) with the reference's own sub-modules at test widths."""

    def __init__(self, enc, dec):
        super().__init__()
        self.encoder, self.decoder = enc, dec

    def forward(self, x):
        x, tgt, tgt_lens = x
        dec_output, _ = self.decoder((self.encoder(x), tgt, tgt_lens))
        return dec_output, None, None, None


def main():
    refenv.setup()
    torch.manual_seed(0)
    import importlib.util
    import modeling_pretrain_vit as V
    spec = importlib.util.spec_from_file_location("ref_attn_decoder", os.path.join(refenv.REF, "models", "attn_decoder.py"))
    ref_ad = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref_ad)        # (models/__init__ pulls unrelated deps)
    spec = importlib.util.spec_from_file_location("ref_seq_ce", os.path.join(refenv.REF, "loss", "seqCrossEntropyLoss.py"))
    ref_ce = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref_ce)
    ecfg = O.DiGConfig(**O.TINY)
    c = A.AttnDecConfig(**{**A.TINY, "in_planes": ecfg.embed_dim})
    enc = V.PretrainVisionTransformerEncoder(img_size=(32, 128), patch_size=4, embed_dim=ecfg.embed_dim, depth=ecfg.depth,
                                             num_heads=ecfg.heads, mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                             num_classes=0, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0)
    dec = ref_ad.AttentionRecognitionHead(num_classes=c.num_classes, in_planes=c.in_planes, sDim=c.sDim, attDim=c.attDim, max_len_labels=c.max_len)
    model = TinyAttnRec(enc, dec).train()
    P = {**D.det_encoder_state(ecfg, 42), **A.det_state(c, 43)}
    sd = model.state_dict()
    assert [k for k in sd if k.startswith("decoder.")] == list(A.param_shapes(c)), [k for k in sd if k.startswith("decoder.")]
    for k, v in P.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        sd[k].copy_(v)
    B = 6
    images = O.synthetic_batch(B, ecfg, 777)[0]
    rng = np.random.RandomState(11)
    lens = torch.from_numpy(rng.randint(1, c.max_len, size=B))                   # max(lengths) < max_len: the zero-padded tail is exercised
    targets = torch.from_numpy(rng.randint(0, 94, size=(B, c.max_len)))
    for b in range(B):
        targets[b, int(lens[b]) - 1] = 94
        targets[b, int(lens[b]):] = 95
    outputs, _, _, _ = model((images, targets, lens))
    loss = ref_ce.SeqCrossEntropyLoss()(outputs, targets, lens)
    loss.backward()
    ref_grads = {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in model.named_parameters()}
    o_loss, o_grads, o_logits = A.loss_and_grads(P, ecfg, c, images, targets, lens)
    assert abs(o_loss - loss.item()) < 1e-5 * abs(loss.item()), (o_loss, loss.item())
    assert (o_logits - outputs.detach()).abs().max() < 2e-5
    worst = 0.0
    gmax = max(g.abs().max().item() for g in ref_grads.values() if g is not None)
    for n, g in ref_grads.items():
        if g is None:
            assert n == "encoder.mask_token", n
            continue
        # (wEmbed.bias shifts every score of a softmax row: its true gradient is 0 and both sides hold round-off)
        e = (o_grads[n] - g).abs().max().item() / max(g.abs().max().item(), 1e-4 * gmax)
        worst = max(worst, e)
        assert e < 2e-3, (n, e)
    model.eval()
    with torch.no_grad():
        probs, _, _, _ = model((images, None, None))
        o_probs = A.head_sample(P, c, D.encoder_features(P, ecfg, images))
    assert (o_probs - probs).abs().max() < 2e-5 and torch.equal(o_probs.argmax(-1), probs.argmax(-1))
    srt = probs.sort(-1, descending=True)[0]
    print(f"GRU attention head: oracle == reference (loss {o_loss:.6f}, worst gradient rel-to-max err {worst:.2e}; greedy sample equal, "
          f"min top-2 margin {float((srt[..., 0] - srt[..., 1]).min()):.3e})")
    names = [n for n in P if ref_grads.get(n) is not None]
    np.savez_compressed(os.path.join(GOLD, "attn_decoder_tiny.npz"), seed_enc=42, seed_dec=43, B=B, batch_seed=777, targets=targets.numpy(),
                        lens=lens.numpy(), loss=np.float64(loss.item()), logits=outputs.detach().numpy(), sample_probs=probs.numpy(),
                        grad_names=np.array(names), grad_norms=np.array([ref_grads[n].double().norm().item() for n in names]),
                        grad_samples=np.stack([np.resize(ref_grads[n].reshape(-1)[sample_index(ref_grads[n].numel())].numpy(), 8) for n in names]))
    print("wrote tests/golden/attn_decoder_tiny.npz")


if __name__ == "__main__":
    main()
