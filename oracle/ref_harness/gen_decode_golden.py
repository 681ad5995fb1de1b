"""Golden vectors for the recognition forward + greedy decode (row N4) from the UNMODIFIED reference classes:
models/decoder.py TFDecoder.forward_test, modeling_pretrain_vit.py PretrainVisionTransformerEncoder, and the real RecModel
(models/model_builder.py) for the state_dict key order.  Asserts oracle/decode_oracle.py == reference, writes
tests/golden/decode_tiny.npz (decoder alone), tests/golden/decode_beam_tiny.npz (TFDecoder.beam_search, width 5) and
tests/golden/recognize_tiny.npz (encoder + linear_norm + decoder).

    python oracle/ref_harness/gen_decode_golden.py        # needs /root/reference (build container only)"""
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
import refenv

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    refenv.setup()
    torch.manual_seed(0)
    from models.decoder import TFDecoder
    import modeling_pretrain_vit as V
    c = D.DecoderConfig(**D.TINY)
    P = D.det_decoder_state(c, 21)
    dec = TFDecoder(n_layers=c.n_layers, d_embedding=c.d_model, n_head=c.n_head, d_k=c.d_k, d_v=c.d_k, d_model=c.d_model, d_inner=c.d_inner,
                    num_classes=c.num_classes, max_seq_len=c.max_seq_len).eval()
    sd = dec.state_dict()
    for k, v in P.items():
        if k.startswith("decoder."):
            sd[k[len("decoder."):]].copy_(v)
    assert torch.equal(sd["position_enc.position_table"][0], D.position_table(c.n_position, c.d_model))
    B, Nm = 5, 96
    mem = O.det_tensor("memory", (B, Nm, c.d_model), 4, 1.0)
    with torch.no_grad():
        ref_out, ref_maps = dec.forward_test(None, mem, None, None, None)
        out, maps, toks = D.greedy_decode(P, c, mem)
        out2, maps2, toks2 = D.greedy_decode_cached(P, c, mem)
    assert torch.equal(ref_out.argmax(-1), toks) and torch.equal(toks, toks2)
    assert (ref_out - out).abs().max() < 2e-6 and (ref_maps - maps).abs().max() < 2e-6, ((ref_out - out).abs().max(), (ref_maps - maps).abs().max())
    assert (ref_out - out2).abs().max() < 2e-6 and (ref_maps - maps2).abs().max() < 2e-6
    np.savez_compressed(os.path.join(GOLD, "decode_tiny.npz"), seed=21, B=B, Nm=Nm, probs=ref_out.numpy(), maps=ref_maps.numpy(),
                        tokens=toks.numpy(), cfg_keys=np.array(list(D.TINY.keys())), cfg_vals=np.array(list(D.TINY.values())))
    print("decoder: oracle == reference forward_test; tokens", toks[0].tolist())

    # ---- beam search (decoder.py:254-370) from the unmodified TFDecoder, beam width 5: (a) the random-weight decoder (EOS is rare),
    # (b) the same decoder with the classifier bias of EOS (94) raised, so that hypotheses end early and the back-tracking's
    # replacement of live beams by ended ones (decoder.py:333-353) is exercised
    beam = {}
    for tag, boost in (("plain", 0.0), ("eos", 3.0)):
        Pb = {k: v.clone() for k, v in P.items()}
        Pb["decoder.classifier.bias"][94] += boost
        sd["classifier.bias"].copy_(Pb["decoder.classifier.bias"])
        with torch.no_grad():
            ref_ids, ref_ones = dec.beam_search(None, mem, None, None, None, 5, eos=94)
            mine = D.beam_search(Pb, c, mem, 5, eos=94)
        assert torch.equal(ref_ids, mine), (tag, ref_ids, mine)
        assert torch.equal(ref_ones, torch.ones_like(ref_ids))
        beam[tag] = ref_ids.numpy()
        print(f"beam search ({tag}): oracle == reference; sample 0", ref_ids[0].tolist(), " EOS positions:", int((ref_ids == 94).sum()))
    sd["classifier.bias"].copy_(P["decoder.classifier.bias"])
    np.savez_compressed(os.path.join(GOLD, "decode_beam_tiny.npz"), seed=21, B=B, Nm=Nm, beam_width=5, eos=94, eos_bias_boost=3.0,
                        tokens_plain=beam["plain"], tokens_eos=beam["eos"], cfg_keys=np.array(list(D.TINY.keys())),
                        cfg_vals=np.array(list(D.TINY.values())))

    # ---- encoder + linear_norm + decoder, tiny widths, reference classes wired as RecModel.forward does in eval mode
    ecfg = O.DiGConfig(**O.TINY)                                                     # embed 128, depth 2, heads 2
    enc = V.PretrainVisionTransformerEncoder(img_size=(32, 128), patch_size=4, embed_dim=ecfg.embed_dim, depth=ecfg.depth,
                                             num_heads=ecfg.heads, mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                             num_classes=0).eval()
    EP = D.det_encoder_state(ecfg, 22)
    esd = enc.state_dict()
    assert ["encoder." + k for k in esd.keys()] == list(EP.keys()), (list(esd.keys())[:5], list(EP.keys())[:5])
    for k, v in EP.items():
        esd[k[len("encoder."):]].copy_(v)
    ln = nn.Sequential(nn.Linear(ecfg.embed_dim, c.d_model), nn.LayerNorm(c.d_model)).eval()
    ln[0].weight.data.copy_(P["linear_norm.0.weight"]); ln[0].bias.data.copy_(P["linear_norm.0.bias"])
    ln[1].weight.data.copy_(P["linear_norm.1.weight"]); ln[1].bias.data.copy_(P["linear_norm.1.bias"])
    images = O.synthetic_batch(3, ecfg, 77)[0]
    with torch.no_grad():
        dec_in = ln(enc(images))
        ref_out, ref_maps = dec(dec_in, dec_in, targets=None, tgt_lens=None, train_mode=False, cls_query_attn_maps=None, trg_word_emb=None,
                                beam_width=0)
        out, maps, toks = D.recognize({**EP, **P}, ecfg, c, images, cached=True)
    assert torch.equal(ref_out.argmax(-1), toks), (ref_out.argmax(-1), toks)
    assert (ref_out - out).abs().max() < 5e-6 and (ref_maps - maps).abs().max() < 5e-6, ((ref_out - out).abs().max(), (ref_maps - maps).abs().max())
    np.savez_compressed(os.path.join(GOLD, "recognize_tiny.npz"), seed_dec=21, seed_enc=22, B=3, batch_seed=77, probs=ref_out.numpy(),
                        maps=ref_maps.numpy(), tokens=toks.numpy(), memory_norm=np.float64(dec_in.double().norm().item()))
    print("recognizer: oracle == reference encoder + linear_norm + TFDecoder; tokens", toks[0].tolist())

    # ---- the real RecModel at full size: key order / shapes of its state_dict (what a fine-tune checkpoint holds)
    from models.model_builder import RecModel
    args = types.SimpleNamespace(model="simmim_vit_small_patch4_32x128", decoder_name="tf_decoder", nb_classes=97, max_len=25, drop=0.0,
                                 drop_path=0.0, attn_drop_rate=0.0, use_mean_pooling=False, init_scale=0.001, use_seq_cls_token=False,
                                 use_1d_attdec=False, text_cond_vis=False, beam_width=0)
    rec = RecModel(args).eval()
    keys = [k for k in rec.state_dict().keys()]
    full = D.DecoderConfig()
    want = list(D.finetune_encoder_shapes(O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128")).keys())
    dshape = D.decoder_param_shapes(full)
    mine = want + [k for k in dshape.keys() if k.startswith("decoder.")] + [k for k in dshape.keys() if k.startswith("linear_norm.")]
    ref_keys = [k for k in keys if not k.endswith("position_table") and not k.startswith("patch_embed.")]
    assert sorted(ref_keys) == sorted(mine), (set(ref_keys) ^ set(mine))
    sdr = rec.state_dict()
    for k, s in {**D.finetune_encoder_shapes(O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128")), **dshape}.items():
        assert tuple(sdr[k].shape) == tuple(s), (k, sdr[k].shape, s)
    np.savez_compressed(os.path.join(GOLD, "recmodel_keys.npz"), keys=np.array(keys))
    print("RecModel state_dict:", len(keys), "keys; shapes match the oracle inventory")


if __name__ == "__main__":
    main()
