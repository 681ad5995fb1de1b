"""Row N4 (recognition forward + greedy decode): the fp32 oracle against fixtures written by the unmodified reference classes
(CPU), and the device path (K/V-cached decode, bf16) against the oracle / fixtures (GPU)."""
import os

import numpy as np
import pytest
import torch

import decode_oracle as D
import dig_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _tiny():
    g = np.load(os.path.join(GOLD, "decode_tiny.npz"))
    c = D.DecoderConfig(**{k: int(v) for k, v in zip(g["cfg_keys"].tolist(), g["cfg_vals"].tolist())})
    return g, c


def test_oracle_decode_matches_reference_fixture():
    g, c = _tiny()
    P = D.det_decoder_state(c, int(g["seed"]))
    mem = O.det_tensor("memory", (int(g["B"]), int(g["Nm"]), c.d_model), 4, 1.0)
    for fn in (D.greedy_decode, D.greedy_decode_cached):                 # literal forward_test, and the K/V-cache form
        probs, maps, toks = fn(P, c, mem)
        assert np.array_equal(toks.numpy(), g["tokens"])
        np.testing.assert_allclose(probs.numpy(), g["probs"], atol=3e-6)
        np.testing.assert_allclose(maps.numpy(), g["maps"], atol=3e-6)


def _beam_fixture():
    g = np.load(os.path.join(GOLD, "decode_beam_tiny.npz"))
    c = D.DecoderConfig(**{k: int(v) for k, v in zip(g["cfg_keys"].tolist(), g["cfg_vals"].tolist())})
    return g, c


def test_oracle_beam_search_matches_reference_fixture():
    """TFDecoder.beam_search of the unmodified reference (oracle/ref_harness/gen_decode_golden.py), beam width 5: the random-weight
    decoder and the one whose EOS bias is raised so that hypotheses end early (back-tracking with replacement)."""
    g, c = _beam_fixture()
    mem = O.det_tensor("memory", (int(g["B"]), int(g["Nm"]), c.d_model), 4, 1.0)
    for tag, boost in (("plain", 0.0), ("eos", float(g["eos_bias_boost"]))):
        P = D.det_decoder_state(c, int(g["seed"]))
        P["decoder.classifier.bias"][int(g["eos"])] += boost
        ids = D.beam_search(P, c, mem, int(g["beam_width"]), eos=int(g["eos"]))
        assert np.array_equal(ids.numpy(), g["tokens_" + tag]), tag
    assert (g["tokens_eos"] == int(g["eos"])).sum() > 0


def test_host_backtracking_matches_oracle():
    """dig_amd.recognizer.beam_backtrack (numpy, the product's host side of beam search) against the oracle's literal restatement of
    decoder.py:311-369 on random decision tables with many EOS events (distinct scores: ties are unspecified in the reference too)."""
    from dig_amd.recognizer import beam_backtrack
    rng = np.random.RandomState(0)
    for B, bw, T, eos in ((3, 5, 8, 94), (1, 2, 25, 7), (6, 4, 12, 3)):
        S = B * bw
        scores = -rng.rand(T, S).astype(np.float32).cumsum(0) - 1e-3 * rng.rand(T, S).astype(np.float32)
        syms = rng.randint(0, 10 if eos < 10 else 97, size=(T, S)).astype(np.int64)
        syms[rng.rand(T, S) < 0.15] = eos
        preds = (rng.randint(0, bw, size=(T, S)) + (np.arange(S) // bw * bw)[None, :]).astype(np.int64)
        want = D.backtrack(torch.from_numpy(scores), torch.from_numpy(preds), torch.from_numpy(syms), B, bw, eos)
        got = beam_backtrack(scores, preds, syms, B, bw, eos)
        assert torch.equal(got, want), (B, bw, T)


def test_oracle_recognizer_matches_reference_fixture():
    g = np.load(os.path.join(GOLD, "recognize_tiny.npz"))
    _, c = _tiny()
    ecfg = O.DiGConfig(**O.TINY)
    P = {**D.det_encoder_state(ecfg, int(g["seed_enc"])), **D.det_decoder_state(c, int(g["seed_dec"]))}
    images = O.synthetic_batch(int(g["B"]), ecfg, int(g["batch_seed"]))[0]
    probs, maps, toks = D.recognize(P, ecfg, c, images)
    assert np.array_equal(toks.numpy(), g["tokens"])
    np.testing.assert_allclose(probs.numpy(), g["probs"], atol=1e-5)
    np.testing.assert_allclose(maps.numpy(), g["maps"], atol=1e-5)


def test_recmodel_state_dict_inventory_matches_reference():
    """Keys / shapes of the real reference RecModel (simmim_vit_small_patch4_32x128 + tf_decoder, 97 classes, max_len 25)."""
    import types
    from dig_amd.recognizer import RecModel
    keys = np.load(os.path.join(GOLD, "recmodel_keys.npz"))["keys"].tolist()
    args = types.SimpleNamespace(model="simmim_vit_small_patch4_32x128", decoder_name="tf_decoder", nb_classes=97, max_len=25)
    mine = RecModel(args).param_shapes()
    ref = [k for k in keys if not k.endswith("position_table") and not k.startswith("patch_embed.")]
    assert sorted(mine.keys()) == sorted(ref)
    assert mine["decoder.trg_word_emb.weight"] == (98, 512) and mine["decoder.classifier.weight"] == (97, 512)
    want = {**D.finetune_encoder_shapes(O.make_config("pretrain_simmim_moco_ori_vit_small_patch4_32x128")), **D.decoder_param_shapes(D.DecoderConfig())}
    assert {k: tuple(v) for k, v in mine.items()} == {k: tuple(v) for k, v in want.items()}


# ------------------------------------------------------------------------------------------------ device
def _tiny_model(c, ecfg, P):
    from dig_amd.recognizer import RecModel
    m = RecModel(embed_dim=ecfg.embed_dim, depth=ecfg.depth, num_heads=ecfg.heads, n_layers=c.n_layers, d_model=c.d_model, n_head=c.n_head,
                 d_k=c.d_k, d_inner=c.d_inner, nb_classes=c.num_classes, max_len=c.max_seq_len).eval()
    m.load_state_dict(P)
    m._prepare(torch.device("cuda:0"))
    return m


@pytest.mark.gpu
def test_device_decode_steps_vs_reference_fixture():
    g, c = _tiny()
    ecfg = O.DiGConfig(**O.TINY)
    P = {**D.det_encoder_state(ecfg, 22), **D.det_decoder_state(c, int(g["seed"]))}
    m = _tiny_model(c, ecfg, P)
    B, Nm = int(g["B"]), int(g["Nm"])
    mem = O.det_tensor("memory", (B, Nm, c.d_model), 4, 1.0).to("cuda:0").to(torch.bfloat16).reshape(B * Nm, c.d_model).contiguous()
    ref_tok = torch.from_numpy(g["tokens"]).to("cuda:0")
    probs, maps, toks = m.greedy_decode(mem, Nm, force_tokens=ref_tok)          # teacher-forced: every step comparable
    p, r = probs.cpu().numpy(), g["probs"]
    assert np.abs(p - r).max() < 3e-2 and np.abs(p[:, 0] - r[:, 0]).max() < 2e-2
    assert np.abs(maps.cpu().numpy() - g["maps"]).max() < 5e-3
    top2 = np.sort(r, -1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 2.5 * np.abs(p - r).max()             # steps whose arg-max is not a near tie
    assert clear.any() and np.array_equal(toks.cpu().numpy()[clear], g["tokens"][clear])
    probs2, _, toks2 = m.greedy_decode(mem, Nm)                                   # free running: identical while tokens agree
    same = (toks2.cpu().numpy() == g["tokens"]).all(1)
    assert same.any() and np.abs(probs2.cpu().numpy()[same] - r[same]).max() < 3e-2


@pytest.mark.gpu
def test_device_recognizer_vs_reference_fixture():
    g = np.load(os.path.join(GOLD, "recognize_tiny.npz"))
    _, c = _tiny()
    ecfg = O.DiGConfig(**O.TINY)
    P = {**D.det_encoder_state(ecfg, int(g["seed_enc"])), **D.det_decoder_state(c, int(g["seed_dec"]))}
    m = _tiny_model(c, ecfg, P)
    images = O.synthetic_batch(int(g["B"]), ecfg, int(g["batch_seed"]))[0].to("cuda:0")
    enc = m.encoder_features(images)
    mem = m.memory(enc)
    want_enc = D.encoder_features(P, ecfg, images.cpu())
    assert ((enc.float().cpu().reshape(want_enc.shape) - want_enc).norm() / want_enc.norm()).item() < 1e-2
    assert abs(mem.float().norm().item() - float(g["memory_norm"])) < 1e-2 * float(g["memory_norm"])
    probs, maps, toks = m.greedy_decode(mem, 256, force_tokens=torch.from_numpy(g["tokens"]).to("cuda:0"))
    assert np.abs(probs.cpu().numpy() - g["probs"]).max() < 4e-2 and np.abs(maps.cpu().numpy() - g["maps"]).max() < 5e-3
    out = m((images, None, None))                                                 # the RecModel.forward surface
    assert out[0].shape == (int(g["B"]), c.max_seq_len, c.num_classes) and out[3].shape == (int(g["B"]), c.max_seq_len, 256)
    assert abs(float(out[0].sum(-1).mean()) - 1.0) < 1e-4


def test_decode_attention_kernels_vs_torch(abi_dev):
    import ctypes
    from dig_amd import _lib as L
    dev = abi_dev
    torch.manual_seed(0)
    B, T, H, Nm, hk = 7, 9, 3, 200, 192
    qkv = torch.randn(B, T, 3 * hk, device=dev).to(torch.bfloat16)
    out = torch.empty(B, hk, device=dev, dtype=torch.bfloat16)
    for t in (0, 4, 8):
        L.call("dig_decode_self_attn", L.ptr(qkv), L.ptr(out), B, T, H, 64, t, ctypes.c_float(0.125), L.stream())
        q = qkv[:, t, :hk].float().view(B, H, 1, 64)
        k = qkv[:, :t + 1, hk:2 * hk].float().view(B, t + 1, H, 64).permute(0, 2, 1, 3)
        v = qkv[:, :t + 1, 2 * hk:].float().view(B, t + 1, H, 64).permute(0, 2, 1, 3)
        ref = ((q @ k.transpose(-1, -2) * 0.125).softmax(-1) @ v).reshape(B, hk)
        assert (out.float() - ref).abs().max().item() < 2e-2
    q = torch.randn(B, hk, device=dev).to(torch.bfloat16)
    kv = torch.randn(B, Nm, 2 * hk, device=dev).to(torch.bfloat16)
    w = torch.empty(B, H, Nm, device=dev)
    L.call("dig_decode_cross_attn", L.ptr(q), L.ptr(kv), L.ptr(out), L.ptr(w), B, Nm, H, 64, ctypes.c_float(0.125), 1, L.stream())
    qf = q.float().view(B, H, 1, 64)
    kf = kv[:, :, :hk].float().view(B, Nm, H, 64).permute(0, 2, 1, 3)
    vf = kv[:, :, hk:].float().view(B, Nm, H, 64).permute(0, 2, 1, 3)
    wr = (qf @ kf.transpose(-1, -2) * 0.125).softmax(-1)
    assert (w - wr[:, :, 0]).abs().max().item() < 1e-4 and (out.float() - (wr @ vf).reshape(B, hk)).abs().max().item() < 2e-2
    logits = torch.randn(B, 104, device=dev); logits[2, 5] = logits[2, 50] = 9.0       # tie: first index wins (torch.max)
    probs = torch.empty(B, 97, device=dev); tok = torch.empty(B, dtype=torch.int64, device=dev)
    L.call("dig_softmax_argmax", L.ptr(logits), 104, L.ptr(probs), L.ptr(tok), B, 97, L.stream())
    assert (probs - logits[:, :97].softmax(-1)).abs().max().item() < 1e-6 and torch.equal(tok, logits[:, :97].argmax(-1)) and int(tok[2]) == 5


def test_beam_step_kernel_vs_torch(abi_dev):
    """dig_beam_step against the reference's own expressions (log_softmax + topk over [B, bw*C], decoder.py:291-301) on random logits,
    with -inf running scores (dead slots) and EOS masking."""
    from dig_amd import _lib as L
    dev = abi_dev
    torch.manual_seed(1)
    for B, bw, C, ld, eos in ((7, 5, 97, 104, 94), (3, 1, 97, 97, 94), (4, 8, 20, 24, 3)):
        S = B * bw
        logits = torch.randn(S, ld, device=dev) * 3
        seq = (-torch.rand(S, device=dev) * 5)
        seq[torch.rand(S, device=dev) < 0.3] = float("-inf")
        seq[::bw] = 0.0                                                    # at least one live slot per sample
        seq_in = seq.clone()
        sym = torch.empty(S, dtype=torch.int64, device=dev); pred = torch.empty_like(sym); st = torch.empty(S, device=dev)
        L.call("dig_beam_step", L.ptr(logits), ld, L.ptr(seq), B, bw, C, eos, L.ptr(sym), L.ptr(pred), L.ptr(st), L.stream())
        cand = seq_in[:, None] + logits[:, :C].log_softmax(-1)
        sc, ci = cand.view(B, -1).topk(bw, dim=1)
        live = torch.isfinite(sc)                                          # (the order among -inf candidates is unspecified)
        assert torch.equal(sym.view(B, bw)[live], (ci % C)[live])
        assert torch.equal(pred.view(B, bw)[live], (ci // C + (torch.arange(B, device=dev) * bw)[:, None])[live])
        assert (st.view(B, bw)[live] - sc[live]).abs().max().item() < 1e-5
        want_next = sc.masked_fill((ci % C) == eos, float("-inf"))
        assert torch.equal(torch.isfinite(seq.view(B, bw))[live], torch.isfinite(want_next)[live])


@pytest.mark.gpu
def test_device_beam_search_vs_oracle():
    """RecModel.beam_search (K/V-cached decode kernels + dig_beam_step + host back-tracking), beam width 5:
    (1) the search fed the oracle's own fp32 classifier outputs: token-equal with the REFERENCE fixture in both cases;
    (2) end to end on the decoder's own bf16 logits (see below)."""
    g, c = _beam_fixture()
    ecfg = O.DiGConfig(**O.TINY)
    B, Nm, bw, eos = int(g["B"]), int(g["Nm"]), int(g["beam_width"]), int(g["eos"])
    mem32 = O.det_tensor("memory", (B, Nm, c.d_model), 4, 1.0)
    mem = mem32.to("cuda:0").to(torch.bfloat16).reshape(B * Nm, c.d_model).contiguous()
    for tag, boost in (("plain", 0.0), ("eos", float(g["eos_bias_boost"]))):
        P = {**D.det_encoder_state(ecfg, 22), **D.det_decoder_state(c, int(g["seed"]))}
        P["decoder.classifier.bias"][eos] += boost
        # the reference's per-step logits, replayed: run the oracle loop once and record what the classifier produced per slot
        logits = _oracle_beam_logits(P, c, mem32, bw, eos)
        m = _tiny_model(c, ecfg, P)
        m._prepare(torch.device("cuda:0"))
        ids = m.beam_search(mem, Nm, bw, eos=eos, force_logits=logits.to("cuda:0"))
        assert np.array_equal(ids.cpu().numpy(), g["tokens_" + tag]), tag
    # end to end: the decoder's own bf16 logits drive the search.  (a) step 0 (all slots still hold <BOS>) agrees with the fp32
    # oracle's logits; (b) the reference's bookkeeping (log_softmax + topk + back-tracking, fp32 torch) replayed on the DEVICE's
    # per-step logits yields the same hypotheses token for token -- tokens fed back, slot order, EOS masking and back-pointers are
    # wired as in decoder.py:283-369 (a random-weight decoder has too many near ties for a token comparison across precisions)
    P = {**D.det_encoder_state(ecfg, 22), **D.det_decoder_state(c, int(g["seed"]))}
    P["decoder.classifier.bias"][eos] += float(g["eos_bias_boost"])
    m = _tiny_model(c, ecfg, P)
    ids, dev_logits = m.beam_search(mem, Nm, bw, eos=eos, return_logits=True)
    ref_logits = _oracle_beam_logits(P, c, mem32.to(torch.bfloat16).float(), bw, eos)
    assert (dev_logits[0].cpu() - ref_logits[0]).abs().max().item() < 5e-2 * ref_logits[0].abs().max().item()
    nc, T, S = c.num_classes, c.max_seq_len, B * bw
    seq_scores = torch.full((S, 1), -float("inf")); seq_scores[torch.arange(B) * bw] = 0.0
    st_s, st_p, st_y = [], [], []
    for t in range(T):
        cand = seq_scores.repeat(1, nc) + dev_logits[t].cpu().log_softmax(-1)
        sc, ci = cand.view(B, -1).topk(bw, dim=1)
        sym = (ci % nc).view(S)
        st_s.append(sc.view(S)); st_p.append((ci // nc + (torch.arange(B) * bw)[:, None]).view(S)); st_y.append(sym)
        seq_scores = sc.view(S, 1).masked_fill(sym.view(-1, 1).eq(eos), -float("inf"))
    want = D.backtrack(torch.stack(st_s), torch.stack(st_p), torch.stack(st_y), B, bw, eos)
    assert torch.equal(ids.cpu(), want), (ids.cpu(), want)
    assert (ids == eos).any()
    m.beam_width = bw                                                      # the RecModel.forward surface with --beam_width
    m.eos = eos
    images = O.synthetic_batch(2, ecfg, 5)[0].to("cuda:0")
    out = m((images, None, None))
    assert out[0].shape == (2, c.max_seq_len) and out[0].dtype == torch.int64 and torch.equal(out[3], torch.ones_like(out[0]))


def _oracle_beam_logits(P, c, memory, bw, eos):
    """Classifier outputs [T, B*bw, C] of the oracle's beam-search loop (decoder.py:283-290) on `memory`."""
    import torch.nn.functional as F
    B, N, Cm = memory.shape
    nc, T = c.num_classes, c.max_seq_len
    mem = memory.unsqueeze(1).repeat(1, bw, 1, 1).reshape(-1, N, Cm)
    seq = torch.zeros((B * bw, T + 1), dtype=torch.long); seq[:, 0] = c.start_idx
    seq_scores = torch.full((B * bw, 1), -float("inf")); seq_scores[torch.arange(B) * bw] = 0.0
    out = []
    for step in range(T):
        o, _ = D.decoder_attention(P, c, seq, torch.full((B * bw,), step + 1, dtype=torch.long), mem)
        lg = o[:, step] @ P["decoder.classifier.weight"].t() + P["decoder.classifier.bias"]
        out.append(lg)
        cand = seq_scores.repeat(1, nc) + F.log_softmax(lg, dim=-1)
        scores, candidates = cand.view(B, -1).topk(bw, dim=1)
        sym = (candidates % nc).view(B * bw)
        seq_scores = scores.view(B * bw, 1).masked_fill(sym.view(-1, 1).eq(eos), -float("inf"))
        seq[:, step + 1] = sym
    return torch.stack(out)


def test_oracle_string_accuracy_rules():
    voc = D.vocabulary()
    assert len(voc) == 97 and voc[94:] == ["EOS", "PADDING", "UNKNOWN"]
    enc = lambda w, n=8: [voc.index(c) for c in w] + [94] + [95] * (n - len(w) - 1)
    pred = np.array([enc("Hello"), enc("a-b!"), enc("xyz"), [voc.index("q"), 96, voc.index("r"), 94, 1, 2, 3, 4]])
    targ = np.array([enc("hello"), enc("AB"), enc("xy"), enc("qr")])
    assert D.str_list(pred, voc) == ["hello", "ab", "xyz", "qr"] and D.accuracy(pred, targ, voc) == 0.75


def test_device_string_accuracy_matches_oracle(abi_dev):
    dev0 = str(abi_dev)
    from dig_amd.recognizer import accuracy
    voc = D.vocabulary()
    rng = np.random.RandomState(0)
    B, T = 512, 25
    targ = rng.randint(0, 97, size=(B, T))
    pred = targ.copy()
    flip = rng.rand(B, T) < 0.03
    pred[flip] = rng.randint(0, 97, size=int(flip.sum()))
    pred[::7, 3] = 94; targ[::5, 6] = 94                                         # early EOS on either side
    want = D.accuracy(pred, targ, voc)
    got = accuracy(torch.from_numpy(pred).to(dev0), torch.from_numpy(targ).to(dev0), voc).item()
    assert 0.05 < want < 0.95 and abs(got - want) < 1e-7
    p1, t1 = D.str_list(pred, voc), D.str_list(targ, voc)                        # and sample by sample
    from dig_amd import _lib as L
    from dig_amd.recognizer import class_canon
    match = torch.empty(B, dtype=torch.uint8, device=dev0)
    pd, td, cn = torch.from_numpy(pred).to(dev0), torch.from_numpy(targ).to(dev0), class_canon(voc).to(dev0)
    L.call("dig_string_match", L.ptr(pd), L.ptr(td), L.ptr(cn), 97, 94, B, T, L.ptr(match), L.stream())
    assert match.cpu().numpy().astype(bool).tolist() == [a == b for a, b in zip(p1, t1)]


def test_oracle_seq_ce_and_fmeasure_rules():
    voc = D.vocabulary()
    inp = torch.randn(3, 5, 97)
    tgt = torch.randint(0, 97, (3, 5)); lens = torch.tensor([5, 2, 0])
    want = sum(-torch.log_softmax(inp[b, t], -1)[tgt[b, t]] for b in range(3) for t in range(int(lens[b]))) / 3
    assert abs(D.seq_cross_entropy(inp, tgt, lens).item() - want.item()) < 1e-5
    enc = lambda w, n=8: [voc.index(c) for c in w] + [94] + [95] * (n - len(w) - 1)
    f = D.recognition_f_measure(np.array([enc("aab"), enc("xyz")]), np.array([enc("ab"), enc("xw")]), voc)
    assert abs(f - (1.0 * 2 / (2 + 1e-5) * 2 / (2 + 1e-5) * 2 / (2 * 2 / (2 + 1e-5) + 1e-5) + 2 * (1 / (3 + 1e-5)) * (1 / (2 + 1e-5)) / (1 / (3 + 1e-5) + 1 / (2 + 1e-5) + 1e-5)) / 2) < 1e-9


def test_device_seq_ce_and_fmeasure_kernels_match_oracle(abi_dev):
    """dig_seq_cross_entropy and dig_char_fmeasure (the evaluation loop's loss and character F-measure) against the oracle functions on random
    label rows with early EOS, dropped classes and zero-length samples."""
    from dig_amd.recognizer import SeqCrossEntropyLoss, recognition_f_measure
    voc = D.vocabulary()
    g = torch.Generator().manual_seed(3)
    B, T = 37, 25
    inp = torch.randn(B, T, 97, generator=g) * 2
    tgt = torch.randint(0, 97, (B, T), generator=g)
    lens = torch.randint(0, T + 1, (B,), generator=g)
    got = SeqCrossEntropyLoss()(inp.to(abi_dev), tgt.to(abi_dev), lens.to(abi_dev)).item()
    assert abs(got - D.seq_cross_entropy(inp, tgt, lens).item()) < 1e-4 * abs(got)
    rng = np.random.RandomState(5)
    targ = rng.randint(0, 97, size=(B, T))
    pred = targ.copy()
    flip = rng.rand(B, T) < 0.2
    pred[flip] = rng.randint(0, 97, size=int(flip.sum()))
    pred[::4, 5] = 94; targ[::3, 9] = 94
    f = recognition_f_measure(torch.from_numpy(pred).to(abi_dev), torch.from_numpy(targ).to(abi_dev), voc).item()
    assert abs(f - D.recognition_f_measure(pred, targ, voc)) < 1e-12


@pytest.mark.gpu
def test_device_evaluate_loop_matches_oracle():
    """dig_amd.engine_for_finetuning.evaluate (greedy decode + SeqCrossEntropyLoss + Accuracy + recognition_f_measure on the
    device) against the same quantities computed by the oracle functions from the device's own decode output."""
    import types
    from dig_amd.engine_for_finetuning import evaluate
    _, c = _tiny()
    ecfg = O.DiGConfig(**O.TINY)
    P = {**D.det_encoder_state(ecfg, 22), **D.det_decoder_state(c, 21)}
    m = _tiny_model(c, ecfg, P)
    voc = D.vocabulary()
    rng = np.random.RandomState(5)
    batches = []
    for i in range(3):
        B = 6 + i
        images = O.synthetic_batch(B, ecfg, 300 + i)[0]
        target = torch.from_numpy(rng.randint(0, 94, size=(B, c.max_seq_len)))
        lens = torch.from_numpy(rng.randint(1, c.max_seq_len + 1, size=B))
        for b in range(B):
            target[b, int(lens[b]) - 1] = 94
            target[b, int(lens[b]):] = 95
        batches.append((images, target, lens))
    loader = types.SimpleNamespace(dataset=types.SimpleNamespace(idx_to_class={i: ch for i, ch in enumerate(voc)}))
    loader = type("L", (list,), {})(batches); loader.dataset = types.SimpleNamespace(idx_to_class={i: ch for i, ch in enumerate(voc)})
    stats = evaluate(loader, m, torch.device("cuda:0"), types.SimpleNamespace(beam_width=0))
    tot = sum(b[0].shape[0] for b in batches)
    want_loss, want_acc, want_f = [], 0.0, 0.0
    for images, target, lens in batches:
        probs = m((images.to("cuda:0"), None, None))[0].cpu()
        pred = probs.argmax(-1).numpy()
        want_loss.append(D.seq_cross_entropy(probs, target, lens).item())
        want_acc += D.accuracy(pred, target.numpy(), voc) * images.shape[0]
        want_f += D.recognition_f_measure(pred, target.numpy(), voc) * images.shape[0]
    assert abs(stats["loss"] - np.mean(want_loss)) < 1e-4 * abs(np.mean(want_loss))
    assert abs(stats["acc"] - want_acc / tot) < 1e-9 and abs(stats["recognition_fmeasure"] - want_f / tot) < 1e-9
