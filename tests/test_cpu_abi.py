"""cpu_abi/libdig_cpu.so: the plain-C++ build of the entry points of include/dig_hip.h ("a CPU build of
the same ABI so the parity tests run in a GPU-less container").  The operator parity itself is tests/test_gpu_kernels.py, which runs
every test on both builds; here: what the build exports, that its prototypes are the header's, that it rejects bad arguments with
the ABI's error codes, and (on the GPU box) that the integer / byte operators of both builds agree bit for bit."""
import ctypes

import pytest
import torch

from cpu_abi_util import build, cpu_abi_backend, exported
from test_abi_symbols import declared_symbols

def test_exports_every_entry_point_of_the_header():
    """dig_cpu.cpp (the pre-training step) + dig_cpu_rec.cpp (the recognition rows N1 / N3 / N4): the whole header, nothing else."""
    exp, decl = set(exported()), set(declared_symbols())
    assert exp == decl, (sorted(exp - decl), sorted(decl - exp))


def test_product_never_loads_the_cpu_build():
    """No module under dig_amd/ mentions the CPU library: without libdig_hip.so the product raises (test_host_logic::test_no_cpu_fallback)."""
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dig_amd")
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                assert "dig_cpu" not in open(os.path.join(d, f), errors="ignore").read(), f


def test_error_codes_match_the_abi():
    lib = ctypes.CDLL(build())
    null = ctypes.c_void_p(0)
    lib.dig_gemm_effective_splits.restype = ctypes.c_int
    assert lib.dig_gemm_effective_splits(65536, 18) == 18 and lib.dig_gemm_effective_splits(716, 8) == 6
    assert lib.dig_gemm_bf16(null, null, null, 1, 8, 64, 64, 64, 8, 0, 0, 0, null, null, 0, null, 0, ctypes.c_float(1), 0, 0, 1, 0, 0, 0, null, null) == -1
    assert lib.dig_attn_fwd(null, null, null, 1, 6, 384, null) == -1
    lib.dig_layernorm_bwd_workspace_bytes.restype = ctypes.c_longlong
    assert lib.dig_layernorm_bwd_workspace_bytes(65536, 384) == 1024 * 3 * 384 * 4
    with cpu_abi_backend():
        from dig_amd import ops, _lib
        x = torch.zeros(64, 72, dtype=torch.bfloat16)                      # R = 72 is not a multiple of 64: rejected on the host
        w = torch.zeros(64, 72, dtype=torch.bfloat16)
        with pytest.raises(_lib.DigHipError):
            ops.gemm(x, w, 64, 64, 72)
        y = torch.zeros(64, 100, dtype=torch.bfloat16)                     # LayerNorm width outside {64..512}: -4 unsupported
        with pytest.raises(_lib.DigHipError, match="unsupported"):
            ops.layernorm_fwd(y, torch.ones(100), torch.zeros(100), 1e-6)


@pytest.mark.gpu
def test_hip_and_cpu_builds_agree_bit_for_bit_on_integer_and_byte_work():
    """mask -> index, row gather, MIM target, fp32 -> bf16 rounding, padded casts, bf16 adds and the dropout hash: same bits from both builds."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import dig_oracle as O
    from dig_amd import ops, dropout
    dev = torch.device("cuda:0")
    cfg = O.DiGConfig()
    im, _, mk = O.synthetic_batch(5, cfg, 77)
    m8 = mk[:, 0].to(torch.uint8)
    src = torch.randn(5 * 256, 384).bfloat16()
    f32 = torch.randn(333, 48) * 3
    a, b = torch.randn(4096).bfloat16(), torch.randn(4096).bfloat16()
    act = torch.randn(96, 128).bfloat16()
    spec = dropout.DropPlan(1234, 7).spec(11, 0.1, 12, 0.2, 32)

    def run(d):
        idx, cnt = ops.mask_to_index(m8.to(d), 179)
        out = [idx, cnt, ops.mim_target(im.to(d), idx, 5 * 179, 8, 32), ops.gather_rows(src.to(d), idx, 5 * 179, 896)]
        c16 = torch.empty(333 * 48, dtype=torch.bfloat16, device=d)
        ops.cast_f32_to_bf16(f32.to(d).reshape(-1), c16)
        pc = torch.empty(384, 64, dtype=torch.bfloat16, device=d)
        ops.pad_cast_rows(f32.to(d), pc, 333, 48)
        ab = torch.empty(4096, dtype=torch.bfloat16, device=d)
        ops.add_bf16(a.to(d), b.to(d), ab)
        out += [c16, pc, ab, ops.dropout_apply(act.to(d), spec)]
        return [t.cpu() for t in out]
    hip = run(dev)
    with cpu_abi_backend() as cpu:
        ref = run(cpu)
    for i, (x, y) in enumerate(zip(hip, ref)):
        assert torch.equal(x.view(torch.int16) if x.dtype == torch.bfloat16 else x, y.view(torch.int16) if y.dtype == torch.bfloat16 else y), i


@pytest.mark.gpu
def test_hip_and_cpu_builds_agree_bit_for_bit_on_the_recognition_rows_integer_work():
    """bicubic resize + normalise (fixed-point Pillow arithmetic), Philox mask generator, token embedding rows, string match, beam-step symbols /
    predecessors and the greedy token of dig_softmax_argmax: same bits from both builds (cpu_abi/dig_cpu_rec.cpp)."""
    import numpy as np
    from dig_amd import _lib as L
    from dig_amd.datasets import resize_normalize, RandomMaskingGenerator
    rng = np.random.RandomState(3)
    crops = [rng.randint(0, 256, size=(rng.randint(4, 90), rng.randint(8, 400), 3)).astype(np.uint8) for _ in range(12)]
    g = torch.Generator().manual_seed(9)
    tok = torch.randint(-2, 101, (57,), generator=g)                      # (out-of-range tokens are clamped)
    table = torch.randn(99, 40, generator=g)
    pred = torch.randint(0, 97, (64, 25), generator=g); targ = pred.clone()
    targ[torch.rand(64, 25, generator=g) < 0.05] = 3
    canon = torch.tensor([(i % 37) if i < 94 else 0 for i in range(97)], dtype=torch.uint8)
    B, bw, C, ld = 6, 5, 97, 104
    logits = torch.randn(B * bw, ld, generator=g) * 3
    seq0 = -torch.rand(B * bw, generator=g) * 4

    def run(d):
        out = [resize_normalize(crops, 32, 128, device=d), RandomMaskingGenerator((8, 32), 0.7, num_view=2, seed=77, device=d)(9)]
        emb = torch.empty(57, 48, dtype=torch.bfloat16, device=d)
        t_, tb = tok.to(d), table.to(d)
        L.call("dig_embed_rows", L.ptr(t_), L.ptr(tb), L.ptr(emb), 48, 57, 40, 99, L.stream())
        out.append(emb[:, :40].contiguous())
        match = torch.empty(64, dtype=torch.uint8, device=d)
        p_, q_, c_ = pred.to(d), targ.to(d), canon.to(d)
        L.call("dig_string_match", L.ptr(p_), L.ptr(q_), L.ptr(c_), 97, 94, 64, 25, L.ptr(match), L.stream())
        out.append(match)
        lg, seq = logits.to(d), seq0.clone().to(d)
        sym = torch.empty(B * bw, dtype=torch.int64, device=d); prd = torch.empty_like(sym); st = torch.empty(B * bw, device=d)
        L.call("dig_beam_step", L.ptr(lg), ld, L.ptr(seq), B, bw, C, 94, L.ptr(sym), L.ptr(prd), L.ptr(st), L.stream())
        probs = torch.empty(B * bw, C, device=d); tk = torch.empty(B * bw, dtype=torch.int64, device=d)
        L.call("dig_softmax_argmax", L.ptr(lg), ld, L.ptr(probs), L.ptr(tk), B * bw, C, L.stream())
        out += [sym, prd, tk]
        return [t.cpu() for t in out]
    hip = run(torch.device("cuda:0"))
    with cpu_abi_backend() as cpu:
        ref = run(cpu)
    for i, (x, y) in enumerate(zip(hip, ref)):
        assert torch.equal(x.view(torch.int16) if x.dtype == torch.bfloat16 else x, y.view(torch.int16) if y.dtype == torch.bfloat16 else y), i
