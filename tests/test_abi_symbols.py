"""The C-ABI library builds, loads without a GPU and exports every symbol include/*.h declares (no compute)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    inc = os.path.join(ROOT, "include")
    txt = "".join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith(".h"))
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dig_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    import torch  # noqa: F401  (same load order as the product: torch's HIP runtime first)
    from dig_amd import build
    path = build.build(verbose=False)
    return ctypes.CDLL(path)


def test_header_declares_something():
    syms = declared_symbols()
    assert len(syms) >= 40 and "dig_gemm_bf16" in syms and "dig_attn_bwd" in syms


def test_every_declared_symbol_is_exported(lib):
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_every_exported_entry_point_is_declared(lib):
    """No undocumented entry points: extern "C" functions in csrc/ must all appear in the header."""
    src = ""
    csrc = os.path.join(ROOT, "dig_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".inc")):
            src += open(os.path.join(csrc, f)).read()
    exported = set(re.findall(r'extern "C" (?:int|long long) (dig_[a-z0-9_]+)\(', src))
    assert exported == set(declared_symbols())


def test_host_side_argument_validation_needs_no_gpu(lib):
    """Entry points reject bad arguments on the host before any launch (error convention of the ABI)."""
    lib.dig_gemm_effective_splits.restype = ctypes.c_int
    assert lib.dig_gemm_effective_splits(65536, 18) == 18
    assert lib.dig_gemm_effective_splits(716, 8) == 6           # 12 K-tiles of 64 -> 2 per slice -> 6 slices
    assert lib.dig_gemm_effective_splits(0, 1) == 0
    lib.dig_gemm_bf16.restype = ctypes.c_int
    null = ctypes.c_void_p(0)
    rc = lib.dig_gemm_bf16(null, null, null, 1, 8, 64, 64, 64, 8, 0, 0, 0, null, null, 0, null, 0, ctypes.c_float(1), 0, 0, 1, 0, 0, 0, null)
    assert rc == -1
    lib.dig_attn_fwd.restype = ctypes.c_int
    assert lib.dig_attn_fwd(null, null, null, 1, 6, 384, null) == -1
    lib.dig_layernorm_bwd_workspace_bytes.restype = ctypes.c_longlong
    assert lib.dig_layernorm_bwd_workspace_bytes(65536, 384) == 1024 * 3 * 384 * 4


def test_hot_kernels_use_no_scratch():
    """The kernels that own the step (dig_amd.build.HOT_KERNELS: the top rows of profiles/*_kernel_stats.csv) carry no scratch segment in
    the built code objects (private_segment_fixed_size == 0: no spilled registers); tools/check_scratch.py prints the table."""
    from dig_amd import build
    build.build(verbose=False)
    rows = build.kernel_resources()
    assert len(rows) > 150 and sum(1 for r in rows if r["hot"]) >= len(build.HOT_KERNELS)
    assert build.check_scratch(rows) == []
