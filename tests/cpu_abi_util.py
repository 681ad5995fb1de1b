"""Test-only: run dig_amd/ops.py (the host logic above the C ABI) against cpu_abi/libdig_cpu.so, the plain-C++ build of the
pre-training entry points, so that operator parity tests also run in a container without a GPU.  Nothing in dig_amd/ knows about
that library; this context manager swaps the ctypes handle of dig_amd._lib for the duration of a test module and puts it back."""
import contextlib
import ctypes
import os
import subprocess

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU_LIB = os.path.join(ROOT, "cpu_abi", "libdig_cpu.so")


def build():
    src = os.path.join(ROOT, "cpu_abi", "dig_cpu.cpp")
    deps = [src, os.path.join(ROOT, "cpu_abi", "dig_cpu_rec.cpp"), os.path.join(ROOT, "dig_amd", "csrc", "encoder_block.inc"),
            os.path.join(ROOT, "include", "dig_block_types.h")]
    if not os.path.exists(CPU_LIB) or os.path.getmtime(CPU_LIB) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["make", "-C", os.path.join(ROOT, "cpu_abi")], check=True, capture_output=True)
    return CPU_LIB


@contextlib.contextmanager
def cpu_abi_backend():
    from dig_amd import _lib, ops
    lib = ctypes.CDLL(build())
    saved = (_lib._lib, dict(_lib._fns), _lib.stream, ops._workspace, ops._workspace2)
    scratch = {}

    def workspace(tag):
        def get(dev, numel):
            w = scratch.get(tag)
            if w is None or w.numel() < numel:
                w = scratch[tag] = torch.empty(max(numel, 1 << 20), dtype=torch.float32)
            return w
        return get
    _lib._lib = lib
    _lib._fns.clear()
    _lib.stream = lambda: None                      # the CPU build ignores its stream argument
    ops._workspace, ops._workspace2 = workspace("a"), workspace("b")
    try:
        yield torch.device("cpu")
    finally:
        _lib._lib, _lib.stream, ops._workspace, ops._workspace2 = saved[0], saved[2], saved[3], saved[4]
        _lib._fns.clear()
        _lib._fns.update(saved[1])


def exported():
    out = subprocess.run(["nm", "-D", "--defined-only", build()], check=True, capture_output=True, text=True).stdout
    return sorted(l.split()[-1] for l in out.splitlines() if " T dig_" in l)
