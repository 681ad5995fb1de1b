"""ctypes binding of the C-ABI in include/dig_hip.h (libdig_hip.so).  Fails loudly when the extension is
missing: there is no eager/CPU fallback in this package."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdig_hip.so")

_lib = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_f = ctypes.c_float
c_l = ctypes.c_longlong


class DigHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DigHipError(f"{LIB_PATH} not found: build it with `python -m dig_amd.build` "
                              "(hipcc, gfx950). dig_amd has no fallback path.")
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


_ERR = {-1: "bad argument", -2: "misaligned pointer / leading dimension", -3: "kernel launch failed",
        -4: "unsupported configuration"}


def check(rc, what):
    if rc != 0:
        raise DigHipError(f"{what}: {_ERR.get(rc, 'error')} (rc={rc})")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_fns = {}


def call(name, *args):
    f = _fns.get(name)
    if f is None:
        f = getattr(lib(), name)
        f.restype = ctypes.c_int
        _fns[name] = f
    rc = f(*args)
    if rc:
        check(rc, name)
