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


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """The caller's current HIP stream as a handle.  torch.cuda.current_stream() builds a Stream object through three Python layers (9 us,
    200 times per step = a quarter of the host time of a step); the raw getter is one C call."""
    if _raw_stream is not None and _get_device is not None:
        return ctypes.c_void_p(_raw_stream(_get_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_fns = {}
_protos = None
_HEADER = os.path.join(os.path.dirname(_HERE), "include", "dig_hip.h")
_SCALARS = {"int": ctypes.c_int, "unsigned": ctypes.c_uint, "unsigned int": ctypes.c_uint, "float": ctypes.c_float,
            "long long": ctypes.c_longlong, "unsigned long long": ctypes.c_ulonglong, "hipStream_t": ctypes.c_void_p}


def _prototypes():
    """{function name: [ctypes argument types]} parsed from include/dig_hip.h -- the header IS the contract, so the binding takes its
    argument types from it: every `long long` parameter is passed as 64 bits whatever the call site wrapped it in, floats are converted,
    a wrong argument count fails in Python instead of reading garbage off the stack."""
    global _protos
    if _protos is None:
        import re
        _protos = {}
        try:
            with open(_HEADER) as f:
                text = re.sub(r"/\*.*?\*/", " ", f.read(), flags=re.S)
        except OSError:
            return _protos                                    # header not shipped next to the package: untyped calls, as before
        for m in re.finditer(r"\b(?:int|long long)\s+(dig_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
            types = []
            for prm in m.group(2).split(","):
                prm = " ".join(prm.split())
                if "*" in prm:
                    types.append(ctypes.c_void_p)
                    continue
                ty = " ".join(w for w in prm.split()[:-1] if w != "const")
                if ty not in _SCALARS:
                    types = None
                    break
                types.append(_SCALARS[ty])
            if types is not None:
                _protos[m.group(1)] = types
    return _protos


def call(name, *args):
    f = _fns.get(name)
    if f is None:
        f = getattr(lib(), name)
        f.restype = ctypes.c_int
        at = _prototypes().get(name)
        if at is not None:
            f.argtypes = at
        _fns[name] = f
    rc = f(*args)
    if rc:
        check(rc, name)
