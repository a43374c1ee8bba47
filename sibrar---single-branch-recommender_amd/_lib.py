"""ctypes binding of libsibrar_hip.so — the C-ABI boundary of the engine (include/sibrar_hip.h).

The prototypes are parsed from the header itself so that the binding and the ABI cannot drift apart. There is NO
CPU fallback: if the shared library is missing, ``lib()`` raises, and every op of this package fails loudly.
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libsibrar_hip.so')
if os.environ.get('SBR_LAB_LIB'):        # lab: a kernel variant built by tools/lab/build_*_variants.sh (never set in product runs)
    LIB_PATH = os.path.abspath(os.environ['SBR_LAB_LIB'])
    import sys as _sys
    print(f'[sibrar_amd] SBR_LAB_LIB is set: loading the LAB library {LIB_PATH} instead of the product library — lab builds carry '
          f'timing-only ablation switches; never use one for results', file=_sys.stderr, flush=True)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'sibrar_hip.h')


class SibrarHipError(RuntimeError):
    pass


_CTYPES = {
    'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float, 'double': ctypes.c_double,
    'unsigned long long': ctypes.c_ulonglong, 'unsigned int': ctypes.c_uint, 'signed char': ctypes.c_byte,
}


def parse_header(path: str = HEADER_PATH):
    """-> {name: (restype, [argtypes], [argnames])} for every prototype declared in the header."""
    text = open(path).read()
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    text = re.sub(r'//[^\n]*', ' ', text)
    text = re.sub(r'#[^\n]*', ' ', text)
    protos = {}
    for m in re.finditer(r'([A-Za-z_][\w\s\*]*?)\b(sbr_\w+)\s*\(([^;{}]*?)\)\s*;', text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        ret = ret.replace('extern "C"', '').strip()
        restype = (ctypes.c_char_p if 'char' in ret else ctypes.c_void_p) if '*' in ret \
            else _CTYPES.get(ret.replace('const', '').strip(), ctypes.c_int)
        argtypes, argnames = [], []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                mm = re.match(r'(.*?)(\w+)$', a)
                typ, nm = mm.group(1).strip(), mm.group(2)
                if '*' in typ:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_CTYPES[typ.replace('const', '').strip()])
                argnames.append(nm)
        protos[name] = (restype, argtypes, argnames)
    return protos


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SibrarHipError(
                f'{LIB_PATH} is missing: build it with `python __graft_entry__.py` (or `make -C '
                f'"{os.path.join(_HERE, "csrc")}"`). This package has no CPU fallback.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes, _) in parse_header().items():
            fn = getattr(handle, name)        # AttributeError here == the library does not export a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        _LIB = handle
    return _LIB


def check(status: int):
    if status != 0:
        raise SibrarHipError(lib().sbr_last_error().decode())


_FN = {}


_TRACE = os.environ.get('SBR_TRACE_CALLS', '0') == '1'
CALL_LOG = None      # tests: set to a list and every entry point called through ``call`` is appended as (name, args)
CALL_TIMER = None    # ops.KernelTimer: callable(name, thunk) -> status that brackets the launch with HIP events on its stream


def call(name: str, *args):
    """Invoke an int-returning entry point and raise SibrarHipError on a non-zero status."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(lib(), name)
    if CALL_LOG is not None:
        CALL_LOG.append((name, args))
    if _TRACE:
        # SBR_TRACE_CALLS=1 (debug aid): name every entry point before it runs and wait for it, so that a device fault is
        # reported next to the launch that caused it
        import sys
        import torch
        print(f'[sbr] {name} {args}', file=sys.stderr, flush=True)
        rc = fn(*args)
        torch.cuda.synchronize()
    elif CALL_TIMER is not None:
        rc = CALL_TIMER(name, lambda: fn(*args))
    else:
        rc = fn(*args)
    if rc != 0:
        raise SibrarHipError(lib().sbr_last_error().decode())


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


import threading


class _PinnedStream(threading.local):
    """Per-thread pinned stream handle (the loader threads and the launch thread must never see each other's)."""
    value = None


_STREAM = _PinnedStream()


def stream():
    """hipStream_t of torch's current stream. ``pin_stream`` caches it for the duration of a fused step."""
    if _STREAM.value is not None:
        return _STREAM.value
    import torch
    return torch.cuda.current_stream().cuda_stream


class pin_stream:
    """Context manager: resolve torch's current stream once and reuse the handle for every kernel call inside."""

    def __enter__(self):
        import torch
        self.prev = _STREAM.value
        _STREAM.value = torch.cuda.current_stream().cuda_stream
        return _STREAM.value

    def __exit__(self, *exc):
        _STREAM.value = self.prev
        return False


def to_device(t, device):
    """Host -> device copy that is asynchronous only for PINNED sources. An "async" copy from pageable memory may read the
    host buffer after the call returned (observed on ROCm: a temporary staging tensor was recycled before its copy ran — garbage
    indices on the device, then an out-of-bounds access in the kernels that consumed them), so pageable sources are copied
    synchronously."""
    if t.device.type != 'cpu':
        return t.to(device, non_blocking=True)
    return t.to(device, non_blocking=t.is_pinned())
