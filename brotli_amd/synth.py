"""Deterministic synthetic workloads (text / binary log) for tests and bench.

Thin ctypes wrapper over csrc/synth.c (built to brotli_amd/libm4gsynth.so).
Generator definition: SURVEY.md section 8(d); BASELINE.json configs 2-5.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libm4gsynth.so")
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = ctypes.CDLL(path)
        lib.m4g_synth_fill.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64,
                                       ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
        lib.m4g_synth_fill.restype = None
        _LIB = lib
    return _LIB


def fill(out: np.ndarray, chunk_bytes: int, first_chunk: int = 0, kind: str = "text",
         nthreads: int = 8) -> np.ndarray:
    """Fill a uint8 array (len = nchunks*chunk_bytes) with synthetic chunks."""
    assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"]
    assert out.size % chunk_bytes == 0
    _lib().m4g_synth_fill(out.ctypes.data, chunk_bytes, first_chunk, out.size // chunk_bytes,
                          {"text": 0, "log": 1}[kind], nthreads)
    return out


def text(nbytes: int, chunk_index: int = 0) -> bytes:
    a = np.empty(nbytes, dtype=np.uint8)
    if nbytes:
        fill(a, nbytes, chunk_index, "text", 1)
    return a.tobytes()


def log(nbytes: int, chunk_index: int = 0) -> bytes:
    a = np.empty(nbytes, dtype=np.uint8)
    if nbytes:
        fill(a, nbytes, chunk_index, "log", 1)
    return a.tobytes()
