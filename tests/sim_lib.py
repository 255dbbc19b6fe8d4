"""ctypes access to tests/sim/libparse_sim.so (host build of the parse header;
test harness only)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_L = None


def lib():
    global _L
    if _L is None:
        L = ctypes.CDLL(os.path.join(_HERE, "sim", "libparse_sim.so"))
        L.sim_find_matches.restype = ctypes.c_int64
        L.sim_find_matches.argtypes = [ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_int] * 6 + \
            [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int64]
        L.sim_find_matches_spec.restype = ctypes.c_int64
        L.sim_find_matches_spec.argtypes = [ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_int] * 6 + \
            [ctypes.c_uint32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        _L = L
    return _L


def _args(data, MaxDistance=0, MinLength=0, HashLen=0, TableBits=0, ChainLength=0,
          DistanceBitCost=0):
    a = np.frombuffer(bytes(data), dtype=np.uint8)
    buf = np.concatenate([a, np.zeros(16, np.uint8)])
    out = np.zeros((len(a) // (MinLength or 4) + 8, 4), dtype=np.uint32)
    p = (MaxDistance or 65535, MinLength or 4, HashLen or 6, TableBits or 17, ChainLength,
         DistanceBitCost)
    return a, buf, out, p


def sim(data, seed=0, **kw):
    a, buf, out, p = _args(data, **kw)
    r = lib().sim_find_matches(buf.ctypes.data, len(a), *p, seed, out.ctypes.data, len(out))
    assert r >= 0, r
    return out[:r, :3].copy()


def sim_spec(data, sub_block, seed=0, **kw):
    a, buf, out, p = _args(data, **kw)
    stats = np.zeros(4, dtype=np.int64)
    r = lib().sim_find_matches_spec(buf.ctypes.data, len(a), *p, seed, sub_block,
                                    out.ctypes.data, len(out), stats.ctypes.data)
    assert r >= 0, r
    return out[:r, :3].copy(), stats


def sim_k3(data, sub_block, ext_limit=4096, legacy_limit=False, seed=0, **kw):
    """The whole K3 stage (k3a + k3b + k3c + k4) as the device runs it, on the host."""
    a, buf, out, p = _args(data, **kw)
    L = lib()
    L.sim_find_matches_k3.restype = ctypes.c_int64
    L.sim_find_matches_k3.argtypes = [ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_int] * 6 + \
        [ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
         ctypes.c_void_p]
    stats = np.zeros(5, dtype=np.int64)
    r = L.sim_find_matches_k3(buf.ctypes.data, len(a), *p, seed, sub_block, ext_limit,
                              1 if legacy_limit else 0, out.ctypes.data, len(out), stats.ctypes.data)
    assert r >= 0, r
    return out[:r, :3].copy(), stats


def sim_k2(data, seed=0, **kw):
    """sim() with the chain walks taken from K2's per-position results (m4_k2.h), computed on the host."""
    a, buf, out, p = _args(data, **kw)
    L = lib()
    L.sim_find_matches_k2.restype = ctypes.c_int64
    L.sim_find_matches_k2.argtypes = [ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_int] * 6 + \
        [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int64]
    r = L.sim_find_matches_k2(buf.ctypes.data, len(a), *p, seed, out.ctypes.data, len(out))
    assert r >= 0, r
    return out[:r, :3].copy()


def k2_reference(data, **kw):
    """K2's results for one fresh chunk by chain walks on the host: ((n, 4) uint32 results, (n,) uint8 marks)."""
    a, buf, _, p = _args(data, **kw)
    L = lib()
    L.sim_k2_reference.restype = None
    L.sim_k2_reference.argtypes = [ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_int] * 6 + [ctypes.c_void_p] * 2
    res = np.zeros((len(a), 4), dtype=np.uint32)
    mark = np.zeros(len(a), dtype=np.uint8)
    L.sim_k2_reference(buf.ctypes.data, len(a), *p, res.ctypes.data, mark.ctypes.data)
    return res, mark


def sim_sw(data, seed=0, **kw):
    """sim() with the walks' candidates read from the bucket-sorted position list (SortEnt), built on the host."""
    a, buf, out, p = _args(data, **kw)
    L = lib()
    L.sim_find_matches_sw.restype = ctypes.c_int64
    L.sim_find_matches_sw.argtypes = [ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_int] * 6 + \
        [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int64]
    r = L.sim_find_matches_sw(buf.ctypes.data, len(a), *p, seed, out.ctypes.data, len(out))
    assert r >= 0, r
    return out[:r, :3].copy()
