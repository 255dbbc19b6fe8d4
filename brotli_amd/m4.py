"""Host-side mirror of the reference's plugin interface over libm4g.so.

``M4`` has the exported fields of ``matchfinder.M4`` (matchfinder/m4.go:15-43)
and the two methods of ``matchfinder.MatchFinder`` (matchfinder/matchfinder.go:23-30):
``FindMatches(dst, src)`` and ``Reset()``.  ``Writer`` restates
``matchfinder.Writer`` (matchfinder/matchfinder.go:44-103) so the parity tests
read like the reference's ``test()`` helper (brotli_test.go:626-649).

Everything here calls the C-ABI of include/m4g.h through ctypes; there is no
Python or CPU implementation of the match finder in this package.  Loading
fails loudly when the HIP library is missing.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("M4G_LIB") or os.path.join(_HERE, "libm4g.so")

M4G_OK = 0
ERR_BASE = -1000000


class BargainParams(ctypes.Structure):
    """m4g_bargain_params: matchfinder.Bargain1 / Bargain2 and the Writer's BlockSize."""
    _fields_ = [("variant", ctypes.c_int32), ("max_distance", ctypes.c_int32),
                ("skip", ctypes.c_int32), ("block_size", ctypes.c_int32)]


class Params(ctypes.Structure):
    _fields_ = [("max_distance", ctypes.c_int32), ("min_length", ctypes.c_int32),
                ("hash_len", ctypes.c_int32), ("table_bits", ctypes.c_int32),
                ("chain_length", ctypes.c_int32), ("distance_bit_cost", ctypes.c_int32)]


EXPORTS = [
    "m4g_strerror", "m4g_last_error", "m4g_device_count", "m4g_create", "m4g_destroy",
    "m4g_reset", "m4g_find_matches", "m4g_find_matches_blocks", "m4g_debug_dump_tables",
    "m4g_history_len",
    "m4g_batch_create", "m4g_batch_destroy", "m4g_batch_run_device", "m4g_batch_run_device_codes",
    "m4g_find_matches_batch",
    "m4g_batch_last_timings", "m4g_batch_set_timing", "m4g_host_alloc", "m4g_host_free",
    "m4g_batch_status", "m4g_batch_info", "m4g_debug_batch_set_k2", "m4g_debug_batch_set_sorted_walk", "m4g_debug_batch_dump_k2", "m4g_batch_run_device_noalloc", "m4g_debug_noalloc_poison", "m4g_find_matches_blocks_multi", "m4g_debug_span_plan", "m4g_bargain_ctx_create", "m4g_bargain_ctx_destroy", "m4g_bargain_ctx_reset", "m4g_bargain_ctx_find_blocks", "m4g_bargain_find_blocks_multi",
    "m4g_batch_encode", "m4g_batch_flate_encode", "m4g_flate_encode_blocks", "m4g_brotli_encode_blocks",
    "m4g_hostbatch_compress", "m4g_writer_create", "m4g_writer_write", "m4g_writer_close", "m4g_writer_reset", "m4g_writer_destroy", "m4g_hostbatch_create", "m4g_hostbatch_run", "m4g_hostbatch_destroy",
    "m4g_bargain_create", "m4g_bargain_destroy", "m4g_bargain_block_count", "m4g_bargain_run_device",
    "m4g_bargain_last_timings", "m4g_bargain_find_blocks", "m4g_brotli_compress_v2", "m4g_writer_create_v2",
]

_LIB = None


class M4GError(RuntimeError):
    pass


def lib():
    """Load libm4g.so (built by __graft_entry__.build()); raises if absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise M4GError(f"{LIB_PATH} is missing: build it with "
                           "`python -c 'import __graft_entry__ as g; g.build()'`. "
                           "There is no CPU fallback.")
        # One HIP runtime per process: PyTorch wheels bundle their own
        # libamdhip64.so.7.  If libm4g.so were loaded first it would pull in the
        # system runtime and a later `import torch` would bring a second HSA
        # runtime into the process, which then sees no GPU.  Importing torch
        # first makes the dynamic linker resolve our NEEDED libamdhip64.so.7
        # to the runtime torch already loaded (same SONAME).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        vp, sz, u64p = ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)
        L.m4g_strerror.argtypes = [ctypes.c_int]
        L.m4g_strerror.restype = ctypes.c_char_p
        L.m4g_last_error.restype = ctypes.c_char_p
        L.m4g_device_count.restype = ctypes.c_int
        L.m4g_create.argtypes = [ctypes.POINTER(Params), ctypes.c_int, ctypes.POINTER(vp)]
        L.m4g_destroy.argtypes = [vp]
        L.m4g_destroy.restype = None
        L.m4g_reset.argtypes = [vp]
        L.m4g_find_matches.argtypes = [vp, vp, sz, ctypes.c_uint32, vp, sz]
        L.m4g_find_matches.restype = ctypes.c_int64
        L.m4g_debug_dump_tables.argtypes = [vp, vp, vp, sz]
        L.m4g_find_matches_blocks.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_size_t, vp, ctypes.c_size_t, vp]
        L.m4g_find_matches_blocks.restype = ctypes.c_int64
        L.m4g_history_len.argtypes = [vp]
        L.m4g_history_len.restype = ctypes.c_int64
        L.m4g_batch_create.argtypes = [ctypes.POINTER(Params), ctypes.c_int, sz, sz,
                                       ctypes.POINTER(vp)]
        L.m4g_batch_destroy.argtypes = [vp]
        L.m4g_batch_destroy.restype = None
        L.m4g_batch_run_device.argtypes = [vp, vp, vp, sz, vp, vp, sz, vp, vp]
        if hasattr(L, "m4g_batch_run_device_codes"):  # absent only in older experiment builds (M4G_LIB)
            L.m4g_batch_run_device_codes.argtypes = [vp, vp, vp, sz, vp, vp, sz, vp, vp, vp]
        L.m4g_find_matches_batch.argtypes = [ctypes.POINTER(Params), vp, vp, sz, vp, ctypes.c_int,
                                             vp, vp, sz, vp]
        L.m4g_find_matches_batch.restype = ctypes.c_int64
        L.m4g_batch_last_timings.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.c_int]
        L.m4g_batch_set_timing.argtypes = [vp, ctypes.c_int]
        L.m4g_batch_set_timing.restype = None
        L.m4g_batch_status.argtypes = [vp]
        L.m4g_batch_info.argtypes = [vp, vp, ctypes.c_int]
        L.m4g_debug_batch_set_k2.argtypes = [vp, ctypes.c_int]
        L.m4g_debug_batch_set_sorted_walk.argtypes = [vp, ctypes.c_int]
        L.m4g_debug_batch_dump_k2.argtypes = [vp, sz, vp, vp, sz]
        L.m4g_hostbatch_create.argtypes = [ctypes.POINTER(Params), vp, ctypes.c_int, sz, sz, ctypes.POINTER(vp)]
        L.m4g_hostbatch_run.argtypes = [vp, vp, vp, sz, vp, vp, sz, vp]
        L.m4g_hostbatch_run.restype = ctypes.c_int64
        L.m4g_hostbatch_destroy.argtypes = [vp]
        L.m4g_hostbatch_compress.argtypes = [vp, ctypes.c_int, ctypes.c_uint32, vp, vp, sz, vp, sz]
        L.m4g_hostbatch_compress.restype = ctypes.c_int64
        L.m4g_hostbatch_destroy.restype = None
        L.m4g_batch_encode.argtypes = [vp, vp, sz, vp, vp, vp, vp, ctypes.c_int, vp, sz, vp,
                                       ctypes.POINTER(ctypes.c_float)]
        L.m4g_batch_encode.restype = ctypes.c_int64
        L.m4g_batch_flate_encode.argtypes = [vp, vp, sz, vp, vp, ctypes.c_int, ctypes.c_uint32, vp, sz, vp,
                                             ctypes.POINTER(ctypes.c_float), vp]
        L.m4g_batch_flate_encode.restype = ctypes.c_int64
        L.m4g_flate_encode_blocks.argtypes = [ctypes.c_int, vp, sz, sz, vp, vp, ctypes.c_int, ctypes.c_uint32, vp, sz, vp]
        L.m4g_flate_encode_blocks.restype = ctypes.c_int64
        L.m4g_brotli_encode_blocks.argtypes = [ctypes.c_int, vp, sz, sz, vp, vp, vp, sz]
        L.m4g_brotli_encode_blocks.restype = ctypes.c_int64
        L.m4g_writer_create.argtypes = [ctypes.POINTER(Params), ctypes.c_int, ctypes.c_int, sz, ctypes.c_uint32,
                                        ctypes.POINTER(vp)]
        L.m4g_writer_write.argtypes = [vp, vp, sz, vp, sz]
        L.m4g_writer_write.restype = ctypes.c_int64
        L.m4g_writer_close.argtypes = [vp, vp, sz]
        L.m4g_writer_close.restype = ctypes.c_int64
        L.m4g_writer_reset.argtypes = [vp]
        L.m4g_writer_destroy.argtypes = [vp]
        L.m4g_writer_destroy.restype = None
        L.m4g_batch_run_device_noalloc.argtypes = [ctypes.POINTER(Params), ctypes.c_int, vp, vp, sz, vp, vp, sz, vp, vp]
        L.m4g_debug_noalloc_poison.argtypes = [ctypes.c_int, ctypes.c_int, vp]
        L.m4g_find_matches_blocks_multi.argtypes = [ctypes.POINTER(Params), vp, ctypes.c_int, vp, sz, sz, vp, sz, vp]
        L.m4g_find_matches_blocks_multi.restype = ctypes.c_int64
        L.m4g_debug_span_plan.argtypes = [ctypes.POINTER(Params), sz, sz, ctypes.c_int, vp, sz, vp]
        L.m4g_debug_span_plan.restype = ctypes.c_int64
        L.m4g_bargain_create.argtypes = [ctypes.POINTER(BargainParams), ctypes.c_int, sz, sz, sz, ctypes.POINTER(vp)]
        L.m4g_bargain_destroy.argtypes = [vp]
        L.m4g_bargain_destroy.restype = None
        L.m4g_bargain_block_count.argtypes = [vp, vp, sz]
        L.m4g_bargain_block_count.restype = sz
        L.m4g_bargain_run_device.argtypes = [vp, vp, vp, sz, vp, sz, vp, vp]
        L.m4g_bargain_run_device.restype = ctypes.c_int64
        L.m4g_bargain_last_timings.argtypes = [vp, vp, ctypes.c_int]
        L.m4g_bargain_find_blocks.argtypes = [ctypes.POINTER(BargainParams), ctypes.c_int, vp, sz, vp, sz, vp]
        L.m4g_bargain_find_blocks.restype = ctypes.c_int64
        L.m4g_bargain_ctx_create.argtypes = [ctypes.POINTER(BargainParams), ctypes.c_int, ctypes.POINTER(vp)]
        L.m4g_bargain_ctx_destroy.argtypes = [vp]
        L.m4g_bargain_ctx_destroy.restype = None
        L.m4g_bargain_ctx_reset.argtypes = [vp]
        L.m4g_bargain_ctx_find_blocks.argtypes = [vp, vp, sz, vp, sz, vp]
        L.m4g_bargain_ctx_find_blocks.restype = ctypes.c_int64
        L.m4g_bargain_find_blocks_multi.argtypes = [ctypes.POINTER(BargainParams), vp, ctypes.c_int, vp, sz, vp, sz, vp]
        L.m4g_bargain_find_blocks_multi.restype = ctypes.c_int64
        L.m4g_writer_create_v2.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]
        L.m4g_brotli_compress_v2.argtypes = [ctypes.c_int, ctypes.c_int, vp, sz, vp, sz]
        L.m4g_brotli_compress_v2.restype = ctypes.c_int64
        L.m4g_host_alloc.argtypes = [sz]
        L.m4g_host_alloc.restype = vp
        L.m4g_host_free.argtypes = [vp]
        L.m4g_host_free.restype = None
        _LIB = L
    return _LIB


STREAM_NEEDED_BIAS = 2000000  # M4G_STREAM_NEEDED_BIAS


def _check(rc: int, what: str):
    if -STREAM_NEEDED_BIAS < rc < ERR_BASE:
        L = lib()
        raise M4GError(f"{what}: {L.m4g_strerror(int(rc)).decode()} "
                       f"({L.m4g_last_error().decode()})")


def _as_u8(src) -> np.ndarray:
    if isinstance(src, np.ndarray):
        return np.ascontiguousarray(src, dtype=np.uint8)
    return np.frombuffer(bytes(src), dtype=np.uint8)


def _params(MaxDistance, MinLength, HashLen, TableBits, ChainLength, DistanceBitCost) -> Params:
    return Params(MaxDistance, MinLength, HashLen, TableBits, ChainLength, DistanceBitCost)


Match = Tuple[int, int, int]  # (Unmatched, Length, Distance)  matchfinder.go:16-20


class M4:
    """``matchfinder.M4`` on an MI355X.  Zero-valued fields take the reference's
    defaults (m4.go:64-75)."""

    def __init__(self, MaxDistance=0, MinLength=0, HashLen=0, TableBits=0, ChainLength=0,
                 DistanceBitCost=0, device: int = 0):
        self.MaxDistance = MaxDistance
        self.MinLength = MinLength
        self.HashLen = HashLen
        self.TableBits = TableBits
        self.ChainLength = ChainLength
        self.DistanceBitCost = DistanceBitCost
        self.device = device
        self._h = None

    # created lazily on first use, like the reference allocates its table on
    # the first FindMatches call (m4.go:76-78)
    def _handle(self):
        if self._h is None:
            h = ctypes.c_void_p()
            p = _params(self.MaxDistance, self.MinLength, self.HashLen, self.TableBits,
                        self.ChainLength, self.DistanceBitCost)
            _check(lib().m4g_create(ctypes.byref(p), self.device, ctypes.byref(h)), "m4g_create")
            self._h = h
        return self._h

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().m4g_destroy(self._h)
            except Exception:
                pass
            self._h = None

    def Reset(self):
        """matchfinder.M4.Reset (m4.go:51-57)."""
        if self._h is not None:
            _check(lib().m4g_reset(self._h), "m4g_reset")

    def find_matches(self, src, seed_prev_distance: int = 0) -> np.ndarray:
        """One FindMatches call; returns an (M, 3) uint32 array."""
        a = _as_u8(src)
        min_len = self.MinLength or 4
        cap = a.size // min_len + 2
        out = np.empty((cap, 3), dtype=np.uint32)
        r = lib().m4g_find_matches(self._handle(), a.ctypes.data if a.size else None, a.size,
                                   seed_prev_distance, out.ctypes.data, cap)
        _check(r, "m4g_find_matches")
        if r < 0:
            raise M4GError(f"m4g_find_matches: out_cap too small, need {-r}")
        return out[:r].copy()

    def history_len(self) -> int:
        """len(M4.history) after the last call."""
        return int(lib().m4g_history_len(self._handle()))

    def find_matches_blocks(self, src, block_size: int):
        """FindMatches on every consecutive block of src in one launch
        (m4g_find_matches_blocks); returns (matches[M,3], block_off[nblocks+1])."""
        a = _as_u8(src)
        min_len = self.MinLength or 4
        nblocks = (a.size + block_size - 1) // block_size
        cap = sum(min(block_size, a.size - b * block_size) // min_len + 2 for b in range(nblocks)) + 1
        out = np.empty((cap, 3), dtype=np.uint32)
        off = np.zeros(nblocks + 1, dtype=np.uint64)
        r = lib().m4g_find_matches_blocks(self._handle(), a.ctypes.data if a.size else None, a.size,
                                          block_size, out.ctypes.data, cap, off.ctypes.data)
        _check(r, "m4g_find_matches_blocks")
        if r < 0:
            raise M4GError(f"m4g_find_matches_blocks: out_cap too small, need {-r}")
        return out[:r], off

    def FindMatches(self, dst: List[Match], src) -> List[Match]:
        """matchfinder.MatchFinder.FindMatches: append to dst, return dst."""
        seed = dst[-1][2] if dst else 0
        if dst and seed == 0:
            raise M4GError("dst ends in a Match with Distance 0; the reference's repeat probe "
                           "degenerates on that input (SURVEY.md section 3.2) - unsupported")
        m = self.find_matches(src, seed)
        dst.extend((int(u), int(l), int(d)) for u, l, d in m.tolist())
        return dst

    def dump_tables(self) -> Tuple[np.ndarray, np.ndarray]:
        L = lib()
        h = self._handle()
        tb = self.TableBits or 17
        n = int(L.m4g_history_len(h))
        table = np.zeros(1 << tb, dtype=np.uint32)
        chain = np.zeros(max(n, 1), dtype=np.uint32)
        _check(L.m4g_debug_dump_tables(h, table.ctypes.data, chain.ctypes.data, n),
               "m4g_debug_dump_tables")
        # bit 31 of a chain entry is the kernels' "4 bytes may match" hint, not part
        # of the reference's chain value
        return table, chain[:n] & np.uint32(0x7FFFFFFF)


class AutoReset:
    """matchfinder.AutoReset (matchfinder/textencoder.go:46-53)."""

    def __init__(self, mf):
        self.MatchFinder = mf

    def Reset(self):
        self.MatchFinder.Reset()

    def FindMatches(self, dst, src):
        self.MatchFinder.Reset()
        return self.MatchFinder.FindMatches(dst, src)


class Writer:
    """matchfinder.Writer (matchfinder/matchfinder.go:44-103).  `Encoder` is any
    object with Encode(src, matches, last_block) -> bytes and Reset(); the
    bit-stream encoder stays on the CPU and is not part of this package."""

    def __init__(self, Dest, MatchFinder, Encoder, BlockSize: int = 0, BlocksPerLaunch: int = 1):
        """BlocksPerLaunch > 1 (not in the reference): full blocks are collected and
        match-found together with M4.find_matches_blocks, then encoded in order.
        The output stream is the same; only the latency of Write changes."""
        self.Dest = Dest
        self.MatchFinder = MatchFinder
        self.Encoder = Encoder
        self.BlockSize = BlockSize
        self.BlocksPerLaunch = BlocksPerLaunch
        self._in = bytearray()

    def Write(self, p) -> int:
        p = bytes(p)
        if self.BlockSize == 0:
            self._write_block(p, False)
            return len(p)
        self._in += p
        pos = 0
        if self.BlocksPerLaunch > 1 and hasattr(self.MatchFinder, "find_matches_blocks"):
            span = self.BlockSize * self.BlocksPerLaunch
            while pos + span <= len(self._in):
                self._write_blocks(bytes(self._in[pos:pos + span]))
                pos += span
        else:
            while pos + self.BlockSize <= len(self._in):
                self._write_block(bytes(self._in[pos:pos + self.BlockSize]), False)
                pos += self.BlockSize
        if pos:
            del self._in[:pos]
        return len(p)

    def _write_blocks(self, p: bytes):
        m, off = self.MatchFinder.find_matches_blocks(p, self.BlockSize)
        for b in range(len(off) - 1):
            blk = p[b * self.BlockSize:(b + 1) * self.BlockSize]
            self.Dest.write(self.Encoder.Encode(blk, m[int(off[b]):int(off[b + 1])], False))

    def _write_block(self, p: bytes, last: bool):
        matches = self.MatchFinder.FindMatches([], p)
        m = np.array(matches, dtype=np.uint32).reshape(-1, 3)
        self.Dest.write(self.Encoder.Encode(p, m, last))

    def Close(self):
        full = len(self._in) // self.BlockSize * self.BlockSize if self.BlockSize else 0
        if full and self.BlocksPerLaunch > 1 and hasattr(self.MatchFinder, "find_matches_blocks"):
            self._write_blocks(bytes(self._in[:full]))
            del self._in[:full]
        while self.BlockSize and len(self._in) >= self.BlockSize:  # cannot happen after Write; kept for safety
            self._write_block(bytes(self._in[:self.BlockSize]), False)
            del self._in[:self.BlockSize]
        self._write_block(bytes(self._in), True)
        self._in.clear()

    def Reset(self, newDest):
        self.MatchFinder.Reset()
        self.Encoder.Reset()
        self._in.clear()
        self.Dest = newDest


class BatchPlan:
    """Independent chunks, device resident (m4g_batch_*).  Tensors are torch
    CUDA(HIP) tensors; torch is used only as the device allocator."""

    def __init__(self, max_chunks: int, max_chunk_bytes: int, device: int = 0, **m4_fields):
        self.params = _params(m4_fields.get("MaxDistance", 0), m4_fields.get("MinLength", 0),
                              m4_fields.get("HashLen", 0), m4_fields.get("TableBits", 0),
                              m4_fields.get("ChainLength", 0), m4_fields.get("DistanceBitCost", 0))
        self.device = device
        self.max_chunks = max_chunks
        self.max_chunk_bytes = max_chunk_bytes
        h = ctypes.c_void_p()
        _check(lib().m4g_batch_create(ctypes.byref(self.params), device, max_chunks,
                                      max_chunk_bytes, ctypes.byref(h)), "m4g_batch_create")
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().m4g_batch_destroy(self._h)
            except Exception:
                pass
            self._h = None

    def set_timing(self, on: bool):
        lib().m4g_batch_set_timing(self._h, 1 if on else 0)

    def last_timings(self) -> List[float]:
        ms = (ctypes.c_float * 4)()
        n = lib().m4g_batch_last_timings(self._h, ms, 4)
        return [float(ms[k]) for k in range(n)]

    def status(self):
        """Kernel-side outcome of the last run (m4g_batch_status): synchronises the
        run's stream; raises M4GError if records were dropped or a repeat distance
        was out of range."""
        _check(lib().m4g_batch_status(self._h), "m4g_batch_status")

    def info(self) -> dict:
        """Device memory and record-slot layout of the plan (m4g_batch_info)."""
        v = (ctypes.c_uint64 * 8)()
        n = lib().m4g_batch_info(self._h, v, 8)
        keys = ("bytes", "slot", "slot_worst", "group_chunks", "reruns", "k2", "k2_bytes", "sorted_walk")
        return {k: int(v[i]) for i, k in enumerate(keys[:n])}

    def set_k2(self, on: bool) -> bool:
        """Test hook: run with / without K2 (the chain walks as a kernel of their own); False when the plan has none."""
        return bool(lib().m4g_debug_batch_set_k2(self._h, 1 if on else 0))

    def set_sorted_walk(self, on: bool) -> bool:
        """Test hook: the parse reads its walks from the sorted position list (True) or chases the chains."""
        return bool(lib().m4g_debug_batch_set_sorted_walk(self._h, 1 if on else 0))

    def dump_k2(self, c: int, n: int):
        """Test hook: (chain entries, K2 results as an (n, 4) uint32 array) of chunk c of the last run."""
        chain = np.empty(n, dtype=np.uint32)
        res = np.empty((n, 4), dtype=np.uint32)
        _check(lib().m4g_debug_batch_dump_k2(self._h, c, chain.ctypes.data, res.ctypes.data, n), "m4g_debug_batch_dump_k2")
        return chain, res

    def run(self, d_src, chunk_off: np.ndarray, d_out, d_out_off, d_histos=None, stream: int = 0,
            d_codes=None):
        """d_src: uint8 tensor; chunk_off: host uint64 array (nchunks+1);
        d_out: int32/uint32-view tensor with 3*out_cap elements; d_out_off:
        int64 tensor (nchunks+1); d_histos: optional int32 tensor (nchunks*1024);
        d_codes: optional tensor of 8 bytes per record of d_out (m4g_code: u16
        command, u8 dist_code, u8 n_extra, u32 extra_bits; encoder.go:57-97)."""
        off = np.ascontiguousarray(chunk_off, dtype=np.uint64)
        n = off.size - 1
        out_cap = d_out.numel() * d_out.element_size() // 12
        if d_codes is not None:
            if d_codes.numel() * d_codes.element_size() < 8 * out_cap:
                raise M4GError("d_codes holds fewer records than d_out")
            rc = lib().m4g_batch_run_device_codes(
                self._h, d_src.data_ptr(), off.ctypes.data, n, d_out.data_ptr(),
                d_out_off.data_ptr(), out_cap,
                d_histos.data_ptr() if d_histos is not None else None, d_codes.data_ptr(),
                stream if stream else None)
            _check(rc, "m4g_batch_run_device_codes")
            return
        rc = lib().m4g_batch_run_device(self._h, d_src.data_ptr(), off.ctypes.data, n,
                                        d_out.data_ptr(), d_out_off.data_ptr(), out_cap,
                                        d_histos.data_ptr() if d_histos is not None else None,
                                        stream if stream else None)
        _check(rc, "m4g_batch_run_device")


    def encode(self, d_src, nchunks: int, d_out, d_out_off, d_histos, d_codes, first: bool = True,
               last: bool = True, stream: int = 0, cap: "int | None" = None, out: "np.ndarray | None" = None):
        """m4g_batch_encode: brotli.Encoder.Encode for every chunk of the plan's last
        run (which must have produced d_histos and d_codes), pass 2 on the GPU.
        Returns (stream bytes, [host ms, kernel ms, total ms]); with `out` (a uint8
        array, e.g. PinnedBuffer.array) the stream is left there and its length is
        returned in place of the bytes."""
        if out is not None:
            buf, cap = out, out.size
        else:
            if cap is None:
                cap = int(d_src.numel()) + (nchunks + 1) * 1024 + 64
            buf = np.zeros(cap, dtype=np.uint8)
        ms = (ctypes.c_float * 3)()
        r = lib().m4g_batch_encode(self._h, d_src.data_ptr(), nchunks, d_out.data_ptr(), d_out_off.data_ptr(),
                                   d_histos.data_ptr(), d_codes.data_ptr(), (1 if first else 0) | (2 if last else 0),
                                   buf.ctypes.data, cap, stream if stream else None, ms)
        _check(r, "m4g_batch_encode")
        if r < 0:
            raise M4GError(f"m4g_batch_encode: stream_cap too small, need {-r - STREAM_NEEDED_BIAS}")
        return (int(r) if out is not None else buf[:r].tobytes()), [float(ms[k]) for k in range(3)]

    def flate_encode(self, d_src, nchunks: int, d_out, d_out_off, first: bool = True, last: bool = True,
                     gzip: bool = False, mtime: int = 0, stream: int = 0, cap: "int | None" = None,
                     out: "np.ndarray | None" = None):
        """m4g_batch_flate_encode: flate.NewEncoder() / NewGZIPEncoder() Encode for every
        chunk of the plan's last run (flate/huffman_bit_writer.go, flate/gzip.go), symbol
        counting and token writing on the GPU.  Returns (stream bytes or, with `out`, the
        length; [host ms, kernel ms, total ms]; block kind per chunk: 0 stored, 1 fixed,
        2 dynamic).  A match outside Length 3..258 / Distance 1..32768 raises M4GError
        (the reference panics there, flate/token.go:59-71)."""
        if out is not None:
            buf, cap = out, out.size
        else:
            if cap is None:
                cap = int(d_src.numel()) + (nchunks + 1) * 16 + 64
            buf = np.zeros(cap, dtype=np.uint8)
        ms = (ctypes.c_float * 3)()
        kinds = np.zeros(max(nchunks, 1), dtype=np.uint8)
        r = lib().m4g_batch_flate_encode(self._h, d_src.data_ptr(), nchunks, d_out.data_ptr(), d_out_off.data_ptr(),
                                         (1 if first else 0) | (2 if last else 0) | (4 if gzip else 0), mtime,
                                         buf.ctypes.data, cap, stream if stream else None, ms, kinds.ctypes.data)
        _check(r, "m4g_batch_flate_encode")
        if r < 0:
            raise M4GError(f"m4g_batch_flate_encode: stream_cap too small, need {-r - STREAM_NEEDED_BIAS}")
        return ((int(r) if out is not None else buf[:r].tobytes()), [float(ms[k]) for k in range(3)],
                kinds[:nchunks].copy())


FORMAT_BROTLI, FORMAT_FLATE, FORMAT_GZIP = 0, 1, 2


class BargainPlan:
    """matchfinder.Bargain1 / Bargain2 over independent streams (chunks), every stream cut into
    blocks of BlockSize bytes as matchfinder.Writer does (m4g_bargain_*).  Device resident."""

    def __init__(self, max_chunks: int, max_total_bytes: int, max_chunk_bytes: int, variant: int = 1,
                 MaxDistance: int = 1 << 20, Skip: bool = False, BlockSize: int = 1 << 16, device: int = 0):
        self.params = BargainParams(variant, MaxDistance, 1 if Skip else 0, BlockSize)
        h = ctypes.c_void_p()
        _check(lib().m4g_bargain_create(ctypes.byref(self.params), device, max_chunks, max_total_bytes,
                                        max_chunk_bytes, ctypes.byref(h)), "m4g_bargain_create")
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().m4g_bargain_destroy(self._h)
            except Exception:
                pass
            self._h = None

    def block_count(self, chunk_off) -> int:
        off = np.ascontiguousarray(chunk_off, dtype=np.uint64)
        return int(lib().m4g_bargain_block_count(self._h, off.ctypes.data, off.size - 1))

    def run(self, d_src, chunk_off, d_out, d_block_off, stream: int = 0) -> int:
        """d_src: uint8 tensor; d_out: tensor of 12-byte records; d_block_off: int64 tensor with
        block_count() + 1 elements.  Returns the number of Matches."""
        off = np.ascontiguousarray(chunk_off, dtype=np.uint64)
        out_cap = d_out.numel() * d_out.element_size() // 12
        rc = lib().m4g_bargain_run_device(self._h, d_src.data_ptr(), off.ctypes.data, off.size - 1,
                                          d_out.data_ptr(), out_cap, d_block_off.data_ptr(),
                                          stream if stream else None)
        _check(rc, "m4g_bargain_run_device")
        if rc < 0:
            raise M4GError(f"m4g_bargain_run_device: out_cap too small, {-rc - STREAM_NEEDED_BIAS} records needed")
        return int(rc)

    def last_timings(self) -> List[float]:
        ms = (ctypes.c_float * 4)()
        n = lib().m4g_bargain_last_timings(self._h, ms, 4)
        return [float(ms[k]) for k in range(n)]


class Bargain:
    """``matchfinder.Bargain1`` / ``Bargain2`` / ``Bargain3`` (variant 1 / 2 / 3; bargain1.go:15-31) as a
    ``matchfinder.MatchFinder`` on an MI355X: ``FindMatches(dst, src)`` / ``Reset()`` (m4g_bargain_ctx_*).  The blocks
    of a stream must be ``BlockSize`` bytes each, a shorter one only as the last before ``Reset`` -- the shape
    ``matchfinder.Writer{BlockSize}`` produces; ``find_matches_blocks`` is the same for many blocks in one launch."""

    def __init__(self, variant: int = 1, MaxDistance: int = 0, Skip: bool = False, BlockSize: int = 1 << 16,
                 device: int = 0):
        self.variant, self.MaxDistance, self.Skip, self.BlockSize, self.device = variant, MaxDistance, Skip, BlockSize, device
        h = ctypes.c_void_p()
        P = BargainParams(variant, MaxDistance, 1 if Skip else 0, BlockSize)
        _check(lib().m4g_bargain_ctx_create(ctypes.byref(P), device, ctypes.byref(h)), "m4g_bargain_ctx_create")
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib().m4g_bargain_ctx_destroy(self._h)
            self._h = None

    def Reset(self):
        _check(lib().m4g_bargain_ctx_reset(self._h), "m4g_bargain_ctx_reset")

    def find_matches_blocks(self, src, block_size: "int | None" = None):
        """FindMatches on every consecutive block of src, continuing the stream: (matches[M,3], block_off)."""
        a = _as_u8(src)
        bs = self.BlockSize or (1 << 16)
        if block_size not in (None, bs):
            raise M4GError("Bargain: the block size is fixed when the finder is made (BlockSize)")
        nb = (a.size + bs - 1) // bs
        cap = a.size // 3 + 2 * nb + 8
        out = np.empty((cap, 3), dtype=np.uint32)
        off = np.zeros(nb + 1, dtype=np.uint64)
        r = lib().m4g_bargain_ctx_find_blocks(self._h, a.ctypes.data if a.size else None, a.size, out.ctypes.data, cap,
                                              off.ctypes.data)
        _check(r, "m4g_bargain_ctx_find_blocks")
        if r < 0:
            raise M4GError(f"m4g_bargain_ctx_find_blocks: out_cap too small, need {-r - STREAM_NEEDED_BIAS}")
        return out[:r], off

    def FindMatches(self, dst: List[Match], src) -> List[Match]:
        """matchfinder.MatchFinder.FindMatches: append to dst, return dst (one block)."""
        m, _ = self.find_matches_blocks(src)
        dst.extend((int(u), int(l), int(d)) for u, l, d in m.tolist())
        return dst


def bargain_find_blocks_multi(data, devices: Sequence[int], variant: int = 1, MaxDistance: int = 1 << 20,
                              Skip: bool = False, BlockSize: int = 1 << 16):
    """m4g_bargain_find_blocks_multi: bargain_find_blocks with the blocks spread over `devices`."""
    a = _as_u8(data)
    P = BargainParams(variant, MaxDistance, 1 if Skip else 0, BlockSize)
    bs = BlockSize or (1 << 16)
    nb = (a.size + bs - 1) // bs
    cap = a.size // 3 + 2 * nb + 8
    out = np.empty((cap, 3), dtype=np.uint32)
    off = np.zeros(nb + 1, dtype=np.uint64)
    devs = (ctypes.c_int * len(devices))(*devices)
    r = lib().m4g_bargain_find_blocks_multi(ctypes.byref(P), devs, len(devices), a.ctypes.data if a.size else None, a.size,
                                            out.ctypes.data, cap, off.ctypes.data)
    _check(r, "m4g_bargain_find_blocks_multi")
    if r < 0:
        raise M4GError(f"m4g_bargain_find_blocks_multi: out_cap too small, need {-r - STREAM_NEEDED_BIAS}")
    return out[:r], off


def bargain_find_blocks(data, variant: int = 1, MaxDistance: int = 1 << 20, Skip: bool = False,
                        BlockSize: int = 1 << 16, device: int = 0):
    """The FindMatches calls of matchfinder.Writer{MatchFinder: &Bargain1{...} | &Bargain2{...}, BlockSize}
    over `data` (one Write, then Close), on the GPU.  Returns (matches (M, 3) uint32, block_off)."""
    a = _as_u8(data)
    n = a.size
    P = BargainParams(variant, MaxDistance, 1 if Skip else 0, BlockSize)
    bs = BlockSize or (1 << 16)
    nb = (n + bs - 1) // bs
    cap = n // 3 + 2 * nb + 8
    out = np.empty((cap, 3), dtype=np.uint32)
    off = np.zeros(nb + 1, dtype=np.uint64)
    rc = lib().m4g_bargain_find_blocks(ctypes.byref(P), device, a.ctypes.data if n else None, n,
                                       out.ctypes.data, cap, off.ctypes.data)
    _check(rc, "m4g_bargain_find_blocks")
    if rc < 0:
        raise M4GError("m4g_bargain_find_blocks: out_cap too small")
    return out[:rc].copy(), off


class GPUWriter:
    """``matchfinder.Writer`` (matchfinder/matchfinder.go:44-103) with BOTH plug-ins on the GPU (m4g_writer):
    MatchFinder = matchfinder.M4{**m4_fields}, Encoder = brotli.Encoder / flate.NewEncoder() / flate.NewGZIPEncoder().
    ``Dest`` needs a ``write`` method; Write / Close / Reset as in the reference."""

    def __init__(self, Dest, format: int = FORMAT_BROTLI, BlockSize: int = 1 << 16, mtime: int = 0, device: int = 0,
                 **m4_fields):
        self.Dest = Dest
        self.BlockSize = BlockSize
        h = ctypes.c_void_p()
        p = _params(m4_fields.get("MaxDistance", 0), m4_fields.get("MinLength", 0), m4_fields.get("HashLen", 0),
                    m4_fields.get("TableBits", 0), m4_fields.get("ChainLength", 0), m4_fields.get("DistanceBitCost", 0))
        _check(lib().m4g_writer_create(ctypes.byref(p), device, format, BlockSize, mtime, ctypes.byref(h)),
               "m4g_writer_create")
        self._h = h
        self._buf = np.zeros(1 << 20, dtype=np.uint8)

    def _emit(self, call):
        r = call(self._buf)
        if r < -STREAM_NEEDED_BIAS:  # the output is kept inside: fetch it with a buffer that is large enough
            self._buf = np.zeros(-r - STREAM_NEEDED_BIAS + 4096, dtype=np.uint8)
            r = call(self._buf, again=True)
        _check(r, "m4g_writer")
        if r > 0:
            self.Dest.write(self._buf[:r].tobytes())

    def Write(self, p) -> int:
        a = _as_u8(p)

        def call(buf, again=False):
            return lib().m4g_writer_write(self._h, None if again or not a.size else a.ctypes.data, 0 if again else a.size,
                                          buf.ctypes.data, buf.size)
        self._emit(call)
        return a.size

    def Close(self):
        self._emit(lambda buf, again=False: lib().m4g_writer_close(self._h, buf.ctypes.data, buf.size))

    def Reset(self, newDest):
        _check(lib().m4g_writer_reset(self._h), "m4g_writer_reset")
        self.Dest = newDest

    def __del__(self):
        if getattr(self, "_h", None):
            lib().m4g_writer_destroy(self._h)
            self._h = None


class NewWriterV2(GPUWriter):
    """``brotli.NewWriterV2(dst, level)`` for levels 5-9 (writer.go:126-161): the matchfinder.Writer with Bargain1 /
    Bargain2 / Bargain3 and brotli.Encoder, both on the GPU (m4g_writer_create_v2).  Write / Close / Reset as in the
    reference; same bytes after the same calls."""

    def __init__(self, Dest, level: int, device: int = 0):
        self.Dest = Dest
        self.BlockSize = 1 << 16
        h = ctypes.c_void_p()
        _check(lib().m4g_writer_create_v2(level, device, ctypes.byref(h)), "m4g_writer_create_v2")
        self._h = h
        self._buf = np.zeros(1 << 20, dtype=np.uint8)


def brotli_compress(data, block_size: int = 1 << 16, device: int = 0, devices: "Sequence[int] | None" = None,
                    **m4_fields) -> bytes:
    """matchfinder.Writer{MatchFinder: &matchfinder.M4{**m4_fields}, Encoder: &brotli.Encoder{}, BlockSize}:
    w.Write(data); w.Close() -- the writer of the reference's tests (brotli_test.go:626-649, 730-744) -- with
    match finding (m4g_find_matches_blocks) and both passes of Encoder.Encode (m4g_brotli_encode_blocks) on the
    GPU.  Returns the brotli stream."""
    a = _as_u8(data)
    if devices:  # the stream's blocks over several devices (m4g_find_matches_blocks_multi), the encoder on the first
        device = devices[0]
        matches, off = find_matches_blocks_multi(a, block_size, devices, **m4_fields)
    else:
        matches, off = M4(device=device, **m4_fields).find_matches_blocks(a, block_size)
    cap = a.size + (a.size // block_size + 2) * 1024 + 64
    buf = np.zeros(cap, dtype=np.uint8)
    m = np.ascontiguousarray(matches, dtype=np.uint32)
    r = lib().m4g_brotli_encode_blocks(device, a.ctypes.data if a.size else None, a.size, block_size,
                                       m.ctypes.data if m.size else None, off.ctypes.data, buf.ctypes.data, cap)
    _check(r, "m4g_brotli_encode_blocks")
    if r < 0:
        raise M4GError(f"m4g_brotli_encode_blocks: stream_cap too small, need {-r - STREAM_NEEDED_BIAS}")
    return buf[:r].tobytes()


def writer_v2_match_finder(level: int) -> dict:
    """The MatchFinder brotli.NewWriterV2(dst, level) builds for the levels that use Bargain1 / Bargain2
    (writer.go:142-150), as keyword arguments of bargain_find_blocks / BargainPlan."""
    table = {5: (1, False), 6: (1, False), 7: (2, True), 8: (2, False), 9: (3, False)}
    if level not in table:
        raise M4GError(f"NewWriterV2 level {level}: levels 5-9 (Bargain1 / Bargain2 / Bargain3) are built")
    v, skip = table[level]
    return {"variant": v, "MaxDistance": 1 << 20, "Skip": skip, "BlockSize": 1 << 16}


def brotli_compress_v2(data, level: int = 5, device: int = 0, devices: "Sequence[int] | None" = None) -> bytes:
    """w := brotli.NewWriterV2(dst, level); w.Write(data); w.Close() for level 5-9 (writer.go:126-161):
    matchfinder.Writer{Bargain1 | Bargain2 | Bargain3, &Encoder{}, BlockSize 1 << 16} with the match finder and both
    passes of Encoder.Encode on the GPU (m4g_brotli_compress_v2)."""
    a = _as_u8(data)
    cap = a.size + (a.size // (1 << 16) + 2) * 1024 + 64
    buf = np.zeros(cap, dtype=np.uint8)
    if devices:  # the stream's blocks over several devices (m4g_bargain_find_blocks_multi), the encoder on the first
        m, off = bargain_find_blocks_multi(a, devices, **writer_v2_match_finder(min(level, 9)))
        m = np.ascontiguousarray(m, dtype=np.uint32)
        r = lib().m4g_brotli_encode_blocks(devices[0], a.ctypes.data if a.size else None, a.size, 1 << 16,
                                           m.ctypes.data if m.size else None, off.ctypes.data, buf.ctypes.data, cap)
        _check(r, "m4g_brotli_encode_blocks")
        return buf[:r].tobytes()
    r = lib().m4g_brotli_compress_v2(level, device, a.ctypes.data if a.size else None, a.size, buf.ctypes.data, cap)
    _check(r, "m4g_brotli_compress_v2")
    if r < 0:
        raise M4GError(f"m4g_brotli_compress_v2: out_cap too small, need {-r - STREAM_NEEDED_BIAS}")
    return buf[:r].tobytes()


def flate_compress(data, level: int = 6, gzip: bool = False, mtime: int = 0, block_size: int = 1 << 16,
                   device: int = 0, devices: "Sequence[int] | None" = None) -> bytes:
    """flate.NewWriter(dst, level) / flate.NewGZIPWriter(dst, level) for the levels whose match finder is
    matchfinder.M4 (5..7; flate/writer.go:23-66): w.Write(data); w.Close().  Match finding for all blocks in
    one launch (m4g_find_matches_blocks: the Match lists of the writer's block-after-block FindMatches calls,
    history carried over), then the encoder on the GPU (m4g_flate_encode_blocks).  `mtime` stands in for the
    time.Now() of the gzip header (flate/gzip.go:46)."""
    a = _as_u8(data)
    if devices:  # (as brotli_compress)
        device = devices[0]
        matches, off = find_matches_blocks_multi(a, block_size, devices, **flate_m4_params(level))
    else:
        matches, off = M4(device=device, **flate_m4_params(level)).find_matches_blocks(a, block_size)
    cap = a.size + (a.size // block_size + 2) * 16 + 64
    buf = np.zeros(cap, dtype=np.uint8)
    m = np.ascontiguousarray(matches, dtype=np.uint32)
    r = lib().m4g_flate_encode_blocks(device, a.ctypes.data if a.size else None, a.size, block_size,
                                      m.ctypes.data if m.size else None, off.ctypes.data, 1 if gzip else 0, mtime,
                                      buf.ctypes.data, cap, None)
    _check(r, "m4g_flate_encode_blocks")
    if r < 0:
        raise M4GError(f"m4g_flate_encode_blocks: stream_cap too small, need {-r - STREAM_NEEDED_BIAS}")
    return buf[:r].tobytes()


def flate_m4_params(level: int) -> dict:
    """The M4 fields flate.NewWriter / NewGZIPWriter use at levels 5..7 (flate/writer.go:33-46)."""
    if not 5 <= level <= 7:
        raise ValueError("flate levels 5..7 are the ones that use matchfinder.M4")
    return dict(MaxDistance=1 << 15, ChainLength={5: 8, 6: 16, 7: 32}[level], HashLen=5, DistanceBitCost=66)


def find_matches_blocks_multi(src, block_size: int, devices: Sequence[int] = (0,), **m4_fields):
    """m4g_find_matches_blocks_multi: a fresh M4's FindMatches over every consecutive block of src, the blocks
    spread over `devices` (one stream over several GPUs); returns (matches[M,3], block_off[nblocks+1])."""
    a = _as_u8(src)
    p = _params(m4_fields.get("MaxDistance", 0), m4_fields.get("MinLength", 0), m4_fields.get("HashLen", 0),
                m4_fields.get("TableBits", 0), m4_fields.get("ChainLength", 0), m4_fields.get("DistanceBitCost", 0))
    min_len = p.min_length or 4
    nblocks = (a.size + block_size - 1) // block_size
    cap = sum(min(block_size, a.size - b * block_size) // min_len + 2 for b in range(nblocks)) + 1
    out = np.empty((cap, 3), dtype=np.uint32)
    off = np.zeros(nblocks + 1, dtype=np.uint64)
    devs = (ctypes.c_int * len(devices))(*devices)
    r = lib().m4g_find_matches_blocks_multi(ctypes.byref(p), devs, len(devices), a.ctypes.data if a.size else None,
                                            a.size, block_size, out.ctypes.data, cap, off.ctypes.data)
    _check(r, "m4g_find_matches_blocks_multi")
    if r < 0:
        raise M4GError(f"m4g_find_matches_blocks_multi: out_cap too small, need {-r - STREAM_NEEDED_BIAS}")
    return out[:r], off


def span_plan(n: int, block_size: int, ndevices: int, **m4_fields):
    """m4g_debug_span_plan: (pieces[k, 4] = device slot, first block, end block, first byte; origins[nblocks])."""
    p = _params(m4_fields.get("MaxDistance", 0), m4_fields.get("MinLength", 0), m4_fields.get("HashLen", 0),
                m4_fields.get("TableBits", 0), m4_fields.get("ChainLength", 0), m4_fields.get("DistanceBitCost", 0))
    nblocks = (n + block_size - 1) // block_size
    org = np.zeros(max(nblocks, 1), dtype=np.uint64)
    k = lib().m4g_debug_span_plan(ctypes.byref(p), n, block_size, ndevices, None, 0, org.ctypes.data)
    _check(k, "m4g_debug_span_plan")
    pieces = np.zeros((max(k, 1), 4), dtype=np.uint64)
    lib().m4g_debug_span_plan(ctypes.byref(p), n, block_size, ndevices, pieces.ctypes.data, k, None)
    return pieces[:k], org[:nblocks]


def noalloc_poison(value: int = 0xFF, device: int = 0, stream: int = 0):
    """Test hook: fill the no-alloc entry point's scratch array on `device` with a byte value."""
    _check(lib().m4g_debug_noalloc_poison(device, value, stream), "m4g_debug_noalloc_poison")


def run_device_noalloc(d_src, chunk_off: np.ndarray, d_out, d_out_off, d_histos=None, stream: int = 0,
                       device: int = 0, **m4_fields):
    """m4g_batch_run_device_noalloc: BatchPlan.run without a plan -- the workspace
    is a fixed array inside libm4g.so, the call allocates nothing (limits:
    include/m4g.h M4G_NOALLOC_*)."""
    p = _params(m4_fields.get("MaxDistance", 0), m4_fields.get("MinLength", 0),
                m4_fields.get("HashLen", 0), m4_fields.get("TableBits", 0),
                m4_fields.get("ChainLength", 0), m4_fields.get("DistanceBitCost", 0))
    off = np.ascontiguousarray(chunk_off, dtype=np.uint64)
    out_cap = d_out.numel() * d_out.element_size() // 12
    rc = lib().m4g_batch_run_device_noalloc(
        ctypes.byref(p), device, d_src.data_ptr(), off.ctypes.data, off.size - 1, d_out.data_ptr(),
        d_out_off.data_ptr(), out_cap, d_histos.data_ptr() if d_histos is not None else None,
        stream if stream else None)
    _check(rc, "m4g_batch_run_device_noalloc")


class PinnedBuffer:
    """Page-locked host memory from m4g_host_alloc, as a numpy array (`.array`).
    Copies between such a buffer and the GPU run at PCIe speed; pageable memory
    (numpy's own) is staged by the HIP runtime at a fraction of it, and a fresh
    numpy output array pays a page fault per 4 KiB on top."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        self._ptr = lib().m4g_host_alloc(self.nbytes)
        if not self._ptr:
            raise M4GError("m4g_host_alloc failed")
        buf = (ctypes.c_uint8 * self.nbytes).from_address(self._ptr)
        self.array = np.frombuffer(buf, dtype=np.uint8)

    def close(self):
        if self._ptr:
            self.array = None
            lib().m4g_host_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostBatch:
    """m4g_hostbatch_*: find_matches_batch on a persistent handle (plans, device
    buffers, pinned staging and streams created once; one host thread per device)."""

    def __init__(self, max_chunks: int, max_chunk_bytes: int, devices: Sequence[int] = (0,), **m4_fields):
        self.params = _params(m4_fields.get("MaxDistance", 0), m4_fields.get("MinLength", 0),
                              m4_fields.get("HashLen", 0), m4_fields.get("TableBits", 0),
                              m4_fields.get("ChainLength", 0), m4_fields.get("DistanceBitCost", 0))
        devs = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_void_p()
        _check(lib().m4g_hostbatch_create(ctypes.byref(self.params), devs, len(devices), max_chunks,
                                          max_chunk_bytes, ctypes.byref(h)), "m4g_hostbatch_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().m4g_hostbatch_destroy(self._h)
            self._h = None

    __del__ = close

    def run(self, base, chunk_off: Sequence[int], want_histos: bool = False, out: "np.ndarray | None" = None):
        """Returns (matches[M,3], out_off[nchunks+1], histos or None), as find_matches_batch."""
        a = _as_u8(base)
        off = np.ascontiguousarray(chunk_off, dtype=np.uint64)
        n = off.size - 1
        min_len = self.params.min_length or 4
        if out is None:
            cap = int(sum(int(off[c + 1] - off[c]) // min_len + 2 for c in range(n))) + 1
            out = np.empty((cap, 3), dtype=np.uint32)
        else:
            assert out.dtype == np.uint32 and out.flags.c_contiguous
            out = out.reshape(-1)[:out.size // 3 * 3].reshape(-1, 3)
            cap = out.shape[0]
        out_off = np.zeros(n + 1, dtype=np.uint64)
        histos = np.zeros((max(n, 1), 1024), dtype=np.uint32) if want_histos else None
        r = lib().m4g_hostbatch_run(self._h, a.ctypes.data if a.size else None, off.ctypes.data, n, out.ctypes.data,
                                    out_off.ctypes.data, cap, histos.ctypes.data if want_histos else None)
        _check(r, "m4g_hostbatch_run")
        if r < 0:
            raise M4GError(f"m4g_hostbatch_run: out_cap too small, need {-r - STREAM_NEEDED_BIAS}")
        return out[:int(out_off[n])], out_off, (histos[:n] if want_histos else None)


    def compress(self, base, chunk_off: Sequence[int], format: int = 0, mtime: int = 0,
                 out: "np.ndarray | None" = None):
        """m4g_hostbatch_compress: the chunks as ONE brotli (format 0) / deflate (1) / gzip (2) stream, one
        meta-block / deflate block per chunk (fresh M4 per chunk); Match lists stay on the device.  Returns the
        stream bytes, or with `out` (a uint8 array, e.g. PinnedBuffer.array) its length."""
        a = _as_u8(base)
        off = np.ascontiguousarray(chunk_off, dtype=np.uint64)
        n = off.size - 1
        buf = out if out is not None else np.zeros(int(a.size + a.size // 4 + (n + 1) * 1024 + 256), dtype=np.uint8)
        r = lib().m4g_hostbatch_compress(self._h, format, mtime, a.ctypes.data if a.size else None, off.ctypes.data, n,
                                         buf.ctypes.data, buf.size)
        _check(r, "m4g_hostbatch_compress")
        if r < 0:
            raise M4GError(f"m4g_hostbatch_compress: stream_cap too small, {-r - STREAM_NEEDED_BIAS} always suffices")
        return int(r) if out is not None else buf[:r].tobytes()


def find_matches_batch(base, chunk_off: Sequence[int], devices: Sequence[int] = (0,),
                       want_histos: bool = False, out: "np.ndarray | None" = None, **m4_fields):
    """Host-memory batch (m4g_find_matches_batch): returns (matches[M,3],
    out_off[nchunks+1], histos[nchunks,1024] or None).  `out`: optional
    caller-provided uint32 array for the matches (e.g. a view of a PinnedBuffer)."""
    a = _as_u8(base)
    off = np.ascontiguousarray(chunk_off, dtype=np.uint64)
    n = off.size - 1
    p = _params(m4_fields.get("MaxDistance", 0), m4_fields.get("MinLength", 0),
                m4_fields.get("HashLen", 0), m4_fields.get("TableBits", 0),
                m4_fields.get("ChainLength", 0), m4_fields.get("DistanceBitCost", 0))
    min_len = p.min_length or 4
    cap = int(sum(int(off[c + 1] - off[c]) // min_len + 2 for c in range(n))) + 1
    if out is None:
        out = np.empty((cap, 3), dtype=np.uint32)
    else:
        assert out.dtype == np.uint32 and out.flags.c_contiguous
        out = out.reshape(-1)[:out.size // 3 * 3].reshape(-1, 3)
        cap = out.shape[0]
    out_off = np.zeros(n + 1, dtype=np.uint64)
    histos = np.zeros((max(n, 1), 1024), dtype=np.uint32) if want_histos else None
    devs = (ctypes.c_int * len(devices))(*devices)
    r = lib().m4g_find_matches_batch(ctypes.byref(p), a.ctypes.data if a.size else None,
                                     off.ctypes.data, n, devs, len(devices), out.ctypes.data,
                                     out_off.ctypes.data, cap,
                                     histos.ctypes.data if want_histos else None)
    _check(r, "m4g_find_matches_batch")
    if r < 0:
        raise M4GError(f"m4g_find_matches_batch: out_cap too small, need {-r - STREAM_NEEDED_BIAS}")
    total = int(out_off[n])
    return out[:total], out_off, (histos[:n] if want_histos else None)
