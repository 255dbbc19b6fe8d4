"""ctypes access to the CPU oracle (oracle/libm4oracle.so) -- checker only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package (brotli_amd/) never does.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

INT64_MIN = -(1 << 63)


class HistoStruct(ctypes.Structure):
    _fields_ = [("lit", ctypes.c_uint32 * 256),
                ("cmd", ctypes.c_uint32 * 704),
                ("dist", ctypes.c_uint32 * 64)]


def lib():
    global _LIB
    if _LIB is None:
        # M4G_ORACLE_SO: an instrumented build of the same sources (tools/oracle_coverage.py)
        path = os.environ.get("M4G_ORACLE_SO") or os.path.join(_ROOT, "oracle", "libm4oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run __graft_entry__.build() (or `make -C oracle`)")
        L = ctypes.CDLL(path)
        L.m4ref_new.argtypes = [ctypes.c_int64] * 6
        L.m4ref_new.restype = ctypes.c_void_p
        L.m4ref_free.argtypes = [ctypes.c_void_p]
        L.m4ref_reset.argtypes = [ctypes.c_void_p]
        L.m4ref_find_matches_flat.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                              ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t]
        L.m4ref_find_matches_flat.restype = ctypes.c_int64
        L.m4ref_dump_table.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.m4ref_dump_table.restype = ctypes.c_size_t
        L.m4ref_dump_chain.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.m4ref_dump_chain.restype = ctypes.c_size_t
        L.m4ref_history_len.argtypes = [ctypes.c_void_p]
        L.m4ref_history_len.restype = ctypes.c_size_t
        L.encref_histograms.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                        ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p]
        L.encref_compress_m4.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t),
                                         ctypes.POINTER(ctypes.c_size_t)]
        L.encref_compress_m4.restype = ctypes.c_void_p
        L.encref_free_buf.argtypes = [ctypes.c_void_p]
        L.encref_encoder_new.restype = ctypes.c_void_p
        L.encref_encoder_free.argtypes = [ctypes.c_void_p]
        L.encref_encoder_reset.argtypes = [ctypes.c_void_p]
        L.encref_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                    ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        L.encref_encoder_take.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        L.encref_encoder_take.restype = ctypes.c_void_p
        L.encref_insert_length_code.argtypes = [ctypes.c_uint32]
        L.encref_insert_length_code.restype = ctypes.c_uint16
        L.encref_copy_length_code.argtypes = [ctypes.c_uint32]
        L.encref_copy_length_code.restype = ctypes.c_uint16
        L.encref_combine_length_codes.argtypes = [ctypes.c_uint16, ctypes.c_uint16, ctypes.c_int]
        L.encref_combine_length_codes.restype = ctypes.c_uint16
        # flate / gzip encoders (oracle/flate_ref.c)
        L.flateref_length_code.argtypes = [ctypes.c_uint32]
        L.flateref_offset_code.argtypes = [ctypes.c_uint32]
        L.flateref_generate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.flateref_statistics.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                          ctypes.c_void_p, ctypes.c_void_p]
        L.flateref_encoder_new.argtypes = [ctypes.c_int, ctypes.c_uint32]
        L.flateref_encoder_new.restype = ctypes.c_void_p
        L.flateref_encoder_free.argtypes = [ctypes.c_void_p]
        L.flateref_encoder_reset.argtypes = [ctypes.c_void_p]
        L.flateref_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.flateref_encoder_take.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        L.flateref_encoder_take.restype = ctypes.c_void_p
        L.flateref_compress_m4.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                           ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(ctypes.c_size_t)]
        L.flateref_compress_m4.restype = ctypes.c_void_p
        L.flateref_free_buf.argtypes = [ctypes.c_void_p]
        L.flateref_crc32.argtypes = [ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t]
        L.flateref_crc32.restype = ctypes.c_uint32
        # oracle/bargain_ref.h
        L.bargainref_new.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int]
        L.bargainref_new.restype = ctypes.c_void_p
        L.bargainref_free.argtypes = [ctypes.c_void_p]
        L.bargainref_reset.argtypes = [ctypes.c_void_p]
        L.bargainref_find_matches_flat.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                                   ctypes.c_void_p, ctypes.c_size_t]
        L.bargainref_find_matches_flat.restype = ctypes.c_int64
        L.bargainref_find_blocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                             ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.bargainref_find_blocks.restype = ctypes.c_int64
        L.bargainref_byte_cost_bits.argtypes = [ctypes.c_uint64, ctypes.c_uint32]
        L.bargainref_byte_cost_bits.restype = ctypes.c_uint32
        L.bargainref_log2.argtypes = [ctypes.c_double]
        L.bargainref_log2.restype = ctypes.c_double
        _LIB = L
    return _LIB


def _buf(data):
    """bytes / bytearray / uint8 ndarray -> (address, length, keepalive)."""
    if isinstance(data, np.ndarray):
        a = np.ascontiguousarray(data, dtype=np.uint8)
    else:
        a = np.frombuffer(bytes(data), dtype=np.uint8)
    return (a.ctypes.data if a.size else None), a.size, a


class OracleM4:
    """matchfinder.M4 restated in C (oracle/m4_ref.c); same field names."""

    def __init__(self, MaxDistance=0, MinLength=0, HashLen=0, TableBits=0, ChainLength=0,
                 DistanceBitCost=0):
        self.params = (MaxDistance, MinLength, HashLen, TableBits, ChainLength, DistanceBitCost)
        self._h = lib().m4ref_new(*self.params)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().m4ref_free(self._h)
            self._h = None

    def Reset(self):
        lib().m4ref_reset(self._h)

    def FindMatches(self, src, seed_prev_distance: int = 0) -> np.ndarray:
        """Returns an (M, 3) uint32 array of (Unmatched, Length, Distance)."""
        addr, n, keep = _buf(src)
        cap = n // 4 + 8
        out = np.empty((cap, 3), dtype=np.uint32)
        r = lib().m4ref_find_matches_flat(self._h, addr, n, seed_prev_distance,
                                          out.ctypes.data, cap)
        if r == INT64_MIN:
            raise IndexError("oracle: repeat distance out of range (Go would panic)")
        if r < 0:
            cap = -r
            raise RuntimeError("oracle: out_cap too small (state already advanced)")
        return out[:r].copy()

    def history_len(self) -> int:
        return int(lib().m4ref_history_len(self._h))

    def dump_table(self) -> np.ndarray:
        tb = self.params[3] or 17
        out = np.zeros(1 << tb, dtype=np.uint32)
        lib().m4ref_dump_table(self._h, out.ctypes.data, out.size)
        return out

    def dump_chain(self) -> np.ndarray:
        n = lib().m4ref_history_len(self._h)
        out = np.zeros(n, dtype=np.uint32)
        k = lib().m4ref_dump_chain(self._h, out.ctypes.data, n)
        return out[:k]

    def compress(self, data, block_size: int = 1 << 16):
        """matchfinder.Writer{M4, brotli.Encoder, BlockSize}: returns (stream, nmatches)."""
        addr, n, keep = _buf(data)
        osz = ctypes.c_size_t()
        nm = ctypes.c_size_t()
        p = lib().encref_compress_m4(self._h, addr, n, block_size, ctypes.byref(osz),
                                     ctypes.byref(nm))
        out = ctypes.string_at(p, osz.value)
        lib().encref_free_buf(p)
        return out, nm.value


class OracleBargain:
    """matchfinder.Bargain1 / Bargain2 / Bargain3 restated in C (oracle/bargain_ref.c)."""

    def __init__(self, variant: int = 1, MaxDistance: int = 0, Skip: bool = False):
        self.variant, self.MaxDistance, self.Skip = variant, MaxDistance, Skip
        self._h = lib().bargainref_new(variant, MaxDistance, 1 if Skip else 0)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().bargainref_free(self._h)
            self._h = None

    def Reset(self):
        lib().bargainref_reset(self._h)

    def FindMatches(self, src) -> np.ndarray:
        addr, n, keep = _buf(src)
        cap = n // 3 + 8
        out = np.empty((cap, 3), dtype=np.uint32)
        r = lib().bargainref_find_matches_flat(self._h, addr, n, out.ctypes.data, cap)
        if r < 0:
            raise RuntimeError("oracle: out_cap too small")
        return out[:r].copy()

    def find_blocks(self, data, block_size: int = 1 << 16):
        """The Writer's loop (matchfinder.go:59-93): one FindMatches per block.  Returns (matches, block_off)."""
        addr, n, keep = _buf(data)
        nb = (n + block_size - 1) // block_size
        cap = n // 3 + 8 + nb
        out = np.empty((cap, 3), dtype=np.uint32)
        off = np.zeros(nb + 1, dtype=np.uint64)
        r = lib().bargainref_find_blocks(self._h, addr, n, block_size, out.ctypes.data, cap, off.ctypes.data)
        if r < 0:
            raise RuntimeError("oracle: out_cap too small")
        return out[:r].copy(), off


def flate_m4_params(level: int) -> dict:
    """flate.newWriter's match finder for the levels that use matchfinder.M4 (flate/writer.go:33-46)."""
    if not 5 <= level <= 7:
        raise ValueError("flate levels 5..7 use matchfinder.M4")
    return dict(MaxDistance=1 << 15, ChainLength={5: 8, 6: 16, 7: 32}[level], HashLen=5, DistanceBitCost=66)


class OracleFlateEncoder:
    """flate.NewEncoder() / flate.NewGZIPEncoder() restated (oracle/flate_ref.c)."""

    KINDS = ("stored", "fixed", "dynamic")

    def __init__(self, gzip: bool = False, mtime: int = 0):
        self._h = lib().flateref_encoder_new(1 if gzip else 0, mtime)
        self.last_kind = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().flateref_encoder_free(self._h)
            self._h = None

    def Reset(self):
        lib().flateref_encoder_reset(self._h)

    def Encode(self, src, matches: np.ndarray, last_block: bool, nil_src: bool = False) -> bytes:
        addr, n, keep = _buf(src)
        m = np.ascontiguousarray(matches, dtype=np.uint32).reshape(-1, 3)
        kind = ctypes.c_int(-1)
        rc = lib().flateref_encode(self._h, addr, n, 1 if nil_src else 0, m.ctypes.data if len(m) else None, len(m),
                                   1 if last_block else 0, ctypes.byref(kind))
        if rc:
            raise ValueError("flate: match too long or distance too high (the reference panics, token.go:59-71)")
        self.last_kind = self.KINDS[kind.value]
        sz = ctypes.c_size_t()
        p = lib().flateref_encoder_take(self._h, ctypes.byref(sz))
        return ctypes.string_at(p, sz.value) if sz.value else b""


def flate_compress(m4: "OracleM4", data, block_size: int = 1 << 16, gzip: bool = False, mtime: int = 0) -> bytes:
    """matchfinder.Writer{M4, flate encoder, BlockSize}: Write(data); Close()."""
    addr, n, keep = _buf(data)
    osz = ctypes.c_size_t()
    p = lib().flateref_compress_m4(m4._h, addr, n, block_size, 1 if gzip else 0, mtime, ctypes.byref(osz))
    if not p:
        raise ValueError("flate: match too long or distance too high (the reference panics)")
    out = ctypes.string_at(p, osz.value)
    lib().flateref_free_buf(p)
    return out


def flate_statistics(src, matches: np.ndarray):
    """makeStatistics' frequencies (huffman_bit_writer.go:425-446): (lit[286], off[30]) as int32."""
    addr, n, keep = _buf(src)
    m = np.ascontiguousarray(matches, dtype=np.uint32).reshape(-1, 3)
    lit = np.zeros(286, dtype=np.int32)
    off = np.zeros(30, dtype=np.int32)
    if lib().flateref_statistics(addr, n, m.ctypes.data if len(m) else None, len(m), lit.ctypes.data, off.ctypes.data):
        raise ValueError("flate: match too long or distance too high")
    return lit, off


def flate_generate(freq, max_bits: int):
    """huffmanEncoder.generate (huffman_code.go:271-309): (codes, lengths) as uint16."""
    f = np.ascontiguousarray(freq, dtype=np.int32)
    code = np.zeros(len(f), dtype=np.uint16)
    ln = np.zeros(len(f), dtype=np.uint16)
    lib().flateref_generate(f.ctypes.data, len(f), max_bits, code.ctypes.data, ln.ctypes.data)
    return code, ln


def histograms(src, matches: np.ndarray):
    """Pass 1 of brotli.Encoder.Encode (encoder.go:44-105).
    Returns (lit[256], cmd[704], dist[64], dist_codes[M,3], cmds[M])."""
    addr, n, keep = _buf(src)
    m = np.ascontiguousarray(matches, dtype=np.uint32).reshape(-1, 3)
    h = HistoStruct()
    dc = np.zeros((max(len(m), 1), 3), dtype=np.uint32)
    cm = np.zeros(max(len(m), 1), dtype=np.uint16)
    lib().encref_histograms(addr, n, m.ctypes.data if len(m) else None, len(m),
                            ctypes.byref(h), dc.ctypes.data, cm.ctypes.data)
    return (np.array(h.lit, dtype=np.uint32), np.array(h.cmd, dtype=np.uint32),
            np.array(h.dist, dtype=np.uint32), dc[:len(m)], cm[:len(m)])


class OracleEncoder:
    """brotli.Encoder restated (oracle/encoder_ref.c)."""

    def __init__(self):
        self._h = lib().encref_encoder_new()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().encref_encoder_free(self._h)
            self._h = None

    def Reset(self):
        lib().encref_encoder_reset(self._h)

    def Encode(self, src, matches: np.ndarray, last_block: bool) -> bytes:
        addr, n, keep = _buf(src)
        m = np.ascontiguousarray(matches, dtype=np.uint32).reshape(-1, 3)
        lib().encref_encode(self._h, addr, n, m.ctypes.data if len(m) else None, len(m),
                            1 if last_block else 0)
        sz = ctypes.c_size_t()
        p = lib().encref_encoder_take(self._h, ctypes.byref(sz))
        return ctypes.string_at(p, sz.value) if sz.value else b""


# ---------------------------------------------------------------------------
# Stand-in for brotli.NewReader: the system's libbrotlidec (Google's C decoder).
_DEC = None


def brotli_decompress(stream: bytes, max_out: int) -> bytes:
    global _DEC
    if _DEC is None:
        for name in ("libbrotlidec.so.1", "libbrotlidec.so"):
            try:
                _DEC = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _DEC is None:
            raise RuntimeError("libbrotlidec not found")
        _DEC.BrotliDecoderDecompress.argtypes = [ctypes.c_size_t, ctypes.c_char_p,
                                                 ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]
        _DEC.BrotliDecoderDecompress.restype = ctypes.c_int
    out = ctypes.create_string_buffer(max_out + 16)
    sz = ctypes.c_size_t(max_out + 16)
    r = _DEC.BrotliDecoderDecompress(len(stream), stream, ctypes.byref(sz), out)
    if r != 1:
        raise ValueError(f"brotli decode failed (result {r})")
    return out.raw[:sz.value]


def replay(matches: np.ndarray, src: bytes, history: bytes = b"") -> bytes:
    """Format-free validity check: rebuild the block from its Match list."""
    buf = bytearray(history)
    base = len(history)
    pos = 0
    for u, l, d in np.asarray(matches).reshape(-1, 3).tolist():
        buf += src[pos:pos + u]
        pos += u
        if l:
            assert 0 < d <= len(buf), (u, l, d, len(buf))
            for _ in range(l):
                buf.append(buf[-d])
            pos += l
    return bytes(buf[base:])
