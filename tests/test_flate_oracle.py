"""The flate / gzip oracle (oracle/flate_ref.c) against what can pin it here.

The reference's own criterion (flate/flate_test.go:14-38, 44-90): the output of
matchfinder.Writer{M4 at the level's parameters, flate encoder, BlockSize 1<<16}
over Isaac.Newton-Opticks.txt inflates back to the file -- zlib / gzip of the
Python standard library stand in for Go's compress/flate and compress/gzip
readers.  Beyond that: the token tables against RFC 1951, the length-limited
codes against an independent package-merge (Kraft equality, same total cost),
and the three block kinds.  Byte identity with the Go encoder is unpinned (no Go
toolchain); the encoder has no unstable sort, so it is expected.
"""
import gzip
import heapq
import os
import struct
import zlib

import numpy as np
import pytest

from brotli_amd import synth

from tests import cases
from tests.oracle_lib import (OracleFlateEncoder, OracleM4, flate_compress, flate_generate, flate_m4_params,
                              flate_statistics, lib)

OPTICKS = os.path.join(os.path.dirname(__file__), "golden", "Isaac.Newton-Opticks.txt")


def inflate(raw: bytes) -> bytes:
    d = zlib.decompressobj(-15)
    out = d.decompress(raw) + d.flush()
    assert d.eof and not d.unused_data
    return out


# RFC 1951 section 3.2.5, written out independently of the oracle's tables
RFC_LEN = [(257 + i, e, b) for i, (e, b) in enumerate([
    (0, 3), (0, 4), (0, 5), (0, 6), (0, 7), (0, 8), (0, 9), (0, 10), (1, 11), (1, 13), (1, 15), (1, 17), (2, 19), (2, 23),
    (2, 27), (2, 31), (3, 35), (3, 43), (3, 51), (3, 59), (4, 67), (4, 83), (4, 99), (4, 115), (5, 131), (5, 163),
    (5, 195), (5, 227), (0, 258)])]
RFC_DIST = [(i, e, b) for i, (e, b) in enumerate([
    (0, 1), (0, 2), (0, 3), (0, 4), (1, 5), (1, 7), (2, 9), (2, 13), (3, 17), (3, 25), (4, 33), (4, 49), (5, 65), (5, 97),
    (6, 129), (6, 193), (7, 257), (7, 385), (8, 513), (8, 769), (9, 1025), (9, 1537), (10, 2049), (10, 3073),
    (11, 4097), (11, 6145), (12, 8193), (12, 12289), (13, 16385), (13, 24577)])]


def test_token_codes_follow_rfc1951():
    L = lib()
    for ln in range(3, 259):
        code = [c for c, e, b in RFC_LEN if b <= ln < b + (1 << e)]
        want = 285 if ln == 258 else min(code)  # 258 also falls into code 284's range; it has its own code
        assert 257 + L.flateref_length_code(ln) == want, ln
    for d in list(range(1, 2000)) + list(range(2000, 32769, 7)) + [32768]:
        want = [c for c, e, b in RFC_DIST if b <= d < b + (1 << e)]
        assert [L.flateref_offset_code(d)] == want, d
    assert L.flateref_length_code(259) == -1 and L.flateref_length_code(2) == -1
    assert L.flateref_offset_code(32769) == -1 and L.flateref_offset_code(0) == -1


def package_merge_cost(freq, max_bits):
    """Cost of an optimal length-limited prefix code (Larmore-Hirschberg), independent of the oracle."""
    items = sorted(f for f in freq if f > 0)
    n = len(items)
    if n <= 2:
        return sum(items)  # one bit each
    leaves = [(f, (i,)) for i, f in enumerate(items)]
    packages = list(leaves)
    for _ in range(max_bits - 1):
        paired = [(packages[k][0] + packages[k + 1][0], packages[k][1] + packages[k + 1][1])
                  for k in range(0, len(packages) - 1, 2)]
        packages = sorted(leaves + paired, key=lambda x: x[0])
    depth = [0] * n
    for _, members in packages[:2 * n - 2]:
        for i in members:
            depth[i] += 1
    return sum(f * d for f, d in zip(items, depth))


@pytest.mark.parametrize("seed", range(12))
def test_generate_is_an_optimal_length_limited_prefix_code(seed):
    rng = np.random.default_rng(seed)
    for nsym, max_bits in ((286, 15), (30, 15), (19, 7)):
        kind = seed % 4
        if kind == 0:
            freq = rng.integers(0, 50, nsym)
        elif kind == 1:  # geometric: the unlimited Huffman code would be deeper than max_bits
            freq = (2.0 ** rng.uniform(0, 24, nsym)).astype(np.int64)
        elif kind == 2:  # Fibonacci-like, the classic worst case for depth
            fib = [1, 1]
            while len(fib) < nsym:
                fib.append(min(fib[-1] + fib[-2], 1 << 26))
            freq = np.array(fib[:nsym])
        else:
            freq = rng.integers(0, 4, nsym) * rng.integers(0, 2, nsym)
        freq = freq.astype(np.int32)
        code, ln = flate_generate(freq, max_bits)
        used = freq > 0
        assert (ln[~used] == 0).all()
        if used.sum() == 0:
            continue
        assert (ln[used] >= 1).all() and ln.max() <= max_bits
        if used.sum() <= 2:
            assert (ln[used] == 1).all()
            continue
        assert sum(2.0 ** -int(x) for x in ln[used]) == 1.0  # Kraft equality: a complete prefix code
        assert int((freq.astype(np.int64) * ln).sum()) == package_merge_cost(freq.tolist(), max_bits)
        # canonical (RFC 1951 3.2.2): codes of one length ascend with the symbol, written bit-reversed
        nxt, bl_count = {}, np.bincount(ln[used], minlength=max_bits + 2)
        c = 0
        for bits in range(1, max_bits + 1):
            c = (c + int(bl_count[bits - 1] if bits > 1 else 0)) << 1
            nxt[bits] = c
        for sym in np.nonzero(used)[0]:
            b = int(ln[sym])
            want = int(format(nxt[b], f"0{b}b")[::-1], 2)
            nxt[b] += 1
            assert int(code[sym]) == want, (sym, b)


def opticks():
    return open(OPTICKS, "rb").read()


@pytest.mark.parametrize("level", [5, 6, 7])
def test_writer_levels_round_trip(level):
    """flate_test.go:44-65 (TestWriterLevels) for the levels whose match finder is M4."""
    data = opticks()
    out = flate_compress(OracleM4(**flate_m4_params(level)), data)
    assert inflate(out) == data
    assert len(out) < len(data) * 0.45


@pytest.mark.parametrize("level", [5, 7])
def test_gzip_writer_levels_round_trip(level):
    """flate_test.go:67-90 (TestGZIPWriterLevels)."""
    data = opticks()
    out = flate_compress(OracleM4(**flate_m4_params(level)), data, gzip=True, mtime=1700000000)
    assert out[:4] == b"\x1f\x8b\x08\x00" and struct.unpack("<I", out[4:8])[0] == 1700000000 and out[8:10] == b"\x00\xff"
    assert gzip.decompress(out) == data
    assert struct.unpack("<II", out[-8:]) == (zlib.crc32(data), len(data))
    assert out[10:-8] == flate_compress(OracleM4(**flate_m4_params(level)), data)  # the deflate body is the flate writer's


def test_block_kinds_and_edge_inputs():
    rng = np.random.default_rng(5)
    text = opticks()[:60000]
    noise = rng.integers(0, 256, 60000, dtype=np.uint8).tobytes()
    for data in (text, noise, b"abc", b"", b"a" * 200, noise[:60000] + text):
        for bs in (0, 1 << 16, 4096):
            assert inflate(flate_compress(OracleM4(**flate_m4_params(6)), data, bs)) == data
    for bs in (0, 1 << 16, 4096):  # one block with a match of 299 bytes: the reference panics (token.go:60-62)
        with pytest.raises(ValueError):
            flate_compress(OracleM4(**flate_m4_params(6)), b"a" * 300, bs)
    assert inflate(flate_compress(OracleM4(**flate_m4_params(6)), b"a" * 300, 128)) == b"a" * 300
    # the kind chosen per block, through the Encoder interface
    for data, kind in ((text, "dynamic"), (noise, "stored"), (b"abc", "fixed")):
        mf = OracleM4(**flate_m4_params(5))
        enc = OracleFlateEncoder()
        m = mf.FindMatches(data)
        out = enc.Encode(data, m, True)
        assert enc.last_kind == kind
        assert inflate(out) == data
    # blocks are concatenated at bit granularity (the writer keeps its pending bits, huffman_bit_writer.go:506-517)
    mf, enc, out = OracleM4(**flate_m4_params(5)), OracleFlateEncoder(), b""
    parts = [text[:1000], noise[:500], text[1000:30000], b"", text[30000:]]
    for k, part in enumerate(parts):
        out += enc.Encode(part, mf.FindMatches(part), k == len(parts) - 1)
    assert inflate(out) == b"".join(parts)


def test_match_limits_are_the_reference_panics():
    src = b"x" * 400
    enc = OracleFlateEncoder()
    with pytest.raises(ValueError):
        enc.Encode(src, np.array([[1, 399, 1]], dtype=np.uint32), True)  # token.go:60-62
    big = bytes(40000) + b"y"
    with pytest.raises(ValueError):
        OracleFlateEncoder().Encode(big, np.array([[1, 100, 1], [32900, 10, 32800], [6990, 0, 0]], dtype=np.uint32), True)
    # the whole Writer: a block in the middle of the stream with a match the format cannot hold
    data = synth.text(70000, 1) + b"a" * 70000 + synth.text(70000, 2)
    with pytest.raises(ValueError):
        flate_compress(OracleM4(**flate_m4_params(5)), data, 1 << 16)
    lit, off = flate_statistics(b"abcabcabc", np.array([[3, 6, 3]], dtype=np.uint32))
    assert lit[ord("a")] == lit[ord("b")] == lit[ord("c")] == 1 and lit[257 + 3] == 1 and off[2] == 1
    assert lit.sum() == 4 and off.sum() == 1


def test_streaming_m4_matches_of_flate_levels_stay_inside_the_token_limits_on_text():
    """What makes flate.NewWriter(level 5..7) work on the reference's test file: MaxDistance 1<<15 bounds the
    distances; nothing bounds the lengths, and Opticks simply has no repeat longer than 258."""
    data = opticks()
    mf = OracleM4(**flate_m4_params(7))
    for pos in range(0, len(data), 1 << 16):
        m = mf.FindMatches(data[pos:pos + (1 << 16)])
        assert m[:, 1].max() <= 258 and m[:, 2].max() <= 32768
