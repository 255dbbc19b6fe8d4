"""CPU checks of the restated Bargain1/2/3 (oracle/bargain_ref.c) -- SURVEY §8(f).3.

The reference holds no golden Match lists and cannot be run here (no Go toolchain), so the
restatement is pinned by (a) a second restatement written independently in another form
(absolute stream positions, no history buffer, no slides; numpy float32), (b) the
properties the algorithm guarantees, and (c) the reference's own test criterion
(brotli_test.go:918-932 -> `test`, brotli_test.go:730-744): the stream the Encoder makes from
the matches decodes to the input.
"""
import math
import os
import struct

import numpy as np
import pytest

from oracle_lib import OracleBargain, OracleEncoder, brotli_decompress, lib, replay

HERE = os.path.dirname(os.path.abspath(__file__))
OPTICKS = os.path.join(HERE, "golden", "Isaac.Newton-Opticks.txt")

F32 = np.float32
MASK64 = (1 << 64) - 1


def py_hashes(variant, u, e):
    if variant == 1:
        return [((((u << 16) & MASK64) * 227718039650203) & MASK64) >> (64 - 18)]
    h5 = (((u << 24) & MASK64) * 889523592379) & MASK64
    h8 = (u * 0xcf1bbcdcb7a56463) & MASK64
    if variant == 2:
        return [h5 >> (64 - 18), h8 >> (64 - 18)]
    h12 = (u * 0xcf1bbcdcb7a56463 + e * (2654435761 << 32)) & MASK64
    return [h5 >> (64 - 17), h8 >> (64 - 18), h12 >> (64 - 19)]


def py_bargain_stream(data: bytes, block_size: int, variant: int, max_distance: int, skip: bool):
    """Second restatement.  Tables are dicts of absolute stream positions and never slide: a
    candidate further back than MaxDistance is rejected by bargain1.go:166 whether or not a slide
    (bargain1.go:66-80) has cleared it, so the only trace a slide leaves is the origin of the
    history coordinates, which Skip's `i%2` (bargain1.go:159) and the empty table entry {0, 0} see."""
    max_distance = max_distance or (1 << 16)
    tail = 12 if variant == 3 else 8
    overlap = 4 if variant == 1 else 6
    ntab = variant
    tables = [dict() for _ in range(ntab)]
    out, block_off = [], [0]
    origin = 0  # stream position of history[0]
    for hs in range(0, len(data), block_size):
        he = min(len(data), hs + block_size)
        n = he - hs
        if hs - origin > 2 * max_distance:
            origin = hs - max_distance
        hist = [0] * 256
        for b in data[hs:he]:
            hist[b] += 1
        byte_cost = []
        for c in hist:
            if c == 0:
                byte_cost.append(F32(np.inf))
            else:
                byte_cost.append(F32(max(float(np.log2(np.float64(n) / np.float64(c))), 1.0)))
        length = [0] * (n + 1)   # arrival at position hs + k, k = 1..n
        dist = [0] * (n + 1)
        cost = [F32(0)] * (n + 1)

        def add_match(start, end, match, unmatched, repeat):
            c = cost[start - hs] if start > hs else F32(0)
            c = F32(c + F32(4))
            c = F32(c + F32(int(unmatched).bit_length()))
            c = F32(c + F32(0 if repeat else int(start - match).bit_length()))
            j = end
            while j >= start + 3:
                k = j - hs
                if cost[k] > 0 and cost[k] <= c:
                    break
                length[k], dist[k], cost[k] = j - start, start - match, c
                j -= 1

        def extend(start, cand, lo):
            a, b = cand + 4, start + 4
            while b < he and data[a] == data[b]:
                a, b = a + 1, b + 1
            while start > lo and cand > origin and data[start - 1] == data[cand - 1]:
                start, cand = start - 1, cand - 1
            return start, b, cand

        next_overlap = 0
        for i in range(hs, he):
            k = i - hs
            a_len, a_dist, a_cost = (length[k], dist[k], cost[k]) if k > 0 else (0, 0, F32(0))
            unmatched = a_len if a_dist == 0 else 0
            prev_distance = dist[k - unmatched] if (unmatched != 0 and i - unmatched > hs) else 0
            lc = F32(a_cost + byte_cost[data[i]])
            if cost[k + 1] == 0 or lc < cost[k + 1]:
                length[k + 1], dist[k + 1], cost[k + 1] = unmatched + 1, 0, lc
            if i > he - tail:
                continue
            u = int.from_bytes(data[i:i + 8], "little")
            e = int.from_bytes(data[i + 8:i + 12], "little") if variant == 3 else 0
            cv32 = u & 0xFFFFFFFF
            cands = []
            for t, h in enumerate(py_hashes(variant, u, e)):
                p = tables[t].get(h)
                if p is None or p < origin:
                    cands.append((origin, 0))
                else:
                    cands.append((p, int.from_bytes(data[p:p + 4], "little")))
                tables[t][h] = i
            if prev_distance != 0 and prev_distance != dist[k + 4]:
                r = i - prev_distance
                if r >= origin and int.from_bytes(data[r:r + 4], "little") == cv32:
                    s, en, m = extend(i, r, i)
                    add_match(s, en, m, unmatched, True)
            if skip and (i - origin) % 2 != 0:
                continue
            next_unmatched = dist[k + 1] == 0
            if unmatched > 0 or i >= next_overlap or next_unmatched:
                for p, val in cands:
                    if p < i and i - p < max_distance and cv32 == val and \
                            int.from_bytes(data[p:p + 4], "little") == cv32:
                        s, en, m = extend(i, p, hs)
                        delta = i - s
                        if delta == 0:
                            add_match(s, en, m, unmatched, False)
                        else:
                            add_match(s, en, m, max(unmatched - delta, 0), False)
                            s, m = s + delta, m + delta
                            add_match(s, en, m, unmatched, False)
                        next_overlap = max(next_overlap, s + 1, en - overlap)
        ms = []
        k = n
        while k >= 1:
            if dist[k] > 0:
                ms.append([0, length[k], dist[k]])
            else:
                if not ms:
                    ms.append([0, 0, 0])
                ms[-1][0] = length[k]
            k -= length[k]
        ms.reverse()
        out.extend(ms)
        block_off.append(len(out))
    return np.array(out, dtype=np.uint32).reshape(-1, 3), np.array(block_off, dtype=np.uint64)


def text(n, seed=1):
    rng = np.random.default_rng(seed)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 9)), dtype=np.uint8)) for _ in range(200)]
    out = bytearray()
    while len(out) < n:
        out += words[int(rng.integers(0, len(words)))] + b" "
    return bytes(out[:n])


def check_tiles(matches, block_off, data, block_size, max_distance):
    """The Match list of every block tiles the block and copies equal bytes from within MaxDistance."""
    for b in range(len(block_off) - 1):
        ms = matches[int(block_off[b]):int(block_off[b + 1])]
        hs = b * block_size
        he = min(len(data), hs + block_size)
        pos = hs
        for u, l, d in ms.tolist():
            pos += u
            if l:
                assert l >= 3 and 0 < d < max_distance and pos - d >= 0
                assert data[pos - d:pos - d + l] == data[pos:pos + l] or d < l  # overlapping copies checked by replay
                pos += l
        assert pos == he
        assert replay(ms, data[hs:he], history=data[:hs]) == data[hs:he]


def test_go_log2_restatement():
    L = lib()
    for k in range(-5, 40):
        assert L.bargainref_log2(float(2.0 ** k)) == float(k)  # exact powers of two, log10.go:33-35
    rng = np.random.default_rng(3)
    for x in np.concatenate([rng.uniform(1, 70000, 2000), rng.uniform(1, 2, 500)]):
        got, want = L.bargainref_log2(float(x)), math.log2(float(x))
        # Log(frac)*(1/Ln2) + exp cancels near 1 (frac just above 1/2, exp 1): absolute, not relative, accuracy
        assert abs(got - want) <= 4 * math.ulp(max(want, 1.0))


def test_byte_cost_values():
    L = lib()
    f = lambda bits: struct.unpack("<f", struct.pack("<I", bits))[0]
    assert f(L.bargainref_byte_cost_bits(65536, 65536)) == 1.0   # log2(1) = 0, floor of 1 (bargain1.go:49)
    assert f(L.bargainref_byte_cost_bits(65536, 40000)) == 1.0
    assert f(L.bargainref_byte_cost_bits(65536, 16384)) == 2.0
    assert f(L.bargainref_byte_cost_bits(65536, 1)) == 16.0
    assert math.isinf(f(L.bargainref_byte_cost_bits(65536, 0)))
    assert f(L.bargainref_byte_cost_bits(1000, 3)) == np.float32(math.log2(1000 / 3))


def test_hand_cases():
    # no 8 bytes to hash: one literal run (bargain1.go:136-139, 195-200)
    assert OracleBargain(1, 1 << 20).FindMatches(b"abcdefg").tolist() == [[7, 0, 0]]
    assert OracleBargain(1, 1 << 20).FindMatches(b"").tolist() == []
    # "abcdefgh" x 4: position 8 finds position 0 (hash of 6 bytes), the match runs to the end of the block
    m = OracleBargain(1, 1 << 20).FindMatches(b"abcdefgh" * 4).tolist()
    assert m == [[8, 24, 8]]
    # zeros: the empty table entry {offset 0, val 0} is a candidate for position 1 (bargain1.go:166-167:
    # offset 0 < i, val 0 == the four zero bytes): distance 1 from the first searched position on
    m = OracleBargain(1, 1 << 20).FindMatches(bytes(64)).tolist()
    assert m == [[1, 63, 1]]
    # the repeat-distance probe (bargain1.go:148-157), traced by hand: P, 8 other bytes, P[:10], '#', P[11:], 8 other
    # bytes.  Position 28 finds P through its hash: match of 10 at distance 28 (cost 4 + Len(28) + Len(28) = 14 bits
    # against ~48 for the literals).  '#' is a literal; at position 39 the arrival is that literal (unmatched 1), the
    # previous distance is 28 and src[11:15] == src[39:43]: a repeat match of 9 at cost 4 + Len(1) + 0, which the hash
    # candidate of the same position (the same match at 5 bits more) cannot replace (:99-100).  No room to hash in the
    # last 7 positions: the tail is one literal run, attached to an empty Match (:195-200).
    P = b"0123456789abcdefghij"
    for variant in (1, 2, 3):
        m = OracleBargain(variant, 1 << 20).FindMatches(P + b"ABCDEFGH" + P[:10] + b"#" + P[11:] + b"KLMNOPQR").tolist()
        assert m == [[28, 10, 28], [1, 9, 28], [8, 0, 0]]
    # a second block continues with the first block's table and history
    z = OracleBargain(1, 1 << 20)
    z.FindMatches(b"0123456789abcdef" * 2)
    m = z.FindMatches(b"0123456789abcdef").tolist()
    assert m == [[0, 16, 16]]


@pytest.mark.parametrize("variant", [1, 2, 3])
@pytest.mark.parametrize("skip", [False, True])
def test_second_restatement_agrees(variant, skip):
    rng = np.random.default_rng(10 * variant + skip)
    inputs = [
        (text(6000, 1), 2048, 1 << 20),
        (text(5000, 2) + bytes(700) + text(3000, 2), 4096, 1 << 20),          # repeats + zero page
        (bytes(rng.integers(0, 4, size=5000, dtype=np.uint8)), 1000, 1 << 20),  # 4 symbols, ragged last block
        (bytes(rng.integers(0, 256, size=3000, dtype=np.uint8)), 1024, 0),     # incompressible, default MaxDistance
        (b"abcdefg" * 900, 1500, 1 << 20),                                     # period 7
        (text(9000, 5), 512, 1024),   # history slides every few blocks (len > 2 * MaxDistance)
        (text(9001, 6), 514, 1026),   # even deltas: Skip's parity survives the slides
    ]
    for data, bs, md in inputs:
        got, goff = OracleBargain(variant, md, skip).find_blocks(data, bs)
        want, woff = py_bargain_stream(data, bs, variant, md, skip)
        assert np.array_equal(goff, woff)
        assert np.array_equal(got, want)
        check_tiles(got, goff, data, bs, md or (1 << 16))


def test_odd_slides_move_skip_parity():
    """With Skip the history origin's parity is visible (bargain1.go:159): block sizes that make a slide's
    delta odd are covered by the second restatement's origin tracking too."""
    data = text(9000, 7)
    for variant in (1, 2):
        got, goff = OracleBargain(variant, 1001, True).find_blocks(data, 517)
        want, woff = py_bargain_stream(data, 517, variant, 1001, True)
        assert np.array_equal(goff, woff) and np.array_equal(got, want)


def test_reset_forgets_history():
    z = OracleBargain(1, 1 << 20)
    a = z.FindMatches(text(3000, 9))
    z.FindMatches(text(3000, 10))
    z.Reset()
    assert np.array_equal(z.FindMatches(text(3000, 9)), a)


@pytest.mark.parametrize("variant,skip", [(1, False), (1, True), (2, False), (2, True), (3, False), (3, True)])
def test_reference_criterion_opticks(variant, skip):
    """brotli_test.go:886-932: Writer{Bargain*, &Encoder{}, 1<<16} on Opticks must decode to the input."""
    data = open(OPTICKS, "rb").read()
    z = OracleBargain(variant, 1 << 20, skip)
    enc = OracleEncoder()
    out = bytearray()
    bs = 1 << 16
    for p in range(0, len(data), bs):
        blk = data[p:p + bs]
        out += enc.Encode(blk, z.FindMatches(blk), last_block=False)
    out += enc.Encode(b"", np.zeros((0, 3), np.uint32), last_block=True)
    assert brotli_decompress(bytes(out), len(data)) == data
    assert len(out) < len(data) * 0.36  # NewWriterV2 level 5-9 territory
