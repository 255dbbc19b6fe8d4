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

from cases import bargain_kat_phrase_at_odd_position, bargain_kat_twelve_byte_candidate
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


def test_hand_traced_skip_and_back_extension():
    """Bargain1, one block of 64 bytes (cases.bargain_kat_phrase_at_odd_position): 64 distinct bytes except a phrase
    at position 8 (even) and again at 33 (odd).  byteCost = max(log2(64 / n), 1): 5 for the phrase's bytes (n = 2),
    6 for the others; baseMatchCost = 4.  All sums below are exact in float32 (small integers).

    (a) phrase "UVWXYZ" (6 bytes), no Skip.  Arrival costs by literals: position 8: 8 x 6 = 48; 14: 48 + 6 x 5 = 78;
    33: 78 + 19 x 6 = 192, a literal run of 33 (arrivals[32] = {length 33, distance 0, cost 192}).  i = 33: unmatched
    = 33; i - unmatched = 0 is not > historyLen, so no previous distance and no repeat probe (bargain1.go:123-125);
    table6[hash6("UVWXYZ")] = {8, "UVWX"}: extendMatch2(src, 33, 8, 0) runs forward over "YZ" and stops at
    src[14] != src[39], backward src[32] != src[7]: m = {33, 39, 8}; addMatch: 192 + 4 + Len(33) + Len(25) =
    192 + 4 + 6 + 5 = 207 for the arrivals at 39, 38, 37, 36 (lengths 6..3; the loop stops at Start + 3,
    bargain1.go:97).  Literals from 33: 197, 202, then 202 + 5 = 207 at 36, which is not < 207, so the match arrival
    stays (:130) and the literal path ends there; position 39 costs 207 against 192 + 30 = 222.  Positions 34..38 hash
    six bytes that occur nowhere else.  From 39 on 25 literals.  Backward walk (:187-203): literal run 25 (attached
    to an empty Match), match {6, 25}, literal run 33: [[33, 6, 25], [25, 0, 0]].

    (b) the same with Skip: at i = 33 the search is skipped (:159, 33 is odd); at i = 34 the six bytes "VWXYZ" +
    src[39] differ from "VWXYZ" + src[14], the only other place with these first bytes: no candidate.  Nothing
    else repeats: [[64, 0, 0]].

    (c) phrase "TUVWXYZ" (7 bytes) with Skip: position 15: 48 + 35 = 83; 33: 83 + 18 x 6 = 191; 34: 196.  i = 33 is
    skipped.  i = 34 (even): unmatched = 34, table6[hash6("UVWXYZ")] = {9, "UVWX"}: forward over "YZ" to 40,
    BACKWARD src[33] == src[8] ('T'), src[32] != src[7]: m = {33, 40, 8}, delta = 1, so the match is added twice
    (:173-178): first from its extended start with unmatched - delta = 33 literals before it: 191 + 4 + Len(33) +
    Len(25) = 206 for the arrivals at 40..36; then from i itself, {34, 40, 9} with 34 literals: 196 + 4 + 6 + 5 =
    211, which finds 206 <= 211 at its first arrival and stops (the early exit, :99-100).  Literals: 35: 201; 36:
    206 is not < 206.  Position 40: 206 against 191 + 35 = 226: [[33, 7, 25], [24, 0, 0]] -- the same as without
    Skip, where i = 33 finds {33, 40, 8} directly at the same 206.  A restatement that only added the second form
    would return [[34, 6, 25], ...]; one without the early exit would let 211 overwrite 206 with the same effect."""
    six, seven = bargain_kat_phrase_at_odd_position(b"UVWXYZ"), bargain_kat_phrase_at_odd_position(b"TUVWXYZ")
    assert OracleBargain(1, 1 << 20, False).FindMatches(six).tolist() == [[33, 6, 25], [25, 0, 0]]
    assert OracleBargain(1, 1 << 20, True).FindMatches(six).tolist() == [[64, 0, 0]]
    assert OracleBargain(1, 1 << 20, True).FindMatches(seven).tolist() == [[33, 7, 25], [24, 0, 0]]
    assert OracleBargain(1, 1 << 20, False).FindMatches(seven).tolist() == [[33, 7, 25], [24, 0, 0]]
    # Bargain2 / Bargain3 hash 5 bytes as well: "VWXYZ" at the even position 34 leads to position 9 and the match is
    # extended back over 'U' (the same two-form addMatch): with Skip they still find the 6-byte phrase
    for variant in (2, 3):
        assert OracleBargain(variant, 1 << 20, True).FindMatches(six).tolist() == [[33, 6, 25], [25, 0, 0]]
    for data, skip in ((six, False), (six, True), (seven, True)):
        for variant in (1, 2, 3):
            want = OracleBargain(variant, 1 << 20, skip).FindMatches(data).tolist()
            assert py_bargain_stream(data, 64, variant, 1 << 20, skip)[0].tolist() == want


def test_hand_traced_twelve_byte_hash_is_the_only_way_to_the_best_match():
    """Bargain3 against Bargain2, two blocks of 64 bytes (cases.bargain_kat_twelve_byte_candidate).  Block 1 holds
    Q (12 distinct bytes) at 4, Q[:11] + x at 24, y + Q[1:] at 44; block 2 (positions 64..127 of the history) is 16
    distinct bytes, Q at 80, 36 distinct bytes.  Every byte of block 2 occurs once in it: byteCost = log2(64) = 6.
    Tables after block 1: table5 / table8 entries of Q's windows point at the LATEST copy (24.. for windows that
    start at Q[0], 45.. for those that start at Q[1] or later); table12[hash12(Q)] = 4, the only place with all 12.

    Block 2, Bargain3 (bargain3.go:153-254).  Literals: position p costs 6 (p - 64); arrivals[15] (position 80) =
    {length 16, distance 0, cost 96}.  i = 80: unmatched 16, i - unmatched = 64 is not > historyLen: no repeat probe.
      candidate5 = {24}: forward Q[4..10], src[35] = x != src[91] = Q[11]: m = {80, 91, 24};
                   96 + 4 + Len(16) + Len(56) = 96 + 4 + 5 + 6 = 111 for the arrivals at 91..83;
      candidate8 = {24}: the same match, 111 <= 111 at its first arrival: early exit;
      candidate12 = {4}: forward to src[16] != src[92]: m = {80, 92, 4}; 96 + 4 + 5 + Len(76) = 96 + 4 + 5 + 7 =
                   112 for the arrival at 92; at 91 it finds 111 <= 112 and stops.
      nextOverlapSearch = max(81, 91 - 6, 92 - 6) = 86.
    i = 81 (unmatched 17, cost 102): candidate5 = candidate8 = {45}: {81, 92, 45}: 102 + 4 + Len(17) + Len(36) =
      102 + 4 + 5 + 6 = 117, stops at 112 <= 117.  i = 82: {82, 92, 46} extends back to {81, 92, 45}: 117 again, and
      108 + 4 + 5 + 6 = 123 from 82 itself.  Literals 102, 108, then 108 + 6 = 114 is not < 111 at position 83.
    i = 83, 84, 85: the arrival is a match (unmatched = 0), i < nextOverlapSearch = 86 and the next arrival is a
      match too: the gate of :191 (bargain1.go:165) is closed, nothing is searched (the table entries are still written).
    i = 86, 87: {86, 92, 50} / {87, 92, 51} extend back to {81, 92, 45} with delta 5 / 6: added with
      max(0 - delta, 0) = 0 literals: 102 + 4 + Len(0) + Len(36) = 102 + 4 + 0 + 6 = 112: a TIE with the arrival at 92,
      `a.cost <= cost` keeps the first comer (the 12-byte match); from 86 / 87 themselves 111 + 4 + 0 + 6 = 121.
    Position 92 costs 112 through {length 12, distance 76}; then 36 literals; the repeat probes after it
      (previous distance 76) compare distinct bytes.  Walk back: [[16, 12, 76], [36, 0, 0]].

    Bargain2 (no table12; the same gate with End - 6): i = 80 gives only the 111 arrivals up to 91; i = 81 sets the
    arrival at 92 to {11, 36, 117}; i = 86 lowers it to 112 ({81, 92, 45} with 0 literals, as above: 117 > 112).
    Position 91 + a literal would be 117, not better.  Walk back from 92: length 11 from 81, and position 81 is a
    literal arrival of run 17: [[17, 11, 36], [36, 0, 0]] -- one byte less matched than Bargain3 at the same cost."""
    b1, b2 = bargain_kat_twelve_byte_candidate()
    z3, z2 = OracleBargain(3, 1 << 20), OracleBargain(2, 1 << 20)
    z3.FindMatches(b1)
    z2.FindMatches(b1)
    assert z3.FindMatches(b2).tolist() == [[16, 12, 76], [36, 0, 0]]
    assert z2.FindMatches(b2).tolist() == [[17, 11, 36], [36, 0, 0]]
    for variant, want in ((3, [[16, 12, 76], [36, 0, 0]]), (2, [[17, 11, 36], [36, 0, 0]])):
        m, off = py_bargain_stream(b1 + b2, 64, variant, 1 << 20, False)
        assert m[int(off[1]):].tolist() == want


def test_hand_traced_repeat_probe_is_suppressed_under_a_match_of_that_distance():
    """The input of test_hand_cases' repeat-probe case, one step further (bargain1.go:150).  i = 39 found the
    repeat {39, 48, 11} (9 bytes at the previous distance 28) at cost c(39) + 4 + Len(1) + 0.  i = 40 can only be
    reached by a literal from 39 (a match's first arrival is Start + 3): unmatched = 2, the match before the run is
    the first one, previous distance 28 -- and arrivals[i + 3] (position 44, under the repeat match) already has
    distance 28: the probe is skipped.  It could not have changed anything: it would re-add {40, 48, 12} at
    c(39) + byteCost + 4 + Len(2), more than the arrivals it meets.  So the Match list is the one of i = 39 alone;
    the branch is reached (profiles/oracle_coverage.txt: both outcomes of bargain_ref.c's distance test)."""
    P = b"0123456789abcdefghij"
    data = P + b"ABCDEFGH" + P[:10] + b"#" + P[11:] + b"KLMNOPQR"
    for variant in (1, 2, 3):
        assert OracleBargain(variant, 1 << 20).FindMatches(data).tolist() == [[28, 10, 28], [1, 9, 28], [8, 0, 0]]
        assert py_bargain_stream(data, 1 << 16, variant, 1 << 20, False)[0].tolist() == [[28, 10, 28], [1, 9, 28], [8, 0, 0]]


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
