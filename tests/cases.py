"""Shared inputs and parameter sets for the parity tests."""
import numpy as np

from brotli_amd import synth

# BASELINE.json parameter sets: G = "greedy", D = "deep".
PARAMS_G = dict(MaxDistance=1 << 20, DistanceBitCost=66)
PARAMS_D = dict(MaxDistance=16777200, ChainLength=64, HashLen=5, DistanceBitCost=66)
# reference test parameter sets: brotli_test.go:730-744, flate/writer.go:33-46
PARAMS_REF_TESTS = [
    dict(MaxDistance=1 << 18, DistanceBitCost=66),
    dict(MaxDistance=1 << 18, DistanceBitCost=66, ChainLength=256),
    dict(MaxDistance=1 << 18, ChainLength=1, DistanceBitCost=66),
    dict(MaxDistance=1 << 15, ChainLength=8, HashLen=5, DistanceBitCost=66),
]
PARAMS_MISC = [
    dict(),  # all defaults
    dict(ChainLength=4, MinLength=5),
    dict(HashLen=8, TableBits=12),
    dict(HashLen=7, ChainLength=3, DistanceBitCost=30),
    dict(HashLen=4, TableBits=10, ChainLength=16, MaxDistance=1000),
    dict(TableBits=18, HashLen=5, MaxDistance=4096, ChainLength=2),
]


def opticks() -> bytes:
    """testdata/Isaac.Newton-Opticks.txt, the file every M4 test and benchmark of the
    reference runs on (brotli_test.go:730-780); committed as a data fixture."""
    import os
    return open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                             "Isaac.Newton-Opticks.txt"), "rb").read()


def edge_inputs():
    rng = np.random.default_rng(7)
    return {
        "empty": b"",
        "short3": b"abc",
        "len7": b"abcdefg",
        "len8": b"abcdefgh",
        "len9": b"abcdefghi",
        "kat_2x16": b"0123456789abcdef" * 2,
        "kat_2x16_tail": b"0123456789abcdef" * 2 + b"XYZ",
        "kat_run40": b"a" * 40,
        "kat_repeat": b"ABCDEFGHIJKLMNOP" + b"qr" + b"BCDEFGHIJ" + b"Z" + b"LMNOP" + b"x" * 8,
        "run_5000": b"a" * 5000,
        "period2": b"ab" * 3000 + b"abc" * 2000,
        "period_mix": (b"0123456789" * 50 + b"xyz") * 40,
        "random_20k": bytes(rng.integers(0, 256, 20000, dtype=np.uint8)),
        "lowent_50k": bytes(rng.integers(0, 3, 50000, dtype=np.uint8)),
        "zeros_70k": bytes(70000),
        "text_1000": synth.text(1000, 11),
        "text_70001": synth.text(70001, 12),
        "log_65536": synth.log(65536, 3),
    }


# SURVEY.md section 8(c) hand-derived known answers (default parameters)
KATS = {
    "kat_2x16": [(16, 16, 16)],
    "kat_2x16_tail": [(16, 16, 16), (3, 0, 0)],
    "kat_run40": [(1, 39, 1)],
    "kat_repeat": [(18, 9, 17), (1, 5, 17), (8, 0, 0)],
    "empty": [],
    "short3": [(3, 0, 0)],
}


# Parameters of fast_path_kat below.
PARAMS_FAST_KAT = dict(HashLen=4)


def fast_path_kat(P: int = 64, seed: int = 1) -> bytes:
    """Input for the fast path of the validation pass (kernels.h k3c, m4_spec.h
    fast_probe_limit) with sub-blocks of P bytes, parameters PARAMS_FAST_KAT.

    Block 7 is entered by the TRUE parse with nothing pending and repeat distance
    3P (block 5 holds a match at that distance, block 6 is unique bytes, so block
    6 cannot synchronise).  In block 7 the parse that assumes "no repeat distance"
    finds match A (u0..u9 v0..v5, source in block 1) at the position of u4 --
    u0..u3 only have 4-byte decoy candidates, which m4.go:161 rejects -- extends it
    back to u0, then finds the longer B (u4.. v19, source in block 0) at
    A.End+2-HashLen and emits A trimmed to [u0,u4) (m4.go:233-234): the first
    record ENDS at the position where A was FOUND.  The true parse probes there
    (m4.go:128-138, i = position of u4) with distance 3P and hits the planted
    "u5 u6 u7 u8" of block 4, so its result differs: a validation that checks the
    probe only below the end of the first record accepts the wrong list."""
    rng = np.random.default_rng(seed)
    buf = bytearray(rng.integers(0, 0x80, 8 * P, dtype=np.uint8).tobytes())
    u = bytes(0xA0 + t for t in range(10))
    v = bytes(0xC0 + t for t in range(20))

    def put(pos, s):
        buf[pos:pos + len(s)] = s

    put(0 * P + 8, b"\xE0" + u[4:] + v + b"\xE1")                    # SB
    put(1 * P + 4, b"\xE2" + u + v[:6] + b"\xE3")                    # SA
    o = 1 * P + 24
    for t in range(4):                                               # decoys for u0..u3
        put(o + 6 * t, bytes([0xE4 + t]) + u[t:t + 4] + bytes([0xE8 + t]))
    c = 8 + 5                                                        # offset of u5 inside block 7
    put(2 * P + 20, bytes(range(0xF0, 0xF8)))                        # source of the distance-3P match
    put(4 * P + c - 1, b"\xEC" + u[5:9] + b"\xED")                   # planted repeat-probe target
    put(5 * P + 20, bytes(range(0xF0, 0xF8)))                        # the match at distance 3P
    put(7 * P + 8, u + v)                                            # block 7
    return bytes(buf)


# ---------------------------------------------------------------------------
# Hand-traced known answers for the pending-match queue of m4.go:187-259
# (VERDICT r1, item 1d).  The expected lists below were derived by walking
# m4.go line by line over the layouts documented in each builder -- they are
# NOT outputs of any restatement.  Filler bytes (0x80..) are each used once, the
# letters of R are pairwise distinct, so two 4-byte windows are equal only when
# they are copies of the same piece of R: every hash candidate and every repeat
# probe in the traces can be decided by reading the layout.  The tests also
# check numerically that no two distinct 6-byte windows of an input share a
# 17-bit hash bucket (a collision could replace a table entry a trace relies on).
_R30 = b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcd"


def _layout(parts):
    """parts: bytes objects, or ints = that many fresh filler bytes."""
    out, nxt = bytearray(), 0x80
    for p in parts:
        if isinstance(p, int):
            for _ in range(p):
                out.append(nxt)
                nxt += 1
        else:
            out += p
    assert nxt <= 0x100
    return bytes(out)


def hand_kats():
    """name -> (params, input, expected [(Unmatched, Length, Distance)], reaches)."""
    R = _R30
    k = {}
    # -- case m4.go:213-219 (newest overlaps oldest: drop the middle one), penalty 275 and 0 --------
    # 0:u | 1-6 A=R[0:6] | 7:u | 8-15 B=R[2:10] | 16:u | 17-25 C=R[5:14] | 26,27:u | 28-41 R[0:14] | 8 u
    #  i=28 cand 1  -> A {28,34,1}   score 1536, queue empty: pushed
    #  i=30 (=A.End-4) cand 8 -> B {30,38,8} 2048 > 1536+275: pushed             (penalty 275, m4.go:189)
    #  i=34 (=B.End-4) cand 18, back-extended over 'F' -> C {33,42,17} 2304; C.Start 33 <= A.End 34 so
    #       penalty 0 (m4.go:190-194): 2304 > 2048 pushed; three pending, C.Start 33 < A.End 34: B dropped
    #  i=42 flush: A clipped to C.Start: {28,33} emitted (28,5,27), then C (0,9,16); tail 8
    k["q213_drop_middle"] = (dict(), _layout([1, R[0:6], 1, R[2:10], 1, R[5:14], 2, R[0:14], 8]),
                             [(28, 5, 27), (0, 9, 16), (8, 0, 0)], "m4.go:213-219, 190-194, 117-119")
    # -- case m4.go:221-229 (no room between oldest and newest: emit oldest, drop middle) -----------
    # 0:u | 1-6 A=R[0:6] | 7:u | 8-16 B=R[2:11] | 17:u | 18-28 C=R[7:18] | 29,30:u | 31-48 R[0:18] | 8 u
    #  i=31 A {31,37,1} 1536; i=33 B {33,42,8} 2304 > 1811; i=38 C {38,49,18} 2816 > 2304+275 (C.Start 38 >
    #  A.End 37: penalty stays 275); 38 >= A.End, 38 < A.End+4: A emitted whole (31,6,30), B dropped;
    #  i=49 flush C: (1,11,20); tail 8
    k["q221_emit_oldest"] = (dict(), _layout([1, R[0:6], 1, R[2:11], 1, R[7:18], 2, R[0:18], 8]),
                             [(31, 6, 30), (1, 11, 20), (8, 0, 0)], "m4.go:221-229")
    # -- default case m4.go:231-258 with the clip, ChainLength 0 ------------------------------------
    # 0:u | 1-8 A=R[0:8] | 9:u | 10-23 B=R[4:18] | 24:u | 25-40 C=R[14:30] | 41,42:u | 43-72 R | 8 u
    #  i=43 A {43,51,1} 2048; i=47 B {47,61,10} 3584; i=57 C {57,73,25} 4096 > 3584+275; C.Start 57 >=
    #  A.End+4: default; A.End clipped to B.Start 47: length 4 >= MinLength, score 1024 > 0: (43,4,42);
    #  i=73 flush: B clipped to C.Start: (0,10,37), C (0,16,32); tail 8
    lay = _layout([1, R[0:8], 1, R[4:18], 1, R[14:30], 2, R, 8])
    k["q231_clip_emit"] = (dict(), lay, [(43, 4, 42), (0, 10, 37), (0, 16, 32), (8, 0, 0)], "m4.go:231-234, 255-257")
    #  same input, DistanceBitCost 200: every distance here has 6 bits, so scores drop by 1200: the
    #  pushes still pass (848; 2384 > 848+275; 2896 > 2384+275) but the clipped A scores 1024-1200 <= 0
    #  and is NOT emitted (m4.go:255): its bytes become literals of B
    k["q231_clip_score_rejects"] = (dict(DistanceBitCost=200), lay, [(47, 10, 37), (0, 16, 32), (8, 0, 0)],
                                    "m4.go:255 score <= 0")
    # -- re-search of a trimmed match, m4.go:235-253 (ChainLength 1) --------------------------------
    # 0:u | 1-8 A1=R[0:8] | 9:u | 10-15 A2=R[0:6] | 16:u | 17-30 B=R[4:18] | 31:u | 32-47 C=R[14:30] |
    # 48,49:u | 50-79 R | 8 u
    #  i=10 cand 1: {10,16,1} pushed, flushed at i=16: (10,6,9); probes with distance 9 never hit
    #  i=50 cand 10 (A2): {50,56,10} 1536; chain hop chain[10]=9 -> 1 (A1): {50,58,1} 2048 > 1536: A
    #  i=54 B {54,68,17}; i=64 C {64,80,32}; default case: A clipped to {50,54}, still 4 bytes, so the
    #  chain is walked from A.Start: chain[50]=40 -> pos 10 > A.Match 1 and src[10:14]==src[50:54]:
    #  A.Match = 10: (34,4,40) instead of distance 49; flush: (0,10,37), (0,16,32); tail 8
    k["q235_research_closer"] = (dict(ChainLength=1),
                                 _layout([1, R[0:8], 1, R[0:6], 1, R[4:18], 1, R[14:30], 2, R, 8]),
                                 [(10, 6, 9), (34, 4, 40), (0, 10, 37), (0, 16, 32), (8, 0, 0)], "m4.go:235-253, 166-181")
    # -- first-candidate score gate m4.go:161 (DistanceBitCost 300) ---------------------------------
    # 0:u | 1-6 R[0:6] | 34 u | 41-46 R[0:6] | 4 u | 51-56 R[0:6] | 8 u
    #  i=41 cand 1: 6 bytes at distance 40 (6 bits): 1536-1800 <= 0 rejected, nothing pending
    #  i=51 cand 41: distance 10 (4 bits): 1536-1200 = 336 > 0: {51,57,41}; flush at 57: (51,6,10); tail 8
    k["q161_score_gate"] = (dict(DistanceBitCost=300), _layout([1, R[0:6], 34, R[0:6], 4, R[0:6], 8]),
                            [(51, 6, 10), (8, 0, 0)], "m4.go:161 score <= 0, 183-185")
    # -- m4.go:197-199: a candidate that does not beat the pending match + penalty ------------------
    # 0:u | 1-10 A=R[0:10] | 11:u | 12-21 B=R[6:16] | 22,23:u | 24-39 R[0:16] | 8 u
    #  i=24 A {24,34,1} 2560; i=30 B {30,40,12} 2560 <= 2560+275: dropped; flush at 34: (24,10,23);
    #  the same i=34 is then searched (nothing pending): cand 16 (R[10:16] inside B's source), no backward
    #  step because Start == NextEmit (m4.go:330 start > min): {34,40,16} pushed; flush at 40: (0,6,18)
    k["q197_not_better"] = (dict(), _layout([1, R[0:10], 1, R[6:16], 2, R[0:16], 8]),
                            [(24, 10, 23), (0, 6, 18), (8, 0, 0)], "m4.go:197-199, 330")
    # -- re-search that runs off the end of the chain, m4.go:241-243 (ChainLength 1) -----------------
    # 0:'x' | 1-8 R[0:8] | 9:u | 10-23 B=R[4:18] | 24:u | 25-40 C=R[14:30] | 41:u | 42:'x' | 43-72 R | 8 u
    #  i=42 window "xABCDE": its only predecessor is position 0, which the table cannot hold (0 = empty,
    #       m4.go:152): no candidate, chain[42] stays 0
    #  i=43 cand 1, extended back over 'x' to Start 42 / Match 0 (m4.go:330 stops at candidate 0):
    #       A {42,51,0} 2304; chain[1] = 0 ends the walk; i=47 B {47,61,10}; i=57 C {57,73,25}; default
    #       case: A clipped to {42,47}: 5 bytes, re-search from pos 42: chain[42] == 0: break, Match
    #       stays 0: (42,5,42); flush (0,10,37), (0,16,32); tail 8
    k["q241_chain_ends"] = (dict(ChainLength=1),
                            _layout([b"x", R[0:8], 1, R[4:18], 1, R[14:30], 1, b"x", R, 8]),
                            [(42, 5, 42), (0, 10, 37), (0, 16, 32), (8, 0, 0)], "m4.go:241-243, 330 candidate > 0")
    # -- end of input with two matches pending, m4.go:262-273 ---------------------------------------
    # 0:u | 1-8 R[0:8] | 9:u | 10-23 R[4:18] | 24,25:u | 26-43 R[0:18]   (no tail: len 44, loop i < 37)
    #  i=26 A {26,34,1} 2048; i=30 B {30,44,10}: the extension stops at len (m4.go:321); 3584 > 2323;
    #  the next searched position 40 is beyond the loop; after the loop A is clipped to 4 bytes and
    #  emitted (26,4,25), then B (0,14,20); NextEmit == len: no literal record (m4.go:276)
    lay = _layout([1, R[0:8], 1, R[4:18], 2, R[0:18]])
    k["q264_tail_two_pending"] = (dict(), lay, [(26, 4, 25), (0, 14, 20)], "m4.go:263-273, 276")
    #  same input, DistanceBitCost 300 (5-bit distances: -1500): 548; 2084 > 548+275; the clipped A scores
    #  1024-1500 <= 0 and is dropped (m4.go:267): its bytes join B's literals
    k["q267_tail_score_rejects"] = (dict(DistanceBitCost=300), lay, [(30, 14, 20)], "m4.go:267 score <= 0")
    # 0:u | 1-6 R[0:6] | 7:u | 8-16 R[2:11] | 17,18:u | 19-29 R[0:11]   (len 30, loop i < 23)
    #  i=19 A {19,25,1} (bytewise tail of extendMatch, m4.go:321); i=21 B {21,30,8} 2304 > 1811; after
    #  the loop A is clipped to 2 bytes < MinLength: dropped; B (21,9,13)
    k["q267_tail_clip_too_short"] = (dict(), _layout([1, R[0:6], 1, R[2:11], 2, R[0:11]]), [(21, 9, 13)],
                                     "m4.go:267 length < MinLength")
    return k


# ---- Bargain1 / 2 / 3: hand-traced known answers (matchfinder/bargain1.go, bargain3.go) --------------------
# Every expected Match list below was derived ON PAPER from the reference's source (the traces are in the
# docstrings of tests/test_bargain_oracle.py::test_hand_traced_*), not taken from a run of the restatement.
def _distinct(n, start, skip=()):
    """n distinct byte values from `start` upwards, none of them in `skip`."""
    out = bytearray()
    v = start
    while len(out) < n:
        if v not in skip:
            out.append(v)
        v += 1
    assert v <= 256
    return bytes(out)


def bargain_kat_phrase_at_odd_position(phrase: bytes) -> bytes:
    """64 bytes, all distinct, except `phrase` at position 8 (even) and again at position 33 (odd)."""
    body = bytearray(_distinct(64, 100, skip=set(phrase)))
    body[8:8 + len(phrase)] = phrase
    body[33:33 + len(phrase)] = phrase
    assert len(set(body)) == 64 - len(phrase)
    return bytes(body)


def bargain_kat_twelve_byte_candidate():
    """Two blocks of 64 bytes.  Block 1: Q (12 distinct bytes) at 4, then Q[:11] + x at 24, then y + Q[1:] at 44,
    everything else distinct: every 5- and 8-byte window of Q occurs again AFTER position 4, so table5 / table8
    point at the later copies, and only the 12-byte hash of the whole of Q still leads to position 4.
    Block 2: 16 distinct bytes, Q, 36 distinct bytes."""
    Q = bytes(range(65, 77))  # 'A'..'L'
    fill = _distinct(30 + 52, 100)
    f1, f2 = fill[:30], fill[30:]
    b1 = f1[0:4] + Q + f1[4:12] + Q[:11] + f1[12:13] + f1[13:21] + f1[21:22] + Q[1:] + f1[22:30]
    b2 = f2[:16] + Q + f2[16:]
    assert len(b1) == 64 and len(b2) == 64 and len(set(b1 + b2)) == 12 + 82
    return b1, b2
