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
