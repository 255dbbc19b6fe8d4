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
