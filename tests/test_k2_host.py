"""CPU tests of the K2 form of the chain walks (brotli_amd/csrc/m4_k2.h): the parse that takes a position's
candidates from K2's per-position results must produce the Match list of the parse that walks the chain
(m4.go:156-181), i.e. the oracle's, for every parameter set.  The results are computed on the host here by the
same functions (k2_pair, k2_add) the kernel uses; tests/test_gpu_k2.py checks the kernel's against them."""
import numpy as np
import pytest

from brotli_amd import synth
from cases import edge_inputs, opticks
from oracle_lib import OracleM4
from sim_lib import k2_reference, sim, sim_k2, sim_sw

# (the sorted walk also serves DistanceBitCost < 0 and ChainLength up to 126, which K2 does not)
PARAMS_SW_ONLY = [
    dict(MaxDistance=1 << 16, ChainLength=126, HashLen=4, TableBits=8),
    dict(MaxDistance=1 << 20, ChainLength=5, HashLen=5, DistanceBitCost=-40),
]
PARAMS = [
    dict(MaxDistance=16777200, ChainLength=64, HashLen=5, DistanceBitCost=66),   # BASELINE config 4
    dict(MaxDistance=1 << 20, MinLength=4, HashLen=5, TableBits=17, ChainLength=8),
    dict(MaxDistance=65535, ChainLength=1, HashLen=6),
    dict(MaxDistance=1000, MinLength=3, HashLen=4, TableBits=12, ChainLength=16, DistanceBitCost=20),
    dict(MaxDistance=1 << 18, MinLength=6, HashLen=8, TableBits=15, ChainLength=32, DistanceBitCost=300),
    dict(MaxDistance=70000, MinLength=5, HashLen=5, TableBits=10, ChainLength=79),
]


def same(a, b):
    a = np.asarray(a, dtype=np.uint32).reshape(-1, 3)
    b = np.asarray(b, dtype=np.uint32).reshape(-1, 3)
    return a.shape == b.shape and bool((a == b).all())


def _inputs():
    rng = np.random.default_rng(11)
    text = np.empty(150001, dtype=np.uint8)
    synth.fill(text, text.size, 5, "text", 2)
    log = np.empty(90000, dtype=np.uint8)
    synth.fill(log, log.size, 6, "log", 2)
    para = bytes(rng.integers(97, 123, 700, dtype=np.uint8))
    return {
        "opticks": opticks()[:200000],
        "text": text.tobytes(),
        "log": log.tobytes(),
        "four_symbols": rng.integers(0, 4, 60000, dtype=np.uint8).tobytes(),
        "paragraphs": b"".join(para[int(rng.integers(0, 300)):] for _ in range(150)),
        "zeros": bytes(50000),
        "period3": b"abc" * 20000,
    }


@pytest.mark.parametrize("pi", range(len(PARAMS)))
def test_parse_with_k2_results_equals_the_chain_walking_parse(pi):
    kw = PARAMS[pi]
    for name, data in _inputs().items():
        ref = OracleM4(**kw).FindMatches(data)
        assert same(sim(data, **kw), ref), f"{name}: host parse differs from the oracle"
        assert same(sim_k2(data, **kw), ref), f"{name}: parse with K2 results differs from the oracle"
        assert same(sim_sw(data, **kw), ref), f"{name}: parse over the sorted position list differs from the oracle"


def test_edge_inputs_with_k2_results():
    kw = PARAMS[0]
    for name, data in edge_inputs().items():
        if len(data) < 8:  # (the host harness reads 8 bytes per hash; shorter buffers have no loop position)
            continue
        assert same(sim_k2(data, **kw), OracleM4(**kw).FindMatches(data)), name
        assert same(sim_sw(data, **kw), OracleM4(**kw).FindMatches(data)), name


def test_results_keep_few_candidates_on_text_and_fall_back_on_long_repeats():
    inp = _inputs()
    res, mark = k2_reference(inp["text"], **PARAMS[0])
    fb = (res[:, 0] == 0xFFFFFFFF)
    assert fb.mean() < 0.02, "text: (almost) every walk fits the kept candidates"
    assert ((res[:, 0] != 0) == (mark != 0)).all(), "marked = has a result or falls back"
    # entries are in walk order: distances ascend
    two = (~fb) & (res[:, 1] != 0)
    assert ((res[two, 0] & 0x3FFFFF) < (res[two, 1] & 0x3FFFFF)).all()
    res, _ = k2_reference(inp["zeros"], **PARAMS[0])
    assert (res[100:40000, 0] == 0xFFFFFFFF).all(), "a zero page: every candidate agrees beyond the compared bytes"


@pytest.mark.parametrize("pi", range(len(PARAMS_SW_ONLY)))
def test_sorted_walk_parameter_sets_k2_does_not_serve(pi):
    kw = PARAMS_SW_ONLY[pi]
    for name, data in _inputs().items():
        assert same(sim_sw(data, **kw), OracleM4(**kw).FindMatches(data)), name
