"""GPU tests of K2 (brotli_amd/csrc/k2_eval.h): the chain walks of m4.go:156-181 as kernels of their own.

* the per-position results the kernels leave (m4g_debug_batch_dump_k2) equal the ones computed on the host by
  walking the chains (tests/sim/parse_sim.cpp: sim_k2_reference) -- word for word, marks included;
* the Match lists of a plan that runs K2 equal the oracle's, and equal the same plan's with K2 switched off
  (m4g_debug_batch_set_k2), on text, logs, long repeats, ragged and tiny chunks, for several parameter sets."""
import numpy as np
import pytest

from brotli_amd import m4, synth
from oracle_lib import OracleM4
from sim_lib import k2_reference

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _k2_on(monkeypatch):
    """K2 is opt-in (M4G_K2=1, read when a plan is created): on text it costs more than it saves."""
    monkeypatch.setenv("M4G_K2", "1")

D = dict(MaxDistance=16777200, ChainLength=64, HashLen=5, DistanceBitCost=66)  # BASELINE config 4
PARAMS = [
    D,
    dict(MaxDistance=1 << 20, MinLength=4, HashLen=5, TableBits=17, ChainLength=8),
    dict(MaxDistance=3000, MinLength=3, HashLen=4, TableBits=14, ChainLength=16, DistanceBitCost=20),
    dict(MaxDistance=1 << 18, MinLength=6, HashLen=8, TableBits=15, ChainLength=79, DistanceBitCost=300),
]


def same(a, b):
    a = np.asarray(a, dtype=np.uint32).reshape(-1, 3)
    b = np.asarray(b, dtype=np.uint32).reshape(-1, 3)
    return a.shape == b.shape and bool((a == b).all())


def _chunks(seed=3):
    rng = np.random.default_rng(seed)
    out = []
    for k, (kind, n) in enumerate([("text", 1 << 16), ("log", 1 << 16), ("text", 40001), ("text", 65536 - 7),
                                   ("log", 12345), ("text", 9), ("text", 8), ("text", 300)]):
        a = np.empty(n, dtype=np.uint8)
        synth.fill(a, n, 100 + k, kind, 1)
        out.append(a.tobytes())
    para = bytes(rng.integers(97, 123, 900, dtype=np.uint8))
    out.append(b"".join(para[int(rng.integers(0, 400)):] for _ in range(80))[:1 << 16])
    out.append(bytes(30000))
    out.append(rng.integers(0, 4, 50000, dtype=np.uint8).tobytes())
    out.append(b"abc" * 10000)
    out.append(b"")
    out.append(b"abcdefg")
    return out


def _run(plan, chunks):
    import torch
    host = np.frombuffer(b"".join(chunks), dtype=np.uint8)
    off = np.cumsum([0] + [len(c) for c in chunks]).astype(np.uint64)
    d_src = torch.zeros(host.size + 64, dtype=torch.uint8, device="cuda:0")
    d_src[:host.size].copy_(torch.from_numpy(host.copy()))
    cap = host.size // 3 + 2 * len(chunks) + 1
    d_out = torch.empty(3 * cap, dtype=torch.int32, device="cuda:0")
    d_off = torch.empty(len(chunks) + 1, dtype=torch.int64, device="cuda:0")
    plan.run(d_src, off, d_out, d_off)
    plan.status()
    o = d_off.cpu().numpy()
    out = d_out.cpu().numpy().view(np.uint32).reshape(-1, 3)
    return [out[o[c]:o[c + 1]].copy() for c in range(len(chunks))]


@pytest.mark.parametrize("pi", range(len(PARAMS)))
def test_k2_results_and_match_lists(pi):
    kw = PARAMS[pi]
    chunks = _chunks()
    plan = m4.BatchPlan(len(chunks), 1 << 16, 0, **kw)
    assert plan.info()["k2"] == 1, "a chain-walking plan of small chunks runs K2"
    with_k2 = _run(plan, chunks)
    # the kernels' per-position results against the chain walks on the host
    for c, data in enumerate(chunks):
        n = len(data)
        if n < 8:
            continue
        chain, res = plan.dump_k2(c, n)
        ref, mark = k2_reference(data, **kw)
        live = np.arange(n) < n - 7  # positions the loop visits (m4.go:112)
        got_mark = (chain >> 31).astype(np.uint8)
        assert (got_mark[live] == mark[live]).all(), f"chunk {c}: marks differ at {np.flatnonzero(got_mark[live] != mark[live])[:5]}"
        m = live & (mark != 0)
        bad = np.flatnonzero((res[m] != ref[m]).any(axis=1))
        assert bad.size == 0, f"chunk {c}: results differ at {np.flatnonzero(m)[bad[:5]]}: {res[m][bad[:2]]} vs {ref[m][bad[:2]]}"
    for c, data in enumerate(chunks):
        assert same(with_k2[c], OracleM4(**kw).FindMatches(data)), f"chunk {c} (with K2) differs from the oracle"
    assert plan.set_k2(False)
    without = _run(plan, chunks)
    assert plan.info()["k2"] == 0
    for c in range(len(chunks)):
        assert same(with_k2[c], without[c]), f"chunk {c}: with and without K2 differ"
    assert plan.set_k2(True)
    again = _run(plan, chunks)
    for c in range(len(chunks)):
        assert same(with_k2[c], again[c]), f"chunk {c}: second run with K2 differs"


def test_plans_without_k2(monkeypatch):
    assert m4.BatchPlan(4, 1 << 16, 0, MaxDistance=1 << 20, DistanceBitCost=66).info()["k2"] == 0, "ChainLength 0"
    assert m4.BatchPlan(2, 1 << 22, 0, **D).info()["k2"] == 0, "chunks of 4 MiB: distances do not fit the entries"
    p = m4.BatchPlan(2, (1 << 22) - 1, 0, **D)
    assert p.info()["k2"] == 1
    assert not m4.BatchPlan(2, 1 << 16, 0, MaxDistance=1 << 20).set_k2(True), "no buffers, nothing to switch"
    monkeypatch.delenv("M4G_K2")
    info = m4.BatchPlan(2, 1 << 16, 0, **D).info()
    assert info["k2"] == 0 and info["sorted_walk"] == 0, "not asked for"


def test_config4_64_chunks_of_one_mib_default_plan(monkeypatch):
    """BASELINE config 4 through the default plan (the parse chases the chains; 128-byte sub-blocks, which
    spec_alloc picks for chain-walking plans of 64 MiB and more): 64 x 1 MiB of text, eight chunks against the
    oracle, the rest through their sums."""
    monkeypatch.delenv("M4G_K2")
    n, C = 1 << 20, 64
    host = np.empty(n * C, dtype=np.uint8)
    synth.fill(host, n, 0, "text", 8)
    chunks = [host[c * n:(c + 1) * n].tobytes() for c in range(C)]
    plan = m4.BatchPlan(C, n, 0, **D)
    info = plan.info()
    assert info["k2"] == 0 and info["sorted_walk"] == 0 and info["slot_worst"] == 128 // 4 + 18
    got = _run(plan, chunks)
    for c in list(range(0, C, 9)):
        assert same(got[c], OracleM4(**D).FindMatches(chunks[c])), f"chunk {c}"
    for c in range(C):
        assert int(got[c][:, 0].sum() + got[c][:, 1].sum()) == n, f"chunk {c}: the list does not tile the chunk"


def test_config4_one_mib_chunks_with_k2():
    """BASELINE config 4 at its chunk size: 8 x 1 MiB of text, all chunks against the oracle."""
    n, C = 1 << 20, 8
    host = np.empty(n * C, dtype=np.uint8)
    synth.fill(host, n, 0, "text", 4)
    chunks = [host[c * n:(c + 1) * n].tobytes() for c in range(C)]
    plan = m4.BatchPlan(C, n, 0, **D)
    assert plan.info()["k2"] == 1
    got = _run(plan, chunks)
    for c in range(C):
        assert same(got[c], OracleM4(**D).FindMatches(chunks[c])), f"chunk {c}"


@pytest.mark.parametrize("pi", range(len(PARAMS) + 1))
def test_sorted_walk_equals_the_chain_chase_and_the_oracle(pi, monkeypatch):
    """M4G_SW=1: the parse reads a walk's candidates from the bucket-sorted position list (k2_place's SortEnt),
    four per round of its loop, instead of chasing the chain.  Same Match lists as the oracle and as the same plan
    with the list switched off."""
    monkeypatch.delenv("M4G_K2")
    monkeypatch.setenv("M4G_SW", "1")
    kw = (PARAMS + [dict(MaxDistance=1 << 16, ChainLength=126, HashLen=4, TableBits=8)])[pi]
    chunks = _chunks(seed=5 + pi)
    plan = m4.BatchPlan(len(chunks), 1 << 16, 0, **kw)
    info = plan.info()
    assert info["sorted_walk"] == 1 and info["k2"] == 0
    got = _run(plan, chunks)
    for c, data in enumerate(chunks):
        assert same(got[c], OracleM4(**kw).FindMatches(data)), f"chunk {c} differs from the oracle"
    assert plan.set_sorted_walk(False)
    chase = _run(plan, chunks)
    assert plan.info()["sorted_walk"] == 0
    for c in range(len(chunks)):
        assert same(got[c], chase[c]), f"chunk {c}: sorted walk and chain chase differ"


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_the_three_forms_of_the_walk(seed, monkeypatch):
    """Random parameter sets with ChainLength > 0 x random structured inputs (tests/test_gpu_fuzz.py's generators):
    the plan with K2, the plan with the sorted walk and the plan that chases the chain, each against the oracle."""
    from test_gpu_fuzz import random_input, random_params
    rng = np.random.default_rng(7000 + seed)
    kw = random_params(rng)
    kw["ChainLength"] = int(rng.choice([1, 3, 16, 64, 79]))
    kw["TableBits"] = int(rng.choice([0, 6, 10, 14, 16, 17]))  # (tables above 2^17 have no split K1, hence no list)
    sizes = [int(rng.choice([0, 1, 7, 8, 9, 63, 64, 3071, 3072, 3073, 4097])) for _ in range(3)] + \
            [int(rng.integers(10, 60000)) for _ in range(6)]
    chunks = [random_input(rng, s) for s in sizes]
    ref = [OracleM4(**kw).FindMatches(c) for c in chunks]
    for env in (dict(M4G_K2="1"), dict(M4G_K2="0", M4G_SW="1"), dict(M4G_K2="0", M4G_SW="0")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        plan = m4.BatchPlan(len(chunks), 1 << 16, 0, **kw)
        info = plan.info()
        assert info["k2"] == int(env.get("M4G_K2") == "1") and info["sorted_walk"] == int("1" in env.values()), (env, info)
        got = _run(plan, chunks)
        for c in range(len(chunks)):
            assert same(got[c], ref[c]), f"seed {seed} {kw} {env}: chunk {c} (len {sizes[c]}) differs from the oracle"
