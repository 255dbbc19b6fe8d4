"""GPU parity: the HIP path (through the C-ABI) vs the CPU oracle, bit-exact."""
import io
import os

import numpy as np
import pytest

from brotli_amd import m4, synth
from cases import (KATS, PARAMS_D, PARAMS_FAST_KAT, PARAMS_G, PARAMS_MISC, PARAMS_REF_TESTS, edge_inputs,
                   fast_path_kat, hand_kats, opticks)
from oracle_lib import OracleEncoder, OracleM4, brotli_decompress, histograms, replay

pytestmark = pytest.mark.gpu


def same(a, b):
    a = np.asarray(a, dtype=np.uint32).reshape(-1, 3)
    b = np.asarray(b, dtype=np.uint32).reshape(-1, 3)
    return a.shape == b.shape and bool((a == b).all())


def test_kats_streaming_api():
    inputs = edge_inputs()
    for name, exp in KATS.items():
        got = m4.M4().find_matches(inputs[name])
        assert same(got, exp), name


def test_hand_traced_queue_kats_gpu():
    """cases.hand_kats: answers traced by hand through m4.go:187-259, through the
    streaming handle and as one batch (all of them as chunks of one launch)."""
    kats = hand_kats()
    for name, (kw, data, exp, reaches) in kats.items():
        assert same(m4.M4(**kw).find_matches(data), exp), (name, reaches)
    for kw in (dict(), dict(ChainLength=1)):
        sel = [(n, v) for n, v in kats.items() if v[0] == kw]
        base = np.frombuffer(b"".join(v[1] for _, v in sel), dtype=np.uint8)
        coff = np.cumsum([0] + [len(v[1]) for _, v in sel])
        out, off, _ = m4.find_matches_batch(base, coff, **kw)
        off = off.astype(np.int64)
        for c, (n, v) in enumerate(sel):
            assert same(out[off[c]:off[c + 1]], v[2]), n


@pytest.mark.parametrize("kw", [PARAMS_G, PARAMS_D] + PARAMS_REF_TESTS + PARAMS_MISC,
                         ids=lambda k: ",".join(f"{a}={b}" for a, b in k.items()) or "defaults")
def test_edge_inputs_fresh_state(kw):
    """P0: fresh handle per input (== AutoReset), every edge input."""
    g = m4.M4(**kw)
    for name, data in edge_inputs().items():
        g.Reset()
        got = g.find_matches(data)
        ref = OracleM4(**kw).FindMatches(data)
        assert same(got, ref), f"{name}: {len(got)} vs {len(ref)} matches"
        assert replay(got, data) == data


@pytest.mark.parametrize("kw", [PARAMS_G, dict(MaxDistance=1 << 18, DistanceBitCost=66, ChainLength=16),
                                dict(MaxDistance=40000, ChainLength=4, HashLen=5, DistanceBitCost=66)],
                         ids=["G", "chain16", "slide40000"])
def test_streaming_blocks_p1(kw):
    """P1: one handle, successive blocks without Reset (matchfinder.Writer),
    including the history slide (m4.go:82-99) for the small-window set."""
    data = synth.text(700000, 5)
    bs = 1 << 16
    g, o = m4.M4(**kw), OracleM4(**kw)
    pos = 0
    while pos < len(data):
        blk = data[pos:pos + bs]
        assert same(g.find_matches(blk), o.FindMatches(blk)), f"block at {pos}"
        pos += bs
    # Close() with an empty remainder
    assert same(g.find_matches(b""), o.FindMatches(b""))
    # chain/table parity (K1) after the last block
    table, chain = g.dump_tables()
    assert (table == o.dump_table()).all()
    if kw.get("ChainLength", 0) > 0:
        assert (chain == o.dump_chain()).all()
    # Reset then reuse: determinism pattern of brotli_test.go:135-149
    g.Reset()
    o.Reset()
    assert same(g.find_matches(data[:bs]), o.FindMatches(data[:bs]))


def test_roundtrip_through_writer():
    """brotli_test.go:626-649 `test()`: Writer{M4, Encoder, 1<<16} -> decoder."""
    data = opticks()  # testdata/Isaac.Newton-Opticks.txt (BASELINE config 1)
    for kw in PARAMS_REF_TESTS[:3] + [PARAMS_G]:
        buf = io.BytesIO()
        w = m4.Writer(buf, m4.M4(**kw), OracleEncoder(), 1 << 16)
        w.Write(data)
        w.Close()
        assert brotli_decompress(buf.getvalue(), len(data)) == data
        ref_stream, _ = OracleM4(**kw).compress(data, 1 << 16)
        assert buf.getvalue() == ref_stream


def _run_batch(host, n, nchunks, kw, histos=True):
    import torch
    dev = torch.device("cuda:0")
    d_src = torch.from_numpy(host).to(dev)
    plan = m4.BatchPlan(nchunks, n, 0, **kw)
    off = np.arange(nchunks + 1, dtype=np.uint64) * n
    min_len = kw.get("MinLength", 0) or 4
    d_out = torch.empty(3 * (n // min_len + 2) * nchunks, dtype=torch.int32, device=dev)
    d_off = torch.empty(nchunks + 1, dtype=torch.int64, device=dev)
    d_h = torch.empty(nchunks * 1024, dtype=torch.int32, device=dev) if histos else None
    plan.run(d_src, off, d_out, d_off, d_h)
    plan.status()
    o = d_off.cpu().numpy()
    out = d_out.cpu().numpy().view(np.uint32).reshape(-1, 3)
    h = d_h.cpu().numpy().view(np.uint32).reshape(nchunks, 1024) if histos else None
    return out, o, h


@pytest.mark.parametrize("kind", ["text", "log"])
@pytest.mark.parametrize("kw", [PARAMS_G, PARAMS_D], ids=["G", "D"])
def test_batch_device_vs_oracle(kind, kw):
    n, nchunks = 1 << 18, 12
    host = np.empty(n * nchunks, dtype=np.uint8)
    synth.fill(host, n, 100, kind, 8)
    out, o, h = _run_batch(host, n, nchunks, kw)
    for c in range(nchunks):
        chunk = host[c * n:(c + 1) * n]
        ref = OracleM4(**kw).FindMatches(chunk)
        assert same(out[o[c]:o[c + 1]], ref), f"chunk {c}"
        lit, cmd, dist, _, _ = histograms(chunk, ref)
        assert (h[c, :256] == lit).all(), f"chunk {c} literal histogram"
        assert (h[c, 256:960] == cmd).all(), f"chunk {c} command histogram"
        assert (h[c, 960:] == dist).all(), f"chunk {c} distance histogram"


def test_batch_host_api_ragged():
    """m4g_find_matches_batch: host buffers, ragged chunk sizes incl. empty."""
    sizes = [0, 5, 8, 70001, 1 << 16, 12345, 0, 200000]
    blob = b"".join(synth.text(s, 40 + i) for i, s in enumerate(sizes))
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    out, out_off, histos = m4.find_matches_batch(blob, off, (0,), True, **PARAMS_G)
    for c, s in enumerate(sizes):
        chunk = blob[int(off[c]):int(off[c + 1])]
        ref = OracleM4(**PARAMS_G).FindMatches(chunk)
        assert same(out[int(out_off[c]):int(out_off[c + 1])], ref), f"chunk {c}"
        lit, cmd, dist, _, _ = histograms(chunk, ref)
        assert (histos[c, :256] == lit).all() and (histos[c, 256:960] == cmd).all() \
            and (histos[c, 960:] == dist).all()


def test_batch_host_api_pipelined_slices():
    """m4g_find_matches_batch with one device and >= 256 chunks cuts the batch
    into slices whose copies overlap the kernels: same result as the unsliced
    call and as the oracle, ragged sizes incl. empty chunks, pinned and pageable
    output, and the -(needed) answer when the output is too small."""
    rng = np.random.default_rng(7)
    sizes = [int(x) for x in rng.choice([0, 7, 100, 3000, 9000, 20000], 300)]
    sizes[5] = 70001
    blob = b"".join(synth.text(s, 900 + i) for i, s in enumerate(sizes))
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    os.environ["M4G_PIPE_SLICE"] = "100"  # three slices, the last one short (the default slices hold >= 256 chunks)
    try:
        _run_pipelined_checks(blob, off, sizes)
    finally:
        del os.environ["M4G_PIPE_SLICE"]


def _run_pipelined_checks(blob, off, sizes):
    out, out_off, histos = m4.find_matches_batch(blob, off, (0,), True, **PARAMS_G)
    os.environ["M4G_NO_PIPELINE"] = "1"
    try:
        out1, out_off1, histos1 = m4.find_matches_batch(blob, off, (0,), True, **PARAMS_G)
    finally:
        del os.environ["M4G_NO_PIPELINE"]
    assert (out_off == out_off1).all() and same(out, out1) and (histos == histos1).all()
    for c in list(range(0, 300, 13)) + [5, 127, 128, 255, 256, 299]:
        chunk = blob[int(off[c]):int(off[c + 1])]
        ref = OracleM4(**PARAMS_G).FindMatches(chunk)
        assert same(out[int(out_off[c]):int(out_off[c + 1])], ref), f"chunk {c}"
        lit, cmd, dist, _, _ = histograms(chunk, ref)
        assert (histos[c, :256] == lit).all() and (histos[c, 256:960] == cmd).all() \
            and (histos[c, 960:] == dist).all()
    pin = m4.PinnedBuffer(12 * (out.shape[0] + 10))
    out2, out_off2, _ = m4.find_matches_batch(blob, off, (0,), False, out=pin.array.view(np.uint32), **PARAMS_G)
    assert (out_off2 == out_off).all() and same(out2, out)
    with pytest.raises(m4.M4GError, match=f"need {out.shape[0]}"):
        m4.find_matches_batch(blob, off, (0,), False, out=np.empty(3 * (out.shape[0] - 1), dtype=np.uint32), **PARAMS_G)
    pin.close()


@pytest.mark.parametrize("kw", [PARAMS_G, PARAMS_D], ids=["G", "D"])
def test_batch_histograms_edge_inputs(kw):
    """Every edge input as one chunk of a ragged batch: match lists and the
    three histograms of encoder.go:44-105 (rings shorter than four distances,
    repeated distances, literal runs longer than a wave, empty chunks)."""
    inputs = edge_inputs()
    names = sorted(inputs)
    sizes = [len(inputs[k]) for k in names]
    blob = b"".join(inputs[k] for k in names)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    out, out_off, histos = m4.find_matches_batch(blob, off, (0,), True, **kw)
    for c, name in enumerate(names):
        chunk = inputs[name]
        ref = OracleM4(**kw).FindMatches(chunk)
        assert same(out[int(out_off[c]):int(out_off[c + 1])], ref), name
        lit, cmd, dist, _, _ = histograms(chunk, ref)
        assert (histos[c, :256] == lit).all(), f"{name}: literal histogram"
        assert (histos[c, 256:960] == cmd).all(), f"{name}: command histogram"
        assert (histos[c, 960:] == dist).all(), f"{name}: distance histogram"


def test_batch_host_api_sharded_over_device_list():
    """m4g_find_matches_batch with several entries in `devices` (chunk c goes to
    devices[c % n]): the one GPU of the test box listed three times
    exercises the sharding, the per-device offsets and the ordered concatenation
    exactly as three GPUs would."""
    sizes = [70001, 0, 1 << 16, 12345, 200000, 8, 99999, 31, 150000, 5]
    blob = b"".join(synth.text(s, 300 + i) if i % 2 else synth.log(s, 300 + i) for i, s in enumerate(sizes))
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    one = m4.find_matches_batch(blob, off, (0,), True, **PARAMS_G)
    three = m4.find_matches_batch(blob, off, (0, 0, 0), True, **PARAMS_G)
    assert (one[1] == three[1]).all() and same(one[0], three[0]) and (one[2] == three[2]).all()
    for c, s in enumerate(sizes):
        chunk = blob[int(off[c]):int(off[c + 1])]
        ref = OracleM4(**PARAMS_G).FindMatches(chunk)
        assert same(three[0][int(three[1][c]):int(three[1][c + 1])], ref), f"chunk {c}"


@pytest.mark.parametrize("devices", [(0,), (0, 0, 0)], ids=["one_device", "three_lanes"])
def test_persistent_host_batch(devices):
    """m4g_hostbatch_*: same results as m4g_find_matches_batch (and the oracle), over
    several runs of changing shape on one handle; the (0, 0, 0) device list runs the
    several-devices code (per-chunk copies, ordered concatenation, one thread per lane)."""
    hb = m4.HostBatch(700, 1 << 16, devices, **PARAMS_G)
    rng = np.random.default_rng(11)
    for rounds, nchunks in enumerate((600, 7, 0, 1, 700)):
        lens = rng.integers(0, (1 << 16) + 1, nchunks)
        if nchunks > 3:
            lens[1] = 0
            lens[2] = 1 << 16
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        base = np.frombuffer(synth.text(int(off[-1]) + 1, 100 + rounds), dtype=np.uint8)[:int(off[-1])]
        got, goff, gh = hb.run(base, off, want_histos=True)
        ref, roff, rh = m4.find_matches_batch(base, off, (0,), True, **PARAMS_G)
        assert (goff == roff).all() and same(got, ref)
        if nchunks:
            assert (gh == rh).all()
        for c in list(range(min(nchunks, 4))) + ([nchunks - 1] if nchunks else []):
            chunk = base[int(off[c]):int(off[c + 1])]
            assert same(got[int(goff[c]):int(goff[c + 1])], OracleM4(**PARAMS_G).FindMatches(chunk)), c
    hb.close()


def test_batch_inputs_that_defeat_the_speculation():
    """Inputs on which the sub-block speculation cannot synchronise through a
    common flush: incompressible bytes (no matches at all), zero pages and short
    periods (one match spanning many sub-blocks), and mixtures with sharp
    transitions.  Results must still be bit-exact; tools/kind_probe.py tracks
    their speed."""
    rng = np.random.default_rng(5)
    n = 1 << 18
    text = np.frombuffer(synth.text(n, 9), dtype=np.uint8)
    rnd = rng.integers(0, 256, n, dtype=np.uint8)
    chunks = [
        rnd,
        np.zeros(n, dtype=np.uint8),
        np.resize(np.frombuffer(b"abcdefg", dtype=np.uint8), n),
        np.concatenate([text[:n // 2], rnd[:n // 2]]),
        np.concatenate([rnd[:n // 3], text[:n // 3], np.zeros(n - 2 * (n // 3), dtype=np.uint8)]),
        np.concatenate([text[:5000], np.zeros(100000, dtype=np.uint8), text[:n - 105000]]),
        np.concatenate([rnd[:70000], rnd[:70000], rnd[1000:n - 139000]]),  # far repeats inside noise
        np.concatenate([np.resize(np.frombuffer(b"xy", dtype=np.uint8), 50001), rnd[:n - 50001]]),
    ]
    # the three "paragraphs repeated" rows of tools/kind_probe.py: long matches everywhere, long back-extensions
    big = np.frombuffer(synth.text(1 << 20, 10), dtype=np.uint8)
    for plen in (2000, 8192, 40000):
        pool = [big[i * plen:(i + 1) * plen] for i in range(16)]
        chunks.append(np.concatenate([pool[int(j)] for j in rng.integers(0, 16, n // plen + 1)])[:n])
    host = np.concatenate(chunks)
    for kw in (PARAMS_G, dict(MaxDistance=1 << 16, ChainLength=4, HashLen=5, DistanceBitCost=66)):
        out, o, h = _run_batch(host, n, len(chunks), kw)
        for c, chunk in enumerate(chunks):
            ref = OracleM4(**kw).FindMatches(chunk)
            assert same(out[o[c]:o[c + 1]], ref), f"chunk {c} params {kw}"
            lit, cmd, dist, _, _ = histograms(chunk, ref)
            assert (h[c, :256] == lit).all() and (h[c, 256:960] == cmd).all() and (h[c, 960:] == dist).all()


def test_fast_path_kat_gpu():
    """cases.fast_path_kat (ADVICE r1, k3c fast path bound): through the streaming
    handle, which cuts a small input into 64-byte sub-blocks, and as chunks of a
    batch (1 KiB sub-blocks)."""
    for seed in (1, 2, 3):
        data = fast_path_kat(64, seed)
        assert same(m4.M4(**PARAMS_FAST_KAT).find_matches(data), OracleM4(**PARAMS_FAST_KAT).FindMatches(data)), seed
    chunks = [fast_path_kat(1024, seed) for seed in (1, 2, 3, 5)]
    host = np.frombuffer(b"".join(chunks), dtype=np.uint8).copy()
    out, o, _ = _run_batch(host, len(chunks[0]), len(chunks), PARAMS_FAST_KAT)
    for c, chunk in enumerate(chunks):
        assert same(out[o[c]:o[c + 1]], OracleM4(**PARAMS_FAST_KAT).FindMatches(chunk)), f"chunk {c}"


def test_full_size_chunks_properties():
    """BASELINE config 2 shape (1 MiB chunks, params G): size-independent
    properties on every chunk + bit-exactness on a sample."""
    n, nchunks = 1 << 20, 64
    host = np.empty(n * nchunks, dtype=np.uint8)
    synth.fill(host, n, 0, "text", 8)
    out, o, h = _run_batch(host, n, nchunks, PARAMS_G)
    m = out[:o[-1]].astype(np.int64)
    # every chunk's matches cover exactly n bytes
    cover = np.add.reduceat(m[:, 0] + m[:, 1], o[:-1].astype(np.int64))
    assert (cover == n).all()
    # literal histogram total == sum of Unmatched; command total == match count
    lit_tot = np.add.reduceat(m[:, 0], o[:-1].astype(np.int64))
    assert (h[:, :256].sum(axis=1) == lit_tot).all()
    assert (h[:, 256:960].sum(axis=1) == np.diff(o)).all()
    for c in (0, 17, 63):
        chunk = host[c * n:(c + 1) * n]
        got = out[o[c]:o[c + 1]]
        assert replay(got, chunk.tobytes()) == chunk.tobytes()
        assert same(got, OracleM4(**PARAMS_G).FindMatches(chunk)), f"chunk {c}"


def test_golden_vectors_gpu():
    """The committed fixtures (tests/golden) through the HIP path."""
    import os
    from golden.make_golden import PARAM_SETS, golden_inputs
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "m4_matches.npz"))
    for name, data in golden_inputs().items():
        for pname, kw in PARAM_SETS.items():
            assert same(m4.M4(**kw).find_matches(data), g[f"m/{pname}/{name}"]), (pname, name)
    data = synth.text(5 * 16384, 9)
    h = m4.M4(MaxDistance=12000, ChainLength=2, DistanceBitCost=66)
    for b in range(5):
        assert same(h.find_matches(data[b * 16384:(b + 1) * 16384]), g[f"p1/{b}"]), b


def test_seeded_repeat_distance_gpu():
    """FindMatches(dst, src) with a non-empty dst (m4.go:129-130)."""
    b1 = b"abcdefghijklmnop" * 4 + b"XY"
    b2 = b"0123" + b"abcdefghijklmnop" * 2 + b"tail-of-the-block"
    g, o = m4.M4(), OracleM4()
    dst = g.FindMatches([], b1)
    m1 = o.FindMatches(b1)
    assert same(dst, m1)
    seed = dst[-1][2] if dst[-1][1] else [d for d in dst if d[1]][-1][2]
    got = g.find_matches(b2, seed)
    assert same(got, o.FindMatches(b2, seed_prev_distance=seed))
    with pytest.raises(m4.M4GError):
        m4.M4().find_matches(b1, 1000)


def test_cpp_host_mirror():
    """include/m4g_matchfinder.hpp: matchfinder::M4 through the C-ABI from C++."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sim", "host_mirror_check")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_k1_uses_the_exchange_kernel_on_this_device():
    """libm4g checks at create time that the lanes of one LDS exchange
    instruction are performed in lane order (k1_xchg.h) and would fall back to
    the sorting kernel otherwise; on gfx950 the check passes."""
    import os
    L = m4.lib()
    v = L.m4g_debug_k1_variant(0)
    assert v == (0 if os.environ.get("M4G_K1_SORT") else 1), v


def test_histograms_of_long_literal_runs():
    """Literal runs around and above the size from which they are counted by the
    side kernel (k5b_long_literals, 8192 bytes) at odd offsets and lengths, several
    per chunk, next to chunks without any: histograms must equal encoder.go:50-55."""
    rng = np.random.default_rng(99)
    rep = synth.text(300, 5)
    chunks = []
    for lens in ([8191, 8192, 8193], [16384 + 5, 3, 40000], [70001], [100], [8192 * 3 + 1, 8200]):
        b = bytearray(b"x")
        for n in lens:
            b += bytes(rng.integers(0, 256, n, dtype=np.uint8)) + rep + rep
        chunks.append(bytes(b))
    chunks.append(synth.text(50000, 6))
    blob = b"".join(chunks)
    off = np.concatenate([[0], np.cumsum([len(c) for c in chunks])]).astype(np.uint64)
    out, out_off, histos = m4.find_matches_batch(blob, off, (0,), True, **PARAMS_G)
    n_long = 0
    for c, chunk in enumerate(chunks):
        ref = OracleM4(**PARAMS_G).FindMatches(chunk)
        assert same(out[int(out_off[c]):int(out_off[c + 1])], ref), f"chunk {c}"
        n_long += int((np.asarray(ref).reshape(-1, 3)[:, 0] >= 8192).sum())
        lit, cmd, dist, _, _ = histograms(chunk, ref)
        assert (histos[c, :256] == lit).all(), f"chunk {c}: literal histogram"
        assert (histos[c, 256:960] == cmd).all() and (histos[c, 960:] == dist).all(), f"chunk {c}"
    assert n_long >= 8  # the case really exercises the side kernel
