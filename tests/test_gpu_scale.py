"""GPU parity at the sizes BASELINE.json states ("close the parity
gaps"): config 4 exactly as written, and the two K1 kernels against each other
on a large mixed batch."""
import ctypes

import numpy as np
import pytest

from brotli_amd import m4, synth
from cases import PARAMS_D
from oracle_lib import OracleM4, histograms

pytestmark = pytest.mark.gpu


def same(a, b):
    a = np.asarray(a, dtype=np.uint32).reshape(-1, 3)
    b = np.asarray(b, dtype=np.uint32).reshape(-1, 3)
    return a.shape == b.shape and bool((a == b).all())


def test_config4_stream_64mib_chain64_window_16mib():
    """BASELINE config 4, streaming leg, as written: ONE 64 MiB stream, 1 MiB
    blocks, no Reset, params D = M4{MaxDistance 16777200, ChainLength 64,
    HashLen 5, DistanceBitCost 66}: 64 chain hops AND candidates many MiB back
    AND the slide at len(history) > 32 MiB, in one launch, against the oracle
    block by block.  Far-back copies are spliced in so that long matches between
    1 and 16 MiB back exist (independent text chunks only share words)."""
    kw = PARAMS_D
    n, block = 64 << 20, 1 << 20
    host = np.empty(n, dtype=np.uint8)
    synth.fill(host, block, 4000, "text", 8)
    rng = np.random.default_rng(11)
    for b in range(16, 64):  # copies from 2..15 MiB back, inside the window, odd offsets
        for back in (2, 7, 15):
            ln = int(rng.integers(2000, 60000))
            dst = b * block + int(rng.integers(0, block - ln))
            src = dst - back * block - int(rng.integers(0, 5000))
            host[dst:dst + ln] = host[src:src + ln]
    g, o = m4.M4(**kw), OracleM4(**kw)
    got, off = g.find_matches_blocks(host, block)
    far = 0
    for b in range(n // block):
        ref = o.FindMatches(host[b * block:(b + 1) * block])
        assert same(got[int(off[b]):int(off[b + 1])], ref), f"block {b}"
        far += int((ref[:, 2] > (1 << 20)).sum())
    assert far > 1000, "the input does not exercise the large window"
    assert g.history_len() == o.history_len() and g.history_len() < n  # the slide fired
    tail = synth.text(100000, 3)
    assert same(g.find_matches(tail), o.FindMatches(tail))


def test_config4_batch_1mib_chunks_params_d():
    """BASELINE config 4, batch leg: params D on 1 MiB independent chunks (text,
    log, and a chunk with long internal repeats), matches and histograms."""
    import torch
    n = 1 << 20
    parts = [np.frombuffer(synth.text(n, 60), dtype=np.uint8), np.frombuffer(synth.log(n, 61), dtype=np.uint8),
             np.frombuffer(synth.text(n, 62), dtype=np.uint8).copy(), np.frombuffer(synth.text(n, 63), dtype=np.uint8)]
    parts[2][600000:900000] = parts[2][100000:400000]
    host = np.concatenate(parts)
    nchunks = len(parts)
    dev = torch.device("cuda:0")
    d_src = torch.from_numpy(host).to(dev)
    plan = m4.BatchPlan(nchunks, n, 0, **PARAMS_D)
    off = np.arange(nchunks + 1, dtype=np.uint64) * n
    d_out = torch.empty(3 * (n // 4 + 2) * nchunks, dtype=torch.int32, device=dev)
    d_off = torch.empty(nchunks + 1, dtype=torch.int64, device=dev)
    d_h = torch.empty(nchunks * 1024, dtype=torch.int32, device=dev)
    plan.run(d_src, off, d_out, d_off, d_h)
    plan.status()
    o = d_off.cpu().numpy()
    out = d_out.cpu().numpy().view(np.uint32).reshape(-1, 3)
    h = d_h.cpu().numpy().view(np.uint32).reshape(nchunks, 1024)
    for c in range(nchunks):
        chunk = host[c * n:(c + 1) * n]
        ref = OracleM4(**PARAMS_D).FindMatches(chunk)
        assert same(out[o[c]:o[c + 1]], ref), f"chunk {c}"
        lit, cmd, dist, _, _ = histograms(chunk, ref)
        assert (h[c, :256] == lit).all() and (h[c, 256:960] == cmd).all() and (h[c, 960:] == dist).all()


def test_k1_variants_agree_on_1gib_of_mixed_input():
    """The exchange kernel (k1_xchg.h) rests on two measured hardware orderings;
    the sorting kernel (kernels.h k1_chain_build) does not.  Their chain arrays
    must be identical, word for word, on 1 GiB of mixed input (text, log records,
    noise, zero pages, short periods, low entropy) -- compared on the device."""
    import torch
    L = m4.lib()
    L.m4g_debug_chain_words.restype = ctypes.c_uint64
    L.m4g_debug_chain_words.argtypes = [ctypes.c_void_p]
    L.m4g_debug_k1_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p,
                                   ctypes.c_void_p]
    n, C = 1 << 20, 1024
    host = np.empty(n * C, dtype=np.uint8)
    synth.fill(host[:n * 640], n, 7000, "text", 8)
    synth.fill(host[n * 640:n * 896], n, 8000, "log", 8)
    rng = np.random.default_rng(17)
    for c in range(896, C):
        chunk = host[c * n:(c + 1) * n]
        k = c % 4
        if k == 0:
            chunk[:] = rng.integers(0, 256, n, dtype=np.uint8)
        elif k == 1:
            chunk[:] = 0
            chunk[rng.integers(0, n, 50)] = 1
        elif k == 2:
            p = int(rng.integers(1, 40))
            chunk[:] = np.resize(rng.integers(0, 256, p, dtype=np.uint8), n)
        else:
            chunk[:] = rng.integers(0, 3, n, dtype=np.uint8)
    dev = torch.device("cuda:0")
    d_src = torch.empty(n * C + 64, dtype=torch.uint8, device=dev)
    d_src[:n * C].copy_(torch.from_numpy(host))
    kw = dict(MaxDistance=1 << 20, DistanceBitCost=66)
    plan = m4.BatchPlan(C, n, 0, **kw)
    off = np.arange(C + 1, dtype=np.uint64) * n
    d_out = torch.empty(3 * (n // 4 + 2) * C, dtype=torch.int32, device=dev)
    d_off = torch.empty(C + 1, dtype=torch.int64, device=dev)
    plan.run(d_src, off, d_out, d_off, None)
    plan.status()
    del d_out
    words = int(L.m4g_debug_chain_words(plan._h))
    chains = []
    try:
        for variant in (0, 1, 2):  # sorting kernel, exchange kernel, split K1 (k1_split.h)
            L.m4g_debug_set_k1_variant(variant)
            ch = torch.zeros(words, dtype=torch.int32, device=dev)
            rc = L.m4g_debug_k1_run(plan._h, d_src.data_ptr(), n * C, C, ch.data_ptr(), None)
            assert rc == 0, L.m4g_last_error()
            torch.cuda.synchronize()
            chains.append(ch)
    finally:
        L.m4g_debug_set_k1_variant(-1)
    assert int((chains[0] != 0).sum()) > n * C // 2  # the arrays hold real entries
    assert torch.equal(chains[0], chains[1])
    assert torch.equal(chains[1], chains[2])
    # and they are the reference's chain (low 31 bits = delta), on a sample of chunks
    for c in (0, 700, 897, 898, 899):
        chunk = host[c * n:(c + 1) * n]
        o = OracleM4(ChainLength=1, **kw)
        o.FindMatches(chunk)
        got = chains[1][c * n:(c + 1) * n].cpu().numpy().view(np.uint32) & np.uint32(0x7FFFFFFF)
        assert (got[:n - 7] == o.dump_chain()[:n - 7]).all(), f"chunk {c}"
