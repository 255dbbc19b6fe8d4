"""Randomised parity: random M4 parameter sets x random structured inputs,
HIP path vs the oracle, through the batch API (fresh state per chunk) and the
streaming handle (carried history, slides).  Seeds are fixed: a failure names
its case and reproduces."""
import numpy as np
import pytest

from brotli_amd import m4, synth
from oracle_lib import OracleM4, histograms, lib as oracle_clib

pytestmark = pytest.mark.gpu


def same(got, ref):
    return got.shape == ref.shape and (got == ref).all()


def random_params(rng):
    kw = dict(
        MaxDistance=int(rng.choice([64, 1000, 4096, 65535, 1 << 18, 1 << 20])),
        MinLength=int(rng.choice([0, 3, 4, 5, 6, 8])),
        HashLen=int(rng.choice([0, 4, 5, 6, 7, 8])),
        TableBits=int(rng.choice([0, 6, 10, 14, 16, 17, 18, 20])),
        ChainLength=int(rng.choice([0, 0, 1, 3, 16])),
        DistanceBitCost=int(rng.choice([0, 30, 66, 200])),
    )
    return kw


def random_input(rng, n):
    kind = rng.integers(0, 8)
    if n == 0:
        return b""
    if kind == 6:  # incompressible
        return bytes(rng.integers(0, 256, n, dtype=np.uint8))
    if kind == 7:  # regions of different character, sharp transitions
        out = bytearray()
        while len(out) < n:
            m = int(rng.integers(1, 30000))
            out += random_input(rng, m) if rng.random() < 0.7 else bytes(m)
        return bytes(out[:n])
    if kind == 0:  # small alphabet noise
        return bytes(rng.integers(0, int(rng.integers(2, 6)), n, dtype=np.uint8))
    if kind == 1:  # phrases from a small dictionary, with mutations
        words = [bytes(rng.integers(97, 123, int(rng.integers(2, 12)), dtype=np.uint8)) for _ in range(40)]
        out = bytearray()
        while len(out) < n:
            out += words[int(rng.integers(0, len(words)))]
            if rng.random() < 0.1:
                out.append(int(rng.integers(0, 256)))
        return bytes(out[:n])
    if kind == 2:  # runs and short periods
        out = bytearray()
        while len(out) < n:
            period = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
            out += period * int(rng.integers(1, 400))
        return bytes(out[:n])
    if kind == 3:
        return synth.text(n, int(rng.integers(0, 1 << 30)))
    if kind == 4:
        return synth.log(n, int(rng.integers(0, 1 << 30)))
    # a block copied around with gaps of noise (long-distance matches)
    block = bytes(rng.integers(0, 256, int(rng.integers(16, 3000)), dtype=np.uint8))
    out = bytearray()
    while len(out) < n:
        out += block[: int(rng.integers(8, len(block) + 1))]
        out += bytes(rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8))
    return bytes(out[:n])


@pytest.mark.parametrize("seed", range(32))
def test_fuzz_batch(seed):
    rng = np.random.default_rng(1000 + seed)
    kw = random_params(rng)
    sizes = [int(rng.choice([0, 1, 7, 8, 9, 63, 64, 3071, 3072, 3073, 4097])) for _ in range(4)] + \
            [int(rng.integers(10, 150000)) for _ in range(8)]
    chunks = [random_input(rng, s) for s in sizes]
    blob = b"".join(chunks)
    off = np.concatenate([[0], np.cumsum([len(c) for c in chunks])]).astype(np.uint64)
    out, out_off, histos = m4.find_matches_batch(blob, off, (0,), True, **kw)
    for c, chunk in enumerate(chunks):
        ref = OracleM4(**kw).FindMatches(chunk)
        got = out[int(out_off[c]):int(out_off[c + 1])]
        assert same(got, ref), f"seed {seed} chunk {c} (len {len(chunk)}) params {kw}"
        lit, cmd, dist, _, _ = histograms(chunk, ref)
        assert (histos[c, :256] == lit).all() and (histos[c, 256:960] == cmd).all() and \
            (histos[c, 960:] == dist).all(), f"seed {seed} chunk {c} histograms, params {kw}"


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_streaming(seed):
    rng = np.random.default_rng(2000 + seed)
    kw = random_params(rng)
    kw["MaxDistance"] = int(rng.choice([64, 1000, 4096, 30000]))  # small windows: the slide fires
    data = b"".join(random_input(rng, int(rng.integers(1000, 60000))) for _ in range(6))
    g, o = m4.M4(**kw), OracleM4(**kw)
    pos, prev = 0, 0
    while pos < len(data):
        n = int(rng.choice([1, 5, 8, 100, 4096, 20000, 65536]))
        block = data[pos:pos + n]
        seed_prev = prev if rng.random() < 0.3 else 0  # caller passes a non-empty dst sometimes
        if seed_prev and seed_prev > min(oracle_clib().m4ref_history_len(o._h), kw["MaxDistance"]):
            seed_prev = 0
        ref = o.FindMatches(block, seed_prev)
        got = g.find_matches(block, seed_prev)
        assert same(got, ref), f"seed {seed} block at {pos} (len {len(block)}) params {kw}"
        prev = int(ref[-1][2]) if len(ref) and ref[-1][2] else prev
        pos += n
