"""Randomised parity of the GPU encoders and of m4g_writer against the restated reference encoders: random mixes of
chunk kinds (text, log records, noise, low entropy, short periods, splices of earlier data) and sizes, random M4
fields, every format.  Where the restated flate encoder refuses an input (a match longer than 258 bytes -- the
reference panics, flate/token.go:59-71) the GPU encoder must refuse it too, and nowhere else."""
import io
import zlib

import numpy as np
import pytest

from brotli_amd import m4, synth
from oracle_lib import OracleEncoder, OracleFlateEncoder, OracleM4, brotli_decompress, flate_compress as oracle_flate_compress
from test_gpu_encode import _encode_on_gpu as brotli_on_gpu
from test_gpu_flate import _encode_on_gpu as flate_on_gpu

pytestmark = pytest.mark.gpu


def piece(rng, size, flate_safe):
    kind = int(rng.integers(0, 7))
    if kind == 0:
        return synth.text(size, int(rng.integers(0, 1000)))
    if kind == 1:
        return synth.log(size, int(rng.integers(0, 1000)))
    if kind == 2:
        return bytes(rng.integers(0, 256, size, dtype=np.uint8))
    if kind == 3:
        return bytes(rng.integers(0, int(rng.integers(2, 20)), size, dtype=np.uint8))
    if kind == 4:  # words from a small vocabulary: many short matches, repeat distances
        vocab = [bytes(rng.integers(97, 123, int(rng.integers(2, 12)), dtype=np.uint8)) for _ in range(int(rng.integers(4, 200)))]
        out = bytearray()
        while len(out) < size:
            out += vocab[int(rng.integers(0, len(vocab)))] + b" "
        return bytes(out[:size])
    if kind == 5 and not flate_safe:  # a short period: one very long match
        unit = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
        return (unit * (size // len(unit) + 1))[:size]
    # splices: copies of 20..200 bytes from earlier in the piece
    base = bytearray(synth.text(size, int(rng.integers(0, 1000))))
    for _ in range(size // 300):
        ln = int(rng.integers(20, 200))
        if size <= 2 * ln:
            break
        a, b = int(rng.integers(0, size - ln)), int(rng.integers(0, size - ln))
        base[b:b + ln] = base[a:a + ln]
    return bytes(base)


def chunks_for(rng, flate_safe):
    n = int(rng.integers(1, 7))
    sizes = [int(rng.choice([0, 1, 7, 8, 9, 100, 3000, 40000, 65535, 65536, 70000, 150000])) for _ in range(n)]
    out = []
    for s in sizes:
        parts, left = [], s
        while left > 0:
            k = left if rng.random() < 0.6 else int(rng.integers(1, left + 1))
            parts.append(piece(rng, k, flate_safe))
            left -= k
        out.append(b"".join(parts))
    return out


def m4_fields(rng, flate):
    if flate:
        return m4.flate_m4_params(int(rng.integers(5, 8)))
    kw = dict(MaxDistance=int(rng.choice([1 << 15, 1 << 16, 1 << 18, 1 << 20])), DistanceBitCost=int(rng.choice([0, 66])))
    if rng.random() < 0.4:
        kw.update(ChainLength=int(rng.choice([2, 8, 32])), HashLen=int(rng.choice([4, 5, 6])))
    if rng.random() < 0.3:
        kw.update(MinLength=int(rng.choice([3, 4, 6])))
    return kw


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_batch_brotli_encoder(seed):
    rng = np.random.default_rng(1000 + seed)
    kw, chunks = m4_fields(rng, False), chunks_for(rng, False)
    stream, _ = brotli_on_gpu(chunks, kw)
    enc, want = OracleEncoder(), b""
    for c in chunks:
        want += enc.Encode(c, OracleM4(**kw).FindMatches(c), False)
    want += enc.Encode(b"", np.zeros((0, 3), np.uint32), True)
    data = b"".join(chunks)
    assert stream == want
    assert brotli_decompress(stream, len(data)) == data


@pytest.mark.parametrize("seed", range(14))
def test_fuzz_batch_flate_encoder(seed):
    rng = np.random.default_rng(2000 + seed)
    gz = bool(rng.integers(0, 2))
    kw, chunks = m4_fields(rng, True), chunks_for(rng, flate_safe=seed % 3 != 0)
    enc, want, refused = OracleFlateEncoder(gz, 5), b"", False
    try:
        for c in chunks:
            want += enc.Encode(c, OracleM4(**kw).FindMatches(c), False)
        want += enc.Encode(b"", np.zeros((0, 3), np.uint32), True)
    except ValueError:
        refused = True
    if refused:
        with pytest.raises(m4.M4GError):
            flate_on_gpu(chunks, kw, gzip=gz, mtime=5)
        return
    stream, _, _ = flate_on_gpu(chunks, kw, gzip=gz, mtime=5)
    assert stream == want
    assert zlib.decompress(stream, 31 if gz else -15) == b"".join(chunks)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_writer(seed):
    rng = np.random.default_rng(3000 + seed)
    fmt = int(rng.integers(0, 3))
    kw = m4_fields(rng, fmt != m4.FORMAT_BROTLI)
    bs = int(rng.choice([4096, 10000, 1 << 16, 1 << 16, 1 << 17]))
    data = b"".join(chunks_for(rng, flate_safe=seed % 4 != 0))
    try:
        if fmt == m4.FORMAT_BROTLI:
            want = OracleM4(**kw).compress(data, bs)[0]
        else:
            want = oracle_flate_compress(OracleM4(**kw), data, bs, gzip=fmt == m4.FORMAT_GZIP, mtime=9)
    except ValueError:
        want = None
    dest = io.BytesIO()
    w = m4.GPUWriter(dest, fmt, bs, mtime=9, **kw)

    def feed():
        p = 0
        while p < len(data):
            k = int(rng.choice([1, 50, bs - 1, bs, bs + 1, 3 * bs, 10 * bs + 17, len(data)]))
            w.Write(data[p:p + k])
            p += k
        w.Close()
    if want is None:
        with pytest.raises(m4.M4GError):
            feed()
        return
    feed()
    assert dest.getvalue() == want
