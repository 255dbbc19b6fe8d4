"""m4g_writer: matchfinder.Writer (matchfinder/matchfinder.go:44-103) with the match finder and the encoder on
the GPU, against the restated Writer (oracle/encoder_ref.c, oracle/flate_ref.c): the bytes handed to Dest must be
the same after any sequence of Write calls, Close and Reset, for the three encoders of the reference."""
import gzip as pygzip
import io
import zlib

import numpy as np
import pytest

from brotli_amd import m4, synth
from cases import PARAMS_G, PARAMS_REF_TESTS, opticks
from oracle_lib import OracleM4, brotli_decompress, flate_compress as oracle_flate_compress

pytestmark = pytest.mark.gpu


def oracle_stream(fmt, kw, data, bs):
    if fmt == m4.FORMAT_BROTLI:
        return OracleM4(**kw).compress(data, bs)[0]
    return oracle_flate_compress(OracleM4(**kw), data, bs, gzip=fmt == m4.FORMAT_GZIP, mtime=77)


def decode(fmt, stream, n):
    if fmt == m4.FORMAT_BROTLI:
        return brotli_decompress(stream, n)
    return pygzip.decompress(stream) if fmt == m4.FORMAT_GZIP else zlib.decompress(stream, -15)


FORMATS = [(m4.FORMAT_BROTLI, PARAMS_G), (m4.FORMAT_BROTLI, PARAMS_REF_TESTS[0]), (m4.FORMAT_FLATE, m4.flate_m4_params(5)),
           (m4.FORMAT_GZIP, m4.flate_m4_params(7))]


@pytest.mark.parametrize("fmt,kw", FORMATS, ids=["brotli-G", "brotli-ref0", "flate5", "gzip7"])
def test_writer_equals_the_restated_writer_for_any_write_pattern(fmt, kw):
    data = opticks()
    want = oracle_stream(fmt, kw, data, 1 << 16)
    rng = np.random.default_rng(4)
    for pattern in ("one", "blocks", "ragged", "bytes-then-rest"):
        dest = io.BytesIO()
        w = m4.GPUWriter(dest, fmt, 1 << 16, mtime=77, **kw)
        if pattern == "one":
            w.Write(data)
        elif pattern == "blocks":
            for p in range(0, len(data), 1 << 16):
                w.Write(data[p:p + (1 << 16)])
        elif pattern == "ragged":
            p = 0
            while p < len(data):
                k = int(rng.choice([1, 100, 65535, 65536, 65537, 200000, 300001]))
                w.Write(data[p:p + k])
                p += k
        else:
            w.Write(data[:3])
            w.Write(b"")
            w.Write(data[3:])
        w.Close()
        got = dest.getvalue()
        assert got == want, pattern
        assert decode(fmt, got, len(data)) == data
    # Reset: a second stream on the same writer
    dest2 = io.BytesIO()
    w.Reset(dest2)
    w.Write(data[:200000])
    w.Close()
    assert dest2.getvalue() == oracle_stream(fmt, kw, data[:200000], 1 << 16)


@pytest.mark.parametrize("fmt,kw", FORMATS[::2] + FORMATS[3:], ids=["brotli-G", "flate5", "gzip7"])
def test_writer_edge_streams(fmt, kw):
    rng = np.random.default_rng(8)
    noise = bytes(rng.integers(0, 256, 150000, dtype=np.uint8))
    text = synth.text(400000, 9)
    for data, bs in ((b"", 1 << 16), (b"abc", 1 << 16), (text[:1 << 16], 1 << 16), (text[:1 << 17], 1 << 16),
                     (noise + text + noise, 1 << 16), (text[:100000], 4096), (noise[:4096 * 3], 4096),
                     (text[:150000], 0), (text[:70000], 1000)):
        dest = io.BytesIO()
        w = m4.GPUWriter(dest, fmt, bs, mtime=77, **kw)
        half = len(data) // 2
        w.Write(data[:half])
        w.Write(data[half:])
        w.Close()
        if bs == 0:  # every Write is one block
            want = None
        else:
            want = oracle_stream(fmt, kw, data, bs)
        got = dest.getvalue()
        assert decode(fmt, got, len(data)) == data, (len(data), bs)
        if want is not None:
            assert got == want, (len(data), bs)


@pytest.mark.parametrize("fmt,kw", FORMATS[::2] + FORMATS[3:], ids=["brotli-G", "flate5", "gzip7"])
def test_writer_block_size_0_empty_write_and_kept_output(fmt, kw):
    """BlockSize 0: every Write is one block (matchfinder.go:64-66), an empty one included; and a Write whose
    output exceeds the wrapper's 1 MiB buffer goes through the kept-output retry (a call with n = 0)."""
    from oracle_lib import OracleEncoder, OracleFlateEncoder
    rng = np.random.default_rng(12)
    noise = bytes(rng.integers(0, 256, 1_300_000, dtype=np.uint8))  # incompressible: > 1 MiB of output
    text = synth.text(200000, 5)
    writes = [text[:70000], b"", noise, b"", text[70000:]]
    mf = OracleM4(**kw)
    enc = OracleEncoder() if fmt == m4.FORMAT_BROTLI else OracleFlateEncoder(gzip=fmt == m4.FORMAT_GZIP, mtime=77)
    want = b""
    for p in writes:
        want += enc.Encode(p, mf.FindMatches(p), False)
    want += enc.Encode(b"", mf.FindMatches(b""), True)
    dest = io.BytesIO()
    w = m4.GPUWriter(dest, fmt, 0, mtime=77, **kw)
    for p in writes:
        w.Write(p)
    w.Close()
    got = dest.getvalue()
    data = b"".join(writes)
    assert decode(fmt, got, len(data)) == data
    assert got == want


def test_writer_large_stream_and_small_output_buffer():
    """32 MiB through the one-launch path in two Writes (flate: the 32 KiB window slides at every block); the
    Python wrapper starts with a 1 MiB output buffer, so the kept-output / retry protocol is exercised too."""
    n = 32 << 20
    host = np.empty(n, dtype=np.uint8)
    synth.fill(host, 1 << 20, 0, "text", 16)
    data = host.tobytes()
    for fmt, kw in ((m4.FORMAT_BROTLI, PARAMS_G), (m4.FORMAT_GZIP, m4.flate_m4_params(6))):
        dest = io.BytesIO()
        w = m4.GPUWriter(dest, fmt, 1 << 16, mtime=77, **kw)
        w.Write(data[:n // 2 + 12345])
        w.Write(data[n // 2 + 12345:])
        w.Close()
        got = dest.getvalue()
        assert decode(fmt, got, n) == data
        k = 3 << 20
        want = oracle_stream(fmt, kw, data[:k + 70000], 1 << 16)
        m = len(oracle_stream(fmt, kw, data[:k], 1 << 16)) - 64
        assert got[:m] == want[:m]


def test_writer_token_range_error_and_reset():
    dest = io.BytesIO()
    w = m4.GPUWriter(dest, m4.FORMAT_FLATE, 1 << 16, **m4.flate_m4_params(5))
    with pytest.raises(m4.M4GError):
        w.Write(synth.text(70000, 1) + b"a" * 70000 + synth.text(70000, 2))
        w.Close()
    dest2 = io.BytesIO()
    w.Reset(dest2)
    data = synth.text(150000, 3)
    w.Write(data)
    w.Close()
    assert zlib.decompress(dest2.getvalue(), -15) == data


@pytest.mark.parametrize("fmt,kw", [(m4.FORMAT_BROTLI, PARAMS_G), (m4.FORMAT_FLATE, m4.flate_m4_params(5)),
                                    (m4.FORMAT_GZIP, m4.flate_m4_params(6))], ids=["brotli-G", "flate5", "gzip6"])
def test_hostbatch_compress_equals_the_restated_encoders(fmt, kw):
    """m4g_hostbatch_compress: chunks in host memory -> one stream, sliced and overlapped on the device; the bytes
    must be what the restated encoder writes for the same chunks (fresh M4 per chunk), however the batch is sliced."""
    from oracle_lib import OracleEncoder, OracleFlateEncoder
    sizes = [70000, 0, 1 << 17, 3, 100000, 65536, 1, 90000] * 40 + [50000]  # 321 chunks
    n = sum(sizes)
    host = np.frombuffer(synth.text(n, 21), dtype=np.uint8)
    off = np.cumsum([0] + sizes).astype(np.uint64)
    chunks = [host[int(off[c]):int(off[c + 1])].tobytes() for c in range(len(sizes))]
    enc = OracleEncoder() if fmt == m4.FORMAT_BROTLI else OracleFlateEncoder(fmt == m4.FORMAT_GZIP, 77)
    want = b""
    for c in chunks:
        want += enc.Encode(c, OracleM4(**kw).FindMatches(c), False)
    want += enc.Encode(b"", np.zeros((0, 3), np.uint32), True)
    import os
    # default slices (one here) and 6 slices on one device; the device list (0, 0, 0): three lanes with a contiguous
    # third of the chunks each (107 chunks = one or four slices), the bit writer passing from lane to lane
    for slice_env, devices in ((None, (0,)), ("64", (0,)), (None, (0, 0, 0)), ("32", (0, 0, 0))):
        if slice_env:
            os.environ["M4G_PIPE_SLICE"] = slice_env
        try:
            hb = m4.HostBatch(len(sizes), 1 << 17, devices, **kw)
            got = hb.compress(host, off, fmt, mtime=77)
            assert got == want, (slice_env, devices)
            if len(devices) > 1:  # fewer chunks than lanes, and a lane without chunks
                assert hb.compress(host[:int(off[2])], off[:3], fmt, mtime=77) == \
                    m4.HostBatch(2, 1 << 17, (0,), **kw).compress(host[:int(off[2])], off[:3], fmt, mtime=77)
            got2 = hb.compress(host[:int(off[5])], off[:6], fmt, mtime=77)  # the handle again, fewer chunks
            assert decode(fmt, got2, int(off[5])) == host[:int(off[5])].tobytes()
            assert decode(fmt, hb.compress(b"", np.zeros(1, np.uint64), fmt, mtime=77), 0) == b""
            hb.close()
        finally:
            os.environ.pop("M4G_PIPE_SLICE", None)
    assert decode(fmt, want, n) == host.tobytes()
