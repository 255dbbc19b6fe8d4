"""GPU parity for SURVEY.md section 8(f).4: the reference's flate / gzip encoders
(flate/huffman_bit_writer.go, flate/gzip.go) with symbol counting and token
writing on the GPU (m4g_batch_flate_encode: k5f -> host code construction ->
k6f), against the restated encoder (oracle/flate_ref.c) byte for byte and
through zlib / gzip inflation (the reference's own criterion,
flate/flate_test.go:14-38, 44-90)."""
import gzip as pygzip
import zlib

import numpy as np
import pytest

from brotli_amd import m4, synth
from cases import edge_inputs, opticks
from oracle_lib import OracleFlateEncoder, OracleM4

pytestmark = pytest.mark.gpu

LEVELS = {lv: m4.flate_m4_params(lv) for lv in (5, 6, 7)}


def _inflate(raw: bytes) -> bytes:
    d = zlib.decompressobj(-15)
    out = d.decompress(raw) + d.flush()
    assert d.eof and not d.unused_data
    return out


def _encode_on_gpu(chunks, kw, first=True, last=True, gzip=False, mtime=0):
    import torch
    dev = torch.device("cuda:0")
    host = np.frombuffer(b"".join(chunks), dtype=np.uint8)
    off = np.cumsum([0] + [len(c) for c in chunks]).astype(np.uint64)
    C, nmax = len(chunks), max(1, max(len(c) for c in chunks))
    d_src = torch.zeros(host.size + 64, dtype=torch.uint8, device=dev)
    if host.size:
        d_src[:host.size].copy_(torch.from_numpy(host.copy()))
    plan = m4.BatchPlan(C, nmax, 0, **kw)
    cap = sum(len(c) // 4 + 2 for c in chunks) + 1
    d_out = torch.empty(3 * cap, dtype=torch.int32, device=dev)
    d_off = torch.empty(C + 1, dtype=torch.int64, device=dev)
    plan.run(d_src, off, d_out, d_off, None)
    plan.status()
    return plan.flate_encode(d_src, C, d_out, d_off, first=first, last=last, gzip=gzip, mtime=mtime)


def _encode_with_oracle(chunks, kw, gzip=False, mtime=0):
    enc, out, kinds = OracleFlateEncoder(gzip, mtime), b"", []
    for c in chunks:
        out += enc.Encode(c, OracleM4(**kw).FindMatches(c), False)  # fresh M4 per chunk = AutoReset
        kinds.append(OracleFlateEncoder.KINDS.index(enc.last_kind))
    return out + enc.Encode(b"", np.zeros((0, 3), np.uint32), True), kinds  # Writer.Close(), nothing buffered


def _check(chunks, kw):
    stream, _, kinds = _encode_on_gpu(chunks, kw)
    data = b"".join(chunks)
    assert _inflate(stream) == data
    want, want_kinds = _encode_with_oracle(chunks, kw)
    assert kinds.tolist() == want_kinds
    assert stream == want
    return kinds


@pytest.mark.parametrize("level", [5, 6, 7])
def test_flate_text_and_log_chunks(level):
    chunks = [synth.text(65536, 40), synth.text(100000, 41), synth.log(65536, 42), synth.text(70001, 43),
              synth.text(1 << 17, 44)]
    kinds = _check(chunks, LEVELS[level])
    assert set(kinds.tolist()) == {2}


def test_flate_block_kinds_and_edge_chunks():
    """stored blocks (incompressible chunks of at most 65535 bytes: byte-aligned copies between bit-aligned
    neighbours), fixed blocks (tiny chunks), empty chunks, one literal run per chunk (written by whole waves),
    chunks too long to be stored"""
    rng = np.random.default_rng(3)
    e = edge_inputs()
    noise = bytes(rng.integers(0, 256, 200000, dtype=np.uint8))
    chunks = [e["text_1000"], b"", e["short3"], noise[:65535], e["len8"], noise[:4000], synth.text(30000, 7),
              noise[:65536], e["random_20k"], noise, e["lowent_50k"], b"ab" * 40, b"abcd" * 50, noise[:1], e["log_65536"]]
    kinds = _check(chunks, LEVELS[6])
    assert {0, 1, 2} <= set(kinds.tolist())
    assert kinds[3] == 0 and kinds[7] != 0  # 65535 bytes can be stored, 65536 cannot (huffman_bit_writer.go:26,273)
    _check([b""], LEVELS[5])
    _check([b"", b"x"], LEVELS[5])
    _check([noise[:100], noise[:3000]], LEVELS[5])


@pytest.mark.parametrize("level", [5, 7])
def test_gzip_framing(level):
    chunks = [synth.text(65536, 50), synth.log(40000, 51), b"", synth.text(100001, 52)]
    data = b"".join(chunks)
    stream, _, _ = _encode_on_gpu(chunks, LEVELS[level], gzip=True, mtime=1700000001)
    assert pygzip.decompress(stream) == data
    assert stream == _encode_with_oracle(chunks, LEVELS[level], gzip=True, mtime=1700000001)[0]
    # a stream made by two calls: the second continues at the bit where the first stopped
    a, _, _ = _encode_on_gpu(chunks[:2], LEVELS[level], first=True, last=False)
    whole, _, _ = _encode_on_gpu(chunks[:2], LEVELS[level], first=True, last=True)
    assert whole[:len(a) - 1] == a[:len(a) - 1]


def test_flate_opticks_in_writer_blocks():
    """The reference's test file in the blocks flate.NewWriter uses (BlockSize 1 << 16, flate/writer.go:62-66),
    every block with a fresh M4 (the batch semantics of this library)."""
    data = opticks()
    chunks = [data[p:p + (1 << 16)] for p in range(0, len(data), 1 << 16)]
    _check(chunks, LEVELS[7])


def test_match_longer_than_258_is_refused_like_the_reference_panics():
    chunks = [synth.text(5000, 1), b"a" * 300, synth.text(5000, 2)]
    with pytest.raises(m4.M4GError, match="chunk 1"):
        _encode_on_gpu(chunks, LEVELS[5])
    with pytest.raises(ValueError):
        _encode_with_oracle(chunks, LEVELS[5])


def test_flate_full_size_batch_inflates():
    """64 x 1 MiB of text at level 6: the stream inflates to the input, its gzip CRC is zlib's (size-independent
    properties); byte identity with the restated encoder on the first chunks."""
    n, C = 1 << 20, 64
    host = np.empty(n * C, dtype=np.uint8)
    synth.fill(host, n, 0, "text", 16)
    chunks = [host[c * n:(c + 1) * n].tobytes() for c in range(C)]
    stream, ms, kinds = _encode_on_gpu(chunks, LEVELS[6], gzip=True, mtime=7)
    assert pygzip.decompress(stream) == host.tobytes()
    want, _ = _encode_with_oracle(chunks[:3], LEVELS[6], gzip=True, mtime=7)
    assert stream[:len(want) - 16] == want[:len(want) - 16]


@pytest.mark.parametrize("level,gz", [(5, False), (6, True), (7, False), (7, True)])
def test_flate_writer_drop_in_on_the_reference_test_file(level, gz):
    """flate_test.go:44-90 (TestWriterLevels / TestGZIPWriterLevels) for the M4 levels, end to end on the GPU:
    match finding of all 64 KiB blocks with the history carried over (m4g_find_matches_blocks) + the encoder
    (m4g_flate_encode_blocks).  The stream inflates to the file and equals, byte for byte, the restated
    matchfinder.Writer{M4, flate encoder, BlockSize 1 << 16}."""
    from oracle_lib import flate_compress as oracle_flate_compress
    data = opticks()
    out = m4.flate_compress(data, level, gzip=gz, mtime=12345)
    assert (pygzip.decompress(out) if gz else _inflate(out)) == data
    assert out == oracle_flate_compress(OracleM4(**LEVELS[level]), data, 1 << 16, gzip=gz, mtime=12345)


def test_flate_writer_drop_in_edge_sizes():
    from oracle_lib import flate_compress as oracle_flate_compress
    rng = np.random.default_rng(9)
    noise = bytes(rng.integers(0, 256, 70000, dtype=np.uint8))
    text = synth.text(300000, 3)
    for data, bs in ((b"", 1 << 16), (b"abc", 1 << 16), (text[:1 << 16], 1 << 16), (text[:(1 << 17)], 1 << 16),
                     (text, 1 << 16), (noise + text[:50000] + noise, 1 << 16), (text[:100000], 4096),
                     (noise[:4096 * 3], 4096)):
        out = m4.flate_compress(data, 6, block_size=bs)
        assert _inflate(out) == data
        assert out == oracle_flate_compress(OracleM4(**LEVELS[6]), data, bs), (len(data), bs)
