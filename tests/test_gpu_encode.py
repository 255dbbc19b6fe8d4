"""GPU parity for SURVEY.md section 8(f).2: brotli.Encoder.Encode with its second
pass on the GPU (m4g_batch_encode: K5 codes -> host prefix codes -> K6 bit
emission), against the restated Encoder (oracle/encoder_ref.c) byte for byte and
through a brotli decoder (the reference's own criterion, brotli_test.go:626-649)."""
import numpy as np
import pytest

from brotli_amd import m4, synth
from cases import PARAMS_D, PARAMS_G, edge_inputs, hand_kats, opticks
from oracle_lib import OracleEncoder, OracleM4, brotli_decompress

pytestmark = pytest.mark.gpu


def _encode_on_gpu(chunks, kw, first=True, last=True):
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
    d_h = torch.empty(C * 1024, dtype=torch.int32, device=dev)
    d_codes = torch.empty(2 * cap, dtype=torch.int32, device=dev)
    plan.run(d_src, off, d_out, d_off, d_h, d_codes=d_codes)
    plan.status()
    stream, ms = plan.encode(d_src, C, d_out, d_off, d_h, d_codes, first=first, last=last)
    return stream, ms


def _encode_with_oracle(chunks, kw):
    enc, out = OracleEncoder(), b""
    for c in chunks:
        out += enc.Encode(c, OracleM4(**kw).FindMatches(c), False)  # fresh M4 per chunk = AutoReset
    return out + enc.Encode(b"", np.zeros((0, 3), np.uint32), True)


def _check(chunks, kw):
    stream, _ = _encode_on_gpu(chunks, kw)
    data = b"".join(chunks)
    assert brotli_decompress(stream, len(data)) == data
    assert stream == _encode_with_oracle(chunks, kw)


@pytest.mark.parametrize("kw", [PARAMS_G, dict(MaxDistance=1 << 16, ChainLength=8, HashLen=5, DistanceBitCost=66)],
                         ids=["G", "chain8"])
def test_encode_text_and_log_chunks(kw):
    chunks = [synth.text(65536, 40), synth.text(100000, 41), synth.log(65536, 42), synth.text(70001, 43),
              synth.text(1 << 17, 44)]
    _check(chunks, kw)


def test_encode_edge_chunks():
    """empty / tiny chunks, one-symbol alphabets (count <= 1 and <= 4 trees,
    brotli_bit_stream.go:655-701), incompressible bytes (one literal run per chunk:
    written by whole waves), zero pages, the hand-traced KAT inputs"""
    rng = np.random.default_rng(3)
    e = edge_inputs()
    chunks = [e["text_1000"], b"", e["short3"], e["len8"], e["kat_run40"], e["zeros_70k"], e["random_20k"],
              bytes(rng.integers(0, 256, 200000, dtype=np.uint8)), e["lowent_50k"], e["period_mix"], b"ab" * 40,
              b"abcd" * 50, e["log_65536"]] + [v[1] for v in hand_kats().values() if v[0] == dict()]
    # alphabets of 2..4 literals in every order of frequencies: the 'simple' prefix codes with their depth sort and
    # tree-select bit (tests/test_oracle.py: test_simple_prefix_codes_of_two_to_four_symbols_decode)
    for counts in ((9, 1), (1, 9), (5, 3, 1), (1, 3, 5), (8, 4, 2, 1), (1, 2, 4, 8), (3, 3, 3, 3), (1, 1, 1, 6), (2, 9, 1, 3)):
        raw = b"".join(bytes([s]) * (c * 13) for s, c in zip(b"qcxk", counts))
        chunks.append(bytes(rng.permutation(np.frombuffer(raw, dtype=np.uint8))))
    _check(chunks, PARAMS_G)
    _check([b""], PARAMS_G)
    _check([b"", b"x"], PARAMS_G)


def test_encode_opticks_1mib_chunks_params_d():
    data = opticks()
    _check([data[:300000], data[300000:]], PARAMS_D)


def test_encode_full_size_batch_decodes():
    """64 x 1 MiB of text: the stream decodes to the input (size-independent property;
    byte identity with the restated Encoder on the first 1 MiB-chunk prefix)."""
    n, C = 1 << 20, 64
    host = np.empty(n * C, dtype=np.uint8)
    synth.fill(host, n, 500, "text", 8)
    chunks = [host[c * n:(c + 1) * n].tobytes() for c in range(C)]
    stream, ms = _encode_on_gpu(chunks, PARAMS_G)
    assert brotli_decompress(stream, n * C) == host.tobytes()
    ref = _encode_with_oracle(chunks[:2], PARAMS_G)
    # same bytes up to the last full byte of chunk 1's meta-block (the oracle's stream ends there with ISLAST)
    assert stream[:len(ref) - 2] == ref[:len(ref) - 2]


@pytest.mark.parametrize("kw", __import__("cases").PARAMS_REF_TESTS[:3] + [PARAMS_G],
                         ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()) or "defaults")
def test_writer_drop_in_on_the_reference_test_file(kw):
    """SURVEY config 1 end to end on the GPU: matchfinder.Writer{M4{...}, brotli.Encoder, BlockSize 1 << 16} over
    Isaac.Newton-Opticks.txt (brotli_test.go:626-649, 730-744): match finding of all blocks with the history carried
    over (m4g_find_matches_blocks) + both passes of Encoder.Encode (m4g_brotli_encode_blocks).  The stream decodes to
    the file and equals, byte for byte, the restated Writer's."""
    data = opticks()
    out = m4.brotli_compress(data, 1 << 16, **kw)
    assert brotli_decompress(out, len(data)) == data
    want, _ = OracleM4(**kw).compress(data, 1 << 16)
    assert out == want


def test_writer_drop_in_edge_sizes():
    rng = np.random.default_rng(11)
    noise = bytes(rng.integers(0, 256, 70000, dtype=np.uint8))
    text = synth.text(300000, 5)
    for data, bs in ((b"", 1 << 16), (b"abc", 1 << 16), (text[:1 << 16], 1 << 16), (text[:1 << 17], 1 << 16),
                     (text, 1 << 16), (noise + text[:50000] + noise, 1 << 16), (text[:100000], 4096),
                     (noise[:4096 * 3], 4096), (bytes(200000), 1 << 16)):
        out = m4.brotli_compress(data, bs, **PARAMS_G)
        assert brotli_decompress(out, len(data)) == data
        assert out == OracleM4(**PARAMS_G).compress(data, bs)[0], (len(data), bs)


ERR_BAD_PARAM = -1000002  # M4G_ERR_BAD_PARAM (include/m4g.h)


def test_lists_that_run_past_their_block_are_refused():
    """Encoder.Encode slices src[pos : pos + Unmatched] for every match (encoder.go:45-52) and panics on a list that
    runs past the block; the C-ABI forms that take a caller's lists refuse them before any kernel reads by them."""
    import ctypes
    data = np.frombuffer(synth.text(100000, 3), dtype=np.uint8)
    bs = 1 << 16
    matches, off = m4.M4(**PARAMS_G).find_matches_blocks(data, bs)
    out = np.zeros(data.size + 4096, dtype=np.uint8)

    def brotli(m, o):
        m = np.ascontiguousarray(m, dtype=np.uint32)
        o = np.ascontiguousarray(o, dtype=np.uint64)
        return m4.lib().m4g_brotli_encode_blocks(0, data.ctypes.data, data.size, bs, m.ctypes.data, o.ctypes.data,
                                                 out.ctypes.data, out.size)

    def flate(m, o):
        m = np.ascontiguousarray(m, dtype=np.uint32)
        o = np.ascontiguousarray(o, dtype=np.uint64)
        return m4.lib().m4g_flate_encode_blocks(0, data.ctypes.data, data.size, bs, m.ctypes.data, o.ctypes.data, 0, 0,
                                                out.ctypes.data, out.size, None)

    assert brotli(matches, off) > 0
    bad = matches.copy()
    bad[5, 0] += 70000                      # literals beyond the block
    assert brotli(bad, off) == ERR_BAD_PARAM and flate(bad, off) == ERR_BAD_PARAM
    bad = matches.copy()
    bad[int(off[1]) - 1, 1] += 1            # the last match of block 0 one byte too long
    assert brotli(bad, off) == ERR_BAD_PARAM
    bad = matches.copy()
    bad[3, 0] = 0xFFFFFFFF                  # sums that wrap 32 bits
    assert brotli(bad, off) == ERR_BAD_PARAM
    bad = matches.copy()
    bad[7, 2] = (1 << 26) - 3               # distance symbol 64: past the Encoder's histogram (encoder.go:35, 97)
    assert brotli(bad, off) == ERR_BAD_PARAM
    bad[7, 2] = (1 << 26) - 4               # symbol 63 is the last one
    assert brotli(bad, off) > 0
    boff = off.copy()
    boff[1] = off[2] + 1                    # offsets not ascending
    assert brotli(matches, boff) == ERR_BAD_PARAM
