"""GPU parity of m4g_bargain_* against the restated Bargain1 / Bargain2 (oracle/bargain_ref.c) -- SURVEY §8(f).3.
Match for Match, block for block (the costs are float32: equality of the Match lists means every rounding agreed)."""
import os

import numpy as np
import pytest

from brotli_amd import m4
from oracle_lib import OracleBargain, OracleEncoder, brotli_decompress
from test_bargain_oracle import OPTICKS, text

pytestmark = pytest.mark.gpu


def oracle_blocks(data, variant, md, skip, bs):
    return OracleBargain(variant, md, skip).find_blocks(data, bs)


def same(data, variant=1, md=1 << 20, skip=False, bs=1 << 16):
    got, goff = m4.bargain_find_blocks(data, variant, md, skip, bs)
    want, woff = oracle_blocks(data, variant, md, skip, bs)
    assert np.array_equal(goff, woff), (goff[:8], woff[:8])
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, (int(bad[0]), got[bad[0]].tolist(), want[bad[0]].tolist())


@pytest.mark.parametrize("variant", [1, 2, 3])
@pytest.mark.parametrize("skip", [False, True])
def test_opticks(variant, skip):
    """brotli_test.go:886-932: the reference's own test input and settings (MaxDistance 1 << 20, 1 << 16 blocks)."""
    data = open(OPTICKS, "rb").read()
    same(data, variant, 1 << 20, skip)


@pytest.mark.parametrize("variant", [1, 2, 3])
def test_kinds(variant):
    rng = np.random.default_rng(5)
    same(text(300000, 3), variant, bs=4096)
    same(text(200001, 4), variant, bs=5000)                                       # ragged last block
    same(bytes(rng.integers(0, 256, size=150000, dtype=np.uint8)), variant)       # no matches
    same(bytes(rng.integers(0, 4, size=150000, dtype=np.uint8)), variant, bs=8192)
    same(bytes(200000), variant, bs=65536)                                        # zeros: the empty table entry matches
    same(b"abcdefg" * 30000, variant, bs=65536)                                   # one long match per block
    same(text(70000, 8) * 3, variant, bs=65536)                                   # long repeats across blocks
    same(text(40000, 9) + bytes(30000) + text(40000, 9), variant, bs=16384)
    same(text(100000, 10), variant, md=0, bs=4096)                                # default MaxDistance 1 << 16
    same(text(100000, 11), variant, md=5000, bs=4096)
    same(text(600000, 14), variant, bs=1 << 18)                                   # blocks longer than the Writer's default


@pytest.mark.parametrize("variant,md", [(1, 1 << 20), (2, 1 << 16), (3, 1 << 20)])
def test_long_stream_is_segmented(variant, md):
    """A stream longer than 6 MiB is cut into 4 MiB segments for K1, each primed with the MaxDistance bytes before it;
    the second half repeats the first beyond MaxDistance (predecessors that exist but must be rejected)."""
    a = text(4_700_000, 30 + variant)
    same(a + a[:4_300_001], variant, md=md)


def test_hand_traced_kats_on_the_gpu():
    """The single-block known answers derived on paper in test_bargain_oracle.py (Skip at an odd position, the
    two-form addMatch of a back-extended candidate, its early exit), here straight against the GPU path."""
    from cases import bargain_kat_phrase_at_odd_position
    six, seven = bargain_kat_phrase_at_odd_position(b"UVWXYZ"), bargain_kat_phrase_at_odd_position(b"TUVWXYZ")
    for data, variant, skip, want in ((six, 1, False, [[33, 6, 25], [25, 0, 0]]), (six, 1, True, [[64, 0, 0]]),
                                      (seven, 1, True, [[33, 7, 25], [24, 0, 0]]), (seven, 1, False, [[33, 7, 25], [24, 0, 0]]),
                                      (six, 2, True, [[33, 6, 25], [25, 0, 0]]), (six, 3, True, [[33, 6, 25], [25, 0, 0]])):
        got, off = m4.bargain_find_blocks(data, variant, 1 << 20, skip)
        assert got.tolist() == want and off.tolist() == [0, len(want)], (variant, skip)
    P = b"0123456789abcdefghij"
    data = P + b"ABCDEFGH" + P[:10] + b"#" + P[11:] + b"KLMNOPQR"
    for variant in (1, 2, 3):
        assert m4.bargain_find_blocks(data, variant, 1 << 20)[0].tolist() == [[28, 10, 28], [1, 9, 28], [8, 0, 0]]


def test_tiny_and_edges():
    for n in (1, 7, 8, 9, 11, 12, 13, 63, 64, 65, 4095, 4096, 4097):
        for variant in (1, 3):
            same(text(n, n), variant, bs=4096)
    got, off = m4.bargain_find_blocks(b"", 1)
    assert got.shape == (0, 3) and off.tolist() == [0]
    with pytest.raises(m4.M4GError):
        m4.bargain_find_blocks(b"x" * 100, 4)
    with pytest.raises(m4.M4GError):
        m4.bargain_find_blocks(b"x" * 100, 1, BlockSize=1000)


def test_stream_position_zero_is_a_candidate():
    """Position 0's table slot holds {0, its bytes} and untouched slots hold {0, 0} (zfast.go:8-11): a later
    position with position 0's hash matches it; one with another hash matches offset 0 only if its bytes are zero."""
    head = b"0123456789"
    same(head + text(5000, 12) + head + text(3000, 13), 1, bs=4096)
    same(bytes(8) + text(5000, 12) + bytes(12) + text(3000, 13), 1, bs=4096)
    same(b"\0\0\0\0abcd" + text(5000, 12) + bytes(12) + text(3000, 13), 2, bs=4096)
    same(head + b"ab" + text(5000, 12) + head + b"ab" + text(3000, 13), 3, bs=4096)
    same(bytes(12) + text(5000, 12) + bytes(16) + text(3000, 13), 3, bs=4096)


def test_batch_of_streams_device():
    import torch
    rng = np.random.default_rng(1)
    chunks = [text(int(rng.integers(1, 200000)), 20 + k) for k in range(9)] + [b""] + [text(65536 * 2, 40)]
    bs = 1 << 16
    off = np.zeros(len(chunks) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(c) for c in chunks])
    total = int(off[-1])
    plan = m4.BargainPlan(len(chunks), total, 1 << 20, variant=1, BlockSize=bs)
    nb = plan.block_count(off)
    d_src = torch.frombuffer(bytearray(b"".join(chunks)), dtype=torch.uint8).cuda()
    d_out = torch.empty((total // 3 + 2 * nb + 8) * 3, dtype=torch.int32, device="cuda")
    d_off = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
    for rep in range(2):  # the plan's workspace is reused
        nm = plan.run(d_src, off, d_out, d_off)
        got = d_out.cpu().numpy().view(np.uint32).reshape(-1, 3)[:nm]
        goff = d_off.cpu().numpy().astype(np.uint64)
        want, woff, base = [], [0], 0
        for c in chunks:
            m, o = OracleBargain(1, 1 << 20, False).find_blocks(c, bs)
            want.append(m)
            woff.extend((o[1:] + base).tolist())
            base += len(m)
        assert goff.tolist() == woff
        assert np.array_equal(got, np.concatenate(want))
    assert len(plan.last_timings()) == 4


def test_reference_criterion_on_gpu_matches():
    """Writer{Bargain1, &Encoder{}, 1 << 16} with the GPU's Match lists decodes to the input."""
    data = open(OPTICKS, "rb").read()
    ms, off = m4.bargain_find_blocks(data, 1, 1 << 20, False, 1 << 16)
    enc = OracleEncoder()
    out = bytearray()
    for b in range(len(off) - 1):
        blk = data[b << 16:(b + 1) << 16]
        out += enc.Encode(blk, ms[int(off[b]):int(off[b + 1])], last_block=False)
    out += enc.Encode(b"", np.zeros((0, 3), np.uint32), last_block=True)
    assert brotli_decompress(bytes(out), len(data)) == data


def oracle_writer_v2(data: bytes, level: int) -> bytes:
    """brotli.NewWriterV2(dst, level) restated: Write(data); Close() (writer.go:126-161, matchfinder.go:59-93)."""
    kw = m4.writer_v2_match_finder(level)
    z = OracleBargain(kw["variant"], kw["MaxDistance"], kw["Skip"])
    enc, out, bs = OracleEncoder(), bytearray(), kw["BlockSize"]
    full = len(data) // bs * bs
    for p in range(0, full, bs):
        out += enc.Encode(data[p:p + bs], z.FindMatches(data[p:p + bs]), last_block=False)
    rest = data[full:]
    out += enc.Encode(rest, z.FindMatches(rest) if rest else np.zeros((0, 3), np.uint32), last_block=True)
    return bytes(out)


@pytest.mark.parametrize("level", [5, 6, 7, 8, 9])
def test_new_writer_v2_levels_on_gpu(level):
    """NewWriterV2(dst, 5..9) end to end on the GPU: same bytes as the restated Writer, and they decode."""
    data = open(OPTICKS, "rb").read()
    for blob in (data, data[:3 << 16], data[:70000], b""):
        got = m4.brotli_compress_v2(blob, level)
        assert got == oracle_writer_v2(blob, level)
        assert brotli_decompress(got, len(blob)) == blob
    with pytest.raises(m4.M4GError):
        m4.brotli_compress_v2(data[:1000], 4)   # ZFast .. Trio are not built
    if level == 9:
        assert m4.brotli_compress_v2(data[:70000], 11) == m4.brotli_compress_v2(data[:70000], 9)  # writer.go:129-131


class Sink:
    def __init__(self):
        self.parts = []

    def write(self, b):
        self.parts.append(bytes(b))

    def value(self):
        return b"".join(self.parts)


@pytest.mark.parametrize("level", [5, 7, 9])
def test_new_writer_v2_streaming_handle(level):
    """m4g_writer_create_v2: Write in ragged pieces, Close, Reset -- the bytes the restated Writer hands to Dest,
    whatever the sizes of the Write calls (matchfinder.go:59-103)."""
    data = open(OPTICKS, "rb").read()
    big = (data * 4)[:2_500_000]   # longer than MaxDistance + a few blocks: the handle's history gets cut
    want = oracle_writer_v2(data, level)
    dst = Sink()
    w = m4.NewWriterV2(dst, level)
    w.Write(data)
    w.Close()
    assert dst.value() == want
    # ragged Writes, including empty ones and pieces that complete no, one and many blocks
    dst2 = Sink()
    w.Reset(dst2)
    rng = np.random.default_rng(level)
    p = 0
    for n in [0, 1, 65535, 1, 65536, 200000, 3, 0, 131072 + 5]:
        w.Write(data[p:p + n])
        p += n
    while p < len(data):
        n = int(rng.integers(1, 150000))
        w.Write(data[p:p + n])
        p += n
    w.Close()
    assert dst2.value() == want
    # a stream whose history is cut between calls; block-sized Writes; an exact multiple of the block size
    for blob, step in ((big, 1 << 16), (big, 700001), (big[:32 << 16], 5 << 16), (b"", 1)):
        dst3 = Sink()
        w.Reset(dst3)
        for q in range(0, len(blob), step):
            w.Write(blob[q:q + step])
        w.Close()
        assert dst3.value() == oracle_writer_v2(blob, level)
        assert brotli_decompress(dst3.value(), len(blob)) == blob


# ---- Bargain as a matchfinder.MatchFinder (m4g_bargain_ctx_*): FindMatches / Reset block after block ----------------
class _Sink:
    def __init__(self):
        self.b = bytearray()

    def write(self, p):
        self.b += p


@pytest.mark.parametrize("level", [5, 7, 9])
def test_bargain_is_a_match_finder_behind_the_unchanged_writer(level):
    """matchfinder.Writer{MatchFinder: Bargain on the GPU, Encoder: the restated brotli.Encoder, BlockSize 1 << 16}
    -- the Writer restated in m4.Writer calls FindMatches once per block, as matchfinder.go:81-87 does -- gives the
    bytes of the restated NewWriterV2; then Reset and a second stream."""
    kw = m4.writer_v2_match_finder(level)
    data = open(OPTICKS, "rb").read()
    mf = m4.Bargain(kw["variant"], kw["MaxDistance"], kw["Skip"], kw["BlockSize"])
    for blob in (data, data[:1 << 16], data[:70001]):
        dst = _Sink()
        w = m4.Writer(dst, mf, OracleEncoder(), kw["BlockSize"])
        w.Write(blob[:100000])
        w.Write(blob[100000:])
        w.Close()
        assert bytes(dst.b) == oracle_writer_v2(blob, level), len(blob)
        w.Reset(dst)  # MatchFinder.Reset + Encoder.Reset


@pytest.mark.parametrize("variant,md,skip", [(1, 1 << 17, False), (2, 1 << 17, True), (3, 1 << 18, False), (1, 0, False)])
def test_bargain_handle_over_a_stream_that_slides(variant, md, skip):
    """Block-by-block calls and many-block calls on one handle, over a stream several times longer than
    2 * MaxDistance (bargain1.go:66-80 slides again and again): Match for Match the sequential finder."""
    data = text((3 << 20) + 1000, 21) if md else text(600000, 21)  # MaxDistance 0 = the default 1 << 16
    bs = 1 << 16
    want, woff = oracle_blocks(data, variant, md or (1 << 16), skip, bs)
    mf = m4.Bargain(variant, md, skip, bs)
    got, p = [], 0
    rng = np.random.default_rng(variant)
    while p < len(data):
        k = int(rng.choice([1, 1, 2, 5, 9])) * bs  # FindMatches, or FindMatchesBlocks over k blocks
        m, off = mf.find_matches_blocks(data[p:p + k])
        got.append(m)
        p += k
    got = np.concatenate(got)
    assert got.shape == want.shape and (got == want).all()
    with pytest.raises(m4.M4GError):  # the stream's short last block has been seen: only Reset goes on
        mf.find_matches_blocks(data[:bs])
    mf.Reset()
    m, off = mf.find_matches_blocks(data[:3 * bs])
    assert (m == want[:int(woff[3])]).all() and (off == woff[:4]).all()


@pytest.mark.parametrize("level", [5, 8, 9])
def test_new_writer_v2_over_three_handles_on_one_card(level):
    """One stream over several devices for the Bargain finders: the device list (0, 0, 0)."""
    kw = m4.writer_v2_match_finder(level)
    data = text((5 << 20) + 4321, 22)
    want, woff = oracle_blocks(data, kw["variant"], kw["MaxDistance"], kw["Skip"], kw["BlockSize"])
    for devices in ((0, 0, 0), (0,), (0, 0, 0, 0, 0, 0, 0)):
        got, goff = m4.bargain_find_blocks_multi(data, devices, **kw)
        assert np.array_equal(goff, woff) and (got == want).all(), devices
    blob = open(OPTICKS, "rb").read()
    assert m4.brotli_compress_v2(blob, level, devices=(0, 0, 0)) == oracle_writer_v2(blob, level)
