"""m4g_find_matches_blocks: all blocks of one stream in one launch must equal
block-after-block FindMatches calls on one M4 (what matchfinder.Writer does,
matchfinder.go:59-93), including slides, the uninserted block tails and the
state the handle is left in."""
import numpy as np
import pytest

from brotli_amd import m4, synth
from oracle_lib import OracleM4

pytestmark = pytest.mark.gpu


def same(got, ref):
    return got.shape == ref.shape and (got == ref).all()


def oracle_blocks(o, data, block):
    lists = []
    for p in range(0, len(data), block):
        lists.append(o.FindMatches(data[p:p + block]))
    return lists


CASES = [
    # (params, stream bytes, block size, bytes fed block-by-block before the call)
    (dict(MaxDistance=1 << 16, DistanceBitCost=66), 1 << 20, 1 << 14, 0),          # many slides
    (dict(MaxDistance=1 << 16, DistanceBitCost=66), 700001, 20000, 150000),        # prior history, ragged end
    (dict(MaxDistance=1 << 18, DistanceBitCost=66), 3 << 20, 1 << 16, 0),          # Writer default block
    (dict(MaxDistance=1 << 20, DistanceBitCost=66), 6 << 20, 1 << 16, 70000),      # params G, slide at 2 MiB
    (dict(MaxDistance=1 << 16, ChainLength=8, HashLen=5, DistanceBitCost=66), 900000, 1 << 15, 40000),
    (dict(MaxDistance=50000, MinLength=5, ChainLength=2, TableBits=14), 400000, 5000, 123456),
    (dict(MaxDistance=1 << 18, DistanceBitCost=66), 100000, 1 << 20, 0),           # one block: per-block path
    (dict(MaxDistance=1 << 18, DistanceBitCost=66), 50000, 1000, 0),               # small blocks: per-block path
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("kind", ["text", "log"])
def test_blocks_equal_sequential_calls(case, kind):
    kw, n, block, pre = CASES[case]
    data = synth.text(pre + n + 50000, 77 + case) if kind == "text" else synth.log(pre + n + 50000, 77 + case)
    g, o = m4.M4(**kw), OracleM4(**kw)
    # history from earlier, ordinary calls
    for p in range(0, pre, 30000):
        blk = data[p:min(pre, p + 30000)]
        assert same(g.find_matches(blk), o.FindMatches(blk))
    got, off = g.find_matches_blocks(data[pre:pre + n], block)
    ref = oracle_blocks(o, data[pre:pre + n], block)
    assert len(off) == len(ref) + 1
    for b, r in enumerate(ref):
        assert same(got[int(off[b]):int(off[b + 1])], r), f"block {b} of {len(ref)}"
    assert g.history_len() == o.history_len()
    # the handle goes on exactly like the reference's
    tail = data[pre + n:]
    assert same(g.find_matches(tail[:20000]), o.FindMatches(tail[:20000]))
    got2, off2 = g.find_matches_blocks(tail[20000:], 10000)
    for b, r in enumerate(oracle_blocks(o, tail[20000:], 10000)):
        assert same(got2[int(off2[b]):int(off2[b + 1])], r), f"second call, block {b}"


def test_blocks_more_slides_than_one_launch_holds():
    """A small MaxDistance slides the history at (almost) every block -- flate's 32 KiB window does; the slide list
    of one launch holds 4096 entries, so a long stream is cut into spans of whole blocks inside the call.  Here:
    MaxDistance 4096, 18 MiB in 8 KiB blocks = 4608 slides."""
    kw = dict(MaxDistance=4096, HashLen=5, ChainLength=4, DistanceBitCost=66)
    n = 18 << 20
    data = synth.text(n + 30000, 123)
    g, o = m4.M4(**kw), OracleM4(**kw)
    got, off = g.find_matches_blocks(data[:n], 8192)
    ref = oracle_blocks(o, data[:n], 8192)
    assert len(off) == len(ref) + 1
    for b, r in enumerate(ref):
        assert same(got[int(off[b]):int(off[b + 1])], r), f"block {b} of {len(ref)}"
    assert g.history_len() == o.history_len()
    assert same(g.find_matches(data[n:]), o.FindMatches(data[n:]))


def test_blocks_degenerate_inputs():
    kw = dict(MaxDistance=1 << 16, DistanceBitCost=66)
    for data in (bytes(300000), b"ab" * 150000, (b"0123456789" * 50 + b"xyz") * 700):
        g, o = m4.M4(**kw), OracleM4(**kw)
        got, off = g.find_matches_blocks(data, 8192)
        for b, r in enumerate(oracle_blocks(o, data, 8192)):
            assert same(got[int(off[b]):int(off[b + 1])], r), f"block {b}"


def test_writer_blocks_per_launch_same_stream():
    """matchfinder.Writer over the GPU handle with many blocks per launch writes
    the byte-identical stream the block-by-block Writer writes, and it decodes."""
    import io
    from oracle_lib import OracleEncoder, brotli_decompress
    kw = dict(MaxDistance=1 << 18, DistanceBitCost=66)
    data = synth.text(3 * (1 << 20) + 12345, 5)
    bufs = []
    for per_launch in (1, 16):
        buf = io.BytesIO()
        w = m4.Writer(buf, m4.M4(**kw), OracleEncoder(), BlockSize=1 << 16, BlocksPerLaunch=per_launch)
        for p in range(0, len(data), 300000):  # ragged Write sizes
            w.Write(data[p:p + 300000])
        w.Close()
        bufs.append(buf.getvalue())
    assert bufs[0] == bufs[1]
    ref_stream, _ = OracleM4(**kw).compress(data, 1 << 16)
    assert bufs[1] == ref_stream
    assert brotli_decompress(bufs[1], len(data)) == data


def test_blocks_largest_window_slide():
    """BASELINE config 4's streaming shape: one 48 MiB stream in 1 MiB blocks with
    the largest window the V2 header allows (MaxDistance 16777200), so the slide at
    len(history) > 32 MiB fires inside the launch.  ChainLength 2 instead of 64
    keeps the CPU oracle to seconds; the chain-walk itself is covered at full
    depth by the batch tests."""
    kw = dict(MaxDistance=16777200, ChainLength=2, HashLen=5, DistanceBitCost=66)
    n, block = 48 << 20, 1 << 20
    host = np.empty(n, dtype=np.uint8)
    synth.fill(host, 1 << 20, 900, "text", 8)
    g, o = m4.M4(**kw), OracleM4(**kw)
    got, off = g.find_matches_blocks(host, block)
    for b in range(n // block):
        ref = o.FindMatches(host[b * block:(b + 1) * block])
        assert same(got[int(off[b]):int(off[b + 1])], ref), f"block {b}"
    assert g.history_len() == o.history_len()
    tail = synth.text(100000, 3)
    assert same(g.find_matches(tail), o.FindMatches(tail))


@pytest.mark.parametrize("tail", [1, 6, 7, 8, 100, 3000, 3841])
@pytest.mark.parametrize("kw", [dict(MaxDistance=1 << 16, DistanceBitCost=66),
                                dict(MaxDistance=1 << 16, ChainLength=4, HashLen=5, DistanceBitCost=66)],
                         ids=["chain0", "chain4"])
def test_blocks_short_last_block_leaves_the_sequential_tables(tail, kw):
    """ADVICE r1 (k1_xchg.h): a last block shorter than a K1 tile.  Its last 7
    positions must not be inserted (m4.go:112): the handle's table and chain
    after the call are what block-after-block calls leave (table entries further
    back than MaxDistance may be empty, include/m4g.h), and a follow-up call finds
    the same matches."""
    block = 20000
    n = 5 * block + tail
    data = synth.text(n + 30000, 500 + tail)
    g, o = m4.M4(**kw), OracleM4(**kw)
    got, off = g.find_matches_blocks(data[:n], block)
    ref = oracle_blocks(o, data[:n], block)
    for b, r in enumerate(ref):
        assert same(got[int(off[b]):int(off[b + 1])], r), f"block {b}"
    assert g.history_len() == o.history_len()
    table, chain = g.dump_tables()
    rt, rc = o.dump_table(), o.dump_chain()
    if kw.get("ChainLength", 0) > 0:  # the reference keeps a chain only then
        assert (chain == rc).all()
    else:  # never-inserted positions (block tails) have no chain entry here either
        h = g.history_len()
        start = h - n
        for b in range(5 + 1):
            e = start + min(n, (b + 1) * block)
            assert (chain[max(start, e - 7):e] == 0).all(), f"tail of block {b}"
    diff = table != rt
    W = kw["MaxDistance"]
    assert ((table[diff] == 0) & (g.history_len() - rt[diff].astype(np.int64) > W)).all()
    nxt = data[n:n + 30000]
    assert same(g.find_matches(nxt), o.FindMatches(nxt))
