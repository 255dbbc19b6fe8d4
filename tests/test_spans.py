"""One stream over several devices (protocol P1; m4g_find_matches_blocks_multi): the arithmetic the
GPU path rests on, checked on the CPU.

The claim: a device that is handed the stream from the buffer origin in force for its first block, and that
inserts only the positions from MaxDistance below that block on (empty table before), produces for its blocks
exactly the Match lists of the sequential run.  Here "the device" is the second restatement (oracle/m4_py.py) put
into that state by hand, and the sequential run is the C restatement; the plan (which blocks, from which byte,
which origins) comes from the library itself (m4g_debug_span_plan touches no device)."""
import os
import sys

import numpy as np
import pytest

from brotli_amd import m4, synth
from m4_py import M4Py
from oracle_lib import OracleM4

KW = dict(MaxDistance=2000, ChainLength=4, HashLen=5, DistanceBitCost=66)
BS = 700


def sequential(data, bs, kw):
    o = OracleM4(**kw)
    return [[tuple(int(x) for x in r) for r in o.FindMatches(data[p:p + bs])] for p in range(0, len(data), bs)]


def span_on_a_fresh_finder(data, bs, kw, b0, b1, lo, origins):
    """Blocks [b0, b1) the way a device computes them: bytes from `lo` (<= the origin of block b0) on, insertion
    from MaxDistance below block b0 with an empty table, the last 7 positions of every block left out."""
    W = kw["MaxDistance"]
    x0 = b0 * bs
    org = int(origins[b0])
    assert lo <= org and lo % bs == 0 and (org == 0 or org <= x0 - W)
    m = M4Py(**kw)
    m.MinLength, m.TableBits, m.HashLen = 4, 17, kw.get("HashLen", 6)  # (the defaults FindMatches would fill in)
    m.history = bytearray(data[org:x0])
    m.chain = [0] * len(m.history)
    first = max(org, x0 - W)
    for k in range(first // bs, b0):  # the blocks of the stream that lie in the primed range
        s, e = max(k * bs, first), min(len(data), (k + 1) * bs)
        hs = m._hashes(bytes(data[org:e]), s - org, e - 7 - org)
        for j, h in enumerate(hs.tolist()):
            i = s - org + j
            p = m.last.get(h, 0)
            if p != 0:
                m.chain[i] = i - p
            m.last[h] = i
    return [m.FindMatches([], data[b * bs:(b + 1) * bs]) for b in range(b0, b1)]


def test_origins_follow_the_slide_schedule():
    # m4.go:82-99: the slide happens when len(history) > 2 * MaxDistance at the START of a call and keeps MaxDistance
    n, bs, W = 20000, 700, 2000
    _, org = m4.span_plan(n, bs, 1, **KW)
    origin, hlen = 0, 0
    for b in range((n + bs - 1) // bs):
        if hlen > 2 * W:
            origin += hlen - W
            hlen = W
        assert int(org[b]) == origin
        hlen += min(bs, n - b * bs)
    assert int(org[-1]) > 0


@pytest.mark.parametrize("ndev", [1, 2, 3, 5])
def test_plan_tiles_the_stream_and_hands_over_enough_history(ndev):
    for n, bs, kw in ((30000, BS, KW), (64 * 65536 + 777, 65536, dict(MaxDistance=32768, HashLen=5, ChainLength=16)),
                      (9 << 20, 1 << 16, dict(MaxDistance=1 << 20))):
        pieces, org = m4.span_plan(n, bs, ndev, **kw)
        nblocks = (n + bs - 1) // bs
        W = kw["MaxDistance"]
        nxt = 0
        for dev, b0, b1, lo in pieces.tolist():
            assert b0 == nxt and b1 > b0 and dev < ndev
            nxt = b1
            assert lo % bs == 0 and lo <= org[b0]
            assert org[b0] == 0 or org[b0] + W <= b0 * bs  # the window below the first block is there
        assert nxt == nblocks
        assert [p[0] for p in pieces.tolist()] == sorted(p[0] for p in pieces.tolist())  # devices take ascending ranges


@pytest.mark.parametrize("kw", [KW, dict(MaxDistance=1500, DistanceBitCost=66), dict(MaxDistance=3000, ChainLength=64, HashLen=5)],
                         ids=["chain4", "chain0", "chain64"])
def test_a_span_equals_the_sequential_run(kw):
    data = synth.text(26000, 31) + bytes(3000) + synth.text(9000, 32)[::-1] + synth.text(9000, 31)
    want = sequential(data, BS, kw)
    for ndev in (2, 3):
        pieces, org = m4.span_plan(len(data), BS, ndev, **kw)
        got = []
        for dev, b0, b1, lo in pieces.tolist():
            got += span_on_a_fresh_finder(data, BS, kw, b0, b1, lo, org)
        assert got == want


def _rank_main(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = synth.text(24000, 77)
    pieces, org = m4.span_plan(len(data), BS, world, **KW)
    mine = []
    for dev, b0, b1, lo in pieces.tolist():
        if dev == rank:
            mine += span_on_a_fresh_finder(data, BS, KW, b0, b1, lo, org)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)  # (test plumbing: the product path joins on the host, no collective)
    if rank == 0:
        q.put([blk for part in gathered for blk in part] == sequential(data, BS, KW))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_share_one_stream_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=300)
    for p in procs:
        p.join(60)
    assert ok
