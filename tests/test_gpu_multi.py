"""One stream over several devices (m4g_find_matches_blocks_multi, SURVEY 8(e) protocol P1) on the GPU: the device
list (0, 0, 0) -- three handles and three host threads on the one card of the test box -- must give the Match
lists of the single handle and of the sequential oracle, and the streams built from them must be the restated
Writer's bytes.  tests/test_spans.py checks the same arithmetic on the CPU."""
import zlib

import numpy as np
import pytest

from brotli_amd import m4, synth
from cases import PARAMS_G
from oracle_lib import OracleM4, brotli_decompress, flate_compress as oracle_flate_compress

pytestmark = pytest.mark.gpu


def text(n, seed):
    host = np.empty((n + (1 << 20) - 1) >> 20 << 20, dtype=np.uint8)
    synth.fill(host, 1 << 20, seed, "text", 8)
    return host[:n].tobytes()


def oracle_blocks(kw, data, bs):
    o = OracleM4(**kw)
    lists = [o.FindMatches(data[p:p + bs]) for p in range(0, len(data), bs)]
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.uint64)
    return np.concatenate([np.asarray(x, dtype=np.uint32).reshape(-1, 3) for x in lists]), off


@pytest.mark.parametrize("kw,n,bs", [
    (PARAMS_G, 7 * (1 << 20) + 12345, 1 << 16),                      # MaxDistance 1 MiB: slides from 2 MiB on
    (m4.flate_m4_params(6), 3 * (1 << 20) + 777, 1 << 16),            # 32 KiB window: a slide at every block, chain walks
    (dict(MaxDistance=1 << 18, ChainLength=8, HashLen=5, DistanceBitCost=66), (5 << 20) + 4000, 1 << 17),
    (PARAMS_G, (1 << 16) + 5, 1 << 16),                               # two blocks, three devices
], ids=["G", "flate6", "chain8-128k", "tiny"])
def test_three_handles_on_one_card_equal_one_handle_and_the_oracle(kw, n, bs):
    data = text(n, 40)
    want_m, want_off = oracle_blocks(kw, data, bs)
    one_m, one_off = m4.M4(**kw).find_matches_blocks(data, bs)
    assert (one_off == want_off).all() and (one_m == want_m).all()
    for devices in ((0, 0, 0), (0, 0), (0,), (0, 0, 0, 0, 0)):
        got_m, got_off = m4.find_matches_blocks_multi(data, bs, devices, **kw)
        assert (got_off == want_off).all(), devices
        assert (got_m == want_m).all(), devices


def test_inputs_that_are_hard_for_the_speculation_over_three_handles():
    rng = np.random.default_rng(5)
    t = text(3 << 20, 41)
    data = t[:1 << 20] + bytes(300000) + bytes(rng.integers(0, 256, 400000, dtype=np.uint8)) + t[:1 << 20] + b"ab" * 200000
    want_m, want_off = oracle_blocks(PARAMS_G, data, 1 << 16)
    got_m, got_off = m4.find_matches_blocks_multi(data, 1 << 16, (0, 0, 0), **PARAMS_G)
    assert (got_off == want_off).all() and (got_m == want_m).all()


def test_streams_from_a_shared_stream_are_the_restated_writers():
    data = text((4 << 20) + 999, 42)
    # matchfinder.Writer{M4 G, brotli.Encoder, 1 << 16}
    got = m4.brotli_compress(data, 1 << 16, devices=(0, 0, 0), **PARAMS_G)
    assert got == OracleM4(**PARAMS_G).compress(data, 1 << 16)[0]
    assert brotli_decompress(got, len(data)) == data
    # flate.NewWriter(dst, 6)
    got = m4.flate_compress(data, 6, devices=(0, 0, 0))
    assert got == oracle_flate_compress(OracleM4(**m4.flate_m4_params(6)), data, 1 << 16)
    assert zlib.decompress(got, -15) == data
