"""CPU suite: pins the oracle (KATs, golden vectors, independent restatement,
round trip through a brotli decoder) and checks the host build of the parse
header -- including the sub-block speculation -- against it."""
import ctypes
import hashlib
import os

import numpy as np
import pytest

from brotli_amd import synth
from cases import (KATS, PARAMS_D, PARAMS_FAST_KAT, PARAMS_G, PARAMS_MISC, PARAMS_REF_TESTS, edge_inputs,
                   fast_path_kat, hand_kats, opticks)
from m4_py import M4Py
from oracle_lib import (OracleEncoder, OracleM4, brotli_decompress, histograms, lib as olib,
                        replay)
from sim_lib import sim, sim_k3, sim_spec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "m4_matches.npz")


def same(a, b):
    a = np.asarray(a, dtype=np.uint32).reshape(-1, 3)
    b = np.asarray(b, dtype=np.uint32).reshape(-1, 3)
    return a.shape == b.shape and bool((a == b).all())


def ids(kw):
    return ",".join(f"{a}={b}" for a, b in kw.items()) or "defaults"


def test_kats_surveyed_known_answers():
    """hand-traced known answers, default parameters."""
    ins = edge_inputs()
    for name, exp in KATS.items():
        assert same(OracleM4().FindMatches(ins[name]), exp), name
        assert same(M4Py().FindMatches([], ins[name]), exp), name
        # DistanceBitCost 66 gives the same result on these
        assert same(OracleM4(DistanceBitCost=66).FindMatches(ins[name]), exp), name


def _bucket_collisions(data, hash_len=6, table_bits=17):
    """distinct hash_len-byte windows that share a table bucket (m4.go:141)"""
    seen, n = {}, 0
    for i in range(len(data) - 7):
        v = int.from_bytes(data[i:i + 8], "little") & ((1 << (8 * hash_len)) - 1)
        h = ((v * 0x1E35A7BD1E35A7BD) & ((1 << 64) - 1)) >> (64 - table_bits)
        w = data[i:i + hash_len]
        n += h in seen and seen[h] != w
        seen[h] = w
    return n


def test_hand_traced_queue_kats():
    """known answers traced by hand through m4.go:187-259 (three
    pending matches: cases 213 / 221 / default with clip, the re-search of a
    trimmed match, the score gates, penalty 275 vs 0) -- see cases.hand_kats.
    Both restatements and the host build of the product's parse header must
    reproduce them; the traces assume no accidental bucket collision."""
    for name, (kw, data, exp, reaches) in hand_kats().items():
        assert _bucket_collisions(data) == 0, name
        assert same(OracleM4(**kw).FindMatches(data), exp), (name, reaches)
        assert same(M4Py(**kw).FindMatches([], data), exp), (name, reaches)
        assert same(sim(data, **kw), exp), (name, reaches)
        for sub in (16, 64):
            assert same(sim_k3(data, sub, **kw)[0], exp), (name, sub)


def test_golden_vectors():
    from golden.make_golden import PARAM_SETS, golden_inputs
    g = np.load(GOLDEN)
    ins = golden_inputs()
    for name, data in ins.items():
        assert hashlib.sha256(data).digest() == g[f"sha256/{name}"].tobytes(), \
            f"input generator drifted: {name}"
        for pname, kw in PARAM_SETS.items():
            assert same(OracleM4(**kw).FindMatches(data), g[f"m/{pname}/{name}"]), (pname, name)
    data = synth.text(5 * 16384, 9)
    o = OracleM4(MaxDistance=12000, ChainLength=2, DistanceBitCost=66)
    for b in range(5):
        assert same(o.FindMatches(data[b * 16384:(b + 1) * 16384]), g[f"p1/{b}"]), b


@pytest.mark.parametrize("kw", [PARAMS_G, PARAMS_D] + PARAMS_MISC, ids=ids)
def test_two_restatements_agree(kw):
    """line-faithful C restatement vs the differently structured Python one"""
    ins = dict(edge_inputs())
    ins["text_120k"] = synth.text(120000, 21)
    for name, data in ins.items():
        a = OracleM4(**kw).FindMatches(data)
        b = M4Py(**kw).FindMatches([], data)
        assert same(a, b), name
        assert replay(a, data) == data, name


def test_two_restatements_agree_streaming_with_slide():
    kw = dict(MaxDistance=5000, ChainLength=3, DistanceBitCost=66)
    data = synth.text(60000, 22)
    o, p = OracleM4(**kw), M4Py(**kw)
    hist = b""
    for pos in range(0, len(data), 4096):
        blk = data[pos:pos + 4096]
        a = o.FindMatches(blk)
        assert same(a, p.FindMatches([], blk)), pos
        assert replay(a, blk, hist) == blk
        hist += blk
    assert same(o.FindMatches(b""), p.FindMatches([], b""))


def test_streaming_slide_without_chain():
    """history slide (m4.go:82-99) with ChainLength 0: only the table is rebased"""
    kw = dict(MaxDistance=3000, DistanceBitCost=66)
    data = synth.text(30000, 28)
    o, p = OracleM4(**kw), M4Py(**kw)
    hist = b""
    for pos in range(0, len(data), 2048):
        blk = data[pos:pos + 2048]
        a = o.FindMatches(blk)
        assert same(a, p.FindMatches([], blk)), pos
        assert replay(a, blk, hist) == blk
        hist += blk
    assert o.history_len() < len(data)  # it did slide


def test_seeded_repeat_distance():
    """non-empty dst: the repeat probe uses dst[len-1].Distance (m4.go:129-130)"""
    b1 = b"abcdefghijklmnop" * 4 + b"XY"
    b2 = b"0123" + b"abcdefghijklmnop" * 2 + b"tail-of-the-block"
    o, p = OracleM4(), M4Py()
    m1 = o.FindMatches(b1)
    p.FindMatches([], b1)
    # a caller that keeps appending to the same dst: seed = last Distance so far
    seed = int(m1[m1[:, 1] > 0][-1, 2])
    a = o.FindMatches(b2, seed_prev_distance=seed)
    b = p.FindMatches([(0, 4, seed)], b2)[1:]
    assert same(a, b)
    assert replay(a, b2, b1) == b2
    with pytest.raises(IndexError):  # Go would panic: src[i+1-prevDistance:] out of range
        OracleM4().FindMatches(b1, seed_prev_distance=1000)


@pytest.mark.parametrize("kw", [PARAMS_G, PARAMS_D] + PARAMS_MISC + PARAMS_REF_TESTS, ids=ids)
def test_parse_header_host_build(kw):
    """brotli_amd/csrc/m4_parse.h + m4_spec.h compiled for the host: serial
    parse and speculative sub-block parse are both bit-exact vs the oracle."""
    ins = dict(edge_inputs())
    ins["text_150k"] = synth.text(150000, 23)
    ins["log_100k"] = synth.log(100000, 24)
    for name, data in ins.items():
        ref = OracleM4(**kw).FindMatches(data)
        assert same(sim(data, **kw), ref), name
        for b in (64, 1000, 4096):
            got, stats = sim_spec(data, b, **kw)
            assert same(got, ref), (name, b)


def test_speculation_synchronises_quickly_on_text():
    data = synth.text(1 << 18, 25)
    got, stats = sim_spec(data, 1024, **PARAMS_G)
    blocks, synced, walked = int(stats[0]), int(stats[1]), int(stats[2])
    assert synced == blocks - 1
    assert walked / (blocks - 1) < 64


def _hard_for_speculation():
    """Inputs on which sub-blocks do not synchronise through a common flush: the
    validation pass (k3c) has to use its skip / fast path / serial re-stitch."""
    rng = np.random.default_rng(5)
    n = 40000
    text = np.frombuffer(synth.text(n, 9), dtype=np.uint8)
    rnd = rng.integers(0, 256, n, dtype=np.uint8)
    para = [synth.text(int(rng.integers(200, 3000)), 50 + k) for k in range(6)]
    return {
        "random": rnd.tobytes(),
        "zeros": bytes(n),
        "period7": (b"abcdefg" * (n // 7 + 1))[:n],
        "text_random": np.concatenate([text[:n // 2], rnd[:n // 2]]).tobytes(),
        "random_text_zeros": np.concatenate([rnd[:n // 3], text[:n // 3], np.zeros(n // 3, np.uint8)]).tobytes(),
        "far_repeats_in_noise": np.concatenate([rnd[:9000], rnd[:9000], rnd[1000:20000]]).tobytes(),
        "paragraphs_repeated": b"".join(para[int(k)] for k in rng.integers(0, 6, 40)),
        "lowent": rng.integers(0, 3, n, dtype=np.uint8).tobytes(),
    }


@pytest.mark.parametrize("kw", [PARAMS_G, dict(MaxDistance=1 << 16, ChainLength=4, HashLen=5, DistanceBitCost=66),
                                dict(HashLen=4, ChainLength=1)], ids=ids)
def test_k3_stage_model_vs_oracle(kw):
    """The K3 stage as the device runs it (k3a give-up rule, k3b stitch + chase,
    k3c scan / skip / fast path / re-stitch, k4 layout), modelled on the host from
    the same headers (tests/sim/parse_sim.cpp), is bit-exact on inputs that
    defeat the speculation, for several sub-block sizes and give-up distances."""
    ins = dict(_hard_for_speculation())
    ins["text_60k"] = synth.text(60000, 31)
    paths = np.zeros(5, dtype=np.int64)
    for name, data in ins.items():
        ref = OracleM4(**kw).FindMatches(data)
        for sub, ext in ((64, 64), (128, 4096), (1024, 4096), (1024, 256)):
            got, stats = sim_k3(data, sub, ext_limit=ext, **kw)
            assert same(got, ref), (name, sub, ext)
            paths += stats
    # the cases really reach every path of the validation pass
    assert paths[1] > 0 and paths[2] > 0 and paths[3] > 0 and paths[4] > 0, paths


@pytest.mark.parametrize("P", [64, 128, 1024])
def test_fast_path_kat(P):
    """ADVICE r1 (kernels.h k3c): the repeat probe must be checked up to and
    including the position where the speculative parse FOUND its first match,
    not only below the end of its first record.  cases.fast_path_kat builds the
    input; the model run with the old bound returns a different list."""
    for seed in (1, 2, 3):
        data = fast_path_kat(P, seed)
        ref = OracleM4(**PARAMS_FAST_KAT).FindMatches(data)
        assert same(M4Py(**PARAMS_FAST_KAT).FindMatches([], data), ref)
        got, stats = sim_k3(data, P, **PARAMS_FAST_KAT)
        assert same(got, ref)
        old, stats_old = sim_k3(data, P, legacy_limit=True, **PARAMS_FAST_KAT)
        assert not same(old, ref), "the KAT no longer reaches the case it was built for"
        assert stats_old[1] == stats[1] + 1  # the old bound took the fast path once more: on block 7


def test_static_code_length_header_matches_reference_constant():
    n = ctypes.c_uint()
    olib().encref_static_code_length_header.restype = ctypes.c_uint64
    v = olib().encref_static_code_length_header(ctypes.byref(n))
    # entropy_encode_static.go:785-788: writeBits(32, 0x55555554); writeBits(8, 0xFF)
    assert n.value == 40 and v == (0xFF << 32 | 0x55555554)


def test_code_helpers():
    L = olib()
    # RFC 7932 section 5: insert length code boundaries / copy length code boundaries
    for ins, code in [(0, 0), (5, 5), (6, 6), (7, 6), (8, 7), (9, 7), (10, 8), (129, 15), (130, 16),
                      (2113, 20), (2114, 21), (6209, 21), (6210, 22), (22593, 22), (22594, 23)]:
        assert L.encref_insert_length_code(ins) == code, ins
    for cp, code in [(2, 0), (9, 7), (10, 8), (11, 8), (12, 9), (133, 17), (134, 18), (2117, 22),
                     (2118, 23)]:
        assert L.encref_copy_length_code(cp) == code, cp
    assert L.encref_combine_length_codes(0, 0, 1) == 0
    assert L.encref_combine_length_codes(0, 0, 0) == 128
    assert L.encref_combine_length_codes(7, 15, 1) == (7 | 7 << 3 | 64)


def test_histograms_totals():
    data = synth.text(100000, 26)
    m = OracleM4(**PARAMS_G).FindMatches(data)
    lit, cmd, dist, dc, cm = histograms(data, m)
    assert lit.sum() == m[:, 0].sum()
    assert cmd.sum() == len(m)
    assert dist.sum() == int(((cm >= 128) & (m[:, 1] != 0)).sum())


@pytest.mark.parametrize("kw", PARAMS_REF_TESTS, ids=ids)
def test_roundtrip_reference_criterion(kw):
    """brotli_test.go:626-649 test(): Writer{M4, Encoder, 1<<16} -> decoder -> equal."""
    data = opticks()  # the reference's own test file (tests/golden/)
    assert len(data) == 567198
    stream, _ = OracleM4(**kw).compress(data, 1 << 16)
    assert brotli_decompress(stream, len(data)) == data


@pytest.mark.parametrize("name", ["empty", "short3", "kat_run40", "zeros_70k", "random_20k",
                                  "period_mix", "log_65536"])
def test_roundtrip_edge_inputs(name):
    data = edge_inputs()[name]
    for bs in (0, 1 << 12, 1 << 16):
        stream, _ = OracleM4(**PARAMS_G).compress(data, bs)
        assert brotli_decompress(stream, len(data)) == data, (name, bs)


def test_encoder_reset_determinism():
    """brotli_test.go:135-149: compress, Reset, compress again -> identical."""
    data = synth.text(200000, 27)
    mf, enc = OracleM4(**PARAMS_G), OracleEncoder()

    def run():
        out = b""
        for pos in range(0, len(data), 1 << 16):
            blk = data[pos:pos + (1 << 16)]
            out += enc.Encode(blk, mf.FindMatches(blk), False)
        return out + enc.Encode(b"", np.zeros((0, 3), np.uint32), True)

    a = run()
    mf.Reset()
    enc.Reset()
    assert run() == a
    assert brotli_decompress(a, len(data)) == data


@pytest.mark.skipif(bool(os.environ.get("M4G_ORACLE_SO")), reason="already inside the coverage run")
def test_oracle_branch_coverage():
    """tools/oracle_coverage.py: every reachable branch of the oracle's parse is
    taken by this file's inputs (report: profiles/oracle_coverage.txt)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "oracle_coverage.py"), "--check"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-500:]


def _literal_only(data: bytes):
    return np.array([[len(data), 0, 0]], dtype=np.uint32)


@pytest.mark.parametrize("counts", [(9, 1), (1, 9), (5, 3, 1), (1, 3, 5), (8, 4, 2, 1), (1, 2, 4, 8), (3, 3, 3, 3),
                                    (1, 1, 1, 6), (6, 1, 1, 1), (2, 9, 1, 3)],
                         ids=lambda c: "-".join(map(str, c)))
def test_simple_prefix_codes_of_two_to_four_symbols_decode(counts):
    """buildAndStoreHuffmanTreeFast (brotli_bit_stream.go:637-741) writes alphabets of at most 4 used symbols as a
    'simple' prefix code: the symbols sorted by depth and, for four, the tree-select bit.  Literal-only blocks
    over 2..4 distinct bytes with every order of frequencies; the system decoder must give the bytes back."""
    rng = np.random.default_rng(sum(counts))
    syms = b"qcxk"[:len(counts)]
    data = bytearray()
    for s, c in zip(syms, counts):
        data += bytes([s]) * (c * 13)
    data = bytes(rng.permutation(np.frombuffer(bytes(data), dtype=np.uint8)))
    out = OracleEncoder().Encode(data, _literal_only(data), True)
    assert brotli_decompress(out, len(data)) == data


def test_meta_block_longer_than_a_mebibyte_decodes():
    """storeMetaBlockHeaderBW (compress_fragment_two_pass.go:222-241): MLEN above 1 << 20 takes six nibbles."""
    data = synth.text((1 << 20) + 70000, 3)
    m = OracleM4(MaxDistance=1 << 20, DistanceBitCost=66)
    out = OracleEncoder().Encode(data, m.FindMatches(data), True)
    assert brotli_decompress(out, len(data)) == data
