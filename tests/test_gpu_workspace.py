"""GPU tests of the batch plan's compact record slots (include/m4g.h, m4g_batch_info).

A plan keeps one Match record per 8 positions of a sub-block; the reference has
no such limit (FindMatches appends to a Go slice, m4.go:63-66), so an input
with more matches than that must come out the same -- through the re-run in the
worst-case layout that m4g_batch_status (or the encoders over the plan) starts.
Random bytes of 4 symbols under a chain-walking parameter set are such an
input: about 150 matches per KiB (measured with the oracle), the slots hold
136 per KiB sub-block."""
import numpy as np
import pytest

from brotli_amd import m4
from oracle_lib import OracleEncoder, OracleM4, brotli_decompress, histograms

pytestmark = pytest.mark.gpu

DENSE = dict(MaxDistance=1 << 20, MinLength=4, HashLen=5, TableBits=17, ChainLength=8)


def same(a, b):
    a = np.asarray(a, dtype=np.uint32).reshape(-1, 3)
    b = np.asarray(b, dtype=np.uint32).reshape(-1, 3)
    return a.shape == b.shape and bool((a == b).all())


def _buffers(plan_chunks, nmax, total):
    import torch
    dev = torch.device("cuda:0")
    cap = total // 4 + 2 * plan_chunks + 1
    return (torch.empty(3 * cap, dtype=torch.int32, device=dev), torch.empty(plan_chunks + 1, dtype=torch.int64, device=dev),
            torch.empty(plan_chunks * 1024, dtype=torch.int32, device=dev), torch.empty(2 * cap, dtype=torch.int32, device=dev))


def _run(plan, chunks, bufs, settle=True):
    import torch
    host = np.frombuffer(b"".join(chunks), dtype=np.uint8)
    off = np.cumsum([0] + [len(c) for c in chunks]).astype(np.uint64)
    d_src = torch.zeros(host.size + 64, dtype=torch.uint8, device="cuda:0")
    d_src[:host.size].copy_(torch.from_numpy(host.copy()))
    d_out, d_off, d_h, d_codes = bufs
    plan.run(d_src, off, d_out, d_off, d_h, d_codes=d_codes)
    if settle:
        plan.status()
    return d_src


def _check_against_oracle(chunks, bufs, kw):
    d_out, d_off, d_h, _ = bufs
    C = len(chunks)
    o = d_off.cpu().numpy()
    out = d_out.cpu().numpy().view(np.uint32).reshape(-1, 3)
    h = d_h.cpu().numpy().view(np.uint32).reshape(-1, 1024)
    assert o[0] == 0
    for c in range(C):
        ref = OracleM4(**kw).FindMatches(chunks[c])
        assert same(out[o[c]:o[c + 1]], ref), f"chunk {c}"
        lit, cmd, dist, _, _ = histograms(np.frombuffer(chunks[c], dtype=np.uint8), ref)
        assert (h[c, :256] == lit).all() and (h[c, 256:960] == cmd).all() and (h[c, 960:] == dist).all(), f"chunk {c} histograms"


def test_dense_input_overflows_the_compact_slots_and_is_rerun():
    rng = np.random.default_rng(7)
    C, n = 64, 1 << 16
    sizes = [n - (997 * c) % 5000 for c in range(C)]  # ragged
    dense = [rng.integers(0, 4, s, dtype=np.uint8).tobytes() for s in sizes]
    plan = m4.BatchPlan(C, n, 0, **DENSE)
    before = plan.info()
    assert before["slot"] < before["slot_worst"] and before["group_chunks"] == 0 and before["reruns"] == 0
    bufs = _buffers(C, n, sum(sizes))
    _run(plan, dense, bufs)
    after = plan.info()
    assert after["reruns"] == 1 and after["slot"] == after["slot_worst"]
    assert 0 < after["group_chunks"] < C, "the worst-case layout holds the batch in several groups"
    assert after["bytes"] == before["bytes"], "same memory"
    _check_against_oracle(dense, bufs, DENSE)
    # the switch is sticky: the next batches (fewer chunks; more symbols, denser still) run once, group after group
    dense8 = [rng.integers(0, 8, s, dtype=np.uint8).tobytes() for s in sizes[:41]]
    _run(plan, dense8, bufs)
    assert plan.info()["reruns"] == 1
    _check_against_oracle(dense8, bufs, DENSE)
    # long literal runs (K4 hands them to the side kernel with the chunk's index in the BATCH, not in its group)
    mixed = [rng.integers(0, 256, 40000, dtype=np.uint8).tobytes() if c % 7 == 3 else dense[c] for c in range(C)]
    _run(plan, mixed, bufs)
    _check_against_oracle(mixed, bufs, DENSE)


@pytest.mark.parametrize("kw,symbols", [(dict(MaxDistance=1 << 20, HashLen=5), 3), (dict(MaxDistance=1 << 20, HashLen=4, MinLength=3), 4),
                                        (dict(MaxDistance=1 << 16, HashLen=5, ChainLength=64, DistanceBitCost=66), 8)],
                         ids=["chain0-hash5", "chain0-minlen3", "chain64"])
def test_overflow_under_other_parameter_sets(kw, symbols):
    """ChainLength 0 takes the event-driven K3a (k3a_fast): 140-220 records per KiB with HashLen 4 / 5 on 3-4 symbols
    (tools/analysis/record_density.py's method).  Ragged chunks, some of them tiny or empty."""
    rng = np.random.default_rng(11 + symbols)
    n = 1 << 16
    sizes = [n, 0, 5, n - 1, 4097, n, 1000, n - 333] * 5
    chunks = [rng.integers(0, symbols, s, dtype=np.uint8).tobytes() for s in sizes]
    plan = m4.BatchPlan(len(sizes), n, 0, **kw)
    assert plan.info()["slot"] < plan.info()["slot_worst"]
    bufs = _buffers(len(sizes), n, sum(sizes) * 4 // (kw.get("MinLength") or 4))
    _run(plan, chunks, bufs)
    assert plan.info()["reruns"] == 1
    _check_against_oracle(chunks, bufs, kw)
    _run(plan, chunks[::-1], bufs)
    assert plan.info()["reruns"] == 1
    _check_against_oracle(chunks[::-1], bufs, kw)


def test_overflow_in_a_plan_with_2_kib_sub_blocks():
    """Plans of a GiB and more use sub-blocks of 2048 positions (264 records per slot); 1024 x 1 MiB of 3-symbol
    bytes under HashLen 5: 153 records per KiB.  A sample of chunks against the oracle, all of them through the
    tiling property (Unmatched + Length sums to the chunk length) and the histograms' literal totals."""
    import torch
    kw = dict(MaxDistance=1 << 20, HashLen=5)
    C, n = 1024, 1 << 20
    rng = np.random.default_rng(21)
    host = rng.integers(0, 3, C * n, dtype=np.uint8)
    d_src = torch.from_numpy(host).to("cuda:0")
    plan = m4.BatchPlan(C, n, 0, **kw)
    info = plan.info()
    assert info["slot"] == 264 and info["slot_worst"] == 530
    off = np.arange(C + 1, dtype=np.uint64) * n
    d_out = torch.empty(3 * (n // 4 + 2) * C, dtype=torch.int32, device="cuda:0")
    d_off = torch.empty(C + 1, dtype=torch.int64, device="cuda:0")
    d_h = torch.empty(C * 1024, dtype=torch.int32, device="cuda:0")
    plan.run(d_src, off, d_out, d_off, d_h)
    plan.status()
    info = plan.info()
    assert info["reruns"] == 1 and 0 < info["group_chunks"] < C
    o = d_off.cpu().numpy()
    out = d_out[:3 * int(o[C])].cpu().numpy().view(np.uint32).reshape(-1, 3)
    seg = np.repeat(np.arange(C), np.diff(o).astype(np.int64))
    covered = np.bincount(seg, weights=out[:, 0].astype(np.float64) + out[:, 1], minlength=C)
    assert (covered == n).all()
    h = d_h.cpu().numpy().view(np.uint32).reshape(C, 1024)
    lits = np.bincount(seg, weights=out[:, 0].astype(np.float64), minlength=C)
    assert (h[:, :256].sum(axis=1) == lits).all()
    for c in (0, 509, 510, 1019, 1020, 1023):  # around the group boundaries (510 chunks per group)
        ref = OracleM4(**kw).FindMatches(host[c * n:(c + 1) * n])
        assert same(out[o[c]:o[c + 1]], ref), c


def test_encoder_over_the_plan_settles_an_overflowed_run_itself():
    rng = np.random.default_rng(8)
    C, n = 48, 1 << 16
    chunks = [rng.integers(0, 4, n, dtype=np.uint8).tobytes() for _ in range(C)]
    plan = m4.BatchPlan(C, n, 0, **DENSE)
    bufs = _buffers(C, n, C * n)
    d_src = _run(plan, chunks, bufs, settle=False)  # no m4g_batch_status in between
    d_out, d_off, d_h, d_codes = bufs
    stream, _ = plan.encode(d_src, C, d_out, d_off, d_h, d_codes, first=True, last=True)
    assert plan.info()["reruns"] == 1
    data = b"".join(chunks)
    assert brotli_decompress(stream, len(data)) == data
    enc, ref = OracleEncoder(), b""
    for c in chunks:
        ref += enc.Encode(c, OracleM4(**DENSE).FindMatches(c), False)
    assert stream == ref + enc.Encode(b"", np.zeros((0, 3), np.uint32), True)


def test_inputs_below_the_bound_never_rerun():
    """4-symbol random bytes under M4's default parameters: 90 records per KiB."""
    rng = np.random.default_rng(9)
    C, n = 32, 1 << 16
    chunks = [rng.integers(0, 4, n, dtype=np.uint8).tobytes() for _ in range(C)]
    plan = m4.BatchPlan(C, n, 0)
    bufs = _buffers(C, n, C * n)
    _run(plan, chunks, bufs)
    assert plan.info()["reruns"] == 0
    _check_against_oracle(chunks, bufs, {})


def test_headline_plan_fits_34_gib():
    """the 4096 x 1 MiB plan in at most 34 GiB (it was 47)."""
    plan = m4.BatchPlan(4096, 1 << 20, 0)
    info = plan.info()
    assert info["bytes"] <= 34 << 30, info
    assert info["bytes"] / (4096 << 20) < 8.5
    del plan


def test_small_plans_keep_the_worst_case_layout():
    """One chunk: the allocation must hold a whole job in the worst-case layout anyway."""
    info = m4.BatchPlan(1, 1 << 16, 0).info()
    assert info["slot"] == info["slot_worst"] and info["group_chunks"] == 0


def test_two_halves_on_two_streams_equal_the_one_stream_form():
    """A batch of a GiB and more runs as two halves on two streams (m4g.hip run_batch_overlapped) unless the plan's
    stage timing is on.  Ragged chunks (so that the halves' chain offsets are not round numbers), Match lists,
    offsets and histograms of both forms equal, and equal to the oracle around the cut and at the ends."""
    import torch
    from brotli_amd import synth
    C, n = 1100, 1 << 20
    sizes = np.array([n - (7919 * c) % 30011 for c in range(C)], dtype=np.uint64)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    host = np.empty(C * n, dtype=np.uint8)
    synth.fill(host, n, 500, "text", 16)
    host = host[:int(off[C])]
    d_src = torch.from_numpy(host).to("cuda:0")
    plan = m4.BatchPlan(C, n, 0, MaxDistance=1 << 20, DistanceBitCost=66)
    cap = (n // 6 + 2) * C
    outs = []
    for timing in (False, True, False):
        d_out = torch.zeros(3 * cap, dtype=torch.int32, device="cuda:0")
        d_off = torch.zeros(C + 1, dtype=torch.int64, device="cuda:0")
        d_h = torch.zeros(C * 1024, dtype=torch.int32, device="cuda:0")
        plan.set_timing(timing)
        plan.run(d_src, off, d_out, d_off, d_h)
        plan.status()
        outs.append((d_off, d_out, d_h))
    assert plan.info()["reruns"] == 0
    for k in (1, 2):
        assert torch.equal(outs[0][0], outs[k][0]) and torch.equal(outs[0][1], outs[k][1]) and torch.equal(outs[0][2], outs[k][2])
    o = outs[0][0].cpu().numpy()
    assert int(o[C]) <= cap
    out = outs[0][1][:3 * int(o[C])].cpu().numpy().view(np.uint32).reshape(-1, 3)
    kw = dict(MaxDistance=1 << 20, DistanceBitCost=66)
    for c in (0, 1, 548, 549, 550, 551, C - 1):
        ref = OracleM4(**kw).FindMatches(host[int(off[c]):int(off[c + 1])])
        assert same(out[o[c]:o[c + 1]], ref), c
