"""GPU parity of the per-match code output (m4g_batch_run_device_codes): the
command symbol and the distCache entry that pass 1 of brotli.Encoder.Encode
computes for every match (encoder.go:57-97, 163-177), bit-exact against the
oracle's restatement of that pass."""
import numpy as np
import pytest

from brotli_amd import m4, synth
from cases import PARAMS_D, PARAMS_G, edge_inputs
from oracle_lib import OracleM4, histograms

pytestmark = pytest.mark.gpu

CODE_DTYPE = np.dtype([("command", "<u2"), ("dist_code", "u1"), ("n_extra", "u1"),
                       ("extra_bits", "<u4")])


def _run(blob: np.ndarray, off: np.ndarray, kw, codes=True):
    import torch
    dev = torch.device("cuda:0")
    nchunks = len(off) - 1
    sizes = np.diff(off.astype(np.int64))
    max_n = max(int(sizes.max()) if nchunks else 1, 1)
    d_src = torch.from_numpy(blob).to(dev) if blob.size else torch.empty(1, dtype=torch.uint8, device=dev)
    plan = m4.BatchPlan(max(nchunks, 1), max_n, 0, **kw)
    min_len = kw.get("MinLength", 0) or 4
    cap = int((sizes // min_len + 2).sum()) if nchunks else 1
    d_out = torch.empty(3 * cap, dtype=torch.int32, device=dev)
    d_off = torch.empty(nchunks + 1, dtype=torch.int64, device=dev)
    d_h = torch.empty(max(nchunks, 1) * 1024, dtype=torch.int32, device=dev)
    # poisoned, so records the kernel does not write show up
    d_c = torch.full((2 * cap,), -1, dtype=torch.int32, device=dev) if codes else None
    plan.run(d_src, off, d_out, d_off, d_h, d_codes=d_c)
    plan.status()
    o = d_off.cpu().numpy()
    out = d_out.cpu().numpy().view(np.uint32).reshape(-1, 3)
    h = d_h.cpu().numpy().view(np.uint32).reshape(-1, 1024)
    c = d_c.cpu().numpy().view(CODE_DTYPE) if codes else None
    return out, o, h, c


def _check_chunk(chunk, matches, codes, name):
    lit, cmd, dist, dc, cm = histograms(chunk, matches)
    assert len(codes) == len(matches), name
    assert (codes["command"] == cm).all(), f"{name}: command symbols"
    assert (codes["dist_code"] == dc[:, 0]).all(), f"{name}: distance codes"
    assert (codes["n_extra"] == dc[:, 1]).all(), f"{name}: nExtra"
    assert (codes["extra_bits"] == dc[:, 2]).all(), f"{name}: extraBits"
    return lit, cmd, dist


@pytest.mark.parametrize("kind", ["text", "log"])
@pytest.mark.parametrize("kw", [PARAMS_G, PARAMS_D], ids=["G", "D"])
def test_codes_device_vs_oracle(kind, kw):
    n, nchunks = 1 << 18, 6
    host = np.empty(n * nchunks, dtype=np.uint8)
    synth.fill(host, n, 300, kind, 8)
    off = np.arange(nchunks + 1, dtype=np.uint64) * n
    out, o, h, codes = _run(host, off, kw)
    out0, o0, h0, _ = _run(host, off, kw, codes=False)
    assert (o == o0).all() and (out[:o[-1]] == out0[:o[-1]]).all() and (h == h0).all(), \
        "the code output must not change matches or histograms"
    for c in range(nchunks):
        chunk = host[c * n:(c + 1) * n]
        ref = OracleM4(**kw).FindMatches(chunk)
        got = out[o[c]:o[c + 1]]
        assert got.shape == np.asarray(ref).reshape(-1, 3).shape and (got == ref).all(), f"chunk {c}"
        lit, cmd, dist = _check_chunk(chunk, ref, codes[o[c]:o[c + 1]], f"chunk {c}")
        assert (h[c, :256] == lit).all() and (h[c, 256:960] == cmd).all() and (h[c, 960:] == dist).all()
    # nothing written past the last match
    tail = codes[o[-1]:].view(np.uint32)
    assert (tail == 0xFFFFFFFF).all()


@pytest.mark.parametrize("kw", [PARAMS_G, PARAMS_D], ids=["G", "D"])
def test_codes_edge_inputs(kw):
    """Ragged batch of the edge inputs: empty chunks, a lone literal run
    (Length 0: dummy copy code 2, no distance code), rings shorter than four
    distances, repeated distances (command < 128: no distance code)."""
    inputs = edge_inputs()
    names = sorted(inputs)
    sizes = [len(inputs[k]) for k in names]
    blob = np.frombuffer(b"".join(inputs[k] for k in names), dtype=np.uint8).copy()
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    out, o, h, codes = _run(blob, off, kw)
    for c, name in enumerate(names):
        chunk = inputs[name]
        ref = np.asarray(OracleM4(**kw).FindMatches(chunk), dtype=np.uint32).reshape(-1, 3)
        got = out[o[c]:o[c + 1]]
        assert got.shape == ref.shape and (got == ref).all(), name
        _check_chunk(chunk, ref, codes[o[c]:o[c + 1]], name)


def test_codes_need_histograms():
    import torch
    dev = torch.device("cuda:0")
    plan = m4.BatchPlan(1, 4096, 0, **PARAMS_G)
    d_src = torch.zeros(4096, dtype=torch.uint8, device=dev)
    d_out = torch.empty(3 * 1026, dtype=torch.int32, device=dev)
    d_off = torch.empty(2, dtype=torch.int64, device=dev)
    d_c = torch.empty(2 * 1026, dtype=torch.int32, device=dev)
    with pytest.raises(m4.M4GError):
        plan.run(d_src, np.array([0, 4096], dtype=np.uint64), d_out, d_off, None, d_codes=d_c)


def test_device_api_with_an_output_that_is_too_small():
    """m4g_batch_run_device contract (include/m4g.h): when the total exceeds
    out_cap nothing is written past out_cap, d_out_off still holds the true
    offsets, chunks whose lists fit are complete (matches, histograms, codes)
    and a chunk whose list is cut reports an all-zero histogram -- also when
    its literals are counted by the side kernel (a chunk of random bytes)."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    n = 1 << 16
    chunks = [synth.text(n, 21), bytes(rng.integers(0, 256, n, dtype=np.uint8)), synth.text(n, 22),
              bytes(rng.integers(0, 256, n, dtype=np.uint8)),
              # cut below: its long literal run is listed for the side kernel, which must skip it
              bytes(rng.integers(0, 256, 20000, dtype=np.uint8)) + synth.text(n - 20000, 23)]
    host = np.frombuffer(b"".join(chunks), dtype=np.uint8).copy()
    off = np.arange(len(chunks) + 1, dtype=np.uint64) * n
    refs = [np.asarray(OracleM4(**PARAMS_G).FindMatches(c), dtype=np.uint32).reshape(-1, 3) for c in chunks]
    true_off = np.concatenate([[0], np.cumsum([len(r) for r in refs])])
    # cut inside chunk 4: chunks 0..3 fit, chunk 4 does not
    cap = int(true_off[4]) + 10
    assert cap < true_off[5] and refs[4][0, 0] >= 8192
    d_src = torch.from_numpy(host).to(dev)
    plan = m4.BatchPlan(len(chunks), n, 0, **PARAMS_G)
    d_out = torch.full((3 * cap + 300,), -1, dtype=torch.int32, device=dev)
    d_off = torch.empty(len(chunks) + 1, dtype=torch.int64, device=dev)
    d_h = torch.full((len(chunks) * 1024,), -1, dtype=torch.int32, device=dev)
    d_c = torch.full((2 * cap + 200,), -1, dtype=torch.int32, device=dev)
    # the tensors are larger than `cap` records on purpose: the guard words behind must stay untouched
    rc = m4.lib().m4g_batch_run_device_codes(plan._h, d_src.data_ptr(), off.ctypes.data, len(chunks),
                                             d_out.data_ptr(), d_off.data_ptr(), cap, d_h.data_ptr(),
                                             d_c.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert (d_off.cpu().numpy() == true_off).all()
    out = d_out.cpu().numpy().view(np.uint32)
    assert (out[3 * cap:] == 0xFFFFFFFF).all(), "records written past out_cap"
    codes = d_c.cpu().numpy()
    assert (codes[2 * cap:] == -1).all(), "codes written past out_cap"
    h = d_h.cpu().numpy().view(np.uint32).reshape(-1, 1024)
    out = out[:3 * cap].reshape(-1, 3)
    cv = codes[:2 * cap].view(CODE_DTYPE)
    for c in range(4):
        a, b = int(true_off[c]), int(true_off[c + 1])
        assert (out[a:b] == refs[c]).all(), f"chunk {c}"
        lit, cmd, dist = _check_chunk(chunks[c], refs[c], cv[a:b], f"chunk {c}")
        assert (h[c, :256] == lit).all() and (h[c, 256:960] == cmd).all() and (h[c, 960:] == dist).all()
    assert (out[int(true_off[4]):cap] == refs[4][:10]).all()
    assert (h[4] == 0).all(), "a truncated chunk reports no histogram"
