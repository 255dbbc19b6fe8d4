#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: input GB/s through M4 match finding
(+ the Encoder histogram pass) on MI355X, bit-exact vs the oracle.

A "step" = one pass of the hot path (K1 chain build -> K3 parse -> K4
compaction -> K5 histograms) over one batch of synthetic text chunks that is
already resident in HBM.  Default workload = BASELINE.json configs[1]:
4 GiB of synthetic Markov text in 4096 independent 1 MiB chunks, parameters
G = M4{MaxDistance: 1<<20, DistanceBitCost: 66}.

Multi-GPU (configs[2]): one process per GPU.  `--scaling weak` (default): every
rank runs --chunks chunks (rank r owns chunk indices [r*C, (r+1)*C)).
`--scaling strong`: the SAME --chunks chunks are split over the ranks (config 3
as written: "the same 4 GiB stream sharded across 2/4/8").  There is no
data-path collective -- torch.distributed is used only for the barrier and the
max-over-ranks of the elapsed time.

`python bench.py --gpus N` without a launcher starts the N ranks itself (child
processes through torch.distributed.run, before this process touches a GPU) and
fails if the node has fewer than N devices.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PARAM_SETS = {
    # G = "M4 greedy", D = "deep-search stress"
    "G": dict(MaxDistance=1 << 20, DistanceBitCost=66),
    "D": dict(MaxDistance=16777200, ChainLength=64, HashLen=5, DistanceBitCost=66),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def shard_chunks(chunks: int, rank: int, world: int, scaling: str) -> range:
    """Global chunk indices of `rank`.  weak: `chunks` per rank, rank r owns
    [r*C, (r+1)*C).  strong: `chunks` in total, contiguous near-equal ranges."""
    if scaling == "weak":
        return range(rank * chunks, (rank + 1) * chunks)
    lo = chunks * rank // world
    hi = chunks * (rank + 1) // world
    return range(lo, hi)


def host_threads() -> int:
    """Threads the CPU legs may use: the cores this process may run on, capped by
    the box's OMP_NUM_THREADS share (a one-GPU box exports 16; os.cpu_count()
    there shows the whole machine)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    cap = os.environ.get("OMP_NUM_THREADS")
    if cap and cap.isdigit() and int(cap) > 0:
        n = min(n, int(cap))
    return max(1, n)


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_find_matches(host, chunk_bytes: int, kw: dict, chunks, threads: int):
    """The CPU oracle (C restatement of the reference's M4; Go is not installed)
    over `chunks` (local indices), one M4 per thread, Reset per chunk -- the
    shape of the reference's benchmark() (brotli_test.go:651-676).  Returns
    (seconds spent in FindMatches, {chunk: match list})."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle_lib import OracleM4
    chunks = list(chunks)

    def work(t):
        o = OracleM4(**kw)
        res = {}
        for c in chunks[t::threads]:
            o.Reset()
            res[c] = o.FindMatches(host[c * chunk_bytes:(c + 1) * chunk_bytes])
        return res

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(work, range(threads)))
    dt = time.perf_counter() - t0
    out = {}
    for p in parts:
        out.update(p)
    return dt, out


def pmc_traffic(kernel: str, args, input_bytes: int):
    """HBM-side bytes per launch of `kernel` (GB), from the rocprofv3 --pmc passes
    committed under profiles/ (FETCH_SIZE and WRITE_SIZE cannot be collected
    inside this process; tools/pmc_traffic.py turns the two passes into
    bytes per input byte for one workload, scaled here to this launch).  A
    stored measurement, not one of this run.  None when none is on file."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except OSError:
        return None
    key = f"{args.kind}/{args.params}"
    per_byte = t.get(key, {}).get(kernel)
    return None if per_byte is None else round(per_byte * input_bytes / 1e9, 3)


def flate_leg(host, n: int, chunks: int, dev, device_index: int, no_cpu: bool) -> dict:
    """flate.NewGZIPWriter(level 6) over `chunks` chunks (one deflate block per chunk, fresh M4 per chunk): match
    finding at the level's M4 parameters (flate/writer.go:33-46) and the gzip encoder on the GPU
    (m4g_batch_flate_encode).  Reported beside `value`; checked by inflating the whole stream."""
    import gzip
    import numpy as np
    import torch
    from brotli_amd import m4
    kw = m4.flate_m4_params(6)
    try:
        plan = m4.BatchPlan(chunks, n, device_index, **kw)
        plan.set_timing(True)
        off = np.arange(chunks + 1, dtype=np.uint64) * n
        d_src = torch.empty(n * chunks + 64, dtype=torch.uint8, device=dev)
        d_src[:n * chunks].copy_(torch.from_numpy(host[:n * chunks]))
        d_out = torch.empty(3 * (n // 4 + 2) * chunks, dtype=torch.int32, device=dev)
        d_off = torch.empty(chunks + 1, dtype=torch.int64, device=dev)
        plan.run(d_src, off, d_out, d_off, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.run(d_src, off, d_out, d_off, None)
        torch.cuda.synchronize()
        dt_find = time.perf_counter() - t0
        stages = plan.last_timings()
        plan.status()
        pin = m4.PinnedBuffer(n * chunks + 4096)
        plan.flate_encode(d_src, chunks, d_out, d_off, gzip=True, mtime=1, out=pin.array)  # warm: allocates the stream
        nbytes, ms, kinds = plan.flate_encode(d_src, chunks, d_out, d_off, gzip=True, mtime=1, out=pin.array)
        stream = bytes(pin.array[:nbytes])
        pin.close()
    except m4.M4GError as e:
        return {"error": str(e)}
    ok = gzip.decompress(stream) == host[:n * chunks].tobytes()
    res = {
        "level": 6, "m4_params": kw, "chunks": chunks,
        "find_matches_value": round(chunks * n / dt_find / 1e9, 3), "find_matches_stage_ms": [round(x, 3) for x in stages],
        "encode_value": round(chunks * n / (ms[2] * 1e-3) / 1e9, 3), "unit": "GB/s", "encode_ms_total": round(ms[2], 2),
        "encode_ms_host_codes": round(ms[0], 2), "encode_ms_k6f_tokens": round(ms[1], 2),
        "stream_bytes": nbytes, "compression_ratio": round(chunks * n / nbytes, 4),
        "block_kinds": {k: int((kinds == i).sum()) for i, k in enumerate(("stored", "fixed", "dynamic"))},
        "gzip_stream_inflates_to_input": ok,
        "sample": f"the first {chunks} chunks of the workload as one gzip stream, one deflate block per chunk: "
                  f"m4g_batch_run_device at flate level 6's M4 parameters, then m4g_batch_flate_encode (k5f symbol "
                  f"counts, codes on host threads, k6f tokens, CRC-32 on the GPU, D2H into page-locked memory)",
    }
    if not no_cpu:
        from oracle_lib import OracleFlateEncoder, OracleM4
        k = min(chunks, 6)
        lists = [OracleM4(**kw).FindMatches(host[c * n:(c + 1) * n]) for c in range(k)]
        enc, ref = OracleFlateEncoder(True, 1), b""
        t0 = time.perf_counter()
        for c in range(k):
            ref += enc.Encode(host[c * n:(c + 1) * n], lists[c], False)
        dt = time.perf_counter() - t0
        res["cpu_encode_s_per_GiB"] = round(dt / (k * n / 2**30), 3)
        res["first_blocks_equal_restated_encoder"] = stream[:len(ref) - 8] == ref[:len(ref) - 8]
    return res


def bargain_leg(host, n: int, chunks: int, dev, device_index: int, no_cpu: bool) -> dict:
    """matchfinder.Bargain1 (what brotli.NewWriterV2 levels 5-6 plug in, writer.go:142-143) over the
    first `chunks` chunks of the workload, every chunk one stream cut into 64 KiB blocks as matchfinder.Writer does.
    Reported beside `value`; the first chunks are compared Match for Match with the restated finder."""
    import numpy as np
    import torch
    from brotli_amd import m4
    bs = 1 << 16
    try:
        plan = m4.BargainPlan(chunks, n * chunks, n, variant=1, MaxDistance=1 << 20, BlockSize=bs, device=device_index)
        off = np.arange(chunks + 1, dtype=np.uint64) * n
        nb = plan.block_count(off)
        d_src = torch.empty(n * chunks + 64, dtype=torch.uint8, device=dev)
        d_src[:n * chunks].copy_(torch.from_numpy(host[:n * chunks]))
        d_out = torch.empty((n * chunks // 3 + 2 * nb + 8) * 3, dtype=torch.int32, device=dev)
        d_off = torch.empty(nb + 1, dtype=torch.int64, device=dev)
        plan.run(d_src, off, d_out, d_off)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nm = plan.run(d_src, off, d_out, d_off)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = plan.last_timings()
    except m4.M4GError as e:
        return {"error": str(e)}
    res = {
        "finder": "Bargain1{MaxDistance: 1 << 20}", "block_size": bs, "chunks": chunks, "blocks": nb,
        "value": round(chunks * n / dt / 1e9, 3), "unit": "GB/s", "ms": round(dt * 1e3, 2),
        "stage_ms": {"k1_links": round(st[0], 3), "byte_costs": round(st[1], 3), "search": round(st[2], 3),
                     "compact": round(st[3], 3)},
        "matches": nm, "bytes_per_match": round(chunks * n / max(nm, 1), 2),
        "sample": f"the first {chunks} chunks of the workload, each a fresh stream of {n // bs} blocks; one wavefront "
                  f"per block runs the shortest-path search (m4g_bargain_run_device, inputs and outputs in HBM)",
    }
    if not no_cpu:
        from oracle_lib import OracleBargain
        k = min(chunks, 4)
        got = d_out.cpu().numpy().view(np.uint32).reshape(-1, 3)
        base, ok = 0, True
        t0 = time.perf_counter()
        for c in range(k):
            want, _ = OracleBargain(1, 1 << 20, False).find_blocks(host[c * n:(c + 1) * n], bs)
            ok = ok and bool(np.array_equal(got[base:base + len(want)], want))
            base += len(want)
        dt = time.perf_counter() - t0
        res["cpu_restated_MB_s_one_core"] = round(k * n / dt / 1e6, 1)
        res["first_chunks_equal_restated_finder"] = ok
        res["chunks_compared"] = k
    # brotli.NewWriterV2(dst, 5) + Write + Close on ONE stream, host memory to host memory (m4g_brotli_compress_v2)
    try:
        nbytes = min(chunks * n, 256 << 20)
        m4.brotli_compress_v2(host[:1 << 20], 5)
        best, stream = None, b""
        for _ in range(2):
            t0 = time.perf_counter()
            stream = m4.brotli_compress_v2(host[:nbytes], 5)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        v2 = {"value": round(nbytes / best / 1e9, 3), "unit": "GB/s", "ms": round(best * 1e3, 1), "stream_bytes_in": nbytes,
              "compression_ratio": round(nbytes / max(len(stream), 1), 4),
              "sample": "one stream, pageable host memory in and out, plan creation and every allocation inside the call"}
        if not no_cpu:
            from oracle_lib import brotli_decompress
            v2["decodes_to_input"] = brotli_decompress(stream, nbytes) == host[:nbytes].tobytes()
        res["new_writer_v2_level5"] = v2
    except m4.M4GError as e:
        res["new_writer_v2_level5"] = {"error": str(e)}
    return res


def writer_leg(host, nbytes: int, device_index: int) -> dict:
    """matchfinder.Writer{M4, Encoder, BlockSize 1 << 16} over the first nbytes of the workload as ONE stream, host
    memory in, host memory out (m4g_writer: match finding with the history carried from block to block AND the
    encoder on the GPU): the reference tests' brotli writer (params G) and flate.NewGZIPWriter(level 6)."""
    import gzip
    from brotli_amd import m4
    from oracle_lib import brotli_decompress  # (libbrotlidec, the stand-in for brotli.NewReader)

    class Sink:
        def __init__(self):
            self.parts = []

        def write(self, b):
            self.parts.append(b)

    res = {"stream_bytes_in": nbytes, "unit": "GB/s", "block_size": 1 << 16,
           "sample": "one Write of the whole input from pageable memory + Close, second full-size run on a warm writer; wall "
                     "clock of the Python wrapper (H2D, K1-K4 per span, encoder kernels, host code building, D2H)"}
    data = host[:nbytes]
    for name, fmt, kw in (("brotli_G", m4.FORMAT_BROTLI, PARAM_SETS["G"]), ("gzip_level6", m4.FORMAT_GZIP, m4.flate_m4_params(6))):
        try:
            w = m4.GPUWriter(Sink(), fmt, 1 << 16, mtime=1, device=device_index, **kw)
            w.Write(data[:8 << 20])
            w.Close()
            for _ in range(2):  # the first full-size run grows the writer's device buffers
                sink = Sink()
                w.Reset(sink)
                t0 = time.perf_counter()
                w.Write(data)
                w.Close()
                dt = time.perf_counter() - t0
            stream = b"".join(sink.parts)
            back = brotli_decompress(stream, nbytes) if fmt == m4.FORMAT_BROTLI else gzip.decompress(stream)
            res[name] = {"value": round(nbytes / dt / 1e9, 3), "ms": round(dt * 1e3, 1),
                         "compression_ratio": round(nbytes / len(stream), 4), "decodes_to_input": back == data.tobytes()}
            del w
        except m4.M4GError as e:
            res[name] = {"error": str(e)}
    return res


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chunks", type=int, default=4096,
                    help="chunks per GPU (weak scaling) or in total (strong scaling)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--chunk-bytes", type=int, default=1 << 20)
    ap.add_argument("--params", choices=sorted(PARAM_SETS), default="G")
    ap.add_argument("--kind", choices=["text", "log"], default="text")
    ap.add_argument("--no-histos", action="store_true")
    ap.add_argument("--codes", action="store_true",
                    help="also write the per-match command / distCache records (m4g_batch_run_device_codes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify-chunks", type=int, default=-1,
                    help="chunks per rank compared with the oracle (-1 = every chunk)")
    ap.add_argument("--cpu-1thread-chunks", type=int, default=384)
    ap.add_argument("--encode-sample-chunks", type=int, default=48)
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="time the steps in the library's one-stream form (each kernel alone on the GPU: what a kernel "
                         "trace should see); by default a large batch runs as two halves on two streams")
    ap.add_argument("--no-gpu-encode", action="store_true",
                    help="skip the m4g_batch_encode leg (Encoder pass 2 on the GPU, SURVEY 8f.2)")
    ap.add_argument("--flate-chunks", type=int, default=256,
                    help="chunks of the flate leg (M4 at flate level 6 + m4g_batch_flate_encode, SURVEY 8f.4); 0 = skip")
    ap.add_argument("--bargain-chunks", type=int, default=512,
                    help="chunks of the Bargain1 leg (SURVEY 8f.3: the shortest-path match finder of NewWriterV2 5-6); 0 = skip")
    ap.add_argument("--writer-mib", type=int, default=256,
                    help="MiB of the workload compressed as ONE stream by m4g_writer (matchfinder.Writer on the GPU); 0 = skip")
    ap.add_argument("--e2e-sample-chunks", type=int, default=512)
    ap.add_argument("--e2e-persistent-chunks", type=int, default=2048)
    return ap.parse_args()


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # No launcher: start the ranks ourselves, before this process has made
        # any GPU call (device_count() does not initialise the GPU on this image).
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but this node has {have} GPU(s)", file=sys.stderr)
            raise SystemExit(2)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    import numpy as np
    import torch
    from brotli_amd import m4, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: no GPU {local_rank} on this node ({torch.cuda.device_count()} visible)")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    kw = PARAM_SETS[args.params]
    n = args.chunk_bytes
    mine = shard_chunks(args.chunks, rank, world, args.scaling)
    C = len(mine)
    total_chunks = args.chunks * world if args.scaling == "weak" else args.chunks
    if C == 0:
        raise SystemExit("fewer chunks than ranks")

    # ---- synthetic input, generated on the host, then resident in HBM ------
    nthreads = max(1, host_threads() // max(1, world))
    host = np.empty(n * C, dtype=np.uint8)
    synth.fill(host, n, mine.start, args.kind, nthreads)
    d_src = torch.empty(n * C + 64, dtype=torch.uint8, device=dev)
    piece = 256 << 20
    for o in range(0, host.size, piece):
        d_src[o:o + min(piece, host.size - o)].copy_(torch.from_numpy(host[o:o + piece]))
    off = np.arange(C + 1, dtype=np.uint64) * n

    plan = m4.BatchPlan(C, n, local_rank, **kw)
    # Stage timing (HIP events between the kernels) belongs to the one-stream form of the pipeline; with it off the
    # library runs a batch of a GiB and more as two halves on two streams (m4g.hip run_batch_overlapped).
    plan.set_timing(args.no_overlap)
    # measured match density is ~N/12; size the output for N/6 (checked below)
    out_cap = (n // 6 + 2) * C
    d_out = torch.empty(3 * out_cap, dtype=torch.int32, device=dev)
    d_off = torch.empty(C + 1, dtype=torch.int64, device=dev)
    d_h = None if args.no_histos else torch.empty(C * 1024, dtype=torch.int32, device=dev)
    d_codes = torch.empty(2 * out_cap, dtype=torch.int32, device=dev) if args.codes and d_h is not None else None
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        plan.run(d_src, off, d_out, d_off, d_h, stream, d_codes=d_codes)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # one settled run first: an input that overflows the plan's compact record slots switches it to the
    # worst-case layout here (m4g_batch_info), not inside the timed region
    step()
    plan.status()
    for _ in range(args.warmup):
        step()
    stage_ms = np.zeros(4)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    plan.status()  # kernel-side errors of the device-resident API (records dropped, repeat range)
    # per-kernel durations (HIP events recorded by the library on the launch
    # stream): re-run untimed steps so the timed region is not touched
    reps = max(1, min(args.steps, 3))
    # what the timed (overlapped) steps left behind, to compare with the one-stream form's output below
    nm_timed = int(d_off[C].item())
    keep = (d_off.clone(), d_out[:3 * min(nm_timed, out_cap)].clone(), d_h.clone() if d_h is not None else None)
    plan.set_timing(True)
    step()  # (the first one-stream run after overlapped ones: untimed)
    barrier()
    same_out = bool(torch.equal(keep[0], d_off) and torch.equal(keep[1], d_out[:keep[1].numel()]) and
                    (d_h is None or torch.equal(keep[2], d_h)))
    del keep
    if not same_out:
        raise SystemExit("the overlapped and the one-stream form of the pipeline disagree")
    t1 = time.perf_counter()
    for _ in range(reps):
        step()
        stage_ms += np.array(plan.last_timings())
    one_stream_ms = (time.perf_counter() - t1) / reps * 1e3
    stage_ms /= reps
    plan.status()

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- Encoder pass 2 on the GPU (untimed for `value`; reported beside the CPU Encode time) ----
    gpu_encode = None
    if rank == 0 and d_h is not None and not args.no_gpu_encode:
        dc = d_codes if d_codes is not None else torch.empty(2 * out_cap, dtype=torch.int32, device=dev)
        plan.run(d_src, off, d_out, d_off, d_h, stream, d_codes=dc)
        plan.status()
        pin = m4.PinnedBuffer(n * C // 2 + (C + 1) * 1024 + 64)
        try:
            plan.encode(d_src, C, d_out, d_off, d_h, dc, out=pin.array)  # warm: allocates the stream buffer
            nbytes, ems = plan.encode(d_src, C, d_out, d_off, d_h, dc, out=pin.array)
            gpu_encode = {"stream_bytes": nbytes, "ms": ems, "first": bytes(pin.array[:1 << 20])}
        except m4.M4GError as e:  # (incompressible input does not fit the half-size buffer)
            gpu_encode = {"error": str(e)}
        pin.close()
        if d_codes is None:
            del dc

    total_matches = int(d_off[C].item())
    if total_matches > out_cap:
        raise SystemExit(f"output capacity too small: {total_matches} > {out_cap}")

    # ---- CPU leg + parity gate (untimed for the GPU) ------------------------
    # The oracle runs over the chunks anyway for the CPU baseline; its Match
    # lists are kept and compared, with the histograms, against the GPU output of
    # EVERY one of them ("bit-exact Match lists vs oracle on every chunk").
    from oracle_lib import OracleEncoder, brotli_decompress, histograms
    nverify = C if args.verify_chunks < 0 else min(C, args.verify_chunks)
    if args.no_cpu_baseline and args.verify_chunks < 0:
        nverify = min(C, 8)
    vchunks = [int(c) for c in np.unique(np.linspace(0, C - 1, num=nverify, dtype=int))] if nverify else []
    threads = nthreads
    cpu_dt, ref_lists = cpu_find_matches(host, n, kw, vchunks, threads)
    o = d_off.cpu().numpy()
    got_all = d_out[:3 * total_matches].cpu().numpy().view(np.uint32).reshape(-1, 3)
    hist_all = None if d_h is None else d_h.cpu().numpy().view(np.uint32).reshape(C, 1024)

    def check(c):
        chunk = host[c * n:(c + 1) * n]
        ref = ref_lists[c]
        got = got_all[int(o[c]):int(o[c + 1])]
        if got.shape != ref.shape or not (got == ref).all():
            return f"PARITY FAILURE on local chunk {c} (rank {rank})"
        if hist_all is not None:
            lit, cmd, dst, _, _ = histograms(chunk, ref)
            h = hist_all[c]
            if not ((h[:256] == lit).all() and (h[256:960] == cmd).all() and (h[960:] == dst).all()):
                return f"HISTOGRAM PARITY FAILURE on local chunk {c} (rank {rank})"
        return None

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(threads) as ex:
        for msg in ex.map(check, vchunks):
            if msg:
                raise SystemExit(msg)
    verified = len(vchunks)

    # the reference's own criterion (brotli_test.go:626-649) on a sample: the
    # stream the restated Encoder builds from the GPU's matches decodes back to
    # the input; also times the CPU Encode a real pipeline would run after us
    stream_bytes = roundtrips = 0
    enc_s = 0.0
    enc_chunks = [c for c in vchunks[::max(1, len(vchunks) // max(1, args.encode_sample_chunks))]][:args.encode_sample_chunks]
    for c in enc_chunks:
        chunk = host[c * n:(c + 1) * n]
        got = got_all[int(o[c]):int(o[c + 1])]
        enc = OracleEncoder()
        t1 = time.perf_counter()
        bs = enc.Encode(chunk, got, True)
        enc_s += time.perf_counter() - t1
        stream_bytes += len(bs)
        try:
            if brotli_decompress(bs, n) != chunk.tobytes():
                raise SystemExit(f"ROUND TRIP FAILURE on local chunk {c} (rank {rank})")
            roundtrips += 1
        except RuntimeError:
            pass  # no libbrotlidec on this box: ratio only

    if dist is not None:
        v = torch.tensor([verified, roundtrips], dtype=torch.int64, device=dev)
        dist.all_reduce(v)
        verified_all, roundtrips_all = int(v[0]), int(v[1])
    else:
        verified_all, roundtrips_all = verified, roundtrips

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    total_bytes = n * total_chunks
    value = total_bytes / (elapsed / args.steps) / 1e9

    # roofline of the dominant kernel: algorithmic bytes of one launch =
    # N (each input byte once) + 12*M (each Match record once) [+ 4096 B/chunk]
    names = ["k1_chain_build", "k3_parse", "k4_compact", "k5_histograms"]
    dom = int(np.argmax(stage_ms))
    alg_bytes = n * C + 12 * total_matches + (0 if d_h is None else 4096 * C)
    achieved = alg_bytes / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
    out = {
        "metric": "input GB/s through M4 match finding + Encoder histogram pass (bit-exact Matches)",
        "value": round(value, 3),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": f"{total_chunks} x {n} B independent chunks of synthetic Markov {args.kind}"
                        f" ({total_chunks * n / 2**30:.2f} GiB over {world} GPU(s), {C} chunks on rank 0),"
                        f" fresh M4 state per chunk",
            "params": args.params,
            "m4": kw,
            "histograms": d_h is not None,
            "per_match_codes": d_codes is not None,
            "matches_per_MiB": round(total_matches / (C * n / 2**20), 1),
            "parity_checked_chunks": verified_all,
            "parity_checked_of": total_chunks,
            "roundtrip_checked_chunks": roundtrips_all,
            "compression_ratio": round(len(enc_chunks) * n / stream_bytes, 3) if stream_bytes else None,
            "workspace": dict(plan.info(), bytes_per_input_byte=round(plan.info()["bytes"] / (C * n), 3)),
        },
        "stage_ms": {names[k]: round(float(stage_ms[k]), 3) for k in range(4)},
        "overlap": {"timed_steps_overlapped": bool(not args.no_overlap and C * n >= 1 << 29 and C >= 512),
                    "one_stream_ms_per_step": round(one_stream_ms, 3),
                    "outputs_equal_one_stream_form": same_out,
                    "note": "stage_ms and roofline come from one-stream steps (each kernel alone on the GPU); `value` "
                            "is the library's default: two halves of the batch on two streams"},
        "roofline": {
            "bound": "hbm",
            "kernel": names[dom],
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": pmc_traffic(names[dom], args, n * C),
            "traffic_source": "profiles/traffic.json (separate rocprofv3 --pmc passes, not this run)",
            "algorithmic_bytes_per_launch": alg_bytes,
            "whole_path_frac": round(alg_bytes / (float(stage_ms.sum()) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            if stage_ms.sum() > 0 else None,
        },
    }
    # rate of the measured HBM-side traffic (GB/s): how close the kernel's REAL bytes are to the peak
    tr = out["roofline"]["traffic"]
    out["roofline"]["traffic_rate"] = round(tr / (stage_ms[dom] * 1e-3), 1) if tr and stage_ms[dom] > 0 else None
    if not args.no_cpu_baseline and world == 1 and verified:
        # one thread, on a bounded prefix of the same chunks
        s1 = min(len(vchunks), args.cpu_1thread_chunks)
        dt1, _ = cpu_find_matches(host, n, kw, vchunks[:s1], 1)
        out["cpu_baseline"] = {
            "value": round(verified * n / cpu_dt / 1e9, 4),
            "unit": "GB/s",
            "cores": threads,
            "kind": "port",
            "value_1thread": round(s1 * n / dt1 / 1e9, 4),
            "cpu": cpu_model(),
            "host_cpus_visible": os.cpu_count(),
            "sample": f"{verified} chunks of the same workload ({verified * n / 2**20:.0f} MiB), C restatement of the "
                      f"reference's M4 (oracle/m4_ref.c; Go is unavailable), {threads} threads (this process's CPU "
                      f"share: affinity capped by OMP_NUM_THREADS), one M4 per thread, Reset per chunk, FindMatches "
                      f"only; value_1thread: the first {s1} of them on one thread",
        }
        if enc_chunks and enc_s > 0:
            out["cpu_encode"] = {
                "s_per_GiB": round(enc_s / (len(enc_chunks) * n / 2**30), 3),
                "value": round(len(enc_chunks) * n / enc_s / 1e9, 4),
                "unit": "GB/s",
                "cores": 1,
                "sample": f"restated brotli.Encoder.Encode (oracle/encoder_ref.c) on {len(enc_chunks)} chunks, one "
                          f"thread: the serial stage a V2 pipeline runs after match finding (SURVEY section 7)",
            }
    if gpu_encode and "ms" in gpu_encode:
        # byte identity of the stream's first meta-block with the restated Encoder (the tests check whole streams)
        enc0 = OracleEncoder()
        ref0 = enc0.Encode(host[:n], ref_lists[0] if 0 in ref_lists else got_all[int(o[0]):int(o[1])], False)
        k = max(0, len(ref0) - 8)
        host_ms, k6_ms, all_ms = gpu_encode["ms"]
        out["gpu_encode"] = {
            "value": round(C * n / (all_ms * 1e-3) / 1e9, 3), "unit": "GB/s", "ms_total": round(all_ms, 2),
            "ms_host_prefix_codes": round(host_ms, 2), "ms_k6_emit_bits": round(k6_ms, 2),
            "stream_bytes": gpu_encode["stream_bytes"], "compression_ratio": round(C * n / gpu_encode["stream_bytes"], 4),
            "first_metablock_equals_restated_encoder": gpu_encode["first"][:k] == ref0[:k],
            "sample": f"m4g_batch_encode over all {C} chunks of rank 0 (one brotli stream, one meta-block per chunk): "
                      f"D2H of the histograms, prefix codes on host threads, K6 on the GPU, D2H of the stream into "
                      f"page-locked memory; ms_total is the whole call",
        }
    elif gpu_encode:
        out["gpu_encode"] = gpu_encode
    if not args.no_end_to_end and world == 1:
        # PCIe-inclusive rate of the host-memory API on a bounded sample: wall
        # clock from the first H2D enqueue to the last D2H completion (SURVEY
        # 8d "end-to-end").  Reported beside `value`, never as `value`.
        sample = min(C, args.e2e_sample_chunks)
        del plan, d_out, d_off, d_h, d_src, d_codes
        torch.cuda.empty_cache()
        sub_off = off[:sample + 1]
        m4.find_matches_batch(host[:n * min(sample, 8)], off[:min(sample, 8) + 1], (local_rank,), True, **kw)  # warm
        t0 = time.perf_counter()
        m4.find_matches_batch(host[:n * sample], sub_off, (local_rank,), True, **kw)
        dt = time.perf_counter() - t0
        # the same call on page-locked buffers from m4g_host_alloc (input copied in beforehand, untimed)
        pin_in = m4.PinnedBuffer(n * sample)
        pin_in.array[:] = host[:n * sample]
        pin_out = m4.PinnedBuffer(12 * (total_matches * sample // C + (1 << 20)))
        pin_out.array[:] = 0
        t0 = time.perf_counter()
        m4.find_matches_batch(pin_in.array, sub_off, (local_rank,), True, out=pin_out.array.view(np.uint32), **kw)
        dt_pin = time.perf_counter() - t0
        pin_in.close()
        pin_out.close()
        # the persistent handle (m4g_hostbatch_*): nothing allocated per run; a larger sample, as a serving
        # loop would see it
        ps = min(C, args.e2e_persistent_chunks)
        hb = m4.HostBatch(ps, n, (local_rank,), **kw)
        pin_in = m4.PinnedBuffer(n * ps)
        pin_in.array[:] = host[:n * ps]
        pin_out = m4.PinnedBuffer(12 * (total_matches * ps // C + (1 << 20)))
        pin_out.array[:] = 0
        hb.run(pin_in.array, off[:ps + 1], True, out=pin_out.array.view(np.uint32))  # warm
        t0 = time.perf_counter()
        pm, poff, ph = hb.run(pin_in.array, off[:ps + 1], True, out=pin_out.array.view(np.uint32))
        dt_hb = time.perf_counter() - t0
        same_as_device = bool((poff.astype(np.int64) == o[:ps + 1]).all()) and \
            bool((pm == got_all[:int(o[ps])]).all())
        # the same chunks as ONE brotli stream (m4g_hostbatch_compress): the stream comes home, not the Match lists
        compress = None
        if args.params == "G" and not args.no_gpu_encode:
            try:
                hb.compress(pin_in.array, off[:ps + 1], m4.FORMAT_BROTLI, out=pin_out.array)  # warm: encoder buffers
                t0 = time.perf_counter()
                nb = hb.compress(pin_in.array, off[:ps + 1], m4.FORMAT_BROTLI, out=pin_out.array)
                dt_c = time.perf_counter() - t0
                compress = {"value": round(ps * n / dt_c / 1e9, 3), "ms": round(dt_c * 1e3, 1), "stream_bytes": nb,
                            "compression_ratio": round(ps * n / nb, 4)}
                if gpu_encode and "first" in gpu_encode:  # same bytes as the device-resident encoder's first MiB
                    k = min(1 << 20, nb) - 8
                    compress["equals_m4g_batch_encode_prefix"] = bytes(pin_out.array[:k]) == gpu_encode["first"][:k]
            except m4.M4GError as e:
                compress = {"error": str(e)}
        hb.close()
        out["end_to_end"] = {
            "value": round(sample * n / dt / 1e9, 3), "unit": "GB/s",
            "value_pinned": round(sample * n / dt_pin / 1e9, 3),
            "value_pinned_persistent": round(ps * n / dt_hb / 1e9, 3),
            "persistent_sample_chunks": ps,
            "persistent_equals_device_path": same_as_device,
            "compress_brotli_pinned_persistent": compress,
            "sample": f"m4g_find_matches_batch on the first {sample} chunks: workspace allocation + H2D + kernels + "
                      f"D2H of matches and histograms; `value` from pageable numpy memory (fresh output array), "
                      f"`value_pinned` with input and output in m4g_host_alloc buffers; `value_pinned_persistent`: "
                      f"m4g_hostbatch_run on {ps} chunks from page-locked buffers, handle created beforehand "
                      f"(H2D + kernels + D2H of matches and histograms, sliced and overlapped); "
                      f"`compress_brotli_pinned_persistent`: m4g_hostbatch_compress on the same chunks and handle "
                      f"(H2D + kernels + encoder + D2H of the brotli stream)",
        }
        pin_in.close()
        pin_out.close()
    if rank == 0 and world == 1 and args.flate_chunks > 0 and args.kind == "text" and not args.no_gpu_encode:
        try:
            out["gpu_flate"] = flate_leg(host, n, min(C, args.flate_chunks), dev, local_rank, args.no_cpu_baseline)
        except Exception as e:  # a side leg must never cost the bench line
            out["gpu_flate"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and args.writer_mib > 0 and args.kind == "text" and not args.no_gpu_encode:
        try:
            out["gpu_writer"] = writer_leg(host, min(n * C, args.writer_mib << 20), local_rank)
        except Exception as e:
            out["gpu_writer"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and args.bargain_chunks > 0 and args.kind == "text" and not args.no_gpu_encode:
        try:
            out["gpu_bargain"] = bargain_leg(host, n, min(C, args.bargain_chunks), dev, local_rank, args.no_cpu_baseline)
        except Exception as e:  # a side leg must never cost the bench line
            out["gpu_bargain"] = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
