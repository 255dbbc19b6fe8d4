"""Throughput of m4g_bargain_run_device on the bench workload (synthetic text, 1 MiB streams in 64 KiB blocks),
with the restated finder on one host core beside it and a Match-for-Match check of the first chunks.
usage: python tools/bargain_probe.py [chunks] [variant] [reps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from brotli_amd import m4, synth  # noqa: E402

chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
kind = sys.argv[4] if len(sys.argv) > 4 else "text"
n, bs = 1 << 20, 1 << 16
host = np.empty(n * chunks, dtype=np.uint8)
if kind in ("text", "log"):
    synth.fill(host, n, 0, kind, 16)
elif kind == "random":
    host[:] = np.random.default_rng(1).integers(0, 256, size=host.size, dtype=np.uint8)
elif kind == "zeros":
    host[:] = 0
else:
    raise SystemExit("kind: text | log | random | zeros")
d_src = torch.empty(n * chunks + 64, dtype=torch.uint8, device="cuda")
d_src[:n * chunks].copy_(torch.from_numpy(host))
off = np.arange(chunks + 1, dtype=np.uint64) * n
plan = m4.BargainPlan(chunks, n * chunks, n, variant=variant, MaxDistance=1 << 20, BlockSize=bs)
nb = plan.block_count(off)
d_out = torch.empty((n * chunks // 3 + 2 * nb + 8) * 3, dtype=torch.int32, device="cuda")
d_off = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
for r in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nm = plan.run(d_src, off, d_out, d_off)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = plan.last_timings()
    print(f"Bargain{variant} {kind} {chunks} x 1 MiB, {nb} blocks: {dt * 1e3:.1f} ms = {n * chunks / dt / 1e9:.2f} GB/s; "
          f"stages ms links {st[0]:.2f} costs {st[1]:.2f} search {st[2]:.2f} compact {st[3]:.2f}; {nm} matches "
          f"({n * chunks / max(nm, 1):.1f} bytes per match)", flush=True)

from oracle_lib import OracleBargain  # noqa: E402
k = min(chunks, 2)
got = d_out.cpu().numpy().view(np.uint32).reshape(-1, 3)
goff = d_off.cpu().numpy()
t0 = time.perf_counter()
base = 0
ok = True
for c in range(k):
    want, woff = OracleBargain(variant, 1 << 20, False).find_blocks(host[c * n:(c + 1) * n], bs)
    ok = ok and np.array_equal(got[base:base + len(want)], want)
    base += len(want)
dt = time.perf_counter() - t0
print(f"restated Bargain{variant}, one core: {k * n / dt / 1e6:.1f} MB/s; first {k} chunks equal: {ok}")
