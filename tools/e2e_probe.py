"""End-to-end rate of the host-memory batch API (pageable and page-locked buffers)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from brotli_amd import m4, synth
C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = 1 << 20
host = np.empty(n * C, dtype=np.uint8); synth.fill(host, n, 0, "text", 16)
off = np.arange(C + 1, dtype=np.uint64) * n
kw = dict(MaxDistance=1 << 20, DistanceBitCost=66)
m4.find_matches_batch(host[:8 * n], off[:9], (0,), True, **kw)
for rep in range(2):
    t = time.perf_counter(); out, _, _ = m4.find_matches_batch(host, off, (0,), True, **kw); dt = time.perf_counter() - t
    print(f"e2e pageable {C} chunks: {dt*1e3:.1f} ms  {C*n/dt/1e9:.2f} GB/s", flush=True)
pin_in = m4.PinnedBuffer(n * C); pin_in.array[:] = host
pin_out = m4.PinnedBuffer(12 * (out.shape[0] + 1024)); pin_out.array[:] = 0
for rep in range(3):
    t = time.perf_counter(); m4.find_matches_batch(pin_in.array, off, (0,), True, out=pin_out.array.view(np.uint32), **kw); dt = time.perf_counter() - t
    print(f"e2e pinned   {C} chunks: {dt*1e3:.1f} ms  {C*n/dt/1e9:.2f} GB/s", flush=True)
