"""Throughput of one stream through the handle: block-by-block calls vs
m4g_find_matches_blocks (one launch)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from brotli_amd import m4, synth
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
block = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
n = mib << 20
host = np.empty(n, dtype=np.uint8); synth.fill(host, 1 << 20, 0, "text", 16)
kw = dict(MaxDistance=1 << 20, DistanceBitCost=66)
g = m4.M4(**kw); g.find_matches_blocks(host[:4 << 20], block)  # warm
g = m4.M4(**kw)
for rep in range(3):
    g.Reset()
    t = time.perf_counter(); out, off = g.find_matches_blocks(host, block); dt = time.perf_counter() - t
    print(f"blocks: {mib} MiB, block {block}: {dt*1e3:.1f} ms  {n/dt/1e9:.2f} GB/s  ({len(out)} matches)", flush=True)
g2 = m4.M4(**kw)
m = min(n, 8 << 20)
t = time.perf_counter()
tot = 0
for p in range(0, m, block):
    tot += len(g2.find_matches(host[p:p + block]))
dt = time.perf_counter() - t
print(f"sequential calls: {m >> 20} MiB, block {block}: {dt*1e3:.1f} ms  {m/dt/1e9:.3f} GB/s", flush=True)
assert tot == int(off[m // block]), (tot, int(off[m // block]))
