"""Block-by-block m4g_find_matches calls only (for a kernel trace of the streaming handle)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from brotli_amd import m4, synth
block = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
m = 8 << 20
host = np.empty(m, dtype=np.uint8); synth.fill(host, 1 << 20, 0, "text", 16)
kw = dict(MaxDistance=1 << 20, DistanceBitCost=66)
g = m4.M4(**kw)
for p in range(0, 1 << 20, block):
    g.find_matches(host[p:p + block])  # warm
g.Reset()
t = time.perf_counter()
for p in range(0, m, block):
    g.find_matches(host[p:p + block])
dt = time.perf_counter() - t
print(f"sequential calls: {m >> 20} MiB, block {block}: {dt*1e3:.1f} ms  {m/dt/1e9:.3f} GB/s  {dt/(m//block)*1e6:.0f} us per call", flush=True)
