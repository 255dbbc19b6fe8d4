"""End-to-end rate of the persistent host batch (m4g_hostbatch_*) from page-locked buffers.
   M4G_PIPE_SLICE=<chunks> python tools/hostbatch_probe.py [chunks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from brotli_amd import m4, synth
C = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = 1 << 20
kw = dict(MaxDistance=1 << 20, DistanceBitCost=66)
pin_in = m4.PinnedBuffer(n * C)
synth.fill(pin_in.array, n, 0, "text", 16)
pin_out = m4.PinnedBuffer(12 * (100000 * C))
off = np.arange(C + 1, dtype=np.uint64) * n
hb = m4.HostBatch(C, n, (0,), **kw)
for r in range(4):
    t0 = time.perf_counter()
    hb.run(pin_in.array, off, True, out=pin_out.array.view(np.uint32))
    dt = time.perf_counter() - t0
    print(f"slice {os.environ.get('M4G_PIPE_SLICE', 'auto')}: run {r}: {dt*1e3:.1f} ms  {C*n/dt/1e9:.2f} GB/s", flush=True)
hb.close()

# the same chunks compressed on the handle: the stream comes home instead of the Match lists
for name, fmt, kw2 in (("brotli (params G)", m4.FORMAT_BROTLI, kw), ("gzip (flate level 6 fields)", m4.FORMAT_GZIP, m4.flate_m4_params(6)),
                       ("deflate (flate level 5 fields)", m4.FORMAT_FLATE, m4.flate_m4_params(5))):
    hb = m4.HostBatch(C, n, (0,), **kw2)
    for r in range(3):
        t0 = time.perf_counter()
        k = hb.compress(pin_in.array, off, fmt, mtime=1, out=pin_out.array)
        dt = time.perf_counter() - t0
        print(f"compress {name}: run {r}: {dt*1e3:.1f} ms  {C*n/dt/1e9:.2f} GB/s  ratio {C*n/k:.3f}", flush=True)
    hb.close()
