"""m4g_writer (matchfinder.Writer with both plug-ins on the GPU): one stream, host memory in, host memory out.
Usage: python tools/writer_probe.py [MiB] [write MiB]"""
import sys, os, time, io, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from brotli_amd import m4, synth
from oracle_lib import brotli_decompress

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
wmib = int(sys.argv[2]) if len(sys.argv) > 2 else mib
n = mib << 20
host = np.empty(n, dtype=np.uint8); synth.fill(host, 1 << 20, 0, "text", 16)


class Sink:
    def __init__(self): self.parts = []
    def write(self, b): self.parts.append(b)


G = dict(MaxDistance=1 << 20, DistanceBitCost=66)
for name, fmt, kw, dec in (("brotli Writer{M4 G}", m4.FORMAT_BROTLI, G, lambda o: brotli_decompress(o, n)),
                           ("flate level 5", m4.FORMAT_FLATE, m4.flate_m4_params(5), lambda o: zlib.decompress(o, -15)),
                           ("gzip level 6", m4.FORMAT_GZIP, m4.flate_m4_params(6), lambda o: zlib.decompress(o, 31)),
                           ("flate level 7", m4.FORMAT_FLATE, m4.flate_m4_params(7), lambda o: zlib.decompress(o, -15))):
    w = m4.GPUWriter(Sink(), fmt, 1 << 16, **kw)
    w.Write(host[:8 << 20]); w.Close()  # warm: buffers
    for rep in range(2):
        sink = Sink(); w.Reset(sink)
        t0 = time.perf_counter()
        for p in range(0, n, wmib << 20):
            w.Write(host[p:p + (wmib << 20)])
        w.Close()
        dt = time.perf_counter() - t0
    out = b"".join(sink.parts)
    print(f"{name:22s} {mib} MiB in Writes of {wmib} MiB: {dt*1e3:8.1f} ms = {n/dt/1e9:6.2f} GB/s  ratio {n/len(out):.3f}  "
          f"round trip {dec(out) == host.tobytes()}", flush=True)
