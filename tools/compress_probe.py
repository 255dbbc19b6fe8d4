"""Whole-stream compression from host memory to host memory: the reference tests' Writer{M4, brotli.Encoder} and
flate.NewWriter / NewGZIPWriter (levels 5..7) with match finding and encoding on the GPU, against the restated CPU
writers on a sample.  Usage: python tools/compress_probe.py [MiB]"""
import sys, os, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from brotli_amd import m4, synth
from oracle_lib import OracleM4, brotli_decompress, flate_compress as oracle_flate

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = mib << 20
host = np.empty(n, dtype=np.uint8); synth.fill(host, 1 << 20, 0, "text", 16)
G = dict(MaxDistance=1 << 20, DistanceBitCost=66)
m4.brotli_compress(host[:1 << 22], **G); m4.flate_compress(host[:1 << 22], 6)  # warm
for name, fn, check in (
        ("brotli Writer{M4 G, Encoder, 64 KiB}", lambda: m4.brotli_compress(host, **G), lambda o: brotli_decompress(o, n) == host.tobytes()),
        ("flate.NewWriter level 5", lambda: m4.flate_compress(host, 5), lambda o: zlib.decompress(o, -15) == host.tobytes()),
        ("flate.NewGZIPWriter level 6", lambda: m4.flate_compress(host, 6, gzip=True), lambda o: zlib.decompress(o, 31) == host.tobytes()),
        ("flate.NewWriter level 7", lambda: m4.flate_compress(host, 7), lambda o: zlib.decompress(o, -15) == host.tobytes())):
    t0 = time.perf_counter(); out = fn(); dt = time.perf_counter() - t0
    print(f"{name:40s} {mib} MiB in {dt*1e3:8.1f} ms = {n/dt/1e9:6.2f} GB/s  ratio {n/len(out):.3f}  round trip {check(out)}", flush=True)
s = 8 << 20
t0 = time.perf_counter(); OracleM4(**G).compress(host[:s], 1 << 16); dt = time.perf_counter() - t0
print(f"CPU restated brotli writer, one thread: {s/dt/1e6:.1f} MB/s")
t0 = time.perf_counter(); oracle_flate(OracleM4(**m4.flate_m4_params(6)), host[:s]); dt = time.perf_counter() - t0
print(f"CPU restated flate level 6 writer, one thread: {s/dt/1e6:.1f} MB/s")
