"""brotli.NewWriterV2(dst, level) + Write + Close on the GPU (m4g_brotli_compress_v2), host memory to host memory:
one stream of the bench's synthetic text per level, decoded again with libbrotlidec, and the restated Writer
(restated Bargain finder + restated Encoder, one host core) on the first MiB beside it.
usage: python tools/v2_probe.py [MiB]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from brotli_amd import m4, synth  # noqa: E402
from oracle_lib import OracleBargain, OracleEncoder, brotli_decompress  # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = mib << 20
host = np.empty(n, dtype=np.uint8)
synth.fill(host, 1 << 20, 0, "text", 16)
data = host.tobytes()
m4.brotli_compress_v2(data[:1 << 20], 5)  # first touch of the device
for level in (5, 7, 8, 9):
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        out = m4.brotli_compress_v2(host, level)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    ok = brotli_decompress(out, n) == data
    kw = m4.writer_v2_match_finder(level)
    z, enc = OracleBargain(kw["variant"], kw["MaxDistance"], kw["Skip"]), OracleEncoder()
    t0 = time.perf_counter()
    ref = bytearray()
    for p in range(0, 1 << 20, 1 << 16):
        blk = data[p:p + (1 << 16)]
        ref += enc.Encode(blk, z.FindMatches(blk), last_block=False)
    cpu = (1 << 20) / (time.perf_counter() - t0)
    print(f"NewWriterV2 level {level} ({'Bargain%d' % kw['variant']}{' Skip' if kw['Skip'] else ''}): {mib} MiB in {best * 1e3:.0f} ms = "
          f"{n / best / 1e9:.2f} GB/s, ratio {n / len(out):.3f}, decodes: {ok}, first MiB equals the restated Writer: "
          f"{out[:len(ref) - 8] == bytes(ref[:len(ref) - 8])}; restated Writer on one core {cpu / 1e6:.1f} MB/s", flush=True)
