"""Experiment: does the batch pipeline gain from running as TWO halves on two streams (K1 of one half next to K3 of
the other), compared with one plan over everything?  Prints ms per pass over C chunks of 1 MiB."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from brotli_amd import m4, synth
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = 1 << 20
host = np.empty(n * C, dtype=np.uint8); synth.fill(host, n, 0, "text", 16)
dev = torch.device("cuda:0")
d_src = torch.empty(n * C + 64, dtype=torch.uint8, device=dev)
for o in range(0, host.size, 256 << 20):
    d_src[o:o + min(256 << 20, host.size - o)].copy_(torch.from_numpy(host[o:o + (256 << 20)]))
kw = dict(MaxDistance=1 << 20, DistanceBitCost=66)

def setup(nparts):
    P = []
    per = C // nparts
    for k in range(nparts):
        plan = m4.BatchPlan(per, n, 0, **kw)
        P.append(dict(plan=plan, src=d_src[k * per * n:], off=np.arange(per + 1, dtype=np.uint64) * n,
                      out=torch.empty(3 * (n // 6 + 2) * per, dtype=torch.int32, device=dev),
                      ooff=torch.empty(per + 1, dtype=torch.int64, device=dev),
                      h=torch.empty(per * 1024, dtype=torch.int32, device=dev), st=torch.cuda.Stream(dev)))
    return P

def run(P, steps):
    for p in P:
        p["plan"].run(p["src"], p["off"], p["out"], p["ooff"], p["h"], p["st"].cuda_stream)
        p["plan"].status()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for p in P:
            p["plan"].run(p["src"], p["off"], p["out"], p["ooff"], p["h"], p["st"].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for nparts in (1, parts, 1, parts):
    P = setup(nparts)
    ms = run(P, 8)
    print(f"{C} chunks as {nparts} plan(s) on {nparts} stream(s): {ms:.2f} ms per pass = {C * n / ms / 1e6:.1f} GB/s", flush=True)
    del P
    torch.cuda.empty_cache()
