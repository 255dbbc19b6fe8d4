"""Batch throughput on inputs that are hard for the speculative parse."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from brotli_amd import m4, synth
C, n = int(os.environ.get("KP_CHUNKS", "256")), 1 << 20
rng = np.random.default_rng(1)
kinds = {
    "random": rng.integers(0, 256, C * n, dtype=np.uint8),
    "zeros": np.zeros(C * n, dtype=np.uint8),
    "alphabet4": rng.integers(0, 4, C * n, dtype=np.uint8),
    "period7": np.resize(np.frombuffer(b"abcdefg", dtype=np.uint8), C * n),
}
t = np.empty(C * n, dtype=np.uint8); synth.fill(t, n, 0, "text", 16)
mix = t.copy(); mix.reshape(C, n)[:, n // 2:] = kinds["random"].reshape(C, n)[:, n // 2:]
kinds["half text half random"] = mix
# long matches everywhere: each chunk is a sequence of paragraphs drawn from a pool of 16
for plen in (2000, 8192, 40000):
    pool = [t[i * plen:(i + 1) * plen] for i in range(16)]
    rep = np.concatenate([pool[int(j)] for j in rng.integers(0, 16, C * n // plen + 1)])[:C * n]
    kinds[f"paragraphs of {plen} repeated"] = np.ascontiguousarray(rep)
dup = t.copy(); dup.reshape(C, n)[:, n // 2:] = dup.reshape(C, n)[:, :n // 2]
kinds["second half = first half"] = dup
dev = torch.device("cuda:0")
kw = dict(MaxDistance=1 << 20, DistanceBitCost=66)
plan = m4.BatchPlan(C, n, 0, **kw); plan.set_timing(True)
off = np.arange(C + 1, dtype=np.uint64) * n
d_out = torch.empty(3 * (n // 4 + 2) * C, dtype=torch.int32, device=dev); d_off = torch.empty(C + 1, dtype=torch.int64, device=dev)
d_h = torch.empty(C * 1024, dtype=torch.int32, device=dev)
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, host in kinds.items():
    if only and only not in name:
        continue
    d_src = torch.empty(C * n + 64, dtype=torch.uint8, device=dev); d_src[:C * n].copy_(torch.from_numpy(host))
    plan.run(d_src, off, d_out, d_off, d_h); torch.cuda.synchronize()
    t0 = time.perf_counter(); plan.run(d_src, off, d_out, d_off, d_h); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:24s} {C*n/dt/1e9:7.2f} GB/s  stages ms {[round(x,2) for x in plan.last_timings()]}  matches/MiB {int(d_off[C].item())/C:.0f}", flush=True)
