"""The three forms of the chain walk (chase / sorted walk / K2) on several inputs and chain lengths:
batch throughput of a plan of KP_CHUNKS x 1 MiB (default 128), stage times from the library's events."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from brotli_amd import m4, synth
C, n = int(os.environ.get("KP_CHUNKS", "128")), 1 << 20
rng = np.random.default_rng(1)
t = np.empty(C * n, dtype=np.uint8); synth.fill(t, n, 0, "text", 16)
lg = np.empty(C * n, dtype=np.uint8); synth.fill(lg, n, 0, "log", 16)
words = [bytes(rng.integers(97, 123, int(rng.integers(2, 9)), dtype=np.uint8)) for _ in range(2000)]
dic = np.frombuffer(b" ".join(words[int(j)] for j in rng.integers(0, 2000, C * n // 5))[:C * n].ljust(C * n, b"x"), dtype=np.uint8)
kinds = {"text": t, "log records": lg, "4 symbols": rng.integers(0, 4, C * n, dtype=np.uint8),
         "16 symbols": rng.integers(0, 16, C * n, dtype=np.uint8), "2000 words at random": dic}
dev = torch.device("cuda:0")
off = np.arange(C + 1, dtype=np.uint64) * n
d_out = torch.empty(3 * (n // 3 + 2) * C, dtype=torch.int32, device=dev); d_off = torch.empty(C + 1, dtype=torch.int64, device=dev)
forms = {"chase": dict(M4G_K2="0", M4G_SW="0"), "sorted walk": dict(M4G_K2="0", M4G_SW="1"), "K2": dict(M4G_K2="1", M4G_SW="0")}
for chain in (8, 64):
    kw = dict(MaxDistance=1 << 20, HashLen=5, ChainLength=chain, DistanceBitCost=66)
    for name, host in kinds.items():
        d_src = torch.empty(C * n + 64, dtype=torch.uint8, device=dev); d_src[:C * n].copy_(torch.from_numpy(np.ascontiguousarray(host)))
        row, counts = [], set()
        for form, env in forms.items():
            os.environ.update(env)
            plan = m4.BatchPlan(C, n, 0, **kw); plan.set_timing(True)
            plan.run(d_src, off, d_out, d_off); plan.status(); torch.cuda.synchronize()
            t0 = time.perf_counter(); plan.run(d_src, off, d_out, d_off); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            counts.add(int(d_off[C].item()))
            ms = plan.last_timings()
            row.append(f"{form} {C*n/dt/1e9:6.2f} GB/s (K1+sort+K2 {ms[0]:6.1f} ms, K3 {ms[1]:6.1f})")
            del plan
        assert len(counts) == 1, counts
        print(f"ChainLength {chain:2d} {name:22s} " + " | ".join(row) + f" | matches/MiB {counts.pop() // C}", flush=True)
