import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from brotli_amd import m4, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 17
t = np.frombuffer(synth.text(1 << 20, 0), dtype=np.uint8)
rng = np.random.default_rng(1)
plen = 2000
pool = [t[i * plen:(i + 1) * plen] for i in range(16)]
rep = np.ascontiguousarray(np.concatenate([pool[int(j)] for j in rng.integers(0, 16, n // plen + 1)])[:n])
dev = torch.device("cuda:0")
plan = m4.BatchPlan(1, n, 0, MaxDistance=1 << 20, DistanceBitCost=66)
d_src = torch.empty(n + 64, dtype=torch.uint8, device=dev); d_src[:n].copy_(torch.from_numpy(rep))
d_out = torch.empty(3 * (n // 4 + 2), dtype=torch.int32, device=dev); d_off = torch.empty(2, dtype=torch.int64, device=dev)
plan.run(d_src, np.array([0, n], dtype=np.uint64), d_out, d_off, None); torch.cuda.synchronize()
print("matches", int(d_off[1].item()))
