"""K3 + K4 stage times of chain-walking plans by sub-block size (M4G_SUB_BLOCK), several chain lengths and windows."""
import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from brotli_amd import m4, synth
C, n = int(os.environ.get("KP_CHUNKS", "128")), 1 << 20
t = np.empty(C * n, dtype=np.uint8); synth.fill(t, n, 0, "text", 16)
lg = np.empty(C * n, dtype=np.uint8); synth.fill(lg, n, 0, "log", 16)
dev = torch.device("cuda:0")
off = np.arange(C + 1, dtype=np.uint64) * n
d_out = torch.empty(3 * (n // 3 + 2) * C, dtype=torch.int32, device=dev); d_off = torch.empty(C + 1, dtype=torch.int64, device=dev)
for chain, hl, md in ((1, 6, 1 << 20), (8, 5, 1 << 20), (8, 4, 32768), (32, 4, 32768), (2, 5, 32768), (16, 5, 32768), (64, 5, 32768), (64, 5, 16777200)):
    kw = dict(MaxDistance=md, HashLen=hl, ChainLength=chain, DistanceBitCost=66)
    for name, host in (("text", t), ("log", lg)):
        d_src = torch.empty(C * n + 64, dtype=torch.uint8, device=dev); d_src[:C * n].copy_(torch.from_numpy(host))
        row = []
        for sb in (1024, 512, 256, 128):
            os.environ["M4G_SUB_BLOCK"] = str(sb)
            plan = m4.BatchPlan(C, n, 0, **kw); plan.set_timing(True)
            plan.run(d_src, off, d_out, d_off); plan.status(); torch.cuda.synchronize()
            plan.run(d_src, off, d_out, d_off); torch.cuda.synchronize()
            ms = plan.last_timings()
            row.append(f"{sb}: K3 {ms[1]:5.1f} K4 {ms[2]:4.2f}")
            del plan
        print(f"chain {chain} hashlen {hl} maxdist {md} {name}: " + " | ".join(row), flush=True)
