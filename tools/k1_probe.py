"""Profiling aid: time kernel K1 alone with parts of it switched off."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from brotli_amd import m4, synth
C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = sys.argv[2] if len(sys.argv) > 2 else "text"
n = 1 << 20
host = np.empty(n * C, dtype=np.uint8); synth.fill(host, n, 0, kind, 16)
dev = torch.device("cuda:0")
d_src = torch.empty(n * C + 64, dtype=torch.uint8, device=dev); d_src[:n * C].copy_(torch.from_numpy(host))
plan = m4.BatchPlan(C, n, 0, MaxDistance=1 << 20, DistanceBitCost=66)
off = np.arange(C + 1, dtype=np.uint64) * n
d_out = torch.empty(3 * (n // 6) * C, dtype=torch.int32, device=dev); d_off = torch.empty(C + 1, dtype=torch.int64, device=dev)
plan.run(d_src, off, d_out, d_off, None); torch.cuda.synchronize()
L = m4.lib(); L.m4g_debug_k1_only.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
for dbg in [int(x) for x in (sys.argv[3].split(',') if len(sys.argv) > 3 else '0,1,4,2'.split(','))]:
    ms = ctypes.c_float()
    rc = L.m4g_debug_k1_only(plan._h, d_src.data_ptr(), n * C, C, dbg, 3, ctypes.byref(ms))
    print(f"{kind} chunks {C} dbg {dbg}: rc {rc} {ms.value:.3f} ms  ({n*C/ms.value/1e6:.1f} GB/s)")
