#!/bin/bash
# Event counts of the Bargain search kernel per position (debug build with -DKB_STATS, not the shipped library).
#   gpurun -- 'bash tools/bargain_stats.sh'
set -eu
cd "$GRAFT_REPO_ROOT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DKB_STATS -o /tmp/libm4g_stats.so brotli_amd/csrc/m4g.hip
M4G_LIB=/tmp/libm4g_stats.so timeout -k 10 300 python3 - <<'PY'
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from brotli_amd import m4, synth
for variant in (1, 2):
    chunks, n, bs = 64, 1 << 20, 1 << 16
    host = np.empty(n * chunks, dtype=np.uint8)
    synth.fill(host, n, 0, "text", 16)
    d_src = torch.empty(n * chunks + 64, dtype=torch.uint8, device="cuda")
    d_src[:n * chunks].copy_(torch.from_numpy(host))
    off = np.arange(chunks + 1, dtype=np.uint64) * n
    plan = m4.BargainPlan(chunks, n * chunks, n, variant=variant)
    nb = plan.block_count(off)
    d_out = torch.empty((n * chunks // 3 + 2 * nb + 8) * 3, dtype=torch.int32, device="cuda")
    d_off = torch.empty(nb + 1, dtype=torch.int64, device="cuda")
    st = (ctypes.c_ulonglong * 8)()
    m4.lib().m4g_debug_bargain_stats(st)
    nm = plan.run(d_src, off, d_out, d_off)
    m4.lib().m4g_debug_bargain_stats(st)
    names = ["repeat-probe evaluations", "addMatch calls", "addMatch parts outside the register window", "long forward extensions",
             "long backward extensions", "prevDistance lookups outside the window", "hash candidates taken", "repeat matches"]
    tot = n * chunks
    print(f"Bargain{variant}, {chunks} MiB of text, search {plan.last_timings()[2]:.1f} ms, {nm} matches")
    for k, nme in enumerate(names):
        print(f"  {nme}: {st[k]} = {st[k] / tot:.4f} per position, {st[k] * 64 / tot:.2f} per batch of 64")
PY
