#!/bin/bash
# Timings of params D, G and the flate level-6 leg for a build with extra compiler definitions.
#   gpurun -- 'bash tools/hop_exp.sh "-DM4G_K3_MINWAVES=4"'
set -u
cd "$GRAFT_REPO_ROOT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared ${1:-} -o /tmp/libm4g_hop.so brotli_amd/csrc/m4g.hip || exit 1
export M4G_LIB=/tmp/libm4g_hop.so
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["stage_ms"], d["config"].get("parity_checked_chunks")); f=d.get("gpu_flate"); print("  flate", f["find_matches_value"], f["find_matches_stage_ms"]) if f else None'
timeout -k 10 300 python3 bench.py --chunks 1024 --params D --no-end-to-end --flate-chunks 0 --writer-mib 0 --bargain-chunks 0 --no-cpu-baseline 2>/dev/null | python3 -c "$P" D || exit 1
timeout -k 10 300 python3 bench.py --chunks 1024 --no-end-to-end --writer-mib 0 --bargain-chunks 0 --no-cpu-baseline --steps 3 2>/dev/null | python3 -c "$P" G || exit 1
