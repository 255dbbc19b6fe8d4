#!/bin/bash
# Where the Bargain search kernel's time goes: debug builds with parts of the walk compiled out
# (results are wrong on purpose; only the search time is read).   gpurun -- 'bash tools/bargain_exp.sh'
set -eu
cd "$GRAFT_REPO_ROOT"
kind=${1:-text}
for e in 0 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DKB_EXP=$e -o /tmp/libm4g_exp$e.so brotli_amd/csrc/m4g.hip
  for c in 64 512; do
    echo "KB_EXP=$e chunks=$c: $(M4G_LIB=/tmp/libm4g_exp$e.so timeout -k 10 300 python3 tools/bargain_probe.py $c 1 1 $kind 2>&1 | grep '^Bargain' | sed 's/.*stages ms/stages ms/')"
  done
done
