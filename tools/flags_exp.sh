#!/bin/bash
# Does a scheduler strategy of the compiler move the headline?  Builds the library with one flag set at a time
# and runs the 1024-chunk bench on each (results are only read, nothing is shipped from here).
#   gpurun -- 'bash tools/flags_exp.sh'
set -u
cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --chunks 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --verify-chunks 4 --flate-chunks 0 --writer-mib 0 --bargain-chunks 0 --no-gpu-encode"
i=0
for f in "" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-mllvm -amdgpu-sched-strategy=iterative-minreg" "-O2" "-mllvm -amdgpu-sched-strategy=iterative-ilp"; do
  i=$((i+1))
  if ! /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $f -o /tmp/libm4g_f$i.so brotli_amd/csrc/m4g.hip 2> /tmp/f$i.err; then echo "[$f] does not build: $(head -1 /tmp/f$i.err)"; continue; fi
  r=$(M4G_LIB=/tmp/libm4g_f$i.so timeout -k 10 200 $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stage_ms'], 'parity', d['config'].get('parity_checked_chunks'))") || { echo "[$f] run failed"; exit 1; }
  echo "[$f] $r"
done
