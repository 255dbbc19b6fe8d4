#!/bin/bash
# One library build under different environment settings (main batch leg only).
#   gpurun -- 'bash tools/k3_env_ab.sh tag variant "VAR=1" "VAR=2 OTHER=3" ...'
set -u
tag=$1; v=$2; shift 2
out=gpurun_out/env_$tag
mkdir -p "$out"
export M4G_LIB=$GRAFT_REPO_ROOT/build_variants/libm4g_$v.so
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 200 python3 bench.py --chunks ${CHUNKS:-1024} --steps ${STEPS:-5} --no-cpu-baseline --no-end-to-end --no-gpu-encode \
    --flate-chunks 0 --bargain-chunks 0 --writer-mib 0 --verify-chunks ${VERIFY:-64} ${BENCH_ARGS:-} > $out/$i.log 2>&1 \
    || { echo "$e FAILED"; tail -5 $out/$i.log; continue; }
  python3 - "$out/$i.log" "$e" <<'PY'
import sys, json
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(f"{sys.argv[2]:28s} value {d['value']:.2f}  stages {d['stage_ms']}")
PY
done
