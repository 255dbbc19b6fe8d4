#!/bin/bash
# A/B of library builds under build_variants/ on the main batch leg only (stage times from the
# library's own HIP events; every variant is also compared with the oracle on VERIFY chunks).
#   gpurun -- 'bash tools/k3_ab.sh tag name1 name2 ...'     [BENCH_ARGS, CHUNKS, VERIFY in the env]
set -u
tag=$1; shift
out=gpurun_out/ab_$tag
mkdir -p "$out"
for v in "$@"; do
  export M4G_LIB=$GRAFT_REPO_ROOT/build_variants/libm4g_$v.so
  timeout -k 10 200 python3 bench.py --chunks ${CHUNKS:-1024} --steps ${STEPS:-5} --no-cpu-baseline --no-end-to-end --no-gpu-encode \
    --flate-chunks 0 --bargain-chunks 0 --writer-mib 0 --verify-chunks ${VERIFY:-64} ${BENCH_ARGS:-} > $out/$v.log 2>&1 \
    || { echo "variant $v FAILED" | tee -a $out/summary.txt; tail -5 $out/$v.log; continue; }
  python3 - "$out/$v.log" "$v" <<'PY' | tee -a $out/summary.txt
import sys, json
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(f"{sys.argv[2]:16s} value {d['value']:.2f}  stages {d['stage_ms']}  verified {d.get('parity_checked_chunks')}")
PY
done
