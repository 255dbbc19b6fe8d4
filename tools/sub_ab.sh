#!/bin/bash
# A/B of the sub-block size (M4G_SUB_BLOCK) on the main batch leg.   gpurun -- 'bash tools/sub_ab.sh 2048 4096 ...'
out=gpurun_out/sub_ab
mkdir -p $out
for v in "$@"; do
  export M4G_SUB_BLOCK=$v
  timeout -k 10 200 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-gpu-encode \
    --flate-chunks 0 --bargain-chunks 0 --writer-mib 0 --verify-chunks 8 > $out/$v.log 2>&1 || { echo "sub $v FAILED"; tail -3 $out/$v.log; exit 1; }
  python3 - "$out/$v.log" "$v" <<'PY' | tee -a $out/summary.txt
import sys, json
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(f"sub {sys.argv[2]:6s} value {d['value']:.2f}  stages {d['stage_ms']}  ws {d['config']['workspace']}")
PY
done
