#!/bin/bash
# A/B of the compact record slot size (M4G_SLOT_RECS) on the main batch leg.
#   gpurun -- 'bash tools/slot_ab.sh 258 256 264 ...'
out=gpurun_out/slot_ab
mkdir -p $out
for v in "$@"; do
  if [ "$v" = worst ]; then export M4G_SLOTS=worst; unset M4G_SLOT_RECS; else unset M4G_SLOTS; export M4G_SLOT_RECS=$v; fi
  timeout -k 10 200 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-gpu-encode \
    --flate-chunks 0 --bargain-chunks 0 --writer-mib 0 --verify-chunks 8 > $out/$v.log 2>&1 || { echo "slot $v FAILED"; tail -3 $out/$v.log; exit 1; }
  python3 - "$out/$v.log" "$v" <<'PY' | tee -a $out/summary.txt
import sys, json
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(f"slot {sys.argv[2]:6s} value {d['value']:.2f}  stages {d['stage_ms']}  bytes/B {d['config']['workspace']['bytes_per_input_byte']}")
PY
done
