#!/bin/bash
# tools/d_budget_ab.sh <tag> -- params D (chain chase, 128-byte sub-blocks) by k3a_spec's round budget (0: lockstep epochs)
out=gpurun_out/$1; mkdir -p $out
for b in 0 8 16 24 48 1000; do
  M4G_K3_BUDGET=$b timeout -k 10 200 python bench.py --no-overlap --chunks 256 --params D --steps 2 --warmup 1 --no-end-to-end --no-cpu-baseline --no-gpu-encode --verify-chunks 2 --flate-chunks 0 --writer-mib 0 --bargain-chunks 0 > $out/b$b.json 2> $out/b$b.err || { echo "budget $b failed"; tail -3 $out/b$b.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$out/b$b.json'))
print('budget $b', d['value'], 'GB/s', d['stage_ms'])"
done
