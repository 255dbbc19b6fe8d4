#!/bin/bash
# tools/d_sub_ab.sh <tag> [env...] -- params D (ChainLength 64) by sub-block size: stage times per 256 x 1 MiB of text
# (extra arguments are NAME=value settings for the runs, e.g. M4G_NO_SW=1 for the chain chase)
out=gpurun_out/$1; mkdir -p $out; shift
for kv in "$@"; do export "$kv"; done
for sb in ${SUBS:-1024 512 256 128}; do
  M4G_SUB_BLOCK=$sb timeout -k 10 200 python bench.py --no-overlap --chunks 256 --params D --steps 2 --warmup 1 --no-end-to-end --no-cpu-baseline --no-gpu-encode --verify-chunks 2 --flate-chunks 0 --writer-mib 0 --bargain-chunks 0 > $out/sb$sb.json 2> $out/sb$sb.err || { echo "sub-block $sb failed"; tail -3 $out/sb$sb.err; exit 1; }
  python3 -c "
import json,sys
d=json.load(open('$out/sb$sb.json'))
print('sub_block $sb', d['value'], 'GB/s', d['stage_ms'], 'ws B/B', d['config']['workspace']['bytes_per_input_byte'])"
done
