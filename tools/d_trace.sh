#!/bin/bash
# tools/d_trace.sh <tag> [NAME=value ...] -- kernel trace of the params D bench (256 x 1 MiB of text)
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-overlap --chunks 256 --params D --steps 2 --warmup 1 --no-end-to-end --no-cpu-baseline --no-gpu-encode --verify-chunks 2 --flate-chunks 0 --writer-mib 0 --bargain-chunks 0 > $out/bench.json 2> $out/trace.err || { tail -5 $out/trace.err; exit 1; }
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/trace
echo "== $tag $@"; python3 tools/kstats.py $out 2>/dev/null | head -9
