#!/bin/bash
# tools/k2_dbg.sh <tag> <walk> -- timing aid: k2_eval with its walk cut to <walk> candidates (results are wrong on purpose)
tag=$1; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp M4G_K2_DEBUG_WALK=$2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-overlap --chunks 256 --params D --steps 2 --warmup 1 --no-end-to-end --no-cpu-baseline --no-gpu-encode --verify-chunks 0 --flate-chunks 0 --writer-mib 0 --bargain-chunks 0 > $out/bench.json 2> $out/trace.err
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/trace
python3 tools/kstats.py $out 2>/dev/null | head -4
