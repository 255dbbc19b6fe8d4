#!/bin/bash
# tools/k2_trace.sh <tag> [chunks] -- on the GPU box: K2's GPU tests, then a kernel trace of the params D bench.
tag=${1:-k2}; chunks=${2:-256}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 python -u -m pytest tests/test_gpu_k2.py -x -q --timeout 180 --timeout-method thread > $out/test.log 2>&1 || { tail -20 $out/test.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-overlap --chunks $chunks --params D --steps 2 --warmup 1 --no-end-to-end --no-cpu-baseline --no-gpu-encode --verify-chunks 4 --flate-chunks 0 --writer-mib 0 --bargain-chunks 0 > $out/bench.json 2> $out/trace.err || { tail -5 $out/trace.err; exit 1; }
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/trace
python3 tools/kstats.py $out 2>/dev/null | head -16 || true
