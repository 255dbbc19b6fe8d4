#!/bin/bash
# One measurement run on the GPU box: tests, bench lines, kernel trace, PMC traffic passes, probes.
#   gpurun --timeout 1200 -- 'bash tools/profile_run.sh r01d'
# then, here:  python tools/make_profile_summary.py gpurun_out/r01d r01
# Every GPU step runs under its own time limit; the script stops at the first failure.
set -u
out=gpurun_out/${1:-run}
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
step() { name=$1; shift; echo "== $name" >&2; "$@" || { echo "step $name failed (rc $?)" >&2; exit 1; }; }
step tests   bash -c "timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $out/gpu_tests.log 2>&1"
step smoke   bash -c "timeout -k 10 120 python __graft_entry__.py smoke > $out/smoke.log 2>&1"
step bench   bash -c "timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err"
step log     bash -c "timeout -k 10 200 python bench.py --chunks 1024 --kind log --no-end-to-end --cpu-sample-chunks 128 > $out/bench_log.json 2>> $out/bench.err"
step D       bash -c "timeout -k 10 300 python bench.py --chunks 1024 --params D --no-end-to-end --cpu-sample-chunks 32 > $out/bench_D.json 2>> $out/bench.err"
step trace   bash -c "timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --no-end-to-end --verify-chunks 2 > $out/trace.log 2>&1"
step fetch   bash -c "timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -- python3 bench.py --chunks 1024 --steps 2 --no-cpu-baseline --no-end-to-end --verify-chunks 0 > $out/pmc_fetch.log 2>&1"
step write   bash -c "timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -- python3 bench.py --chunks 1024 --steps 2 --no-cpu-baseline --no-end-to-end --verify-chunks 0 > $out/pmc_write.log 2>&1"
step kinds   bash -c "timeout -k 10 300 python tools/kind_probe.py > $out/kind_probe.log 2>&1"
step stream  bash -c "timeout -k 10 200 python tools/stream_probe.py 256 > $out/stream_probe.log 2>&1"
tail -1 "$out/bench.json" | cut -c1-400
