#!/bin/bash
# One rocprofv3 --pmc pass over a short bench run, printed per kernel.
#   gpurun -- 'bash tools/pmc_one.sh tag kernel-substring COUNTER [COUNTER ...]'
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; kern=$2; shift 2
mkdir -p gpurun_out/pmc_$tag
timeout -s KILL 90 rocprofv3 --pmc "$@" -d gpurun_out/pmc_$tag/p -- python3 bench.py --chunks 512 --steps 1 --warmup 1 --no-cpu-baseline --verify-chunks 0 --no-end-to-end ${BENCH_ARGS:-} > gpurun_out/pmc_$tag/p.log 2>&1 || { echo "pass failed"; tail -3 gpurun_out/pmc_$tag/p.log; exit 1; }
python3 tools/pmc_db.py pmc_$tag/p $kern
