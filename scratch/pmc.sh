#!/bin/bash
# usage: pmc.sh <outdir> <bench args...>
set -u
out=$1; shift
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$out
run() { # name counters...
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/$out/$name -- python3 bench.py "$BARGS" > gpurun_out/$out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/$out/$name.log; return 1; }
}
