#!/bin/bash
# L2->fabric read request sizes + kernel time of library variants (experiments).
#   gpurun -- 'bash tools/pmc_var.sh tag kernel-substring name1 name2 ...'
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; kern=$2; shift 2
mkdir -p gpurun_out/pv_$tag
for v in "$@"; do
  export M4G_LIB=$GRAFT_REPO_ROOT/build_variants/libm4g_$v.so
  echo "== $v"
  timeout -s KILL 90 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d gpurun_out/pv_$tag/$v -- python3 bench.py --chunks 512 --steps 1 --warmup 1 --no-cpu-baseline --verify-chunks 0 --no-end-to-end > gpurun_out/pv_$tag/$v.log 2>&1 || { echo "pass failed"; tail -3 gpurun_out/pv_$tag/$v.log; exit 1; }
  python3 tools/pmc_db.py pv_$tag/$v $kern
  timeout -k 10 120 python3 bench.py --chunks 1024 --steps 3 --no-cpu-baseline --no-end-to-end --verify-chunks 0 > gpurun_out/pv_$tag/$v.bench 2> gpurun_out/pv_$tag/$v.err || { echo "bench failed"; tail -3 gpurun_out/pv_$tag/$v.err; exit 1; }
  grep -o '"stage_ms": {[^}]*}' gpurun_out/pv_$tag/$v.bench
done
