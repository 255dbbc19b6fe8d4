#!/bin/bash
# bench stage times of ONE library under different environment settings (experiments).
#   gpurun -- 'bash tools/env_probe.sh tag lib "VAR=val ..." "VAR=val ..."'
set -u
tag=$1; lib=$2; shift 2
out=gpurun_out/env_$tag
mkdir -p "$out"
[ "$lib" != "-" ] && export M4G_LIB=$GRAFT_REPO_ROOT/build_variants/libm4g_$lib.so
i=0
for setting in "$@"; do
  i=$((i+1))
  ( [ -n "$setting" ] && export $setting; timeout -k 10 150 python3 bench.py --chunks ${CHUNKS:-1024} --steps 3 --no-cpu-baseline --no-end-to-end --verify-chunks ${VERIFY:-2} ${BENCH_ARGS:-} > $out/$i.log 2> $out/$i.err ) || { echo "setting '$setting' failed" >&2; tail -5 $out/$i.err >&2; exit 1; }
  echo "== $setting" >> $out/summary.txt
  tail -1 $out/$i.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  value', d['value'], d['stage_ms'], 'parity', d['config'].get('parity_checked_chunks'))" >> $out/summary.txt
done
cat $out/summary.txt
