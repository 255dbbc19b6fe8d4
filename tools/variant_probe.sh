#!/bin/bash
# Per-kernel times of library builds under build_variants/ (experiments; not product code).
#   gpurun -- 'bash tools/variant_probe.sh tag name1 name2 ...'   [BENCH_ARGS="--params D" in the env]
set -u
tag=$1; shift
out=gpurun_out/var_$tag
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
for v in "$@"; do
  export M4G_LIB=$GRAFT_REPO_ROOT/build_variants/libm4g_$v.so
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$v -- python3 bench.py --chunks ${CHUNKS:-1024} --steps 3 --no-cpu-baseline --no-end-to-end --verify-chunks ${VERIFY:-0} ${BENCH_ARGS:-} > $out/$v.log 2>&1 || { echo "variant $v failed" >&2; tail -5 $out/$v.log >&2; exit 1; }
  f=$(find $out/$v -name '*kernel_stats.csv' | sort | tail -1)
  echo "== $v" >> $out/summary.txt
  python3 - "$f" >> $out/summary.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"  {r['Name'][:40]:40s} calls {r['Calls']:>4s} avg_ms {float(r['AverageNs'])/1e6:9.3f}")
PY
  tail -1 $out/$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('  value', d['value'], d['stage_ms'])" >> $out/summary.txt
done
cat $out/summary.txt
