#!/bin/bash
# Kernel trace of one bench run under given environment settings (experiments).
#   gpurun -- 'bash tools/trace_env.sh tag "VAR=val VAR=val" [bench args]'
set -u
tag=$1; setting=$2; shift 2
out=gpurun_out/tr_$tag
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
[ -n "$setting" ] && export $setting
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 bench.py --chunks ${CHUNKS:-1024} --steps 3 --no-cpu-baseline --no-end-to-end --verify-chunks ${VERIFY:-2} "$@" > $out/t.log 2>&1 || { echo "failed" >&2; tail -5 $out/t.log >&2; exit 1; }
f=$(find $out/t -name '*kernel_stats.csv' | sort | tail -1)
echo "== $tag: $setting $*"
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['AverageNs']) > 20000:
        print(f"  {r['Name'][:40]:40s} calls {r['Calls']:>4s} avg_ms {float(r['AverageNs'])/1e6:9.3f}")
PY
