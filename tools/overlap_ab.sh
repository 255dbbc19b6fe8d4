#!/bin/bash
# Where does the two-stream form start to pay?  bench.py at several batch sizes, overlapped (threshold lowered) against one stream.
out=gpurun_out/overlap_ab; mkdir -p $out
for c in 256 512 1024 2048; do
  for mode in ov one; do
    if [ $mode = ov ]; then export M4G_OVERLAP_MIN_CHUNKS=128; unset M4G_NO_OVERLAP; else unset M4G_OVERLAP_MIN_CHUNKS; export M4G_NO_OVERLAP=1; fi
    timeout -k 10 200 python3 bench.py --chunks $c --steps 12 --warmup 3 --no-cpu-baseline --no-end-to-end --no-gpu-encode \
      --flate-chunks 0 --bargain-chunks 0 --writer-mib 0 --verify-chunks 4 > $out/$c.$mode.log 2>&1 || { echo "$c $mode FAILED"; tail -3 $out/$c.$mode.log; exit 1; }
    python3 - "$out/$c.$mode.log" "$c" "$mode" <<'PY'
import sys, json
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(f"chunks {sys.argv[2]:5s} {sys.argv[3]:3s} value {d['value']:.2f} ms {d['ms_per_step']}  (one-stream steps {d['overlap']['one_stream_ms_per_step']})")
PY
  done
done
