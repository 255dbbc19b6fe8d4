# K1 (split) on text and log records: kernel durations and the un-split kernel beside them
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in text log; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k1k_$k -- python3 tools/k1_probe.py 256 $k -1 > gpurun_out/k1k_$k.log 2>&1
  python3 tools/kstats.py gpurun_out/k1k_$k | grep -i "k1" || true
  M4G_K1_NOSPLIT=1 timeout -k 10 120 python3 tools/k1_probe.py 256 $k -1 | tail -1
done
