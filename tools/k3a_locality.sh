# K3a against the number of chunks in flight per XCD: occupancy limited with unused LDS, sub-block sizes
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sb in 1024 512 256; do
  for pad in 0 20000 40000 70000; do
    echo "== sub_block $sb lds_pad $pad"
    export M4G_SUB_BLOCK=$sb M4G_K3A_LDS_PAD=$pad
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k3l -- python3 bench.py --chunks 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --verify-chunks 0 --no-gpu-encode > gpurun_out/k3l.log 2>&1
    python3 tools/kstats.py gpurun_out/k3l | grep "k3\|k4_compact"
    rm -rf gpurun_out/k3l
  done
done
