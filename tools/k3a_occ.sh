# K3a with more waves per SIMD: smaller LDS windows (M4G_K3_FWD) and register caps (M4G_K3_MINWAVES) -- variant libraries
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "" f16_w1 f16_w5 f0_w5 f0_w6; do
  echo "== ${lib:-default}"
  if [ -n "$lib" ]; then export M4G_LIB=$GRAFT_REPO_ROOT/build_variants/libm4g_$lib.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k3o -- python3 bench.py --chunks 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --verify-chunks 8 --no-gpu-encode > gpurun_out/k3o.log 2>&1
  python3 tools/kstats.py gpurun_out/k3o | grep "k3a\|k3b"
  rm -rf gpurun_out/k3o
done
