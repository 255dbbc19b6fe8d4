# per-kernel times on repeated-paragraph inputs (optionally with a variant library: M4G_LIB)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in "paragraphs of 2000" "paragraphs of 8192"; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kp -- python3 tools/kind_probe.py "$k" > gpurun_out/kp.log 2>&1
grep GB/s gpurun_out/kp.log
python3 tools/kstats.py gpurun_out/kp | grep "k3"
rm -rf gpurun_out/kp
done
