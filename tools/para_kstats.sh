set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in "paragraphs of 2000" "paragraphs of 40000"; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kp -- python3 tools/kind_probe.py "$k" > gpurun_out/kp.log 2>&1
tail -1 gpurun_out/kp.log
python3 tools/kstats.py gpurun_out/kp | grep "k3\|k1\|k4_c\|k5_h"
rm -rf gpurun_out/kp
done
