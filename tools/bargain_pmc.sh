#!/bin/bash
# SQ counters of the Bargain search kernel (two passes; the guide's limit is 8 SQ counters per pass).
#   gpurun -- 'bash tools/bargain_pmc.sh [variant]'
set -u
v=${1:-1}
out=gpurun_out/bargain_pmc
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p $out
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $out/p1 -- python3 tools/bargain_probe.py 512 $v 1 > $out/p1.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_SALU --output-format csv -d $out/p2 -- python3 tools/bargain_probe.py 512 $v 1 > $out/p2.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2"):
    for f in glob.glob(f"gpurun_out/bargain_pmc/{p}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            if "kb_parse" in row["Kernel_Name"]:
                acc[row["Counter_Name"]] += float(row["Counter_Value"])
        for k, v in sorted(acc.items()):
            print(f"{k} = {v:.4g}")
PY
