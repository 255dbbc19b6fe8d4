#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { name=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" -d gpurun_out/$name -- python3 tools/k1_probe.py 256 text 0 > gpurun_out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/$name.log; }
}
run pk1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU && \
run pk2 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT && \
run pk3 SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_I8 SQ_WAVES SQ_INSTS_SMEM
python3 tools/pmc_db.py pk1 k1x; python3 tools/pmc_db.py pk2 k1x; python3 tools/pmc_db.py pk3 k1x
