#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { name=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" -d gpurun_out/$name -- python3 bench.py --chunks 512 --steps 1 --warmup 1 --no-cpu-baseline --verify-chunks 0 --no-end-to-end > gpurun_out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/$name.log; }
}
run pq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU && \
run pq2 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_WAVES
python3 tools/pmc_db.py pq1 k3a; python3 tools/pmc_db.py pq2 k3a
