#!/bin/bash
# Memory-system counters of the kernels (one rocprofv3 --pmc pass per run).
#   gpurun -- 'bash tools/pmc_mem.sh tag [kernel-substring]'
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; kern=${2:-k3a}
run() { name=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" -d gpurun_out/pmc_$tag/$name -- python3 bench.py --chunks 512 --steps 1 --warmup 1 --no-cpu-baseline --verify-chunks 0 --no-end-to-end ${BENCH_ARGS:-} > gpurun_out/pmc_$tag/$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/pmc_$tag/$name.log; exit 1; }
}
mkdir -p gpurun_out/pmc_$tag
run m1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE && \
run m2 TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCC_REQ_sum TCC_TAG_STALL_sum && \
run m3 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY && \
run m4 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TOTAL_READ_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_READ_sum TCC_WRITE_sum
for p in m1 m2 m3 m4; do python3 tools/pmc_db.py pmc_$tag/$p $kern; done
