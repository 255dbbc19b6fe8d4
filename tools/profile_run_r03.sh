#!/bin/bash
# Round-3 measurement run on the GPU box: tests, bench lines, kernel traces, PMC passes, probes.
#   gpurun --timeout 1200 -- 'bash tools/profile_run_r03.sh r03a'
# then, here:  python tools/make_profile_summary_r03.py gpurun_out/r03a r03
# Every GPU step runs under its own time limit; the script stops at the first failure.
set -u
out=gpurun_out/${1:-run}
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
step() { name=$1; shift; echo "== $name" >&2; "$@" || { echo "step $name failed (rc $?)" >&2; exit 1; }; }
B1K="python3 bench.py --no-overlap --chunks 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --verify-chunks 0 --flate-chunks 0 --writer-mib 0 --bargain-chunks 0"
pmc() { name=$1; shift; args=$1; shift
  step "pmc_$name" bash -c "timeout -s KILL 150 rocprofv3 --pmc $* -d $out/pmc_$name -- $B1K $args > $out/pmc_$name.log 2>&1"; }
step tests   bash -c "timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $out/gpu_tests.log 2>&1"
step smoke   bash -c "timeout -k 10 120 python __graft_entry__.py smoke > $out/smoke.log 2>&1"
step bench   bash -c "timeout -k 10 500 python bench.py > $out/bench.json 2> $out/bench.err"
step log     bash -c "timeout -k 10 200 python bench.py --chunks 1024 --kind log --no-end-to-end --writer-mib 0 > $out/bench_log.json 2>> $out/bench.err"
step D       bash -c "timeout -k 10 300 python bench.py --chunks 1024 --params D --no-end-to-end --flate-chunks 0 --writer-mib 0 --bargain-chunks 0 > $out/bench_D.json 2>> $out/bench.err"
step trace   bash -c "timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-overlap --no-cpu-baseline --no-end-to-end --verify-chunks 2 --flate-chunks 0 --writer-mib 0 --bargain-chunks 0 > $out/trace.log 2>&1"
step traceov bash -c "timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_ov -- python3 bench.py --no-cpu-baseline --no-end-to-end --no-gpu-encode --verify-chunks 2 --flate-chunks 0 --writer-mib 0 --bargain-chunks 0 > $out/trace_ov.log 2>&1"
step tracel  bash -c "timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_legs -- python3 bench.py --chunks 256 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --verify-chunks 0 > $out/trace_legs.log 2>&1"
step trace1k bash -c "timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace1k -- $B1K > $out/trace1k.log 2>&1"
pmc fetch "" FETCH_SIZE
pmc write "" WRITE_SIZE
pmc sq1 "" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
pmc sq2 "" SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_SCA
pmc fetch_D "--params D --no-gpu-encode" FETCH_SIZE
pmc write_D "--params D --no-gpu-encode" WRITE_SIZE
pmc fetch_log "--kind log --no-gpu-encode" FETCH_SIZE
pmc write_log "--kind log --no-gpu-encode" WRITE_SIZE
step pmcx    bash -c "python3 tools/pmc_extract.py $out > $out/pmc_extract.log 2>&1"
step kinds   bash -c "timeout -k 10 300 python tools/kind_probe.py > $out/kind_probe.log 2>&1"
step stream  bash -c "timeout -k 10 200 python tools/stream_probe.py 256 > $out/stream_probe.log 2>&1"
step hostb   bash -c "timeout -k 10 200 python tools/hostbatch_probe.py 2048 > $out/hostbatch_probe.log 2>&1"
step writer  bash -c "timeout -k 10 200 python tools/writer_probe.py 256 > $out/writer_probe.log 2>&1"
step bargain bash -c "(for a in '64 1' '512 1' '512 2' '512 3'; do timeout -k 10 300 python tools/bargain_probe.py \$a 1 || exit 1; done) > $out/bargain_probe.log 2>&1"
step bstats  bash -c "timeout -k 10 400 bash tools/bargain_stats.sh > $out/bargain_stats.log 2>&1"
step v2      bash -c "timeout -k 10 400 python tools/v2_probe.py 256 > $out/v2_probe.log 2>&1"
step compr   bash -c "timeout -k 10 200 python tools/compress_probe.py 256 > $out/compress_probe.log 2>&1"
tail -1 "$out/bench.json" | cut -c1-400
