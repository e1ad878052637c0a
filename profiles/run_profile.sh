#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   profiles/run_profile.sh <tag> [sets]        sets: any of  trace serial fetch write sq sq2 sq3   (default: all)
# kernel-trace/stats and each PMC set are separate runs (never combined with other trace domains).
#   trace   --kernel-trace --stats of the default bench (two batch slots)
#   serial  the same with --streams 1 (each kernel alone on the device)
#   fetch / write   FETCH_SIZE / WRITE_SIZE (HBM traffic, MI355X_MICROARCH.md)
#   sq      instruction counts and wave cycles          sq2  what the waves wait for, per instruction class
#   sq3     thread-level VALU cycles, SALU cycles, instruction fetch, the clock (GRBM_GUI_ACTIVE)
set -e
TAG=${1:-r03}
SETS=${2:-"trace serial fetch write sq sq2 sq3"}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-other-configs --no-e2e ${BENCH_EXTRA:-}"
for s in $SETS; do
  case $s in
    trace)  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 ;;
    serial) rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -- $BENCH --streams 1 > $OUT/trace_serial.log 2>&1 ;;
    fetch)  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH --streams 1 > $OUT/pmc_fetch.log 2>&1 ;;
    write)  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH --streams 1 > $OUT/pmc_write.log 2>&1 ;;
    sq)     rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- $BENCH --streams 1 > $OUT/pmc_sq.log 2>&1 ;;
    sq2)    rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $BENCH --streams 1 > $OUT/pmc_sq2.log 2>&1 ;;
    sq3)    rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VMEM SQ_IFETCH SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_LEVEL_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq3 -- $BENCH --streams 1 > $OUT/pmc_sq3.log 2>&1 ;;
  esac
  echo "done: $s"
done
find $OUT -name "*.csv" | head -50
