#!/bin/bash
# Collects rocprofv3 PMC counters for the bench workload, one pass per run (rocprofv3 does
# not split counters over passes).  Usage (on the GPU box, from the repo root):
#   bash tools/profile_pmc.sh [extra bench.py flags]
# Results land under gpurun_out/prof/pmc_<pass>/.
set -u
OUT=gpurun_out/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
run_pass() {
  name=$1; shift
  ctrs=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc_$name" -- \
      python3 bench.py --warmup 1 --no-cpu-baseline "$@" > "$OUT/pmc_$name.log" 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
run_pass fetch  "FETCH_SIZE" "$@" &&
run_pass write  "WRITE_SIZE" "$@" &&
run_pass l2     "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "$@" &&
run_pass ea     "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "$@" &&
run_pass sq1    "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "$@" &&
run_pass sq2    "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES" "$@" &&
run_pass tcp    "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "$@" &&
run_pass grbm   "GRBM_GUI_ACTIVE GRBM_COUNT" "$@"
