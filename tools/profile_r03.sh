#!/bin/bash
# Counter record of the bench workload's trace kernel: rocprofv3 PMC passes, one run of the same command per pass
# (rocprofv3 does not split counters over passes), then tools/make_traffic.py turns them into the record bench.py
# prices (traffic.json) and a readable summary.  On the GPU box, from the repo root:
#   bash tools/profile_r03.sh [tag]        -> gpurun_out/<tag>/{traffic.json,summary.txt}
# The command of every pass: bench.py --warmup 1 --steps 200 --probe 16 --settle-ms 0 --no-cpu-baseline --no-literal, i.e. one set-up
# launch, 201 overlapped launches (two streams, rows in plain order) and 16 launches one at a time (measured dispatch order).
TAG=${1:-r03prof}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
run_pass() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- \
      python3 bench.py --warmup 1 --steps 200 --probe 16 --settle-ms 0 --no-cpu-baseline --no-literal > "$OUT/pmc_$name.log" 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  [ $rc -eq 0 ] || tail -n 5 "$OUT/pmc_$name.log"
  return $rc
}
run_pass fetch FETCH_SIZE &&
run_pass write WRITE_SIZE &&
run_pass l2    TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
run_pass ea    TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum &&
run_pass sq1   SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY &&
run_pass sq2   SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM &&
run_pass tcp   TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum &&
python3 tools/make_traffic.py "$OUT" | tee "$OUT/summary.txt"
