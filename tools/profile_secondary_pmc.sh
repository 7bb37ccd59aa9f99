#!/bin/bash
# rocprofv3 counter passes (one run per pass) for the two secondary-ray kernels on the secondary-ray workload.
#   bash tools/profile_secondary_pmc.sh   -> gpurun_out/prof_sec/<variant>/pmc_<pass>/ and a summary per variant
set -u
export TMPDIR=/tmp
for variant in ${VARIANTS:-pixel binned}; do
  OUT=gpurun_out/prof_sec/$variant
  mkdir -p "$OUT"
  run_pass() {
    name=$1; ctrs=$2
    timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc_$name" -- python3 tools/sec_run.py $variant 8 > "$OUT/pmc_$name.log" 2>&1
    rc=$?; echo "$variant pass $name rc=$rc"; return $rc
  }
  run_pass sq1 "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" &&
  run_pass sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" &&
  run_pass l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" &&
  run_pass tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" &&
  run_pass fetch "FETCH_SIZE" && run_pass write "WRITE_SIZE" || exit 1
  python3 tools/summarize_pmc.py "$OUT" k_trace_secondary > "$OUT/summary.txt"
  (cd /tmp && timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/ktrace" -- python3 "$OLDPWD/tools/sec_run.py" $variant 12 > "$OLDPWD/$OUT/ktrace.log" 2>&1) || exit 1
  cat "$OUT/summary.txt"
  find "$OUT/ktrace" -name "*kernel_stats.csv" | head -n 1 | xargs -r grep secondary
done
