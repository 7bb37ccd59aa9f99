#!/bin/bash
# Every long parity run against the CPU oracle in one GPU-box session (each prints progress every few seconds):
#   bash tools/soak_all.sh [tag] [seed]   -> gpurun_out/<tag>/soak_*.txt
TAG=${1:-soak}; SEED=${2:-31}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
for kind in default odd sparse axis; do
  timeout -k 10 600 python -u tools/soak_parity.py 40 "$SEED" $kind > "$OUT/soak_parity_$kind.txt" 2>&1
  rc=$?; tail -n 1 "$OUT/soak_parity_$kind.txt"; [ $rc -eq 0 ] || exit $rc
done
timeout -k 10 600 python -u tools/soak_secondary.py 60 "$SEED" > "$OUT/soak_secondary.txt" 2>&1
rc=$?; tail -n 1 "$OUT/soak_secondary.txt"; exit $rc
