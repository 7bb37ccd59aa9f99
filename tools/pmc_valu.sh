#!/bin/bash
# Vector / scalar instruction counts per launch of the trace kernel of the bench workload: one rocprofv3 counter
# pass over `bench.py --warmup 1 --steps 200 --probe 16` (201 overlapped launches, which run their rows in plain
# order, then the set-up and probe launches one at a time, which run the measured dispatch order).
#   bash tools/pmc_valu.sh [tag] [extra bench.py flags]     -> gpurun_out/<tag>/valu.txt
TAG=${1:-pmc_valu}; shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES --output-format csv \
    -d "$OUT/pmc" -- python3 bench.py --warmup 1 --steps 200 --probe 16 --no-cpu-baseline --no-literal "$@" > "$OUT/pmc.log" 2>&1
rc=$?; echo "pmc pass rc=$rc"; [ $rc -eq 0 ] || { tail -n 20 "$OUT/pmc.log"; exit $rc; }
python3 tools/pmc_valu_summary.py "$OUT/pmc" | tee "$OUT/valu.txt"
