#!/bin/bash
# A short GPU-box session while a kernel is being worked on: the world and parity tests, the default bench line without
# the CPU baseline, the other configurations.  Every step has its own time limit; the chain stops at the first failure.
#   gpurun --timeout 900 -- bash tools/gpu_quick.sh [tag] [pytest -k expression]
TAG=${1:-quick}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread ${2:+-k "$2"} > "$OUT/tests.log" 2>&1
rc=$?; tail -n 6 "$OUT/tests.log"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 240 python bench.py --no-cpu-baseline --no-literal > "$OUT/bench.json" 2> "$OUT/bench.err"
rc=$?; [ $rc -eq 0 ] || { tail -n 20 "$OUT/bench.err"; exit $rc; }
python3 -c "import json; d=json.load(open('$OUT/bench.json')); r=d['roofline']; print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'parity', d.get('parity'))"
timeout -k 10 400 python -u tools/bench_configs.py > "$OUT/other_configs.json" 2> "$OUT/other_configs.err"
rc=$?; cat "$OUT/other_configs.json"; exit $rc
