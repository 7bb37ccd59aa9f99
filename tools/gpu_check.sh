#!/bin/bash
# One GPU-box session: the GPU test suite, the smoke check, the default bench line and its kernel-trace profile.
# Every step has its own time limit and the chain stops at the first failure.
#   gpurun --timeout 1100 -- bash tools/gpu_check.sh [tag]
TAG=${1:-check}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 840 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread --durations=12 > "$OUT/tests.log" 2>&1
rc=$?; tail -n 25 "$OUT/tests.log"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python -u __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1
rc=$?; tail -n 3 "$OUT/smoke.log"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 240 python bench.py --steps 20 --warmup 5 > "$OUT/bench_20.json" 2> "$OUT/bench_20.err"
rc=$?; cat "$OUT/bench_20.json"; [ $rc -eq 0 ] || { tail -n 20 "$OUT/bench_20.err"; exit $rc; }
timeout -k 10 240 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
rc=$?; cat "$OUT/bench_default.json"; [ $rc -eq 0 ] || { tail -n 20 "$OUT/bench_default.err"; exit $rc; }
(cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/ktrace" -- python3 "$OLDPWD/bench.py" --no-cpu-baseline > "$OLDPWD/$OUT/ktrace.log" 2>&1)
rc=$?; echo "kernel trace rc=$rc"; find "$OUT/ktrace" -name "*kernel_stats.csv" | head -n 1 | xargs -r head -n 12
exit $rc
