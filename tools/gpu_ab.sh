#!/bin/bash
# One GPU-box session for a kernel change: the GPU test suite, then alternating bench runs of the working tree's
# library and another build of it (tools/build_variant.sh) on the same box.
#   gpurun --timeout 900 -- bash tools/gpu_ab.sh [tag] [other.so] [rounds] [pytest selection]
TAG=${1:-ab}; OTHER=${2:-raytracing_amd/ab_prev.so}; ROUNDS=${3:-3}; SEL=${4:-tests}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 840 python -u -m pytest $SEL -m gpu -x -q --timeout 600 --timeout-method thread > "$OUT/tests.log" 2>&1
rc=$?; tail -n 15 "$OUT/tests.log"; [ $rc -eq 0 ] || exit $rc
bash tools/ab_libs.sh "$OTHER" "$ROUNDS" --no-literal 2>&1 | tee "$OUT/ab.txt"
