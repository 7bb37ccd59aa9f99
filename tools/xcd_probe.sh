#!/bin/bash
# XCD stripes (a build with EXTRA=-DVRT_XCD_STRIPES, raytracing_amd/ab_xcd.so from tools/build_flags.sh) against the shipped
# mapping: bench step and kernel time, alternating, and the L2 fill traffic of the trace kernel (FETCH_SIZE, KiB per launch).
set -u
export TMPDIR=/tmp
bash tools/ab_many.sh 2 raytracing_amd/libvrt.so raytracing_amd/ab_xcd.so -- --overlap 1 || exit 1
bash tools/ab_many.sh 1 raytracing_amd/libvrt.so raytracing_amd/ab_xcd.so || exit 1
for lib in libvrt ab_xcd; do
  export VRT_LIB=$PWD/raytracing_amd/$lib.so
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/xcd/$lib -- python3 bench.py --warmup 1 --steps 10 --no-cpu-baseline --no-literal > gpurun_out/xcd_$lib.log 2>&1 || exit 1
  python3 tools/summarize_pmc.py gpurun_out/xcd/$lib k_trace_skip | sed "s|^|$lib |"
done
