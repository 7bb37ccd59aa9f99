#!/bin/bash
# A/B of several builds of the library on the secondary-ray workload (tools/ab_secondary.py per build, which also checks that the
# binned kernel's planes equal the per-pixel kernel's).  usage: bash tools/ab_sec_libs.sh rounds lib1.so lib2.so ...
set -u; mkdir -p gpurun_out
rounds=$1; shift
for lib in "$@"; do
  echo "$lib"
  VRT_LIB=$PWD/$lib timeout -k 10 200 python tools/ab_secondary.py "$rounds" 2>>gpurun_out/ab_sec_err.log || { tail -5 gpurun_out/ab_sec_err.log; exit 1; }
done
