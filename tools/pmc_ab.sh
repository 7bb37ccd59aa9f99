set -u
mkdir -p gpurun_out/prof2; export TMPDIR=/tmp
for v in 0 1; do
  VRT_SKIP_VARIANT=$v timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/prof2/v$v -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof2/v$v.log 2>&1 || exit 1
done
