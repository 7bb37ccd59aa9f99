set -u
mkdir -p gpurun_out/pose; export TMPDIR=/tmp
for pitch in 1.2 -0.35 -1.4; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pose/p$pitch -- python3 tools/pose_probe.py $pitch > gpurun_out/pose/p$pitch.log 2>&1 || exit 1
  grep -E "pitch|kernel us" gpurun_out/pose/p$pitch.log
done
