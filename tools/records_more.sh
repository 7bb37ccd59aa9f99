#!/bin/bash
# The round's other records in one GPU-box session: chunk-crossing costs, the inputs of the multi-GPU prediction, the
# single-rank rehearsal of the gather, counters of the secondary-ray kernels.
#   gpurun -- bash tools/records_more.sh [tag]   -> gpurun_out/<tag>/...
TAG=${1:-more}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 200 python -u tools/crossing_probe.py 160 480 1056 > "$OUT/chunk_crossing.txt" 2> "$OUT/chunk_crossing.err"
rc=$?; cat "$OUT/chunk_crossing.txt"; [ $rc -eq 0 ] || { tail -n 5 "$OUT/chunk_crossing.err"; exit $rc; }
timeout -k 10 300 python -u tools/scale_inputs.py > "$OUT/scale_inputs.json" 2> "$OUT/scale_inputs.err"
rc=$?; cut -c1-600 "$OUT/scale_inputs.json"; [ $rc -eq 0 ] || { tail -n 5 "$OUT/scale_inputs.err"; exit $rc; }
for px in bgr bgra; do
  timeout -k 10 200 python bench.py --force-gather --gather-pixels $px --no-cpu-baseline --no-literal > "$OUT/force_gather_$px.json" 2> "$OUT/force_gather_$px.err"
  rc=$?; [ $rc -eq 0 ] || { tail -n 5 "$OUT/force_gather_$px.err"; exit $rc; }
  python3 -c "import json; d=json.load(open('$OUT/force_gather_$px.json')); print('force-gather $px: ms_per_step', d['ms_per_step'], 'host_enqueue', d['host_enqueue_ms_per_step'], 'value', d['value'], d['gather'])"
done
timeout -k 10 200 python bench.py --shard strips --force-gather --no-cpu-baseline --no-literal > "$OUT/force_gather_strips.json" 2> "$OUT/force_gather_strips.err"
rc=$?; [ $rc -eq 0 ] || { tail -n 5 "$OUT/force_gather_strips.err"; exit $rc; }
python3 -c "import json; d=json.load(open('$OUT/force_gather_strips.json')); print('force-gather strips: ms_per_step', d['ms_per_step'], d['gather'])"
VARIANTS=binned bash tools/profile_secondary_pmc.sh > "$OUT/secondary_pmc.log" 2>&1
rc=$?; tail -n 40 "$OUT/secondary_pmc.log"; exit $rc
