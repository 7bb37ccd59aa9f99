#!/bin/bash
# The records a round commits under profiles/: the default bench line, its kernel-trace statistics and the other configurations.
#   gpurun -- bash tools/final_records.sh [tag]   -> gpurun_out/<tag>/{bench_line.json,kernel_stats.csv,other_configs.json}
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"
rc=$?; [ $rc -eq 0 ] || { tail -n 20 "$OUT/bench_line.err"; exit $rc; }
python3 -c "import json,sys; d=json.load(open('$OUT/bench_line.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['frac_datasheet'], r['valu_count_stale'])"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/ktrace" -- python3 "$OLDPWD/bench.py" --no-cpu-baseline > "$OLDPWD/$OUT/ktrace.log" 2>&1)
rc=$?; echo "kernel trace rc=$rc"; [ $rc -eq 0 ] || exit $rc
find "$OUT/ktrace" -name "*kernel_stats.csv" | head -n 1 | xargs -r -I{} cp {} "$OUT/kernel_stats.csv"
head -n 4 "$OUT/kernel_stats.csv"
timeout -k 10 400 python -u tools/bench_configs.py > "$OUT/other_configs.json" 2> "$OUT/other_configs.err"
rc=$?; cat "$OUT/other_configs.json"; exit $rc
