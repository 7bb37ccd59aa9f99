#!/bin/bash
# Vector instructions per wavefront of k_trace_skip by stage, measured: builds of the library that stop the kernel after
# ray generation (1), after the ray's set-up (2), after the traversal trips (3), after the ray's outputs (4) and the whole
# kernel (full), one rocprofv3 counter pass each over single launches of the bench frame (tools/stage_probe.py).
# The stage builds are made beforehand on the build host:  bash tools/stage_budget.sh build [rev]   (rev: a git revision whose
# sources carry the VRT_DIAG_STAGE switches; default the working tree) -> raytracing_amd/ab_stage{1,2,3,4,full}.so
# and measured on the GPU box:                            bash tools/stage_budget.sh run [tag] [prefix]  -> gpurun_out/<tag>/stages.txt
# (prefix: the libraries are raytracing_amd/<prefix>{1,2,3,4,full}.so, default ab_stage)
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
if [ "${1:-}" = build ]; then
  src=${2:-}
  for st in 1 2 3 4 full; do
    tmp=$(mktemp -d /tmp/vrt_stage.XXXXXX)
    if [ -n "$src" ]; then git -C "$root" archive "$src" raytracing_amd/csrc include | tar -x -C "$tmp"
    else mkdir -p "$tmp/raytracing_amd" && cp -r "$root/raytracing_amd/csrc" "$tmp/raytracing_amd/" && cp -r "$root/include" "$tmp/" && rm -f "$tmp"/raytracing_amd/csrc/*.o; fi
    extra=""; [ $st = full ] || extra="-DVRT_DIAG_STAGE=$st"
    make -s -j4 -C "$tmp/raytracing_amd/csrc" ../libvrt.so EXTRA="$extra" 2>&1 | grep -v "warning\|^ \|generated" | tail -3
    cp "$tmp/raytracing_amd/libvrt.so" "$root/raytracing_amd/ab_stage$st.so"
    rm -rf "$tmp"
  done
  ls -la "$root"/raytracing_amd/ab_stage*.so
  exit 0
fi
TAG=${2:-stages}
PREFIX=${3:-ab_stage}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
: > "$OUT/stages.txt"
for st in 1 2 3 4 full; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d "$OUT/pmc_$st" -- \
      python3 tools/stage_probe.py "raytracing_amd/$PREFIX$st.so" > "$OUT/pmc_$st.log" 2>&1 || { echo "stage $st failed"; tail -n 5 "$OUT/pmc_$st.log"; exit 1; }
  python3 - "$OUT/pmc_$st" "$st" <<'PY' | tee -a "$OUT/stages.txt"
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace_skip" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print(f"stage {sys.argv[2]:>4}: launches {len(acc['SQ_WAVES'])}  SQ_INSTS_VALU {m['SQ_INSTS_VALU']:.0f} = {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:.1f} per wave   SQ_INSTS_SALU {m['SQ_INSTS_SALU'] / m['SQ_WAVES']:.1f} per wave")
PY
done
