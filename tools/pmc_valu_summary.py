#!/usr/bin/env python3
"""Per-launch instruction counters of the trace kernels from one rocprofv3 --pmc pass (tools/pmc_valu.sh): launches
are grouped by their SQ_INSTS_VALU count (overlapped launches run rows in plain order and skip the dispatch-order
bookkeeping, so they issue fewer instructions than launches that run one at a time)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
rows = defaultdict(dict)   # (kernel, dispatch id) -> counter -> value
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace" not in r["Kernel_Name"]:
            continue
        rows[(r["Kernel_Name"], int(r["Dispatch_Id"]))][r["Counter_Name"]] = float(r["Counter_Value"])
by_kernel = defaultdict(list)
for (k, d), c in sorted(rows.items(), key=lambda kv: kv[0][1]):
    by_kernel[k].append(c)
for k, launches in by_kernel.items():
    print(k, f"{len(launches)} launches")
    groups = defaultdict(list)
    for c in launches:
        groups[round(c.get("SQ_INSTS_VALU", 0) / 2e5)].append(c)   # bins of 0.2 M instructions
    for g, cs in sorted(groups.items()):
        n = len(cs)
        mean = {name: sum(c.get(name, 0.0) for c in cs) / n for name in cs[0]}
        waves = mean.get("SQ_WAVES", 0) or 1
        util = mean.get("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * mean["SQ_ACTIVE_INST_VALU"]) if mean.get("SQ_ACTIVE_INST_VALU") else float("nan")
        print(f"  {n:4d} launches: SQ_INSTS_VALU {mean.get('SQ_INSTS_VALU', 0):.0f} ({mean.get('SQ_INSTS_VALU', 0) / waves:.1f} per wave)  "
              f"SQ_INSTS_SALU {mean.get('SQ_INSTS_SALU', 0):.0f} ({mean.get('SQ_INSTS_SALU', 0) / waves:.1f} per wave)  "
              f"SQ_WAVES {waves:.0f}  lanes active per vector instruction {util:.3f}")
