#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per dispatch, per kernel.
usage: summarize_pmc.py gpurun_out/prof [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if want and want not in k:
            continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, ctrs in acc.items():
    print(k)
    for c in sorted(ctrs):
        v = ctrs[c]
        print(f"  {c:34s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
