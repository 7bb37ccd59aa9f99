#!/usr/bin/env python3
"""Turns the counter passes of tools/profile_r03.sh into the record bench.py prices (traffic.json, written into the
directory given) and prints a readable summary.  Launches of k_trace_skip are told apart by their place in the run:
the first (set-up) and the last 16 run one at a time with the measured dispatch order ("alone"), the 201 between them
overlap on two streams with rows in plain order ("overlapped")."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd.kernel_sources import blob_hashes  # noqa: E402

out_dir = sys.argv[1]
kernel = "k_trace_skip"
modes = {"alone": defaultdict(list), "overlapped": defaultdict(list)}
for d in sorted(glob.glob(os.path.join(out_dir, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    per_dispatch = defaultdict(dict)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                per_dispatch[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    order = sorted(per_dispatch)
    if len(order) != 218:
        print(f"# {d}: {len(order)} launches of {kernel} (218 expected): pass ignored")
        continue
    for i, did in enumerate(order):
        mode = "alone" if i == 0 or i >= 202 else "overlapped"
        for name, v in per_dispatch[did].items():
            modes[mode][name].append(v)

W, H = 1920, 1080
record = {"_note": "Counters per launch of k_trace_skip<u8, pow2> on the bench.py workload (1920x1080 over 1024^3), rocprofv3 --pmc passes of "
                   "`bench.py --warmup 1 --steps 200 --probe 16 --settle-ms 0 --no-cpu-baseline --no-literal` (tools/profile_r03.sh, one run per pass), "
                   "means over the 17 launches that run one at a time (measured dispatch order) and over the 201 that overlap on two "
                   "streams (rows in plain order, no dispatch-order bookkeeping).  FETCH_SIZE and WRITE_SIZE are KiB; FETCH_SIZE is doubled "
                   "(gfx950 tallies 128-byte fills as 64 bytes; TCC_EA0_RDREQ_32B = 0), WRITE_SIZE taken as reported.  kernel_sources: git "
                   "blob hashes of the files the kernel was compiled from; bench.py uses the record only for a tree with the same hashes.",
          "kernel_sources": blob_hashes()}
print(f"bench.py workload (1920x1080 primary rays over 1024^3 u8, pose B at y=40.3), {kernel}<u8, pow2>: rocprofv3 counter passes,")
print("one run per pass (tools/profile_r03.sh), mean per launch.  Kernel sources (git blob hashes):")
for k, v in record["kernel_sources"].items():
    print(f"  {v}  {k}")
for mode, key in (("alone", "skip"), ("overlapped", "skip_plain_order")):
    c = {name: sum(v) / len(v) for name, v in modes[mode].items()}
    n = {name: len(v) for name, v in modes[mode].items()}
    print(f"\n== launches {mode} ==")
    for name in sorted(c):
        print(f"  {name:34s} n={n[name]:4d} mean={c[name]:.6g}")
    if not c:
        continue
    fetch = c["FETCH_SIZE"] * 1024 * 2
    write = c["WRITE_SIZE"] * 1024
    waves = c["SQ_WAVES"]
    derived = {
        "fetch_bytes": fetch, "write_bytes": write, "hbm_bytes": fetch + write,
        "write_over_frame": write / (W * H * 4),
        "valu_instructions": c["SQ_INSTS_VALU"], "salu_instructions": c["SQ_INSTS_SALU"],
        "valu_per_wave": c["SQ_INSTS_VALU"] / waves, "salu_per_wave": c["SQ_INSTS_SALU"] / waves,
        "lanes_active_per_vector_instruction": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]),
        "issue_wait_fraction_of_wave_cycles": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
        "memory_wait_fraction_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
        "l1_hit_rate": 1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"],
        "l2_hit_rate": c["TCC_HIT_sum"] / c["TCC_REQ_sum"],
        "l1_accesses": c["TCP_TOTAL_CACHE_ACCESSES_sum"], "l2_requests": c["TCC_REQ_sum"],
        "vmem_read_instructions": c["SQ_INSTS_VMEM_RD"], "vmem_write_instructions": c["SQ_INSTS_VMEM_WR"],
        "waves": waves,
    }
    for name, v in derived.items():
        print(f"  derived {name:42s} {v:.6g}")
    record[key] = fetch + write
    record[key + "_valu_instructions"] = c["SQ_INSTS_VALU"]
    record[key + "_salu_instructions"] = c["SQ_INSTS_SALU"]
    record[key + "_counters"] = derived
json.dump(record, open(os.path.join(out_dir, "traffic.json"), "w"), indent=1)
