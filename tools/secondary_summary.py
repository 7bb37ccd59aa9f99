#!/usr/bin/env python3
"""Puts the counter summaries that tools/profile_secondary_pmc.sh left under gpurun_out/prof_sec/<variant>/ into one
text with the derived figures (lane utilisation, cache hit rates).  usage: secondary_summary.py "<note>" > profiles/..."""
import glob, re, sys

def parse(p):
    d = {}
    for l in open(p):
        m = re.match(r"\s+(\S+)\s+n=\s*\d+ mean=(\S+)", l)
        if m: d[m.group(1)] = float(m.group(2))
    return d

print("Secondary-ray workload (1920x1080 primary rays + one shadow and four ambient-occlusion rays per hit, 1024^3 u8 terrain, pose B at y=40.3)")
print("rocprofv3 counter passes, one run per pass (tools/profile_secondary_pmc.sh), 8 launches each; kernel times from a separate")
print("--kernel-trace --stats run of 12 launches.  " + (sys.argv[1] if len(sys.argv) > 1 else ""))
print()
import json, os
# the primary kernel alone on this frame, one launch at a time: the counter record of the bench workload (same frame, same pose)
_c = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "traffic.json")))["skip_counters"]
PRIMARY_VALU = _c["valu_instructions"]
PRIMARY_THREAD = _c["lanes_active_per_vector_instruction"] * 64 * PRIMARY_VALU
for v in ("pixel", "binned"):
    path = f"gpurun_out/prof_sec/{v}/summary.txt"
    if not os.path.exists(path): continue
    d = parse(path)
    print(f"== VRT_SECONDARY={v}")
    print(open(path).read().rstrip())
    print(f"  derived: lane utilisation SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) = {d['SQ_THREAD_CYCLES_VALU'] / (64 * d['SQ_ACTIVE_INST_VALU']):.3f}")
    print(f"  derived: the secondary rays alone (primary kernel's counts subtracted): {(d['SQ_THREAD_CYCLES_VALU'] - PRIMARY_THREAD) / (64 * (d['SQ_ACTIVE_INST_VALU'] - PRIMARY_VALU)):.3f}")
    print(f"  derived: L2 hit rate TCC_HIT/TCC_REQ = {d['TCC_HIT_sum'] / d['TCC_REQ_sum']:.3f}; L1 hit rate 1 - TCP_TCC_READ_REQ/TCP_TOTAL_CACHE_ACCESSES = {1 - d['TCP_TCC_READ_REQ_sum'] / d['TCP_TOTAL_CACHE_ACCESSES_sum']:.3f}")
    if d.get("SQ_ACTIVE_INST_LDS", 0) > 1e4:
        print(f"  derived: SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = {d['SQ_LDS_BANK_CONFLICT'] / d['SQ_ACTIVE_INST_LDS']:.2f}; per wavefront: "
              f"{d['SQ_LDS_BANK_CONFLICT'] / d['SQ_WAVES']:.0f} conflict cycles, {d['SQ_INSTS_LDS'] / d['SQ_WAVES']:.0f} LDS instructions, "
              f"{d['SQ_WAVE_CYCLES'] / d['SQ_WAVES']:.0f} cycles of lifetime")
    print(f"  derived: vector instructions per wavefront {d['SQ_INSTS_VALU'] / d['SQ_WAVES']:.0f}")
    stats = sorted(glob.glob(f"gpurun_out/prof_sec/{v}/ktrace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
    for f in stats[-1:]:   # (the newest run's)
        for l in open(f):
            if "secondary" in l: print("  kernel_stats: " + l.strip())
    print()
