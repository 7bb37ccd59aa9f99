#!/usr/bin/env python3
"""Static instruction budget of one trace kernel from the compiler's ISA listing.

  python tools/isa_budget.py [--kernel k_trace_skipIhLi1E] [--extra=-DX] [--cuts 124,196,...]

Compiles raytracing_amd/csrc/vrt_trace.hip for gfx950 with -save-temps into a directory under /tmp (never the
tree), cuts the kernel's listing into basic blocks and prints, per block, the vector / scalar / memory
instruction counts, whether the block belongs to the traversal loop (the compiler marks those "in Loop"), and
the label comment the compiler left (inlined functions end in a block named after them).  With --cuts the
listing is summed over the line ranges between the given line numbers of the kernel's own listing instead
(written to /tmp/vrt_isa/kernel.s for reading).  Straight-line code before and after the loop runs once per
wavefront, so its static count is its dynamic count; the loop's share of a measured SQ_INSTS_VALU total is
what remains.
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/vrt_isa"


def classify(tok):
    if tok.startswith("v_"):
        return "valu"
    if tok.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if tok.startswith("ds_"):
        return "lds"
    if tok.startswith("s_load") or tok.startswith("s_buffer_load"):
        return "smem"
    if tok.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if tok.startswith("s_"):
        return "salu"
    return None


# Issue cost of a vector instruction in shader cycles per wave64 instruction per SIMD, as tools/op_rates measured it on the
# MI355X with eight wavefronts per SIMD: a few 32-bit operations issue at about 2.3 cycles as long as no operand comes from a
# scalar register (VCC excepted), transcendentals at 8.2, everything else at about 4.15.
FAST = {"v_fma_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32"}
TRANS = ("v_rcp_", "v_sqrt_", "v_rsq_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def cycles(line):
    parts = line.strip().replace("\t", " ").split(None, 1)
    if not parts or not parts[0].startswith("v_"):
        if parts and parts[0] == "s_nop":
            return 1.0 + float(parts[1].split()[0])
        return 0.0
    op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", parts[0])
    ops = parts[1] if len(parts) > 1 else ""
    if op.startswith(TRANS):
        return 8.2
    if op == "v_cndmask_b32" and parts[0].endswith("_e32"):
        return 2.3   # (the VCC form, measured behind the compare that writes VCC)
    scalar_src = re.search(r"(?<![\w.])s(\d+|\[\d+:\d+\])", ops.split(",", 1)[1] if "," in ops else "") is not None
    if op in FAST and not scalar_src and not parts[0].endswith(("_sdwa", "_dpp")):
        return 2.3
    return 4.15


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_trace_skipIhLi1E", help="substring of the mangled kernel name")
    ap.add_argument("--extra", default="", help="extra compiler flags, e.g. --extra=-DVRT_DIAG_ITERS")
    ap.add_argument("--cuts", default="", help="comma-separated line numbers of kernel.s: sum per range instead of per block")
    ap.add_argument("--no-build", action="store_true", help="reuse the listing of the last run")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, "raytracing_amd", "csrc", "vrt_trace.hip")
    listing = os.path.join(OUT, "vrt_trace-hip-amdgcn-amd-amdhsa-gfx950.s")
    if not args.no_build:
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
               "-fno-fast-math", "-c", src, "-o", os.path.join(OUT, "vrt_trace.o"), "-save-temps=obj"] + args.extra.split()
        subprocess.run(cmd, check=True, cwd=OUT)
    lines = open(listing).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN3vrt\w*" + re.escape(args.kernel) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    open(os.path.join(OUT, "kernel.s"), "w").write("\n".join(body) + "\n")
    meta = {}
    for l in lines[end:end + 120]:
        m = re.match(r"\s*;\s*(NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize):\s*(\d+)", l)
        if m:
            meta.setdefault(m.group(1), m.group(2))
    print(f"{lines[start].split(':')[0]}  ({len(body)} lines)  " + "  ".join(f"{k}={v}" for k, v in meta.items()))

    def tally(chunk):
        c = {"valu": 0, "salu": 0, "vmem": 0, "lds": 0, "smem": 0, "wait": 0, "cyc": 0.0}
        for l in chunk:
            t = l.strip().split(" ")[0].split("\t")[0]
            k = classify(t)
            if k:
                c[k] += 1
            c["cyc"] += cycles(l)
        return c

    if args.cuts:
        cuts = [1] + [int(x) for x in args.cuts.split(",")] + [len(body) + 1]
        for a, b in zip(cuts, cuts[1:]):
            c = tally(body[a - 1:b - 1])
            print(f"  lines {a:5d}-{b - 1:5d}  valu {c['valu']:4d} ({c['cyc']:6.0f} cycles)  salu {c['salu']:4d}  vmem {c['vmem']:3d}  smem {c['smem']:3d}")
        return
    blocks, cur = [], {"label": "entry", "note": "", "loop": False, "first": 1, "lines": []}
    for i, l in enumerate(body, 1):
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "note": (m.group(2) or "").strip("; "), "loop": False, "first": i, "lines": []}
            continue
        if "in Loop:" in l or "Inner Loop Header" in l:
            cur["loop"] = True
        m = re.match(r"^; %bb\.(\d+):\s*(;.*)?$", l)
        if m:
            blocks.append(cur)
            cur = {"label": "bb." + m.group(1), "note": (m.group(2) or "").strip("; "), "loop": False, "first": i, "lines": []}
            if "in Loop:" in l:
                cur["loop"] = True
            continue
        cur["lines"].append(l)
    blocks.append(cur)
    tot = {"loop": tally([]), "rest": tally([])}
    for b in blocks:
        c = tally(b["lines"])
        if not any(v for k, v in c.items() if k != "cyc"):
            continue
        where = "loop" if b["loop"] else "rest"
        for k in c:
            tot[where][k] += c[k]
        note = re.sub(r"in Loop: Header=\S+ Depth=\d+", "", b["note"]).strip()
        print(f"  {b['first']:5d} {b['label']:10s} {'L' if b['loop'] else ' '} valu {c['valu']:3d} ({c['cyc']:4.0f} cyc) salu {c['salu']:3d} vmem {c['vmem']:2d}  {note[:90]}")
    for k, c in tot.items():
        print(f"  total {k}: valu {c['valu']} ({c['cyc']:.0f} cycles) salu {c['salu']} vmem {c['vmem']} smem {c['smem']}")


if __name__ == "__main__":
    sys.exit(main())
