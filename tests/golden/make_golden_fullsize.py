#!/usr/bin/env python3
"""Freezes the CPU oracle's outputs for the benchmark frame (1920x1080 over the 1024^3 u8 terrain) as
tests/golden/fullsize/headline_1080p_n1024.json.  The oracle needs minutes to FILL a 1024^3 world, so the
world comes from the HIP terrain generator (bit-identical to the oracle's: tests/test_gpu_world.py checks that
at sizes the oracle fills quickly, and the fixture pins its SHA-256 here); every traced output is the
oracle's.  Needs a GPU for that one step:

  python tests/golden/make_golden_fullsize.py        (on the GPU box; about 10 s)

Like the small fixtures these are oracle outputs, not reference outputs."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, scenes  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    n, w, h = 1024, 1920, 1080
    p = scenes.pose_headline()
    wmin = scenes.window_min(p.get_position(), n)
    with GraphicsCore(grid_n=n, voxel_format=VOXEL_PACKED_U8) as g:
        g.generate_window(wmin)
        block = g.download_world()
    cam = scenes.cam16(p)
    world = O.World(block)
    out = O.trace(world, cam, w, h, nthreads=O.host_threads())
    sec = O.trace_secondary(world, cam, w, h, nthreads=O.host_threads())
    rng = np.random.default_rng(3)
    px = []
    for r, c in zip(rng.integers(0, h, 256).tolist(), rng.integers(0, w, 256).tolist()):
        px.append({"row": r, "col": c, "voxel": out["voxel"][r, c].tolist(), "face": int(out["face"][r, c]),
                   "steps": int(out["steps"][r, c]), "bgra8": out["bgra8"][r, c].tolist(),
                   "t_bits": int(out["t"][r, c].view(np.uint32)),
                   "hit_point_bits": out["hit_point"][r, c].view(np.uint32).tolist(),
                   "shadow": int(sec["shadow"][r, c]), "ao": int(sec["ao"][r, c]), "sec_steps": int(sec["sec_steps"][r, c])})
    fx = {"name": "headline_1080p_n1024", "grid_n": n, "voxel_dtype": "uint8", "window_min": [int(v) for v in wmin],
          "cam_col_major_bits": cam.view(np.uint32).tolist(), "width": w, "height": h,
          "sha256": {k: sha(out[k]) for k in ("voxel", "face", "steps", "bgra8", "t", "hit_point", "rgba")}
                    | {k: sha(sec[k]) for k in ("shadow", "ao", "sec_steps")} | {"block": sha(block)},
          "totals": {"hits": int((out["face"] >= 0).sum()), "steps": int(out["steps"].astype(np.int64).sum()),
                     "solid_voxels": int(np.count_nonzero(block)), "secondary_steps": int(sec["sec_steps"].astype(np.int64).sum())},
          "pixels": px}
    os.makedirs(os.path.join(HERE, "fullsize"), exist_ok=True)
    path = os.path.join(HERE, "fullsize", fx["name"] + ".json")
    with open(path, "w") as f:
        json.dump(fx, f, indent=0)
    print(path, fx["totals"])


if __name__ == "__main__":
    main()
