#!/usr/bin/env python3
"""Freezes outputs of the CPU oracle as regression fixtures (tests/golden/*.json).

The reference ships no golden vectors and cannot run here, so these are NOT reference
outputs: they pin the oracle (and through it the HIP kernels) against drift between rounds.
Re-run only when the canonical semantics change on purpose, and say so in the commit.

  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
from raytracing_amd import scenes  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def freeze(name, n, dtype, player, w, h, samples=256, seed=1):
    wmin = scenes.window_min(player.get_position(), n)
    block = O.terrain_fill(n, wmin, dtype)
    cam = scenes.cam16(player)
    out = O.trace(O.World(block), cam, w, h)
    sec = O.trace_secondary(O.World(block), cam, w, h)
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, h, samples)
    cols = rng.integers(0, w, samples)
    px = []
    for r, c in zip(rows.tolist(), cols.tolist()):
        px.append({"row": r, "col": c,
                   "voxel": out["voxel"][r, c].tolist(), "face": int(out["face"][r, c]),
                   "steps": int(out["steps"][r, c]), "bgra8": out["bgra8"][r, c].tolist(),
                   "t_bits": int(out["t"][r, c].view(np.uint32)),
                   "hit_point_bits": out["hit_point"][r, c].view(np.uint32).tolist(),
                   "rgba_bits": out["rgba"][r, c].view(np.uint32).tolist(),
                   "shadow": int(sec["shadow"][r, c]), "ao": int(sec["ao"][r, c]),
                   "sec_steps": int(sec["sec_steps"][r, c])})
    fx = {"name": name, "grid_n": n, "voxel_dtype": np.dtype(dtype).name, "window_min": list(wmin),
          "cam_col_major_bits": cam.view(np.uint32).tolist(), "width": w, "height": h,
          "sha256": {"block": sha(block), "voxel": sha(out["voxel"]), "face": sha(out["face"]),
                     "steps": sha(out["steps"]), "bgra8": sha(out["bgra8"]), "t": sha(out["t"]),
                     "hit_point": sha(out["hit_point"]), "rgba": sha(out["rgba"]),
                     "shadow": sha(sec["shadow"]), "ao": sha(sec["ao"]), "sec_steps": sha(sec["sec_steps"])},
          "totals": {"hits": int((out["face"] >= 0).sum()), "steps": int(out["steps"].astype(np.int64).sum()),
                     "solid_voxels": int(block.astype(bool).sum())},
          "pixels": px}
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(fx, f, indent=0)
    print(name, fx["totals"])


def freeze_flythrough_poses():
    """The 64 camera transforms of the 2048^3 flythrough (tests/golden/fullsize/flythrough_poses_64x16_f32.json)."""
    cams = np.stack([np.asarray(scenes.cam16(p), np.float32) for p in scenes.flythrough_poses(64)])
    note = ("64 camera transforms of the flythrough (circle of radius 300 voxels at y = 60.3, yaw_k = 2*pi*k/64 + 0.013, "
            "pitch -0.3), column major, as uint32 bit patterns of the 16 f32 each; made from "
            "raytracing_amd.scenes.flythrough_poses(64) by tests/golden/make_golden.py")
    with open(os.path.join(HERE, "fullsize", "flythrough_poses_64x16_f32.json"), "w") as f:
        json.dump({"note": note, "cam_col_major_bits": cams.view(np.uint32).tolist()}, f)


if __name__ == "__main__":
    freeze_flythrough_poses()
    freeze("config1_pose_b_640x480_n160", 160, np.uint16, scenes.pose_b(), 640, 480)
    freeze("config1_spawn_640x480_n160", 160, np.uint16, scenes.pose_spawn(), 640, 480)
    freeze("config2_pose_b_1920x1080_n256", 256, np.uint16, scenes.pose_b(), 1920, 1080)
