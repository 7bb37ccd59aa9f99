"""Frozen oracle outputs (tests/golden/*.json, made by tests/golden/make_golden.py): the CPU
test pins the oracle against drift; the GPU test pins the HIP path against the same files
without needing anything but the repository.  These fixtures are oracle outputs, not
reference outputs (the reference has none and cannot run here)."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def check(fx, block, out, sec):
    assert sha(block) == fx["sha256"]["block"]
    for k in ("voxel", "face", "steps", "bgra8", "t", "hit_point", "rgba"):
        assert sha(out[k]) == fx["sha256"][k], k
    for k in ("shadow", "ao", "sec_steps"):
        assert sha(sec[k]) == fx["sha256"][k], k
    assert int((out["face"] >= 0).sum()) == fx["totals"]["hits"]
    for p in fx["pixels"][:64]:
        r, c = p["row"], p["col"]
        assert out["voxel"][r, c].tolist() == p["voxel"] and int(out["steps"][r, c]) == p["steps"]
        assert int(out["t"][r, c].view(np.uint32)) == p["t_bits"]
        assert out["rgba"][r, c].view(np.uint32).tolist() == p["rgba_bits"]
        assert int(sec["ao"][r, c]) == p["ao"] and int(sec["shadow"][r, c]) == p["shadow"]


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-5] for p in FIXTURES])
def test_oracle_matches_golden(path):
    fx = json.load(open(path))
    dtype = np.dtype(fx["voxel_dtype"])
    block = O.terrain_fill(fx["grid_n"], fx["window_min"], dtype)
    cam = np.array(fx["cam_col_major_bits"], np.uint32).view(np.float32)
    world = O.World(block)
    out = O.trace(world, cam, fx["width"], fx["height"])
    sec = O.trace_secondary(world, cam, fx["width"], fx["height"])
    check(fx, block, out, sec)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-5] for p in FIXTURES])
def test_gpu_matches_golden(path):
    from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, VOXEL_REF_U16U16
    fx = json.load(open(path))
    fmt = VOXEL_PACKED_U8 if fx["voxel_dtype"] == "uint8" else VOXEL_REF_U16U16
    cam = np.array(fx["cam_col_major_bits"], np.uint32).view(np.float32)
    with GraphicsCore(grid_n=fx["grid_n"], voxel_format=fmt) as g:
        g.generate_window(fx["window_min"])
        g.reset_distance_field()
        block = g.download_world()
        out = g.trace(cam, fx["width"], fx["height"])
        sec = g.trace_secondary(cam, fx["width"], fx["height"])
    check(fx, block, out, sec)


FULLSIZE = sorted(glob.glob(os.path.join(HERE, "golden", "fullsize", "headline_*.json")))


@pytest.mark.gpu
@pytest.mark.parametrize("path", FULLSIZE, ids=[os.path.basename(p)[:-5] for p in FULLSIZE])
def test_gpu_matches_fullsize_golden(path):
    """The benchmark frame (made by tests/golden/make_golden_fullsize.py: the oracle's trace of the 1024^3
    world): every plane of the HIP path, primary and secondary, has the oracle's SHA-256."""
    from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8
    fx = json.load(open(path))
    cam = np.array(fx["cam_col_major_bits"], np.uint32).view(np.float32)
    with GraphicsCore(grid_n=fx["grid_n"], voxel_format=VOXEL_PACKED_U8) as g:
        g.generate_window(fx["window_min"])
        g.reset_distance_field()
        assert sha(g.download_world()) == fx["sha256"]["block"]
        out = g.trace(cam, fx["width"], fx["height"])
        sec = g.trace_secondary(cam, fx["width"], fx["height"])
    for k in ("voxel", "face", "steps", "bgra8", "t", "hit_point", "rgba"):
        assert sha(out[k]) == fx["sha256"][k], k
    for k in ("shadow", "ao", "sec_steps"):
        assert sha(sec[k]) == fx["sha256"][k], k
    assert int(out["steps"].astype(np.int64).sum()) == fx["totals"]["steps"]
    for p in fx["pixels"]:
        r, c = p["row"], p["col"]
        assert out["voxel"][r, c].tolist() == p["voxel"] and int(out["face"][r, c]) == p["face"]
        assert int(out["t"][r, c].view(np.uint32)) == p["t_bits"] and int(sec["sec_steps"][r, c]) == p["sec_steps"]


def test_flythrough_poses_match_the_committed_fixture():
    """The 64 x 16 f32 camera transforms of the pose-sharded flythrough, bit for bit."""
    from raytracing_amd import scenes
    fx = json.load(open(os.path.join(HERE, "golden", "fullsize", "flythrough_poses_64x16_f32.json")))
    want = np.array(fx["cam_col_major_bits"], np.uint32)
    got = np.stack([np.asarray(scenes.cam16(p), np.float32) for p in scenes.flythrough_poses(64)]).view(np.uint32)
    assert want.shape == (64, 16) and np.array_equal(got, want)
