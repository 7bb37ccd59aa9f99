"""Cross-check of the C oracle against the independent numpy restatement
(tests/pyref.py) on seeded random inputs: rays through random sparse grids
(u8 and u16, with and without a non-zero distance image), and the terrain
noise on random points.  Bit-exact."""
import numpy as np
import pytest

import oracle_lib as O
import pyref

f32 = np.float32


@pytest.mark.parametrize("seed,n,dtype,with_dist", [
    (0, 32, np.uint16, False), (1, 64, np.uint8, False), (2, 40, np.uint16, False),
    (3, 32, np.uint16, True), (4, 160, np.uint16, False)])
def test_random_rays(seed, n, dtype, with_dist):
    rng = np.random.default_rng(seed)
    block = ((rng.random((n, n, n)) < 0.01) * rng.integers(1, 200, (n, n, n))).astype(dtype)
    dist = rng.integers(0, 3, (n, n, n)).astype(np.uint16) if with_dist else None
    world = O.World(block, dist)
    for _ in range(60):
        ro = rng.uniform(-40, 40, 3).astype(f32)
        rd = rng.normal(size=3).astype(f32)
        rd /= np.linalg.norm(rd)
        if rng.random() < 0.15:
            rd[rng.integers(0, 3)] = 0.0      # degenerate component
        if rng.random() < 0.15:
            ro[rng.integers(0, 3)] = np.round(ro[rng.integers(0, 3)])  # integer coordinate
        a = O.trace_ray(world, ro, rd)
        b = pyref.trace_ray(block, dist, ro, rd, world.max_steps)
        for k in ("hit", "capped", "voxel", "face", "steps", "block_id"):
            assert a[k] == b[k], (k, a[k], b[k], ro, rd)
        assert a["t"].tobytes() == b["t"].tobytes()
        assert a["hit_point"].tobytes() == b["hit_point"].tobytes()


def test_snoise_matches_pyref():
    rng = np.random.default_rng(7)
    pts = np.concatenate([rng.uniform(-40, 40, (300, 3)), rng.integers(-2048, 2048, (300, 3)) * 0.015]).astype(f32)
    for p in pts:
        a = O.snoise(*p)
        b = pyref.snoise(p)
        assert a.tobytes() == b.tobytes(), (p, a, b)
    vals = np.array([O.snoise(*p) for p in pts])
    assert np.abs(vals).max() <= 1.0 and vals.std() > 0.2  # looks like noise in [-1, 1]


def test_terrain_fill_matches_pyref_samples():
    n = 64
    wmin = (-48, -16, -16)
    blk = O.terrain_fill(n, wmin, np.uint16)
    rng = np.random.default_rng(3)
    for _ in range(300):
        p = [int(rng.integers(wmin[k], wmin[k] + n)) for k in range(3)]
        p[1] = int(rng.integers(-8, 8))  # stay near the surface band |y| <= 6
        want = pyref.terrain_block(*p)
        assert blk[p[2] % n, p[1] % n, p[0] % n] == want
    # the terrain is a height field around y = 0: everything above y = 6 is air, below -6 solid
    for y in range(wmin[1], wmin[1] + n):
        layer = blk[:, y % n, :]
        if y > 6:
            assert not layer.any()
        if y < -6:
            assert layer.all()


def test_terrain_chunk_equals_fill():
    n = 64
    wmin = (-16, -16, -16)
    full = O.terrain_fill(n, wmin, np.uint8)
    # the window [-16, 48) is not chunk aligned: fill a 96^3 image by chunks over [-32, 64) and compare the overlap
    blk2 = np.zeros((96, 96, 96), np.uint8)
    for cz in (-1, 0, 1):
        for cy in (-1, 0, 1):
            for cx in (-1, 0, 1):
                O.terrain_chunk(blk2, cx, cy, cz)
    idx = (np.arange(-16, 48) % 96)
    sub = blk2[np.ix_(idx, idx, idx)]
    idn = (np.arange(-16, 48) % n)
    assert np.array_equal(sub, full[np.ix_(idn, idn, idn)])
