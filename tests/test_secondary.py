"""Secondary rays (shadow + 4 ambient-occlusion rays per primary hit): an extension with no
counterpart in the reference, defined by the CPU oracle.  CPU part: the definition itself
(directions are unit, lie in the hemisphere of the face normal, are cosine distributed, and
the hash chain is the published pcg hash).  GPU part: bit-exact against the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from raytracing_amd import scenes


def test_pcg_chain_and_direction_properties():
    light = O.secondary_dir(0, 0, 3)
    v = np.array([0.4, 0.8, 0.45], np.float32)
    assert np.allclose(light, v / np.linalg.norm(v), atol=1e-7)
    rng = np.random.default_rng(0)
    dots = []
    for _ in range(400):
        px = int(rng.integers(0, 1920 * 1080))
        face = int(rng.integers(0, 6))
        normal = np.zeros(3)
        normal[face // 2] = 1.0 if face % 2 else -1.0
        for idx in range(1, 5):
            d = O.secondary_dir(px, idx, face).astype(np.float64)
            assert abs(np.linalg.norm(d) - 1.0) < 1e-6
            assert d @ normal >= 0.0
            dots.append(d @ normal)
    # cosine-weighted hemisphere: E[cos] = 2/3
    assert abs(np.mean(dots) - 2.0 / 3.0) < 0.03
    # the four AO rays of a pixel differ, and pixels differ
    a = [tuple(O.secondary_dir(7, i, 3)) for i in range(1, 5)]
    assert len(set(a)) == 4 and tuple(O.secondary_dir(8, 1, 3)) != a[0]


def test_oracle_secondary_on_a_flat_floor():
    # flat floor at y = 10, nothing else: no secondary ray can hit -> all zero; with a ceiling,
    # every shadow ray (the light is above) hits
    n = 64
    block = np.zeros((n, n, n), np.uint8)
    block[:, 10, :] = 1
    from raytracing_amd import Player
    cam = scenes.cam16(Player.at((16.5, 20.5, 16.5), 0.3, -0.8))
    out = O.trace_secondary(O.World(block), cam, 64, 48)
    prim = O.trace(O.World(block), cam, 64, 48, want=("face",))
    hit = prim["face"] >= 0
    assert hit.any() and not out["shadow"].any() and not out["ao"].any()
    assert (out["sec_steps"][hit] > 0).all() and (out["sec_steps"][~hit] == 0).all()
    block[:, 30, :] = 2
    out = O.trace_secondary(O.World(block), cam, 64, 48)
    floor_hit = prim["face"] == 3
    # (shadow rays that leave the loaded window before they reach the ceiling are misses)
    assert floor_hit.any() and out["shadow"][floor_hit].mean() > 0.5 and not out["shadow"][~hit].any()


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["binned", "pixel"])  # rays staged and binned in LDS (the default), or on their pixel's lane
@pytest.mark.parametrize("n,dtype", [(160, np.uint16), (256, np.uint8)])
def test_gpu_secondary_matches_oracle(n, dtype, kernel, monkeypatch):
    from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, VOXEL_REF_U16U16
    monkeypatch.setenv("VRT_SECONDARY", kernel)  # read when the context is created
    p = scenes.pose_b()
    block = O.terrain_fill(n, scenes.window_min(p.get_position(), n), dtype)
    cam = scenes.cam16(p)
    W, H = 480, 270
    want = O.trace_secondary(O.World(block), cam, W, H)
    fmt = VOXEL_PACKED_U8 if dtype == np.uint8 else VOXEL_REF_U16U16
    with GraphicsCore(grid_n=n, voxel_format=fmt) as g:
        g.upload_world(block)
        got = g.trace_secondary(cam, W, H)
        assert g.last_kernel_name() == "secondary"
        band = g.trace_secondary(cam, W, H, 100, 140)
    for k in ("shadow", "ao", "sec_steps"):
        assert np.array_equal(got[k], want[k]), k
        assert np.array_equal(band[k], want[k][100:140]), k
    assert want["shadow"].any() and want["ao"].any()


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["binned", "pixel"])
def test_gpu_secondary_on_random_grid_with_degenerate_rays(kernel, monkeypatch):
    from raytracing_amd import GraphicsCore, Player, VOXEL_PACKED_U8
    monkeypatch.setenv("VRT_SECONDARY", kernel)
    rng = np.random.default_rng(9)
    n = 64
    block = ((rng.random((n, n, n)) < 0.02) * rng.integers(1, 9, (n, n, n))).astype(np.uint8)
    with GraphicsCore(grid_n=n, voxel_format=VOXEL_PACKED_U8) as g:
        g.upload_world(block)
        for pos, yaw, pitch in [((16.0, 16.0, 16.0), 0.0, 0.0), ((20.3, 11.2, 9.9), 1.1, -0.4)]:
            cam = scenes.cam16(Player.at(pos, yaw, pitch))
            want = O.trace_secondary(O.World(block), cam, 96, 64)
            got = g.trace_secondary(cam, 96, 64)
            for k in ("shadow", "ao", "sec_steps"):
                assert np.array_equal(got[k], want[k]), (k, pos)


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["binned", "pixel"])
def test_gpu_secondary_with_a_small_step_cap(kernel, monkeypatch):
    """A step cap below 3n selects the one-step-per-trip form of the traversal, which the binned kernel runs whole
    for each ray it draws; capped secondary rays count as misses with max_steps iterations."""
    from raytracing_amd import GraphicsCore, Player, VOXEL_PACKED_U8
    monkeypatch.setenv("VRT_SECONDARY", kernel)
    rng = np.random.default_rng(21)
    n, cap = 96, 60
    block = ((rng.random((n, n, n)) < 0.004) * rng.integers(1, 9, (n, n, n))).astype(np.uint8)
    block[:, :20, :] = 3  # a floor, so that most pixels hit
    cam = scenes.cam16(Player.at((40.2, 30.7, 44.9), 0.7, -0.6))
    want = O.trace_secondary(O.World(block, max_steps=cap), cam, 160, 96)
    with GraphicsCore(grid_n=n, voxel_format=VOXEL_PACKED_U8, max_steps=cap) as g:
        g.upload_world(block)
        got = g.trace_secondary(cam, 160, 96)
    for k in ("shadow", "ao", "sec_steps"):
        assert np.array_equal(got[k], want[k]), k
    assert want["sec_steps"].max() >= cap and want["shadow"].any()


@pytest.mark.gpu
def test_trace_kernels_keep_eight_workgroups_per_compute_unit():
    """The skip kernel is held to 64 registers and the binned secondary-ray kernel to 20 KB of LDS so that a compute unit
    holds eight workgroups of either (all 32 wavefront slots); a build that loses one still traces every pixel right,
    so the runtime's occupancy calculation is asked directly."""
    from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8
    with GraphicsCore(grid_n=64, voxel_format=VOXEL_PACKED_U8) as g:
        assert g.kernel_occupancy(0) == 8
        assert g.kernel_occupancy(1) == 8
        assert g.kernel_occupancy(2) >= 8
