"""Known-answer tests for the CPU oracle, derived by hand from the GLSL
(src/graphics/shaders/rendering/raytrace.glsl:26-77, util.glsl:28-42,
render.comp:6-14).  The reference ships no golden vectors, so these are the
pins: every expected value below was worked out on paper from the shader text
with inputs chosen so that all intermediate floats are exact in binary32.

Geometry used throughout: n = 32 and a camera inside chunk (0,0,0), so the
loaded area is lower = 16-16 = 0, upper = 16+16-1 = 31 and the loop runs while
0 < check_point < 31 on every axis (raytrace.glsl:19-24).
"""
import numpy as np
import pytest

import oracle_lib as O
import pyref

f32 = np.float32


def empty(n=32, dtype=np.uint16):
    return np.zeros((n, n, n), dtype)


def solid_at(n, voxels, dtype=np.uint16, value=1):
    b = empty(n, dtype)
    for (x, y, z) in voxels:
        b[z % n, y % n, x % n] = value
    return b


RO = (16.5, 16.5, 16.5)


def both(block, ro, rd, dist=None, max_steps=None):
    w = O.World(block, dist, max_steps)
    a = O.trace_ray(w, ro, rd)
    b = pyref.trace_ray(block, dist, ro, rd, w.max_steps)
    for k in ("hit", "capped", "voxel", "face", "steps", "block_id"):
        assert a[k] == b[k], (k, a[k], b[k])
    assert a["t"].tobytes() == b["t"].tobytes()
    assert a["hit_point"].tobytes() == b["hit_point"].tobytes()
    assert a["normal"].tobytes() == b["normal"].tobytes()
    return a


def test_axis_dominant_plus_x():
    # t per axis: x planes at 0.5+i, y planes at 2+4j, z planes at 4+8m.
    # visited: (17,16,16) t=.5, (18,16,16) t=1.5, (18,17,16) t=2 [y], (19,17,16) t=2.5, (20,17,16) t=3.5
    r = both(solid_at(32, [(20, 17, 16)], value=7), RO, (1.0, 0.25, 0.125))
    assert r["hit"] == 1 and r["steps"] == 5 and r["voxel"] == (20, 17, 16)
    assert r["face"] == 0 and r["block_id"] == 7  # normal = -x
    assert r["t"] == f32(3.5)
    assert tuple(r["hit_point"]) == (20.0, 17.375, 16.9375)
    assert tuple(r["normal"]) == (-1.0, 0.0, 0.0)


def test_axis_dominant_minus_x():
    # s01 = 0, s11 = -1: check_point = anchor; x planes at .5,1.5,2.5; y at 2, 6; z at 4
    # visited: (15,16,16) .5, (14,16,16) 1.5, (14,15,16) 2 [y], (13,15,16) 2.5
    r = both(solid_at(32, [(13, 15, 16)]), RO, (-1.0, -0.25, -0.125))
    assert r["hit"] == 1 and r["steps"] == 4 and r["voxel"] == (13, 15, 16)
    assert r["face"] == 1  # normal = +x
    assert r["t"] == f32(2.5)
    assert tuple(r["hit_point"]) == (14.0, 15.875, 16.1875)
    assert tuple(r["normal"]) == (1.0, 0.0, 0.0)


@pytest.mark.parametrize("axis", [0, 1, 2])
@pytest.mark.parametrize("sign", [1, -1])
def test_each_face(axis, sign):
    # dominant axis `axis`; the two others are slow enough never to step before the hit
    rd = [0.001953125, 0.0009765625, 0.00048828125]
    rd[axis] = float(sign)
    target = [16, 16, 16]
    target[axis] += 3 * sign
    r = both(solid_at(32, [tuple(target)]), RO, rd)
    assert r["hit"] == 1 and r["steps"] == 3 and r["voxel"] == tuple(target)
    assert r["face"] == 2 * axis + (1 if sign < 0 else 0)
    assert r["t"] == f32(2.5)


def test_tie_order_x_then_y_then_z():
    # all three t equal -> argmin's strict '<' picks x, then y, then z (util.glsl:28-42)
    order = [(17, 16, 16), (17, 17, 16), (17, 17, 17), (18, 17, 17), (18, 18, 17), (18, 18, 18)]
    for steps, v in enumerate(order, start=1):
        r = both(solid_at(32, [v]), RO, (1.0, 1.0, 1.0))
        assert r["hit"] == 1 and r["voxel"] == v and r["steps"] == steps
        assert r["face"] == 2 * ((steps - 1) % 3)
        assert r["t"] == f32(0.5 + (steps - 1) // 3)


def test_miss_runs_to_window_edge():
    # empty grid: x steps at t=.5..13.5 (14 of them, check_point.x 17 -> 31 ends the loop),
    # y steps at 2,6,10 (3), z steps at 4,12 (2): 19 iterations, last anchor (30,19,18)
    r = both(empty(), RO, (1.0, 0.25, 0.125))
    assert r["hit"] == 0 and r["capped"] == 0 and r["face"] == -1
    assert r["steps"] == 19 and r["voxel"] == (30, 19, 18)
    assert tuple(r["normal"]) == (0.0, 0.0, 0.0)


def test_start_voxel_is_never_tested():
    r = both(solid_at(32, [(16, 16, 16)]), RO, (1.0, 0.25, 0.125))
    assert r["hit"] == 0 and r["steps"] == 19


def test_toroidal_addressing_negative_chunk():
    # camera in chunk x = -1: mid.x = -16, window x in (-32, -1); world x = -14 -> texel 18
    block = empty()
    block[16, 16, 18] = 3
    r = both(block, (-15.5, 16.5, 16.5), (1.0, 0.25, 0.125))
    assert r["hit"] == 1 and r["voxel"] == (-14, 16, 16) and r["steps"] == 2 and r["block_id"] == 3
    assert r["t"] == f32(1.5)


def test_u8_and_u16_agree():
    for dt in (np.uint8, np.uint16):
        r = both(solid_at(32, [(20, 17, 16)], dtype=dt), RO, (1.0, 0.25, 0.125))
        assert r["steps"] == 5 and r["voxel"] == (20, 17, 16)


def test_thousand_step_miss_is_teal():
    # n = 704: window half-width 352 -> a diagonal miss takes > 1000 iterations
    # (raytrace.glsl:73-75).  The zero grid is never touched except along the ray.
    n = 704
    block = np.zeros((n, n, n), np.uint8)
    w = O.World(block)
    r = O.trace_ray(w, RO, (1.0, 1.0, 1.0))
    # x,y,z alternate; check_point.x = 17 + i leaves at 16 + 352 - 1 = 367 -> 350 x-steps,
    # and the y and z steps with the same t run before the loop test fails: 3*350 - 2
    assert r["hit"] == 0 and r["capped"] == 0
    assert r["steps"] == 1048
    assert np.allclose(r["normal"], (0.2, 1.0, 0.8))
    r = O.trace_ray(w, RO, (1.0, 0.25, 0.125))
    assert r["steps"] < 1000 and tuple(r["normal"]) == (0.0, 0.0, 0.0)


def test_step_cap_on_degenerate_ray():
    # rd.x == 0, s11.x = -1, check_point.x = anchor.x.  t.x = (cp.x - ro.x)*inf = -inf wins
    # argmin; ro + rd*t is NaN on x, so anchor.x = int(round(NaN)) - 1 + 0 = -1 (canonical
    # NaN -> 0).  From then on t.x = (-1 - ro.x)*inf = -inf again and the anchor never moves:
    # the reference would spin for ever, the canonical semantics stop at max_steps.
    # n = 64, camera in chunk 0: window (-16, 47), so x = -1 is inside.
    r = both(empty(64), (0.5, 16.5, 16.5), (0.0, 0.5, 1.0), max_steps=77)
    assert r["hit"] == 0 and r["capped"] == 1 and r["face"] == -2 and r["steps"] == 77
    assert r["voxel"] == (-1, 16, 16)
    # integer ro.x: first t.x = 0*inf = NaN; NaN wins because it sits in .x (util.glsl:28-42)
    r = both(empty(64), (0.0, 16.5, 16.5), (0.0, 0.5, 1.0), max_steps=50)
    assert r["capped"] == 1 and r["steps"] == 50 and r["voxel"] == (-1, 16, 16)
    # with n = 32 the window is (0, 31): x = -1 is outside and the same ray ends after one step
    r = both(empty(32), RO, (0.0, 0.5, 1.0), max_steps=77)
    assert r["capped"] == 0 and r["steps"] == 1 and r["voxel"] == (-1, 16, 16)


def test_zero_component_that_never_wins_is_a_plain_2d_walk():
    # rd.x = -0: inv.x = -inf, check_point.x = anchor.x = 16, t.x = (16 - 16.5)*(-inf) = +inf never wins argmin,
    # and ro.x + (-0)*t = 16.5 keeps anchor.x.  y planes 17, 18, 19 at t = .5, 1.5, 2.5; z planes 17, 18 at t = 1, 3.
    # visited: (16,17,16) .5 [y], (16,17,17) 1 [z], (16,18,17) 1.5 [y], (16,19,17) 2.5 [y]
    r = both(solid_at(32, [(16, 19, 17)], value=5), RO, (-0.0, 1.0, 0.5))
    assert r["hit"] == 1 and r["steps"] == 4 and r["voxel"] == (16, 19, 17) and r["block_id"] == 5
    assert r["face"] == 2 and r["t"] == f32(2.5)          # normal = -y
    assert tuple(r["hit_point"]) == (16.5, 19.0, 17.75)


def test_zero_component_that_wins_once_moves_the_anchor_to_minus_one():
    # rd.x = +0, ro.x = -5.5 (chunk -1: window x in (-48, 15), n = 64).  Iteration 1: t.x = (-6 + 5.5)*inf = -inf
    # wins, the position is NaN on x, anchor.x = f2i(NaN) - 1 + 0 = -1 (raytrace.glsl:50), y and z keep 16; voxel
    # (-1,16,16) is air.  From there t.x = (-1 + 5.5)*inf = +inf never wins again and the ray walks y and z as
    # above, at x = -1: (-1,17,16) .5, (-1,17,17) 1, (-1,18,17) 1.5, (-1,19,17) 2.5 -- five iterations in all.
    r = both(solid_at(64, [(-1, 19, 17)], value=9), (-5.5, 16.5, 16.5), (0.0, 1.0, 0.5))
    assert r["hit"] == 1 and r["steps"] == 5 and r["voxel"] == (-1, 19, 17) and r["block_id"] == 9
    assert r["face"] == 2 and r["t"] == f32(2.5)
    assert tuple(r["hit_point"]) == (-5.5, 19.0, 17.75)


def test_free_dist_general_formula():
    # documentation of the unfinished upstream skip (never active as shipped): with
    # free_dist = 2 at the start voxel the first iteration jumps to the far face of the
    # 5-wide cube: check_point = anchor + 1 + 2 = 19, t.x = 2.5, and the new anchor is
    # clamped into (16..18) on y,z and set to 19 on x (raytrace.glsl:35,48-50).
    block = solid_at(32, [(19, 17, 16)])
    dist = np.zeros((32, 32, 32), np.uint16)
    dist[16, 16, 16] = 2
    r = both(block, RO, (1.0, 0.25, 0.125), dist=dist)
    assert r["hit"] == 1 and r["steps"] == 1 and r["voxel"] == (19, 17, 16) and r["t"] == f32(2.5)


def test_raygen_identity_centre_pixel():
    cam = np.eye(4, dtype=f32).T.reshape(16)
    ro, rd = O.raygen(cam, 640, 480, 320, 240)
    ln = np.sqrt(f32(0.015625) + f32(1.0))
    assert tuple(ro) == (0.0, 0.0, 0.0)
    assert rd[0] == 0.0 and rd[1] == f32(0.125) / ln and rd[2] == f32(1.0) / ln
    # asymmetric y range: nc.y = gy/H - H/(2W)
    _, rd0 = O.raygen(cam, 640, 480, 0, 0)
    d = np.array([-0.5, -0.375, 1.0], dtype=f32)
    assert np.array_equal(rd0, d / np.sqrt(f32(f32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])))


def test_raygen_matches_pyref_random():
    rng = np.random.default_rng(1)
    from raytracing_amd.player import Player
    for _ in range(50):
        p = Player.at(rng.uniform(-50, 50, 3), rng.uniform(-3, 3), rng.uniform(-1.5, 1.5))
        cam = p.push_constants().column_major()
        gx, gy = int(rng.integers(0, 641)), int(rng.integers(0, 481))
        a = O.raygen(cam, 640, 480, gx, gy)
        b = pyref.raygen(cam, 640, 480, gx, gy)
        assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()


def test_frame_flip_and_crosshair():
    # render.comp:10-13: invocation (x, y) is stored at row H - y; the crosshair is the
    # five invocations within 1.2 of (W/2, H/2).
    w_, h_ = 64, 48
    block = empty()
    cam = np.eye(4, dtype=f32)
    cam[:3, 3] = RO
    out = O.trace(O.World(block), cam.T.reshape(16), w_, h_)
    red = np.array([77, 77, 255, 255], np.uint8)  # B,G,R,A of (1, .3, .3, 1)
    rows, cols = np.nonzero((out["bgra8"] == red).all(axis=2))
    got = sorted(zip(rows.tolist(), cols.tolist()))
    gy = h_ // 2
    want = sorted([(h_ - gy, 32), (h_ - (gy - 1), 32), (h_ - (gy + 1), 32), (h_ - gy, 31), (h_ - gy, 33)])
    assert got == want
    # everything else is a miss: black, alpha 255
    mask = np.ones((h_, w_), bool)
    mask[rows, cols] = False
    assert (out["bgra8"][mask] == np.array([0, 0, 0, 255], np.uint8)).all()
    assert (out["face"] == -1).sum() + (out["face"] == -2).sum() == w_ * h_


def test_frame_row_mapping_against_single_rays():
    rng = np.random.default_rng(5)
    block = (rng.random((32, 32, 32)) < 0.02).astype(np.uint16) * 5
    from raytracing_amd.player import Player
    p = Player.at((16.3, 17.1, 15.2), 0.7, -0.2)
    cam = p.push_constants().column_major()
    w_, h_ = 40, 30
    world = O.World(block)
    out = O.trace(world, cam, w_, h_)
    band = O.trace(world, cam, w_, h_, 7, 19)
    for k in out:
        assert np.array_equal(out[k][7:19], band[k])
    for row, col in [(0, 0), (29, 39), (11, 3), (15, 20)]:
        ro, rd = O.raygen(cam, w_, h_, col, h_ - row)
        r = O.trace_ray(world, ro, rd)
        assert tuple(out["voxel"][row, col]) == r["voxel"]
        assert out["steps"][row, col] == r["steps"] and out["face"][row, col] == r["face"]


def test_canonical_sine_is_a_sine():
    xs = np.concatenate([np.arange(-4096, 4097, 2), np.arange(-2**20 + 2, 2**20, 2 * 9973)]).astype(np.float32)
    got = np.array([O.sin(x) for x in xs], np.float32)
    assert np.abs(got - np.sin(xs.astype(np.float64))).max() < 3e-7
    xs = np.linspace(-7, 7, 2001).astype(np.float32)
    got = np.array([O.sin(x) for x in xs], np.float32)
    assert np.abs(got - np.sin(xs.astype(np.float64))).max() < 3e-7
    assert np.isnan(O.sin(np.float32("nan"))) and np.isnan(O.sin(np.float32("inf")))


def test_shading_of_a_hit_pixel():
    # one solid wall at x = 20: colour = normal + 0.5*sin(floor(hit + 0.001)*2) (raytrace.glsl:130)
    block = empty()
    block[:, :, 20] = 1
    cam = np.eye(4, dtype=f32)
    cam[:3, 3] = RO
    # look along +x: yaw so that (0,0,1) maps to +x: rd = R^T (nc,1)
    from raytracing_amd.player import Player
    p = Player.at(RO, -np.pi / 2, 0.0)
    out = O.trace(O.World(block), p.push_constants().column_major(), 32, 24)
    row, col = 5, 20
    assert out["face"][row, col] == 0 and out["voxel"][row, col, 0] == 20
    hp = out["hit_point"][row, col]
    want = np.array([-1.0, 0.0, 0.0], np.float32) + np.float32(0.5) * np.array(
        [O.sin(np.floor(h + np.float32(0.001)) * np.float32(2.0)) for h in hp], np.float32)
    assert np.array_equal(out["rgba"][row, col, :3], want)
    q = np.floor(np.clip(want, 0, 1) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
    assert tuple(out["bgra8"][row, col]) == (q[2], q[1], q[0], 255)
