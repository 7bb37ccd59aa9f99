"""GPU world producers against the CPU oracle: the terrain generator kernel
(≙ upload_chunk_gpu, src/graphics/mod.rs:245-291 running terrain_gen.comp) must produce
the same voxels bit for bit, so that hit-map parity never depends on who built the scene."""
import numpy as np
import pytest

import oracle_lib as O
from raytracing_amd import GraphicsCore, TRACE_SKIP, VOXEL_PACKED_U8, VOXEL_REF_U16U16, scenes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,fmt,dtype", [(160, VOXEL_REF_U16U16, np.uint16), (256, VOXEL_PACKED_U8, np.uint8)])
def test_generate_window_matches_oracle(n, fmt, dtype):
    wmin = scenes.window_min(scenes.POSE_B_POSITION, n)
    want = O.terrain_fill(n, wmin, dtype)
    with GraphicsCore(grid_n=n, voxel_format=fmt) as g:
        g.generate_window(wmin)
        got = g.download_world()
    assert got.dtype == want.dtype and np.array_equal(got, want)
    assert 0.2 < want.astype(bool).mean() < 0.8


def test_upload_chunks_around_player_builds_the_default_scene():
    # GameState::new -> upload_chunks_around_player(None): 5^3 chunks around chunk (0,0,0)
    # fill the whole 160^3 torus (src/game_logic/game_state.rs:50-83).
    with GraphicsCore() as g:
        assert g.grid_n == 160
        count = g.upload_chunks_around_player((0.0, 0.0, 0.0), None)
        assert count == 125
        got = g.download_world()
        want = O.terrain_fill(160, (-64, -64, -64), np.uint16)
        assert np.array_equal(got, want)
        # moving one chunk in +x regenerates only the newly exposed 25 chunks
        count = g.upload_chunks_around_player((33.0, 0.0, 0.0), (0, 0, 0))
        assert count == 25
        got = g.download_world()
        for cz in range(-2, 3):
            for cy in range(-2, 3):
                O.terrain_chunk(want, 3, cy, cz)
        assert np.array_equal(got, want)
        assert g.upload_chunks_around_player((40.0, 1.0, 2.0), (1, 0, 0)) == 0


def test_generated_world_traces_like_uploaded_world():
    p = scenes.pose_b()
    wmin = scenes.window_min(p.get_position(), 160)
    cam = scenes.cam16(p)
    block = O.terrain_fill(160, wmin, np.uint16)
    with GraphicsCore(trace_mode=TRACE_SKIP) as g:
        g.generate_window(wmin)
        g.reset_distance_field()
        got = g.redraw(p.push_constants(), outputs=("bgra8", "voxel", "face", "steps", "t"))
    full = O.trace(O.World(block), cam, 640, 480)
    for k in got:
        a, b = got[k], full[k]
        if a.dtype == np.float32:
            a, b = a.view(np.uint32), b.view(np.uint32)
        assert np.array_equal(a, b), k


def test_generator_far_from_the_surface_matches_oracle():
    # the kernel skips the noise where |y| > 128 (its sign cannot matter there); the oracle does not
    n = 512
    wmin = (-100, -250, 40)
    with GraphicsCore(grid_n=n, voxel_format=VOXEL_PACKED_U8) as g:
        g.generate_window(wmin)
        got = g.download_world()
    rng = np.random.default_rng(5)
    import pyref
    ys = np.concatenate([np.arange(-250, -120), np.arange(120, 262)])
    for y in ys[::3]:
        for _ in range(4):
            x = int(rng.integers(wmin[0], wmin[0] + n)); z = int(rng.integers(wmin[2], wmin[2] + n))
            noise = float(O.snoise(np.float32(x) * np.float32(0.015), np.float32(y) * np.float32(0.015), np.float32(z) * np.float32(0.015)))
            want = 1 if np.float32(np.float32(noise) * np.float32(6.0)) - np.float32(y) > 0 else 0
            assert got[z % n, y % n, x % n] == want, (x, y, z)
    # whole layers: solid below, air above
    assert got[:, (-200) % n, :].all() and not got[:, 200 % n, :].any()


@pytest.mark.parametrize("render_distance", [1, 3, 6])  # grids of 160, 288 and 480 voxels: 40, 72 and 120 cells per axis
def test_index_updated_chunk_by_chunk_equals_an_index_built_whole(render_distance):
    """After a chunk crossing only the regenerated chunks' cells and the cells within 31 cells of them are
    re-indexed (on the 160 grid that is everything, on the 288 grid a box that wraps round the torus, on the 480
    grid a slab).  Every record, and every pyramid word after a trace from the new position, must equal those of
    a context that was handed the same voxels whole; the frames of both must be equal on every plane."""
    from raytracing_amd import GraphicsSettings, Player
    n = 32 * (2 * render_distance + 3)
    walk = [(0.0, 10.0, 0.0), (33.0, 10.0, 0.0), (33.0, 10.0, -1.0), (33.0, -2.0, -1.0), (70.0, -2.0, -1.0),
            (100.0, 33.0, -40.0),   # two chunks in x, one in y, one in z at once
            (100.0, 33.0, -40.0 - 32.0 * (2 * render_distance + 4)),  # further than the window is wide: every chunk is new
            (60.0, 33.0, -40.0 - 32.0 * (2 * render_distance + 4))]
    planes = ("bgra8", "voxel", "face", "steps", "t")
    with GraphicsCore(settings=GraphicsSettings(render_distance=render_distance), trace_mode=TRACE_SKIP) as g, \
         GraphicsCore(settings=GraphicsSettings(render_distance=render_distance), trace_mode=TRACE_SKIP) as fresh:
        assert g.grid_n == n
        old = None
        for i, pos in enumerate(walk):
            count = g.upload_chunks_around_player(pos, old, batched=i % 2 == 1)  # chunk by chunk, or 64 chunks per launch
            old = tuple(int(np.floor(v / 32.0)) for v in pos)
            assert i == 0 or count > 0
            fresh.upload_world(g.download_world())
            fresh.reset_distance_field()
            cam = scenes.cam16(Player.at(pos, 0.3 + 0.9 * i, -0.35))
            a = g.trace(cam, 320, 240, outputs=planes)
            b = fresh.trace(cam, 320, 240, outputs=planes)
            for k in planes:
                assert np.array_equal(a[k].view(np.uint32) if a[k].dtype == np.float32 else a[k],
                                      b[k].view(np.uint32) if b[k].dtype == np.float32 else b[k]), (i, k)
            ca, pa = g.download_index(pyramid=True)
            cb, pb = fresh.download_index(pyramid=True)
            assert np.array_equal(ca, cb), (i, int((ca != cb).any(axis=1).sum()))
            assert np.array_equal(pa, pb), (i, int((pa != pb).any(axis=1).sum()))
        # the walk ends on frames the oracle agrees with (the 160 grid is small enough for it)
        if n == 160:
            full = O.trace(O.World(g.download_world()), cam, 320, 240)
            for k in planes:
                x, y = a[k], full[k]
                if x.dtype == np.float32:
                    x, y = x.view(np.uint32), y.view(np.uint32)
                assert np.array_equal(x, y), k


@pytest.mark.parametrize("n,kind", [(64, "random"), (96, "terrain"), (160, "terrain")])
def test_box_words_against_their_definition_in_numpy(n, kind):
    """Every entry of the cells' box words, as downloaded, against the definition worked out here from the voxels (and
    the records' octant distances, which the parity tests cover): (side - 1) << 8 | rise, or 0xffff for none.  The column
    square of level L -- 2^L x 2^L cell columns with the cell's own at the corner, extending the way the word's rays
    travel -- is valid for a voxel of window level v when it holds no solid voxel at or above v - drop; rising words take
    the widest valid square (rise 0xff: to the top of the window), descending words the one with the largest
    min(side in voxels, 4 * (drop + 1)), the wider of equals (rise = drop, at most 254).  The octant cube of the record
    (side d) takes its place unless twice the square's side exceeds d; its rise is what is left of the voxel's cell plus
    4 (d - 1)."""
    rng = np.random.default_rng(11)
    pose = scenes.pose_b(9.3)
    wmin = scenes.window_min(pose.get_position(), n)
    if kind == "terrain":
        block = O.terrain_fill(n, wmin, np.uint8)
    else:
        block = (rng.random((n, n, n)) < 0.004).astype(np.uint8)
        block[:, : n // 3, :] |= (rng.random((n, n // 3, n)) < 0.3).astype(np.uint8)
    nc = n // 4
    with GraphicsCore(grid_n=n, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
        g.upload_world(block)
        g.reset_distance_field()
        g.trace(scenes.cam16(pose), 64, 48, outputs=("bgra8",))   # levels the words for this camera's window
        records, words = g.download_index(pyramid=True)
    words = words.reshape(nc, nc, nc, 4, 2, 2).astype(np.int64)          # [tcz, tcy, tcx, quadrant, rising/descending, half]
    entries = np.stack([words[..., 0] & 0xffff, words[..., 0] >> 16, words[..., 1] & 0xffff, words[..., 1] >> 16], -1)  # [..., j]
    dist = records[:, 2:4].copy().view(np.uint8).reshape(nc, nc, nc, 8).astype(np.int64)   # byte o = sx | sy << 1 | sz << 2
    # window level of every texel row: texel = rem_euclid(world, n), world = wmin + level
    off = int(wmin[1]) % n
    solid = np.roll(block != 0, -off, axis=1)                              # [tz, level, tx]
    lvl = np.arange(1, n + 1)[None, :, None]
    top = (solid * lvl).max(axis=1).reshape(nc, 4, nc, 4).max(axis=(1, 3))  # [tcz, tcx]: level just above the highest solid voxel
    tables = [top]
    while (2 << (len(tables) - 1)) <= nc and len(tables) < 9:
        h, t = 1 << (len(tables) - 1), tables[-1]
        t = np.maximum(t, np.roll(t, -h, axis=1))
        tables.append(np.maximum(t, np.roll(t, -h, axis=0)))
    offc = off // 4
    wl = (np.arange(nc) - offc) % nc                                       # window layer of torus cell row tcy
    j = np.arange(4)[None, None, None, :]
    for q in range(4):
        negx, negz = q & 1, q >> 1
        up_side = np.zeros((nc, nc, nc, 4), np.int64)
        down_side = np.zeros((nc, nc, nc, 4), np.int64)
        down_drop = np.zeros((nc, nc, nc, 4), np.int64)
        score = np.full((nc, nc, nc, 4), -1, np.int64)
        v = 4 * wl[None, :, None, None] + j
        for L, t in enumerate(tables):
            back = (1 << L) - 1
            tq = np.roll(t, (back if negz else 0, back if negx else 0), axis=(0, 1))[:, None, :, None]   # [tcz, 1, tcx, 1]
            ok = tq <= v
            sc = np.minimum(4 << L, 4 * (v - tq + 1))
            up_side = np.where(ok, 1 << L, up_side)
            better = ok & (sc >= score)
            down_side = np.where(better, 1 << L, down_side)
            down_drop = np.where(better, v - tq, down_drop)
            score = np.where(better, sc, score)
        o = (0 if negx else 1) | (0 if negz else 4)
        d_up, d_down = dist[..., o | 2][..., None], dist[..., o][..., None]
        up = np.where(2 * up_side > d_up, ((np.minimum(up_side, 255) - 1) << 8) | 0xff,
                      np.where(d_up > 0, ((d_up - 1) << 8) | ((3 - j) + 4 * (d_up - 1)), 0xffff))
        down = np.where(2 * down_side > d_down, ((np.minimum(down_side, 255) - 1) << 8) | np.minimum(down_drop, 254),
                        np.where(d_down > 0, ((d_down - 1) << 8) | (j + 4 * (d_down - 1)), 0xffff))
        assert np.array_equal(entries[:, :, :, q, 0, :], up), (q, "rising")
        assert np.array_equal(entries[:, :, :, q, 1, :], down), (q, "descending")
