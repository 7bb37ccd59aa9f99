"""Independent numpy-float32 restatement of the reference shaders, written
directly from the GLSL in vector form (the C oracle is written in scalar form).
Used only to cross-check the C oracle on small cases; pure-Python loops, slow.

  trace:   src/graphics/shaders/rendering/raytrace.glsl:19-77, util.glsl:1-8,28-42
  raygen:  src/graphics/shaders/rendering/raytrace.glsl:120-123
  snoise:  src/terrain/terrain_gen/noise.glsl:4-74
  terrain: src/terrain/terrain_gen/terrain_gen.comp:14-24
"""
import numpy as np

f32 = np.float32
i32 = np.int32
CHUNK_SIZE = 32


def _f2i(v):
    """float -> int, saturating, NaN -> 0 (canonical choice)."""
    v = np.asarray(v, dtype=f32)
    out = np.zeros(v.shape, dtype=np.int64)
    ok = ~np.isnan(v)
    with np.errstate(invalid="ignore", over="ignore"):
        c = np.clip(v.astype(np.float64), -2147483648.0, 2147483647.0)
    out[ok] = c[ok].astype(np.int64)
    return out.astype(i32)


def _wrap(v):
    return ((np.asarray(v, dtype=np.int64) + 2**31) % 2**32 - 2**31).astype(i32)


def rem_euclid_ivec3(p, n):  # util.glsl:1-8
    p = np.asarray(p, dtype=np.int64)
    r = np.fmod(p, n)  # truncated remainder, like GLSL % on what glslang emits for negatives + fix-up
    return np.where(r >= 0, r, r + n).astype(i32)


def argmin(a):  # util.glsl:28-42
    min_index, min_value = 0, a[0]
    if a[1] < min_value:
        min_value, min_index = a[1], 1
    if a[2] < min_value:
        min_index = 2
    return min_index


def raygen(m16, w, h, gx, gy):
    m = np.asarray(m16, dtype=f32).reshape(4, 4)  # m[c] = column c
    size = np.array([w, h], dtype=f32)
    nc = np.array([gx, gy], dtype=f32) / size - np.array([0.5, size[1] / size[0] * f32(0.5)], dtype=f32)
    v = np.array([nc[0], nc[1], 1.0, 1.0], dtype=f32)
    d = np.zeros(3, dtype=f32)
    for i in range(3):
        acc = f32(v[0] * m[i][0])
        acc = f32(acc + f32(v[1] * m[i][1]))
        acc = f32(acc + f32(v[2] * m[i][2]))
        acc = f32(acc + f32(v[3] * m[i][3]))
        d[i] = acc
    ln = np.sqrt(f32(f32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]))
    return m[3][:3].copy(), (d / ln).astype(f32)


def trace_ray(block, dist, ro, rd, max_steps):
    """block/dist indexed [z,y,x]; returns dict like the oracle's orc_ray."""
    n = block.shape[0]
    ro = np.asarray(ro, dtype=f32)
    rd = np.asarray(rd, dtype=f32)

    def texel(img, p):
        t = rem_euclid_ivec3(p, n)
        return int(img[t[2], t[1], t[0]])

    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        mid = _wrap(_f2i(np.floor(ro / f32(CHUNK_SIZE))).astype(np.int64) * CHUNK_SIZE + CHUNK_SIZE // 2)
        lower = _wrap(mid.astype(np.int64) - n // 2)
        upper = _wrap(mid.astype(np.int64) + n // 2 - 1)

        def inside(pos):
            return bool(np.all(lower < pos) and np.all(pos < upper))

        inv_rd = f32(1.0) / rd
        s01 = (rd > 0).astype(i32)
        s11 = s01 * 2 - 1
        anchor = _f2i(np.floor(ro))
        free = texel(dist, anchor) if dist is not None else 0
        cp = _wrap(anchor.astype(np.int64) + s01 + s11.astype(np.int64) * free)
        steps = 0
        capped = False
        while inside(cp):
            if steps >= max_steps:
                capped = True
                break
            steps += 1
            t = ((cp.astype(f32) - ro) * inv_rd).astype(f32)
            k = argmin(t)
            p = (ro + (rd * t[k]).astype(f32)).astype(f32)
            clamp_p = _wrap(cp.astype(np.int64) - s11.astype(np.int64) * (2 * free + 1))
            lo = _wrap(np.minimum(cp, clamp_p).astype(np.int64) + 1)
            hi = _wrap(np.maximum(cp, clamp_p).astype(np.int64) - 1)
            anchor = np.minimum(np.maximum(_f2i(np.floor(p)), lo), hi).astype(i32)
            anchor[k] = _wrap(int(_f2i(np.rint(p[k]))) - 1 + int(s01[k]))
            b = texel(block, anchor)
            if b > 0:
                normal = np.zeros(3, dtype=f32)
                normal[k] = -s11[k]
                return dict(hit=1, capped=0, voxel=tuple(int(v) for v in anchor), face=2 * k + (1 if s11[k] < 0 else 0),
                            steps=steps, block_id=b, t=f32(t[k]), hit_point=p, normal=normal)
            free = texel(dist, anchor) if dist is not None else 0
            cp = _wrap(anchor.astype(np.int64) + s01 + s11.astype(np.int64) * free)
    normal = np.zeros(3, dtype=f32)
    if steps >= 1000:
        normal[:] = (0.2, 1.0, 0.8)
    return dict(hit=0, capped=int(capped), voxel=tuple(int(v) for v in anchor), face=-2 if capped else -1,
                steps=steps, block_id=0, t=f32(-1.0), hit_point=np.zeros(3, f32), normal=normal)


# ---------------------------------------------------------------- noise

def _mod(x, y):
    return (x - y * np.floor(x / y)).astype(f32)


def _permute(x):
    return _mod(((x * f32(34.0)) + f32(1.0)) * x, f32(289.0))


def _taylor_inv_sqrt(r):
    return (f32(1.79284291400159) - f32(0.85373472095314) * r).astype(f32)


def _step(edge, x):
    return np.where(x < edge, f32(0.0), f32(1.0)).astype(f32)


def _dot(a, b):
    acc = f32(a[0] * b[0])
    for j in range(1, len(a)):
        acc = f32(acc + f32(a[j] * b[j]))
    return acc


def snoise(v):
    v = np.asarray(v, dtype=f32)
    C = np.array([1.0 / 6.0, 1.0 / 3.0]).astype(f32)
    D = np.array([0.0, 0.5, 1.0, 2.0], dtype=f32)
    Cyyy = np.array([C[1]] * 3, dtype=f32)
    Cxxx = np.array([C[0]] * 3, dtype=f32)
    i = np.floor(v + _dot(v, Cyyy)).astype(f32)
    x0 = ((v - i) + _dot(i, Cxxx)).astype(f32)
    g = _step(x0[[1, 2, 0]], x0)
    l = (f32(1.0) - g).astype(f32)
    i1 = np.minimum(g, l[[2, 0, 1]])
    i2 = np.maximum(g, l[[2, 0, 1]])
    x1 = ((x0 - i1) + f32(1.0) * Cxxx).astype(f32)
    x2 = ((x0 - i2) + f32(2.0) * Cxxx).astype(f32)
    x3 = ((x0 - f32(1.0)) + f32(3.0) * Cxxx).astype(f32)
    i = _mod(i, f32(289.0))
    p = _permute(i[2] + np.array([0.0, i1[2], i2[2], 1.0], dtype=f32))
    p = _permute((p + i[1]) + np.array([0.0, i1[1], i2[1], 1.0], dtype=f32))
    p = _permute((p + i[0]) + np.array([0.0, i1[0], i2[0], 1.0], dtype=f32))
    n_ = f32(1.0 / 7.0)
    ns = (n_ * D[[3, 1, 2]] - D[[0, 2, 0]]).astype(f32)
    j = (p - f32(49.0) * np.floor(p * ns[2] * ns[2])).astype(f32)
    x_ = np.floor(j * ns[2]).astype(f32)
    y_ = np.floor(j - f32(7.0) * x_).astype(f32)
    x = (x_ * ns[0] + ns[1]).astype(f32)
    y = (y_ * ns[0] + ns[1]).astype(f32)
    h = ((f32(1.0) - np.abs(x)) - np.abs(y)).astype(f32)
    b0 = np.array([x[0], x[1], y[0], y[1]], dtype=f32)
    b1 = np.array([x[2], x[3], y[2], y[3]], dtype=f32)
    s0 = (np.floor(b0) * f32(2.0) + f32(1.0)).astype(f32)
    s1 = (np.floor(b1) * f32(2.0) + f32(1.0)).astype(f32)
    sh = (-_step(h, np.zeros(4, dtype=f32))).astype(f32)
    a0 = (b0[[0, 2, 1, 3]] + s0[[0, 2, 1, 3]] * sh[[0, 0, 1, 1]]).astype(f32)
    a1 = (b1[[0, 2, 1, 3]] + s1[[0, 2, 1, 3]] * sh[[2, 2, 3, 3]]).astype(f32)
    p0 = np.array([a0[0], a0[1], h[0]], dtype=f32)
    p1 = np.array([a0[2], a0[3], h[1]], dtype=f32)
    p2 = np.array([a1[0], a1[1], h[2]], dtype=f32)
    p3 = np.array([a1[2], a1[3], h[3]], dtype=f32)
    norm = _taylor_inv_sqrt(np.array([_dot(p0, p0), _dot(p1, p1), _dot(p2, p2), _dot(p3, p3)], dtype=f32))
    p0 = (p0 * norm[0]).astype(f32)
    p1 = (p1 * norm[1]).astype(f32)
    p2 = (p2 * norm[2]).astype(f32)
    p3 = (p3 * norm[3]).astype(f32)
    m = np.maximum(f32(0.6) - np.array([_dot(x0, x0), _dot(x1, x1), _dot(x2, x2), _dot(x3, x3)], dtype=f32), f32(0.0)).astype(f32)
    m = (m * m).astype(f32)
    return f32(f32(42.0) * _dot((m * m).astype(f32), np.array([_dot(p0, x0), _dot(p1, x1), _dot(p2, x2), _dot(p3, x3)], dtype=f32)))


def terrain_block(px, py, pz):
    pos = np.array([px, py, pz], dtype=f32)
    noise_val = f32(f32(snoise(pos * f32(0.015)) * f32(6.0)) - f32(py))
    return 1 if noise_val > 0 else 0
