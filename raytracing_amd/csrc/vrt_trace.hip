// vrt_trace.hip -- primary-ray kernels for gfx950.
//
// One thread per pixel; a 256-thread workgroup covers a 16x16 pixel tile as four
// wavefronts of 8x8 pixels, so the 64 rays of a wavefront stay spatially coherent.
//
//   k_trace_literal : one voxel per iteration, straight restatement of
//                     raytrace.glsl:26-77 (reads the linear block / distance images).
//   k_trace_skip    : the same visit order reproduced exactly, but empty space is
//                     crossed in one iteration per empty box.  Reads one 16-byte record
//                     per 4^3 cell visited (occupancy mask + octant distances) instead
//                     of one voxel per step; the voxel image is read once, at the hit.
//
// Reference lines followed: render.comp:6-14 (entry, crosshair, flipped store),
// raytrace.glsl:19-24 (loaded area), :26-77 (traversal), :117-134 (ray generation,
// shading), util.glsl:1-8,28-42.
#include "vrt_device.h"
#include "vrt_kernels.h"

namespace vrt {

struct RayHit {
    int      hit, capped;
    int      vx, vy, vz;
    int      k;      // axis of the last step
    int      sk;     // s11 of that axis
    unsigned steps;
    unsigned iters;  // iterations of the kernel's own loop (diagnostic)
    unsigned block_id;
    float    t, hx, hy, hz;
};

VRT_DEV size_t texel_index(const TraceParams &P, int x, int y, int z)
{
    if (P.n_pow2_mask >= 0) {
        x &= P.n_pow2_mask; y &= P.n_pow2_mask; z &= P.n_pow2_mask;
    } else {
        x = rem_euclid(x, P.n); y = rem_euclid(y, P.n); z = rem_euclid(z, P.n);
    }
    return ((size_t)z * (size_t)P.n + (size_t)y) * (size_t)P.n + (size_t)x;
}

template <typename T>
VRT_DEV unsigned load_block(const TraceParams &P, int x, int y, int z)
{
    return (unsigned)((const T *)P.block)[texel_index(P, x, y, z)];
}

// raytrace.glsl:19-24; lower = window minimum, upper = window minimum + n - 1.
VRT_DEV void loaded_area(const TraceParams &P, float ro, int &lower, int &upper)
{
    int mid = wadd(wmul(f2i(floorf(ro / (float)kChunk)), kChunk), kChunk / 2);
    lower   = wsub(mid, P.n / 2);
    upper   = wsub(wadd(mid, P.n / 2), 1);
}

// ---------------------------------------------------------------------------
// literal traversal (raytrace.glsl:26-77)
// ---------------------------------------------------------------------------
template <typename T, bool HAS_DIST>
VRT_DEV void traverse_literal(const TraceParams &P, float rox, float roy, float roz, float rdx, float rdy,
                              float rdz, int lox, int loy, int loz, int upx, int upy, int upz, RayHit &H)
{
    const float invx = 1.0f / rdx, invy = 1.0f / rdy, invz = 1.0f / rdz;          // :27
    const int   s01x = rdx > 0.0f ? 1 : 0, s01y = rdy > 0.0f ? 1 : 0, s01z = rdz > 0.0f ? 1 : 0;  // :29
    const int   s11x = 2 * s01x - 1, s11y = 2 * s01y - 1, s11z = 2 * s01z - 1;     // :30
    int ax = f2i(floorf(rox)), ay = f2i(floorf(roy)), az = f2i(floorf(roz));        // :32
    int fd = 0;
    if (HAS_DIST) fd = (int)P.dist[texel_index(P, ax, ay, az)];                     // :33
    int cpx = wadd(wadd(ax, s01x), wmul(s11x, fd));                                 // :35
    int cpy = wadd(wadd(ay, s01y), wmul(s11y, fd));
    int cpz = wadd(wadd(az, s01z), wmul(s11z, fd));
    unsigned steps = 0;
    H.hit = 0; H.capped = 0; H.block_id = 0; H.k = 0; H.sk = 0;
    for (;;) {
        if (!(lox < cpx && loy < cpy && loz < cpz && cpx < upx && cpy < upy && cpz < upz)) break;  // :43
        if (steps >= P.max_steps) { H.capped = 1; break; }
        ++steps;                                                                    // :44
        const float tx = ((float)cpx - rox) * invx;                                 // :45
        const float ty = ((float)cpy - roy) * invy;
        const float tz = ((float)cpz - roz) * invz;
        const int   k  = argmin3(tx, ty, tz);                                       // :46
        const float tk = k == 0 ? tx : (k == 1 ? ty : tz);
        const float px = rox + rdx * tk, py = roy + rdy * tk, pz = roz + rdz * tk;
        if (HAS_DIST) {                                                             // :48-49
            const int w2 = wadd(wmul(2, fd), 1);
            const int qx = wsub(cpx, wmul(s11x, w2)), qy = wsub(cpy, wmul(s11y, w2)), qz = wsub(cpz, wmul(s11z, w2));
            ax = min(max(f2i(floorf(px)), wadd(min(cpx, qx), 1)), wsub(max(cpx, qx), 1));
            ay = min(max(f2i(floorf(py)), wadd(min(cpy, qy), 1)), wsub(max(cpy, qy), 1));
            az = min(max(f2i(floorf(pz)), wadd(min(cpz, qz), 1)), wsub(max(cpz, qz), 1));
        }
        // With a zero distance the clamp bounds are [a+1, a] and min(max(x, a+1), a) == a
        // for every x, so the anchor keeps its value on all axes before line :50.
        const float pk  = k == 0 ? px : (k == 1 ? py : pz);
        const int   s01 = k == 0 ? s01x : (k == 1 ? s01y : s01z);
        const int   ak  = wadd(wsub(f2i(rintf(pk)), 1), s01);                       // :50
        if (k == 0) ax = ak; else if (k == 1) ay = ak; else az = ak;
        const unsigned b = load_block<T>(P, ax, ay, az);                            // :52
        if (b > 0) {                                                                // :53-60
            H.hit = 1; H.block_id = b; H.k = k; H.sk = k == 0 ? s11x : (k == 1 ? s11y : s11z);
            H.t = tk; H.hx = px; H.hy = py; H.hz = pz;
            break;
        }
        if (HAS_DIST) fd = (int)P.dist[texel_index(P, ax, ay, az)];                 // :62
        cpx = wadd(wadd(ax, s01x), wmul(s11x, fd));                                 // :67
        cpy = wadd(wadd(ay, s01y), wmul(s11y, fd));
        cpz = wadd(wadd(az, s01z), wmul(s11z, fd));
    }
    H.vx = ax; H.vy = ay; H.vz = az; H.steps = steps; H.iters = steps;
    if (!H.hit) { H.t = -1.0f; H.hx = H.hy = H.hz = 0.0f; }
}

// ---------------------------------------------------------------------------
// exact empty-space skipping
// ---------------------------------------------------------------------------
// The reference advances one voxel per iteration: the axis is argmin over the three
// plane parameters t_k(i) = (float(i) - ro_k) * inv_k of the next plane i on each axis,
// ties resolved x, y, z.  Each t_k(i) is recomputed from the integer plane (no
// accumulation) and is monotone in the stepping direction, so the visit order is the
// merge of three sorted event lists under the key (t, axis).  Crossing an empty
// axis-aligned box can therefore be done in one go: the exit event is the smallest
// (t, axis) among the three exit planes, and every other axis advances by the number of
// its events that sort before the exit event.
//
// That count comes from floor(ro + rd*t_exit) whenever the exit point is provably on the
// right side of every plane: the sign of t_j(i) - t_exit equals the sign of
// (i - pos_j)/rd_j as soon as |pos_j - i| exceeds u*(3|i - ro_j| + |rd_j t| + |pos_j|) with
// u = 2^-24 (three roundings in t_j(i), two in pos_j); the kernel uses the wider margin
// 2^-20*(|rd_j t| + |pos_j| + 1).  Within the margin only the nearest plane is in doubt and
// it is settled by comparing t_j(i) with t_exit directly, with the reference's tie rule.
//
// Empty boxes come from the cell records: for the ray's direction octant, distance d
// means the d^3 cells that start at the current cell and extend in the ray's direction
// are empty.  A step inside a non-empty cell is a plain unit step that tests the cell's
// 64-bit mask from registers.  Anchors are kept window relative (u = a - lower), so the
// loop test lower < a + s01 < upper is one unsigned range compare per axis.

// New window-relative anchor on a non-exit axis after leaving the box at (tk, k).
// Branch free: every lane evaluates the plane in doubt and selects.
VRT_DEV int axis_after_exit(float ro, float rd, float inv, int lo, int s01, int u, int bl, int bh, float tk,
                            bool inclusive)
{
    const float rt   = rd * tk;
    const float pos  = ro + rt;
    const float i0   = rintf(pos);
    const float eps  = 0x1.0p-20f * ((fabsf(rt) + fabsf(pos)) + 1.0f);
    const float tt   = (i0 - ro) * inv;  // t_j of the nearest plane, as the reference computes it
    const bool  crossed = inclusive ? tt <= tk : tt < tk;
    const int   a_near  = f2i_small(i0) - ((crossed != (s01 != 0)) ? 1 : 0);
    const int   a_far   = f2i_small(floorf(pos));
    const int   a       = fabsf(pos - i0) <= eps ? a_near : a_far;
    int un = a - lo;
    un = min(max(un, bl), bh);
    return s01 ? max(un, u) : min(un, u);
}

template <typename T>
VRT_DEV void traverse_skip(const TraceParams &P, float rox, float roy, float roz, float rdx, float rdy,
                           float rdz, int lox, int loy, int loz, int upx, int upy, int upz, RayHit &H)
{
    const float invx = 1.0f / rdx, invy = 1.0f / rdy, invz = 1.0f / rdz;
    // The box algebra needs finite, non-zero directions and finite plane parameters.
    // Anything else takes the literal loop (NaN / inf semantics of the reference).
    const float big = 0x1.0p100f;
    const bool regular = (rdx != 0.0f) && (rdy != 0.0f) && (rdz != 0.0f) && (fabsf(invx) < big) &&
                         (fabsf(invy) < big) && (fabsf(invz) < big);
    if (!regular) {
        traverse_literal<T, false>(P, rox, roy, roz, rdx, rdy, rdz, lox, loy, loz, upx, upy, upz, H);
        return;
    }
    const int s01x = rdx > 0.0f ? 1 : 0, s01y = rdy > 0.0f ? 1 : 0, s01z = rdz > 0.0f ? 1 : 0;
    const int s11x = 2 * s01x - 1, s11y = 2 * s01y - 1, s11z = 2 * s01z - 1;
    const int n = P.n, nc = P.nc;
    // texel of a window-relative anchor: u + off (mod n), off = rem_euclid(lower, n), a multiple of 4
    const int offx = rem_euclid(lox, n), offy = rem_euclid(loy, n), offz = rem_euclid(loz, n);
    // loop test lower < a + s01 < upper  <=>  alo <= u <= ahi
    const int alox = 1 - s01x, aloy = 1 - s01y, aloz = 1 - s01z;
    const unsigned span = (unsigned)(n - 3);  // ahi - alo
    const int ahix = alox + (n - 3), ahiy = aloy + (n - 3), ahiz = aloz + (n - 3);
    // world index of the next plane on each axis = u + pl
    const int plx = lox + s01x, ply = loy + s01y, plz = loz + s01z;
    // byte of the record's distance word that belongs to this ray's octant
    const int oct = s01x | (s01y << 1) | (s01z << 2);
    const int dshift = 8 * (oct & 3);
    const bool dhigh = oct >= 4;
    // the box grows only forward: r cells on the side the ray moves to, none behind
    const int mlx = 1 - s01x, mhx = s01x, mly = 1 - s01y, mhy = s01y, mlz = 1 - s01z, mhz = s01z;

    int ux = f2i_small(floorf(rox)) - lox, uy = f2i_small(floorf(roy)) - loy, uz = f2i_small(floorf(roz)) - loz;
    unsigned steps = 0, iters = 0;
    H.hit = 0; H.capped = 0; H.block_id = 0; H.k = 0; H.sk = 0;
    int      cur_cell = -1, d = 0;
    uint64_t cur_mask = 0;
    auto load_cell = [&](int cell) {
        const uint4 r = P.cells[cell];
        cur_cell = cell;
        cur_mask = (uint64_t)r.x | ((uint64_t)r.y << 32);
        d        = (int)(((dhigh ? r.w : r.z) >> dshift) & 0xffu);
    };
    auto cell_of = [&](int vx, int vy, int vz) {
        int qx = vx + offx; qx -= qx >= n ? n : 0;
        int qy = vy + offy; qy -= qy >= n ? n : 0;
        int qz = vz + offz; qz -= qz >= n ? n : 0;
        return ((qz >> 2) * nc + (qy >> 2)) * nc + (qx >> 2);
    };
    if ((unsigned)(ux - alox) <= span && (unsigned)(uy - aloy) <= span && (unsigned)(uz - aloz) <= span)
        load_cell(cell_of(ux, uy, uz));  // the start voxel's cell (the voxel itself is never tested)
    for (;;) {
        if ((unsigned)(ux - alox) > span || (unsigned)(uy - aloy) > span || (unsigned)(uz - aloz) > span) break;
        if (steps >= P.max_steps) { H.capped = 1; break; }
        int   k;
        float tk;
        bool  moved_cell;
        if (d == 0 || steps + (unsigned)(12 * d) > P.max_steps) {
            // unit step (raytrace.glsl:45-50 with a zero distance)
            iters += P.debug_iters == 2 ? 0u : 0x10001u;
            const float tx = ((float)(ux + plx) - rox) * invx;
            const float ty = ((float)(uy + ply) - roy) * invy;
            const float tz = ((float)(uz + plz) - roz) * invz;
            k  = argmin3(tx, ty, tz);
            tk = k == 0 ? tx : (k == 1 ? ty : tz);
            const int oldu = k == 0 ? ux : (k == 1 ? uy : uz);
            const int newu = oldu + (k == 0 ? s11x : (k == 1 ? s11y : s11z));
            ux = k == 0 ? newu : ux; uy = k == 1 ? newu : uy; uz = k == 2 ? newu : uz;
            moved_cell = ((oldu ^ newu) & ~3) != 0;
            steps += 1;
        } else {
            iters += P.debug_iters == 2 ? (d == 1 ? 1u : (d == 2 ? 0x100u : (d <= 4 ? 0x10000u : 0x1000000u))) : 1u;
            const int r = d - 1;
            const int blx = max(((ux >> 2) - r * mlx) * 4, alox), bhx = min(((ux >> 2) + r * mhx) * 4 + 3, ahix);
            const int bly = max(((uy >> 2) - r * mly) * 4, aloy), bhy = min(((uy >> 2) + r * mhy) * 4 + 3, ahiy);
            const int blz = max(((uz >> 2) - r * mlz) * 4, aloz), bhz = min(((uz >> 2) + r * mhz) * 4 + 3, ahiz);
            // first anchor outside the box on each axis, and the plane crossed to get there
            const int nux = s01x ? bhx + 1 : blx - 1, nuy = s01y ? bhy + 1 : bly - 1, nuz = s01z ? bhz + 1 : blz - 1;
            const float tx = ((float)(nux + (lox + 1 - s01x)) - rox) * invx;
            const float ty = ((float)(nuy + (loy + 1 - s01y)) - roy) * invy;
            const float tz = ((float)(nuz + (loz + 1 - s01z)) - roz) * invz;
            k  = argmin3(tx, ty, tz);
            tk = k == 0 ? tx : (k == 1 ? ty : tz);
            // axis j's events sort before the exit event of axis k on ties iff j < k
            const int wx = axis_after_exit(rox, rdx, invx, lox, s01x, ux, blx, bhx, tk, true);
            const int wy = axis_after_exit(roy, rdy, invy, loy, s01y, uy, bly, bhy, tk, k == 2);
            const int wz = axis_after_exit(roz, rdz, invz, loz, s01z, uz, blz, bhz, tk, false);
            const int vx = k == 0 ? nux : wx, vy = k == 1 ? nuy : wy, vz = k == 2 ? nuz : wz;
            steps += (unsigned)(abs(vx - ux) + abs(vy - uy) + abs(vz - uz));
            ux = vx; uy = vy; uz = vz;
            moved_cell = true;
        }
        // test the voxel just entered
        if (moved_cell) {
            const int cell = cell_of(ux, uy, uz);
            if (cell != cur_cell) load_cell(cell);
        }
        if (d == 0) {
            const int bit = ((uz & 3) << 4) | ((uy & 3) << 2) | (ux & 3);  // off is a multiple of 4
            if ((cur_mask >> bit) & 1ull) {
                H.hit = 1; H.k = k; H.sk = k == 0 ? s11x : (k == 1 ? s11y : s11z);
                H.t = tk;
                H.hx = rox + rdx * tk; H.hy = roy + rdy * tk; H.hz = roz + rdz * tk;
                int qx = ux + offx; qx -= qx >= n ? n : 0;
                int qy = uy + offy; qy -= qy >= n ? n : 0;
                int qz = uz + offz; qz -= qz >= n ? n : 0;
                H.block_id = (unsigned)((const T *)P.block)[((size_t)qz * (size_t)n + (size_t)qy) * (size_t)n + (size_t)qx];
                break;
            }
        }
    }
    H.vx = ux + lox; H.vy = uy + loy; H.vz = uz + loz; H.steps = steps; H.iters = iters;
    if (!H.hit) { H.t = -1.0f; H.hx = H.hy = H.hz = 0.0f; }
}

// ---------------------------------------------------------------------------
// per-pixel prologue / epilogue shared by both kernels
// ---------------------------------------------------------------------------
VRT_DEV bool pixel_of_thread(const TraceParams &P, unsigned &col, unsigned &q, unsigned &row)
{
    const unsigned tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    col = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    q   = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    if (col >= P.w || q >= P.rows.rows) return false;
    if (P.rows.strip_h == 0) {
        row = P.rows.y0 + q;
    } else {
        row = ((q / P.rows.strip_h) * P.rows.parts + P.rows.part) * P.rows.strip_h + q % P.rows.strip_h;
    }
    return true;
}

// raytrace.glsl:120-123
VRT_DEV void raygen(const TraceParams &P, unsigned gx, unsigned gy, float &rdx, float &rdy, float &rdz)
{
    const float fw = (float)P.w, fh = (float)P.h;
    const float ncx = (float)gx / fw - 0.5f;
    const float ncy = (float)gy / fh - (fh / fw * 0.5f);
    const float *m = P.m;
    const float d0 = ((ncx * m[0] + ncy * m[1]) + 1.0f * m[2]) + 1.0f * m[3];
    const float d1 = ((ncx * m[4] + ncy * m[5]) + 1.0f * m[6]) + 1.0f * m[7];
    const float d2 = ((ncx * m[8] + ncy * m[9]) + 1.0f * m[10]) + 1.0f * m[11];
    const float len = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
    rdx = d0 / len; rdy = d1 / len; rdz = d2 / len;
}

// raytrace.glsl:129-133 + render.comp:9-13
VRT_DEV void finish_pixel(const TraceParams &P, unsigned gx, unsigned gy, size_t i, const RayHit &H)
{
    float r, g, b, a = 1.0f;
    if (H.hit) {
        const float nk = (float)(-H.sk);
        r = (H.k == 0 ? nk : 0.0f) + 0.5f * canon_sin(floorf(H.hx + 0.001f) * 2.0f);
        g = (H.k == 1 ? nk : 0.0f) + 0.5f * canon_sin(floorf(H.hy + 0.001f) * 2.0f);
        b = (H.k == 2 ? nk : 0.0f) + 0.5f * canon_sin(floorf(H.hz + 0.001f) * 2.0f);
    } else if (H.steps >= 1000u) {  // raytrace.glsl:73-75
        r = 0.2f; g = 1.0f; b = 0.8f;
    } else {
        r = g = b = 0.0f;
    }
    const float dx = (float)gx - (float)(int)(P.w / 2);
    const float dy = (float)gy - (float)(int)(P.h / 2);
    if (sqrtf(dx * dx + dy * dy) < 1.2f) { r = 1.0f; g = 0.3f; b = 0.3f; a = 1.0f; }  // render.comp:10-12

    if (P.bgra8) {
        const unsigned px = unorm8(b) | (unorm8(g) << 8) | (unorm8(r) << 16) | (unorm8(a) << 24);
        ((unsigned *)P.bgra8)[i] = px;
    }
    if (P.voxel) { P.voxel[3 * i] = H.vx; P.voxel[3 * i + 1] = H.vy; P.voxel[3 * i + 2] = H.vz; }
    if (P.face) P.face[i] = H.hit ? 2 * H.k + (H.sk < 0 ? 1 : 0) : (H.capped ? -2 : -1);
    if (P.steps) P.steps[i] = P.debug_iters ? H.iters : H.steps;
    if (P.t) P.t[i] = H.t;
    if (P.hit_point) { P.hit_point[3 * i] = H.hx; P.hit_point[3 * i + 1] = H.hy; P.hit_point[3 * i + 2] = H.hz; }
    if (P.rgba) { P.rgba[4 * i] = r; P.rgba[4 * i + 1] = g; P.rgba[4 * i + 2] = b; P.rgba[4 * i + 3] = a; }
}

template <typename T, bool HAS_DIST>
__global__ __launch_bounds__(256) void k_trace_literal(const TraceParams P)
{
    unsigned col, q, row;
    if (!pixel_of_thread(P, col, q, row)) return;
    const unsigned gx = col, gy = P.h - row;  // render.comp:13
    float rdx, rdy, rdz;
    raygen(P, gx, gy, rdx, rdy, rdz);
    const float rox = P.m[12], roy = P.m[13], roz = P.m[14];  // raytrace.glsl:16
    int lox, loy, loz, upx, upy, upz;
    loaded_area(P, rox, lox, upx); loaded_area(P, roy, loy, upy); loaded_area(P, roz, loz, upz);
    RayHit H;
    traverse_literal<T, HAS_DIST>(P, rox, roy, roz, rdx, rdy, rdz, lox, loy, loz, upx, upy, upz, H);
    finish_pixel(P, gx, gy, (size_t)q * P.w + col, H);
}

template <typename T>
__global__ __launch_bounds__(256) void k_trace_skip(const TraceParams P)
{
    unsigned col, q, row;
    if (!pixel_of_thread(P, col, q, row)) return;
    const unsigned gx = col, gy = P.h - row;
    float rdx, rdy, rdz;
    raygen(P, gx, gy, rdx, rdy, rdz);
    const float rox = P.m[12], roy = P.m[13], roz = P.m[14];
    int lox, loy, loz, upx, upy, upz;
    loaded_area(P, rox, lox, upx); loaded_area(P, roy, loy, upy); loaded_area(P, roz, loz, upz);
    RayHit H;
    traverse_skip<T>(P, rox, roy, roz, rdx, rdy, rdz, lox, loy, loz, upx, upy, upz, H);
    finish_pixel(P, gx, gy, (size_t)q * P.w + col, H);
}

static dim3 trace_grid(const TraceParams &p) { return dim3((p.w + 15) / 16, (p.rows.rows + 15) / 16, 1); }

hipError_t launch_trace_literal(const TraceParams &p, int voxel_bytes, hipStream_t s)
{
    if (p.rows.rows == 0 || p.w == 0) return hipSuccess;
    const dim3 grid = trace_grid(p), block(256);
    if (voxel_bytes == 1) {
        k_trace_literal<uint8_t, false><<<grid, block, 0, s>>>(p);
    } else if (p.dist) {
        k_trace_literal<uint16_t, true><<<grid, block, 0, s>>>(p);
    } else {
        k_trace_literal<uint16_t, false><<<grid, block, 0, s>>>(p);
    }
    return hipGetLastError();
}

hipError_t launch_trace_skip(const TraceParams &p, int voxel_bytes, hipStream_t s)
{
    if (p.rows.rows == 0 || p.w == 0) return hipSuccess;
    const dim3 grid = trace_grid(p), block(256);
    if (voxel_bytes == 1) k_trace_skip<uint8_t><<<grid, block, 0, s>>>(p);
    else k_trace_skip<uint16_t><<<grid, block, 0, s>>>(p);
    return hipGetLastError();
}

}  // namespace vrt
