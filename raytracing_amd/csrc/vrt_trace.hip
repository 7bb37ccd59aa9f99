// vrt_trace.hip -- primary-ray kernels for gfx950.
//
// One thread per pixel; a wavefront covers a 16x4-pixel tile, so its 64 rays stay spatially
// coherent; a 256-thread workgroup stacks four such tiles.
//
//   k_trace_literal   : one voxel per iteration, straight restatement of
//                       raytrace.glsl:26-77 (reads the linear block / distance images).
//   k_trace_skip      : the same visit order reproduced exactly, but empty space is crossed
//                       in one step per empty box (traverse_cells: two-phase loop; traverse_skip:
//                       one step per trip, kept for callers that set a small step cap).  Reads one
//                       16-byte record per 4^3 cell visited (occupancy mask + octant distances)
//                       instead of one voxel per step; the voxel image is read once, at the hit.
//   k_trace_secondary : the primary ray and, per hit, a shadow ray and four ambient-occlusion
//                       rays through the same traversal (no reference counterpart).
//
// The skip kernels also choose the order in which their workgroup rows are dispatched from
// what earlier launches measured (see "dispatch order of workgroup rows" below).
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
    unsigned work;   // trips of the two-phase loop this lane was live for (0 on the other paths)
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
    H.vx = ax; H.vy = ay; H.vz = az; H.steps = steps; H.iters = steps; H.work = 0;
    if (!H.hit) { H.t = -1.0f; H.hx = H.hy = H.hz = 0.0f; }
}

// The literal loop for a ray whose direction has a zero component, which the box algebra below cannot
// take as it is: 1 / 0 puts infinities and NaNs into the plane parameters (raytrace.glsl:27,45).  What
// the reference's arithmetic does with them, per zero axis k (s01 = 0, s11 = -1, cp_k = anchor_k):
//   t_k = (cp_k - ro_k) * (+-inf) is -inf, +inf or NaN (cp_k == ro_k).
//   +inf, or NaN on y or z, never wins argmin (util.glsl:28-42) and the position on axis k stays ro_k:
//        the axis is "passive", the anchor keeps its coordinate and the ray is an ordinary ray in the
//        other axes.
//   -inf, or NaN on x, wins: the position becomes NaN, line :50 sets anchor_k = f2i(NaN) - 1 = -1, and
//        the next t_k is taken from there.  Either the axis is passive then, or the same voxel is tested
//        again and again until the step cap.
// So the loop runs literally, and ends early in two provable ways: when an iteration leaves cp as it
// was, every later iteration repeats it and the ray ends at the step cap; when every zero axis has
// become passive (and the others are regular), the rest of the ray is handed to the box traversal,
// frozen on the zero axes (CellRay::resume).  Returns true on such a hand-over, with the anchor and the
// iterations made so far; false when the ray ended here and H holds its result.
template <typename T>
VRT_DEV bool traverse_degenerate(const TraceParams &P, float rox, float roy, float roz, float rdx, float rdy,
                                 float rdz, int lox, int loy, int loz, int upx, int upy, int upz, RayHit &H,
                                 int &ax, int &ay, int &az, unsigned &steps)
{
    const float invx = 1.0f / rdx, invy = 1.0f / rdy, invz = 1.0f / rdz;          // :27
    const int   s01x = rdx > 0.0f ? 1 : 0, s01y = rdy > 0.0f ? 1 : 0, s01z = rdz > 0.0f ? 1 : 0;  // :29
    const int   s11x = 2 * s01x - 1, s11y = 2 * s01y - 1, s11z = 2 * s01z - 1;     // :30
    const float big = 0x1.0p100f;
    const bool  zx = rdx == 0.0f, zy = rdy == 0.0f, zz = rdz == 0.0f;
    // every axis either zero or fit for the box algebra, and at least one of the latter
    const bool  can_resume = (zx || fabsf(invx) < big) && (zy || fabsf(invy) < big) && (zz || fabsf(invz) < big) &&
                             !(zx && zy && zz);
    ax = f2i(floorf(rox)); ay = f2i(floorf(roy)); az = f2i(floorf(roz));           // :32
    int cpx = wadd(ax, s01x), cpy = wadd(ay, s01y), cpz = wadd(az, s01z);          // :35, zero distance
    steps = 0;
    H.hit = 0; H.capped = 0; H.block_id = 0; H.k = 0; H.sk = 0;
    for (;;) {
        if (can_resume && steps <= 8u) {
            const float fx = ((float)cpx - rox) * invx, fy = ((float)cpy - roy) * invy, fz = ((float)cpz - roz) * invz;
            const float inf = __builtin_inff();
            if ((!zx || fx == inf) && (!zy || !(fy < inf)) && (!zz || !(fz < inf))) return true;
        }
        if (!(lox < cpx && loy < cpy && loz < cpz && cpx < upx && cpy < upy && cpz < upz)) break;  // :43
        if (steps >= P.max_steps) { H.capped = 1; break; }
        ++steps;                                                                    // :44
        const float tx = ((float)cpx - rox) * invx;                                 // :45
        const float ty = ((float)cpy - roy) * invy;
        const float tz = ((float)cpz - roz) * invz;
        const int   k  = argmin3(tx, ty, tz);                                       // :46
        const float tk = k == 0 ? tx : (k == 1 ? ty : tz);
        const float px = rox + rdx * tk, py = roy + rdy * tk, pz = roz + rdz * tk;
        const float pk  = k == 0 ? px : (k == 1 ? py : pz);
        const int   s01 = k == 0 ? s01x : (k == 1 ? s01y : s01z);
        const int   ak  = wadd(wsub(f2i(rintf(pk)), 1), s01);                       // :50 (:49 keeps the anchor, see traverse_literal)
        if (k == 0) ax = ak; else if (k == 1) ay = ak; else az = ak;
        const unsigned b = load_block<T>(P, ax, ay, az);                            // :52
        if (b > 0) {                                                                // :53-60
            H.hit = 1; H.block_id = b; H.k = k; H.sk = k == 0 ? s11x : (k == 1 ? s11y : s11z);
            H.t = tk; H.hx = px; H.hy = py; H.hz = pz;
            break;
        }
        const int ncx = wadd(ax, s01x), ncy = wadd(ay, s01y), ncz = wadd(az, s01z);  // :67
        if (ncx == cpx && ncy == cpy && ncz == cpz) {
            // nothing changed and the voxel is air: the loop repeats this iteration up to the cap
            steps = P.max_steps; H.capped = 1;
            break;
        }
        cpx = ncx; cpy = ncy; cpz = ncz;
    }
    H.vx = ax; H.vy = ay; H.vz = az; H.steps = steps; H.iters = steps; H.work = 0;
    if (!H.hit) { H.t = -1.0f; H.hx = H.hy = H.hz = 0.0f; }
    return false;
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
// it is settled by comparing t_j(i) with t_exit directly, with the reference's tie rule
// (rare: taken by a wavefront only when one of its lanes is that close to a plane).
//
// Direction-normalised coordinates.  Per axis the kernel works in a mirrored frame in
// which the ray always moves towards +: ro' = s*ro, rd' = |rd|, inv' = |inv| and the
// window-relative anchor u = a - lower becomes w = u (s > 0) or ~u = -u-1 (s < 0).  Voxel w
// spans [w + ln, w + ln + 1] in the mirrored frame (ln = s*lower), the next plane is
// w + ln + 1, a unit step is w += 1 and a cell boundary is crossed when (w & 3) == 0.
// Negation is exact in IEEE arithmetic and commutes with every rounding, so
// ((float)P' - ro') * inv' has the bits of the reference's ((float)P - ro) * inv -- except for
// the sign of an exact zero, which no comparison sees; the reported t is recomputed in the
// original frame.
//
// Empty boxes come from the cell records (vrt_kernels.h): the octant distance d says the
// d^3 cells that start at the current cell and extend in the ray's direction are empty;
// for empty cells the record also names the largest aligned block of cell columns whose
// occupied layers all lie below the cell (the column-top pyramid), which gives a box as
// wide as that block and as high as the window.

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
    const int n = P.n, nc = P.nc;
    const bool px = rdx > 0.0f, py = rdy > 0.0f, pz = rdz > 0.0f;
    const int  smx = px ? 0 : -1, smy = py ? 0 : -1, smz = pz ? 0 : -1;            // u = w ^ sm
    const float rnx = px ? rox : -rox, rny = py ? roy : -roy, rnz = pz ? roz : -roz;  // mirrored origin
    const float dnx = fabsf(rdx), dny = fabsf(rdy), dnz = fabsf(rdz);
    const float inx = fabsf(invx), iny = fabsf(invy), inz = fabsf(invz);
    const int  lnx = px ? lox : -lox, lny = py ? loy : -loy, lnz = pz ? loz : -loz;
    // loop test lower < a + s01 < upper  <=>  wlo <= w <= wlo + n - 3
    const int  wlox = px ? 0 : 1 - n, wloy = py ? 0 : 1 - n, wloz = pz ? 0 : 1 - n;
    const int  whix = wlox + (n - 3), whiy = wloy + (n - 3), whiz = wloz + (n - 3);
    const unsigned span = (unsigned)(n - 3);
    // texel of an anchor: u + off (mod n), off = rem_euclid(lower, n), a multiple of 4
    const int  offx = P.off[0], offy = P.off[1], offz = P.off[2];
    // byte of the record's distance word that belongs to this ray's octant
    const int  oct = (px ? 1 : 0) | (py ? 2 : 0) | (pz ? 4 : 0);
    const int  dshift = 8 * (oct & 3);
    const bool dhigh = oct >= 4;

    int wx = (f2i_small(floorf(rox)) - lox) ^ smx;
    int wy = (f2i_small(floorf(roy)) - loy) ^ smy;
    int wz = (f2i_small(floorf(roz)) - loz) ^ smz;
    unsigned steps = 0, iters = 0;
    H.hit = 0; H.capped = 0; H.block_id = 0; H.k = 0; H.sk = 0;
    int      cur_cell = -1, d = 0;
    uint64_t cur_mask = 0;   // non-empty cells: occupancy
    int qx = 0, qy = 0, qz = 0;  // texel of the anchor at the last cell lookup
    const bool pow2 = P.n_pow2_mask >= 0;  // wave uniform
    auto locate = [&]() {
        qx = (wx ^ smx) + offx; qy = (wy ^ smy) + offy; qz = (wz ^ smz) + offz;
        if (pow2) {
            qx &= P.n_pow2_mask; qy &= P.n_pow2_mask; qz &= P.n_pow2_mask;
        } else {
            qx -= qx >= n ? n : 0; qy -= qy >= n ? n : 0; qz -= qz >= n ? n : 0;
        }
        const int cell = ((qz >> 2) * nc + (qy >> 2)) * nc + (qx >> 2);
        if (cell != cur_cell) {
            const uint4 r = P.cells[CELL_U4 * (size_t)cell];
            cur_cell = cell;
            cur_mask = (uint64_t)r.x | ((uint64_t)r.y << 32);
            d        = (int)(((dhigh ? r.w : r.z) >> dshift) & 0xffu);
        }
    };
    if ((unsigned)(wx - wlox) <= span && (unsigned)(wy - wloy) <= span && (unsigned)(wz - wloz) <= span)
        locate();  // the start voxel's cell (the voxel itself is never tested)
    for (;;) {
        if ((unsigned)(wx - wlox) > span || (unsigned)(wy - wloy) > span || (unsigned)(wz - wloz) > span) break;
        if (P.need_cap && steps >= P.max_steps) { H.capped = 1; break; }
        int   k;
        float tk;
        bool  moved_cell;
        bool  unit = d == 0;
        int   fwx = wx, fwy = wy, fwz = wz;  // last anchor of the empty box on each axis
        if (!unit) {
            const int r4 = (d - 1) << 2;  // the octant cube (this form does not use the column-top pyramid)
            fwx = min((wx | 3) + r4, whix);
            fwy = min((wy | 3) + r4, whiy);
            fwz = min((wz | 3) + r4, whiz);
            // the box may not carry the ray past the step cap (only with tiny caps)
            if (P.need_cap && steps + (unsigned)((fwx - wx) + (fwy - wy) + (fwz - wz) + 3) > P.max_steps) {
                unit = true; fwx = wx; fwy = wy; fwz = wz;
            }
        }
        // The box code below also performs a unit step when the box is the single voxel
        // (fw = w), so a wavefront runs it whenever any of its lanes has a box to cross and
        // the cheaper unit-step code only when none has.
        if (__any(unit ? 0 : 1)) {
#ifdef VRT_DIAG_ITERS
            iters += unit ? 0x10001u : 1u;  // low half: loop iterations, high half: unit steps among them
#endif
            // exit planes = lower faces of the first anchors outside the box
            const float tx = ((float)(fwx + lnx + 1) - rnx) * inx;
            const float ty = ((float)(fwy + lny + 1) - rny) * iny;
            const float tz = ((float)(fwz + lnz + 1) - rnz) * inz;
            k  = argmin3(tx, ty, tz);
            tk = k == 0 ? tx : (k == 1 ? ty : tz);
            // anchors at the exit from the exit point, where it is clear of every plane
            const float rtx = dnx * tk, rty = dny * tk, rtz = dnz * tk;
            const float psx = rnx + rtx, psy = rny + rty, psz = rnz + rtz;
            const float i0x = rintf(psx), i0y = rintf(psy), i0z = rintf(psz);
            int vx = f2i_small(floorf(psx)) - lnx, vy = f2i_small(floorf(psy)) - lny, vz = f2i_small(floorf(psz)) - lnz;
            const bool nearx = fabsf(psx - i0x) <= 0x1.0p-20f * ((rtx + fabsf(psx)) + 1.0f);
            const bool neary = fabsf(psy - i0y) <= 0x1.0p-20f * ((rty + fabsf(psy)) + 1.0f);
            const bool nearz = fabsf(psz - i0z) <= 0x1.0p-20f * ((rtz + fabsf(psz)) + 1.0f);
            if (__any((!unit && (nearx || neary || nearz)) ? 1 : 0)) {
                // a plane within the margin: has the reference crossed it before the exit event?
                // axis j's events sort before the exit event of axis k on ties iff j < k
                const float ttx = (i0x - rnx) * inx, tty = (i0y - rny) * iny, ttz = (i0z - rnz) * inz;
                const bool  cx = ttx <= tk;
                const bool  cy = k == 2 ? tty <= tk : tty < tk;
                const bool  cz = ttz < tk;
                vx = nearx ? f2i_small(i0x) - lnx - (cx ? 0 : 1) : vx;
                vy = neary ? f2i_small(i0y) - lny - (cy ? 0 : 1) : vy;
                vz = nearz ? f2i_small(i0z) - lnz - (cz ? 0 : 1) : vz;
            }
            vx = k == 0 ? fwx + 1 : min(max(vx, wx), fwx);
            vy = k == 1 ? fwy + 1 : min(max(vy, wy), fwy);
            vz = k == 2 ? fwz + 1 : min(max(vz, wz), fwz);
            steps += (unsigned)((vx - wx) + (vy - wy) + (vz - wz));
            wx = vx; wy = vy; wz = vz;
            const int wk = k == 0 ? wx : (k == 1 ? wy : wz);
            moved_cell = !unit || (wk & 3) == 0;
        } else {
            // unit step (raytrace.glsl:45-50 with a zero distance)
#ifdef VRT_DIAG_ITERS
            iters += 0x10001u;
#endif
            const float tx = ((float)(wx + lnx + 1) - rnx) * inx;
            const float ty = ((float)(wy + lny + 1) - rny) * iny;
            const float tz = ((float)(wz + lnz + 1) - rnz) * inz;
            k  = argmin3(tx, ty, tz);
            tk = k == 0 ? tx : (k == 1 ? ty : tz);
            wx += k == 0 ? 1 : 0; wy += k == 1 ? 1 : 0; wz += k == 2 ? 1 : 0;
            const int wk = k == 0 ? wx : (k == 1 ? wy : wz);
            moved_cell = (wk & 3) == 0;
            steps += 1;
        }
        // test the voxel just entered
        if (moved_cell) locate();
        if (d == 0) {
            const int bit = (((wz ^ smz) & 3) << 4) | (((wy ^ smy) & 3) << 2) | ((wx ^ smx) & 3);
            if ((cur_mask >> bit) & 1ull) {
                H.hit = 1; H.k = k; H.sk = k == 0 ? (px ? 1 : -1) : (k == 1 ? (py ? 1 : -1) : (pz ? 1 : -1));
                // t as the reference computes it, in the unmirrored frame: the mirrored value has the
                // same bits except for the sign of an exact zero ((x - x) is +0 in either frame)
                const int   pk = k == 0 ? (wx ^ smx) + lox + (px ? 0 : 1)
                               : (k == 1 ? (wy ^ smy) + loy + (py ? 0 : 1) : (wz ^ smz) + loz + (pz ? 0 : 1));
                const float th = ((float)pk - (k == 0 ? rox : (k == 1 ? roy : roz))) * (k == 0 ? invx : (k == 1 ? invy : invz));
                H.t = th;
                H.hx = rox + rdx * th; H.hy = roy + rdy * th; H.hz = roz + rdz * th;
                locate();
                H.block_id = (unsigned)((const T *)P.block)[((size_t)qz * (size_t)n + (size_t)qy) * (size_t)n + (size_t)qx];
                break;
            }
        }
    }
    H.vx = (wx ^ smx) + lox; H.vy = (wy ^ smy) + loy; H.vz = (wz ^ smz) + loz;
    H.steps = steps; H.iters = iters; H.work = 0;
    if (!H.hit) { H.t = -1.0f; H.hx = H.hy = H.hz = 0.0f; }
}

// ---------------------------------------------------------------------------
// exact empty-space skipping, two-phase loop (the form the kernels run unless the caller asked
// for a step cap below 3n, which only traverse_skip above implements)
// ---------------------------------------------------------------------------
// Same boxes, same arithmetic and therefore the same visit order as traverse_skip, arranged
// so that a wavefront does not pay for the box algebra on every voxel of a non-empty cell.
// Each trip of the outer loop has two phases:
//   A  lanes standing in an empty cell, or in a non-empty cell above all solid voxels of the cell
//      columns around them, cross their box (one exact multi-voxel step) and look up the cell
//      they land in;
//   B  lanes standing in a non-empty cell at or below its solid voxels take up to two unit steps
//      (the reference's own step) against the 64-bit occupancy mask held in registers: about a
//      quarter of the instructions of phase A each and no memory access until the walk leaves the
//      cell.  The bound keeps a long walk of one lane from stalling the boxes of the other lanes.
// Every step moves towards + in the mirrored frame, so only the stepped axis can leave the
// window and only its upper bound needs a test; the voxel just entered is tested for a hit
// before that, as the reference does (raytrace.glsl:52 comes before the next :43).
// The ray as a resumable object: begin() sets it up, trip() runs one trip of the loop for the lanes
// whose ray is live, finish() turns the end state into the reference's outputs.  traverse_cells
// below runs one ray per lane to the end; the stream kernel keeps lanes busy by starting a new ray
// in a lane whose ray has ended.
#ifndef VRT_WALK_STEPS
#define VRT_WALK_STEPS 2  // unit steps per trip of a lane without a box (measured: 1, 3, 4 and 6 are no better)
#endif
// POW2: 0 = any grid (texel wrap by subtraction, 64-bit cell addresses); 1 = power-of-two grid up to 1024 (wrap by bit
// fields, 32-bit byte offsets); 2 = power-of-two grid above that (bit fields, 64-bit byte offsets)
template <typename T, int POW2>
struct CellRay {
    // per ray constants (mirrored frame)
    float rnx, rny, rnz, dnx, dny, dnz, inx, iny, inz, thr0x, thr0y, thr0z;
    int   lnx, lny, lnz, ln1x, ln1y, ln1z, smx, smy, smz;  // (ln1 = ln + 1 + PLANE_BIAS, see plane_float)
    unsigned psel;  // byte offset, in a cell, of the box word for this ray's direction of travel
    // state
    int      wx, wy, wz, w0, k, bit;
    unsigned pyr_lo, pyr_hi;  // the cell's box word for this ray's direction of travel
    unsigned e;               // that word's entry for the current voxel's level (BOX_NONE: no box, unit steps), kept beside `bit`
    unsigned work, iters;
    uint64_t cur_mask;
    bool     live, hit;
#ifdef VRT_DIAG_SEC_UTIL
    // diagnostic build: lanes x trips in phase A, in the first and second unit step of phase B, and trips in which
    // the wavefront ran phase A / phase B at all (wave uniform, counted by every lane)
    unsigned dA = 0, dB1 = 0, dB2 = 0, dAx = 0, dBx = 0, dNear = 0, dNearX = 0;  // lanes near a plane; trips that ran the settling code
#endif

    // float(i) of a plane coordinate |i| < 2^22 that arrives with PLANE_BIAS added: the biased integer is the bit pattern of
    // 1.5 * 2^23 + i, and taking 1.5 * 2^23 off is exact -- an integer add and a float subtract, both of which this chip
    // issues at twice the rate of the conversion.  (Plane coordinates are window coordinates: |i| <= 2^20 + 2 * 4096.)
    static constexpr int PLANE_BIAS = 0x4B400000;
    static VRT_DEV float plane_float(int biased) { return __int_as_float(biased) - 12582912.0f; }

    // entry of the box word for voxel level j (0..3): bytes 2j (rise) and 2j + 1 (side - 1) of the 64-bit word, cut out by
    // one byte permute into bits 0..7 and 16..23 (selector bytes: 0..3 = pyr_lo, 4..7 = pyr_hi, 0x0c = zero), so that
    // 4 * (side - 1) is one right shift.  The selector's constant half is OR-ed in as a literal of the instruction itself:
    // held in a vector register across the secondary-ray kernel's loop (what the compiler does with it when it sees a
    // multiply-add) the binned kernel's results came out wrong on the device.
    static constexpr unsigned NO_BOX = ((BOX_NONE >> 8) << 16) | (BOX_NONE & 0xffu);  // BOX_NONE spread the same way
    VRT_DEV unsigned entry_of(unsigned j) const
    {
        unsigned sel;
        const unsigned t = 0x00020002u * j;
        asm("v_or_b32 %0, 0x0c010c00, %1" : "=v"(sel) : "v"(t));
        return __builtin_amdgcn_perm(pyr_hi, pyr_lo, sel);
    }
    VRT_DEV unsigned entry() const { return entry_of(((unsigned)bit >> 2) & 3u); }

    // occupancy of the current voxel in its cell's mask
    VRT_DEV bool mask_bit() const { return ((unsigned)(cur_mask >> (bit & 63)) & 1u) != 0; }

    VRT_DEV void locate(const TraceParams &P)
    {
        const int n = P.n, nc = P.nc;
        int qx = (wx ^ smx) + P.off[0], qy = (wy ^ smy) + P.off[1], qz = (wz ^ smz) + P.off[2];
        uint2 r, pw;  // the cell's occupancy mask (the record's first half), its box word for this ray
        if (POW2) {
            // The cell's coordinates are bit fields of the texel coordinates, which need no
            // wrap of their own then (the field extraction drops the bits above the grid, off is a multiple of 4), and
            // for n <= 1024 (POW2 == 1) the byte offset of the cell's 128 bytes stays below 2^32: one 32-bit offset on the
            // scalar base address, the column-top word of the ray's direction of travel at psel (16 .. 72) behind it.
            const unsigned L = (unsigned)P.nc_log2;
            const unsigned cx = __builtin_amdgcn_ubfe((unsigned)qx, 2u, L), cy = __builtin_amdgcn_ubfe((unsigned)qy, 2u, L),
                           cz = __builtin_amdgcn_ubfe((unsigned)qz, 2u, L);
            const unsigned cell = (((cz << L) | cy) << L) | cx;
            const char *base = reinterpret_cast<const char *>(P.cells);
            if (POW2 == 1) {
                const unsigned at = cell << CELL_SHIFT;
                r  = *reinterpret_cast<const uint2 *>(base + at);
                pw = *reinterpret_cast<const uint2 *>(base + (at | psel));
            } else {
                const size_t at = (size_t)cell << CELL_SHIFT;
                r  = *reinterpret_cast<const uint2 *>(base + at);
                pw = *reinterpret_cast<const uint2 *>(base + at + psel);
            }
        } else {
            qx -= qx >= n ? n : 0; qy -= qy >= n ? n : 0; qz -= qz >= n ? n : 0;
            const size_t cell = (size_t)(unsigned)(((qz >> 2) * nc + (qy >> 2)) * nc + (qx >> 2));
            const char *rec = reinterpret_cast<const char *>(P.cells) + (cell << CELL_SHIFT);  // the record, then the column-top words
            r  = *reinterpret_cast<const uint2 *>(rec);
            pw = *reinterpret_cast<const uint2 *>(rec + psel);
        }
        cur_mask = (uint64_t)r.x | ((uint64_t)r.y << 32);
        // z * 16 + y * 4 + x of the voxel in its cell (off is a multiple of 4) in bits 0..5; what lies above them is
        // never looked at (mask_bit, entry)
        unsigned lvl = (unsigned)qy & 3u;
        asm("" : "+v"(lvl));  // (kept whole: the compiler would spread the shifts over the three fields, two instructions more)
        bit      = (int)(((((unsigned)qz << 2) | lvl) << 2) | ((unsigned)qx & 3u));
        pyr_lo = pw.x; pyr_hi = pw.y;
        e = entry_of(lvl);
    }

    // Starts (or takes over) a ray at anchor (ax, ay, az) after `steps` iterations of the reference's loop:
    // (floor(ro), 0) for an ordinary ray; the state traverse_degenerate hands over for a ray with zero
    // direction components, all of them passive.  Such an axis is frozen by its constants alone, trip()
    // is the same code: origin -2^30 with an infinite reciprocal puts every plane parameter of the axis
    // at +inf (never the argmin), the exit position -2^30 floors far below the anchor, which the clamp
    // to the box then keeps, and the margin test never fires.  The stored magnitude carries the sign
    // that lets finish() rebuild the direction bit for bit (sm = -1: rd = -dn).
    // (MAY_FREEZE = false: the caller knows that no component is zero)
    // (CAMERA: the origin is the launch's camera position: the margins of the exit-point test come from the host)
    template <bool MAY_FREEZE, bool CAMERA = false>
    VRT_DEV void begin(const TraceParams &P, float rox, float roy, float roz, float rdx, float rdy, float rdz,
                       int lox, int loy, int loz, int ax, int ay, int az, unsigned steps)
    {
        const int   n = P.n;
        const bool  zx = MAY_FREEZE && rdx == 0.0f, zy = MAY_FREEZE && rdy == 0.0f, zz = MAY_FREEZE && rdz == 0.0f;
        const bool  px = rdx > 0.0f, py = rdy > 0.0f, pz = rdz > 0.0f;
        const float far = -0x1.0p30f, never = 0x1.0p40f;
        hit = false; k = 0; work = 0; iters = 0;
        smx = px ? 0 : -1; smy = py ? 0 : -1; smz = pz ? 0 : -1;
        rnx = zx ? far : (px ? rox : -rox); rny = zy ? far : (py ? roy : -roy); rnz = zz ? far : (pz ? roz : -roz);
        dnx = zx ? -rdx : fabsf(rdx); dny = zy ? -rdy : fabsf(rdy); dnz = zz ? -rdz : fabsf(rdz);
        if (MAY_FREEZE) {
            inx = fabsf(1.0f / rdx); iny = fabsf(1.0f / rdy); inz = fabsf(1.0f / rdz);  // +inf on a frozen axis
        } else {  // ray_is_regular holds: the same bits, and the same instructions as in that test
            inx = fabsf(recip_regular(rdx)); iny = fabsf(recip_regular(rdy)); inz = fabsf(recip_regular(rdz));
        }
        // (the magnitudes as values of their own: left to itself the compiler keeps the signed originals in registers
        // across the loop and takes the magnitude again in every trip)
        asm volatile("" : "+v"(dnx), "+v"(dny), "+v"(dnz), "+v"(inx), "+v"(iny), "+v"(inz));
        lnx = px ? lox : -lox; lny = py ? loy : -loy; lnz = pz ? loz : -loz;
        ln1x = lnx + 1 + PLANE_BIAS; ln1y = lny + 1 + PLANE_BIAS; ln1z = lnz + 1 + PLANE_BIAS;
        // word = descending | negative x << 1 | negative z << 2, eight bytes each, behind the 16-byte record
        psel = 16u + ((unsigned)smy & 8u) + ((unsigned)smx & 16u) + ((unsigned)smz & 32u);
        // margin of the exit-point test, as 0.5 - VRT_MARGIN * (|pos| + |origin| + 1) = thr0 - VRT_MARGIN * |pos|
        thr0x = zx ? never : (CAMERA ? P.thr0[0] : 0.5f - VRT_MARGIN * (fabsf(rnx) + 1.0f));
        thr0y = zy ? never : (CAMERA ? P.thr0[1] : 0.5f - VRT_MARGIN * (fabsf(rny) + 1.0f));
        thr0z = zz ? never : (CAMERA ? P.thr0[2] : 0.5f - VRT_MARGIN * (fabsf(rnz) + 1.0f));
        wx = wsub(ax, lox) ^ smx; wy = wsub(ay, loy) ^ smy; wz = wsub(az, loz) ^ smz;
        // every step moves one axis by +1 in the mirrored frame: the step count is the distance from here
        w0 = wx + wy + wz - (int)steps;
        const unsigned span = (unsigned)(n - 3);
        // loop test lower < a + s01 < upper  <=>  wlo <= w <= wlo + n - 3, wlo = 0 (s > 0) or 1 - n (s < 0)
        live = (unsigned)(wx - (smx & (1 - n))) <= span && (unsigned)(wy - (smy & (1 - n))) <= span &&
               (unsigned)(wz - (smz & (1 - n))) <= span;
        pyr_lo = pyr_hi = 0; e = NO_BOX; cur_mask = 0; bit = 0;
        if (live) locate(P);  // the anchor's cell (the start voxel itself is never tested)
    }

#ifdef VRT_DIAG_STAGE
    // instruction-budget builds (tools/stage_budget.sh): a word that depends on everything begin() worked out
    VRT_DEV unsigned digest() const
    {
        return __float_as_uint(rnx) ^ __float_as_uint(rny) ^ __float_as_uint(rnz) ^ __float_as_uint(dnx) ^ __float_as_uint(dny) ^
               __float_as_uint(dnz) ^ __float_as_uint(inx) ^ __float_as_uint(iny) ^ __float_as_uint(inz) ^ __float_as_uint(thr0x) ^
               __float_as_uint(thr0y) ^ __float_as_uint(thr0z) ^ (unsigned)lnx ^ (unsigned)lny ^ (unsigned)lnz ^ (unsigned)smx ^
               ((unsigned)smy << 1) ^ ((unsigned)smz << 2) ^ psel ^ (unsigned)wx ^ ((unsigned)wy << 8) ^ ((unsigned)wz << 16) ^
               (unsigned)w0 ^ (unsigned)bit ^ pyr_lo ^ pyr_hi ^ e ^ (unsigned)cur_mask ^ (unsigned)(cur_mask >> 32) ^
               (live ? 1u : 0u);
    }
#endif

    // a ray that ended elsewhere: the loop has nothing to do for this lane
    VRT_DEV void idle() { live = false; hit = false; work = 0; iters = 0; wx = wy = wz = w0 = 0; e = NO_BOX; }

    // iterations of the reference's loop so far (every step moves one axis by +1 in the mirrored frame)
    VRT_DEV unsigned steps_done() const { return (unsigned)(wx + wy + wz - w0); }

    // One trip for the lanes with a live ray.  On entry the current voxel has been tested (or is the
    // start voxel) and lies inside the loaded area.
    // FOLD > 0: when fewer than FOLD of the lanes that make this trip have no box to cross (and some lane has one),
    // those lanes take their unit step inside phase A, as the crossing of a box of one voxel -- the same arithmetic,
    // for no instruction more than the wavefront issues anyway -- and phase B is not run in this trip, neither for
    // them nor for the lanes that phase A leaves without a box.  Among rays that have little to do with each other
    // a handful of lanes per trip need a unit step, and phase B for them alone costs the wavefront half a phase A.
    template <int FOLD = 0>
    VRT_DEV void trip(const TraceParams &P)
    {
        const int n = P.n;
        const int whix = (smx & (1 - n)) + (n - 3), whiy = (smy & (1 - n)) + (n - 3), whiz = (smz & (1 - n)) + (n - 3);
        work += 1u;
#ifdef VRT_DIAG_ITERS
        iters += 1u;
#endif
#ifdef VRT_DIAG_SEC_UTIL
        dA += e != NO_BOX ? 1u : 0u;
#endif
        const bool boxed = e != NO_BOX;
        bool fold = false;  // wave uniform
        if (FOLD > 0) {
            const unsigned walkers = (unsigned)__popcll(__ballot(boxed ? 0 : 1));
            fold = walkers != 0 && walkers < (unsigned)FOLD && __any(boxed ? 1 : 0);
        }
        if (boxed || fold) {
            // ---- phase A: cross an empty box that holds the current voxel: the column-top block of the
            // voxel's level (also inside a non-empty cell, above its solid voxels) or the octant cube
            int fwx, fwy, fwz;  // last anchor of the box on each axis
            if (FOLD > 0 && !boxed) {
                fwx = wx; fwy = wy; fwz = wz;  // the voxel itself
            } else {
                // The box starts at the ray's own cell and extends the way the ray travels (+ in the mirrored frame): `side`
                // cells along x and z, and from the voxel on `rise` voxels more along y, or to the top of the window (the
                // builder has chosen between the square of cell columns and the octant cube, k_voxel_pyramid).
                const int      r4 = (int)(e >> 14);  // 4 * (side - 1): the voxels across the box, less those of the first cell
                const unsigned rise = e & 0xffu;
                fwx = min((wx | 3) + r4, whix);
                fwz = min((wz | 3) + r4, whiz);
                fwy = rise == BOX_RISE_ALL ? whiy : min(wy + (int)rise, whiy);
            }
            // plane parameters of the box's far faces (ln1 = ln + 1: the plane behind anchor fw)
            const float tx = (plane_float(fwx + ln1x) - rnx) * inx;
            const float ty = (plane_float(fwy + ln1y) - rny) * iny;
            const float tz = (plane_float(fwz + ln1z) - rnz) * inz;
            // util.glsl:28-42 on finite values (an ordinary ray's, or +inf on a frozen axis): y wins over x only when
            // smaller, z over both only when smaller; the winner's value is the minimum
            const float txy = __builtin_fminf(tx, ty);
            const bool  cy = ty < tx, cz = tz < txy;
            const float tk = __builtin_fminf(txy, tz);
            k = cz ? 2 : (cy ? 1 : 0);
            const bool kx = !cy && !cz, ky = cy && !cz;
            // anchors at the exit from the exit point, where it is clear of every plane.  The point is
            // off by at most 2^-24 * (3|i - ro| + |rd t| + |pos|) < 2^-21 * (|pos| + |ro| + 1); a plane
            // within twice that is settled by comparing plane parameters instead (see traverse_skip).
            // (near: distance of the position's fraction from 0.5, plus the margin's share of |pos|, reaches
            // thr0 = 0.5 - margin * (|ro| + 1); a fraction that rounds to 1 counts as near either way)
            // (fused: this point is the kernel's own, not a value of the reference's; one rounding instead of two keeps it
            // inside the error bound above)
            const float psx = __builtin_fmaf(dnx, tk, rnx), psy = __builtin_fmaf(dny, tk, rny), psz = __builtin_fmaf(dnz, tk, rnz);
            const int vx = f2i_floor(psx) - lnx, vy = f2i_floor(psy) - lny, vz = f2i_floor(psz) - lnz;
            // (not on the exit axis: there the point lies on its plane by construction, and the anchor is known)
            const bool mx = fma_abs(psx, VRT_MARGIN, __builtin_amdgcn_fractf(psx) - 0.5f) >= thr0x;
            const bool my = fma_abs(psy, VRT_MARGIN, __builtin_amdgcn_fractf(psy) - 0.5f) >= thr0y;
            const bool mz = fma_abs(psz, VRT_MARGIN, __builtin_amdgcn_fractf(psz) - 0.5f) >= thr0z;
            const bool nearx = ((int)mx & (int)!kx) != 0, neary = ((int)my & (int)!ky) != 0, nearz = ((int)mz & (int)!cz) != 0;
#ifdef VRT_DIAG_SEC_UTIL
            dNear += (nearx || neary || nearz) ? 1u : 0u;
            dNearX += __any((nearx || neary || nearz) ? 1 : 0) ? 1u : 0u;
#endif
            // The exit axis lands one past the box: its computed anchor is the box's last or the one behind it (the
            // exit point lies on that plane up to rounding), so the clamp yields the last and the step is a carry --
            // added here, next to the comparisons it comes from, where it costs no instruction of its own.
            const int ox = wx, oy = wy, oz = wz;
            wx = clamp_ordered(vx, ox, fwx) + (kx ? 1 : 0);   // (a box's last anchor is not before the current one)
            wy = clamp_ordered(vy, oy, fwy) + (ky ? 1 : 0);
            wz = clamp_ordered(vz, oz, fwz) + (cz ? 1 : 0);
            if (((int)nearx | (int)neary | (int)nearz) != 0) {  // (skipped by the wavefront when no lane is near)
                // a plane within the margin: has the reference crossed it before the exit event?  Axis j's events
                // sort before the exit event of axis k on ties iff j < k.
                const float i0x = rintf(psx), i0y = rintf(psy), i0z = rintf(psz);
                const float ttx = (i0x - rnx) * inx, tty = (i0y - rny) * iny, ttz = (i0z - rnz) * inz;
                const bool  bx = ttx <= tk;
                const bool  by = cz ? tty <= tk : tty < tk;
                const bool  bz = ttz < tk;
                const int sx = f2i_small(i0x) - lnx - (bx ? 0 : 1), sy = f2i_small(i0y) - lny - (by ? 0 : 1),
                          sz = f2i_small(i0z) - lnz - (bz ? 0 : 1);
                wx = nearx ? clamp_ordered(sx, ox, fwx) : wx;
                wy = neary ? clamp_ordered(sy, oy, fwy) : wy;
                wz = nearz ? clamp_ordered(sz, oz, fwz) : wz;
            }
            // the voxel just entered is tested even when it lies outside the loaded area; the ray
            // ends after that test (raytrace.glsl:52 comes before the next :43).  Only the exit axis can have
            // left the area (the other two stay inside the box), so "some axis beyond its bound" is that test.
            locate(P);  // a box ends at a cell boundary on its exit axis
            if (mask_bit()) { hit = true; live = false; }  // (an empty cell's mask is zero)
            else if (((int)(wx > whix) | (int)(wy > whiy) | (int)(wz > whiz)) != 0) live = false;
        }
#ifdef VRT_DIAG_SEC_UTIL
        dB1 += (!fold && live && e == NO_BOX) ? 1u : 0u;
#endif
        if (!fold && live && e == NO_BOX) {
            // ---- phase B: a few unit steps through the non-empty cell(s)
            // change of the mask bit index (z*16 + y*4 + x of the unmirrored voxel) per unit step
            const int dbx = 1 | smx, dby = 4 ^ (smy & ~7), dbz = 16 ^ (smz & ~31);  // +-1, +-4, +-16
            for (int i = 0; i < VRT_WALK_STEPS; ++i) {
#ifdef VRT_DIAG_ITERS
                iters += 0x10000u;
#endif
#ifdef VRT_DIAG_SEC_UTIL
                dB2 += i;
#endif
                const float tx = (plane_float(wx + ln1x) - rnx) * inx;
                const float ty = (plane_float(wy + ln1y) - rny) * iny;
                const float tz = (plane_float(wz + ln1z) - rnz) * inz;
                const bool  cy = ty < tx, cz = tz < __builtin_fminf(tx, ty);  // util.glsl:28-42, as in phase A
                const bool  kx = !cy && !cz, ky = cy && !cz;
                k = cz ? 2 : (cy ? 1 : 0);
                const int nx = wx + (kx ? 1 : 0), ny = wy + (ky ? 1 : 0), nz = wz + (cz ? 1 : 0);
                wx = nx; wy = ny; wz = nz;
                const int nxy = ky ? ny : nx, wk = cz ? nz : nxy;  // (locals: selects, not branches)
                if ((wk & 3) == 0) locate(P);
                else { const int dxy = ky ? dby : dbx; bit += cz ? dbz : dxy; e = entry(); }
                if (mask_bit()) { hit = true; live = false; break; }
                if (((int)(nx > whix) | (int)(ny > whiy) | (int)(nz > whiz)) != 0) { live = false; break; }
                if (e != NO_BOX) break;  // an empty box again: back to phase A
            }
        }
    }

    // outputs of a ray that has ended (live == false) after a successful begin().  The direction and its
    // reciprocal are rebuilt from their magnitudes and signs (negation is exact), so the caller's copies
    // need not stay in registers across the loop.
    VRT_DEV void finish(const TraceParams &P, float rox, float roy, float roz, int lox, int loy, int loz, RayHit &H) const
    {
        // (members copied to locals: three-way choices between locals compile to selects, between members to branches)
        const int   sx = smx, sy = smy, sz = smz, kk = k;
        const float ax_in = inx, ay_in = iny, az_in = inz;
        const float rdx = sx ? -dnx : dnx, rdy = sy ? -dny : dny, rdz = sz ? -dnz : dnz;
        H.hit = hit ? 1 : 0; H.capped = 0; H.block_id = 0; H.k = 0; H.sk = 0;
        const int ax = (wx ^ sx) + lox, ay = (wy ^ sy) + loy, az = (wz ^ sz) + loz;
        if (hit) {
            const int   syz = kk == 1 ? sy : sz, sk = kk == 0 ? sx : syz;            // sign mask of the last step's axis
            const int   ayz = kk == 1 ? ay : az, ak = kk == 0 ? ax : ayz;            // the hit voxel on that axis
            const float iyz = kk == 1 ? ay_in : az_in, imag = kk == 0 ? ax_in : iyz;  // |1 / rd_k|
            const float oyz = kk == 1 ? roy : roz, ok = kk == 0 ? rox : oyz;
            H.k = kk; H.sk = 1 | sk;
            // t as the reference computes it, in the unmirrored frame (see traverse_skip)
            const int   pk = ak - sk;                    // the entry plane: a + (s < 0)
            const float ik = sk ? -imag : imag;          // 1 / rd_k
            const float th = ((float)pk - ok) * ik;
            H.t = th;
            H.hx = rox + rdx * th; H.hy = roy + rdy * th; H.hz = roz + rdz * th;
            H.block_id = load_block<T>(P, ax, ay, az);
        } else {
            H.t = -1.0f; H.hx = H.hy = H.hz = 0.0f;
        }
        H.vx = ax; H.vy = ay; H.vz = az;
        H.steps = (unsigned)(wx + wy + wz - w0); H.iters = iters; H.work = work;
    }
};

// The box algebra needs non-zero directions and finite plane parameters.
VRT_DEV bool ray_is_regular(float rdx, float rdy, float rdz)
{
    // (all three reciprocals are taken on every path, so that the traversal's own 1 / rd are these)
    const float big = 0x1.0p100f;
    const float ix = fabsf(recip_regular(rdx)), iy = fabsf(recip_regular(rdy)), iz = fabsf(recip_regular(rdz));
    return ((rdx != 0.0f) & (rdy != 0.0f) & (rdz != 0.0f) & (ix < big) & (iy < big) & (iz < big)) != 0;
}

#ifndef VRT_PRIMARY_FOLD
#define VRT_PRIMARY_FOLD 0
#endif
constexpr int PRIMARY_FOLD = VRT_PRIMARY_FOLD;  // CellRay::trip<FOLD> of rays traced one per lane to the end

// A ray for which ray_is_regular holds, from the camera (CAMERA: origin = the launch's camera position, whose share
// of the set-up the host has done) or from anywhere.
template <typename T, int POW2, bool CAMERA = false>
VRT_DEV void traverse_cells_regular(const TraceParams &P, float rox, float roy, float roz, float rdx, float rdy,
                                    float rdz, int lox, int loy, int loz, RayHit &H)
{
    CellRay<T, POW2> ray;
    if (CAMERA) ray.template begin<false, true>(P, rox, roy, roz, rdx, rdy, rdz, lox, loy, loz, P.a0[0], P.a0[1], P.a0[2], 0u);
    else ray.template begin<false>(P, rox, roy, roz, rdx, rdy, rdz, lox, loy, loz, f2i_small(floorf(rox)),
                                   f2i_small(floorf(roy)), f2i_small(floorf(roz)), 0u);
#if defined(VRT_DIAG_STAGE) && VRT_DIAG_STAGE == 2
    H.steps = ray.digest(); return;  // (instruction-budget build: stops after the set-up)
#endif
    while (ray.live) ray.template trip<PRIMARY_FOLD>(P);
#if defined(VRT_DIAG_STAGE) && VRT_DIAG_STAGE == 3
    H.steps = ray.steps_done() ^ (unsigned)ray.k ^ (ray.hit ? 4u : 0u); return;  // (... after the trips)
#endif
    ray.finish(P, rox, roy, roz, lox, loy, loz, H);
}

// any ray: one that is not regular starts in the literal loop (NaN / inf semantics of the reference), which
// hands it back when the rest of it is an ordinary ray in the other axes
template <typename T, int POW2>
VRT_DEV void traverse_cells(const TraceParams &P, float rox, float roy, float roz, float rdx, float rdy,
                            float rdz, int lox, int loy, int loz, int upx, int upy, int upz, RayHit &H)
{
    CellRay<T, POW2> ray;
    int      ax = f2i_small(floorf(rox)), ay = f2i_small(floorf(roy)), az = f2i_small(floorf(roz));
    unsigned steps = 0;
    bool     boxed = true;
    if (!ray_is_regular(rdx, rdy, rdz))
        boxed = traverse_degenerate<T>(P, rox, roy, roz, rdx, rdy, rdz, lox, loy, loz, upx, upy, upz, H, ax, ay, az, steps);
    if (boxed) ray.template begin<true>(P, rox, roy, roz, rdx, rdy, rdz, lox, loy, loz, ax, ay, az, steps);
    else ray.idle();
    while (ray.live) ray.trip(P);
    if (boxed) ray.finish(P, rox, roy, roz, lox, loy, loz, H);
}

// The traversal a launch runs is fixed on the host (one kernel instance per form keeps the uniform
// state of the other forms out of the scalar registers):
//   FORM_CAPPED  one step per trip with the step cap tested (max_steps < 3n)
//   FORM_POW2    two-phase loop, texel wrap by mask;   FORM_ANY_N  two-phase loop, wrap by subtraction
//   FORM_POW2_WIDE  as FORM_POW2 for grids whose cells take more than 2^32 bytes (2048^3, 4096^3)
enum { FORM_CAPPED = 0, FORM_POW2 = 1, FORM_ANY_N = 2, FORM_POW2_WIDE = 3 };
constexpr int pow2_of(int form) { return form == FORM_POW2 ? 1 : (form == FORM_POW2_WIDE ? 2 : 0); }

template <typename T, int FORM>
VRT_DEV void traverse_auto(const TraceParams &P, float rox, float roy, float roz, float rdx, float rdy,
                           float rdz, int lox, int loy, int loz, int upx, int upy, int upz, RayHit &H)
{
    if (FORM == FORM_CAPPED) traverse_skip<T>(P, rox, roy, roz, rdx, rdy, rdz, lox, loy, loz, upx, upy, upz, H);
    else traverse_cells<T, pow2_of(FORM)>(P, rox, roy, roz, rdx, rdy, rdz, lox, loy, loz, upx, upy, upz, H);
}

// ---------------------------------------------------------------------------
// per-pixel prologue / epilogue shared by both kernels
// ---------------------------------------------------------------------------
// The launch parameters as they lie in the kernel-argument segment (TraceParams is the first parameter of
// every trace kernel, at offset 0: vrt_selftest_kernarg checks that), through a pointer the optimiser
// cannot trace back: loads through it are scalar loads issued where the code stands, not hoisted to the
// start of the kernel and held in scalar registers across the traversal loop.  Only valid in a kernel.
typedef const __attribute__((address_space(4))) TraceParams KernargParams;
VRT_DEV KernargParams *kernarg_again()
{
    KernargParams *p = (KernargParams *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// Wave tile = (64 >> tile_log2h) x (1 << tile_log2h) pixels; a workgroup stacks its four waves
// vertically.  tile_log2h = 3 gives 8x8-pixel waves in a 8x32 block.  `by` is the workgroup row.
template <typename PARAMS>
VRT_DEV bool pixel_of_thread(const PARAMS &P, unsigned by, unsigned tid, unsigned &col, unsigned &q, unsigned &row)
{
    const unsigned wave = tid >> 6, lane = tid & 63;
    const unsigned lanes = P.tile_lanes, lanes_log2 = lanes == 64u ? 6u : 5u;
    const unsigned lh = P.tile_log2h, tw = lanes >> lh;
#ifdef VRT_XCD_STRIPES
    // An experiment (measured, not adopted): workgroups are dealt round-robin over the eight XCDs, each with an L2 of its
    // own, so with a row of G workgroups (G a multiple of 8) workgroup x runs on XCD x % 8; tile column
    // (x % 8) * (G / 8) + x / 8 gives every XCD a vertical stripe of the frame instead of every eighth column.
    const unsigned G = gridDim.x, bx = (G & 7u) == 0 ? (blockIdx.x & 7u) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
#else
    const unsigned bx = blockIdx.x;
#endif
    col = bx * tw + (lane & (tw - 1));
    q   = ((by * 4 + wave) << lh) + (lane >> (lanes_log2 - lh));
    if (lane >= lanes || col >= P.w || q >= P.rows.rows) return false;
    if (P.rows.strip_h == 0) {
        row = P.rows.y0 + q;
    } else {
        row = ((q / P.rows.strip_h) * P.rows.parts + P.rows.part) * P.rows.strip_h + q % P.rows.strip_h;
    }
    return true;
}

// ---------------------------------------------------------------------------
// dispatch order of workgroup rows
// ---------------------------------------------------------------------------
// The hardware starts workgroups in blockIdx order, and once every SIMD is full a frame takes as
// long as its last wavefront.  Wavefronts differ by a factor of thirty (sky rows end after two or
// three boxes, rows under the horizon skim the terrain for a hundred steps), so the frame is
// shortest when the long rows start first and cheap ones fill the end.  Which rows are long
// depends on the scene and the camera, so it is measured: each wavefront stores how many trips of
// the traversal loop it made; the following launch reduces these to the longest time per row; the launch
// after that finds the topmost row that is not cheap; and the one after that starts there, runs
// down to the bottom and finishes with the rows above, all of which are cheap.  (One step per
// launch keeps every step a single round trip to memory for a single wavefront.)  Only the order of
// execution changes; every pixel is computed exactly as before.  Plain stores to a slot per
// wavefront: nothing is shared inside a launch, so no atomics, barriers or fences are needed,
// and the reader is a later kernel.
constexpr unsigned SCHED_MAX_ROWS = 256;

// Workgroup row of this workgroup.
VRT_DEV unsigned sched_row(const TraceParams &P)
{
    unsigned s = 0;
    if (P.sched.fixed_start >= 0) {
        s = (unsigned)P.sched.fixed_start % gridDim.y;
    } else if (P.sched.use_order) {
        const unsigned v = P.sched.start[P.sched.generation & 1u];
        if ((v >> 16) == P.sched.generation && (v & 0xffffu) < gridDim.y) s = v & 0xffffu;
    }
    unsigned by = blockIdx.y + s;
    if (by >= gridDim.y) by -= gridDim.y;
    return by;
}

// Maximum over the 64 lanes (all active), as a scalar: rotations inside each row of 16 lanes on the
// DPP path of the VALU, then one lane per row.
VRT_DEV unsigned wave_max(unsigned v)
{
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false));  // row_ror:8
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false));  // row_ror:4
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false));  // row_ror:2
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false));  // row_ror:1
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}

// Called by every lane of a wavefront at its end: one plain store to the wavefront's own slot.
VRT_DEV void sched_end(const TraceParams &P, unsigned by, unsigned work)
{
    const unsigned trips = wave_max(work);  // a wavefront runs the loop as often as its longest lane
    if ((threadIdx.x & 63u) != 0) return;
    const unsigned slot = (by * gridDim.x + blockIdx.x) * (blockDim.x / 64u) + (threadIdx.x >> 6);
    P.sched.cost[(P.sched.generation & 1u) * P.sched.capacity + slot] = (P.sched.generation << 20) | min(trips + 1u, 0xfffffu);
}

// One wavefront per workgroup row: the longest wavefront of this row in the previous launch.
VRT_DEV void sched_reduce_row(const TraceParams &P, unsigned by)
{
    const unsigned lane = threadIdx.x & 63u, per_row = gridDim.x * (blockDim.x / 64u);
    const unsigned src = P.sched.generation - 1u;
    const uint32_t *slots = P.sched.cost + (size_t)(src & 1u) * P.sched.capacity + (size_t)by * per_row;
    unsigned c = 0;
    bool known = true;
    for (unsigned x = lane; x < per_row; x += 64u) {
        const unsigned v = slots[x];
        if ((v >> 20) == src) c = max(c, v & 0xfffffu); else known = false;
    }
    for (int o = 32; o > 0; o >>= 1) c = max(c, (unsigned)__shfl_xor((int)c, o));
    if (__any(known ? 0 : 1)) c = 0;  // a wavefront without a time: the row counts as unknown
    if (lane == 0) P.sched.row_longest[(P.sched.generation & 1u) * SCHED_MAX_ROWS + by] = (P.sched.generation << 20) | c;
}

// One wavefront per launch: from the row times the previous launch reduced, the start row of the next.
VRT_DEV void sched_plan(const TraceParams &P)
{
    const unsigned lane = threadIdx.x & 63u, GY = gridDim.y;
    const unsigned src = P.sched.generation - 1u;
    const uint32_t *rows = P.sched.row_longest + (src & 1u) * SCHED_MAX_ROWS;
    unsigned mine[SCHED_MAX_ROWS / 64];  // lane l keeps rows l, l + 64, ...
    unsigned longest = 0, cheapest = 0xffffffffu;
    for (unsigned i = 0; i < SCHED_MAX_ROWS / 64; ++i) {
        const unsigned r = lane + 64u * i;
        mine[i] = 0;
        if (r < GY) {
            const unsigned v = rows[r];
            mine[i] = (v >> 20) == src ? (v & 0xfffffu) : 0u;
            longest = max(longest, mine[i]);
            cheapest = min(cheapest, mine[i]);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        longest = max(longest, (unsigned)__shfl_xor((int)longest, o));
        cheapest = min(cheapest, (unsigned)__shfl_xor((int)cheapest, o));
    }
    // "not cheap": more than twice the cheapest row and more than an eighth of the longest.  (The rows
    // that start a launch run with cold caches and come out a few microseconds slower than they are.)
    const unsigned threshold = max(2u * cheapest, longest / 8u);
    unsigned first = GY;
    for (unsigned i = 0; i < SCHED_MAX_ROWS / 64; ++i) {
        const unsigned r = lane + 64u * i;
        if (r < GY && mine[i] > threshold) first = min(first, r);
    }
    for (int o = 32; o > 0; o >>= 1) first = min(first, (unsigned)__shfl_xor((int)first, o));
    if (cheapest == 0) first = 0;  // a row without a time: keep the plain order
    if (lane == 0) {
        const unsigned next = P.sched.generation + 1u;
        P.sched.start[next & 1u] = (next << 16) | (first < GY ? first : 0u);
    }
}

// raytrace.glsl:120-122 without the normalisation: the direction before it and its squared length
VRT_DEV void raygen_raw(const TraceParams &P, unsigned gx, unsigned gy, float &d0, float &d1, float &d2, float &len2)
{
    const float fw = (float)P.w, fh = (float)P.h;
    const float ncx = P.ncx ? P.ncx[gx] : (float)gx / fw - 0.5f;
    const float ncy = P.ncy ? P.ncy[gy] : (float)gy / fh - P.ncy_off;
    const float *m = P.m;
    d0 = ((ncx * m[0] + ncy * m[1]) + 1.0f * m[2]) + 1.0f * m[3];
    d1 = ((ncx * m[4] + ncy * m[5]) + 1.0f * m[6]) + 1.0f * m[7];
    d2 = ((ncx * m[8] + ncy * m[9]) + 1.0f * m[10]) + 1.0f * m[11];
    len2 = (d0 * d0 + d1 * d1) + d2 * d2;
}

// raytrace.glsl:120-123
VRT_DEV void raygen(const TraceParams &P, unsigned gx, unsigned gy, float &rdx, float &rdy, float &rdz)
{
    float d0, d1, d2, len2;
    raygen_raw(P, gx, gy, d0, d1, d2, len2);
    const float len = sqrtf(len2);
    rdx = d0 / len; rdy = d1 / len; rdz = d2 / len;
}

// The same direction for the lanes of a wavefront together, and the verdict of ray_is_regular on it.  When every lane's
// components are tame (2^-40 <= |d_k|, squared length <= 2^80: any camera but one that looks exactly along a coordinate
// plane through this tile, or one with a matrix of absurd scale) the root and the three quotients are taken in their
// short forms, which have the same bits there (vrt_device.h), one reciprocal of the length serves the three
// divisions, and the verdict needs no test: 2^-40 <= len <= 2^40 then, so every |rd_k| is at least 2^-81, none is zero
// and no reciprocal reaches 2^100.  Any other wavefront takes the general forms, as raygen does.
VRT_DEV bool raygen_regular(const TraceParams &P, unsigned gx, unsigned gy, float &rdx, float &rdy, float &rdz)
{
    float d0, d1, d2, len2;
    raygen_raw(P, gx, gy, d0, d1, d2, len2);
    const float least = 0x1.0p-40f;
    const bool  tame = ((int)(fabsf(d0) >= least) & (int)(fabsf(d1) >= least) & (int)(fabsf(d2) >= least) & (int)(len2 <= 0x1.0p80f)) != 0;
    if (__builtin_amdgcn_ballot_w64(tame) == __builtin_amdgcn_ballot_w64(true)) {  // every active lane
        const float len = sqrt_tame(len2), r = tame_rcp(len);
        rdx = div_tame(d0, len, r); rdy = div_tame(d1, len, r); rdz = div_tame(d2, len, r);
        return true;
    }
    const float len = sqrtf(len2);
    rdx = d0 / len; rdy = d1 / len; rdz = d2 / len;
    return ray_is_regular(rdx, rdy, rdz);
}

// 0.5 * sin(floor(hit + 0.001) * 2) of raytrace.glsl:131 for axis k (halving a float is exact): from the frame's table
// when the floored coordinate is in it, else evaluated.  IN_WINDOW: the caller's ray is an ordinary one that hit a voxel
// of the loaded area or the first voxel beyond it, so the hit point lies on that voxel's face up to rounding and its
// floored coordinate within the table's margin of four voxels; the index is used as it is (held to the table's length,
// so that no address can leave the table whatever the arithmetic did).
template <bool IN_WINDOW, typename PARAMS>
VRT_DEV float shade_half_sine(const PARAMS &P, int k, float h)
{
    const float f = floorf(h + 0.001f);
    if (IN_WINDOW) {
        const unsigned i = (unsigned)(f2i_small(f) - P.sin_base[k]);
        return P.sin2[(unsigned)k * P.sin_len + min(i, P.sin_len - 1u)];
    }
    if (P.sin2) {
        const float i = f - (float)P.sin_base[k];  // exact; NaN (a degenerate ray) fails both tests
        if (i >= 0.0f && i < (float)P.sin_len) return P.sin2[(unsigned)k * P.sin_len + (unsigned)f2i_small(i)];
    }
    return 0.5f * canon_sin(f * 2.0f);
}

// the same arithmetic, once per frame shape / window position instead of once per pixel
__global__ __launch_bounds__(256) void k_frame_tables(float *ncx, float *ncy, float *sin2, unsigned w, unsigned h,
                                                      float ncy_off, int b0, int b1, int b2, unsigned sin_len)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    const float fw = (float)w, fh = (float)h;
    if (i < w) ncx[i] = (float)i / fw - 0.5f;
    if (i <= h) ncy[i] = (float)i / fh - ncy_off;
    if (i < sin_len) {
        sin2[i]               = 0.5f * canon_sin((float)(b0 + (int)i) * 2.0f);
        sin2[sin_len + i]     = 0.5f * canon_sin((float)(b1 + (int)i) * 2.0f);
        sin2[2 * sin_len + i] = 0.5f * canon_sin((float)(b2 + (int)i) * 2.0f);
    }
}

hipError_t launch_frame_tables(float *ncx, float *ncy, float *sin2, uint32_t w, uint32_t h, float ncy_off,
                               const int32_t sin_base[3], uint32_t sin_len, hipStream_t s)
{
    unsigned count = w > h + 1 ? w : h + 1;
    if (sin_len > count) count = sin_len;
    k_frame_tables<<<(count + 255) / 256, 256, 0, s>>>(ncx, ncy, sin2, w, h, ncy_off, sin_base[0], sin_base[1],
                                                        sin_base[2], sin_len);
    return hipGetLastError();
}

// raytrace.glsl:129-133 + render.comp:9-13
// (IN_WINDOW: see shade_half_sine)
template <bool IN_WINDOW = false, typename PARAMS>
VRT_DEV void finish_pixel(const PARAMS &P, unsigned gx, unsigned gy, size_t i, const RayHit &H)
{
    // The packed pixel, and the shader's float colour only where its plane is asked for.  The three constant colours
    // are packed constants (unorm8 of 0.2, 1.0, 0.8; of 0; of 1.0, 0.3, 0.3), so a wavefront without a hit -- every
    // tile of sky -- converts nothing.  Alpha is 1 on every path.
    float hr = 0.0f, hg = 0.0f, hb = 0.0f;
    unsigned px;
    const bool teal = H.steps >= 1000u;  // raytrace.glsl:73-75
    if (H.hit) {
        const float nk = (float)(-H.sk);
        const int   k = H.k;
        hr = (k == 0 ? nk : 0.0f) + shade_half_sine<IN_WINDOW>(P, 0, H.hx);
        hg = (k == 1 ? nk : 0.0f) + shade_half_sine<IN_WINDOW>(P, 1, H.hy);
        hb = (k == 2 ? nk : 0.0f) + shade_half_sine<IN_WINDOW>(P, 2, H.hz);
        px = unorm8_into(hb, 0u, unorm8_into(hg, 1u, unorm8_into(hr, 2u, 0xff000000u)));
    } else {
        px = teal ? 0xff33ffccu : 0xff000000u;
    }
    // render.comp:10-12: length(vec2(gid) - vec2(size/2)) < 1.2.  The offsets are integers: the length is below 1.2
    // exactly when the squared length is 0 or 1, that is when |dx| + |dy| <= 1.
    const bool cross = __sad(gy, P.h / 2, __sad(gx, P.w / 2, 0u)) <= 1u;
    if (cross) px = 0xffff4d4du;

    if (P.bgra8) ((unsigned *)P.bgra8)[i] = px;
    if (P.bgr8) { uint8_t *d = P.bgr8 + 3 * i; d[0] = (uint8_t)px; d[1] = (uint8_t)(px >> 8); d[2] = (uint8_t)(px >> 16); }
    if (P.voxel) { P.voxel[3 * i] = H.vx; P.voxel[3 * i + 1] = H.vy; P.voxel[3 * i + 2] = H.vz; }
    if (P.face) P.face[i] = H.hit ? 2 * H.k + (H.sk < 0 ? 1 : 0) : (H.capped ? -2 : -1);
#ifdef VRT_DIAG_ITERS
    if (P.steps) P.steps[i] = H.iters;  // diagnostic build: the kernel's own loop count
#else
    if (P.steps) P.steps[i] = H.steps;
#endif
    if (P.t) P.t[i] = H.t;
    if (P.hit_point) { P.hit_point[3 * i] = H.hx; P.hit_point[3 * i + 1] = H.hy; P.hit_point[3 * i + 2] = H.hz; }
    if (P.rgba) {
        float r = H.hit ? hr : (teal ? 0.2f : 0.0f), g = H.hit ? hg : (teal ? 1.0f : 0.0f), b = H.hit ? hb : (teal ? 0.8f : 0.0f);
        if (cross) { r = 1.0f; g = 0.3f; b = 0.3f; }
        P.rgba[4 * i] = r; P.rgba[4 * i + 1] = g; P.rgba[4 * i + 2] = b; P.rgba[4 * i + 3] = 1.0f;
    }
}

template <typename T, bool HAS_DIST>
__global__ __launch_bounds__(256) void k_trace_literal(const TraceParams P)
{
    unsigned col, q, row;
    if (!pixel_of_thread(P, blockIdx.y, threadIdx.x, col, q, row)) return;
    const unsigned gx = col, gy = P.h - row;  // render.comp:13
    float rdx, rdy, rdz;
    raygen(P, gx, gy, rdx, rdy, rdz);
    const float rox = P.m[12], roy = P.m[13], roz = P.m[14];  // raytrace.glsl:16
    // loaded area (raytrace.glsl:19-24), evaluated once per frame on the host
    const int lox = P.lo[0], loy = P.lo[1], loz = P.lo[2], upx = P.up[0], upy = P.up[1], upz = P.up[2];
    RayHit H;
    traverse_literal<T, HAS_DIST>(P, rox, roy, roz, rdx, rdy, rdz, lox, loy, loz, upx, upy, upz, H);
    finish_pixel(P, gx, gy, (size_t)q * P.w + col, H);
}

// A pixel whose ray has a zero direction component (a pixel or two in one frame out of twenty; the middle
// column with the reference's start orientation), from ray generation to the stores.  A function of its own,
// called after the loop of the other lanes, with the launch parameters read through a pointer to the kernel
// argument segment (TraceParams is the kernel's only argument): this seldom used code then has its own
// registers and costs the loop of the ordinary rays none.  Returns the trips of the two-phase loop.
template <typename T, int FORM>
__device__ __attribute__((noinline)) unsigned degenerate_pixel(const TraceParams *params, unsigned gx, unsigned gy, unsigned q)
{
    const TraceParams &P = *params;
    float rdx, rdy, rdz;
    raygen(P, gx, gy, rdx, rdy, rdz);
    const float rox = P.m[12], roy = P.m[13], roz = P.m[14];
    const int lox = P.lo[0], loy = P.lo[1], loz = P.lo[2], upx = P.up[0], upy = P.up[1], upz = P.up[2];
    RayHit H;
    traverse_auto<T, FORM>(P, rox, roy, roz, rdx, rdy, rdz, lox, loy, loz, upx, upy, upz, H);
    finish_pixel(P, gx, gy, (size_t)q * P.w + gx, H);
    return H.work;
}

// Camera rays that can meet nothing but air.  *P.sky is the level of the highest solid voxel of the whole window, plus one
// (written with the box words): a ray whose every voxel lies at or above that level ends where it leaves the loaded area,
// a miss, and its pixel depends on one thing only: whether it got there in 1000 iterations or more (raytrace.glsl:73-75).
// A ray that rises from a camera at or above that level is such a ray; so is one that descends but leaves the loaded
// area while still two levels above it (a ray's lowest voxel is its last one).  An iteration is a step of one voxel along
// one axis, so the count is the sum over the axes of the distance between the start voxel and the last one; here the
// last voxel is taken from the exit point worked out with approximate reciprocals -- off by at most one voxel per axis
// (where the exit point rounds to, which plane the reference's own rounding puts first), hence the two levels of margin
// -- and the count is only trusted when it is not within 6 of 1000.  Wave uniform: true when every lane of the
// wavefront has such a ray and a trusted count (H is then the ray's result, with that count); any other wavefront
// traces its rays as usual, and a wavefront with a ray too steep to stay above the highest voxel as far as the loaded
// area reaches does not try.  Not used when the launch writes the planes that need the last voxel or the count itself.
VRT_DEV bool sky_miss(const TraceParams &P, bool regular, float rdx, float rdy, float rdz, RayHit &H)
{
    if (P.sky == nullptr || P.voxel != nullptr || P.steps != nullptr) return false;
    const int top = (int)*P.sky, room = (P.a0[1] - P.lo[1]) - top;  // levels between the camera's voxel and the highest solid one
    if (room < 0) return false;
    // (a ray from the middle of the loaded area leaves it after 0.5 n to 0.9 n voxels: one that loses more than `room`
    // levels over 0.6 n may as well not be looked at)
    const float steepest = -(float)room * P.sky_slope;
    if (__builtin_amdgcn_ballot_w64(regular && rdy > steepest) != __builtin_amdgcn_ballot_w64(true)) return false;
    const float tx = (rdx > 0.0f ? P.sky_pos[0] : P.sky_neg[0]) * __builtin_amdgcn_rcpf(rdx);
    const float ty = (rdy > 0.0f ? P.sky_pos[1] : P.sky_neg[1]) * __builtin_amdgcn_rcpf(rdy);
    const float tz = (rdz > 0.0f ? P.sky_pos[2] : P.sky_neg[2]) * __builtin_amdgcn_rcpf(rdz);
    const float t = __builtin_fminf(__builtin_fminf(tx, ty), tz);
    const int   ex = f2i_floor(__builtin_fmaf(rdx, t, P.m[12])), ey = f2i_floor(__builtin_fmaf(rdy, t, P.m[13])),
                ez = f2i_floor(__builtin_fmaf(rdz, t, P.m[14]));
    // (|a - b| of signed coordinates by the unsigned instruction: both moved up by 2^31, which keeps their order)
    const unsigned up = 0x80000000u;
    const unsigned steps = __sad((unsigned)ez ^ up, (unsigned)P.a0[2] ^ up, __sad((unsigned)ey ^ up, (unsigned)P.a0[1] ^ up,
                                 __sad((unsigned)ex ^ up, (unsigned)P.a0[0] ^ up, 0u)));
    const bool unsure = steps - 994u <= 12u;  // 994 .. 1006
    const bool clear = rdy > 0.0f || ey - P.lo[1] >= top + 2;
    if (__builtin_amdgcn_ballot_w64(unsure || !clear) != 0) return false;
    H.hit = 0; H.capped = 0; H.vx = H.vy = H.vz = 0; H.k = 0; H.sk = 0; H.steps = steps; H.iters = 0; H.work = 0; H.block_id = 0;
    H.t = -1.0f; H.hx = H.hy = H.hz = 0.0f;
    return true;
}

// (eight wavefronts per SIMD, i.e. at most 64 VGPRs: measured 1.8 % faster than the 65 the compiler takes unasked)
template <typename T, int FORM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_trace_skip(const TraceParams P)
{
    const unsigned by = sched_row(P);
    unsigned col, q, row, work = 0;
    if (pixel_of_thread(P, by, threadIdx.x, col, q, row)) {
#ifdef VRT_DIAG_CLOCK
        const long long d0 = wall_clock64();
#endif
        float rdx, rdy, rdz;
        bool  regular = true;  // (the capped form sorts its rays out itself)
        if (FORM == FORM_CAPPED) raygen(P, col, P.h - row, rdx, rdy, rdz);
        else regular = raygen_regular(P, col, P.h - row, rdx, rdy, rdz);
#if defined(VRT_DIAG_STAGE) && VRT_DIAG_STAGE == 1
        // instruction-budget build: the kernel stops after ray generation (one word per pixel keeps it alive)
        ((unsigned *)P.bgra8)[(size_t)q * P.w + col] = __float_as_uint(rdx) ^ __float_as_uint(rdy) ^ __float_as_uint(rdz) ^ (regular ? 1u : 0u);
        return;
#endif
        const float rox = P.m[12], roy = P.m[13], roz = P.m[14];
        // loaded area (raytrace.glsl:19-24), evaluated once per frame on the host
        const int lox = P.lo[0], loy = P.lo[1], loz = P.lo[2], upx = P.up[0], upy = P.up[1], upz = P.up[2];
        RayHit H;
        const bool sky = FORM != FORM_CAPPED && sky_miss(P, regular, rdx, rdy, rdz, H);  // (wave uniform)
        if (regular && !sky) {
            if (FORM == FORM_CAPPED) traverse_skip<T>(P, rox, roy, roz, rdx, rdy, rdz, lox, loy, loz, upx, upy, upz, H);
            else traverse_cells_regular<T, pow2_of(FORM), true>(P, rox, roy, roz, rdx, rdy, rdz, lox, loy, loz, H);
            work = H.work;
        }
#if defined(VRT_DIAG_STAGE) && VRT_DIAG_STAGE >= 2
        // instruction-budget builds: stop after the set-up (2), the trips (3), the ray's outputs (4)
        ((unsigned *)P.bgra8)[(size_t)q * P.w + col] =
            VRT_DIAG_STAGE == 4 ? H.steps ^ __float_as_uint(H.t) ^ __float_as_uint(H.hx) ^ __float_as_uint(H.hy) ^ __float_as_uint(H.hz) ^
                                      H.block_id ^ (unsigned)H.k ^ (unsigned)H.sk ^ (unsigned)H.vx ^ (unsigned)H.vy ^ (unsigned)H.vz
                                : H.steps;
        return;
#endif
        // What follows the traversal loop takes nothing over from before it but the ray's result: the pixel's
        // coordinates are worked out again from the thread index, and the launch parameters are read again
        // from the kernel-argument segment.  The loop has no register to spare at 64 per lane (kept across it,
        // one of the coordinates went to scratch memory: 4 bytes per lane written and read back per launch).
        const KernargParams &K = *kernarg_again();
        pixel_of_thread(K, by, opaque(threadIdx.x), col, q, row);
        const unsigned gx = col, gy = K.h - row;
        // (the capped form also traces rays that are not ordinary ones)
        if (regular) finish_pixel<FORM != FORM_CAPPED>(K, gx, gy, (size_t)q * K.w + col, H);
        if (!regular)  // (the builtin is only valid in the kernel itself; TraceParams sits at offset 0 of the segment)
            work = degenerate_pixel<T, FORM>((const TraceParams *)__builtin_amdgcn_kernarg_segment_ptr(), gx, gy, q);
#ifdef VRT_DIAG_CLOCK
        // diagnostic build: start tick (100 MHz) in the face plane, duration in the steps plane
        const long long d1 = wall_clock64();
        if (P.face) P.face[(size_t)q * P.w + col] = (int)(d0 & 0x7fffffff);
        if (P.steps) P.steps[(size_t)q * P.w + col] = (unsigned)(d1 - d0);
        if (P.t) P.t[(size_t)q * P.w + col] = (float)H.iters;  // with -DVRT_DIAG_ITERS as well: the loop counts
#endif
    }
    if (P.sched.cost) {
        sched_end(P, by, work);
        // the bookkeeping is done by the workgroups that start first, which are far from the end of the launch
        const unsigned first = blockIdx.y * gridDim.x + blockIdx.x;
        if (first < gridDim.y && threadIdx.x < 64) {
            sched_reduce_row(P, first);
            if (first == 0) sched_plan(P);
        }
    }
}

#ifdef VRT_PRIMARY_STREAM
// ---------------------------------------------------------------------------
// primary rays with lane refill (an experiment, built only with EXTRA=-DVRT_PRIMARY_STREAM=R; measured slower than
// k_trace_skip and kept for that record)
// ---------------------------------------------------------------------------
// A wavefront traces R tiles of 16 x 4 pixels -- the workgroup rows blockIdx.y, blockIdx.y + gridDim.y, ... -- and a
// lane whose ray has ended starts on its pixel of its next tile as soon as VRT_STREAM_REFILL lanes wait (ballot of the
// waiting lanes), instead of idling until the wavefront's longest ray ends.  The ray is the resumable CellRay of the
// binned secondary-ray kernel; an ended ray is finished (shading, stores) when its lane next draws.
#ifndef VRT_STREAM_REFILL
#define VRT_STREAM_REFILL 24
#endif
#ifndef VRT_STREAM_WAVES
#define VRT_STREAM_WAVES 8
#endif
template <typename T, int FORM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VRT_STREAM_WAVES, 8))) void k_trace_skip_stream(const TraceParams P)
{
    constexpr unsigned R = VRT_PRIMARY_STREAM;
    const unsigned tid = threadIdx.x;
    const unsigned block_rows = 4u << P.tile_log2h, total_rows = (P.rows.rows + block_rows - 1) / block_rows;
    const float rox = P.m[12], roy = P.m[13], roz = P.m[14];
    const int lox = P.lo[0], loy = P.lo[1], loz = P.lo[2];
    CellRay<T, pow2_of(FORM)> ray;
    ray.idle();
    unsigned j = 0;        // tiles this lane has started
    bool holding = false;  // the lane holds a ray of tile j - 1, live or ended and not yet finished
    for (;;) {
        const bool waiting = !ray.live && (holding || j < R);
        const unsigned nwait = (unsigned)__popcll(__ballot(waiting ? 1 : 0));
        const bool none_live = !__any(ray.live ? 1 : 0);
        if (nwait == 0 && none_live) break;
        if (nwait >= VRT_STREAM_REFILL || none_live) {
            if (waiting) {
                unsigned col, q, row;
                if (holding) {
                    pixel_of_thread(P, blockIdx.y + (j - 1u) * gridDim.y, tid, col, q, row);
                    RayHit H;
                    ray.finish(P, rox, roy, roz, lox, loy, loz, H);
                    finish_pixel(P, col, P.h - row, (size_t)q * P.w + col, H);
                    holding = false;
                }
                while (j < R && !holding) {
                    const unsigned by = blockIdx.y + j * gridDim.y;
                    ++j;
                    if (by >= total_rows || !pixel_of_thread(P, by, tid, col, q, row)) continue;
                    float rdx, rdy, rdz;
                    if (raygen_regular(P, col, P.h - row, rdx, rdy, rdz)) {
                        ray.template begin<false>(P, rox, roy, roz, rdx, rdy, rdz, lox, loy, loz, f2i_small(floorf(rox)),
                                                  f2i_small(floorf(roy)), f2i_small(floorf(roz)), 0u);
                        holding = true;
                    } else {
                        degenerate_pixel<T, FORM>((const TraceParams *)__builtin_amdgcn_kernarg_segment_ptr(), col, P.h - row, q);
                    }
                }
            }
        }
        if (ray.live) ray.template trip<PRIMARY_FOLD>(P);
    }
}
#endif

// ---------------------------------------------------------------------------
// secondary rays per primary hit (no reference counterpart: the shader casts primary rays
// only).  One shadow ray towards normalize(0.4, 0.8, 0.45) and four cosine-distributed
// ambient-occlusion rays about the face normal, from the hash chain pcg(0x5EED ^ pixel);
// origin = hit_point + normal*1e-3; same traversal, same loaded area as the camera.
// ---------------------------------------------------------------------------
VRT_DEV unsigned pcg_hash(unsigned v)
{
    const unsigned state = v * 747796405u + 2891336453u;
    const unsigned word  = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

// Ambient-occlusion direction from the ray's hash h1 (the second hash of the pair follows from it), about `face`.
VRT_DEV void ao_dir_from_hash(unsigned h1, int face, float &dx, float &dy, float &dz)
{
    const unsigned h2 = pcg_hash(h1);
    const float u1 = (float)(h1 >> 8) * 0x1.0p-24f, u2 = (float)(h2 >> 8) * 0x1.0p-24f;
    const float r = sqrtf(u1), phi = 6.2831855f * u2;
    const float tx = r * canon_sin(phi + 1.5707964f);
    const float ty = r * canon_sin(phi);
    const float tz = sqrtf(1.0f - u1);
    const int   k = face >> 1;
    const float nz = (face & 1) ? tz : -tz;
    // normal axis k gets +-tz, the two other axes (in cyclic order) get tx, ty
    dx = k == 0 ? nz : (k == 2 ? tx : ty);
    dy = k == 1 ? nz : (k == 0 ? tx : ty);
    dz = k == 2 ? nz : (k == 1 ? tx : ty);
}

VRT_DEV void secondary_dir(unsigned pixel_index, unsigned index, int face, float &dx, float &dy, float &dz)
{
    if (index == 0) {
        const float lx = 0.4f, ly = 0.8f, lz = 0.45f;
        const float len = sqrtf((lx * lx + ly * ly) + lz * lz);
        dx = lx / len; dy = ly / len; dz = lz / len;
        return;
    }
    unsigned hsh = pcg_hash(0x5EEDu ^ pixel_index);
    for (unsigned j = 1; j < index; ++j) { hsh = pcg_hash(hsh); hsh = pcg_hash(hsh); }
    ao_dir_from_hash(hsh, face, dx, dy, dz);
}

// The secondary-ray kernel keeps its seldom used code out of line for the same reason as k_trace_skip
// (degenerate_pixel): a ray with a zero direction component goes through the complete traversal in a
// function of its own, which reads the launch parameters through the kernel-argument pointer.
// Primary ray of pixel (gx, gy): bit 0 hit, bits 1..2 axis, bit 3 s11 < 0; hit point in hp[0..2].
template <typename T, int FORM>
__device__ __attribute__((noinline)) unsigned degenerate_primary(const TraceParams *params, unsigned gx, unsigned gy, float *hp)
{
    const TraceParams &P = *params;
    float rdx, rdy, rdz;
    raygen(P, gx, gy, rdx, rdy, rdz);
    RayHit H;
    traverse_auto<T, FORM>(P, P.m[12], P.m[13], P.m[14], rdx, rdy, rdz, P.lo[0], P.lo[1], P.lo[2], P.up[0], P.up[1], P.up[2], H);
    hp[0] = H.hx; hp[1] = H.hy; hp[2] = H.hz;
    return (H.hit ? 1u : 0u) | ((unsigned)H.k << 1) | (H.sk < 0 ? 8u : 0u);
}

// Any ray in the camera's loaded area: bit 31 hit, the loop iterations below it.
template <typename T, int FORM>
__device__ __attribute__((noinline)) unsigned degenerate_ray(const TraceParams *params, float rox, float roy, float roz,
                                                             float rdx, float rdy, float rdz)
{
    const TraceParams &P = *params;
    RayHit S;
    traverse_auto<T, FORM>(P, rox, roy, roz, rdx, rdy, rdz, P.lo[0], P.lo[1], P.lo[2], P.up[0], P.up[1], P.up[2], S);
    return (S.hit ? 0x80000000u : 0u) | min(S.steps, 0x7fffffffu);
}

// (eight wavefronts per SIMD here too: 336 us per 1080p frame with secondaries against 359 us with the 85
// VGPRs and five wavefronts the compiler takes unasked, 343 us with six)
template <typename T, int FORM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_trace_secondary(const TraceParams P, uint8_t *shadow, uint8_t *ao, uint32_t *sec_steps)
{
    const unsigned by = sched_row(P);
    unsigned col, q, row, work = 0;
    if (pixel_of_thread(P, by, threadIdx.x, col, q, row)) {
        const unsigned gx = col, gy = P.h - row;
        float rdx, rdy, rdz;
        bool  regular = true;
        if (FORM == FORM_CAPPED) raygen(P, gx, gy, rdx, rdy, rdz);
        else regular = raygen_regular(P, gx, gy, rdx, rdy, rdz);
        const float rox = P.m[12], roy = P.m[13], roz = P.m[14];
        // loaded area (raytrace.glsl:19-24), evaluated once per frame on the host
        const int lox = P.lo[0], loy = P.lo[1], loz = P.lo[2], upx = P.up[0], upy = P.up[1], upz = P.up[2];
        const TraceParams *params = (const TraceParams *)__builtin_amdgcn_kernarg_segment_ptr();  // P sits at offset 0
        RayHit H;
        if (FORM == FORM_CAPPED) {
            traverse_skip<T>(P, rox, roy, roz, rdx, rdy, rdz, lox, loy, loz, upx, upy, upz, H);
        } else if (regular) {
            traverse_cells_regular<T, pow2_of(FORM), true>(P, rox, roy, roz, rdx, rdy, rdz, lox, loy, loz, H);
        } else {
            float hp[3];
            const unsigned r = degenerate_primary<T, FORM>(params, gx, gy, hp);
            H.hit = (int)(r & 1u); H.k = (int)((r >> 1) & 3u); H.sk = (r & 8u) ? -1 : 1;
            H.hx = hp[0]; H.hy = hp[1]; H.hz = hp[2]; H.work = 0;
        }
        unsigned sh = 0, occ = 0, st = 0;
        work = H.work;
        if (H.hit) {
            const float nk = (float)(-H.sk);
            const float sox = H.hx + (H.k == 0 ? nk : 0.0f) * 1e-3f;
            const float soy = H.hy + (H.k == 1 ? nk : 0.0f) * 1e-3f;
            const float soz = H.hz + (H.k == 2 ? nk : 0.0f) * 1e-3f;
            const int face = 2 * H.k + (H.sk < 0 ? 1 : 0);
            // The four ambient-occlusion rays are traced steepest first: within a pass the lanes of a
            // wavefront then hold rays of similar elevation, which end after a similar number of trips
            // (the directions are random per pixel; the sums written below do not depend on the order).
            unsigned order = 0;  // four 4-bit ray indices, steepest first
            {
                unsigned hsh = pcg_hash(0x5EEDu ^ (row * P.w + col));
                unsigned key[4];
                for (unsigned j = 0; j < 4; ++j) {
                    key[j] = ((hsh >> 8) << 3) | (j + 1u);  // u1 of ray j + 1 (24 bits) above its index
                    hsh = pcg_hash(hsh); hsh = pcg_hash(hsh);
                }
                auto cswap = [&](int a, int b) { const unsigned lo = min(key[a], key[b]), hi = max(key[a], key[b]); key[a] = lo; key[b] = hi; };
                cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2);
                order = (key[0] & 7u) | ((key[1] & 7u) << 4) | ((key[2] & 7u) << 8) | ((key[3] & 7u) << 12);
            }
            for (unsigned pass = 0; pass < 5; ++pass) {
                const unsigned s = pass == 0 ? 0u : (order >> (4u * (pass - 1u))) & 7u;
                float sdx, sdy, sdz;
                secondary_dir(row * P.w + col, s, face, sdx, sdy, sdz);
                RayHit S;
                if (FORM == FORM_CAPPED) {
                    traverse_skip<T>(P, sox, soy, soz, sdx, sdy, sdz, lox, loy, loz, upx, upy, upz, S);
                } else if (ray_is_regular(sdx, sdy, sdz)) {
                    traverse_cells_regular<T, pow2_of(FORM)>(P, sox, soy, soz, sdx, sdy, sdz, lox, loy, loz, S);
                } else {
                    const unsigned r = degenerate_ray<T, FORM>(params, sox, soy, soz, sdx, sdy, sdz);
                    S.hit = (int)(r >> 31); S.steps = r & 0x7fffffffu; S.work = 0;
                }
#ifdef VRT_DIAG_SEC_TRIPS
                // diagnostic build: trips of the two-phase loop instead of results (shadow: the shadow ray's,
                // ao: the longest ambient-occlusion ray's, both capped at 255; sec_steps: the sum over the five)
                st += S.work;
                if (s == 0) sh = min(S.work, 255u); else occ = max(occ, min(S.work, 255u));
#else
                st += S.steps;
                if (S.hit) { if (s == 0) sh = 1; else ++occ; }
#endif
                work += S.work;
            }
        }
        const size_t i = (size_t)q * P.w + col;
        if (shadow) shadow[i] = (uint8_t)sh;
        if (ao) ao[i] = (uint8_t)occ;
        if (sec_steps) sec_steps[i] = st;
    }
    if (P.sched.cost) {
        sched_end(P, by, work);
        // the bookkeeping is done by the workgroups that start first, which are far from the end of the launch
        const unsigned first = blockIdx.y * gridDim.x + blockIdx.x;
        if (first < gridDim.y && threadIdx.x < 64) {
            sched_reduce_row(P, first);
            if (first == 0) sched_plan(P);
        }
    }
}

// ---------------------------------------------------------------------------
// secondary rays, staged in LDS and binned by direction class
// ---------------------------------------------------------------------------
// k_trace_secondary gives every pixel's five rays to the pixel's own lane, pass by pass: lanes whose primary
// ray missed idle through all five passes, and the four ambient-occlusion rays a wavefront traces together
// point anywhere in their hemispheres.  Here the workgroup (a 16 x 16 pixel tile) turns its primary hits into
// a list of rays in LDS -- one 16-bit record per ray (pixel of the tile, ray index) with the direction beside it
// (worked out by the pixel's own lane as it places the ray); origin and face once per pixel -- sorted by direction
// class (counting sort over LDS counters): ambient-occlusion rays by elevation above their surface in sixteen
// classes, the shallowest (which skim the terrain and run longest) first, the shadow rays (one direction for
// the whole frame, none stored) last.  The four wavefronts then draw rays from the list through an LDS ticket
// until it is empty: a wavefront draws as soon as SEC_REFILL of its lanes are free, so no lane idles for a pixel
// that missed, a wavefront's rays have alike elevations, and a wavefront that draws short rays takes more of
// them.  Unit steps are folded into the box phase of the traversal (CellRay::trip<SEC_FOLD>): among rays that
// have little to do with each other a handful of lanes per trip need one, which does not pay for a phase of its
// own.  Results are summed per pixel with LDS atomics (integer sums: the order of arrival does not matter), by
// the lanes that are about to draw, and stored by the pixel's own lane.
// Measured on 1920 x 1080 over the 1024^3 terrain, 5.9 M secondary rays: 32.3 Grays/s, 0.67 of the lanes active
// per vector instruction, against 22.2 Grays/s and 0.45 for k_trace_secondary.
constexpr unsigned SEC_BINS = 17;  // 16 elevation classes of ambient-occlusion rays + the shadow rays
constexpr unsigned SEC_AO_HIT = 1u << 24, SEC_SHADOW_HIT = 1u << 28;  // a ray's hit in its pixel's accumulator, above the iterations
#ifndef VRT_SEC_REFILL
#define VRT_SEC_REFILL 24
#endif
#ifndef VRT_SEC_WAVES
#define VRT_SEC_WAVES 8
#endif
#ifndef VRT_SEC_FOLD
#define VRT_SEC_FOLD 64
#endif
constexpr int SEC_FOLD = VRT_SEC_FOLD;  // CellRay::trip<FOLD> of the binned kernel's rays
constexpr unsigned SEC_REFILL = VRT_SEC_REFILL;  // free lanes at which a wavefront draws new rays (1..64; 64: whole bundles)

template <typename T, int FORM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VRT_SEC_WAVES, 8))) void k_trace_secondary_binned(const TraceParams P, uint8_t *shadow, uint8_t *ao, uint32_t *sec_steps)
{
    // 20 112 bytes: eight workgroups (32 wavefronts, the most a CU holds) fit a CU's 160 KB with the 20 480 each may take
    __shared__ float4         s_org[256];      // origin of the pixel's secondary rays; .w: face of the primary hit (bits), 0xff: no hit
    __shared__ unsigned       s_acc[256];      // per pixel: loop iterations summed over its rays (bits 0..23: at most 5 x 4 x 4096),
                                               // ambient-occlusion rays that hit (bits 24..27), the shadow ray hit (bit 28)
    __shared__ unsigned short s_ray[5 * 256];  // the sorted list: pixel | ray index << 8
    __shared__ float          s_dir[3][4 * 256];  // directions of the ambient-occlusion rays, by position in the list
    __shared__ unsigned       s_count[SEC_BINS], s_base[SEC_BINS], s_next, s_total;
    static_assert(sizeof(float4) * 256 + sizeof(unsigned) * 256 + sizeof(unsigned short) * 5 * 256 + sizeof(float) * 3 * 4 * 256 +
                      sizeof(unsigned) * (2 * SEC_BINS + 2) <= 20480,
                  "the binned secondary-ray kernel may take 20 KB of LDS: a ninth of a byte more and a CU holds seven workgroups instead of eight");

    const unsigned tid = threadIdx.x;
    const unsigned by = sched_row(P);
    if (tid < SEC_BINS) s_count[tid] = 0;
    if (tid == 0) s_next = 0;
    s_acc[tid] = 0; s_org[tid].w = __uint_as_float(0xffu);
    __syncthreads();

    // ---- phase 1: the primary ray of this lane's pixel
    unsigned col, q, row, work = 0;
    unsigned bins = 0;  // five 5-bit bin numbers of this pixel's rays, ray 0 lowest; only meaningful after a hit
    unsigned ao_hash[4] = {0, 0, 0, 0};  // first hash of each ambient-occlusion ray's pair
    bool     hit = false;
    const bool inside = pixel_of_thread(P, by, tid, col, q, row);
    if (inside) {
        const unsigned gx = col, gy = P.h - row;
        float rdx, rdy, rdz;
        bool  regular = true;
        if (FORM == FORM_CAPPED) raygen(P, gx, gy, rdx, rdy, rdz);
        else regular = raygen_regular(P, gx, gy, rdx, rdy, rdz);
        const float rox = P.m[12], roy = P.m[13], roz = P.m[14];
        const int lox = P.lo[0], loy = P.lo[1], loz = P.lo[2], upx = P.up[0], upy = P.up[1], upz = P.up[2];
        const TraceParams *params = (const TraceParams *)__builtin_amdgcn_kernarg_segment_ptr();  // P sits at offset 0
        RayHit H;
        if (FORM == FORM_CAPPED) {
            traverse_skip<T>(P, rox, roy, roz, rdx, rdy, rdz, lox, loy, loz, upx, upy, upz, H);
        } else if (regular) {
            traverse_cells_regular<T, pow2_of(FORM), true>(P, rox, roy, roz, rdx, rdy, rdz, lox, loy, loz, H);
        } else {
            float hp[3];
            const unsigned r = degenerate_primary<T, FORM>(params, gx, gy, hp);
            H.hit = (int)(r & 1u); H.k = (int)((r >> 1) & 3u); H.sk = (r & 8u) ? -1 : 1;
            H.hx = hp[0]; H.hy = hp[1]; H.hz = hp[2]; H.work = 0;
        }
        work = H.work;
        hit = H.hit != 0;
        if (hit) {
            const float nk = (float)(-H.sk);
            s_org[tid] = make_float4(H.hx + (H.k == 0 ? nk : 0.0f) * 1e-3f, H.hy + (H.k == 1 ? nk : 0.0f) * 1e-3f,
                                     H.hz + (H.k == 2 ? nk : 0.0f) * 1e-3f, __uint_as_float((unsigned)(2 * H.k + (H.sk < 0 ? 1 : 0))));
            // direction classes: ray j + 1 has elevation sqrt(1 - u1) over its surface, u1 = (hash >> 8) * 2^-24
            // (secondary_dir): class = the top four bits of u1, counted from the shallow end
            unsigned hsh = pcg_hash(0x5EEDu ^ (row * P.w + col));
            bins = SEC_BINS - 1;  // the shadow ray
            for (unsigned j = 0; j < 4; ++j) {
                ao_hash[j] = hsh;
#if defined(VRT_SEC_ORDER) && VRT_SEC_ORDER == 1      // (measurement builds: steepest first / one class for all)
                bins |= (hsh >> 28) << (5u * (j + 1u));
#elif defined(VRT_SEC_ORDER) && VRT_SEC_ORDER == 2
                bins |= 0u << (5u * (j + 1u));
#else
                bins |= (15u - (hsh >> 28)) << (5u * (j + 1u));
#endif
                hsh = pcg_hash(hsh); hsh = pcg_hash(hsh);
            }
            // (the shadow rays of a wavefront are counted and placed by one add: 64 lanes on one counter take turns)
            const unsigned long long hm = __ballot(1);
            if (__builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u)) == 0)
                atomicAdd(&s_count[SEC_BINS - 1], (unsigned)__popcll(hm));
            for (unsigned j = 1; j < 5; ++j) atomicAdd(&s_count[(bins >> (5u * j)) & 31u], 1u);
        }
    }
    __syncthreads();
    if (tid == 0) {
        unsigned run = 0;
#if defined(VRT_SEC_ORDER) && VRT_SEC_ORDER == 3      // (measurement build: the shadow rays at the head of the list)
        s_base[SEC_BINS - 1] = 0; run = s_count[SEC_BINS - 1];
        for (unsigned b = 0; b < SEC_BINS - 1; ++b) { s_base[b] = run; run += s_count[b]; }
#else
        for (unsigned b = 0; b < SEC_BINS; ++b) { s_base[b] = run; run += s_count[b]; }
#endif
        s_total = run;
    }
    __syncthreads();
    // every hit pixel's lane puts its five rays into the list, and the directions of its four ambient-occlusion rays
    // beside them (the list's tail of shadow rays shares one direction and has none stored)
#if defined(VRT_SEC_ORDER) && VRT_SEC_ORDER == 3
    const unsigned ao_base = s_count[SEC_BINS - 1];  // list position of the first ambient-occlusion ray
#else
    const unsigned ao_base = 0;
#endif
    if (hit) {
        const int face = (int)__float_as_uint(s_org[tid].w);
        {
            const unsigned long long hm = __ballot(1);
            const unsigned r = __builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u));
            unsigned b0 = 0;
            if (r == 0) b0 = atomicAdd(&s_base[SEC_BINS - 1], (unsigned)__popcll(hm));
            b0 = (unsigned)__builtin_amdgcn_readfirstlane((int)b0);  // (the first active lane is the one that added)
            s_ray[b0 + r] = (unsigned short)tid;
        }
        for (unsigned j = 0; j < 4; ++j) {
            const unsigned at = atomicAdd(&s_base[(bins >> (5u * (j + 1u))) & 31u], 1u);
            s_ray[at] = (unsigned short)(tid | ((j + 1u) << 8));
            float dx, dy, dz;
            ao_dir_from_hash(ao_hash[j], face, dx, dy, dz);
            s_dir[0][at - ao_base] = dx; s_dir[1][at - ao_base] = dy; s_dir[2][at - ao_base] = dz;
        }
    }
    __syncthreads();
    const unsigned total = s_total, n_ao = total - s_count[SEC_BINS - 1];

    // ---- phase 2: every wavefront traces rays of the list until it is empty.  A lane whose ray has ended
    // takes the next ray of the list as soon as SEC_REFILL lanes of its wavefront are free (one LDS ticket for
    // all of them, ranks by mbcnt over the ballot of free lanes), so a wavefront's trips stay filled while
    // rays of very different lengths -- three trips on average, a hundred at most -- come and go.  Every
    // wavefront sees the list empty sooner or later and leaves the loop, so all reach the barrier below.
    {
        const int lox = P.lo[0], loy = P.lo[1], loz = P.lo[2], upx = P.up[0], upy = P.up[1], upz = P.up[2];
        const TraceParams *params = (const TraceParams *)__builtin_amdgcn_kernarg_segment_ptr();
        const unsigned lane = tid & 63u;
        float lightx, lighty, lightz;
        secondary_dir(0u, 0u, 0, lightx, lighty, lightz);  // the shadow rays' one direction
        CellRay<T, pow2_of(FORM)> ray;
        ray.idle();
        constexpr unsigned NO_RAY = 0xffffffffu;
        unsigned mine = NO_RAY;  // pixel | ray index << 8 of the ray this lane holds, live or ended and not yet delivered
        // the result of an ended ray goes to its pixel when the lane next draws (or at the end), together with the
        // other free lanes' results, rather than in the trip it ended in with the few lanes that ended with it
        auto deliver = [&]() {
            if (mine != NO_RAY) {
                const unsigned pix = mine & 255u;   // one add carries the ray's iterations and its hit
                atomicAdd(&s_acc[pix], ray.steps_done() + (ray.hit ? ((mine >> 8) == 0 ? SEC_SHADOW_HIT : SEC_AO_HIT) : 0u));
                mine = NO_RAY;
            }
        };
        unsigned trips = 0;     // wave uniform: trips of the traversal loop this wavefront ran
#ifdef VRT_DIAG_SEC_UTIL
        unsigned dRefills = 0, dRefillLanes = 0, dLive = 0, dBegun = 0, dTailTrips = 0, dTailLive = 0;
#endif
        bool     drained = false;  // wave uniform: the list has no ray left for this wavefront
        for (;;) {
            const unsigned long long free_lanes = __ballot(ray.live ? 0 : 1);
            const unsigned nfree = (unsigned)__popcll(free_lanes);
            if (!drained && nfree >= SEC_REFILL) {
                unsigned first = 0;
                if (lane == 0) first = atomicAdd(&s_next, nfree);
                first = (unsigned)__builtin_amdgcn_readfirstlane((int)first);
                drained = first + nfree >= total;
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(free_lanes >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)free_lanes, 0u));
#ifdef VRT_DIAG_SEC_UTIL
                dRefills += 1u; dRefillLanes += nfree; dBegun += (!ray.live && first + rank < total) ? 1u : 0u;
#endif
                if (!ray.live) deliver();
                if (!ray.live && first + rank < total) {
                    const unsigned at = first + rank;
                    const unsigned rec = s_ray[at];
                    const unsigned pix = rec & 255u, j = rec >> 8;
                    const bool     is_ao = at - ao_base < n_ao;  // (unsigned: a shadow ray ahead of ao_base wraps round)
                    const unsigned dat = is_ao ? at - ao_base : 0u;
                    const float sdx = is_ao ? s_dir[0][dat] : lightx, sdy = is_ao ? s_dir[1][dat] : lighty,
                                sdz = is_ao ? s_dir[2][dat] : lightz;
                    const float4 so = s_org[pix];   // (one 16-byte read)
                    const float sox = so.x, soy = so.y, soz = so.z;
                    if (FORM == FORM_CAPPED) {  // the one-step-per-trip form is not resumable: the whole ray here
                        RayHit S;
                        traverse_skip<T>(P, sox, soy, soz, sdx, sdy, sdz, lox, loy, loz, upx, upy, upz, S);
                        atomicAdd(&s_acc[pix], S.steps + (S.hit ? (j == 0 ? SEC_SHADOW_HIT : SEC_AO_HIT) : 0u));
                    } else if (ray_is_regular(sdx, sdy, sdz)) {
                        ray.template begin<false>(P, sox, soy, soz, sdx, sdy, sdz, lox, loy, loz, f2i_small(floorf(sox)),
                                                  f2i_small(floorf(soy)), f2i_small(floorf(soz)), 0u);
                        mine = rec;  // (a ray that starts outside the loaded area is not live and is delivered as it is)
                    } else {
                        const unsigned r = degenerate_ray<T, FORM>(params, sox, soy, soz, sdx, sdy, sdz);
                        atomicAdd(&s_acc[pix], (r & 0x7fffffffu) + ((r >> 31) ? (j == 0 ? SEC_SHADOW_HIT : SEC_AO_HIT) : 0u));
                    }
                }
            }
            if (!__any(ray.live ? 1 : 0)) {
                if (drained) break;
                continue;  // every lane is free: the next round refills them
            }
            trips += 1u;
#ifdef VRT_DIAG_SEC_UTIL
            dLive += ray.live ? 1u : 0u;
            if (drained) { dTailTrips += 1u; dTailLive += ray.live ? 1u : 0u; }
            ray.dAx += __any((ray.live && ray.entry() != ray.NO_BOX) ? 1 : 0) ? 1u : 0u;
#endif
            if (ray.live) ray.template trip<SEC_FOLD>(P);
        }
        deliver();
        work += trips;
#ifdef VRT_DIAG_SEC_UTIL
        // counters summed over the launch into sec_steps[0..9] (the plane itself is not written in this build)
        if (sec_steps) {
            auto wsum = [](unsigned v) { for (int o = 32; o > 0; o >>= 1) v += (unsigned)__shfl_xor((int)v, o); return v; };
            const unsigned sA = wsum(ray.dA), sB1 = wsum(ray.dB1), sB2 = wsum(ray.dB2), sLive = wsum(dLive), sBegun = wsum(dBegun);
            const unsigned sNear = wsum(ray.dNear), sNearX = wsum(ray.dNearX), sTailLive = wsum(dTailLive);
            if (lane == 0) {
                atomicAdd(&sec_steps[0], trips); atomicAdd(&sec_steps[1], sLive); atomicAdd(&sec_steps[2], sA);
                atomicAdd(&sec_steps[3], sB1); atomicAdd(&sec_steps[4], sB2); atomicAdd(&sec_steps[5], ray.dAx);
                atomicAdd(&sec_steps[6], dRefills); atomicAdd(&sec_steps[7], dRefillLanes); atomicAdd(&sec_steps[8], sBegun);
                atomicAdd(&sec_steps[9], 1u);
                atomicAdd(&sec_steps[10], sNear); atomicAdd(&sec_steps[11], sNearX);
                atomicAdd(&sec_steps[12], dTailTrips); atomicAdd(&sec_steps[13], sTailLive);
            }
        }
#endif
    }
    __syncthreads();
#ifdef VRT_DIAG_SEC_UTIL
    sec_steps = nullptr;
#endif

    // ---- phase 3: every lane stores its own pixel (coordinates worked out again rather than kept across the phases)
    if (pixel_of_thread(P, by, opaque(tid), col, q, row)) {
        const size_t i = (size_t)q * P.w + col;
        const unsigned acc = s_acc[tid];
        if (shadow) shadow[i] = (uint8_t)((acc >> 28) & 1u);
        if (ao) ao[i] = (uint8_t)((acc >> 24) & 15u);
        if (sec_steps) sec_steps[i] = acc & 0xffffffu;
    }
    if (P.sched.cost) {
        sched_end(P, by, work);
        const unsigned first = blockIdx.y * gridDim.x + blockIdx.x;
        if (first < gridDim.y && threadIdx.x < 64) {
            sched_reduce_row(P, first);
            if (first == 0) sched_plan(P);
        }
    }
}

static int trace_form(const TraceParams &p)
{
    // (FORM_POW2 addresses the cells by 32-bit byte offsets: 128 * (n / 4)^3 <= 2^32)
    if (p.need_cap) return FORM_CAPPED;
    if (p.n_pow2_mask < 0) return FORM_ANY_N;
    return p.n <= 1024 ? FORM_POW2 : FORM_POW2_WIDE;
}

static dim3 trace_grid(const TraceParams &p)
{
    const unsigned tw = p.tile_lanes >> p.tile_log2h, bh = 4u << p.tile_log2h;
    return dim3((p.w + tw - 1) / tw, (p.rows.rows + bh - 1) / bh, 1);
}

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

hipError_t launch_trace_secondary(const TraceParams &p, int voxel_bytes, bool binned, uint8_t *shadow, uint8_t *ao,
                                  uint32_t *sec_steps, hipStream_t s)
{
    if (p.rows.rows == 0 || p.w == 0) return hipSuccess;
    const dim3 grid = trace_grid(p), block(256);
    const int form = trace_form(p);
#define VRT_LAUNCH_SECONDARY(K, T)                                                           \
    (form == FORM_CAPPED ? K<T, FORM_CAPPED><<<grid, block, 0, s>>>(p, shadow, ao, sec_steps) \
     : form == FORM_POW2 ? K<T, FORM_POW2><<<grid, block, 0, s>>>(p, shadow, ao, sec_steps)   \
     : form == FORM_POW2_WIDE ? K<T, FORM_POW2_WIDE><<<grid, block, 0, s>>>(p, shadow, ao, sec_steps) \
                         : K<T, FORM_ANY_N><<<grid, block, 0, s>>>(p, shadow, ao, sec_steps))
    if (binned) {
        if (voxel_bytes == 1) VRT_LAUNCH_SECONDARY(k_trace_secondary_binned, uint8_t);
        else VRT_LAUNCH_SECONDARY(k_trace_secondary_binned, uint16_t);
    } else {
        if (voxel_bytes == 1) VRT_LAUNCH_SECONDARY(k_trace_secondary, uint8_t);
        else VRT_LAUNCH_SECONDARY(k_trace_secondary, uint16_t);
    }
#undef VRT_LAUNCH_SECONDARY
    return hipGetLastError();
}

// Workgroups of 256 threads of one trace kernel that a compute unit holds at once (registers, LDS and wavefront slots
// taken together), as the runtime works it out for the current device: kind 0 the primary skip kernel, 1 the binned
// secondary-ray kernel, 2 the literal kernel; the u8 / power-of-two instances.  Eight (32 wavefronts) is what the
// kernels are built for.
hipError_t trace_kernel_occupancy(int kind, int *workgroups_per_cu)
{
    const void *k = kind == 0   ? reinterpret_cast<const void *>(&k_trace_skip<uint8_t, FORM_POW2>)
                    : kind == 1 ? reinterpret_cast<const void *>(&k_trace_secondary_binned<uint8_t, FORM_POW2>)
                                : reinterpret_cast<const void *>(&k_trace_literal<uint8_t, false>);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(workgroups_per_cu, k, 256, 0);
}

hipError_t launch_trace_skip(const TraceParams &p, int voxel_bytes, hipStream_t s)
{
    if (p.rows.rows == 0 || p.w == 0) return hipSuccess;
    const dim3 grid = trace_grid(p), block(256);
    const int form = trace_form(p);
#ifdef VRT_PRIMARY_STREAM
    if (form != FORM_CAPPED) {
        const dim3 sgrid(grid.x, (grid.y + VRT_PRIMARY_STREAM - 1) / VRT_PRIMARY_STREAM, 1);
        if (voxel_bytes == 1) {
            if (form == FORM_POW2) k_trace_skip_stream<uint8_t, FORM_POW2><<<sgrid, block, 0, s>>>(p);
            else if (form == FORM_POW2_WIDE) k_trace_skip_stream<uint8_t, FORM_POW2_WIDE><<<sgrid, block, 0, s>>>(p);
            else k_trace_skip_stream<uint8_t, FORM_ANY_N><<<sgrid, block, 0, s>>>(p);
        } else {
            if (form == FORM_POW2) k_trace_skip_stream<uint16_t, FORM_POW2><<<sgrid, block, 0, s>>>(p);
            else if (form == FORM_POW2_WIDE) k_trace_skip_stream<uint16_t, FORM_POW2_WIDE><<<sgrid, block, 0, s>>>(p);
            else k_trace_skip_stream<uint16_t, FORM_ANY_N><<<sgrid, block, 0, s>>>(p);
        }
        return hipGetLastError();
    }
#endif
#define VRT_LAUNCH_SKIP(T)                                                              \
    (form == FORM_CAPPED ? k_trace_skip<T, FORM_CAPPED><<<grid, block, 0, s>>>(p)         \
     : form == FORM_POW2 ? k_trace_skip<T, FORM_POW2><<<grid, block, 0, s>>>(p)           \
     : form == FORM_POW2_WIDE ? k_trace_skip<T, FORM_POW2_WIDE><<<grid, block, 0, s>>>(p) \
                         : k_trace_skip<T, FORM_ANY_N><<<grid, block, 0, s>>>(p))
    if (voxel_bytes == 1) VRT_LAUNCH_SKIP(uint8_t);
    else VRT_LAUNCH_SKIP(uint16_t);
#undef VRT_LAUNCH_SKIP
    return hipGetLastError();
}

}  // namespace vrt
