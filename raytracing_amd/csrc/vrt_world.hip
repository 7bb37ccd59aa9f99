// vrt_world.hip -- kernels that produce and index the voxel world in HBM.
//
//   k_terrain       : terrain_gen.comp:14-24 + noise.glsl:4-74 for a box of world voxels,
//                     stored at their toroidal texels (bit-exact restatement: same
//                     operation order as the shader, no FMA contraction).
//   k_cell_masks    : one 64-bit occupancy mask per 4^3 cell of the texel image, into the builders' plane of masks.
//   k_df_x, k_df_y, k_df_z_pack : the separable (min, max) octant distance transform over cells (torus
//                     metric), both directions of an axis per launch; the last packs the eight octant
//                     distances of a cell into the builders' plane of distances.  All of these run on a box of cells.
//   k_column_vtops, k_tops_extend, k_tops_max, k_voxel_pyramid : column tops at voxel precision, their maxima over
//                     squares of 1, 2, .. 256 cell columns, and the cells themselves: a 128-byte line per cell with the
//                     record (mask, distances) and the eight box words of the traversal.
//   k_any_nonzero   : is the uploaded distance image all zero (as shipped)?
#include "vrt_device.h"
#include "vrt_kernels.h"

namespace vrt {

// ---------------------------------------------------------------------------
// noise.glsl
// ---------------------------------------------------------------------------
VRT_DEV float glsl_mod(float x, float y) { return x - y * floorf(x / y); }
VRT_DEV float permute1(float x) { return glsl_mod(((x * 34.0f) + 1.0f) * x, 289.0f); }       // noise.glsl:4
VRT_DEV float taylor_inv_sqrt(float r) { return 1.79284291400159f - 0.85373472095314f * r; }  // noise.glsl:5
VRT_DEV float stepf(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
VRT_DEV float dot3(float ax, float ay, float az, float bx, float by, float bz) { return (ax * bx + ay * by) + az * bz; }

// noise.glsl:7-74
VRT_DEV float snoise(float vx, float vy, float vz)
{
    const float Cx = (float)(1.0 / 6.0), Cy = (float)(1.0 / 3.0);
    // :12-13
    const float s  = (vx * Cy + vy * Cy) + vz * Cy;
    float ix = floorf(vx + s), iy = floorf(vy + s), iz = floorf(vz + s);
    const float u  = (ix * Cx + iy * Cx) + iz * Cx;
    const float x0x = (vx - ix) + u, x0y = (vy - iy) + u, x0z = (vz - iz) + u;
    // :16-19  g = step(x0.yzx, x0.xyz); l = 1 - g; i1 = min(g, l.zxy); i2 = max(g, l.zxy)
    const float gx = stepf(x0y, x0x), gy = stepf(x0z, x0y), gz = stepf(x0x, x0z);
    const float lx = 1.0f - gx, ly = 1.0f - gy, lz = 1.0f - gz;
    const float i1x = fminf(gx, lz), i1y = fminf(gy, lx), i1z = fminf(gz, ly);
    const float i2x = fmaxf(gx, lz), i2y = fmaxf(gy, lx), i2z = fmaxf(gz, ly);
    // :22-24
    const float x1x = (x0x - i1x) + 1.0f * Cx, x1y = (x0y - i1y) + 1.0f * Cx, x1z = (x0z - i1z) + 1.0f * Cx;
    const float x2x = (x0x - i2x) + 2.0f * Cx, x2y = (x0y - i2y) + 2.0f * Cx, x2z = (x0z - i2z) + 2.0f * Cx;
    const float x3x = (x0x - 1.0f) + 3.0f * Cx, x3y = (x0y - 1.0f) + 3.0f * Cx, x3z = (x0z - 1.0f) + 3.0f * Cx;
    // :27-31
    ix = glsl_mod(ix, 289.0f); iy = glsl_mod(iy, 289.0f); iz = glsl_mod(iz, 289.0f);
    const float oz[4] = {0.0f, i1z, i2z, 1.0f};
    const float oy[4] = {0.0f, i1y, i2y, 1.0f};
    const float ox[4] = {0.0f, i1x, i2x, 1.0f};
    // :35-36
    const float n_  = (float)(1.0 / 7.0);
    const float nsx = n_ * 2.0f - 0.0f, nsy = n_ * 0.5f - 1.0f, nsz = n_ * 1.0f - 0.0f;
    float x[4], y[4], h[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float a = permute1(iz + oz[c]);
        const float b = permute1((a + iy) + oy[c]);
        const float p = permute1((b + ix) + ox[c]);
        const float j  = p - 49.0f * floorf(p * nsz * nsz);  // :38
        const float x_ = floorf(j * nsz);                    // :40
        const float y_ = floorf(j - 7.0f * x_);              // :41
        x[c] = x_ * nsx + nsy;                               // :43
        y[c] = y_ * nsx + nsy;                               // :44
        h[c] = (1.0f - fabsf(x[c])) - fabsf(y[c]);           // :45
    }
    // :47-55
    const float b0[4] = {x[0], x[1], y[0], y[1]};
    const float b1[4] = {x[2], x[3], y[2], y[3]};
    float s0[4], s1[4], sh[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        s0[c] = floorf(b0[c]) * 2.0f + 1.0f;
        s1[c] = floorf(b1[c]) * 2.0f + 1.0f;
        sh[c] = -stepf(h[c], 0.0f);
    }
    const float a0x = b0[0] + s0[0] * sh[0], a0y = b0[2] + s0[2] * sh[0], a0z = b0[1] + s0[1] * sh[1], a0w = b0[3] + s0[3] * sh[1];
    const float a1x = b1[0] + s1[0] * sh[2], a1y = b1[2] + s1[2] * sh[2], a1z = b1[1] + s1[1] * sh[3], a1w = b1[3] + s1[3] * sh[3];
    // :57-67
    float p0x = a0x, p0y = a0y, p0z = h[0];
    float p1x = a0z, p1y = a0w, p1z = h[1];
    float p2x = a1x, p2y = a1y, p2z = h[2];
    float p3x = a1z, p3y = a1w, p3z = h[3];
    const float n0 = taylor_inv_sqrt(dot3(p0x, p0y, p0z, p0x, p0y, p0z));
    const float n1 = taylor_inv_sqrt(dot3(p1x, p1y, p1z, p1x, p1y, p1z));
    const float n2 = taylor_inv_sqrt(dot3(p2x, p2y, p2z, p2x, p2y, p2z));
    const float n3 = taylor_inv_sqrt(dot3(p3x, p3y, p3z, p3x, p3y, p3z));
    p0x *= n0; p0y *= n0; p0z *= n0;
    p1x *= n1; p1y *= n1; p1z *= n1;
    p2x *= n2; p2y *= n2; p2z *= n2;
    p3x *= n3; p3y *= n3; p3z *= n3;
    // :70-73
    float m0 = fmaxf(0.6f - dot3(x0x, x0y, x0z, x0x, x0y, x0z), 0.0f);
    float m1 = fmaxf(0.6f - dot3(x1x, x1y, x1z, x1x, x1y, x1z), 0.0f);
    float m2 = fmaxf(0.6f - dot3(x2x, x2y, x2z, x2x, x2y, x2z), 0.0f);
    float m3 = fmaxf(0.6f - dot3(x3x, x3y, x3z, x3x, x3y, x3z), 0.0f);
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2; m3 = m3 * m3;
    const float d0 = dot3(p0x, p0y, p0z, x0x, x0y, x0z), d1 = dot3(p1x, p1y, p1z, x1x, x1y, x1z);
    const float d2 = dot3(p2x, p2y, p2z, x2x, x2y, x2z), d3 = dot3(p3x, p3y, p3z, x3x, x3y, x3z);
    const float acc = (((m0 * m0) * d0 + (m1 * m1) * d1) + (m2 * m2) * d2) + (m3 * m3) * d3;
    return 42.0f * acc;
}

constexpr int NOISE_REACH = 32;  // no voxel further than this from y = 0 depends on the noise (k_terrain)

// terrain_gen.comp:14-24 for world voxels [x0,x0+ex) x [y0,y0+ey) x [z0,z0+ez)
template <typename T>
__global__ __launch_bounds__(256) void k_terrain(T *block, int n, int x0, int y0, int z0, int ex, int ey, int ez)
{
    const int lx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ly = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int lz = blockIdx.z;
    if (lx >= ex || ly >= ey || lz >= ez) return;
    const int px = x0 + lx, py = y0 + ly, pz = z0 + lz;
    // |snoise| < 3.6 for every input.  Each of the four corner terms is 42 * m^4 * dot(p, x) with m = 0.6 - |x|^2 (or 0) and
    // p the corner's gradient: at most 42 * max_r (0.6 - r^2)^4 * r * |p| = 42 * 0.02089 * |p| (the maximum lies at
    // r^2 = 0.6 / 9).  A gradient lies on the octahedron |x| + |y| + |h| = 1 before it is scaled (noise.glsl:49-63 fold the
    // points with h < 0 back onto it), so its squared length s lies in [1/3, 1], and scaled by taylorInvSqrt(s) =
    // 1.79284 - 0.85373 s its length sqrt(s) * (1.79284 - 0.85373 s) is at most 1.0001 (the maximum lies at s = 0.70).  Four
    // corners: 4 * 42 * 0.02089 * 1.0001 = 3.51, and rounding adds a few parts in a million.  So |snoise * 6| < 22, and
    // for |y| > NOISE_REACH = 32 the sign of snoise*6 - y is the sign of -y whatever the noise is: the 500-odd instructions
    // of the noise are skipped (a wavefront shares y).  The CPU oracle evaluates the noise everywhere.
    T v;
    if (py > NOISE_REACH || py < -NOISE_REACH) {
        v = py < 0 ? (T)1 : (T)0;
    } else {
        const float noise_val = snoise((float)px * 0.015f, (float)py * 0.015f, (float)pz * 0.015f) * 6.0f - (float)py;
        v = noise_val > 0.0f ? (T)1 : (T)0;
    }
    const size_t tx = (size_t)rem_euclid(px, n), ty = (size_t)rem_euclid(py, n), tz = (size_t)rem_euclid(pz, n);
    block[(tz * (size_t)n + ty) * (size_t)n + tx] = v;
}

// The same for byte voxels, four along x per thread and one 32-bit store (x0 and ex multiples of 4, as a window's are):
// far from y = 0, where no noise is evaluated, the kernel is its stores, and a wavefront's 64 single bytes fill half a
// cache line where 64 words fill two.
__global__ __launch_bounds__(256) void k_terrain_u8x4(uint8_t *block, int n, int x0, int y0, int z0, int ex4, int ey, int ez)
{
    const int lx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ly = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int lz = blockIdx.z;
    if (lx >= ex4 || ly >= ey || lz >= ez) return;
    const int px = x0 + 4 * lx, py = y0 + ly, pz = z0 + lz;
    unsigned v = 0;
    if (py > NOISE_REACH || py < -NOISE_REACH) {  // see k_terrain
        v = py < 0 ? 0x01010101u : 0u;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float noise_val = snoise((float)(px + i) * 0.015f, (float)py * 0.015f, (float)pz * 0.015f) * 6.0f - (float)py;
            v |= (noise_val > 0.0f ? 1u : 0u) << (8 * i);
        }
    }
    // (px is a multiple of 4 and so is n: the four texels do not wrap apart)
    const size_t tx = (size_t)rem_euclid(px, n), ty = (size_t)rem_euclid(py, n), tz = (size_t)rem_euclid(pz, n);
    *reinterpret_cast<uint32_t *>(block + (tz * (size_t)n + ty) * (size_t)n + tx) = v;
}

// the same for up to 64 chunks of 32^3 voxels per launch (positions in chunk units): a workgroup is 32 x 8 voxels
// of one chunk, so every lane of a wavefront has a voxel
struct ChunkBatch {
    int count;
    int xyz[64][3];
};

template <typename T>
__global__ __launch_bounds__(256) void k_terrain_chunks(T *block, int n, ChunkBatch batch)
{
    const int *c = batch.xyz[blockIdx.x];
    const int px = c[0] * 32 + (int)(threadIdx.x & 31u);
    const int py = c[1] * 32 + (int)(blockIdx.y * 8u + (threadIdx.x >> 5));
    const int pz = c[2] * 32 + (int)blockIdx.z;
    T v;
    if (py > NOISE_REACH || py < -NOISE_REACH) {  // see k_terrain
        v = py < 0 ? (T)1 : (T)0;
    } else {
        const float noise_val = snoise((float)px * 0.015f, (float)py * 0.015f, (float)pz * 0.015f) * 6.0f - (float)py;
        v = noise_val > 0.0f ? (T)1 : (T)0;
    }
    const size_t tx = (size_t)rem_euclid(px, n), ty = (size_t)rem_euclid(py, n), tz = (size_t)rem_euclid(pz, n);
    block[(tz * (size_t)n + ty) * (size_t)n + tx] = v;
}

hipError_t launch_terrain_chunks(void *block, int voxel_bytes, int n, const int32_t *xyz, uint32_t count, hipStream_t s)
{
    for (uint32_t done = 0; done < count;) {
        ChunkBatch b;
        b.count = (int)(count - done < 64u ? count - done : 64u);
        for (int i = 0; i < b.count; ++i)
            for (int k = 0; k < 3; ++k) b.xyz[i][k] = xyz[3 * (done + (uint32_t)i) + (uint32_t)k];
        const dim3 grid((unsigned)b.count, 4, 32), blockdim(256);
        if (voxel_bytes == 1) k_terrain_chunks<uint8_t><<<grid, blockdim, 0, s>>>((uint8_t *)block, n, b);
        else k_terrain_chunks<uint16_t><<<grid, blockdim, 0, s>>>((uint16_t *)block, n, b);
        done += (uint32_t)b.count;
    }
    return hipGetLastError();
}

hipError_t launch_terrain(void *block, int voxel_bytes, int n, int x0, int y0, int z0, int ex, int ey, int ez,
                          hipStream_t s)
{
    if (ex <= 0 || ey <= 0 || ez <= 0) return hipSuccess;
    // grid.z is limited to 65535; ez <= 4096 here.
    const dim3 grid((ex + 63) / 64, (ey + 3) / 4, ez), blockdim(256);
    if (voxel_bytes == 1 && (x0 & 3) == 0 && (ex & 3) == 0 && (n & 3) == 0)
        k_terrain_u8x4<<<dim3((ex / 4 + 63) / 64, (ey + 3) / 4, ez), blockdim, 0, s>>>((uint8_t *)block, n, x0, y0, z0, ex / 4, ey, ez);
    else if (voxel_bytes == 1) k_terrain<uint8_t><<<grid, blockdim, 0, s>>>((uint8_t *)block, n, x0, y0, z0, ex, ey, ez);
    else k_terrain<uint16_t><<<grid, blockdim, 0, s>>>((uint16_t *)block, n, x0, y0, z0, ex, ey, ez);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// cell records: cell (cx,cy,cz) covers texels [4c, 4c+4)^3
// ---------------------------------------------------------------------------
// A cell's record and its box words lie in one 128-byte line: `cells` points at the record of cell 0 and cell
// i's is cells[CELL_U4 * i], its box words behind it; k_voxel_pyramid writes both, from the builders' planes.  A ray that
// lands in a cell reads the record and one of the words, and among rays that have little to do with each other that is
// one miss in the L2 per cell instead of two.
// Every builder below works on a box of cells of the torus: origin o (each coordinate in [0, nc)) and extent e
// (1..nc per axis); the cell of a thread is (o + local) mod nc.  The whole torus is the box (0, nc); after a few
// chunks have been regenerated only the cells whose records can have changed are visited (launch_update_cells).
struct CellBox {
    int ox, oy, oz, ex, ey, ez;
};

VRT_DEV bool box_cell(const CellBox &b, int nc, int &cx, int &cy, int &cz)
{
    const int lx = blockIdx.x * 64 + (threadIdx.x & 63), ly = blockIdx.y * 4 + (threadIdx.x >> 6), lz = blockIdx.z;
    if (lx >= b.ex || ly >= b.ey || lz >= b.ez) return false;
    cx = b.ox + lx; cx -= cx >= nc ? nc : 0;
    cy = b.oy + ly; cy -= cy >= nc ? nc : 0;
    cz = b.oz + lz; cz -= cz >= nc ? nc : 0;
    return true;
}

static dim3 box_grid(const CellBox &b) { return dim3((b.ex + 63) / 64, (b.ey + 3) / 4, b.ez); }

// occupancy mask, bit = z*16 + y*4 + x, into the builders' own plane of masks (8 bytes per cell, side by side: the
// passes below read 32 masks ahead of every cell, which would be 32 lines of the 128-byte cells); k_voxel_pyramid copies
// it into the cell's record
template <typename T>
__global__ __launch_bounds__(256) void k_cell_masks(const T *block, int n, int nc, CellBox box, uint2 *masks)
{
    int cx, cy, cz;
    if (!box_cell(box, nc, cx, cy, cz)) return;
    uint64_t m = 0;
#pragma unroll
    for (int z = 0; z < 4; ++z) {
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const size_t base = ((size_t)(4 * cz + z) * (size_t)n + (size_t)(4 * cy + y)) * (size_t)n + (size_t)(4 * cx);
            unsigned bits = 0;
            if (sizeof(T) == 1) {
                const uint32_t v = *reinterpret_cast<const uint32_t *>(block + base);
                bits = ((v & 0xffu) ? 1u : 0u) | ((v & 0xff00u) ? 2u : 0u) | ((v & 0xff0000u) ? 4u : 0u) | ((v & 0xff000000u) ? 8u : 0u);
            } else {
                const uint2 v = *reinterpret_cast<const uint2 *>(block + base);
                bits = ((v.x & 0xffffu) ? 1u : 0u) | ((v.x >> 16) ? 2u : 0u) | ((v.y & 0xffffu) ? 4u : 0u) | ((v.y >> 16) ? 8u : 0u);
            }
            m |= (uint64_t)bits << (z * 16 + y * 4);
        }
    }
    masks[((size_t)cz * nc + cy) * nc + cx] = make_uint2((unsigned)m, (unsigned)(m >> 32));
}

// The octant distances come from a separable transform.  One axis of it: out(p) = min over r >= 0 of
// max(r, in(p + dir*r)) along the axis (torus).  Applied along x, y, z with directions (sx, sy, sz) to
// in = {0 non-empty, CAP empty} it yields the Chebyshev distance to the nearest non-empty cell q with
// s_k*(q_k - c_k) >= 0 on every axis, i.e. the side of the largest empty cube of cells that has the cell at its
// corner and extends into that octant, capped at DF_CAP cells: wide open space is the column-top pyramid's
// job, and a pass costs up to DF_CAP look-ups per cell and direction (1024^3 voxels: 8.5 ms of passes with a
// cap of 255, 2.5 ms with 32, same trace times).  Three launches: x (both directions from the masks, two
// planes), y (both directions of both x planes, four planes), z (both directions of the four y planes, the eight
// results packed straight into the record).  The x and y planes stay in memory between builds: a value depends
// on the DF_CAP - 1 cells ahead of it only, so after a change inside a box of cells the x planes are redone on
// the box widened by DF_CAP - 1 cells along x, the y planes on that widened along y, the records on that widened
// along z, and every other cell keeps a value that is still right.
constexpr int DF_CAP = 32;

template <typename LOAD>
VRT_DEV int df_scan(LOAD value, int self, int pos, int nc, int dir)
{
    int best = self;
    for (int r = 1; r < best && r < nc; ++r) {
        int q = pos + dir * r;
        q += q < 0 ? nc : 0;
        q -= q >= nc ? nc : 0;
        best = min(best, max(r, value(q)));
    }
    return best;
}

// planes: tx[sx] at tx + sx * ncells
__global__ __launch_bounds__(256) void k_df_x(const uint2 *masks, int nc, CellBox box, uint8_t *tx)
{
    int cx, cy, cz;
    if (!box_cell(box, nc, cx, cy, cz)) return;
    const size_t ncells = (size_t)nc * nc * nc, line = ((size_t)cz * nc + cy) * nc;
    auto value = [&](int q) -> int {
        const uint2 m = masks[line + q];
        return (m.x | m.y) != 0 ? 0 : DF_CAP;
    };
    const int self = value(cx);
    tx[line + cx]          = (uint8_t)df_scan(value, self, cx, nc, -1);
    tx[ncells + line + cx] = (uint8_t)df_scan(value, self, cx, nc, 1);
}

// planes: ty[sx | sy << 1] at ty + (sx | sy << 1) * ncells
__global__ __launch_bounds__(256) void k_df_y(const uint8_t *tx, int nc, CellBox box, uint8_t *ty)
{
    int cx, cy, cz;
    if (!box_cell(box, nc, cx, cy, cz)) return;
    const size_t ncells = (size_t)nc * nc * nc, base = (size_t)cz * nc * nc + cx;
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
        const uint8_t *in = tx + (size_t)sx * ncells + base;
        auto value = [&](int q) -> int { return (int)in[(size_t)q * nc]; };
        const int self = value(cy);
        ty[(size_t)sx * ncells + base + (size_t)cy * nc]       = (uint8_t)df_scan(value, self, cy, nc, -1);
        ty[(size_t)(sx | 2) * ncells + base + (size_t)cy * nc] = (uint8_t)df_scan(value, self, cy, nc, 1);
    }
}

// the eight octant distances of a cell, byte o = sx | sy << 1 | sz << 2: the record's .z and .w words, into the builders'
// own plane of distances (side by side like the masks; k_voxel_pyramid, which every change of the index is followed by,
// writes the cells themselves, a whole cell at a time)
__global__ __launch_bounds__(256) void k_df_z_pack(const uint8_t *ty, int nc, CellBox box, uint2 *dists)
{
    int cx, cy, cz;
    if (!box_cell(box, nc, cx, cy, cz)) return;
    const size_t ncells = (size_t)nc * nc * nc, plane = (size_t)nc * nc, base = (size_t)cy * nc + cx;
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int sxy = 0; sxy < 4; ++sxy) {
        const uint8_t *in = ty + (size_t)sxy * ncells + base;
        auto value = [&](int q) -> int { return (int)in[(size_t)q * plane]; };
        const int self = value(cz);
        lo |= (unsigned)df_scan(value, self, cz, nc, -1) << (8 * sxy);
        hi |= (unsigned)df_scan(value, self, cz, nc, 1) << (8 * sxy);
    }
    dists[(size_t)cz * plane + base] = make_uint2(lo, hi);
}

// box [lo, lo + len) of an axis widened by `grow` cells on both sides, on a torus of nc cells
static void widen(int &o, int &e, int grow, int nc)
{
    if (e + 2 * grow >= nc) { o = 0; e = nc; return; }
    o -= grow; e += 2 * grow;
    o %= nc; if (o < 0) o += nc;
}

hipError_t launch_update_cells(const void *block, int voxel_bytes, int n, uint4 *cells, uint8_t *planes, const int origin[3],
                               const int extent[3], hipStream_t s)
{
    const int nc = n / 4;
    CellBox b{origin[0], origin[1], origin[2], extent[0], extent[1], extent[2]};
    if (b.ex <= 0 || b.ey <= 0 || b.ez <= 0) return hipSuccess;
    const size_t ncells = (size_t)nc * nc * nc;
    const dim3 blockdim(256);
    // planes: the masks and the octant distances (8 bytes per cell each), the two x planes, the four y planes
    uint2   *masks = reinterpret_cast<uint2 *>(planes), *dists = masks + ncells;
    uint8_t *tx = planes + 16 * ncells, *ty = planes + 18 * ncells;
    if (voxel_bytes == 1) k_cell_masks<uint8_t><<<box_grid(b), blockdim, 0, s>>>((const uint8_t *)block, n, nc, b, masks);
    else k_cell_masks<uint16_t><<<box_grid(b), blockdim, 0, s>>>((const uint16_t *)block, n, nc, b, masks);
    widen(b.ox, b.ex, DF_CAP - 1, nc);
    k_df_x<<<box_grid(b), blockdim, 0, s>>>(masks, nc, b, tx);
    widen(b.oy, b.ey, DF_CAP - 1, nc);
    k_df_y<<<box_grid(b), blockdim, 0, s>>>(tx, nc, b, ty);
    widen(b.oz, b.ez, DF_CAP - 1, nc);
    k_df_z_pack<<<box_grid(b), blockdim, 0, s>>>(ty, nc, b, dists);
    return hipGetLastError();
}

hipError_t launch_build_cells(const void *block, int voxel_bytes, int n, uint4 *cells, uint8_t *planes, hipStream_t s)
{
    const int origin[3] = {0, 0, 0}, extent[3] = {n / 4, n / 4, n / 4};
    return launch_update_cells(block, voxel_bytes, n, cells, planes, origin, extent, s);
}

// ---------------------------------------------------------------------------
// column tops at voxel precision (TraceParams::pyr)
// ---------------------------------------------------------------------------
// Levels are counted from the bottom of the loaded window: window layer wl of torus cell row tcy is
// (tcy - offcy) mod nc, offcy = rem_euclid(lower_y, n) / 4, and window level v = 4*wl + (voxel row in the cell).
// tops[0][tcz][tcx] = window level just above the highest solid voxel of the 4x4 voxel columns of cell
// column (tcx, tcz), 0 if they are empty; tops[L][tcz][tcx] = the maximum over the square of 2^L x 2^L cell columns
// (tcx .. tcx + 2^L - 1) x (tcz .. tcz + 2^L - 1) of the torus -- a table per level, every cell column the corner of a
// square of its own (not a pyramid of aligned blocks: a ray uses the square that starts at its own cell and extends the
// way it travels, all of it, where an aligned block serves it only from wherever in the block the ray happens to be).
// A cell is non-empty iff its octant distances are zero.
__global__ __launch_bounds__(256) void k_tops_extend(const uint16_t *fine, uint16_t *coarse, int nc, int half)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int z = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nc || z >= nc) return;
    int x1 = x + half, z1 = z + half;
    x1 -= x1 >= nc ? nc : 0;
    z1 -= z1 >= nc ? nc : 0;
    const unsigned a = fine[(size_t)z * nc + x], b = fine[(size_t)z * nc + x1];
    const unsigned c = fine[(size_t)z1 * nc + x], d = fine[(size_t)z1 * nc + x1];
    coarse[(size_t)z * nc + x] = (uint16_t)max(max(a, b), max(c, d));
}

struct TopsLevels {
    const uint16_t *tops[TOPS_LEVELS];  // nc x nc each
    int             levels;             // highest level present: squares of 2^levels <= nc cell columns
};

__global__ __launch_bounds__(256) void k_column_vtops(const uint2 *masks, int nc, int offcy, uint16_t *tops0)
{
    const int tcx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int tcz = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (tcx >= nc || tcz >= nc) return;
    int v = 0;
    for (int wl = nc - 1; wl >= 0; --wl) {
        int tcy = wl + offcy;
        tcy -= tcy >= nc ? nc : 0;
        const uint2 rec = masks[((size_t)tcz * nc + tcy) * nc + tcx];
        if ((rec.x | rec.y) == 0) continue;  // empty cell
        const uint64_t m = (uint64_t)rec.x | ((uint64_t)rec.y << 32);
        int y = 3;
        while (y > 0 && (m & (0x000F000F000F000Full << (4 * y))) == 0) --y;
        v = 4 * wl + y + 1;
        break;
    }
    tops0[(size_t)tcz * nc + tcx] = (uint16_t)v;
}

// The eight words of a cell: word = descending | negative x << 1 | negative z << 2, an entry per voxel level of the cell:
// the empty box a ray that travels the word's way crosses from a voxel of that level -- BOX_NONE, or
// (side - 1) << 8 | rise: the box spans `side` cells along x and z from the ray's own cell on in its direction of travel,
// and along y from the voxel on for `rise` voxels more (BOX_RISE_ALL: up to the top of the window).  Two kinds of box
// end up in that form, chosen here so that the traversal need not:
//   the column square  2^L x 2^L cell columns with the cell's own at the corner, none of them with a solid voxel at or
//                      above the window level (voxel's level - drop): rise = drop for a ray that descends, all the way
//                      up for one that rises (valid inside a non-empty cell too, above its solid voxels);
//   the octant cube    of the cell's record (side d cells, empty cells only): rise = what is left of the voxel's own
//                      cell + 4 (d - 1).
// The square is taken when twice its side exceeds the cube's (measured in the traversal, when it chose: better than once
// and than four times).  One thread per cell and pair of words (a quadrant of travel, negative x | negative z << 1), the
// four threads of a cell side by side: 64 bytes in a row.
__global__ __launch_bounds__(256) void k_voxel_pyramid(int nc, int offcy, TopsLevels tl, const uint2 *masks, const uint2 *dists, uint4 *cells)
{
    const int tx = blockIdx.x * 256 + threadIdx.x;
    const int tcx = tx >> 2, q = tx & 3;
    const int tcy = blockIdx.y;
    const int tcz = blockIdx.z;
    if (tcx >= nc) return;
    int wl = tcy - offcy;
    wl += wl < 0 ? nc : 0;
    const size_t at = ((size_t)tcz * nc + tcy) * nc + tcx;
    uint4 *cell = cells + CELL_U4 * at;
    // the square of level L that extends towards -x has its corner 2^L - 1 columns before the cell
    const bool negx = (q & 1) != 0, negz = (q & 2) != 0;
    // Any level whose square lies at or below the voxel gives a valid box; two choices are kept.
    //   up[j]    the widest square: for rays that rise, which leave a box through its sides only
    //   down[j]  the largest min(side, 4 * height above the square's top): a wide square with little room
    //            underneath ends a descending ray after a few voxels (measured with aligned blocks on the bench
    //            frame: 58 us against 62 us with the widest; factors 2..4 alike, 1 worse)
    int up_side[4] = {0, 0, 0, 0}, down_side[4] = {0, 0, 0, 0}, down_drop[4] = {0, 0, 0, 0};
    int score[4] = {-1, -1, -1, -1};
    for (int L = 0; L <= tl.levels; ++L) {
        const int back = (1 << L) - 1;
        int x = negx ? tcx - back : tcx, z = negz ? tcz - back : tcz;
        x += x < 0 ? nc : 0;
        z += z < 0 ? nc : 0;
        const int t = (int)tl.tops[L][(size_t)z * nc + x];
        if (t > 4 * wl + 3) break;  // tops grow with L
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int v = 4 * wl + j;
            if (t > v) continue;
            const int sc = min(4 << L, 4 * (v - t + 1));
            up_side[j] = 1 << L;
            if (sc >= score[j]) { score[j] = sc; down_side[j] = 1 << L; down_drop[j] = v - t; }
        }
    }
    // octant distances of the cell's record: byte o = sx | sy << 1 | sz << 2, bit set = positive direction
    const uint2 dist = dists[at];
    const int   oxz = (negx ? 0 : 1) | (negz ? 0 : 4);
    const int   d_up = (int)(((oxz & 4 ? dist.y : dist.x) >> (8 * ((oxz | 2) & 3))) & 0xffu);
    const int   d_down = (int)(((oxz & 4 ? dist.y : dist.x) >> (8 * (oxz & 3))) & 0xffu);
    unsigned up[4], down[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // (a side of 256 cells is stored as 255: the entry of a window-wide square that rises all the way must not read
        // as BOX_NONE)
        up[j] = 2 * up_side[j] > d_up ? ((unsigned)(min(up_side[j], 255) - 1) << 8) | BOX_RISE_ALL
                : d_up > 0           ? ((unsigned)(d_up - 1) << 8) | (unsigned)((3 - j) + 4 * (d_up - 1))
                                     : BOX_NONE;
        down[j] = 2 * down_side[j] > d_down ? ((unsigned)(min(down_side[j], 255) - 1) << 8) | (unsigned)min(down_drop[j], (int)BOX_RISE_ALL - 1)
                  : d_down > 0             ? ((unsigned)(d_down - 1) << 8) | (unsigned)(j + 4 * (d_down - 1))
                                           : BOX_NONE;
    }
    if (q == 0) { const uint2 m = masks[at]; cell[0] = make_uint4(m.x, m.y, dist.x, dist.y); }  // the record
    cell[1 + q] = make_uint4(up[0] | (up[1] << 16), up[2] | (up[3] << 16), down[0] | (down[1] << 16), down[2] | (down[3] << 16));
}

// the highest of all column tops (one workgroup)
__global__ __launch_bounds__(256) void k_tops_max(const uint16_t *tops0, int count, uint32_t *out)
{
    __shared__ unsigned part[4];
    unsigned v = 0;
    for (int i = (int)threadIdx.x; i < count; i += 256) v = max(v, (unsigned)tops0[i]);
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) *out = max(max(part[0], part[1]), max(part[2], part[3]));
}

hipError_t launch_build_voxel_pyramid(const uint8_t *planes, int nc, int offcy, uint16_t *tops, uint4 *cells, uint32_t *sky, hipStream_t s)
{
    const dim3 blockdim(256), grid2((nc + 63) / 64, (nc + 3) / 4);
    TopsLevels tl{};
    uint16_t *level = tops;
    k_column_vtops<<<grid2, blockdim, 0, s>>>(reinterpret_cast<const uint2 *>(planes), nc, offcy, level);
    tl.tops[0] = level; tl.levels = 0;
    k_tops_max<<<1, blockdim, 0, s>>>(level, nc * nc, sky);
    for (int L = 1; L < TOPS_LEVELS; ++L) {
        if ((1 << L) > nc) break;
        uint16_t *coarse = level + (size_t)nc * nc;
        k_tops_extend<<<grid2, blockdim, 0, s>>>(level, coarse, nc, 1 << (L - 1));
        level = coarse;
        tl.tops[L] = level; tl.levels = L;
    }
    const uint2 *masks = reinterpret_cast<const uint2 *>(planes);
    k_voxel_pyramid<<<dim3((4 * nc + 255) / 256, nc, nc), blockdim, 0, s>>>(nc, offcy, tl, masks, masks + (size_t)nc * nc * nc, cells);
    return hipGetLastError();
}

// dst[i * part + j] = src[CELL_U4 * i + j]: the records, or the column-top words, out of the 128-byte cells
__global__ __launch_bounds__(256) void k_copy_cell_part(const uint4 *src, uint4 *dst, size_t count, unsigned part)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= count * part) return;
    const size_t i = t / part, j = t % part;
    dst[t] = src[CELL_U4 * i + j];
}

hipError_t launch_copy_cell_part(const uint4 *src, uint4 *dst, size_t count, unsigned part, hipStream_t s)
{
    if (count == 0 || part == 0) return hipSuccess;
    k_copy_cell_part<<<(unsigned)((count * part + 255) / 256), 256, 0, s>>>(src, dst, count, part);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_any_nonzero(const uint16_t *p, size_t count, uint32_t *flag)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    unsigned any = 0;
    for (; i < count; i += stride) any |= p[i];
    if (__any(any != 0) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

hipError_t launch_any_nonzero_u16(const uint16_t *p, size_t count, uint32_t *flag, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const unsigned blocks = (unsigned)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
    k_any_nonzero<<<blocks ? blocks : 1, 256, 0, s>>>(p, count, flag);
    return hipGetLastError();
}

}  // namespace vrt
