// vrt_kernels.h -- host-side launchers of the HIP kernels and the parameter blocks
// they take.  Internal to libvrt.so (the public surface is include/vrt.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// Margin of the exit-point test of the box traversal (vrt_trace.hip), relative to |position| + |origin| + 1
#ifndef VRT_MARGIN
#define VRT_MARGIN 0x1.0p-20f
#endif

namespace vrt {

// One cell of the skip structure (TraceParams::cells): a 128-byte line
constexpr unsigned CELL_U4 = 8;     // uint4 units per cell
constexpr unsigned CELL_SHIFT = 7;  // log2 of its bytes
constexpr unsigned PYR_U4 = 4;      // uint4 units of column-top words per cell, from cells[CELL_U4 * i + 1]
constexpr unsigned BOX_NONE = 0xffffu;   // entry of a box word: no empty box known, the ray steps voxel by voxel
constexpr unsigned BOX_RISE_ALL = 0xffu;  // ... its rise field: up to the top of the window
constexpr unsigned PLANE_BYTES_PER_CELL = 22;  // launch_build_cells' planes: masks 8, octant distances 8, x planes 2, y planes 4
constexpr int      TOPS_LEVELS = 9; // tables of column tops: squares of 1, 2, .. 256 cell columns

// Records the thread's error message (vrt_last_error) and returns `code`; shared by the files of the C ABI.
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Which rows a launch covers.  Packed row q of the output planes maps to image row
//   row(q) = y0 + q                                   (strip_h == 0: plain band)
//   row(q) = ((q / strip_h) * parts + part) * strip_h + q % strip_h   (interleaved strips)
struct RowMap {
    uint32_t y0;
    uint32_t rows;     // packed rows in this launch
    uint32_t strip_h;  // 0 = plain band
    uint32_t part, parts;
};

// Dispatch-order feedback of the skip kernels.  Launch g records the work of each of its wavefronts
// (trips of the traversal loop); launch g + 1 reduces them to the longest per workgroup row; launch g + 2 derives the row
// at which launch g + 3 starts (the topmost row that is not cheap); rows then run from there to the
// bottom and on from the top.
//   cost[g & 1][(row * grid_x + column) * 4 + wave] = g << 20 | work + 1               by launch g
//   row_longest[g & 1][row]                         = g << 20 | work, of launch g - 1   by launch g
//   start[g & 1]                                    = g << 16 | row                         by launch g - 1
// The generation tags make stale entries read as "nothing known".
struct RowSched {
    uint32_t *cost;         // [2][capacity]; null = feature off
    uint32_t *row_longest;  // [2][256]
    uint32_t *start;        // [2]
    uint32_t  capacity;     // wavefronts per half of `cost`
    uint32_t  generation;   // g of this launch, 1..4095
    uint32_t  use_order;    // 0: earlier launches do not describe this one, run rows top to bottom
    int32_t   fixed_start;  // >= 0: start row given by the host (A/B measurements only)
};

struct TraceParams {
    float    m[16];        // camera transform, column major
    uint32_t w, h;         // frame size
    float    ncy_off;      // (float(h) / float(w)) * 0.5f of raytrace.glsl:121, evaluated on the host
    RowMap   rows;
    int32_t  n;            // voxels per axis
    // loaded area of this camera (raytrace.glsl:19-24), computed once on the host:
    // lo = floor(ro/32)*32 + 16 - n/2, up = lo + n - 1, off = rem_euclid(lo, n)
    int32_t  lo[3], up[3], off[3];
    // what every camera ray's set-up shares, worked out once on the host with the kernels' own arithmetic:
    // a0 = int(floor(ro)), the start voxel (raytrace.glsl:32); thr0 = 0.5 - VRT_MARGIN * (|ro| + 1) per axis
    int32_t  a0[3];
    float    thr0[3];
    int32_t  n_pow2_mask;  // n-1 if n is a power of two, else -1
    int32_t  nc_log2;      // log2(n / 4) if n is a power of two
    uint32_t max_steps;
    uint32_t tile_log2h;   // wave tile is (tile_lanes >> tile_log2h) x (1 << tile_log2h) pixels, 0..5
    uint32_t tile_lanes;   // 64, or 32: wavefronts filled by half (a launch too small to fill the GPU: twice the wavefronts, each
                           // as long as the longest of 32 rays instead of 64)
    uint32_t need_cap;     // max_steps can end a regular ray (max_steps < 3n): test it in the skip loop
    RowSched sched;
    // world
    const void     *block;  // n^3 texels (u8 or u16), linear x fastest
    const uint16_t *dist;   // n^3 texels or nullptr (all zero)
    // skip structure: CELL_BYTES (128) bytes per cell of 4^3 voxels (torus space, linear order), one cache line: the
    // 16-byte record of cell i at cells[CELL_U4 * i], its eight column-top words behind it (bytes 16..79), the rest unused
    //   .x,.y  64-bit occupancy mask, bit = (z&3)*16 + (y&3)*4 + (x&3); zero for an empty cell
    //   .z,.w  eight octant distances, byte o = sx | sy<<1 | sz<<2 (bit set = positive direction):
    //          side of the largest empty cube of cells that starts at the cell and extends into
    //          that octant; 0 = the cell is non-empty, cap 32
    const uint4 *cells;
    int32_t  nc;                // n / 4
    // box words, eight 64-bit words per cell (any cell, empty or not), word w = descending | negative x << 1 |
    // negative z << 2 for the ray's direction of travel, at byte 16 + 8 * w of the cell: for each of the cell's four voxel
    // levels j a 16-bit entry, the empty box a ray crosses from a voxel of that level -- BOX_NONE, or (side - 1) << 8 |
    // rise: `side` cells along x and z from the ray's own cell on the way it travels, and `rise` voxels more along y from
    // the voxel on (BOX_RISE_ALL: up to the top of the window).  The builder (vrt_world.hip, k_voxel_pyramid) puts the
    // better of two boxes there: the square of cell columns that hold no voxel at or above some window level (a table of
    // column tops per square size; valid above the solid voxels of a non-empty cell too), or the octant cube of the
    // record.  Window levels are counted from the bottom of the loaded window, so the words are rebuilt when the window's
    // vertical offset changes, and whenever records change.
    const uint4 *pyr;           // cells + 1
    // Rays that cannot hit anything.  *sky = the window level just above the highest solid voxel of the whole window (written
    // with the box words; null: not known, or the launch wants it left alone): a ray that stays at or above that level
    // meets air only, and all its pixel needs is whether the ray left the window after 1000 iterations or more
    // (raytrace.glsl:73-75; sky_miss in vrt_trace.hip).  sky_pos / sky_neg: per axis the plane through which a ray that travels towards + / - leaves
    // the loaded area (raytrace.glsl:19-24), less the camera's position -- the numerators of the exit parameters.
    const uint32_t *sky;
    float    sky_pos[3], sky_neg[3];
    float    sky_slope;   // 1 / (0.6 n): levels a ray may lose per voxel of path, per level of room, and still be worth the test
    // per-frame tables (device memory, filled by launch_frame_tables with the kernels' own arithmetic;
    // null = evaluate in place): the two divisions of raytrace.glsl:121 per column / per row, and the
    // sine of raytrace.glsl:131 for every voxel coordinate a hit point can floor to
    const float *ncx;       // [w]      float(gx) / float(w) - 0.5
    const float *ncy;       // [h + 1]  float(gy) / float(h) - ncy_off, indexed by gy = h - row
    const float *sin2;      // [3][sin_len]  0.5 * canon_sin(2 * (sin_base[k] + i)): the shading term of raytrace.glsl:131, halved (exactly)
    int32_t  sin_base[3];
    uint32_t sin_len;
    // outputs (device pointers, any may be null)
    uint8_t  *bgra8;
    int32_t  *voxel;
    int32_t  *face;
    uint32_t *steps;
    float    *t;
    float    *hit_point;
    float    *rgba;
    uint8_t  *bgr8;   // three bytes per pixel: the bgra8 pixel without its alpha byte
};

// traversal kernels
hipError_t launch_trace_literal(const TraceParams &p, int voxel_bytes, hipStream_t s);
hipError_t launch_trace_skip(const TraceParams &p, int voxel_bytes, hipStream_t s);
// binned: the workgroup's secondary rays staged in LDS and sorted by direction class (k_trace_secondary_binned);
// else every pixel's rays on the pixel's own lane (k_trace_secondary).  Same outputs.
hipError_t launch_trace_secondary(const TraceParams &p, int voxel_bytes, bool binned, uint8_t *shadow, uint8_t *ao,
                                  uint32_t *sec_steps, hipStream_t s);

hipError_t trace_kernel_occupancy(int kind, int *workgroups_per_cu);  // kind 0: skip, 1: binned secondary, 2: literal

// world kernels
hipError_t launch_terrain(void *block, int voxel_bytes, int n, int x0, int y0, int z0,
                          int ex, int ey, int ez, hipStream_t s);
// `count` chunks of 32^3 voxels at chunk positions xyz[3 * i ..] (host memory), 64 chunks per launch
hipError_t launch_terrain_chunks(void *block, int voxel_bytes, int n, const int32_t *xyz, uint32_t count, hipStream_t s);
// cells = nc^3 x 128 bytes (record, column-top words); planes = PLANE_BYTES_PER_CELL * nc^3 bytes: the cells' occupancy masks side
// by side and the x and y planes of the separable octant distance transform, which must keep their contents between a
// build and the updates that follow it
hipError_t launch_build_cells(const void *block, int voxel_bytes, int n, uint4 *cells, uint8_t *planes, hipStream_t s);
// The same for the cells whose voxels changed, a box of the torus (origin in [0, nc), extent 1..nc per axis), and
// for every cell around it whose distances depend on them.
hipError_t launch_update_cells(const void *block, int voxel_bytes, int n, uint4 *cells, uint8_t *planes, const int origin[3],
                               const int extent[3], hipStream_t s);
// planes and cells: as launch_build_cells left them (the masks and the records' distances are read); tops: scratch of at least
// TOPS_LEVELS * nc * nc uint16_t; offcy = rem_euclid(lower_y, n) / 4; writes the box words of every cell (cells[CELL_U4 * i + 1 .. + 4])
// and *sky, the window level just above the highest solid voxel of all (TraceParams::sky)
hipError_t launch_build_voxel_pyramid(const uint8_t *planes, int nc, int offcy, uint16_t *tops, uint4 *cells, uint32_t *sky, hipStream_t s);
hipError_t launch_frame_tables(float *ncx, float *ncy, float *sin2, uint32_t w, uint32_t h, float ncy_off,
                               const int32_t sin_base[3], uint32_t sin_len, hipStream_t s);
// dst[i * part + j] = src[CELL_U4 * i + j], j < part: the records (src = cells, part = 1) or the column-top words
// (src = pyr, part = PYR_U4) out of the cells
hipError_t launch_copy_cell_part(const uint4 *src, uint4 *dst, size_t count, unsigned part, hipStream_t s);
hipError_t launch_any_nonzero_u16(const uint16_t *p, size_t count, uint32_t *flag, hipStream_t s);

// Gathered strip buffers (parts x part_stride_rows packed rows of row_bytes each) -> frame of h rows.
hipError_t launch_assemble_strips(const void *gathered, void *frame, uint32_t row_bytes, uint32_t h,
                                  uint32_t strip_h, uint32_t parts, uint32_t part_stride_rows, hipStream_t s);

// The same from parts that hold BGR8 rows (3 * w bytes each) into a BGRA8 frame; w a multiple of 4, gathered 4-byte and
// frame 16-byte aligned.
hipError_t launch_assemble_strips_bgr(const void *gathered, void *frame, uint32_t w, uint32_t h, uint32_t strip_h,
                                      uint32_t parts, uint32_t part_stride_rows, hipStream_t s);

// BGRA8 <-> BGR8 (alpha dropped / restored as 255); `pixels` a multiple of 4, both buffers 16- / 4-byte aligned
hipError_t launch_pack_bgr(const void *bgra, void *bgr, size_t pixels, hipStream_t s);
hipError_t launch_unpack_bgra(const void *bgr, void *bgra, size_t pixels, hipStream_t s);

}  // namespace vrt
