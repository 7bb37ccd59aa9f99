/*
 * vrt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, strict IEEE-754 binary32, no FMA contraction) of the
 * reference's GLSL compute shaders:
 *   src/graphics/shaders/rendering/render.comp:6-14        entry, crosshair, flipped store
 *   src/graphics/shaders/rendering/raytrace.glsl:16-24     globals, loaded-area test
 *   src/graphics/shaders/rendering/raytrace.glsl:26-77     single_ray_optimized (live traversal)
 *   src/graphics/shaders/rendering/raytrace.glsl:117-134   raycast: ray generation + shading
 *   src/graphics/shaders/rendering/util.glsl:1-8,28-42     rem_euclid, argmin
 *   src/terrain/terrain_gen/terrain_gen.comp:14-24         terrain function
 *   src/terrain/terrain_gen/noise.glsl:4-74                3-D simplex noise
 *
 * PARITY UNPINNED: the reference ships no tests, golden images or known-answer
 * vectors for this path, and it can neither be compiled nor executed here (GLSL
 * compute shader behind Vulkan; no Rust toolchain, no GLSL compiler, no Vulkan
 * device).  This oracle is therefore pinned only by hand-derived known-answer
 * rays and by an independent numpy restatement kept under tests/.
 *
 * Nothing in the shipped library links, loads or calls this file.  Only tests/,
 * the smoke check and the CPU-baseline leg of the benchmark may use it.
 *
 * Where GLSL leaves a result implementation-defined the oracle fixes one
 * canonical choice (documented at each function in vrt_oracle.c); the HIP
 * kernels implement the same choices and are compared bit for bit.
 */
#ifndef VRT_ORACLE_H
#define VRT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* World as the shader sees it: two N^3 3-D images addressed by the Euclidean
 * remainder of the world voxel coordinate (toroidal window).  Host layout is
 * linear, x fastest: texel (x,y,z) at ((z*n + y)*n + x). */
typedef struct orc_world {
    const void *block;      /* n^3 texels, voxel_bytes each; 0 = air            */
    const void *dist;       /* n^3 u16 texels or NULL (all zero, as shipped)    */
    int32_t     voxel_bytes;/* 1 (packed u8) or 2 (reference r16ui)             */
    int32_t     n;          /* texels per axis                                  */
    uint32_t    max_steps;  /* step cap (the reference has none; see .c)        */
} orc_world;

/* Result of one ray. */
typedef struct orc_ray {
    int32_t  hit;           /* 1 hit, 0 miss                                    */
    int32_t  capped;        /* 1 if the step cap ended the ray                  */
    int32_t  voxel[3];      /* hit voxel (world coords); on a miss the last anchor */
    int32_t  face;          /* 0:-x 1:+x 2:-y 3:+y 4:-z 5:+z (surface normal); -1 miss; -2 capped */
    uint32_t steps;         /* loop iterations executed (= panic_counter)       */
    uint32_t block_id;      /* stored block value at the hit voxel, 0 on miss   */
    float    t;             /* t_distances[next_xyz] at the hit; -1 on miss     */
    float    hit_point[3];  /* ro + rd*t; zeros on miss                         */
    float    normal[3];     /* surface_normal as the shader returns it          */
} orc_ray;

/* Per-pixel output planes for a band of rows [y0,y1); any pointer may be NULL.
 * Pixel (row r, col c) of the band is stored at index (r - y0)*w + c. */
typedef struct orc_outputs {
    uint8_t  *bgra8;        /* 4 bytes per pixel, memory order B,G,R,A          */
    int32_t  *voxel;        /* 3 per pixel                                      */
    int32_t  *face;
    uint32_t *steps;
    float    *t;
    float    *hit_point;    /* 3 per pixel                                      */
    float    *rgba;         /* 4 per pixel, unclamped shader colour             */
} orc_outputs;

/* raytrace.glsl:120-123 */
void  orc_raygen(const float cam_col_major[16], uint32_t w, uint32_t h,
                 uint32_t gid_x, uint32_t gid_y, float ro[3], float rd[3]);
/* raytrace.glsl:26-77 */
void  orc_trace_ray(const orc_world *wd, const float ro[3], const float rd[3], orc_ray *out);
/* the same traversal for a ray that does not start at the camera: the loaded area
 * (raytrace.glsl:19-24) is still the one around `player_position` */
void  orc_trace_ray_from(const orc_world *wd, const float player_position[3], const float ro[3],
                         const float rd[3], orc_ray *out);
/* render.comp:6-14 for every pixel of rows [y0,y1) */
int   orc_trace(const orc_world *wd, const float cam_col_major[16],
                uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                const orc_outputs *out, int nthreads);
/* Secondary rays per primary hit -- NO reference counterpart (the shader casts primary rays
 * only); defined here so that an incoherent-ray workload has an exact CPU statement.
 * For every pixel of rows [y0,y1) whose primary ray hits: one shadow ray towards the light
 * normalize(0.4, 0.8, 0.45) and four ambient-occlusion rays, cosine distributed about the face
 * normal from the hash chain pcg(0x5EED ^ (row*w + col)); origin = hit_point + normal*1e-3;
 * all traced by the same traversal in the camera's loaded area.
 * shadow: 1 if the shadow ray hits, else 0; ao: number of AO rays that hit (0..4);
 * sec_steps: sum of the loop iterations of the five rays.  All 0 for primary misses. */
int   orc_trace_secondary(const orc_world *wd, const float cam_col_major[16], uint32_t w, uint32_t h,
                          uint32_t y0, uint32_t y1, uint8_t *shadow, uint8_t *ao, uint32_t *sec_steps,
                          int nthreads);
/* direction of secondary ray `index` (0 shadow, 1..4 AO) of pixel `pixel_index` on `face` */
void  orc_secondary_dir(uint32_t pixel_index, uint32_t index, int32_t face, float rd[3]);
/* canonical sine used by the shading term (raytrace.glsl:130) */
float orc_sin(float x);
/* noise.glsl:7-74 */
float orc_snoise(float x, float y, float z);
/* terrain_gen.comp:14-24 evaluated for every world voxel of the cube
 * [wmin, wmin+n)^3 and stored at its toroidal texel. */
void  orc_terrain_fill(void *block, int32_t voxel_bytes, int32_t n,
                       const int32_t wmin[3], int nthreads);
/* terrain_gen.comp:14-24 for one chunk (32^3 voxels at chunk*32). */
void  orc_terrain_chunk(void *block, int32_t voxel_bytes, int32_t n,
                        int32_t cx, int32_t cy, int32_t cz);
#ifdef __cplusplus
}
#endif
#endif
