/*
 * vrt_oracle.c -- TEST INFRASTRUCTURE ONLY (see vrt_oracle.h).
 *
 * Plain-C restatement of the reference's trace and terrain shaders.  Every
 * function cites the shader lines it follows.  Build with
 *     gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math
 * so that every float operation below is one IEEE-754 binary32 operation in
 * the order written.
 *
 * PARITY UNPINNED (no golden vectors exist upstream; the shader cannot run
 * here).  Canonical choices where GLSL/SPIR-V leaves the result open:
 *   - float->int conversion saturates and maps NaN to 0;
 *   - integer +,-,* wrap (two's complement);
 *   - clamp(x,lo,hi) = min(max(x,lo),hi), also when lo > hi;
 *   - round() = round-half-to-even (rintf);
 *   - normalize(v) = v / sqrt((x*x + y*y) + z*z), one IEEE division per component;
 *   - vector*matrix and dot products sum left to right;
 *   - sin() = orc_sin below (binary32 three-term reduction against pi + odd
 *     Taylor polynomial through r^11, every operation in the order written);
 *   - UNORM8 store = floor(clamp01(v)*255 + 0.5), NaN -> 0;
 *   - a step cap (max_steps) ends rays the reference would trace forever.
 */
#include "vrt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define CHUNK_SIZE 32 /* raytrace.glsl:10, terrain_gen.comp:12 */

/* ------------------------------------------------------------------ */
/* scalar primitives                                                   */
/* ------------------------------------------------------------------ */

static inline int32_t f2i(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static inline int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
static inline int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

/* util.glsl:1-4 */
static inline int32_t rem_euclid_int(int32_t dividend, int32_t divisor)
{
    int32_t r = dividend % divisor;
    return r >= 0 ? r : r + divisor;
}

/* util.glsl:28-42 -- strict '<', so ties resolve x, then y, then z; a NaN in
 * .x wins because every later comparison is false. */
static inline int argmin3(const float a[3])
{
    int   min_index = 0;
    float min_value = a[0];
    if (a[1] < min_value) { min_value = a[1]; min_index = 1; }
    if (a[2] < min_value) { min_index = 2; }
    return min_index;
}

/* Canonical sine, binary32 throughout.  k = rint(x/pi); the reduction x - k*pi uses pi split
 * into P1 = 3.125 (5 bits: k*P1 is exact for |k| < 2^19, and x - k*P1 is exact), P2 (8 bits)
 * and P3; then the odd Taylor polynomial through r^11 on [-pi/2, pi/2].  Absolute error
 * below 3e-7 for |x| < 2^20; non-finite x or |x| >= 2^20 gives NaN. */
float orc_sin(float x)
{
    const float INV_PI = 0x1.45f306p-2f;
    const float P1 = 3.125f, P2 = 0x1.1p-6f, P3 = -0x1.2aeef4p-17f;
    if (!(fabsf(x) < 1048576.0f)) return NAN;
    const float k  = rintf(x * INV_PI);
    const float r  = ((x - k * P1) - k * P2) - k * P3;
    const float r2 = r * r;
    float p = -0x1.ae6456p-26f;       /* -1/11! */
    p = p * r2 + 0x1.71de3ap-19f;     /*  1/9!  */
    p = p * r2 + -0x1.a01a02p-13f;    /* -1/7!  */
    p = p * r2 + 0x1.111112p-7f;      /*  1/5!  */
    p = p * r2 + -0x1.555556p-3f;     /* -1/3!  */
    float s = r + (r * r2) * p;
    if ((int32_t)k & 1) s = -s;
    return s;
}

static inline uint8_t unorm8(float v)
{
    v = (v > 0.0f) ? v : 0.0f; /* NaN -> 0 */
    v = (v < 1.0f) ? v : 1.0f;
    return (uint8_t)f2i(floorf(v * 255.0f + 0.5f));
}

/* ------------------------------------------------------------------ */
/* world access                                                        */
/* ------------------------------------------------------------------ */

static inline size_t texel_index(const int32_t p[3], int32_t n)
{
    /* util.glsl:6-8 rem_euclid_ivec3 + linear x-fastest image layout */
    size_t x = (size_t)rem_euclid_int(p[0], n);
    size_t y = (size_t)rem_euclid_int(p[1], n);
    size_t z = (size_t)rem_euclid_int(p[2], n);
    return (z * (size_t)n + y) * (size_t)n + x;
}

static inline uint32_t load_block(const orc_world *wd, const int32_t p[3])
{
    size_t i = texel_index(p, wd->n);
    return wd->voxel_bytes == 1 ? ((const uint8_t *)wd->block)[i] : ((const uint16_t *)wd->block)[i];
}

static inline uint32_t load_dist(const orc_world *wd, const int32_t p[3])
{
    if (!wd->dist) return 0;
    return ((const uint16_t *)wd->dist)[texel_index(p, wd->n)];
}

/* raytrace.glsl:19-24.  Bounds depend only on the camera position. */
static void loaded_area_bounds(const float ro[3], int32_t n, int32_t lower[3], int32_t upper[3])
{
    for (int k = 0; k < 3; ++k) {
        int32_t mid = wadd(wmul(f2i(floorf(ro[k] / (float)CHUNK_SIZE)), CHUNK_SIZE), CHUNK_SIZE / 2);
        lower[k] = wsub(mid, n / 2);
        upper[k] = wsub(wadd(mid, n / 2), 1);
    }
}

static inline int inside_loaded_area(const int32_t pos[3], const int32_t lower[3], const int32_t upper[3])
{
    return lower[0] < pos[0] && lower[1] < pos[1] && lower[2] < pos[2] &&
           pos[0] < upper[0] && pos[1] < upper[1] && pos[2] < upper[2];
}

/* ------------------------------------------------------------------ */
/* traversal: raytrace.glsl:26-77 (single_ray_optimized)               */
/* ------------------------------------------------------------------ */

void orc_trace_ray(const orc_world *wd, const float ro[3], const float rd[3], orc_ray *out)
{
    orc_trace_ray_from(wd, ro, ro, rd, out); /* raytrace.glsl:16,123: ro = player_position */
}

void orc_trace_ray_from(const orc_world *wd, const float player_position[3], const float ro[3],
                        const float rd[3], orc_ray *out)
{
    float   inv_rd[3];
    int32_t s01[3], s11[3], anchor[3], cp[3], lower[3], upper[3];
    memset(out, 0, sizeof *out);

    loaded_area_bounds(player_position, wd->n, lower, upper);
    for (int k = 0; k < 3; ++k) {
        inv_rd[k] = 1.0f / rd[k];                 /* :27 */
        s01[k]    = rd[k] > 0.0f ? 1 : 0;         /* :29 */
        s11[k]    = s01[k] * 2 - 1;               /* :30 */
        anchor[k] = f2i(floorf(ro[k]));           /* :32 */
    }
    int32_t free_dist = (int32_t)load_dist(wd, anchor); /* :33 */
    for (int k = 0; k < 3; ++k)                   /* :35 */
        cp[k] = wadd(wadd(anchor[k], s01[k]), wmul(s11[k], free_dist));

    uint32_t steps = 0;                           /* :40 panic_counter */
    int      capped = 0;
    for (;;) {
        if (!inside_loaded_area(cp, lower, upper)) break; /* :43 */
        if (steps >= wd->max_steps) { capped = 1; break; }  /* canonical cap */
        steps++;                                  /* :44 */
        float t[3];
        for (int k = 0; k < 3; ++k) t[k] = ((float)cp[k] - ro[k]) * inv_rd[k]; /* :45 */
        int   nx = argmin3(t);                    /* :46 */
        float tk = t[nx];
        float p[3];
        for (int k = 0; k < 3; ++k) p[k] = ro[k] + rd[k] * tk;
        for (int k = 0; k < 3; ++k) {
            int32_t clamp_p = wsub(cp[k], wmul(s11[k], wadd(wmul(2, free_dist), 1))); /* :48 */
            int32_t lo = wadd(imin(cp[k], clamp_p), 1);
            int32_t hi = wsub(imax(cp[k], clamp_p), 1);
            anchor[k] = imin(imax(f2i(floorf(p[k])), lo), hi);                     /* :49 */
        }
        anchor[nx] = wadd(wsub(f2i(rintf(p[nx])), 1), s01[nx]);                    /* :50 */

        uint32_t stored = load_block(wd, anchor);  /* :52 */
        if (stored > 0) {                          /* :53-60 */
            out->hit       = 1;
            out->block_id  = stored;
            out->normal[nx] = (float)(-s11[nx]);
            out->face      = 2 * nx + (s11[nx] < 0 ? 1 : 0);
            out->t         = tk;
            for (int k = 0; k < 3; ++k) { out->hit_point[k] = p[k]; out->voxel[k] = anchor[k]; }
            out->steps = steps;
            return;
        }
        free_dist = (int32_t)load_dist(wd, anchor); /* :62 */
        for (int k = 0; k < 3; ++k)                 /* :67 */
            cp[k] = wadd(wadd(anchor[k], s01[k]), wmul(s11[k], free_dist));
    }
    /* :72-76 */
    out->hit = 0;
    out->capped = capped;
    out->face = capped ? -2 : -1;
    out->steps = steps;
    out->t = -1.0f;
    for (int k = 0; k < 3; ++k) out->voxel[k] = anchor[k];
    if (steps >= 1000) { out->normal[0] = 0.2f; out->normal[1] = 1.0f; out->normal[2] = 0.8f; }
}

/* ------------------------------------------------------------------ */
/* ray generation: raytrace.glsl:120-123                               */
/* ------------------------------------------------------------------ */

void orc_raygen(const float m[16], uint32_t w, uint32_t h, uint32_t gx, uint32_t gy, float ro[3], float rd[3])
{
    float fw = (float)w, fh = (float)h;
    float ncx = (float)gx / fw - 0.5f;
    float ncy = (float)gy / fh - (fh / fw * 0.5f);
    float d[3];
    /* row vector (ncx, ncy, 1, 1) times column-major mat4: component i is the
     * dot product with column i */
    for (int i = 0; i < 3; ++i)
        d[i] = ((ncx * m[4 * i + 0] + ncy * m[4 * i + 1]) + 1.0f * m[4 * i + 2]) + 1.0f * m[4 * i + 3];
    float len = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    for (int i = 0; i < 3; ++i) rd[i] = d[i] / len;
    ro[0] = m[12]; ro[1] = m[13]; ro[2] = m[14]; /* raytrace.glsl:16 */
}

/* ------------------------------------------------------------------ */
/* per-pixel entry: render.comp:6-14 + raytrace.glsl:117-134           */
/* ------------------------------------------------------------------ */

static void shade_pixel(const orc_world *wd, const float m[16], uint32_t w, uint32_t h,
                        uint32_t row, uint32_t col, orc_ray *r, float rgba[4])
{
    /* render.comp:13 stores invocation (x, y) at row h - y, so the pixel in
     * row `row` is produced by invocation y = h - row. */
    uint32_t gx = col, gy = h - row;
    float ro[3], rd[3];
    orc_raygen(m, w, h, gx, gy, ro, rd);
    orc_trace_ray(wd, ro, rd, r);
    if (r->hit) { /* raytrace.glsl:129-131 */
        for (int k = 0; k < 3; ++k)
            rgba[k] = r->normal[k] + 0.5f * orc_sin(floorf(r->hit_point[k] + 0.001f) * 2.0f);
    } else {
        for (int k = 0; k < 3; ++k) rgba[k] = r->normal[k];
    }
    rgba[3] = 1.0f;
    /* render.comp:10-12 crosshair */
    float dx = (float)gx - (float)(int32_t)(w / 2);
    float dy = (float)gy - (float)(int32_t)(h / 2);
    if (sqrtf(dx * dx + dy * dy) < 1.2f) { rgba[0] = 1.0f; rgba[1] = 0.3f; rgba[2] = 0.3f; rgba[3] = 1.0f; }
}

typedef struct job {
    const orc_world *wd; const float *m; uint32_t w, h, y0, y1; const orc_outputs *out;
    int tid, nthreads;
} job;

static void *trace_rows(void *arg)
{
    job *j = (job *)arg;
    for (uint32_t row = j->y0 + (uint32_t)j->tid; row < j->y1; row += (uint32_t)j->nthreads) {
        for (uint32_t col = 0; col < j->w; ++col) {
            orc_ray r; float rgba[4];
            shade_pixel(j->wd, j->m, j->w, j->h, row, col, &r, rgba);
            size_t i = (size_t)(row - j->y0) * j->w + col;
            const orc_outputs *o = j->out;
            if (o->bgra8) {
                o->bgra8[4 * i + 0] = unorm8(rgba[2]);
                o->bgra8[4 * i + 1] = unorm8(rgba[1]);
                o->bgra8[4 * i + 2] = unorm8(rgba[0]);
                o->bgra8[4 * i + 3] = unorm8(rgba[3]);
            }
            if (o->voxel) { o->voxel[3 * i] = r.voxel[0]; o->voxel[3 * i + 1] = r.voxel[1]; o->voxel[3 * i + 2] = r.voxel[2]; }
            if (o->face) o->face[i] = r.face;
            if (o->steps) o->steps[i] = r.steps;
            if (o->t) o->t[i] = r.t;
            if (o->hit_point) { o->hit_point[3 * i] = r.hit_point[0]; o->hit_point[3 * i + 1] = r.hit_point[1]; o->hit_point[3 * i + 2] = r.hit_point[2]; }
            if (o->rgba) memcpy(o->rgba + 4 * i, rgba, sizeof rgba);
        }
    }
    return NULL;
}

int orc_trace(const orc_world *wd, const float m[16], uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
              const orc_outputs *out, int nthreads)
{
    if (!wd || !wd->block || !out || w == 0 || h == 0 || y0 > y1 || y1 > h) return -1;
    if (wd->voxel_bytes != 1 && wd->voxel_bytes != 2) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256]; job jobs[256];
    for (int i = 0; i < nthreads; ++i) {
        jobs[i] = (job){wd, m, w, h, y0, y1, out, i, nthreads};
        if (i + 1 < nthreads) pthread_create(&th[i], NULL, trace_rows, &jobs[i]);
    }
    trace_rows(&jobs[nthreads - 1]);
    for (int i = 0; i + 1 < nthreads; ++i) pthread_join(th[i], NULL);
    return 0;
}

/* ------------------------------------------------------------------ */
/* secondary rays (extension, no reference counterpart)                */
/* ------------------------------------------------------------------ */

static inline uint32_t pcg_hash(uint32_t v)
{
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word  = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

void orc_secondary_dir(uint32_t pixel_index, uint32_t index, int32_t face, float rd[3])
{
    if (index == 0) { /* light direction */
        const float lx = 0.4f, ly = 0.8f, lz = 0.45f;
        const float len = sqrtf((lx * lx + ly * ly) + lz * lz);
        rd[0] = lx / len; rd[1] = ly / len; rd[2] = lz / len;
        return;
    }
    uint32_t hsh = pcg_hash(0x5EEDu ^ pixel_index);
    for (uint32_t j = 1; j < index; ++j) { hsh = pcg_hash(hsh); hsh = pcg_hash(hsh); }
    const uint32_t h1 = hsh, h2 = pcg_hash(hsh);
    const float u1 = (float)(h1 >> 8) * 0x1.0p-24f, u2 = (float)(h2 >> 8) * 0x1.0p-24f;
    const float r = sqrtf(u1), phi = 6.2831855f * u2;
    const float tx = r * orc_sin(phi + 1.5707964f); /* cos */
    const float ty = r * orc_sin(phi);
    const float tz = sqrtf(1.0f - u1);
    const int   k = face >> 1;
    const float sgn = (face & 1) ? 1.0f : -1.0f;   /* faces 1,3,5 have the + normal */
    /* normal axis k gets +-tz, the two other axes (in cyclic order) get tx, ty */
    rd[k] = sgn * tz; rd[(k + 1) % 3] = tx; rd[(k + 2) % 3] = ty;
}

typedef struct sec_job {
    const orc_world *wd; const float *m; uint32_t w, h, y0, y1; uint8_t *shadow, *ao; uint32_t *steps;
    int tid, nthreads;
} sec_job;

static void *secondary_rows(void *arg)
{
    sec_job *j = (sec_job *)arg;
    for (uint32_t row = j->y0 + (uint32_t)j->tid; row < j->y1; row += (uint32_t)j->nthreads) {
        for (uint32_t col = 0; col < j->w; ++col) {
            const size_t i = (size_t)(row - j->y0) * j->w + col;
            float ro[3], rd[3];
            orc_ray pr;
            orc_raygen(j->m, j->w, j->h, col, j->h - row, ro, rd);
            orc_trace_ray(j->wd, ro, rd, &pr);
            uint32_t sh = 0, ao = 0, st = 0;
            if (pr.hit) {
                float so[3];
                for (int k = 0; k < 3; ++k) so[k] = pr.hit_point[k] + pr.normal[k] * 1e-3f;
                for (uint32_t s = 0; s < 5; ++s) {
                    float sd[3];
                    orc_ray sr;
                    orc_secondary_dir(row * j->w + col, s, pr.face, sd);
                    orc_trace_ray_from(j->wd, ro, so, sd, &sr);
                    st += sr.steps;
                    if (sr.hit) { if (s == 0) sh = 1; else ao++; }
                }
            }
            if (j->shadow) j->shadow[i] = (uint8_t)sh;
            if (j->ao) j->ao[i] = (uint8_t)ao;
            if (j->steps) j->steps[i] = st;
        }
    }
    return NULL;
}

int orc_trace_secondary(const orc_world *wd, const float m[16], uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                        uint8_t *shadow, uint8_t *ao, uint32_t *sec_steps, int nthreads)
{
    if (!wd || !wd->block || w == 0 || h == 0 || y0 > y1 || y1 > h) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256]; sec_job jobs[256];
    for (int i = 0; i < nthreads; ++i) {
        jobs[i] = (sec_job){wd, m, w, h, y0, y1, shadow, ao, sec_steps, i, nthreads};
        if (i + 1 < nthreads) pthread_create(&th[i], NULL, secondary_rows, &jobs[i]);
    }
    secondary_rows(&jobs[nthreads - 1]);
    for (int i = 0; i + 1 < nthreads; ++i) pthread_join(th[i], NULL);
    return 0;
}

/* ------------------------------------------------------------------ */
/* terrain: noise.glsl:4-74 and terrain_gen.comp:14-24                 */
/* ------------------------------------------------------------------ */

/* GLSL mod(x, y) = x - y*floor(x/y) */
static inline float glsl_mod(float x, float y) { return x - y * floorf(x / y); }
/* noise.glsl:4 */
static inline float permute1(float x) { return glsl_mod(((x * 34.0f) + 1.0f) * x, 289.0f); }
/* noise.glsl:5 */
static inline float taylor_inv_sqrt(float r) { return 1.79284291400159f - 0.85373472095314f * r; }
/* GLSL step(edge, x) = x < edge ? 0 : 1 */
static inline float stepf(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
static inline float dot3(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

float orc_snoise(float vx, float vy, float vz)
{
    const float Cx = (float)(1.0 / 6.0), Cy = (float)(1.0 / 3.0); /* :8 */
    const float v[3] = {vx, vy, vz};
    float i[3], x0[3], x1[3], x2[3], x3[3], g[3], l[3], i1[3], i2[3];

    /* :12-13 first corner */
    float s = (v[0] * Cy + v[1] * Cy) + v[2] * Cy;
    for (int k = 0; k < 3; ++k) i[k] = floorf(v[k] + s);
    float u = (i[0] * Cx + i[1] * Cx) + i[2] * Cx;
    for (int k = 0; k < 3; ++k) x0[k] = (v[k] - i[k]) + u;

    /* :16-19 other corners: g = step(x0.yzx, x0.xyz) */
    g[0] = stepf(x0[1], x0[0]); g[1] = stepf(x0[2], x0[1]); g[2] = stepf(x0[0], x0[2]);
    for (int k = 0; k < 3; ++k) l[k] = 1.0f - g[k];
    /* l.zxy */
    const float lz[3] = {l[2], l[0], l[1]};
    for (int k = 0; k < 3; ++k) { i1[k] = fminf(g[k], lz[k]); i2[k] = fmaxf(g[k], lz[k]); }

    /* :22-24 */
    for (int k = 0; k < 3; ++k) {
        x1[k] = (x0[k] - i1[k]) + 1.0f * Cx;
        x2[k] = (x0[k] - i2[k]) + 2.0f * Cx;
        x3[k] = (x0[k] - 1.0f) + 3.0f * Cx;
    }

    /* :27-31 permutations */
    for (int k = 0; k < 3; ++k) i[k] = glsl_mod(i[k], 289.0f);
    const float oz[4] = {0.0f, i1[2], i2[2], 1.0f};
    const float oy[4] = {0.0f, i1[1], i2[1], 1.0f};
    const float ox[4] = {0.0f, i1[0], i2[0], 1.0f};
    float p[4];
    for (int c = 0; c < 4; ++c) {
        float a = permute1(i[2] + oz[c]);
        float b = permute1((a + i[1]) + oy[c]);
        p[c]    = permute1((b + i[0]) + ox[c]);
    }

    /* :35-36 gradients: ns = n_ * D.wyz - D.xzx with D = (0, 0.5, 1, 2) */
    const float n_  = (float)(1.0 / 7.0);
    const float nsx = n_ * 2.0f - 0.0f, nsy = n_ * 0.5f - 1.0f, nsz = n_ * 1.0f - 0.0f;

    float x[4], y[4], h[4];
    for (int c = 0; c < 4; ++c) {
        float j  = p[c] - 49.0f * floorf(p[c] * nsz * nsz); /* :38 */
        float x_ = floorf(j * nsz);                         /* :40 */
        float y_ = floorf(j - 7.0f * x_);                   /* :41 */
        x[c] = x_ * nsx + nsy;                              /* :43 */
        y[c] = y_ * nsx + nsy;                              /* :44 */
        h[c] = (1.0f - fabsf(x[c])) - fabsf(y[c]);          /* :45 */
    }
    const float b0[4] = {x[0], x[1], y[0], y[1]};           /* :47 */
    const float b1[4] = {x[2], x[3], y[2], y[3]};           /* :48 */
    float s0[4], s1[4], sh[4];
    for (int c = 0; c < 4; ++c) {
        s0[c] = floorf(b0[c]) * 2.0f + 1.0f;                /* :50 */
        s1[c] = floorf(b1[c]) * 2.0f + 1.0f;                /* :51 */
        sh[c] = -stepf(h[c], 0.0f);                         /* :52 */
    }
    /* :54-55  a0 = b0.xzyw + s0.xzyw*sh.xxyy ; a1 = b1.xzyw + s1.xzyw*sh.zzww */
    const float a0[4] = {b0[0] + s0[0] * sh[0], b0[2] + s0[2] * sh[0], b0[1] + s0[1] * sh[1], b0[3] + s0[3] * sh[1]};
    const float a1[4] = {b1[0] + s1[0] * sh[2], b1[2] + s1[2] * sh[2], b1[1] + s1[1] * sh[3], b1[3] + s1[3] * sh[3]};
    /* :57-60 */
    float p0[3] = {a0[0], a0[1], h[0]};
    float p1[3] = {a0[2], a0[3], h[1]};
    float p2[3] = {a1[0], a1[1], h[2]};
    float p3[3] = {a1[2], a1[3], h[3]};
    /* :63-67 */
    float n0 = taylor_inv_sqrt(dot3(p0, p0)), n1 = taylor_inv_sqrt(dot3(p1, p1));
    float n2 = taylor_inv_sqrt(dot3(p2, p2)), n3 = taylor_inv_sqrt(dot3(p3, p3));
    for (int k = 0; k < 3; ++k) { p0[k] *= n0; p1[k] *= n1; p2[k] *= n2; p3[k] *= n3; }
    /* :70-73 */
    float m[4] = {fmaxf(0.6f - dot3(x0, x0), 0.0f), fmaxf(0.6f - dot3(x1, x1), 0.0f),
                  fmaxf(0.6f - dot3(x2, x2), 0.0f), fmaxf(0.6f - dot3(x3, x3), 0.0f)};
    for (int c = 0; c < 4; ++c) m[c] = m[c] * m[c];
    const float d[4] = {dot3(p0, x0), dot3(p1, x1), dot3(p2, x2), dot3(p3, x3)};
    float acc = (((m[0] * m[0]) * d[0] + (m[1] * m[1]) * d[1]) + (m[2] * m[2]) * d[2]) + (m[3] * m[3]) * d[3];
    return 42.0f * acc;
}

/* terrain_gen.comp:15-21 */
static inline uint32_t terrain_block(int32_t px, int32_t py, int32_t pz)
{
    float noise_val = orc_snoise((float)px * 0.015f, (float)py * 0.015f, (float)pz * 0.015f) * 6.0f - (float)py;
    return noise_val > 0.0f ? 1u : 0u;
}

static inline void store_block(void *block, int32_t vb, int32_t n, int32_t px, int32_t py, int32_t pz, uint32_t v)
{
    const int32_t p[3] = {px, py, pz};
    size_t i = texel_index(p, n); /* terrain_gen.comp:23 */
    if (vb == 1) ((uint8_t *)block)[i] = (uint8_t)v; else ((uint16_t *)block)[i] = (uint16_t)v;
}

typedef struct fill_job { void *block; int32_t vb, n; const int32_t *wmin; int tid, nthreads; } fill_job;

static void *fill_slices(void *arg)
{
    fill_job *j = (fill_job *)arg;
    for (int32_t z = j->tid; z < j->n; z += j->nthreads)
        for (int32_t y = 0; y < j->n; ++y)
            for (int32_t x = 0; x < j->n; ++x) {
                int32_t px = j->wmin[0] + x, py = j->wmin[1] + y, pz = j->wmin[2] + z;
                store_block(j->block, j->vb, j->n, px, py, pz, terrain_block(px, py, pz));
            }
    return NULL;
}

void orc_terrain_fill(void *block, int32_t vb, int32_t n, const int32_t wmin[3], int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256]; fill_job jobs[256];
    for (int i = 0; i < nthreads; ++i) {
        jobs[i] = (fill_job){block, vb, n, wmin, i, nthreads};
        if (i + 1 < nthreads) pthread_create(&th[i], NULL, fill_slices, &jobs[i]);
    }
    fill_slices(&jobs[nthreads - 1]);
    for (int i = 0; i + 1 < nthreads; ++i) pthread_join(th[i], NULL);
}

void orc_terrain_chunk(void *block, int32_t vb, int32_t n, int32_t cx, int32_t cy, int32_t cz)
{
    /* terrain_gen.comp:15: pos = chunk_position*32 + gl_GlobalInvocationID, dispatch [4,4,4] x local 8^3 */
    for (int32_t z = 0; z < CHUNK_SIZE; ++z)
        for (int32_t y = 0; y < CHUNK_SIZE; ++y)
            for (int32_t x = 0; x < CHUNK_SIZE; ++x) {
                int32_t px = cx * CHUNK_SIZE + x, py = cy * CHUNK_SIZE + y, pz = cz * CHUNK_SIZE + z;
                store_block(block, vb, n, px, py, pz, terrain_block(px, py, pz));
            }
}
