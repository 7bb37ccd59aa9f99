/*
 * vrt.h -- C ABI of the MI355X voxel ray tracer (libvrt.so).
 *
 * Drop-in boundary for ONE path of Solvir777/Raytracing: the per-pixel ray
 * generation + grid traversal + shading that the reference runs as a Vulkan
 * compute dispatch.  The reference has no FFI for this path; its seam is the
 * `GraphicsCore` inherent impl (src/graphics/mod.rs).  Each entry point below
 * names the reference method it replaces.  A Rust (or any other) host binds
 * these symbols; no C++ or torch types cross the boundary.
 *
 * Conventions: every call returns VRT_OK (0) or a negative VRT_E_* code and
 * never throws or aborts; vrt_last_error() gives a thread-local message.
 * Inputs are borrowed for the duration of the call; outputs are caller owned.
 * A context is bound to one HIP device and is not thread-safe.  There is no
 * CPU fallback: without a usable HIP device vrt_create fails.
 */
#ifndef VRT_H
#define VRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRT_OK              0
#define VRT_E_INVALID_ARG  (-1)
#define VRT_E_NO_DEVICE    (-2)
#define VRT_E_OUT_OF_MEMORY (-3)
#define VRT_E_UNSUPPORTED  (-4)
#define VRT_E_HIP          (-5)
#define VRT_E_NO_WORLD     (-6)

/* `GraphicsCore::CHUNK_SIZE` (src/graphics/mod.rs:147). */
#define VRT_CHUNK_SIZE 32

/* Voxel storage, as uploaded and as held in HBM. */
#define VRT_VOXEL_REF_U16U16 0u /* two r16ui images: block ids + distance field (src/graphics/mod.rs:354-382) */
#define VRT_VOXEL_PACKED_U8  1u /* one u8 per voxel, 0 = air; no distance image */

/* Which traversal kernel vrt_trace launches. */
#define VRT_TRACE_AUTO    0u /* SKIP when the distance image is all zero (as shipped), else LITERAL */
#define VRT_TRACE_LITERAL 1u /* one voxel per iteration, straight restatement of raytrace.glsl:26-77 */
#define VRT_TRACE_SKIP    2u /* exact empty-space skipping over 4^3-cell occupancy masks + cell distance field */

#define VRT_MEM_HOST   0u
#define VRT_MEM_DEVICE 1u

typedef struct vrt_ctx vrt_ctx;

typedef struct vrt_config {
    uint32_t struct_size;   /* sizeof(vrt_config) */
    int32_t  device;        /* HIP device ordinal */
    uint32_t grid_n;        /* voxels per axis of the toroidal window; multiple of 8, 8..4096.
                               The reference uses 32*(2*render_distance+3) = 160 (src/graphics/mod.rs:359) */
    uint32_t voxel_format;  /* VRT_VOXEL_* */
    uint32_t max_steps;     /* per-ray iteration cap; 0 = 4*grid_n.  The reference disables its own
                               cap (raytrace.glsl:43) and can spin for ever on degenerate rays. */
    uint32_t trace_mode;    /* VRT_TRACE_* */
} vrt_config;

/* Per-pixel output planes for rows [y0,y1) of the frame; any pointer may be NULL.
 * Pixel (row r, column c) is element (r - y0)*w + c of each plane.  Row r is what
 * the reference stores at image row r: invocation y = h - r (render.comp:13). */
typedef struct vrt_outputs {
    uint32_t struct_size;   /* sizeof(vrt_outputs) */
    uint32_t memory;        /* VRT_MEM_HOST or VRT_MEM_DEVICE: where ALL non-NULL planes live */
    uint8_t  *bgra8;        /* 4 B/pixel, byte order B,G,R,A = the B8G8R8A8_UNORM swapchain image (src/graphics/mod.rs:77) */
    int32_t  *voxel;        /* 3 per pixel: hit voxel in world coordinates; on a miss the last anchor */
    int32_t  *face;         /* 0:-x 1:+x 2:-y 3:+y 4:-z 5:+z = direction of the surface normal; -1 miss; -2 miss by step cap */
    uint32_t *steps;        /* loop iterations (the shader's panic_counter) */
    float    *t;            /* t_distances[next_xyz] at the hit; -1 on a miss */
    float    *hit_point;    /* 3 per pixel; zeros on a miss */
    float    *rgba;         /* 4 per pixel, the shader's colour before the UNORM store */
    uint8_t  *bgr8;         /* 3 B/pixel, byte order B,G,R: the bgra8 pixel without its alpha byte, which is 255 on every path of
                               the shader (render.comp:9-13) -- the frame as it travels between GPUs, written by the trace
                               kernel itself instead of by vrt_pack_bgr behind it.  A caller built against the struct
                               without this member passes the smaller struct_size and is served as before. */
} vrt_outputs;

/* Measurement knobs, read from the environment once per vrt_create and meant for A/B runs only (none changes a
 * pixel): VRT_TILE_LOG2H=k (wavefront tile of (64 >> k) x (1 << k) pixels instead of 32 x 2), VRT_NO_ROW_ORDER=1
 * (workgroup rows always top to bottom), VRT_ROW_BLOCK0=r (always start at workgroup row r), VRT_SCHED_DEBUG=1
 * (print what each launch measured; synchronises), VRT_SECONDARY=pixel (vrt_trace_secondary without the LDS
 * binning), VRT_NO_SKY=1 (wavefronts whose rays can only meet air are traced like any other), VRT_SKY_REACH=f (how steeply a ray
 * may descend and still be tried for that short cut); vrt_comm_create reads VRT_COMM_SELF_SENDRECV=1 (the root's own part through RCCL as well). */
/* ≙ GraphicsCore::new (src/graphics/mod.rs:292-425), minus window/swapchain:
 * picks the device, creates the stream, allocates the N^3 world images. */
int vrt_create(const vrt_config *cfg, vrt_ctx **out);
int vrt_destroy(vrt_ctx *ctx);

/* Replace the whole world.  `block`: n^3 texels (u16 for REF_U16U16, u8 for PACKED_U8),
 * linear, x fastest; world voxel p lives at texel rem_euclid(p, n) per axis
 * (terrain_gen.comp:23).  `dist`: n^3 u16 texels or NULL (all zero); must be NULL for
 * PACKED_U8.  `memory` says where both pointers live.  The data is copied. */
int vrt_upload_world(vrt_ctx *ctx, const void *block, const void *dist, uint32_t n, uint32_t memory);

/* ≙ GraphicsCore::upload_chunk_gpu(chunk_pos) (src/graphics/mod.rs:245-291): run the
 * terrain generator (terrain_gen.comp:14-24, noise.glsl) for the 32^3 voxels of one chunk
 * and store them at their toroidal texels. */
int vrt_generate_chunk(vrt_ctx *ctx, int32_t cx, int32_t cy, int32_t cz);
/* The same for `count` chunks at chunk positions (xyz[3*i], xyz[3*i+1], xyz[3*i+2]), 64 chunks per launch: what a
 * caller that collects the chunk positions of one GameState::upload_chunks_around_player pass
 * (src/game_logic/game_state.rs:61-75) hands over in one call (a crossing at grid_n = 1056 regenerates 1089
 * chunks: 17 ms of launches one by one, 18 launches this way). */
int vrt_generate_chunks(vrt_ctx *ctx, const int32_t *xyz, uint32_t count);

/* The same generator for every world voxel of the cube [wmin, wmin+n)^3 in one launch
 * (what GameState::upload_chunks_around_player, src/game_logic/game_state.rs:50-83,
 * achieves chunk by chunk for its 5^3 chunks). */
int vrt_generate_window(vrt_ctx *ctx, const int32_t wmin[3]);

/* ≙ GraphicsCore::reset_distance_field (src/graphics/mod.rs:150-183): clears the distance
 * image to zero.  Also the point at which the traversal's own skip structure is rebuilt
 * after world edits (the reference calls it after every batch of chunk uploads,
 * game_state.rs:82).  After vrt_generate_chunk calls on an indexed world the rebuild is incremental: the regenerated
 * chunks' cells and the cells within 31 cells of them, whose empty-space distances can see them; after
 * vrt_upload_world or vrt_generate_window it covers the whole grid.  Launches only, no allocation after the first
 * build, no host wait. */
int vrt_reset_distance_field(vrt_ctx *ctx);

/* ≙ the compute dispatch inside GraphicsCore::redraw (src/graphics/mod.rs:581-597) with
 * PushConstants{transform} (src/graphics/push_constants.rs:4-9): trace rows [y0,y1) of a
 * w x h frame.  `cam_col_major`: the 4x4 f32 transform, column major (nalgebra Matrix4 =
 * GLSL mat4).  With HOST outputs the call returns when the planes are filled; with DEVICE
 * outputs it only enqueues work on the context's stream (vrt_sync to wait). */
int vrt_trace(vrt_ctx *ctx, const float cam_col_major[16], uint32_t w, uint32_t h,
              uint32_t y0, uint32_t y1, const vrt_outputs *out);

/* vrt_trace for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  It has NO counterpart in the reference.  Same arguments, same planes, same kernels as
 * vrt_trace, with two differences.  (1) It makes no hipMalloc / hipFree / hipHostMalloc call and none that
 * allocates behind the scenes: the per-frame tables and the dispatch-order slots that a trace launch needs beyond
 * the world and the caller's planes live in a static device array of the library (1 165 576 bytes per GPU, held
 * until the process ends) instead of the context's own buffers.  (2) Therefore: DEVICE planes only; frames of at
 * most 8192 x 8192 pixels and 131 072 wavefronts of 64 pixels (3840 x 2160 has 129 600); grid_n at most 4096;
 * and the traversal index must be current -- call vrt_reset_distance_field (or vrt_trace once) after the last
 * change to the world, else VRT_E_UNSUPPORTED.  Anything beyond these limits is VRT_E_INVALID_ARG with a message.
 * Every context on one GPU shares that GPU's array, which keeps the last call's tables and measurements (the
 * library refills and re-tags what a call needs): launches that overlap on one GPU -- two streams, two threads --
 * would race on it, so the library keeps them apart: a call that arrives while another thread is inside
 * vrt_trace_static on the same GPU returns VRT_E_UNSUPPORTED (nothing enqueued), and a call on another stream than the
 * last one first makes its stream wait for that launch (one event per GPU, created by vrt_create; calls that stay on
 * one stream wait for nothing).  vrt_create and vrt_trace_static may be called from several threads; a context itself
 * is still driven by one thread at a time.  No timing events are recorded (vrt_kernel_time does not see these
 * launches).  The call only enqueues work on the context's stream. */
int vrt_trace_static(vrt_ctx *ctx, const float cam_col_major[16], uint32_t w, uint32_t h,
                     uint32_t y0, uint32_t y1, const vrt_outputs *out);

/* Secondary rays per primary hit: one shadow ray towards normalize(0.4, 0.8, 0.45) and four
 * cosine-distributed ambient-occlusion rays about the face normal (hash chain
 * pcg(0x5EED ^ (row*w + col))), origin hit_point + normal*1e-3, traced by the skip kernel in
 * the camera's loaded area.  NO counterpart in the reference (its shader casts primary rays
 * only); an incoherent-ray stress workload whose exact statement is the CPU oracle's.
 * Needs an all-zero distance image.  Planes cover rows [y0,y1); any pointer may be NULL. */
typedef struct vrt_secondary_outputs {
    uint32_t struct_size;   /* sizeof(vrt_secondary_outputs) */
    uint32_t memory;        /* VRT_MEM_HOST or VRT_MEM_DEVICE */
    uint8_t  *shadow;       /* 1 if the shadow ray hits; 0 otherwise and for primary misses */
    uint8_t  *ao;           /* number of ambient-occlusion rays that hit, 0..4 */
    uint32_t *sec_steps;    /* loop iterations summed over the five secondary rays */
} vrt_secondary_outputs;
int vrt_trace_secondary(vrt_ctx *ctx, const float cam_col_major[16], uint32_t w, uint32_t h,
                        uint32_t y0, uint32_t y1, const vrt_secondary_outputs *out);

/* Rows of interleaved strips for multi-GPU sharding: strip s (rows [s*strip_h, (s+1)*strip_h))
 * belongs to part s % parts; traces the strips of `part`, packed in strip order, so plane
 * element index = (packed_row)*w + c.  vrt_strip_rows gives the packed row count. */
int vrt_trace_strips(vrt_ctx *ctx, const float cam_col_major[16], uint32_t w, uint32_t h,
                     uint32_t strip_h, uint32_t part, uint32_t parts, const vrt_outputs *out);
uint32_t vrt_strip_rows(uint32_t h, uint32_t strip_h, uint32_t part, uint32_t parts);

/* The step after the gather of a strip-sharded frame: `gathered` holds the packed planes of parts
 * 0..parts-1 one after the other, part_stride_rows rows apart (at least the row count of the largest
 * part; rows beyond a part's own are padding), each row row_bytes long (w * bytes per pixel of the
 * plane); writes the h rows of `frame` in image order.  Both pointers are DEVICE memory and must not
 * overlap; the copy is enqueued on the context's stream.  The reference renders on one device and
 * has no counterpart (src/graphics/mod.rs:581-597). */
int vrt_assemble_strips(vrt_ctx *ctx, const void *gathered, void *frame, uint32_t row_bytes, uint32_t h,
                        uint32_t strip_h, uint32_t parts, uint32_t part_stride_rows);

/* vrt_assemble_strips for parts that travelled without their alpha byte (vrt_pack_bgr): `gathered_bgr` holds BGR8 rows of
 * 3 * w bytes, laid out as above; `frame_bgra` receives the h rows as BGRA8, alpha 255 (render.comp:9-13 stores a = 1) --
 * the reorder and the fourth byte in one pass.  w a multiple of 4; the parts 4-byte, the frame 16-byte aligned; DEVICE
 * memory, not overlapping; enqueued on the context's stream.  No counterpart in the reference (one device). */
int vrt_assemble_strips_bgr(vrt_ctx *ctx, const void *gathered_bgr, void *frame_bgra, uint32_t w, uint32_t h,
                            uint32_t strip_h, uint32_t parts, uint32_t part_stride_rows);

/* ---- several GPUs driven by ONE process: the framebuffer exchange ------------------------------------------
 * The reference picks one physical device (src/graphics/mod.rs:325-342) and has no counterpart.  A host that
 * shards a frame (vrt_trace_strips) or a batch of poses over n GPUs creates one context per HIP device, each
 * holding a replica of the world, and one communicator over them: RCCL, one rank per context in the order given
 * (ncclCommInitAll); RCCL is loaded when the first communicator is created and a machine without it gets
 * VRT_E_UNSUPPORTED.  Context 0 is the root, the one that receives.  Both gathers are one group of ncclSend /
 * ncclRecv enqueued on the contexts' own streams -- behind the traces enqueued there before, no host wait -- and
 * a device copy for the root's own part; vrt_sync(ctxs[0]) waits for the result.  Buffers are DEVICE memory of
 * the context they belong to.  (A host with one process per GPU gathers with its own collective and calls
 * vrt_assemble_strips.) */
typedef struct vrt_comm vrt_comm;
/* n = 1..64 contexts on distinct devices.  With VRT_COMM_ALLOW_SHARED_DEVICE=1 in the environment (a rehearsal switch:
 * the n-context control flow -- offsets, strides, joins, stream order -- on a machine with fewer GPUs) contexts may share
 * the root's device; their parts are copied on that device between two events instead of travelling through RCCL.
 * More than one context per process has only been run that way: no machine with two GPUs was available to the build. */
int vrt_comm_create(vrt_ctx *const *ctxs, uint32_t n, vrt_comm **out);
int vrt_comm_destroy(vrt_comm *comm);                                   /* waits for the contexts' streams */
uint32_t vrt_comm_size(const vrt_comm *comm);
/* bytes_each bytes at send[i] (context i's device) -> recv_root + i * bytes_each (root's device): whole frames of
 * a pose-sharded batch. */
int vrt_gather(vrt_comm *comm, const void *const *send, size_t bytes_each, void *recv_root);
/* A strip-sharded frame: packed[i] holds the vrt_strip_rows(h, strip_h, i, n) rows context i traced as part i of
 * n (vrt_trace_strips), row_bytes each.  They are gathered into `gathered` (n * part_stride_rows rows, root's
 * device; part_stride_rows at least the largest part's row count) and put back in image order into `frame` (h
 * rows, root's device) as vrt_assemble_strips does. */
int vrt_gather_strips(vrt_comm *comm, const void *const *packed, void *gathered, void *frame, uint32_t row_bytes,
                      uint32_t h, uint32_t strip_h, uint32_t part_stride_rows);
/* The same for a BGRA8 frame with three bytes per pixel on the links (a quarter less through the root's links, which
 * bound the step: 24.9 instead of 33.2 MB for a 3840 x 2160 frame): context i packs its rows packed_bgra[i] into its own
 * buffer packed_bgr[i] (3 * w bytes per row, vrt_pack_bgr on its stream), the BGR8 parts are gathered into
 * `gathered_bgr` (n * part_stride_rows rows of 3 * w bytes, root's device) and vrt_assemble_strips_bgr writes
 * `frame_bgra`.  w a multiple of 4. */
int vrt_gather_strips_bgr(vrt_comm *comm, const void *const *packed_bgra, void *const *packed_bgr, void *gathered_bgr,
                          void *frame_bgra, uint32_t w, uint32_t h, uint32_t strip_h, uint32_t part_stride_rows);

/* The frame as it travels between GPUs: B, G, R of every pixel without the alpha byte, which is 255 on every path
 * of the shader (render.comp:9-13), so three quarters of the bytes cross the link that bounds a gathered frame.
 * vrt_pack_bgr before the send (`pixels` BGRA8 pixels -> 3 * pixels bytes), vrt_unpack_bgra on the receiver (alpha
 * restored).  `pixels` a multiple of 4; DEVICE memory of the context's GPU, BGRA8 16-byte and BGR8 4-byte aligned,
 * not overlapping; enqueued on the context's stream.  The reference presents on the device that rendered
 * (src/graphics/mod.rs:598-611) and has no counterpart. */
int vrt_pack_bgr(vrt_ctx *ctx, const void *bgra, void *bgr, size_t pixels);
int vrt_unpack_bgra(vrt_ctx *ctx, const void *bgr, void *bgra, size_t pixels);

/* Page-locked host memory for HOST output planes (and for worlds handed to vrt_upload_world): copies
 * between it and the device run at the bus rate, about three times the rate of pageable memory
 * (1080p BGRA8 frame, kernel + copy: 0.23 ms against 0.54 ms).  Any host memory is accepted by the
 * other entry points; this pair only spares the host a HIP binding of its own.  The reference presents
 * from device memory and has no host copy of a frame (src/graphics/mod.rs:531-611). */
int vrt_alloc_host(vrt_ctx *ctx, size_t bytes, void **out);
int vrt_free_host(vrt_ctx *ctx, void *ptr);

/* Wait for everything the context has enqueued (on all of its streams). */
int vrt_sync(vrt_ctx *ctx);

/* Frame overlap, for hosts that trace frames back to back into DEVICE planes (a flythrough, a batch of poses).  One
 * launch of a frame ends with wavefronts that finish one after the other while the GPU runs empty, and a stream starts
 * the next launch only when the last of them is done: with `streams` = 2..4 vrt_trace, vrt_trace_strips and
 * vrt_trace_secondary deal their launches round to that many streams of the context, each with its own scratch, so
 * that the end of one frame runs under the beginning of the next (1080p over 1024^3: 53.4 -> 48.8 us per frame with
 * two streams).  1 (the default) is one stream, every launch behind the one before.  With overlap on, frames may
 * finish in any order: the planes of a frame are complete after vrt_sync, or for work enqueued on vrt_stream(ctx)
 * after vrt_join(ctx), which makes that stream wait for every trace enqueued so far.  The library's own consumers of
 * frames (vrt_assemble_strips, vrt_pack_bgr, vrt_unpack_bgra, vrt_gather*) join by themselves, and so does every
 * change of the world or of its index.  Calls with HOST planes and vrt_trace_static stay on the first stream.  The
 * reference submits one frame at a time and waits for its fence (src/graphics/mod.rs:598-634); no counterpart.
 * The call waits for the context's streams; the streams and their scratch are created here, not by a trace. */
int vrt_set_frame_overlap(vrt_ctx *ctx, uint32_t streams);
int vrt_join(vrt_ctx *ctx);
/* The other direction: every stream of the context waits for what has been enqueued on vrt_stream(ctx) so far.  A host
 * whose own work on that stream still reads a plane calls it before it hands the plane to the next trace, which may run
 * on another stream. */
int vrt_fork(vrt_ctx *ctx);

/* Copy the block image back (tests of the generator).  `block_out`: n^3 texels. */
int vrt_download_world(vrt_ctx *ctx, void *block_out, uint32_t memory);

/* Copy the traversal index out (tests of its builders; the reference has no such structure): `cells_out`, (n/4)^3
 * records of 16 bytes -- the 64-bit occupancy mask of a 4^3 cell, bit z*16 + y*4 + x, then its eight octant
 * distances -- and `pyr_out`, (n/4)^3 x 64 bytes: per cell eight 64-bit words of column-top entries (word = ray
 * descends | travels towards -x << 1 | towards -z << 2; four 16-bit entries per word, one per voxel level of the cell)
 * as levelled for the window of the last trace (VRT_E_UNSUPPORTED before the first trace after a world change);
 * either may be NULL.  The index is
 * brought up to date with the voxel image first, as vrt_reset_distance_field does. */
int vrt_download_index(vrt_ctx *ctx, void *cells_out, void *pyr_out, uint32_t memory);

/* HIP-event time of the trace kernel: sum and count over all vrt_trace calls since the
 * last vrt_kernel_time_reset.  Waits for the stream. */
int vrt_kernel_time(vrt_ctx *ctx, double *total_ms, uint64_t *launches);
int vrt_kernel_time_reset(vrt_ctx *ctx);
/* Time one trace launch in `every_n` (default 1: all of them).  The two events around a launch cost
 * a few microseconds of stream time, which a caller that issues frames back to back may not want on
 * every frame; `launches` of vrt_kernel_time then counts the timed launches only.  every_n = 0 turns
 * the timing off. */
int vrt_kernel_time_sampling(vrt_ctx *ctx, uint32_t every_n);
/* Workgroups (256 threads, four wavefronts) of a trace kernel that one compute unit of the context's GPU holds at once,
 * by the runtime's own occupancy calculation: kind 0 the primary skip kernel, 1 the LDS-binned secondary-ray kernel,
 * 2 the literal kernel.  The kernels are built to reach 8 (every wavefront slot of the CU); a build that loses that --
 * a register or 20 KB of LDS more -- slows down without failing any test of its pixels, so a test asks here. */
int vrt_kernel_occupancy(vrt_ctx *ctx, uint32_t kind, int32_t *workgroups_per_cu);
/* Name of the kernel the last vrt_trace launched ("literal" or "skip"). */
const char *vrt_last_kernel_name(vrt_ctx *ctx);

/* The context's hipStream_t, for callers that enqueue their own work behind a trace, and its HIP device ordinal
 * (-1 for a null context). */
void *vrt_stream(vrt_ctx *ctx);
int vrt_device(vrt_ctx *ctx);

const char *vrt_last_error(void);
const char *vrt_version(void);

#ifdef __cplusplus
}
#endif
#endif
