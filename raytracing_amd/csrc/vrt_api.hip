// vrt_api.hip -- the extern "C" surface of libvrt.so (include/vrt.h): context, world
// residency in HBM, kernel selection, HIP-event timing.  No CPU fallback exists: every
// entry point either runs on the HIP device or returns an error.
#include "../../include/vrt.h"
#include "vrt_kernels.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <new>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(e_ == hipErrorOutOfMemory ? VRT_E_OUT_OF_MEMORY : VRT_E_HIP,        \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                     \
    } while (0)

constexpr int kEventRing = 64;
constexpr int kMaxLanes = 4;  // streams a context deals its frames to (vrt_set_frame_overlap)

constexpr uint32_t kSchedRows = 256;  // = SCHED_MAX_ROWS of the kernels
constexpr size_t   kSchedHead = 2 + 2 * kSchedRows;

// Device memory a trace launch needs beyond the world and the caller's planes, and what the host knows
// about its contents:
//   tables  ncx[max_dim] | ncy[max_dim + 1] | sin2[3][grid_n + 8]        (TraceParams::ncx, ncy, sin2)
//   sched   start[2] | row_longest[2][256] | cost[2][sched_capacity]      (vrt::RowSched)
// A context owns one (allocated in vrt_create); vrt_trace_static uses the one that lives in the library's
// static device array, one per device.
struct Scratch {
    float    *tables        = nullptr;
    uint32_t  max_dim       = 0;         // frame width / height the tables have room for
    uint32_t  sin_cap       = 0;         // entries per axis the sine table has room for
    uint32_t *sched         = nullptr;
    uint32_t  sched_capacity = 0;        // wavefronts per launch the dispatch-order feedback covers
    // contents
    uint32_t  tab_w = 0, tab_h = 0, tab_sin_len = 0;  // frame / grid the tables were filled for (0: none)
    int32_t   tab_base[3]   = {0, 0, 0};
    uint32_t  sched_gen     = 0;             // generation of the last skip launch, 1..4095
    uint32_t  sched_same    = 0;             // consecutive launches of this shape before the next one
    uint64_t  sched_sig[5]  = {0, 0, 0, 0, 0};  // shape of the last launch; [0] == 0: none

    size_t table_floats() const { return (size_t)2 * max_dim + 1 + 3 * (size_t)sin_cap; }
    size_t sched_bytes() const { return (kSchedHead + 2 * (size_t)sched_capacity) * sizeof(uint32_t); }
};

// ---- the static scratch of vrt_trace_static ---------------------------------------------------------
// Limits of that entry point: frames of at most 8192 x 8192 pixels and 131072 wavefronts of 64 pixels
// (3840 x 2160 has 129600), grids of at most 4096 voxels per axis.
constexpr uint32_t kStaticMaxDim      = 8192;
constexpr uint32_t kStaticMaxGrid     = 4096;
constexpr uint32_t kStaticSchedCap    = 131072;
constexpr size_t   kStaticTableBytes  = ((size_t)2 * kStaticMaxDim + 1 + 3 * (size_t)(kStaticMaxGrid + 8)) * sizeof(float);
constexpr size_t   kStaticSchedOffset = (kStaticTableBytes + 255) & ~(size_t)255;
constexpr size_t   kStaticSchedBytes  = (kSchedHead + 2 * (size_t)kStaticSchedCap) * sizeof(uint32_t);
constexpr size_t   kStaticScratchBytes = kStaticSchedOffset + kStaticSchedBytes;  // 1 165 576 bytes

// One copy per GPU the library runs on, held until the process ends.  In host code this name is only a
// stand-in for the symbol: the device address comes from hipGetSymbolAddress (static_scratch_for) and is
// what every kernel argument, memset and copy below uses.
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

constexpr int kMaxDevices = 64;
// What guards a GPU's one static scratch.  Host side: `in_use` is held for the length of a vrt_trace_static call, which
// reads and updates the host's notes on the scratch (tables filled for, dispatch-order generation): a second thread
// that calls meanwhile gets VRT_E_UNSUPPORTED instead of a torn update.  Device side: launches that use the scratch
// from two streams must not run side by side; `done` is recorded behind every launch and a call that arrives on another
// stream than the last one makes its stream wait for it (calls that stay on one stream touch no event wait).
struct StaticScratch {
    std::atomic<bool> ready{false};
    std::atomic<bool> in_use{false};
    Scratch     s;
    hipEvent_t  done = nullptr;          // behind the last launch that used the scratch
    hipStream_t last_stream = nullptr;   // the stream that launch went to (null: none yet)
};
StaticScratch g_static[kMaxDevices];
std::atomic<bool> g_static_setup{false};  // one thread at a time looks a GPU's scratch up for the first time

// The device's own copy of `scratch`, looked up once per device with that device current.
int static_scratch_for(int device, StaticScratch **out)
{
    if (device < 0 || device >= kMaxDevices) return fail(VRT_E_INVALID_ARG, "device %d beyond the %d the static scratch covers", device, kMaxDevices);
    StaticScratch &st = g_static[device];
    if (!st.ready.load(std::memory_order_acquire)) {
        while (g_static_setup.exchange(true, std::memory_order_acquire)) {}  // (set-up of a few microseconds, once per GPU)
        struct Release { ~Release() { g_static_setup.store(false, std::memory_order_release); } } release;
        if (!st.ready.load(std::memory_order_acquire)) {
            HIP_TRY(hipSetDevice(device));
            void *ptr = nullptr;
            HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
            if (!ptr) return fail(VRT_E_HIP, "hipGetSymbolAddress returned no address for the static scratch");
            if (!st.done) HIP_TRY(hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
            st.s = Scratch{};
            st.s.tables = static_cast<float *>(ptr);
            st.s.max_dim = kStaticMaxDim;
            st.s.sin_cap = kStaticMaxGrid + 8;
            st.s.sched = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(ptr) + kStaticSchedOffset);
            st.s.sched_capacity = kStaticSchedCap;
            st.ready.store(true, std::memory_order_release);  // the array is zero when the code object is loaded: generation tags start clean
        }
    }
    *out = &st;
    return VRT_OK;
}

}  // namespace

int vrt::set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

struct vrt_ctx {
    int         device       = 0;
    hipStream_t stream       = nullptr;
    uint32_t    n            = 0;
    uint32_t    format       = 0;
    int         voxel_bytes  = 1;
    uint32_t    max_steps    = 0;
    uint32_t    trace_mode   = VRT_TRACE_AUTO;
    // world in HBM
    void       *d_block      = nullptr;  // n^3 texels, linear x fastest
    uint16_t   *d_dist       = nullptr;  // n^3 u16, allocated on first non-null upload
    bool        dist_nonzero = false;
    bool        have_world   = false;
    // skip structure
    uint4      *d_cells      = nullptr;  // (n/4)^3 x 128 B: per cell the record (occupancy mask + octant distances), then its column-top words
    uint16_t   *d_tops       = nullptr;  // column-top tables (build scratch), TOPS_LEVELS * (n/4)^2, and behind them d_sky
    uint32_t   *d_sky        = nullptr;  // window level just above the highest solid voxel of the window (TraceParams::sky)
    uint4      *d_pyr        = nullptr;  // d_cells + 1: column-top words at voxel precision, cell i's from d_pyr[CELL_U4 * i]
    uint8_t    *d_build      = nullptr;  // 14 x (n/4)^3 B: the cells' masks side by side and the x and y planes of the octant distance transform, kept between builds
    // what the skip structure has not seen yet: nothing (skip_valid), a box of regenerated chunks, or everything
    bool        index_built  = false;    // records and planes describe some earlier state of the whole voxel image
    bool        dirty_box    = false;    // chunks inside [dirty_lo, dirty_hi) (world cells of 4^3 voxels) changed since
    int32_t     dirty_lo[3]  = {0, 0, 0}, dirty_hi[3] = {0, 0, 0};
    Scratch     own;                     // per-frame tables + dispatch-order slots of this context (lane 0)
    // Frame overlap (vrt_set_frame_overlap): traces with DEVICE planes are dealt round to `overlap` lanes, each a stream
    // with a scratch of its own, so that nothing a launch touches beyond the world and the caller's planes is shared
    // between launches that run side by side.  Lane 0 is `stream` / `own`.
    uint32_t    overlap      = 1;
    uint32_t    next_lane    = 0;
    hipStream_t lane_stream[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
    Scratch     lane_scratch[kMaxLanes];            // [0] unused: lane 0 has `own`
    hipEvent_t  lane_ev[kMaxLanes]     = {nullptr, nullptr, nullptr, nullptr};
    bool        lane_busy[kMaxLanes]   = {false, false, false, false};  // launches on the lane since it was last joined
    uint64_t    serial       = 0;        // distinguishes contexts in the dispatch-order history of a shared scratch
    int         pyr_offcy    = -1;       // vertical window offset (cells) the pyramid entries were built for
    bool        skip_valid   = false;
    uint32_t   *d_flag       = nullptr;
    // measurement knobs, read from the environment once, when the context is created
    //   VRT_TILE_LOG2H=k    wave tile of (64 >> k) x (1 << k) pixels instead of 32 x 2
    //   VRT_NO_ROW_ORDER=1  workgroup rows always top to bottom
    //   VRT_ROW_BLOCK0=r    workgroup rows always start at row block r
    //   VRT_SCHED_DEBUG=1   print what each launch measured for the dispatch order (synchronises)
    //   VRT_SECONDARY=pixel vrt_trace_secondary runs every pixel's rays on the pixel's own lane instead of staging
    //                       and binning the workgroup's rays in LDS (same outputs; for A/B measurements)
    uint32_t    knob_tile_log2h = 1;
    uint32_t    knob_tile_lanes = 0;     // VRT_TILE_LANES=32|64: lanes of a wavefront that trace (0: the library chooses per launch)
    bool        knob_no_row_order = false, knob_sched_debug = false, knob_secondary_binned = true, knob_no_sky = false;
    int32_t     knob_row_block0 = -1;
    float       knob_sky_reach = 0.6f;   // VRT_SKY_REACH=f: a descending ray is tried for the sky short cut when it would lose less than the
                                         // camera's height above the highest solid voxel over f * n voxels of path
    // staging for host-resident outputs
    void       *d_stage      = nullptr;
    size_t      stage_bytes  = 0;
    // kernel timing
    hipEvent_t  ev0[kEventRing], ev1[kEventRing];
    int         ev_pending   = 0;
    uint32_t    time_every   = 1;        // one launch in time_every is bracketed by events; 0 = none
    uint64_t    launch_seq   = 0;
    double      total_ms     = 0.0;
    uint64_t    launches     = 0;
    const char *last_kernel  = "none";
};

namespace {

size_t texels(const vrt_ctx *c) { return (size_t)c->n * c->n * c->n; }
size_t cells(const vrt_ctx *c) { return texels(c) / 64; }

int drain_events(vrt_ctx *c)
{
    for (int i = 0; i < c->ev_pending; ++i) {
        HIP_TRY(hipEventSynchronize(c->ev1[i]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0[i], c->ev1[i]));
        c->total_ms += ms;
        c->launches += 1;
    }
    c->ev_pending = 0;
    return VRT_OK;
}

// The context's first stream waits for everything the other lanes have been given so far: whatever is enqueued on
// it from here on (a pack or reorder kernel, a copy, the caller's own work, an index update) finds the frames done.
int join_lanes(vrt_ctx *c)
{
    for (uint32_t i = 1; i < c->overlap; ++i) {
        if (!c->lane_busy[i]) continue;
        HIP_TRY(hipEventRecord(c->lane_ev[i], c->lane_stream[i]));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->lane_ev[i], 0));
        c->lane_busy[i] = false;
    }
    return VRT_OK;
}

// The other lanes wait for what the first stream has been given so far (a change of the world or of its index).
int fork_lanes(vrt_ctx *c)
{
    if (c->overlap <= 1) return VRT_OK;
    HIP_TRY(hipEventRecord(c->lane_ev[0], c->stream));
    for (uint32_t i = 1; i < c->overlap; ++i) HIP_TRY(hipStreamWaitEvent(c->lane_stream[i], c->lane_ev[0], 0));
    return VRT_OK;
}

// The lane of the next trace: lane 0 unless frames overlap and the planes stay on the device.
uint32_t pick_lane(vrt_ctx *c, bool device_planes)
{
    if (c->overlap <= 1 || !device_planes) return 0;
    const uint32_t lane = c->next_lane;
    c->next_lane = (lane + 1) % c->overlap;
    return lane;  // (the caller marks the lane busy once its launch is enqueued: a join before that has nothing to wait for)
}

int sync_lanes(vrt_ctx *c)
{
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (uint32_t i = 1; i < c->overlap; ++i) {
        HIP_TRY(hipStreamSynchronize(c->lane_stream[i]));
        c->lane_busy[i] = false;
    }
    return VRT_OK;
}

hipStream_t stream_of(vrt_ctx *c, uint32_t lane) { return lane ? c->lane_stream[lane] : c->stream; }
Scratch &scratch_of(vrt_ctx *c, uint32_t lane) { return lane ? c->lane_scratch[lane] : c->own; }

// Brings the skip structure up to date with the voxel image.  offcy: rem_euclid(lower_y, n) / 4 of the window
// the next trace uses, or -1 when no camera is known yet (the column-top pyramid, whose entries count levels
// from the bottom of the window, is then left to the next trace).  Buffers are allocated on the first call
// and kept; everything else is launches on the context's stream.
int ensure_skip_structure(vrt_ctx *c, int offcy)
{
    const bool work = !c->skip_valid || (offcy >= 0 && c->pyr_offcy != offcy);
    if (work) { int rc = join_lanes(c); if (rc != VRT_OK) return rc; }  // no trace reads the index while it is rewritten
    if (!c->skip_valid) {
        if (!c->d_cells) { HIP_TRY(hipMalloc(&c->d_cells, cells(c) * vrt::CELL_U4 * sizeof(uint4))); c->d_pyr = c->d_cells + 1; }
        if (!c->d_tops) {
            const size_t tables = (size_t)vrt::TOPS_LEVELS * (c->n / 4) * (c->n / 4) * sizeof(uint16_t);  // (a multiple of 4: n / 4 is even)
            HIP_TRY(hipMalloc(&c->d_tops, tables + 64));
            c->d_sky = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(c->d_tops) + tables);
        }
        if (!c->d_build) HIP_TRY(hipMalloc(&c->d_build, vrt::PLANE_BYTES_PER_CELL * cells(c)));
        const int nc = (int)c->n / 4;
        if (c->index_built && c->dirty_box) {
            // only regenerated chunks since the last build: their cells, and the cells whose distances see them
            int origin[3], extent[3];
            for (int k = 0; k < 3; ++k) {
                const int64_t len = (int64_t)c->dirty_hi[k] - c->dirty_lo[k];
                if (len >= nc) { origin[k] = 0; extent[k] = nc; continue; }
                int o = c->dirty_lo[k] % nc;
                origin[k] = o < 0 ? o + nc : o;
                extent[k] = (int)len;
            }
            HIP_TRY(vrt::launch_update_cells(c->d_block, c->voxel_bytes, (int)c->n, c->d_cells, c->d_build, origin, extent, c->stream));
        } else {
            HIP_TRY(vrt::launch_build_cells(c->d_block, c->voxel_bytes, (int)c->n, c->d_cells, c->d_build, c->stream));
        }
        c->skip_valid = true;
        c->index_built = true;
        c->dirty_box = false;
        c->pyr_offcy = -1;
    }
    if (offcy >= 0 && c->pyr_offcy != offcy) {
        HIP_TRY(vrt::launch_build_voxel_pyramid(c->d_build, (int)c->n / 4, offcy, c->d_tops, c->d_cells, c->d_sky, c->stream));
        c->pyr_offcy = offcy;
    }
    if (work) return fork_lanes(c);
    return VRT_OK;
}

// the voxel image changed everywhere (or nobody knows where): the next build is a whole one
void world_replaced(vrt_ctx *c)
{
    c->have_world = true;
    c->skip_valid = false;
    c->index_built = false;
    c->dirty_box = false;
}

// the voxels of world box [lo, lo + len) changed (voxel coordinates, multiples of 4)
void world_box_changed(vrt_ctx *c, const int32_t lo[3], int32_t len)
{
    c->have_world = true;
    c->skip_valid = false;
    for (int k = 0; k < 3; ++k) {
        const int32_t a = lo[k] >> 2, b = (lo[k] + len) >> 2;  // arithmetic shift: floor for negative coordinates
        if (!c->dirty_box) { c->dirty_lo[k] = a; c->dirty_hi[k] = b; }
        else { c->dirty_lo[k] = std::min(c->dirty_lo[k], a); c->dirty_hi[k] = std::max(c->dirty_hi[k], b); }
    }
    c->dirty_box = true;
}

// vertical offset (in cells) of the loaded window for a camera at height y (raytrace.glsl:19-24)
int window_offcy(const vrt_ctx *c, float y)
{
    const int n = (int)c->n;
    const int lower = (int)std::floor(y / (float)VRT_CHUNK_SIZE) * VRT_CHUNK_SIZE + VRT_CHUNK_SIZE / 2 - n / 2;
    int r = lower % n;
    if (r < 0) r += n;
    return r / 4;
}

struct Plane { size_t off, bytes; void *host; };

// Packed row (of this launch) on which the central ray is horizontal: the rows with the longest rays
// lie just under it, so a launch is only comparable with the previous one if it has not moved much.
uint32_t horizon_packed_row(const float *m, uint32_t w, uint32_t h, const vrt::RowMap &rows)
{
    if (rows.rows == 0) return 0;
    // raytrace.glsl:120-123 at the centre column: rd.y ~ ncy*m[5] + m[6] + m[7], ncy = gy/h - h/(2w), row = h - gy
    float row = (float)h * 0.5f;
    if (m[5] != 0.0f) {
        const float ncy = -(m[6] + m[7]) / m[5];
        row = (float)h - (ncy + (float)h / (float)w * 0.5f) * (float)h;
    }
    if (!(row > 0.0f)) row = 0.0f;                  // also catches NaN
    if (row > (float)(h - 1)) row = (float)(h - 1);
    const uint32_t r = (uint32_t)row;
    uint32_t q;
    if (rows.strip_h == 0) {
        q = r <= rows.y0 ? 0u : r - rows.y0;
    } else {
        q = r / (rows.strip_h * rows.parts) * rows.strip_h;
    }
    return q < rows.rows ? q : rows.rows - 1;
}

constexpr uint32_t kSchedCapacity = 262144;  // wavefronts per launch a context's own feedback covers (1080p: 32640, 4K: 129600)

// Fills p.sched for a skip-kernel launch (kind 1: primary, 2: primary + secondary) from the slots in S.  The
// start row planned by the previous launch from the times of the launch before it is used when both had this
// launch's context, kernel, frame, rows and (to a workgroup row) horizon; either way this launch measures.
int attach_row_sched(vrt_ctx *c, Scratch &S, vrt::TraceParams &p, uint32_t kind, bool overlapped = false)
{
    p.sched = vrt::RowSched{};
    p.sched.fixed_start = -1;
    // Launches that overlap run their rows top to bottom and measure nothing: the end of a frame runs under the
    // beginning of the next, so there is no lone tail to shorten, and the long rows of the next frame are better not
    // started against it (1080p over 1024^3, two streams: 47.9 us per frame against 49.4 with the measured order).
    if (overlapped && c->knob_row_block0 < 0) { S.sched_sig[0] = 0; return VRT_OK; }
    if (c->knob_row_block0 >= 0) { p.sched.fixed_start = c->knob_row_block0; S.sched_sig[0] = 0; return VRT_OK; }
    const uint32_t block_h = 4u << p.tile_log2h;
    const uint32_t grid_y = (p.rows.rows + block_h - 1) / block_h;
    const uint32_t grid_x = (p.w + (p.tile_lanes >> p.tile_log2h) - 1) / (p.tile_lanes >> p.tile_log2h);
    if (!S.sched || grid_y == 0 || grid_y > kSchedRows || (uint64_t)grid_x * grid_y * 4 > S.sched_capacity || c->knob_no_row_order) {
        S.sched_sig[0] = 0;
        return VRT_OK;
    }
    const uint64_t sig[5] = {
        ((uint64_t)kind << 56) | ((uint64_t)(p.tile_lanes == 32u) << 55) | ((uint64_t)p.tile_log2h << 48) | ((uint64_t)p.w << 24) | (uint64_t)p.h,
        ((uint64_t)p.rows.y0 << 32) | (uint64_t)p.rows.rows,
        ((uint64_t)p.rows.strip_h << 40) | ((uint64_t)p.rows.part << 20) | (uint64_t)p.rows.parts,
        (uint64_t)(horizon_packed_row(p.m, p.w, p.h, p.rows) / block_h),
        c->serial};
    if (c->knob_sched_debug && S.sched_gen > 0 && &S == &c->own) {  // diagnostic: what the last launch measured
        std::vector<uint32_t> hbuf(kSchedHead + 2 * (size_t)S.sched_capacity);
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(hbuf.data(), S.sched, S.sched_bytes(), hipMemcpyDeviceToHost));
        const uint32_t g = S.sched_gen;
        std::fprintf(stderr, "[sched] gen %u start[0]=%u/%u start[1]=%u/%u rows:", g, hbuf[0] >> 16, hbuf[0] & 0xffff,
                     hbuf[1] >> 16, hbuf[1] & 0xffff);
        for (uint32_t r = 0; r < grid_y; ++r) {
            uint32_t m = 0;
            for (uint32_t x = 0; x < grid_x * 4; ++x) {
                const uint32_t v = hbuf[kSchedHead + (size_t)(g & 1) * S.sched_capacity + r * grid_x * 4 + x];
                if ((v >> 20) == g) m = std::max(m, v & 0xfffffu);
            }
            std::fprintf(stderr, " %u", m);
        }
        std::fprintf(stderr, "\n");
    }
    S.sched_same = std::memcmp(sig, S.sched_sig, sizeof sig) == 0 ? S.sched_same + 1 : 0;
    if (S.sched_gen >= 4095u) {  // the tags are 12 bits wide: start over with clean slots
        HIP_TRY(hipMemsetAsync(S.sched, 0, S.sched_bytes(), c->stream));
        S.sched_gen = 0;
        S.sched_same = 0;
    }
    S.sched_gen += 1;
    p.sched.start = S.sched;
    p.sched.row_longest = S.sched + 2;
    p.sched.cost = S.sched + kSchedHead;
    p.sched.capacity = S.sched_capacity;
    p.sched.generation = S.sched_gen;
    p.sched.use_order = S.sched_same >= 3 ? 1u : 0u;
    std::memcpy(S.sched_sig, sig, sizeof sig);
    return VRT_OK;
}

// argument checks, kernel choice and the parameter block shared by every trace entry point
// (S: the scratch whose tables the launch reads, st: the stream the launch goes to; may_build: the skip structure may be (re)built here, which
// allocates on its first use)
int prepare_trace(vrt_ctx *c, Scratch &S, hipStream_t st, bool may_build, const float *cam, uint32_t w, uint32_t h,
                  const vrt::RowMap &rows, bool need_skip, vrt::TraceParams &p, bool &use_skip)
{
    if (!c || !cam) return fail(VRT_E_INVALID_ARG, "null argument");
    if (w == 0 || h == 0 || w > 16384 || h > 16384) return fail(VRT_E_INVALID_ARG, "frame size out of range");
    for (int i = 0; i < 16; ++i)
        if (!std::isfinite(cam[i])) return fail(VRT_E_INVALID_ARG, "camera transform is not finite");
    for (int i = 12; i < 15; ++i)
        if (std::fabs(cam[i]) > 1048576.0f) return fail(VRT_E_INVALID_ARG, "camera position beyond +/-2^20");
    if (!c->have_world) return fail(VRT_E_NO_WORLD, "no world uploaded or generated");
    HIP_TRY(hipSetDevice(c->device));

    switch (c->trace_mode) {
        case VRT_TRACE_LITERAL: use_skip = false; break;
        case VRT_TRACE_SKIP:
            if (c->dist_nonzero) return fail(VRT_E_UNSUPPORTED, "skip traversal needs an all-zero distance image");
            use_skip = true;
            break;
        default: use_skip = !c->dist_nonzero; break;
    }
    if (need_skip) {
        if (c->dist_nonzero) return fail(VRT_E_UNSUPPORTED, "secondary rays need an all-zero distance image");
        use_skip = true;
    }
    if (use_skip) {
        if (!may_build && (!c->skip_valid || !c->d_pyr))
            return fail(VRT_E_UNSUPPORTED, "the skip structure is not built: call vrt_reset_distance_field (or vrt_trace once) "
                                           "after the last change to the world");
        int rc = ensure_skip_structure(c, window_offcy(c, cam[13]));  // at most the pyramid's relevelling launches from here
        if (rc != VRT_OK) return rc;
    }

    std::memset(&p, 0, sizeof p);
    std::memcpy(p.m, cam, sizeof p.m);
    p.w = w; p.h = h; p.rows = rows;
    p.ncy_off = (float)h / (float)w * 0.5f;
    p.n = (int)c->n;
    p.n_pow2_mask = (c->n & (c->n - 1)) == 0 ? (int)c->n - 1 : -1;
    for (uint32_t v = c->n / 4; v > 1; v >>= 1) p.nc_log2 += 1;
    p.max_steps = c->max_steps;
    p.tile_log2h = c->knob_tile_log2h;
    p.tile_lanes = 64;   // (do_trace halves it for a primary launch that would leave most of the GPU's wavefront slots empty)
    p.need_cap = c->max_steps < 3u * c->n ? 1u : 0u;
    for (int k = 0; k < 3; ++k) {
        const int n = (int)c->n;
        const int mid = (int)std::floor(cam[12 + k] / (float)VRT_CHUNK_SIZE) * VRT_CHUNK_SIZE + VRT_CHUNK_SIZE / 2;
        p.lo[k] = mid - n / 2;
        p.up[k] = mid + n / 2 - 1;
        int r = p.lo[k] % n;
        p.off[k] = r < 0 ? r + n : r;
        p.a0[k] = (int)std::floor(cam[12 + k]);  // |position| <= 2^20 was checked above
        p.thr0[k] = 0.5f - VRT_MARGIN * (std::fabs(cam[12 + k]) + 1.0f);  // (this file is built without contraction, like the kernels)
        // a ray leaves the loaded area into voxel lo + n - 2 (towards +) or lo (towards -): through planes lo + n - 2, lo + 1
        p.sky_pos[k] = (float)(p.lo[k] + n - 2) - cam[12 + k];
        p.sky_neg[k] = (float)(p.lo[k] + 1) - cam[12 + k];
    }
    p.sky = use_skip && !c->knob_no_sky ? c->d_sky : nullptr;
    p.sky_slope = 1.0f / (c->knob_sky_reach * (float)c->n);
    p.block = c->d_block;
    p.dist = c->dist_nonzero ? c->d_dist : nullptr;
    p.cells = c->d_cells; p.nc = (int)c->n / 4; p.pyr = c->d_pyr;
    {   // per-frame tables: refilled (one small launch) when the frame shape or the window position changes
        const uint32_t sin_len = c->n + 8;
        if (w > S.max_dim || h > S.max_dim || sin_len > S.sin_cap || !S.tables)
            return fail(VRT_E_INVALID_ARG, "frame %u x %u or grid %u beyond what the scratch tables hold (%u, %u)", w, h, c->n,
                        S.max_dim, S.sin_cap > 8 ? S.sin_cap - 8 : 0);
        float *ncx = S.tables, *ncy = ncx + S.max_dim, *sin2 = ncy + S.max_dim + 1;
        const int32_t base[3] = {p.lo[0] - 4, p.lo[1] - 4, p.lo[2] - 4};
        if (S.tab_w != w || S.tab_h != h || S.tab_sin_len != sin_len || std::memcmp(base, S.tab_base, sizeof base) != 0) {
            HIP_TRY(vrt::launch_frame_tables(ncx, ncy, sin2, w, h, p.ncy_off, base, sin_len, st));
            S.tab_w = w; S.tab_h = h; S.tab_sin_len = sin_len;
            std::memcpy(S.tab_base, base, sizeof base);
        }
        p.ncx = ncx; p.ncy = ncy; p.sin2 = sin2; p.sin_len = sin_len;
        std::memcpy(p.sin_base, base, sizeof base);
    }
    return VRT_OK;
}

int ensure_stage(vrt_ctx *c, size_t total)
{
    if (total > c->stage_bytes) {
        if (c->d_stage) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->d_stage)); c->d_stage = nullptr; c->stage_bytes = 0; }
        HIP_TRY(hipMalloc(&c->d_stage, total));
        c->stage_bytes = total;
    }
    return VRT_OK;
}

// vrt_outputs as a caller of either layout passed it: the struct grew by `bgr8` at its end, and a caller built against
// the older header says so with its struct_size.
constexpr uint32_t kOutputsSizeV1 = (uint32_t)offsetof(vrt_outputs, bgr8);
int read_outputs(const vrt_outputs *in, vrt_outputs &o)
{
    if (in->struct_size != sizeof(vrt_outputs) && in->struct_size != kOutputsSizeV1)
        return fail(VRT_E_INVALID_ARG, "vrt_outputs.struct_size mismatch");
    o = vrt_outputs{};
    std::memcpy(&o, in, in->struct_size);
    return VRT_OK;
}

int do_trace(vrt_ctx *c, const float *cam, uint32_t w, uint32_t h, const vrt::RowMap &rows, const vrt_outputs *given)
{
    if (!c || !cam || !given) return fail(VRT_E_INVALID_ARG, "null argument");
    vrt_outputs view;
    { int rc = read_outputs(given, view); if (rc != VRT_OK) return rc; }
    const vrt_outputs *out = &view;
    if (out->memory != VRT_MEM_HOST && out->memory != VRT_MEM_DEVICE) return fail(VRT_E_INVALID_ARG, "bad outputs.memory");
    vrt::TraceParams p;
    bool use_skip = false;
    const uint32_t lane = pick_lane(c, out->memory == VRT_MEM_DEVICE);
    const hipStream_t st = stream_of(c, lane);
    Scratch &S = scratch_of(c, lane);
    { int rc = prepare_trace(c, S, st, true, cam, w, h, rows, false, p, use_skip); if (rc != VRT_OK) return rc; }

    const size_t px = (size_t)rows.rows * w;
    constexpr int kPlanes = 8;
    Plane planes[kPlanes] = {{0, out->bgra8 ? px * 4 : 0, out->bgra8},       {0, out->voxel ? px * 12 : 0, out->voxel},
                             {0, out->face ? px * 4 : 0, out->face},         {0, out->steps ? px * 4 : 0, out->steps},
                             {0, out->t ? px * 4 : 0, out->t},               {0, out->hit_point ? px * 12 : 0, out->hit_point},
                             {0, out->rgba ? px * 16 : 0, out->rgba},        {0, out->bgr8 ? px * 3 : 0, out->bgr8}};
    void *dev[kPlanes];
    if (out->memory == VRT_MEM_DEVICE) {
        for (int i = 0; i < kPlanes; ++i) dev[i] = planes[i].host;
    } else {
        size_t total = 0;
        for (int i = 0; i < kPlanes; ++i) { planes[i].off = total; total += (planes[i].bytes + 255) & ~(size_t)255; }
        { int rc = ensure_stage(c, total); if (rc != VRT_OK) return rc; }
        for (int i = 0; i < kPlanes; ++i) dev[i] = planes[i].bytes ? (char *)c->d_stage + planes[i].off : nullptr;
    }
    p.bgra8 = (uint8_t *)dev[0]; p.voxel = (int32_t *)dev[1]; p.face = (int32_t *)dev[2]; p.steps = (uint32_t *)dev[3];
    p.t = (float *)dev[4]; p.hit_point = (float *)dev[5]; p.rgba = (float *)dev[6]; p.bgr8 = (uint8_t *)dev[7];

    const bool timed = c->time_every != 0 && c->launch_seq++ % c->time_every == 0;
    if (timed && c->ev_pending == kEventRing) { int rc = drain_events(c); if (rc != VRT_OK) return rc; }
    const int e = c->ev_pending;
    if (timed) HIP_TRY(hipEventRecord(c->ev0[e], st));
    if (use_skip && p.tile_log2h <= 4) p.tile_lanes = c->knob_tile_lanes ? c->knob_tile_lanes : 64;
    if (use_skip) { int rc = attach_row_sched(c, S, p, 1, c->overlap > 1 && out->memory == VRT_MEM_DEVICE); if (rc != VRT_OK) return rc; }
    if (use_skip) { HIP_TRY(vrt::launch_trace_skip(p, c->voxel_bytes, st)); c->last_kernel = "skip"; }
    else          { HIP_TRY(vrt::launch_trace_literal(p, c->voxel_bytes, st)); c->last_kernel = "literal"; }
    if (timed) { HIP_TRY(hipEventRecord(c->ev1[e], st)); c->ev_pending = e + 1; }
    if (lane) c->lane_busy[lane] = true;

    if (out->memory == VRT_MEM_HOST) {  // (lane 0: its stream)
        for (int i = 0; i < kPlanes; ++i)
            if (planes[i].bytes) HIP_TRY(hipMemcpyAsync(planes[i].host, dev[i], planes[i].bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return VRT_OK;
}

}  // namespace

extern "C" {

const char *vrt_last_error(void) { return g_err; }
const char *vrt_version(void) { return "vrt 0.1 (gfx950)"; }

int vrt_create(const vrt_config *cfg, vrt_ctx **out)
{
    if (!cfg || !out) return fail(VRT_E_INVALID_ARG, "null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(vrt_config)) return fail(VRT_E_INVALID_ARG, "vrt_config.struct_size mismatch");
    if (cfg->grid_n < 8 || cfg->grid_n > 4096 || cfg->grid_n % 8 != 0)
        return fail(VRT_E_INVALID_ARG, "grid_n must be a multiple of 8 in [8, 4096]");
    if (cfg->voxel_format != VRT_VOXEL_REF_U16U16 && cfg->voxel_format != VRT_VOXEL_PACKED_U8)
        return fail(VRT_E_INVALID_ARG, "unknown voxel_format");
    if (cfg->trace_mode > VRT_TRACE_SKIP) return fail(VRT_E_INVALID_ARG, "unknown trace_mode");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(VRT_E_NO_DEVICE, "no HIP device available");
    if (cfg->device < 0 || cfg->device >= count) return fail(VRT_E_NO_DEVICE, "device %d out of range (%d devices)", cfg->device, count);
    HIP_TRY(hipSetDevice(cfg->device));
    {   // the static scratch of vrt_trace_static: its address on this device, taken now so that the entry point itself
        // has no set-up left to do
        StaticScratch *unused = nullptr;
        int rc = static_scratch_for(cfg->device, &unused);
        if (rc != VRT_OK) return rc;
    }

    vrt_ctx *c = new (std::nothrow) vrt_ctx();
    if (!c) return fail(VRT_E_OUT_OF_MEMORY, "host allocation failed");
    c->device      = cfg->device;
    c->n           = cfg->grid_n;
    c->format      = cfg->voxel_format;
    c->voxel_bytes = cfg->voxel_format == VRT_VOXEL_PACKED_U8 ? 1 : 2;
    c->max_steps   = cfg->max_steps ? cfg->max_steps : 4 * cfg->grid_n;
    c->trace_mode  = cfg->trace_mode;
    for (int i = 0; i < kEventRing; ++i) { c->ev0[i] = nullptr; c->ev1[i] = nullptr; }
    if (const char *v = std::getenv("VRT_TILE_LOG2H")) c->knob_tile_log2h = (uint32_t)std::atoi(v) % 6u;
    if (const char *v = std::getenv("VRT_TILE_LANES")) c->knob_tile_lanes = std::atoi(v) == 32 ? 32 : (std::atoi(v) == 64 ? 64 : 0);
    if (const char *v = std::getenv("VRT_NO_ROW_ORDER")) c->knob_no_row_order = v[0] == '1';
    if (const char *v = std::getenv("VRT_ROW_BLOCK0")) c->knob_row_block0 = std::atoi(v);
    if (const char *v = std::getenv("VRT_SCHED_DEBUG")) c->knob_sched_debug = v[0] == '1';
    if (const char *v = std::getenv("VRT_NO_SKY")) c->knob_no_sky = v[0] == '1';
    if (const char *v = std::getenv("VRT_SKY_REACH")) { const float f = (float)std::atof(v); if (f >= 0.05f && f <= 100.0f) c->knob_sky_reach = f; }
    if (const char *v = std::getenv("VRT_SECONDARY")) c->knob_secondary_binned = std::strcmp(v, "pixel") != 0;
    auto bail = [&](int code) { vrt_destroy(c); return code; };
#define TRY_OR_BAIL(expr)                                                                                     \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return bail(fail(e_ == hipErrorOutOfMemory ? VRT_E_OUT_OF_MEMORY : VRT_E_HIP, "%s failed: %s",    \
                             #expr, hipGetErrorString(e_)));                                                  \
    } while (0)
    TRY_OR_BAIL(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (int i = 0; i < kEventRing; ++i) { TRY_OR_BAIL(hipEventCreate(&c->ev0[i])); TRY_OR_BAIL(hipEventCreate(&c->ev1[i])); }
    TRY_OR_BAIL(hipMalloc(&c->d_block, texels(c) * (size_t)c->voxel_bytes));
    TRY_OR_BAIL(hipMalloc(&c->d_flag, sizeof(uint32_t)));
    {
        static uint64_t next_serial = 0;
        c->serial = ++next_serial;
        Scratch &S = c->own;
        S.max_dim = 16384; S.sin_cap = c->n + 8; S.sched_capacity = kSchedCapacity;
        TRY_OR_BAIL(hipMalloc(&S.tables, S.table_floats() * sizeof(float)));
        TRY_OR_BAIL(hipMalloc(&S.sched, S.sched_bytes()));
        TRY_OR_BAIL(hipMemsetAsync(S.sched, 0, S.sched_bytes(), c->stream));
    }
#undef TRY_OR_BAIL
    *out = c;
    return VRT_OK;
}

int vrt_destroy(vrt_ctx *c)
{
    if (!c) return VRT_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int i = 1; i < kMaxLanes; ++i) {
        if (c->lane_stream[i]) { (void)hipStreamSynchronize(c->lane_stream[i]); (void)hipStreamDestroy(c->lane_stream[i]); }
        if (c->lane_scratch[i].tables) (void)hipFree(c->lane_scratch[i].tables);
        if (c->lane_scratch[i].sched) (void)hipFree(c->lane_scratch[i].sched);
    }
    for (int i = 0; i < kMaxLanes; ++i)
        if (c->lane_ev[i]) (void)hipEventDestroy(c->lane_ev[i]);
    for (int i = 0; i < kEventRing; ++i) {
        if (c->ev0[i]) (void)hipEventDestroy(c->ev0[i]);
        if (c->ev1[i]) (void)hipEventDestroy(c->ev1[i]);
    }
    void *bufs[] = {c->d_block, c->d_dist, c->d_cells, c->d_tops, c->d_build, c->own.tables, c->d_flag, c->own.sched, c->d_stage};
    for (void *b : bufs) if (b) (void)hipFree(b);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return VRT_OK;
}

int vrt_upload_world(vrt_ctx *c, const void *block, const void *dist, uint32_t n, uint32_t memory)
{
    if (!c || !block) return fail(VRT_E_INVALID_ARG, "null argument");
    if (n != c->n) return fail(VRT_E_INVALID_ARG, "n = %u does not match the context's grid_n = %u", n, c->n);
    if (memory != VRT_MEM_HOST && memory != VRT_MEM_DEVICE) return fail(VRT_E_INVALID_ARG, "bad memory kind");
    if (dist && c->format == VRT_VOXEL_PACKED_U8) return fail(VRT_E_INVALID_ARG, "PACKED_U8 has no distance image");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = join_lanes(c); if (rc != VRT_OK) return rc; }  // (the call ends with a wait for the stream: no fork needed)
    const hipMemcpyKind kind = memory == VRT_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    HIP_TRY(hipMemcpyAsync(c->d_block, block, texels(c) * (size_t)c->voxel_bytes, kind, c->stream));
    c->dist_nonzero = false;
    if (dist) {
        if (!c->d_dist) HIP_TRY(hipMalloc(&c->d_dist, texels(c) * sizeof(uint16_t)));
        HIP_TRY(hipMemcpyAsync(c->d_dist, dist, texels(c) * sizeof(uint16_t), kind, c->stream));
        HIP_TRY(vrt::launch_any_nonzero_u16(c->d_dist, texels(c), c->d_flag, c->stream));
        uint32_t flag = 0;
        HIP_TRY(hipMemcpyAsync(&flag, c->d_flag, sizeof flag, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->dist_nonzero = flag != 0;
    } else {
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    world_replaced(c);
    return VRT_OK;
}

int vrt_generate_chunks(vrt_ctx *c, const int32_t *xyz, uint32_t count)
{
    if (!c || (!xyz && count)) return fail(VRT_E_INVALID_ARG, "null argument");
    if (c->n < VRT_CHUNK_SIZE) return fail(VRT_E_UNSUPPORTED, "grid_n smaller than one chunk");
    if (count > (1u << 24)) return fail(VRT_E_INVALID_ARG, "%u chunks in one call", count);
    const int32_t lim = (1 << 20) / VRT_CHUNK_SIZE;
    for (uint32_t i = 0; i < 3 * count; ++i)
        if (xyz[i] < -lim || xyz[i] > lim) return fail(VRT_E_INVALID_ARG, "chunk position beyond +/-2^20 voxels");
    if (count == 0) return VRT_OK;
    HIP_TRY(hipSetDevice(c->device));
    { int rc = join_lanes(c); if (rc != VRT_OK) return rc; }  // traces in flight read the voxels; the next trace re-indexes, which forks
    HIP_TRY(vrt::launch_terrain_chunks(c->d_block, c->voxel_bytes, (int)c->n, xyz, count, c->stream));
    { int rc = fork_lanes(c); if (rc != VRT_OK) return rc; }
    for (uint32_t i = 0; i < count; ++i) {
        const int32_t lo[3] = {xyz[3 * i] * VRT_CHUNK_SIZE, xyz[3 * i + 1] * VRT_CHUNK_SIZE, xyz[3 * i + 2] * VRT_CHUNK_SIZE};
        world_box_changed(c, lo, VRT_CHUNK_SIZE);
    }
    return VRT_OK;
}

int vrt_generate_chunk(vrt_ctx *c, int32_t cx, int32_t cy, int32_t cz)
{
    const int32_t xyz[3] = {cx, cy, cz};
    return vrt_generate_chunks(c, xyz, 1);
}

int vrt_generate_window(vrt_ctx *c, const int32_t wmin[3])
{
    if (!c || !wmin) return fail(VRT_E_INVALID_ARG, "null argument");
    for (int k = 0; k < 3; ++k)
        if (wmin[k] < -(1 << 21) || wmin[k] > (1 << 21)) return fail(VRT_E_INVALID_ARG, "window origin beyond +/-2^21");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = join_lanes(c); if (rc != VRT_OK) return rc; }
    HIP_TRY(vrt::launch_terrain(c->d_block, c->voxel_bytes, (int)c->n, wmin[0], wmin[1], wmin[2], (int)c->n, (int)c->n,
                                (int)c->n, c->stream));
    { int rc = fork_lanes(c); if (rc != VRT_OK) return rc; }
    world_replaced(c);
    return VRT_OK;
}

int vrt_reset_distance_field(vrt_ctx *c)
{
    if (!c) return fail(VRT_E_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    if (c->d_dist) {
        { int rc = join_lanes(c); if (rc != VRT_OK) return rc; }
        HIP_TRY(hipMemsetAsync(c->d_dist, 0, texels(c) * sizeof(uint16_t), c->stream));
        { int rc = fork_lanes(c); if (rc != VRT_OK) return rc; }
    }
    c->dist_nonzero = false;
    if (c->have_world && c->trace_mode != VRT_TRACE_LITERAL) {
        // (the pyramid is levelled by the next trace, which knows the camera's window)
        return ensure_skip_structure(c, -1);
    }
    return VRT_OK;
}

int vrt_trace(vrt_ctx *c, const float cam[16], uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, const vrt_outputs *out)
{
    if (y0 > y1 || y1 > h) return fail(VRT_E_INVALID_ARG, "row band [%u,%u) outside the frame", y0, y1);
    vrt::RowMap rows{y0, y1 - y0, 0, 0, 1};
    return do_trace(c, cam, w, h, rows, out);
}

int vrt_trace_static(vrt_ctx *c, const float cam[16], uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, const vrt_outputs *given)
{
    if (!c || !cam || !given) return fail(VRT_E_INVALID_ARG, "null argument");
    vrt_outputs view;
    { int rc = read_outputs(given, view); if (rc != VRT_OK) return rc; }
    const vrt_outputs *out = &view;
    if (out->memory != VRT_MEM_DEVICE)
        return fail(VRT_E_INVALID_ARG, "vrt_trace_static writes DEVICE planes only (host planes need a staging buffer)");
    if (y0 > y1 || y1 > h) return fail(VRT_E_INVALID_ARG, "row band [%u,%u) outside the frame", y0, y1);
    if (w == 0 || h == 0 || w > kStaticMaxDim || h > kStaticMaxDim)
        return fail(VRT_E_INVALID_ARG, "frame %u x %u beyond the static scratch's %u x %u", w, h, kStaticMaxDim, kStaticMaxDim);
    if (c->n > kStaticMaxGrid) return fail(VRT_E_INVALID_ARG, "grid_n %u beyond the static scratch's %u", c->n, kStaticMaxGrid);
    {   // the dispatch-order slots: one per wavefront of the launch
        const uint32_t lh = c->knob_tile_log2h, tw = 64u >> lh, bh = 4u << lh;
        const uint64_t waves = (uint64_t)((w + tw - 1) / tw) * ((y1 - y0 + bh - 1) / bh) * 4u;
        if (waves > kStaticSchedCap)
            return fail(VRT_E_INVALID_ARG, "%llu wavefronts of 64 pixels beyond the static scratch's %u (3840 x 2160 has 129600)",
                        (unsigned long long)waves, kStaticSchedCap);
    }
    StaticScratch *st = nullptr;
    { int rc = static_scratch_for(c->device, &st); if (rc != VRT_OK) return rc; }
    if (st->in_use.exchange(true, std::memory_order_acquire))
        return fail(VRT_E_UNSUPPORTED, "the static scratch of device %d is in use: another vrt_trace_static call is running on this GPU",
                    c->device);
    struct Release { std::atomic<bool> &f; ~Release() { f.store(false, std::memory_order_release); } } release{st->in_use};
    Scratch *S = &st->s;
    // launches from another stream than the last one: behind that one (its tables and slots are this one's too)
    if (st->last_stream && st->last_stream != c->stream) HIP_TRY(hipStreamWaitEvent(c->stream, st->done, 0));
    vrt::RowMap rows{y0, y1 - y0, 0, 0, 1};
    vrt::TraceParams p;
    bool use_skip = false;
    { int rc = prepare_trace(c, *S, c->stream, false, cam, w, h, rows, false, p, use_skip); if (rc != VRT_OK) return rc; }
    p.bgra8 = out->bgra8; p.voxel = out->voxel; p.face = out->face; p.steps = out->steps;
    p.t = out->t; p.hit_point = out->hit_point; p.rgba = out->rgba; p.bgr8 = out->bgr8;
    if (use_skip) { int rc = attach_row_sched(c, *S, p, 1); if (rc != VRT_OK) return rc; }
    if (use_skip) { HIP_TRY(vrt::launch_trace_skip(p, c->voxel_bytes, c->stream)); c->last_kernel = "skip"; }
    else          { HIP_TRY(vrt::launch_trace_literal(p, c->voxel_bytes, c->stream)); c->last_kernel = "literal"; }
    HIP_TRY(hipEventRecord(st->done, c->stream));
    st->last_stream = c->stream;
    return VRT_OK;
}

int vrt_trace_secondary(vrt_ctx *c, const float cam[16], uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
                        const vrt_secondary_outputs *out)
{
    if (!c || !cam || !out) return fail(VRT_E_INVALID_ARG, "null argument");
    if (out->struct_size != sizeof(vrt_secondary_outputs)) return fail(VRT_E_INVALID_ARG, "vrt_secondary_outputs.struct_size mismatch");
    if (out->memory != VRT_MEM_HOST && out->memory != VRT_MEM_DEVICE) return fail(VRT_E_INVALID_ARG, "bad outputs.memory");
    if (y0 > y1 || y1 > h) return fail(VRT_E_INVALID_ARG, "row band [%u,%u) outside the frame", y0, y1);
    vrt::RowMap rows{y0, y1 - y0, 0, 0, 1};
    vrt::TraceParams p;
    bool use_skip = false;
    const uint32_t lane = pick_lane(c, out->memory == VRT_MEM_DEVICE);
    const hipStream_t st = stream_of(c, lane);
    Scratch &S = scratch_of(c, lane);
    { int rc = prepare_trace(c, S, st, true, cam, w, h, rows, true, p, use_skip); if (rc != VRT_OK) return rc; }
    const size_t px = (size_t)rows.rows * w;
    Plane planes[3] = {{0, out->shadow ? px : 0, out->shadow}, {0, out->ao ? px : 0, out->ao},
                       {0, out->sec_steps ? px * 4 : 0, out->sec_steps}};
    void *dev[3];
    if (out->memory == VRT_MEM_DEVICE) {
        for (int i = 0; i < 3; ++i) dev[i] = planes[i].host;
    } else {
        size_t total = 0;
        for (int i = 0; i < 3; ++i) { planes[i].off = total; total += (planes[i].bytes + 255) & ~(size_t)255; }
        { int rc = ensure_stage(c, total); if (rc != VRT_OK) return rc; }
        for (int i = 0; i < 3; ++i) dev[i] = planes[i].bytes ? (char *)c->d_stage + planes[i].off : nullptr;
    }
    const bool timed = c->time_every != 0 && c->launch_seq++ % c->time_every == 0;
    if (timed && c->ev_pending == kEventRing) { int rc = drain_events(c); if (rc != VRT_OK) return rc; }
    const int e = c->ev_pending;
    if (timed) HIP_TRY(hipEventRecord(c->ev0[e], st));
    { int rc = attach_row_sched(c, S, p, 2, c->overlap > 1 && out->memory == VRT_MEM_DEVICE); if (rc != VRT_OK) return rc; }
    HIP_TRY(vrt::launch_trace_secondary(p, c->voxel_bytes, c->knob_secondary_binned, (uint8_t *)dev[0], (uint8_t *)dev[1],
                                        (uint32_t *)dev[2], st));
    c->last_kernel = "secondary";
    if (timed) { HIP_TRY(hipEventRecord(c->ev1[e], st)); c->ev_pending = e + 1; }
    if (lane) c->lane_busy[lane] = true;
    if (out->memory == VRT_MEM_HOST) {
        for (int i = 0; i < 3; ++i)
            if (planes[i].bytes) HIP_TRY(hipMemcpyAsync(planes[i].host, dev[i], planes[i].bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return VRT_OK;
}

uint32_t vrt_strip_rows(uint32_t h, uint32_t strip_h, uint32_t part, uint32_t parts)
{
    if (strip_h == 0 || parts == 0 || part >= parts) return 0;
    uint32_t rows = 0;
    for (uint64_t s = part; s * strip_h < h; s += parts) {
        const uint64_t a = s * strip_h, b = a + strip_h < h ? a + strip_h : h;
        rows += (uint32_t)(b - a);
    }
    return rows;
}

int vrt_trace_strips(vrt_ctx *c, const float cam[16], uint32_t w, uint32_t h, uint32_t strip_h, uint32_t part,
                     uint32_t parts, const vrt_outputs *out)
{
    if (strip_h == 0 || parts == 0 || part >= parts) return fail(VRT_E_INVALID_ARG, "bad strip partition");
    vrt::RowMap rows{0, vrt_strip_rows(h, strip_h, part, parts), strip_h, part, parts};
    return do_trace(c, cam, w, h, rows, out);
}

int vrt_assemble_strips(vrt_ctx *c, const void *gathered, void *frame, uint32_t row_bytes, uint32_t h,
                        uint32_t strip_h, uint32_t parts, uint32_t part_stride_rows)
{
    if (!c || !gathered || !frame) return fail(VRT_E_INVALID_ARG, "null argument");
    if (strip_h == 0 || parts == 0 || row_bytes == 0 || h == 0 || h > 16384u)
        return fail(VRT_E_INVALID_ARG, "bad strip partition or frame size");
    uint32_t most = 0;
    for (uint32_t part = 0; part < parts; ++part) {
        const uint32_t rows = vrt_strip_rows(h, strip_h, part, parts);
        if (rows > most) most = rows;
    }
    if (part_stride_rows < most)
        return fail(VRT_E_INVALID_ARG, "part_stride_rows %u is less than the %u rows of the largest part", part_stride_rows, most);
    if (gathered == frame) return fail(VRT_E_INVALID_ARG, "gathered and frame must not overlap");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = join_lanes(c); if (rc != VRT_OK) return rc; }
    HIP_TRY(vrt::launch_assemble_strips(gathered, frame, row_bytes, h, strip_h, parts, part_stride_rows, c->stream));
    return VRT_OK;
}

int vrt_assemble_strips_bgr(vrt_ctx *c, const void *gathered_bgr, void *frame_bgra, uint32_t w, uint32_t h, uint32_t strip_h,
                            uint32_t parts, uint32_t part_stride_rows)
{
    if (!c || !gathered_bgr || !frame_bgra) return fail(VRT_E_INVALID_ARG, "null argument");
    if (strip_h == 0 || parts == 0 || w == 0 || h == 0 || h > 16384u || w > 16384u)
        return fail(VRT_E_INVALID_ARG, "bad strip partition or frame size");
    if (w % 4u != 0) return fail(VRT_E_INVALID_ARG, "frame width %u is not a multiple of 4 pixels", w);
    if (((uintptr_t)frame_bgra & 15u) || ((uintptr_t)gathered_bgr & 3u))
        return fail(VRT_E_INVALID_ARG, "frame must be 16-byte and the gathered parts 4-byte aligned");
    uint32_t most = 0;
    for (uint32_t part = 0; part < parts; ++part) most = std::max(most, vrt_strip_rows(h, strip_h, part, parts));
    if (part_stride_rows < most)
        return fail(VRT_E_INVALID_ARG, "part_stride_rows %u is less than the %u rows of the largest part", part_stride_rows, most);
    if (gathered_bgr == frame_bgra) return fail(VRT_E_INVALID_ARG, "gathered and frame must not overlap");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = join_lanes(c); if (rc != VRT_OK) return rc; }
    HIP_TRY(vrt::launch_assemble_strips_bgr(gathered_bgr, frame_bgra, w, h, strip_h, parts, part_stride_rows, c->stream));
    return VRT_OK;
}

static int check_pixel_buffers(vrt_ctx *c, const void *a, const void *b, size_t pixels)
{
    if (!c || !a || !b) return fail(VRT_E_INVALID_ARG, "null argument");
    if (pixels % 4 != 0) return fail(VRT_E_INVALID_ARG, "pixel count %zu is not a multiple of 4", pixels);
    if (pixels > ((size_t)1 << 38)) return fail(VRT_E_INVALID_ARG, "pixel count out of range");
    return VRT_OK;
}

int vrt_pack_bgr(vrt_ctx *c, const void *bgra, void *bgr, size_t pixels)
{
    { int rc = check_pixel_buffers(c, bgra, bgr, pixels); if (rc != VRT_OK) return rc; }
    if (((uintptr_t)bgra & 15u) || ((uintptr_t)bgr & 3u)) return fail(VRT_E_INVALID_ARG, "bgra must be 16-byte and bgr 4-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = join_lanes(c); if (rc != VRT_OK) return rc; }
    HIP_TRY(vrt::launch_pack_bgr(bgra, bgr, pixels, c->stream));
    return VRT_OK;
}

int vrt_unpack_bgra(vrt_ctx *c, const void *bgr, void *bgra, size_t pixels)
{
    { int rc = check_pixel_buffers(c, bgr, bgra, pixels); if (rc != VRT_OK) return rc; }
    if (((uintptr_t)bgra & 15u) || ((uintptr_t)bgr & 3u)) return fail(VRT_E_INVALID_ARG, "bgra must be 16-byte and bgr 4-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = join_lanes(c); if (rc != VRT_OK) return rc; }
    HIP_TRY(vrt::launch_unpack_bgra(bgr, bgra, pixels, c->stream));
    return VRT_OK;
}

int vrt_alloc_host(vrt_ctx *c, size_t bytes, void **out)
{
    if (!c || !out || bytes == 0) return fail(VRT_E_INVALID_ARG, "null argument or zero size");
    *out = nullptr;
    HIP_TRY(hipSetDevice(c->device));
    const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return fail(VRT_E_OUT_OF_MEMORY, "hipHostMalloc(%zu) failed", bytes); }
    HIP_TRY(e);
    return VRT_OK;
}

int vrt_free_host(vrt_ctx *c, void *ptr)
{
    if (!c) return fail(VRT_E_INVALID_ARG, "null argument");
    if (!ptr) return VRT_OK;
    HIP_TRY(hipSetDevice(c->device));
    { int rc = sync_lanes(c); if (rc != VRT_OK) return rc; }  // a copy into it may still be in flight
    HIP_TRY(hipHostFree(ptr));
    return VRT_OK;
}

int vrt_sync(vrt_ctx *c)
{
    if (!c) return fail(VRT_E_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    return sync_lanes(c);
}

int vrt_join(vrt_ctx *c)
{
    if (!c) return fail(VRT_E_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    return join_lanes(c);
}

int vrt_fork(vrt_ctx *c)
{
    if (!c) return fail(VRT_E_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    return fork_lanes(c);
}

int vrt_set_frame_overlap(vrt_ctx *c, uint32_t streams)
{
    if (!c) return fail(VRT_E_INVALID_ARG, "null argument");
    if (streams < 1 || streams > (uint32_t)kMaxLanes) return fail(VRT_E_INVALID_ARG, "frames overlap on 1..%d streams, not %u", kMaxLanes, streams);
    HIP_TRY(hipSetDevice(c->device));
    { int rc = sync_lanes(c); if (rc != VRT_OK) return rc; }
    if (!c->lane_ev[0]) HIP_TRY(hipEventCreateWithFlags(&c->lane_ev[0], hipEventDisableTiming));
    for (uint32_t i = 1; i < streams; ++i) {  // lanes are set up once and kept
        if (!c->lane_stream[i]) HIP_TRY(hipStreamCreateWithFlags(&c->lane_stream[i], hipStreamNonBlocking));
        if (!c->lane_ev[i]) HIP_TRY(hipEventCreateWithFlags(&c->lane_ev[i], hipEventDisableTiming));
        Scratch &S = c->lane_scratch[i];
        if (!S.tables) {
            S.max_dim = c->own.max_dim; S.sin_cap = c->own.sin_cap; S.sched_capacity = c->own.sched_capacity;
            HIP_TRY(hipMalloc(&S.tables, S.table_floats() * sizeof(float)));
        }
        if (!S.sched) {
            HIP_TRY(hipMalloc(&S.sched, S.sched_bytes()));
            HIP_TRY(hipMemsetAsync(S.sched, 0, S.sched_bytes(), c->lane_stream[i]));
            HIP_TRY(hipStreamSynchronize(c->lane_stream[i]));
        }
    }
    c->overlap = streams;
    c->next_lane = 0;
    return VRT_OK;
}

int vrt_download_world(vrt_ctx *c, void *block_out, uint32_t memory)
{
    if (!c || !block_out) return fail(VRT_E_INVALID_ARG, "null argument");
    if (memory != VRT_MEM_HOST && memory != VRT_MEM_DEVICE) return fail(VRT_E_INVALID_ARG, "bad memory kind");
    if (!c->have_world) return fail(VRT_E_NO_WORLD, "no world uploaded or generated");
    HIP_TRY(hipSetDevice(c->device));
    const hipMemcpyKind kind = memory == VRT_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    HIP_TRY(hipMemcpyAsync(block_out, c->d_block, texels(c) * (size_t)c->voxel_bytes, kind, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VRT_OK;
}

int vrt_download_index(vrt_ctx *c, void *cells_out, void *pyr_out, uint32_t memory)
{
    if (!c) return fail(VRT_E_INVALID_ARG, "null argument");
    if (memory != VRT_MEM_HOST && memory != VRT_MEM_DEVICE) return fail(VRT_E_INVALID_ARG, "bad memory kind");
    if (!c->have_world) return fail(VRT_E_NO_WORLD, "no world uploaded or generated");
    if (c->dist_nonzero) return fail(VRT_E_UNSUPPORTED, "a world with a non-zero distance image has no traversal index");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = ensure_skip_structure(c, -1); if (rc != VRT_OK) return rc; }
    if (pyr_out && c->pyr_offcy < 0) return fail(VRT_E_UNSUPPORTED, "the box words are levelled by the first trace after a world change");
    // (the cells, records included, are written together with the box words: before the first trace, for a window at the
    // bottom of the torus; the next trace levels the words for its own window)
    if (c->pyr_offcy < 0) { int rc = ensure_skip_structure(c, 0); if (rc != VRT_OK) return rc; }
    // 16 (the record) and 64 (the column-top words) of every 128 bytes: gathered by a kernel, into the caller's device
    // memory or into a staging buffer
    void *stage = nullptr;
    if (memory == VRT_MEM_HOST) HIP_TRY(hipMalloc(&stage, cells(c) * vrt::PYR_U4 * sizeof(uint4)));
    const uint4 *src[2] = {c->d_cells, c->d_pyr};
    const unsigned part[2] = {1u, vrt::PYR_U4};
    void *dst[2] = {cells_out, pyr_out};
    for (int i = 0; i < 2; ++i) {
        if (!dst[i]) continue;
        const size_t bytes = cells(c) * part[i] * sizeof(uint4);
        hipError_t e = vrt::launch_copy_cell_part(src[i], static_cast<uint4 *>(stage ? stage : dst[i]), cells(c), part[i], c->stream);
        if (e == hipSuccess && stage) e = hipMemcpyAsync(dst[i], stage, bytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && stage) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { if (stage) (void)hipFree(stage); HIP_TRY(e); }
    }
    if (stage) HIP_TRY(hipFree(stage));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VRT_OK;
}

int vrt_kernel_time(vrt_ctx *c, double *total_ms, uint64_t *launches)
{
    if (!c) return fail(VRT_E_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = drain_events(c);
    if (rc != VRT_OK) return rc;
    if (total_ms) *total_ms = c->total_ms;
    if (launches) *launches = c->launches;
    return VRT_OK;
}

int vrt_kernel_time_reset(vrt_ctx *c)
{
    if (!c) return fail(VRT_E_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = drain_events(c);
    c->total_ms = 0.0;
    c->launches = 0;
    return rc;
}

int vrt_kernel_time_sampling(vrt_ctx *c, uint32_t every_n)
{
    if (!c) return fail(VRT_E_INVALID_ARG, "null argument");
    c->time_every = every_n;
    c->launch_seq = 0;
    return VRT_OK;
}

int vrt_kernel_occupancy(vrt_ctx *c, uint32_t kind, int32_t *workgroups_per_cu)
{
    if (!c || !workgroups_per_cu) return fail(VRT_E_INVALID_ARG, "null argument");
    if (kind > 2u) return fail(VRT_E_INVALID_ARG, "kernel kind %u (0: skip, 1: binned secondary, 2: literal)", kind);
    HIP_TRY(hipSetDevice(c->device));
    int n = 0;
    HIP_TRY(vrt::trace_kernel_occupancy((int)kind, &n));
    *workgroups_per_cu = n;
    return VRT_OK;
}

const char *vrt_last_kernel_name(vrt_ctx *c) { return c ? c->last_kernel : "none"; }
void *vrt_stream(vrt_ctx *c) { return c ? (void *)c->stream : nullptr; }
int vrt_device(vrt_ctx *c) { return c ? c->device : -1; }

}  // extern "C"
