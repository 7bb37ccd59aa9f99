// vrt_comm.hip -- the framebuffer exchange of a multi-GPU frame for hosts that drive several GPUs from ONE
// process (a Rust or C++ host has no torch.distributed): n contexts on n distinct HIP devices, one RCCL
// communicator per device (ncclCommInitAll), and per frame one group of ncclSend / ncclRecv that moves every
// context's packed planes to the first context's device, enqueued on the contexts' own streams behind the
// traces that fill them.  The reference renders on one device and has no such step
// (src/graphics/mod.rs:325-342 picks one physical device; :581-597 is its only dispatch).
//
// RCCL is loaded when the first communicator is created (dlopen of librccl.so.1, the copy the process already
// holds if there is one), so a host that never shards a frame needs no RCCL on its machine.
#include "../../include/vrt.h"
#include "vrt_kernels.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclCommInitAll)   CommInitAll = nullptr;
    decltype(&ncclCommDestroy)   CommDestroy = nullptr;
    decltype(&ncclGroupStart)    GroupStart = nullptr;
    decltype(&ncclGroupEnd)      GroupEnd = nullptr;
    decltype(&ncclSend)          Send = nullptr;
    decltype(&ncclRecv)          Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;

int load_rccl()
{
    if (g_rccl.handle) return VRT_OK;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    void *h = nullptr;
    for (const char *name : names)
        if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return vrt::set_error(VRT_E_UNSUPPORTED, "RCCL is not available: %s", dlerror());
    Rccl r;
    r.handle = h;
#define VRT_RCCL_SYM(field, name)                                                                        \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                                       \
    if (!r.field) { dlclose(h); return vrt::set_error(VRT_E_UNSUPPORTED, "RCCL has no symbol %s", name); }
    VRT_RCCL_SYM(CommInitAll, "ncclCommInitAll")
    VRT_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    VRT_RCCL_SYM(GroupStart, "ncclGroupStart")
    VRT_RCCL_SYM(GroupEnd, "ncclGroupEnd")
    VRT_RCCL_SYM(Send, "ncclSend")
    VRT_RCCL_SYM(Recv, "ncclRecv")
    VRT_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef VRT_RCCL_SYM
    g_rccl = r;
    return VRT_OK;
}

#define RCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess) return vrt::set_error(VRT_E_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)

#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return vrt::set_error(VRT_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

}  // namespace

struct vrt_comm {
    std::vector<vrt_ctx *>   ctx;
    std::vector<int>         device;
    std::vector<hipStream_t> stream;
    // One RCCL rank per context that has a device of its own; a context that shares the root's device (accepted only
    // with VRT_COMM_ALLOW_SHARED_DEVICE=1, a switch for rehearsing the n-context control flow on a machine with fewer
    // GPUs) has none: its part is copied on the device, ordered by two events.
    std::vector<ncclComm_t>  comm;      // by context; null for a context without a rank
    std::vector<int>         rank;      // RCCL rank of the context, -1 without one
    std::vector<hipEvent_t>  ready;     // rankless contexts: recorded on the context's stream behind what fills its part
    std::vector<hipEvent_t>  taken;     // ... and on the root's stream behind the copy that took it
    bool self_through_rccl = false;  // VRT_COMM_SELF_SENDRECV=1: the root's own part travels by ncclSend / ncclRecv too
};

namespace {

// One exchange: part i of send[i] (bytes[i] bytes, on context i's device) to recv_root + offset[i] on the root's.  Parts
// of contexts with an RCCL rank travel in one group of sends and receives on the contexts' own streams; a part that
// lives on the root's device is copied by the root's stream between two events.
int exchange(vrt_comm *cm, const void *const *send, const size_t *bytes, const size_t *offset, void *recv_root)
{
    const size_t n = cm->ctx.size();
    const bool   self = cm->self_through_rccl && cm->comm[0] != nullptr;
    for (size_t i = 0; i < n; ++i) {  // a context whose frames overlap: its first stream waits for all of them
        int rc = vrt_join(cm->ctx[i]);
        if (rc != VRT_OK) return rc;
    }
    bool any = false;
    for (size_t i = self ? 0 : 1; i < n; ++i) any = any || (bytes[i] != 0 && cm->comm[i] != nullptr);
    if (any) {
        RCCL_TRY(g_rccl.GroupStart());
        ncclResult_t first = ncclSuccess;
        for (size_t i = self ? 0 : 1; i < n && first == ncclSuccess; ++i) {
            if (bytes[i] == 0 || !cm->comm[i]) continue;
            first = g_rccl.Send(send[i], bytes[i], ncclUint8, 0, cm->comm[i], cm->stream[i]);
            if (first == ncclSuccess)
                first = g_rccl.Recv(static_cast<char *>(recv_root) + offset[i], bytes[i], ncclUint8, cm->rank[i], cm->comm[0], cm->stream[0]);
        }
        const ncclResult_t end = g_rccl.GroupEnd();  // the group is closed whatever happened inside it
        RCCL_TRY(first);
        RCCL_TRY(end);
    }
    HIP_TRY(hipSetDevice(cm->device[0]));
    for (size_t i = 1; i < n; ++i) {
        if (bytes[i] == 0 || cm->comm[i]) continue;
        // the root's stream takes the part once the context's stream has filled it; the context's stream goes on
        // (and may write the buffer again) once the part has been taken
        HIP_TRY(hipEventRecord(cm->ready[i], cm->stream[i]));
        HIP_TRY(hipStreamWaitEvent(cm->stream[0], cm->ready[i], 0));
        HIP_TRY(hipMemcpyAsync(static_cast<char *>(recv_root) + offset[i], send[i], bytes[i], hipMemcpyDeviceToDevice, cm->stream[0]));
        HIP_TRY(hipEventRecord(cm->taken[i], cm->stream[0]));
        HIP_TRY(hipStreamWaitEvent(cm->stream[i], cm->taken[i], 0));
    }
    if (!self && bytes[0] != 0)
        HIP_TRY(hipMemcpyAsync(static_cast<char *>(recv_root) + offset[0], send[0], bytes[0], hipMemcpyDeviceToDevice, cm->stream[0]));
    return VRT_OK;
}

// The calls below walk over the contexts' devices; the device that was current on entry is current again on return.
struct KeepDevice {
    int device = -1;
    KeepDevice() { if (hipGetDevice(&device) != hipSuccess) device = -1; }
    ~KeepDevice() { if (device >= 0) (void)hipSetDevice(device); }
};

void destroy_comm(vrt_comm *cm)
{
    for (size_t i = 0; i < cm->comm.size(); ++i)
        if (cm->comm[i]) (void)g_rccl.CommDestroy(cm->comm[i]);
    for (hipEvent_t e : cm->ready) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : cm->taken) if (e) (void)hipEventDestroy(e);
    delete cm;
}

}  // namespace

extern "C" {

int vrt_comm_create(vrt_ctx *const *ctxs, uint32_t n, vrt_comm **out)
{
    if (!ctxs || !out) return vrt::set_error(VRT_E_INVALID_ARG, "null argument");
    *out = nullptr;
    if (n == 0 || n > 64) return vrt::set_error(VRT_E_INVALID_ARG, "a communicator spans 1..64 contexts, not %u", n);
    bool shared_ok = false;
    if (const char *v = std::getenv("VRT_COMM_ALLOW_SHARED_DEVICE")) shared_ok = v[0] == '1';
    for (uint32_t i = 0; i < n; ++i) {
        if (!ctxs[i]) return vrt::set_error(VRT_E_INVALID_ARG, "context %u is null", i);
        for (uint32_t j = 0; j < i; ++j) {
            if (ctxs[j] == ctxs[i]) return vrt::set_error(VRT_E_INVALID_ARG, "contexts %u and %u are one context", j, i);
            // (with the rehearsal switch: any number of contexts on the root's device, the others one per device)
            if (vrt_device(ctxs[j]) == vrt_device(ctxs[i]) && !(shared_ok && vrt_device(ctxs[i]) == vrt_device(ctxs[0])))
                return vrt::set_error(VRT_E_INVALID_ARG, "contexts %u and %u share HIP device %d: one context per device", j, i, vrt_device(ctxs[i]));
        }
    }
    vrt_comm *cm = new (std::nothrow) vrt_comm();
    if (!cm) return vrt::set_error(VRT_E_OUT_OF_MEMORY, "host allocation failed");
    std::vector<int> rank_device;   // devices of the contexts that get an RCCL rank, in rank order (the root first)
    for (uint32_t i = 0; i < n; ++i) {
        cm->ctx.push_back(ctxs[i]);
        cm->device.push_back(vrt_device(ctxs[i]));
        cm->stream.push_back(static_cast<hipStream_t>(vrt_stream(ctxs[i])));
        const bool on_root_device = i > 0 && cm->device[i] == cm->device[0];
        cm->rank.push_back(on_root_device ? -1 : (int)rank_device.size());
        if (!on_root_device) rank_device.push_back(cm->device[i]);
    }
    cm->comm.assign(n, nullptr);
    cm->ready.assign(n, nullptr);
    cm->taken.assign(n, nullptr);
    if (const char *v = std::getenv("VRT_COMM_SELF_SENDRECV")) cm->self_through_rccl = v[0] == '1';
    KeepDevice keep;
    for (uint32_t i = 1; i < n; ++i) {
        if (cm->rank[i] >= 0) continue;
        if (hipSetDevice(cm->device[0]) != hipSuccess || hipEventCreateWithFlags(&cm->ready[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&cm->taken[i], hipEventDisableTiming) != hipSuccess) {
            destroy_comm(cm);
            return vrt::set_error(VRT_E_HIP, "event creation for context %u failed", i);
        }
    }
    if (rank_device.size() > 1 || cm->self_through_rccl) {  // someone to talk to through RCCL
        { int rc = load_rccl(); if (rc != VRT_OK) { destroy_comm(cm); return rc; } }
        std::vector<ncclComm_t> comms(rank_device.size(), nullptr);
        const ncclResult_t r = g_rccl.CommInitAll(comms.data(), (int)rank_device.size(), rank_device.data());
        if (r != ncclSuccess) {
            destroy_comm(cm);
            return vrt::set_error(VRT_E_HIP, "ncclCommInitAll over %zu devices failed: %s", rank_device.size(), g_rccl.GetErrorString(r));
        }
        for (uint32_t i = 0; i < n; ++i)
            if (cm->rank[i] >= 0) cm->comm[i] = comms[(size_t)cm->rank[i]];
    }
    *out = cm;
    return VRT_OK;
}

int vrt_comm_destroy(vrt_comm *cm)
{
    if (!cm) return VRT_OK;
    KeepDevice keep;
    for (size_t i = 0; i < cm->ctx.size(); ++i) {
        (void)hipSetDevice(cm->device[i]);
        (void)hipStreamSynchronize(cm->stream[i]);
    }
    destroy_comm(cm);
    return VRT_OK;
}

uint32_t vrt_comm_size(const vrt_comm *cm) { return cm ? (uint32_t)cm->ctx.size() : 0u; }

int vrt_gather(vrt_comm *cm, const void *const *send, size_t bytes_each, void *recv_root)
{
    KeepDevice keep;
    if (!cm || !send || !recv_root) return vrt::set_error(VRT_E_INVALID_ARG, "null argument");
    const size_t n = cm->ctx.size();
    std::vector<size_t> bytes(n, bytes_each), offset(n);
    for (size_t i = 0; i < n; ++i) {
        if (!send[i] && bytes_each) return vrt::set_error(VRT_E_INVALID_ARG, "send[%zu] is null", i);
        offset[i] = i * bytes_each;
    }
    return exchange(cm, send, bytes.data(), offset.data(), recv_root);
}

int vrt_gather_strips(vrt_comm *cm, const void *const *packed, void *gathered, void *frame, uint32_t row_bytes,
                      uint32_t h, uint32_t strip_h, uint32_t part_stride_rows)
{
    KeepDevice keep;
    if (!cm || !packed || !gathered || !frame) return vrt::set_error(VRT_E_INVALID_ARG, "null argument");
    if (strip_h == 0 || row_bytes == 0 || h == 0 || h > 16384u) return vrt::set_error(VRT_E_INVALID_ARG, "bad strip partition or frame size");
    const size_t   n = cm->ctx.size();
    const uint32_t parts = (uint32_t)n;
    std::vector<size_t> bytes(n), offset(n);
    for (size_t i = 0; i < n; ++i) {
        const uint32_t rows = vrt_strip_rows(h, strip_h, (uint32_t)i, parts);
        if (rows > part_stride_rows)
            return vrt::set_error(VRT_E_INVALID_ARG, "part_stride_rows %u is less than the %u rows of part %zu", part_stride_rows, rows, i);
        if (rows && !packed[i]) return vrt::set_error(VRT_E_INVALID_ARG, "packed[%zu] is null", i);
        bytes[i]  = (size_t)rows * row_bytes;
        offset[i] = i * (size_t)part_stride_rows * row_bytes;
    }
    { int rc = exchange(cm, packed, bytes.data(), offset.data(), gathered); if (rc != VRT_OK) return rc; }
    return vrt_assemble_strips(cm->ctx[0], gathered, frame, row_bytes, h, strip_h, parts, part_stride_rows);
}

int vrt_gather_strips_bgr(vrt_comm *cm, const void *const *packed_bgra, void *const *packed_bgr, void *gathered_bgr,
                          void *frame_bgra, uint32_t w, uint32_t h, uint32_t strip_h, uint32_t part_stride_rows)
{
    KeepDevice keep;
    if (!cm || !packed_bgra || !packed_bgr || !gathered_bgr || !frame_bgra) return vrt::set_error(VRT_E_INVALID_ARG, "null argument");
    if (strip_h == 0 || w == 0 || h == 0 || h > 16384u || w > 16384u) return vrt::set_error(VRT_E_INVALID_ARG, "bad strip partition or frame size");
    if (w % 4u != 0) return vrt::set_error(VRT_E_INVALID_ARG, "frame width %u is not a multiple of 4 pixels", w);
    const size_t   n = cm->ctx.size();
    const uint32_t parts = (uint32_t)n;
    std::vector<size_t> bytes(n), offset(n);
    for (size_t i = 0; i < n; ++i) {
        const uint32_t rows = vrt_strip_rows(h, strip_h, (uint32_t)i, parts);
        if (rows > part_stride_rows)
            return vrt::set_error(VRT_E_INVALID_ARG, "part_stride_rows %u is less than the %u rows of part %zu", part_stride_rows, rows, i);
        if (rows && (!packed_bgra[i] || !packed_bgr[i])) return vrt::set_error(VRT_E_INVALID_ARG, "packed buffer %zu is null", i);
        bytes[i]  = (size_t)rows * w * 3u;
        offset[i] = i * (size_t)part_stride_rows * w * 3u;
        // every context drops the alpha byte of its own rows, on its own stream behind the trace that made them
        if (rows) { int rc = vrt_pack_bgr(cm->ctx[i], packed_bgra[i], packed_bgr[i], (size_t)rows * w); if (rc != VRT_OK) return rc; }
    }
    { int rc = exchange(cm, packed_bgr, bytes.data(), offset.data(), gathered_bgr); if (rc != VRT_OK) return rc; }
    return vrt_assemble_strips_bgr(cm->ctx[0], gathered_bgr, frame_bgra, w, h, strip_h, parts, part_stride_rows);
}

}  // extern "C"
