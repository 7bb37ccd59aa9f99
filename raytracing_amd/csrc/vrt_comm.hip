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
    std::vector<ncclComm_t>  comm;
    bool self_through_rccl = false;  // VRT_COMM_SELF_SENDRECV=1: the root's own part travels by ncclSend / ncclRecv too
};

namespace {

// One group: part i of send[i] (bytes[i] bytes, on context i's device) to recv_root + offset[i] on the root's.
int exchange(vrt_comm *cm, const void *const *send, const size_t *bytes, const size_t *offset, void *recv_root)
{
    const size_t n = cm->ctx.size();
    const bool   self = cm->self_through_rccl;
    for (size_t i = 0; i < n; ++i) {  // a context whose frames overlap: its first stream waits for all of them
        int rc = vrt_join(cm->ctx[i]);
        if (rc != VRT_OK) return rc;
    }
    bool any = false;
    for (size_t i = self ? 0 : 1; i < n; ++i) any = any || bytes[i] != 0;
    if (any) {
        RCCL_TRY(g_rccl.GroupStart());
        ncclResult_t first = ncclSuccess;
        for (size_t i = self ? 0 : 1; i < n && first == ncclSuccess; ++i) {
            if (bytes[i] == 0) continue;
            first = g_rccl.Send(send[i], bytes[i], ncclUint8, 0, cm->comm[i], cm->stream[i]);
            if (first == ncclSuccess)
                first = g_rccl.Recv(static_cast<char *>(recv_root) + offset[i], bytes[i], ncclUint8, (int)i, cm->comm[0], cm->stream[0]);
        }
        const ncclResult_t end = g_rccl.GroupEnd();  // the group is closed whatever happened inside it
        RCCL_TRY(first);
        RCCL_TRY(end);
    }
    if (!self && bytes[0] != 0) {
        HIP_TRY(hipSetDevice(cm->device[0]));
        HIP_TRY(hipMemcpyAsync(static_cast<char *>(recv_root) + offset[0], send[0], bytes[0], hipMemcpyDeviceToDevice, cm->stream[0]));
    }
    return VRT_OK;
}

}  // namespace

extern "C" {

int vrt_comm_create(vrt_ctx *const *ctxs, uint32_t n, vrt_comm **out)
{
    if (!ctxs || !out) return vrt::set_error(VRT_E_INVALID_ARG, "null argument");
    *out = nullptr;
    if (n == 0 || n > 64) return vrt::set_error(VRT_E_INVALID_ARG, "a communicator spans 1..64 contexts, not %u", n);
    for (uint32_t i = 0; i < n; ++i) {
        if (!ctxs[i]) return vrt::set_error(VRT_E_INVALID_ARG, "context %u is null", i);
        for (uint32_t j = 0; j < i; ++j)
            if (vrt_device(ctxs[j]) == vrt_device(ctxs[i]))
                return vrt::set_error(VRT_E_INVALID_ARG, "contexts %u and %u share HIP device %d: one context per device", j, i, vrt_device(ctxs[i]));
    }
    { int rc = load_rccl(); if (rc != VRT_OK) return rc; }
    vrt_comm *cm = new (std::nothrow) vrt_comm();
    if (!cm) return vrt::set_error(VRT_E_OUT_OF_MEMORY, "host allocation failed");
    for (uint32_t i = 0; i < n; ++i) {
        cm->ctx.push_back(ctxs[i]);
        cm->device.push_back(vrt_device(ctxs[i]));
        cm->stream.push_back(static_cast<hipStream_t>(vrt_stream(ctxs[i])));
    }
    if (const char *v = std::getenv("VRT_COMM_SELF_SENDRECV")) cm->self_through_rccl = v[0] == '1';
    cm->comm.assign(n, nullptr);
    const ncclResult_t r = g_rccl.CommInitAll(cm->comm.data(), (int)n, cm->device.data());
    if (r != ncclSuccess) {
        delete cm;
        return vrt::set_error(VRT_E_HIP, "ncclCommInitAll over %u devices failed: %s", n, g_rccl.GetErrorString(r));
    }
    *out = cm;
    return VRT_OK;
}

int vrt_comm_destroy(vrt_comm *cm)
{
    if (!cm) return VRT_OK;
    for (size_t i = 0; i < cm->comm.size(); ++i) {
        (void)hipSetDevice(cm->device[i]);
        (void)hipStreamSynchronize(cm->stream[i]);
    }
    for (ncclComm_t c : cm->comm)
        if (c) (void)g_rccl.CommDestroy(c);
    delete cm;
    return VRT_OK;
}

uint32_t vrt_comm_size(const vrt_comm *cm) { return cm ? (uint32_t)cm->ctx.size() : 0u; }

int vrt_gather(vrt_comm *cm, const void *const *send, size_t bytes_each, void *recv_root)
{
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

}  // extern "C"
