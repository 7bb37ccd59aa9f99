// vrt_gather.hip -- puts gathered strip buffers back into frame order.
//
// After a frame has been traced in interleaved strips (vrt_trace_strips) and the packed planes of all
// parts have been gathered into one buffer, every image row has to move once: row r of the frame is
// packed row (s / parts) * strip_h + r % strip_h of part s % parts, s = r / strip_h.  One launch copies
// all rows; each lane moves one 16-, 4- or 1-byte unit, consecutive lanes consecutive units of a row.
// The reference renders on one device and has no such step (src/graphics/mod.rs:581-597).
#include "vrt_kernels.h"

namespace vrt {

template <typename U>
__global__ __launch_bounds__(256) void k_assemble_strips(const U *__restrict__ gathered, U *__restrict__ frame,
                                                         uint32_t row_units, uint32_t strip_h, uint32_t parts,
                                                         uint32_t part_stride_rows)
{
    const uint32_t r = blockIdx.y;
    const uint32_t s = r / strip_h;
    const uint32_t part = s % parts;
    const uint32_t q = (s / parts) * strip_h + (r - s * strip_h);
    const U *src = gathered + ((size_t)part * part_stride_rows + q) * row_units;
    U *dst = frame + (size_t)r * row_units;
    for (uint32_t u = blockIdx.x * 256u + threadIdx.x; u < row_units; u += gridDim.x * 256u)
        dst[u] = src[u];
}

template <typename U>
static hipError_t assemble(const void *gathered, void *frame, uint32_t row_units, uint32_t h, uint32_t strip_h,
                           uint32_t parts, uint32_t part_stride_rows, hipStream_t s)
{
    uint32_t gx = (row_units + 255u) / 256u;
    if (gx > 64u) gx = 64u;
    hipLaunchKernelGGL(k_assemble_strips<U>, dim3(gx, h), dim3(256), 0, s, (const U *)gathered, (U *)frame, row_units,
                       strip_h, parts, part_stride_rows);
    return hipGetLastError();
}

hipError_t launch_assemble_strips(const void *gathered, void *frame, uint32_t row_bytes, uint32_t h,
                                  uint32_t strip_h, uint32_t parts, uint32_t part_stride_rows, hipStream_t s)
{
    const uintptr_t align = (uintptr_t)gathered | (uintptr_t)frame | row_bytes;
    if ((align & 15u) == 0) return assemble<uint4>(gathered, frame, row_bytes / 16u, h, strip_h, parts, part_stride_rows, s);
    if ((align & 3u) == 0) return assemble<uint32_t>(gathered, frame, row_bytes / 4u, h, strip_h, parts, part_stride_rows, s);
    return assemble<uint8_t>(gathered, frame, row_bytes, h, strip_h, parts, part_stride_rows, s);
}

// ---------------------------------------------------------------------------
// The frame as it travels between GPUs: B, G, R of every pixel, without the alpha byte, which is 255 on every path
// of the shader (render.comp:9-13 stores a = 1).  Three quarters of the bytes over a link that is the bottleneck of
// a gathered frame.  One lane moves four pixels: 16 bytes in, 12 bytes out, or the other way round.
__global__ __launch_bounds__(256) void k_pack_bgr(const uint4 *__restrict__ bgra, uint32_t *__restrict__ bgr, size_t quads)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= quads) return;
    const uint4 p = bgra[i];
    const uint32_t a = p.x & 0xffffffu, b = p.y & 0xffffffu, c = p.z & 0xffffffu, d = p.w & 0xffffffu;
    uint32_t *o = bgr + 3 * i;
    o[0] = a | (b << 24);
    o[1] = (b >> 8) | (c << 16);
    o[2] = (c >> 16) | (d << 8);
}

__global__ __launch_bounds__(256) void k_unpack_bgra(const uint32_t *__restrict__ bgr, uint4 *__restrict__ bgra, size_t quads)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= quads) return;
    const uint32_t *s = bgr + 3 * i;
    const uint32_t w0 = s[0], w1 = s[1], w2 = s[2];
    bgra[i] = make_uint4((w0 & 0xffffffu) | 0xff000000u, (w0 >> 24) | ((w1 & 0xffffu) << 8) | 0xff000000u,
                         (w1 >> 16) | ((w2 & 0xffu) << 16) | 0xff000000u, (w2 >> 8) | 0xff000000u);
}

// Gathered strip buffers that travelled as BGR8 -> the BGRA8 frame in image order: the row reorder of
// k_assemble_strips and the alpha byte of k_unpack_bgra in one pass (3 bytes read, 4 written per pixel; the frame is
// not read or written a second time).  One lane moves four pixels; a row holds row_quads of them.
__global__ __launch_bounds__(256) void k_assemble_strips_bgr(const uint32_t *__restrict__ gathered, uint4 *__restrict__ frame,
                                                             uint32_t row_quads, uint32_t strip_h, uint32_t parts,
                                                             uint32_t part_stride_rows)
{
    const uint32_t r = blockIdx.y;
    const uint32_t s = r / strip_h;
    const uint32_t part = s % parts;
    const uint32_t q = (s / parts) * strip_h + (r - s * strip_h);
    const uint32_t *src = gathered + ((size_t)part * part_stride_rows + q) * row_quads * 3u;
    uint4 *dst = frame + (size_t)r * row_quads;
    for (uint32_t u = blockIdx.x * 256u + threadIdx.x; u < row_quads; u += gridDim.x * 256u) {
        const uint32_t w0 = src[3u * u], w1 = src[3u * u + 1u], w2 = src[3u * u + 2u];
        dst[u] = make_uint4((w0 & 0xffffffu) | 0xff000000u, (w0 >> 24) | ((w1 & 0xffffu) << 8) | 0xff000000u,
                            (w1 >> 16) | ((w2 & 0xffu) << 16) | 0xff000000u, (w2 >> 8) | 0xff000000u);
    }
}

hipError_t launch_assemble_strips_bgr(const void *gathered, void *frame, uint32_t w, uint32_t h, uint32_t strip_h,
                                      uint32_t parts, uint32_t part_stride_rows, hipStream_t s)
{
    const uint32_t row_quads = w / 4u;
    uint32_t gx = (row_quads + 255u) / 256u;
    if (gx > 64u) gx = 64u;
    hipLaunchKernelGGL(k_assemble_strips_bgr, dim3(gx, h), dim3(256), 0, s, (const uint32_t *)gathered, (uint4 *)frame, row_quads,
                       strip_h, parts, part_stride_rows);
    return hipGetLastError();
}

hipError_t launch_pack_bgr(const void *bgra, void *bgr, size_t pixels, hipStream_t s)
{
    const size_t quads = pixels / 4;
    if (quads == 0) return hipSuccess;
    k_pack_bgr<<<(unsigned)((quads + 255) / 256), 256, 0, s>>>((const uint4 *)bgra, (uint32_t *)bgr, quads);
    return hipGetLastError();
}

hipError_t launch_unpack_bgra(const void *bgr, void *bgra, size_t pixels, hipStream_t s)
{
    const size_t quads = pixels / 4;
    if (quads == 0) return hipSuccess;
    k_unpack_bgra<<<(unsigned)((quads + 255) / 256), 256, 0, s>>>((const uint32_t *)bgr, (uint4 *)bgra, quads);
    return hipGetLastError();
}

}  // namespace vrt
