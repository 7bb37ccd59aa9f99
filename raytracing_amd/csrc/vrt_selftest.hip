// vrt_selftest.hip -- exhaustive checks of device arithmetic that the kernels rely on, in a small library
// of its own (libvrt_selftest.so) that only the tests load.
#include "vrt_device.h"
#include "vrt_kernels.h"

namespace {

// thread i checks the 2^12 arguments whose upper 20 bits are i
__global__ __launch_bounds__(256) void k_check_recip(unsigned long long *bad_value, unsigned long long *bad_class,
                                                     unsigned *first_bad)
{
    const unsigned hi = blockIdx.x * 256u + threadIdx.x;
    const float    big = 0x1.0p100f;
    unsigned       nv = 0, nc = 0;
    for (unsigned lo = 0; lo < 4096u; ++lo) {
        const unsigned bits = (hi << 12) | lo;
        const float    x = __uint_as_float(bits);
        if (fabsf(x) > 2.0f) continue;  // a direction component never exceeds 1 (NaNs pass this test and are checked)
        const float    full = 1.0f / x, fast = vrt::recip_regular(x);
        const bool     reg_full = x != 0.0f && fabsf(full) < big, reg_fast = x != 0.0f && fabsf(fast) < big;
        if (reg_full != reg_fast) { if (nc++ == 0) atomicMin(first_bad, bits); }
        else if (reg_full && __float_as_uint(full) != __float_as_uint(fast)) { if (nv++ == 0) atomicMin(first_bad, bits); }
    }
    if (nv) atomicAdd(bad_value, (unsigned long long)nv);
    if (nc) atomicAdd(bad_class, (unsigned long long)nc);
}

}  // namespace

// All float arguments of magnitude up to 2 and the NaNs: out[0] = arguments of ordinary rays (x != 0, |1/x| < 2^100) for which
// recip_regular(x) differs from 1.0f / x; out[1] = arguments that the two classify differently;
// out[2] = the smallest offending bit pattern (0xffffffff: none).  Returns 0, or the HIP error code.
extern "C" int vrt_selftest_recip(int device, unsigned long long out[3])
{
    unsigned long long *d = nullptr;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return (int)e;
    e = hipMalloc(&d, 3 * sizeof(unsigned long long));
    if (e != hipSuccess) return (int)e;
    const unsigned long long init[3] = {0, 0, 0xffffffffull};
    e = hipMemcpy(d, init, sizeof init, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_check_recip<<<4096, 256>>>(d, d + 1, reinterpret_cast<unsigned *>(d + 2));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d, sizeof init, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e == hipSuccess) out[2] &= 0xffffffffull;
    return (int)e;
}

// ---------------------------------------------------------------------------
// The short forms of ray generation (sqrt_tame, tame_rcp / div_tame) and the packed UNORM8 conversion (unorm8_into)
// against the forms they stand in for.
// ---------------------------------------------------------------------------
namespace {

__device__ unsigned mix32(unsigned v)
{
    v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
    return v;
}

// thread i checks the 2^12 arguments 2^-80 + (i << 12 | lo) ulps: every float of [2^-80, 2^80)
__global__ __launch_bounds__(256) void k_check_sqrt_tame(unsigned long long *bad, unsigned *first_bad)
{
    const unsigned base = 0x17800000u + ((blockIdx.x * 256u + threadIdx.x) << 12);  // 2^-80 = 0x17800000, 2^80 = 0x67800000
    unsigned n = 0;
    for (unsigned lo = 0; lo < 4096u; ++lo) {
        const float x = __uint_as_float(base + lo);
        if (__float_as_uint(vrt::sqrt_tame(x)) != __float_as_uint(sqrtf(x))) { if (n++ == 0) atomicMin(first_bad, base + lo); }
    }
    if (n) atomicAdd(bad, (unsigned long long)n);
}

// thread i checks 2^12 pairs (a, b): b of either sign with 2^-40 <= |b| < 2^40 and any mantissa; a = b scaled by a factor
// that is spread over (0, 1.0001] in the logarithm -- half of the pairs within a factor of two of |b|, where the
// quotient's rounding is tightest -- with mantissa bits of its own, of either sign, and not below 2^-40
__global__ __launch_bounds__(256) void k_check_div_tame(unsigned seed, unsigned long long *bad, unsigned long long *tested,
                                                         unsigned long long *first_bad)
{
    const unsigned tid = blockIdx.x * 256u + threadIdx.x;
    unsigned n = 0, m = 0;
    for (unsigned j = 0; j < 4096u; ++j) {
        const unsigned h0 = mix32(seed ^ (tid * 4096u + j)), h1 = mix32(h0 ^ 0x9e3779b9u), h2 = mix32(h1 + 0x85ebca6bu);
        const unsigned eb = 127u - 40u + (h0 >> 8) % 80u;                       // exponent field of b
        const float    b = __uint_as_float((h0 << 31) | (eb << 23) | (h1 & 0x7fffffu));
        const unsigned down = (h2 & 1u) ? (h2 >> 1) % 2u : (h2 >> 1) % 82u;     // a's exponent below b's
        unsigned       ea = eb - down;
        float          a = __uint_as_float((h1 & 0x80000000u) | (ea << 23) | (h2 >> 9));
        if (fabsf(a) > 1.0001f * fabsf(b)) a = __uint_as_float((__float_as_uint(a) & 0x80000000u) | (__float_as_uint(b) & 0x7fffffffu));  // |a| = |b|
        if (fabsf(a) < 0x1.0p-40f) continue;
        ++m;
        const float fast = vrt::div_tame(a, b, vrt::tame_rcp(b)), full = a / b;
        if (__float_as_uint(fast) != __float_as_uint(full)) {
            if (n++ == 0) atomicMin(first_bad, ((unsigned long long)__float_as_uint(a) << 32) | __float_as_uint(b));
        }
    }
    if (n) atomicAdd(bad, (unsigned long long)n);
    atomicAdd(tested, (unsigned long long)m);
}

// thread i checks the 2^12 floats whose upper 20 bits are i
__global__ __launch_bounds__(256) void k_check_unorm8(unsigned long long *bad, unsigned *first_bad)
{
    const unsigned hi = blockIdx.x * 256u + threadIdx.x;
    unsigned n = 0;
    for (unsigned lo = 0; lo < 4096u; ++lo) {
        const unsigned bits = (hi << 12) | lo;
        const float    v = __uint_as_float(bits);
        const unsigned want = vrt::unorm8(v);
        const unsigned at = bits & 3u;
        const unsigned got = vrt::unorm8_into(v, at, 0xa5c3e17du);
        if (got != ((0xa5c3e17du & ~(0xffu << (8u * at))) | (want << (8u * at)))) { if (n++ == 0) atomicMin(first_bad, bits); }
    }
    if (n) atomicAdd(bad, (unsigned long long)n);
}

// thread i checks the 2^12 floats whose upper 20 bits are i
__global__ __launch_bounds__(256) void k_check_floor_convert(unsigned long long *bad, unsigned *first_bad)
{
    const unsigned hi = blockIdx.x * 256u + threadIdx.x;
    unsigned n = 0;
    for (unsigned lo = 0; lo < 4096u; ++lo) {
        const unsigned bits = (hi << 12) | lo;
        const float    x = __uint_as_float(bits);
        if (!(fabsf(x) < 0x1.0p31f)) continue;
        if (vrt::f2i_floor(x) != (int)floorf(x)) { if (n++ == 0) atomicMin(first_bad, bits); }
    }
    if (n) atomicAdd(bad, (unsigned long long)n);
}

}  // namespace

// out[0] = floats x of magnitude below 2^31 (all of them are tried) with f2i_floor(x) != (int)floorf(x); out[1] = the
// smallest offending bit pattern (0xffffffff: none).  Returns 0, or the HIP error code.
extern "C" int vrt_selftest_floor_convert(int device, unsigned long long out[2])
{
    if (!out) return (int)hipErrorInvalidValue;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return (int)e;
    unsigned long long *d = nullptr;
    e = hipMalloc(&d, 2 * sizeof(unsigned long long));
    if (e != hipSuccess) return (int)e;
    const unsigned long long init[2] = {0, 0xffffffffull};
    e = hipMemcpy(d, init, sizeof init, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_check_floor_convert<<<4096, 256>>>(d, reinterpret_cast<unsigned *>(d + 1));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d, sizeof init, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e == hipSuccess) out[1] &= 0xffffffffull;
    return (int)e;
}

// out[0] = floats x of [2^-80, 2^80) with sqrt_tame(x) != sqrtf(x) (all of them are tried); out[1] = pairs with
// div_tame(a, b, tame_rcp(b)) != a / b among out[2] pseudo-random pairs inside the short form's range (about 3 * 10^9);
// out[3] = the first offending x, or a << 32 | b (all ones: none).  Returns 0, or the HIP error code.
extern "C" int vrt_selftest_normalize(int device, unsigned long long out[4])
{
    if (!out) return (int)hipErrorInvalidValue;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return (int)e;
    unsigned long long *d = nullptr;
    e = hipMalloc(&d, 5 * sizeof(unsigned long long));
    if (e != hipSuccess) return (int)e;
    const unsigned long long init[5] = {0, 0, 0, ~0ull, 0xffffffffull};
    e = hipMemcpy(d, init, sizeof init, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_check_sqrt_tame<<<(0x50000000u >> 12) / 256u, 256>>>(d, reinterpret_cast<unsigned *>(d + 4));
        k_check_div_tame<<<4096, 256>>>(0x5eed1234u, d + 1, d + 2, d + 3);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    unsigned long long h[5] = {0, 0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    out[0] = h[0]; out[1] = h[1]; out[2] = h[2];
    out[3] = h[0] ? (h[4] & 0xffffffffull) : h[3];
    return (int)e;
}

// out[0] = floats v (all 2^32 bit patterns) for which unorm8_into(v, at, word) differs from unorm8(v) put into byte
// `at` of `word`; out[1] = the smallest offending bit pattern (0xffffffff: none).  Returns 0, or the HIP error code.
extern "C" int vrt_selftest_unorm8(int device, unsigned long long out[2])
{
    if (!out) return (int)hipErrorInvalidValue;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return (int)e;
    unsigned long long *d = nullptr;
    e = hipMalloc(&d, 2 * sizeof(unsigned long long));
    if (e != hipSuccess) return (int)e;
    const unsigned long long init[2] = {0, 0xffffffffull};
    e = hipMemcpy(d, init, sizeof init, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_check_unorm8<<<4096, 256>>>(d, reinterpret_cast<unsigned *>(d + 1));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d, sizeof init, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e == hipSuccess) out[1] &= 0xffffffffull;
    return (int)e;
}

// ---------------------------------------------------------------------------
// VALU issue rate of the chip: what bounds a kernel that is neither memory nor matrix bound.
// ---------------------------------------------------------------------------
namespace {

constexpr int kRateUnroll = 64;  // instructions per loop iteration (16 independent chains x 4)

// KIND 0: v_fma_f32   1: v_add_u32   2: the trace kernels' mix (fma, integer add, compare + select, min/max, float->int)
template <int KIND>
__global__ __launch_bounds__(256) void k_valu_rate(float *sink, unsigned iters, long long *clocks)
{
    float    a[16];
    unsigned u[16];
    for (int i = 0; i < 16; ++i) { a[i] = (float)(threadIdx.x + i) * 0x1.0p-10f; u[i] = threadIdx.x * 16u + i; }
    const float    x = 1.0f + (float)blockIdx.x * 0x1.0p-20f, y = 0x1.0p-12f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (unsigned it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < kRateUnroll / 16; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
                } else if (KIND == 1) {
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                } else {
                    switch ((i + 4 * r) & 7) {
                        case 0: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y)); break;
                        case 1: asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15])); break;
                        case 2: asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc"); break;
                        case 3: asm volatile("v_min_i32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 3) & 15])); break;
                        case 4: asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i])); break;
                        case 5: asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x)); break;
                        case 6: asm volatile("v_floor_f32 %0, %0" : "+v"(a[i])); break;
                        default: asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 5) & 15])); break;
                    }
                }
            }
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.0f;
    for (int i = 0; i < 16; ++i) s += a[i] + (float)u[i];
    if (s == 123.456f) sink[0] = s;  // keeps the chains alive; practically never taken
    if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = w1 - w0; }
}

}  // namespace

// Chip-wide rate of wave64 vector instructions with `waves_per_simd` (1..8) wavefronts resident on every
// SIMD of every compute unit, each issuing a long stream of independent instructions of `kind` (0: v_fma_f32,
// 1: v_add_u32, 2: a mix like the trace kernels').  out[0] = wave instructions per second, out[1] = shader
// clock in GHz inside the kernel (shader ticks over 100 MHz wall ticks of the first wavefront's loop), out[2] = shader
// cycles per wave instruction per SIMD at that clock (clock x SIMDs / rate), out[3] = kernel milliseconds (HIP events).  The mix counts its
// compare + select pair as two instructions.  Returns 0, or the HIP error code.
extern "C" int vrt_selftest_valu_rate(int device, int kind, int waves_per_simd, double out[4])
{
    if (kind < 0 || kind > 2 || waves_per_simd < 1 || waves_per_simd > 8 || !out) return (int)hipErrorInvalidValue;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return (int)e;
    const unsigned cus = (unsigned)prop.multiProcessorCount, blocks = cus * (unsigned)waves_per_simd;  // 256 threads = one wave per SIMD
    const unsigned iters = 20000u / (unsigned)waves_per_simd + 1000u;
    float *sink = nullptr;
    long long *clocks = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    e = hipMalloc(&sink, 256);
    if (e == hipSuccess) e = hipMalloc(&clocks, 2 * sizeof(long long));
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float ms = 0.0f;
    long long h[2] = {0, 0};
    for (int rep = 0; rep < 2 && e == hipSuccess; ++rep) {  // the first launch warms the clocks up
        e = hipEventRecord(e0, nullptr);
        if (e != hipSuccess) break;
        if (kind == 0) k_valu_rate<0><<<blocks, 256>>>(sink, iters, clocks);
        else if (kind == 1) k_valu_rate<1><<<blocks, 256>>>(sink, iters, clocks);
        else k_valu_rate<2><<<blocks, 256>>>(sink, iters, clocks);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    }
    if (e == hipSuccess) e = hipMemcpy(h, clocks, sizeof h, hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
        // the mix's compare + select case is two instructions: 1 case in 8
        const double per_iter = kind == 2 ? kRateUnroll * 9.0 / 8.0 : (double)kRateUnroll;
        const double instr = (double)blocks * 4.0 * (double)iters * per_iter;
        const double ghz = h[1] > 0 ? (double)h[0] / ((double)h[1] * 10.0) : 0.0;  // 100 MHz wall ticks
        out[0] = instr / ((double)ms * 1e-3);
        out[1] = ghz;
        out[2] = out[0] > 0.0 ? ghz * 1e9 * (double)cus * 4.0 / out[0] : 0.0;  // cycles the chip's SIMDs spend per instruction
        out[3] = (double)ms;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    (void)hipFree(clocks);
    return (int)e;
}

// ---------------------------------------------------------------------------
// Layout of the kernel-argument segment.  The trace kernels hand their seldom used code a pointer
// to the segment instead of a copy of the launch parameters (degenerate_pixel, degenerate_primary,
// degenerate_ray in vrt_trace.hip), on the assumption that a by-value TraceParams that is the kernel's
// first parameter sits at offset 0 of the segment, with the struct's own layout.
// ---------------------------------------------------------------------------
namespace {

__device__ unsigned count_kernarg_mismatches(const void *segment, const unsigned *expect, unsigned words)
{
    const unsigned *seg = static_cast<const unsigned *>(segment);
    unsigned bad = 0;
    for (unsigned i = threadIdx.x; i < words; i += blockDim.x) bad += seg[i] != expect[i] ? 1u : 0u;
    return bad;
}

// the shape of k_trace_skip and k_trace_literal: the parameter block alone
__global__ __launch_bounds__(64) void k_kernarg_alone(const vrt::TraceParams P, const unsigned *expect, unsigned *bad)
{
    const unsigned n = count_kernarg_mismatches((const void *)__builtin_amdgcn_kernarg_segment_ptr(), expect, sizeof(vrt::TraceParams) / 4);
    if (n) atomicAdd(bad, n);
    if (threadIdx.x == 0 && P.w != expect[16]) atomicAdd(bad, 1u << 16);  // m[16] comes first, then w
}

// the shape of k_trace_secondary: three pointers follow the block
__global__ __launch_bounds__(64) void k_kernarg_with_tail(const vrt::TraceParams P, uint8_t *a, uint8_t *b, uint32_t *c,
                                                          const unsigned *expect, unsigned *bad)
{
    const unsigned n = count_kernarg_mismatches((const void *)__builtin_amdgcn_kernarg_segment_ptr(), expect, sizeof(vrt::TraceParams) / 4);
    if (n) atomicAdd(bad, n);
    if (threadIdx.x == 0 && (P.w != expect[16] || a != nullptr || b != nullptr || c != nullptr)) atomicAdd(bad, 1u << 16);
}

}  // namespace

// out[0], out[1] = 32-bit words of the kernel-argument segment that differ from the TraceParams the host
// passed by value, for the two kernel shapes above (0 = the layout holds); out[2] = sizeof(TraceParams).
// Returns 0, or the HIP error code.
extern "C" int vrt_selftest_kernarg(int device, unsigned out[3])
{
    static_assert(sizeof(vrt::TraceParams) % 4 == 0, "TraceParams is compared word by word");
    if (!out) return (int)hipErrorInvalidValue;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return (int)e;
    constexpr unsigned words = sizeof(vrt::TraceParams) / 4;
    vrt::TraceParams p;
    unsigned pattern[words];
    for (unsigned i = 0; i < words; ++i) pattern[i] = 0x9e3779b9u * (i + 1u);  // every word distinct, padding included
    static_assert(sizeof pattern == sizeof p, "pattern covers the block");
    __builtin_memcpy(&p, pattern, sizeof p);
    unsigned *d = nullptr;
    e = hipMalloc(&d, sizeof pattern + 2 * sizeof(unsigned));
    if (e != hipSuccess) return (int)e;
    unsigned *bad = d + words;
    e = hipMemcpy(d, pattern, sizeof pattern, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(bad, 0, 2 * sizeof(unsigned));
    if (e == hipSuccess) {
        k_kernarg_alone<<<1, 64>>>(p, d, bad);
        k_kernarg_with_tail<<<1, 64>>>(p, nullptr, nullptr, nullptr, d, bad + 1);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, bad, 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
    out[2] = (unsigned)sizeof(vrt::TraceParams);
    (void)hipFree(d);
    return (int)e;
}
