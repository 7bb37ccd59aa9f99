// vrt_device.h -- scalar semantics shared by every device kernel.
//
// The trace kernels must reproduce, bit for bit, a canonical IEEE-754 binary32
// evaluation of the reference shaders (raytrace.glsl, util.glsl, render.comp,
// noise.glsl).  Everything whose result could depend on the compiler lives here,
// and the library is built with -ffp-contract=off so that no multiply-add is fused.
// Division and sqrtf are correctly rounded (hipcc default for HIP).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define VRT_DEV __device__ __forceinline__

namespace vrt {

constexpr int kChunk = 32;  // raytrace.glsl:10, terrain_gen.comp:12

// float -> int: saturating, NaN -> 0.  This is exactly what the hardware conversion
// v_cvt_i32_f32 does (round toward zero, out-of-range values saturate, NaN gives 0); the
// instruction is issued directly so that the optimiser never sees an out-of-range fptosi
// (undefined in C++) and no compare/branch cascade is needed.
VRT_DEV int f2i(float x)
{
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// float -> int for values known to be finite and far inside the int range.
VRT_DEV int f2i_small(float x) { return (int)x; }

// int(floor(x)) in one instruction, for values known to be finite and far inside the int range (the hardware conversion
// that rounds toward minus infinity; vrt_selftest_floor_convert compares it with the two-step form for every float
// of magnitude below 2^31).
VRT_DEV int f2i_floor(float x)
{
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// |a| * m + |b| as one fused multiply-add with the magnitudes taken by the instruction's own operand modifiers (written
// out because the compiler pairs two such operations into a packed instruction, which has no magnitude modifier, and
// pays four extra instructions for the pair; the factor in a vector register: an operand from a scalar register halves
// the instruction's issue rate on this chip).  The result only steers which of two exact paths a lane takes.
VRT_DEV float fma_abs(float a, float m, float b)
{
    float r;
    asm("v_fma_f32 %0, |%1|, %2, |%3|" : "=v"(r) : "v"(a), "v"(m), "v"(b));
    return r;
}

// min(max(v, lo), hi) for lo <= hi, as the one instruction the hardware has for it (the median of the three; the compiler
// forms it only when it can see that lo <= hi).
VRT_DEV int clamp_ordered(int v, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}

// The same value, as far as the optimiser can tell a new one: what is computed from it is computed again
// instead of being kept in registers from an earlier computation.
VRT_DEV unsigned opaque(unsigned v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// wrapping integer arithmetic
VRT_DEV int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
VRT_DEV int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
VRT_DEV int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }

// util.glsl:1-4
VRT_DEV int rem_euclid(int v, int n)
{
    int r = v % n;
    return r >= 0 ? r : r + n;
}

// util.glsl:28-42: strict '<', ties resolve x, y, z; NaN in .x wins.
VRT_DEV int argmin3(float a, float b, float c)
{
    int   idx = 0;
    float mv  = a;
    if (b < mv) { mv = b; idx = 1; }
    if (c < mv) { idx = 2; }
    return idx;
}

// 1 / x for the direction components of an ordinary ray: the hardware's reciprocal estimate and one Newton step on it,
// two fused multiply-adds -- the head of the compiler's own division sequence, without the two quotient corrections
// that follow it there and without the operand scaling and the special-case fix-up around it.  For every x with
// |x| <= 2 and |1 / x| < 2^100 the result has the bits of 1.0f / x all the same (the corrections change nothing when the
// numerator is 1, and the wrappers only act on operands whose quotient is zero, denormal, infinite or NaN), and for
// every other such x (and NaN) it is not below 2^100 in magnitude either -- which is all the traversal asks
// (ray_is_regular; a component of a normalised direction never exceeds 1).  vrt_selftest_recip checks both statements
// for every such argument; tools/rcp_probe.hip counts, for each shorter and longer form, the arguments that differ.
VRT_DEV float recip_regular(float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

// sqrtf(x) and a / b for operands far from the edges of the float range, as ray generation has them for every
// ordinary camera: the compiler's own correctly rounded sequences (hardware estimate; for the root two residuals
// and a step of one ulp either way, for the quotient a Newton step on the reciprocal and two corrections of the
// quotient, all fused multiply-adds) without the steps around them that rescale operands near the ends of the
// exponent range and patch up zeros, infinities, NaNs and denormal results.  Those steps do nothing -- the scaling
// instructions return their operand, the fix-up returns the quotient -- when
//     2^-80 <= x <= 2^80                                         (sqrt_tame)
//     2^-40 <= |b| <= 2^40  and  2^-50 <= |a| <= 1.0001 |b|      (div_tame)
// so the bits are those of sqrtf(x) and a / b there (vrt_selftest_normalize: every x of the range; 3 * 10^9 pairs).  The
// reciprocal of b and its Newton step depend on b alone: tame_rcp once per denominator, div_tame per numerator.
VRT_DEV float sqrt_tame(float x)
{
    const float y  = __builtin_amdgcn_sqrtf(x);
    const float ym = __uint_as_float(__float_as_uint(y) - 1u), yp = __uint_as_float(__float_as_uint(y) + 1u);
    const float em = __builtin_fmaf(-ym, y, x), ep = __builtin_fmaf(-yp, y, x);
    float r = em <= 0.0f ? ym : y;
    r = ep > 0.0f ? yp : r;
    return r;
}

VRT_DEV float tame_rcp(float b)
{
    const float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

VRT_DEV float div_tame(float a, float b, float rcp_b)
{
    const float q0 = a * rcp_b;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, rcp_b, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, rcp_b, q1);
}

// Canonical sine of the shading term (raytrace.glsl:130), binary32 throughout: k = rint(x/pi),
// reduction against pi = P1 + P2 + P3 with P1 = 3.125 (k*P1 and x - k*P1 are exact for
// |k| < 2^19), odd Taylor polynomial through r^11.  |error| < 3e-7 for |x| < 2^20.
VRT_DEV float canon_sin(float x)
{
    const float INV_PI = 0x1.45f306p-2f;
    const float P1 = 3.125f, P2 = 0x1.1p-6f, P3 = -0x1.2aeef4p-17f;
    if (!(fabsf(x) < 1048576.0f)) return __builtin_nanf("");
    const float k  = rintf(x * INV_PI);
    const float r  = ((x - k * P1) - k * P2) - k * P3;
    const float r2 = r * r;
    float p = -0x1.ae6456p-26f;
    p = p * r2 + 0x1.71de3ap-19f;
    p = p * r2 + -0x1.a01a02p-13f;
    p = p * r2 + 0x1.111112p-7f;
    p = p * r2 + -0x1.555556p-3f;
    float s = r + (r * r2) * p;
    if ((int)k & 1) s = -s;
    return s;
}

// B8G8R8A8_UNORM store: clamp to [0,1] (NaN -> 0), round half up.
VRT_DEV unsigned unorm8(float v)
{
    v = (v > 0.0f) ? v : 0.0f;
    v = (v < 1.0f) ? v : 1.0f;
    return (unsigned)f2i(floorf(v * 255.0f + 0.5f));
}

// The same byte, put into byte `at` (0..3) of `word`: the hardware's saturating float -> u8 conversion (NaN -> 0,
// below 0 -> 0, above 255 -> 255) stands in for the clamp and the conversion, and packs as well.  It rounds to nearest,
// so the floor stays (its result converts exactly); v * 255 + 0.5 is evaluated as above, and where the clamp would have
// acted the floored sum lies beyond [0, 255] on the same side and saturates to the same byte.  vrt_selftest_unorm8
// compares the two for every float.
VRT_DEV unsigned unorm8_into(float v, unsigned at, unsigned word)
{
    return __builtin_amdgcn_cvt_pk_u8_f32(floorf(v * 255.0f + 0.5f), at, word);
}

}  // namespace vrt
