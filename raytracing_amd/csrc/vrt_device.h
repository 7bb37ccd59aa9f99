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

// Canonical sine of the shading term (raytrace.glsl:130): double-precision reduction
// against pi = HI + LO (HI has 33 significant bits, so q*HI is exact for |q| < 2^20),
// odd Taylor polynomial through r^17, one rounding to float.
VRT_DEV float canon_sin(float x)
{
    const double INV_PI = 0x1.45f306dc9c883p-2;
    const double PI_HI  = 0x1.921fb54400000p+1;
    const double PI_LO  = 0x1.0b4611a626331p-33;
    double xd = (double)x;
    if (!(fabs(xd) < 1048576.0)) return __builtin_nanf("");
    double q  = rint(xd * INV_PI);
    double r  = (xd - q * PI_HI) - q * PI_LO;
    double r2 = r * r;
    double p  = 0x1.952c77030ad4ap-49;
    p = p * r2 + -0x1.ae7f3e733b81fp-41;
    p = p * r2 + 0x1.6124613a86d09p-33;
    p = p * r2 + -0x1.ae64567f544e4p-26;
    p = p * r2 + 0x1.71de3a556c734p-19;
    p = p * r2 + -0x1.a01a01a01a01ap-13;
    p = p * r2 + 0x1.1111111111111p-7;
    p = p * r2 + -0x1.5555555555555p-3;
    double s = r + (r * r2) * p;
    long long qi = (long long)q;
    if (qi & 1) s = -s;
    return (float)s;
}

// B8G8R8A8_UNORM store: clamp to [0,1] (NaN -> 0), round half up.
VRT_DEV unsigned unorm8(float v)
{
    v = (v > 0.0f) ? v : 0.0f;
    v = (v < 1.0f) ? v : 1.0f;
    return (unsigned)f2i(floorf(v * 255.0f + 0.5f));
}

}  // namespace vrt
