// vrt_selftest.hip -- exhaustive checks of device arithmetic that the kernels rely on, in a small library
// of its own (libvrt_selftest.so) that only the tests load.
#include "vrt_device.h"

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
