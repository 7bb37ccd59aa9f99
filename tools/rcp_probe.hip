// Which short forms of 1 / x have the bits of the division for every float the traversal can meet (|x| <= 2, |1/x| < 2^100)?
// Counts, over all such bit patterns, the arguments for which each candidate differs from 1.0f / x.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/rcp_probe tools/rcp_probe.hip && tools/rcp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ float cand(int which, float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    if (which == 0) return r;
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);                       // one Newton step
    if (which == 1) return r;
    float rem = __builtin_fmaf(-x, r, 1.0f);
    float q = __builtin_fmaf(rem, r, r);               // + one correction
    if (which == 2) return q;
    rem = __builtin_fmaf(-x, q, 1.0f);
    return __builtin_fmaf(rem, r, q);                  // + a second one (the form in use)
}

__global__ void k(unsigned long long *bad)
{
    const unsigned stride = gridDim.x * blockDim.x;
    unsigned long long local[4] = {0, 0, 0, 0};
    for (unsigned long long i = blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const unsigned u = (unsigned)i;
        float x;
        memcpy(&x, &u, 4);
        if (!(fabsf(x) <= 2.0f) || x == 0.0f) continue;
        const float want = 1.0f / x;
        if (!(fabsf(want) < 0x1.0p100f)) continue;
        unsigned w;
        memcpy(&w, &want, 4);
        for (int c = 0; c < 4; ++c) {
            const float g = cand(c, x);
            unsigned gb;
            memcpy(&gb, &g, 4);
            if (gb != w) local[c] += 1;
        }
    }
    for (int c = 0; c < 4; ++c) if (local[c]) atomicAdd(&bad[c], local[c]);
}

int main()
{
    unsigned long long *d, h[4];
    hipMalloc(&d, sizeof h);
    hipMemset(d, 0, sizeof h);
    k<<<4096, 256>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("arguments whose result differs from 1.0f / x: rcp alone %llu, + Newton step %llu, + one correction %llu, + two corrections %llu\n",
           h[0], h[1], h[2], h[3]);
    return 0;
}
