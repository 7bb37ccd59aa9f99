// Issue cost of single gfx950 vector instructions, measured: every SIMD of the chip runs `waves` wavefronts, each a long
// stream of one instruction in 16 independent register chains; the figure printed per instruction is shader cycles per
// wave64 instruction per SIMD (clock x SIMDs / chip-wide rate).  An instruction count is a poor cost model when the
// instructions differ by a factor of four: this table is what the trace kernels' loops are priced with.
//   hipcc -O2 --offload-arch=gfx950 -o tools/op_rates tools/op_rates.hip && tools/op_rates [waves per SIMD, default 8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Op { const char *name; };

template <int OP>
__device__ __forceinline__ void one(float &a, unsigned &u, unsigned long long &q, float x, float y, unsigned k)
{
    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
    else if (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(x));
    else if (OP == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(x));
    else if (OP == 3) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a) : "v"(x));
    else if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(k));
    else if (OP == 5) asm volatile("v_min_i32 %0, %0, %1" : "+v"(u) : "v"(k));
    else if (OP == 6) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u) : "v"(k));
    else if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u) : "v"(k) : "vcc");
    else if (OP == 8) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a), "v"(x) : "vcc");
    else if (OP == 9) asm volatile("v_cmp_lt_i32 s[20:21], %0, %1" : : "v"(u), "v"(k) : "s20", "s21");
    else if (OP == 10) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u) : "v"(a));
    else if (OP == 11) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a) : "v"(u));
    else if (OP == 12) asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(u) : "v"(a));
    else if (OP == 13) asm volatile("v_floor_f32 %0, %0" : "+v"(a));
    else if (OP == 14) asm volatile("v_fract_f32 %0, %0" : "+v"(a));
    else if (OP == 15) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u) : "v"(k));
    else if (OP == 16) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u) : "v"(k));
    else if (OP == 17) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u) : "v"(k));
    else if (OP == 18) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(q) : "v"(k));
    else if (OP == 19) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u) : "v"(k));
    else if (OP == 20) asm volatile("v_bfe_u32 %0, %0, 2, 5" : "+v"(u));
    else if (OP == 21) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(u) : "v"(k));
    else if (OP == 22) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(u) : "v"(k));
    else if (OP == 23) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u) : "v"(k));
    else if (OP == 24) asm volatile("v_cmp_eq_u64 vcc, %0, %0" : : "v"(q) : "vcc");
    else if (OP == 25) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
    else if (OP == 26) asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(u) : "v"(k));
    else if (OP == 27) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(u) : : "vcc");
    else if (OP == 28) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(q));
    else if (OP == 29) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(q));
    else if (OP == 30) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x60" : "+v"(u) : "v"(k));
    else if (OP == 31) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u) : "v"(a));
    else if (OP == 32) asm volatile("v_rndne_f32 %0, %0" : "+v"(a));
    else if (OP == 33) asm volatile("v_mov_b32 %0, %1" : "=v"(u) : "v"(k));
    else if (OP == 34) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u));
    else if (OP == 35) asm volatile("v_max_i32 %0, %0, %1" : "+v"(u) : "v"(k));
    else if (OP == 36) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(x));
    else if (OP == 37) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(x));
    else if (OP == 38) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(a), "v"(x) : "s20", "s21");
    else if (OP == 39) asm volatile("v_cndmask_b32 %0, %0, %1, s[22:23]" : "+v"(u) : "v"(k));
    else if (OP == 40) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a));
    else if (OP == 41) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(u) : "v"(k));
    else if (OP == 42) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(u) : "v"(k));
    else if (OP == 43) asm volatile("v_sad_u32 %0, %0, %1, %1" : "+v"(u) : "v"(k));
    else if (OP == 44) asm volatile("v_fma_f32 %0, |%0|, %1, |%2|" : "+v"(a) : "v"(x), "v"(y));
    else if (OP == 45) asm volatile("v_mad_u32_u24 %0, %0, 2, %1" : "+v"(u) : "v"(k));
    else if (OP == 46) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u) : "v"(k));
    else if (OP == 47) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
    else if (OP == 48) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u) : "v"(k));
    else if (OP == 49) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u) : "v"(k));
    else if (OP == 50) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u) : "v"(k));
    else if (OP == 51) asm volatile("v_or_b32 %0, 0x0c0c0100, %0" : "+v"(u));
    else if (OP == 52) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(u) : "v"(k));
    else if (OP == 53) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(u));
    else if (OP == 54) asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(u) : "v"(k));
    else if (OP == 55) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(u) : "v"(k));
    else if (OP == 56) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
    else if (OP == 57) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
    else if (OP == 58) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(u) : "v"(a), "v"(x), "v"(k) : "vcc");
    else if (OP == 59) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u) : "v"(k));
    else if (OP == 60) asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(u) : "v"(k));
    else if (OP == 61) asm volatile("v_not_b32 %0, %0" : "+v"(u));
    else if (OP == 62) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(u) : "v"(k));
    else if (OP == 63) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a) : "v"(u));
    else if (OP == 64) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(u) : "v"(k));
    else if (OP == 65) asm volatile("v_add_f32 %0, s24, %0" : "+v"(a));
    else if (OP == 66) asm volatile("v_add_u32 %0, 1, %0" : "+v"(u));
    else if (OP == 67) asm volatile("v_add_f32 %0, |%0|, %1" : "+v"(a) : "v"(x));
    else if (OP == 68) asm volatile("v_mul_f32 %0, 0x3f800100, %0" : "+v"(a));
    else if (OP == 69) asm volatile("v_fma_f32 %0, %0, s24, 1.0" : "+v"(a));
    else if (OP == 70) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
    else if (OP == 71) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(u) : "v"(k));
    else if (OP == 72) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(u) : "v"(k));
    else if (OP == 73) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(u) : "s20");
    else if (OP == 74) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u));
    else if (OP == 75) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(u) : "v"(k) : "vcc");
    else if (OP == 76) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
}

static const char *NAMES[] = {
    "v_fma_f32", "v_add_f32", "v_mul_f32", "v_min_f32", "v_add_u32", "v_min_i32", "v_xor_b32", "v_cndmask_b32 (vcc)",
    "v_cmp_lt_f32 -> vcc", "v_cmp_lt_i32 -> sgpr pair", "v_cvt_i32_f32", "v_cvt_f32_i32", "v_cvt_flr_i32_f32", "v_floor_f32",
    "v_fract_f32", "v_mul_lo_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_lshrrev_b64", "v_perm_b32", "v_bfe_u32", "v_xad_u32",
    "v_lshl_or_b32", "v_add3_u32", "v_cmp_eq_u64 -> vcc", "v_rcp_f32", "v_med3_i32", "v_addc_co_u32", "v_pk_fma_f32 (2 fma)",
    "v_pk_add_f32 (2 add)", "v_bitop3_b32", "v_cvt_pk_u8_f32", "v_rndne_f32", "v_mov_b32", "v_lshlrev_b32", "v_max_i32",
    "v_sub_f32", "v_max_f32", "v_cmp_lt_f32 -> sgpr pair", "v_cndmask_b32 (sgpr pair)", "v_sqrt_f32", "v_lshrrev_b32",
    "v_and_or_b32", "v_sad_u32", "v_fma_f32 with |.| modifiers", "v_mad_u32_u24 (inline const)", "v_mul_hi_u32", "v_fmac_f32",
    "v_sub_u32", "v_and_b32", "v_or_b32", "v_or_b32 (literal)", "v_lshlrev_b32 (vgpr amount)", "v_lshrrev_b32 (inline amount)", "v_ashrrev_i32", "v_or3_b32", "v_min3_f32", "v_max3_f32", "v_cmp_lt_f32 vcc + v_cndmask vcc (2)", "v_lshl_add_u32", "v_add_lshl_u32", "v_not_b32", "v_bfi_b32", "v_cvt_f32_u32", "v_subrev_u32", "v_add_f32 (sgpr operand)", "v_add_u32 (inline const)", "v_add_f32 e64 with |.|", "v_mul_f32 (literal)", "v_fma_f32 (sgpr + literal)", "v_med3_f32", "v_bcnt_u32_b32", "v_mbcnt_lo_u32_b32", "v_readfirstlane_b32", "v_mov_b32 dpp quad_perm", "v_add_co_u32", "v_exp_f32"};
constexpr int N_OPS = sizeof(NAMES) / sizeof(NAMES[0]);

template <int OP>
__global__ __launch_bounds__(256) void k_rate(float *sink, unsigned iters, long long *clocks)
{
    float              a[16];
    unsigned           u[16];
    unsigned long long q[16];
    for (int i = 0; i < 16; ++i) {
        a[i] = 1.0f + (float)(threadIdx.x + i) * 0x1.0p-10f; u[i] = threadIdx.x * 16u + i;
        q[i] = ((unsigned long long)u[i] << 32) | (unsigned)__float_as_uint(a[i]);
    }
    const float    x = 1.0f + (float)blockIdx.x * 0x1.0p-20f, y = 0x1.0p-12f;
    const unsigned k = threadIdx.x & 7u;
    asm volatile("s_mov_b64 s[22:23], 0x5555\n\ts_mov_b32 s24, 0x3f800010" : : : "s22", "s23", "s24");
    const long long c0 = clock64(), w0 = wall_clock64();
    for (unsigned it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) one<OP>(a[i], u[i], q[i], x, y, k);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.0f;
    for (int i = 0; i < 16; ++i) s += a[i] + (float)u[i] + (float)q[i];
    if (s == 123.456f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = w1 - w0; }
}

using Kernel = void (*)(float *, unsigned, long long *);
template <int... I>
static void fill(Kernel *t, std::integer_sequence<int, I...>) { ((t[I] = k_rate<I>), ...); }

int main(int argc, char **argv)
{
    const int waves = argc > 1 ? atoi(argv[1]) : 8;
    if (waves < 1 || waves > 8) { fprintf(stderr, "waves per SIMD: 1..8\n"); return 2; }
    Kernel table[N_OPS];
    fill(table, std::make_integer_sequence<int, N_OPS>());
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const unsigned cus = (unsigned)prop.multiProcessorCount, blocks = cus * (unsigned)waves;
    const unsigned iters = 6000u / (unsigned)waves + 500u;
    float *sink; long long *clocks;
    CHECK(hipMalloc(&sink, 256)); CHECK(hipMalloc(&clocks, 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%s, %u CUs, %d wavefronts per SIMD, %u x 64 instructions per wavefront\n", prop.gcnArchName, cus, waves, iters);
    printf("%-34s %10s %8s %8s\n", "instruction", "G instr/s", "GHz", "cycles");
    for (int op = 0; op < N_OPS; ++op) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0, nullptr));
            table[op]<<<blocks, 256>>>(sink, iters, clocks);
            CHECK(hipGetLastError());
            CHECK(hipEventRecord(e1, nullptr));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        long long h[2];
        CHECK(hipMemcpy(h, clocks, sizeof h, hipMemcpyDeviceToHost));
        const double instr = (double)blocks * 4.0 * iters * 64.0, rate = instr / (ms * 1e-3);
        const double ghz = (double)h[0] / ((double)h[1] * 10.0);
        printf("%-34s %10.1f %8.3f %8.2f\n", NAMES[op], rate / 1e9, ghz, ghz * 1e9 * cus * 4.0 / rate);
        fflush(stdout);
    }
    return 0;
}
