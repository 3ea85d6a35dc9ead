// issue_ubench.hip -- per-SIMD issue cost of the instruction classes the scan kernels are made of, at a KNOWN number of
// waves per SIMD (round 3; replaces the wave axis of valu_ubench.hip, whose "k blocks of 256 threads per CU" were not
// co-resident as labelled -- VERDICT r2 weak #6).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/issue_ubench tools/ubench/issue_ubench.hip
//   ./issue_ubench            -> one JSON line per (test, waves per SIMD)
//
// Occupancy is fixed by construction: ONE workgroup of 256 * w threads per CU (grid = number of CUs, 16 KiB of LDS per
// wave-quad so that a second workgroup of a neighbouring dispatch cannot join), i.e. exactly w waves on each of the four
// SIMDs, started together behind a barrier.  Every wave runs ITER x 32 copies of one instruction (8 independent
// destination registers); reported:
//   cyc_wave = s_memtime cycles per instruction as ONE wave sees them (its issue interval at that occupancy)
//   cyc_simd = cyc_wave / w = SIMD cycles per wave-instruction = the throughput cost to price a kernel's histogram with
//   us       = kernel time (cross-check: us * clock / (ITER * 32 * w))
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 1024;

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

#define DEFINE_TEST(NAME, ASM8)                                                                   \
__global__ void __launch_bounds__(1024) k_##NAME(float* out, long long* cyc, float seed) {        \
    extern __shared__ float lds[];                                                                \
    float r0 = seed + threadIdx.x, r1 = r0 * 0.5f, r2 = r0 * 0.25f, r3 = r0 + 1.f, r4 = r0 + 2.f, \
          r5 = r0 + 3.f, r6 = r0 + 4.f, r7 = r0 + 5.f;                                            \
    float s0 = 0.999f, s1 = 1e-3f;                                                                \
    unsigned la = (threadIdx.x * 8u) & 8191u;       /* b64: lane stride 2 dwords */               \
    unsigned la4 = (threadIdx.x * 4u) & 4095u;      /* b32 */                                     \
    unsigned bp = ((threadIdx.x & 48u) << 2) | 20u; /* ds_bpermute: lane 5 of the own DPP row */  \
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = r0;                             \
    __syncthreads();                                                                              \
    long long c0 = __builtin_readcyclecounter();                                                  \
    for (int it = 0; it < ITER; ++it) {                                                           \
        asm volatile(ASM8 ASM8 ASM8 ASM8                                                          \
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                     : "v"(s0), "v"(s1), "v"(la), "v"(la4), "v"(bp) : "memory", "s20", "s21", "s22", "s23", "vcc"); \
    }                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
    long long c1 = __builtin_readcyclecounter();                                                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;          \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = c1 - c0; \
}

// operands: %0..%7 = r0..r7, %8 = s0, %9 = s1, %10 = lds addr b64, %11 = lds addr b32, %12 = bpermute addr
#define R8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
DEFINE_TEST(fma, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
DEFINE_TEST(fmac, "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                  "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n")
DEFINE_TEST(mul, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                 "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
DEFINE_TEST(fma_chain, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"
                       "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n")
DEFINE_TEST(exp, "v_exp_f32 %0, %8\n v_exp_f32 %1, %8\n v_exp_f32 %2, %8\n v_exp_f32 %3, %8\n"
                 "v_exp_f32 %4, %8\n v_exp_f32 %5, %8\n v_exp_f32 %6, %8\n v_exp_f32 %7, %8\n")
DEFINE_TEST(rcp, "v_rcp_f32 %0, %8\n v_rcp_f32 %1, %8\n v_rcp_f32 %2, %8\n v_rcp_f32 %3, %8\n"
                 "v_rcp_f32 %4, %8\n v_rcp_f32 %5, %8\n v_rcp_f32 %6, %8\n v_rcp_f32 %7, %8\n")
DEFINE_TEST(exp1_fma7, "v_exp_f32 %0, %8\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                       "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
DEFINE_TEST(cndmask, "v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n"
                     "v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n")
DEFINE_TEST(fmac_dpp_shr, "v_fmac_f32_dpp %0, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                          "v_fmac_f32_dpp %2, %8, %9 row_shr:2 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %8, %9 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                          "v_fmac_f32_dpp %4, %8, %9 row_shr:4 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %8, %9 row_shr:4 row_mask:0xf bank_mask:0xf\n"
                          "v_fmac_f32_dpp %6, %8, %9 row_shr:8 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %8, %9 row_shr:8 row_mask:0xf bank_mask:0xf\n")
// the scan step as the kernels issue it: dependent fmac_dpp + mul_dpp + s_nop 0 on ONE register pair
DEFINE_TEST(mscan_step, "v_fmac_f32_dpp %1, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 0\n"
                        "v_fmac_f32_dpp %1, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 0\n"
                        "v_fmac_f32_dpp %1, %1, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 0\n"
                        "v_fmac_f32_dpp %1, %1, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n")
DEFINE_TEST(mov_dpp_ror, "v_mov_b32_dpp %0, %8 row_ror:15 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 row_ror:15 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %8 row_ror:15 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 row_ror:15 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %8 row_ror:15 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 row_ror:15 row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %6, %8 row_ror:15 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 row_ror:15 row_mask:0xf bank_mask:0xf\n")
DEFINE_TEST(permlane32_swap, "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                             "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n")
DEFINE_TEST(permlane16_swap, "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                             "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n")
DEFINE_TEST(swap_add, "v_permlane32_swap_b32 %0, %1\n v_add_f32 %0, %0, %1\n v_permlane32_swap_b32 %2, %3\n v_add_f32 %2, %2, %3\n"
                      "v_permlane16_swap_b32 %4, %5\n v_add_f32 %4, %4, %5\n v_permlane16_swap_b32 %6, %7\n v_add_f32 %6, %6, %7\n")
DEFINE_TEST(ds_bpermute, "ds_bpermute_b32 %0, %12, %8\n ds_bpermute_b32 %1, %12, %8\n ds_bpermute_b32 %2, %12, %8\n ds_bpermute_b32 %3, %12, %8\n"
                         "ds_bpermute_b32 %4, %12, %8\n ds_bpermute_b32 %5, %12, %8\n ds_bpermute_b32 %6, %12, %8\n ds_bpermute_b32 %7, %12, %8\n")
DEFINE_TEST(readlane, "v_readlane_b32 s20, %0, 5\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 5\n v_readlane_b32 s23, %3, 5\n"
                      "v_readlane_b32 s20, %4, 5\n v_readlane_b32 s21, %5, 5\n v_readlane_b32 s22, %6, 5\n v_readlane_b32 s23, %7, 5\n")
DEFINE_TEST(s_nop0, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
DEFINE_TEST(s_mov, "s_mov_b32 s20, s21\n s_mov_b32 s21, s22\n s_mov_b32 s22, s23\n s_mov_b32 s23, s20\n s_mov_b32 s20, s21\n s_mov_b32 s21, s22\n s_mov_b32 s22, s23\n s_mov_b32 s23, s20\n")
// VALU beside SALU: does a scalar instruction take a VALU issue slot of the same wave?
DEFINE_TEST(fma4_smov4, "v_fma_f32 %0, %0, %8, %9\n s_mov_b32 s20, s21\n v_fma_f32 %1, %1, %8, %9\n s_mov_b32 s21, s22\n"
                        "v_fma_f32 %2, %2, %8, %9\n s_mov_b32 s22, s23\n v_fma_f32 %3, %3, %8, %9\n s_mov_b32 s23, s20\n")
DEFINE_TEST(ds_write_b32, "ds_write_b32 %11, %0\n ds_write_b32 %11, %1 offset:4096\n ds_write_b32 %11, %2 offset:8192\n ds_write_b32 %11, %3 offset:12288\n"
                          "ds_write_b32 %11, %4\n ds_write_b32 %11, %5 offset:4096\n ds_write_b32 %11, %6 offset:8192\n ds_write_b32 %11, %7 offset:12288\n")
// 1 LDS read per 6 VALU, the backward's state-loop ratio
DEFINE_TEST(fma6_dsread1, "ds_bpermute_b32 %7, %12, %8\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                          "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n s_nop 0\n")

struct Test { const char* name; void (*fn)(float*, long long*, float); int ninstr; };
#define T(NAME, N) {#NAME, k_##NAME, N}
static Test g_tests[] = {
    T(fma, 8), T(fmac, 8), T(mul, 8), T(fma_chain, 8), T(exp, 8), T(rcp, 8), T(exp1_fma7, 8), T(cndmask, 8), T(fmac_dpp_shr, 8),
    T(mscan_step, 11), T(mov_dpp_ror, 8), T(permlane32_swap, 8), T(permlane16_swap, 8), T(swap_add, 8), T(ds_bpermute, 8),
    T(readlane, 8), T(s_nop0, 8), T(s_mov, 8), T(fma4_smov4, 8), T(ds_write_b32, 8), T(fma6_dsread1, 8),
};

int main(int argc, char** argv) {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out; long long* cyc;
    CHECK(hipMalloc(&out, sizeof(float) * cus * 1024));
    CHECK(hipMalloc(&cyc, sizeof(long long) * cus * 16));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", prop.gcnArchName, cus, prop.clockRate / 1000);
    for (const Test& t : g_tests) {
        for (int w : {1, 2, 3, 4}) {
            const int threads = 256 * w;
            const size_t lds = 96 * 1024;                         // > half of the CU's LDS: one workgroup per CU
            CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(t.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(t.fn, dim3(cus), dim3(threads), lds, 0, out, cyc, 1.0f);   // warm
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(t.fn, dim3(cus), dim3(threads), lds, 0, out, cyc, 1.0f);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<long long> h(cus * 4 * w);
            CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            const double med = (double)h[h.size() / 2];
            const double ninstr = (double)ITER * 4 * t.ninstr;
            printf("{\"test\": \"%s\", \"waves_per_simd\": %d, \"cyc_wave\": %.2f, \"cyc_simd\": %.2f, \"us\": %.1f}\n",
                   t.name, w, med / ninstr, med / ninstr / w, ms * 1e3);
        }
    }
    return 0;
}
