// valu_ubench.hip -- issue-rate micro-benchmarks of the gfx950 instructions the scan kernels are made of.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_ubench tools/ubench/valu_ubench.hip
//   ./valu_ubench            -> one JSON line per (test, waves per SIMD)
//
// Every test is a loop of ITER iterations over an unrolled block of UNR copies of one instruction
// (independent destination registers unless the test is a dependent chain).  The kernel runs on every
// CU with k waves per SIMD (k = 1, 2, 4); cycles are read with s_memtime inside the wave, so
// "cyc" = SIMD cycles per wave-instruction at that occupancy = (elapsed cycles * k) / (instructions a wave ran)
// ... reported per SIMD: cycles_per_instr_per_simd = elapsed / (k * n_instr).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 512;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// one kernel per instruction body: 8 independent registers r0..r7, 2 sources s0, s1
#define DEFINE_TEST(NAME, ASM8, NINSTR)                                                          \
__global__ void __launch_bounds__(256) k_##NAME(float* out, long long* cyc, float seed) {         \
    __shared__ float lds[4608];                                                                   \
    float r0 = seed + threadIdx.x, r1 = r0 * 0.5f, r2 = r0 * 0.25f, r3 = r0 + 1.f, r4 = r0 + 2.f, \
          r5 = r0 + 3.f, r6 = r0 + 4.f, r7 = r0 + 5.f;                                            \
    float s0 = 0.999f, s1 = 1e-3f;                                                                \
    float2 q0 = make_float2(r0, r1), q1 = make_float2(r2, r3), q2 = make_float2(r4, r5), q3 = make_float2(r6, r7); \
    float2 t0 = make_float2(0.999f, 0.998f), t1 = make_float2(1e-3f, 2e-3f);                      \
    unsigned la = (threadIdx.x * 8u) & 2047u; unsigned la4 = (threadIdx.x * 4u) & 1023u;                                                    \
    lds[threadIdx.x] = r0; lds[threadIdx.x + 256] = r1; __syncthreads();                          \
    long long c0 = __builtin_readcyclecounter();                                                  \
    for (int it = 0; it < ITER; ++it) {                                                           \
        asm volatile(ASM8 ASM8 ASM8 ASM8                                                          \
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), \
                       "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)                                     \
                     : "v"(s0), "v"(s1), "v"(t0), "v"(t1), "v"(la), "v"(la4) : "memory", "s20", "s21", "s22", "s23");                   \
    }                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
    long long c1 = __builtin_readcyclecounter();                                                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + q0.x + q1.y + q2.x + q3.y; \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = c1 - c0; \
}

// operands: %0..%7 = r0..r7, %8..%11 = q0..q3 (64-bit), %12 = s0, %13 = s1, %14 = t0, %15 = t1, %16 = lds addr
DEFINE_TEST(fma_indep,
    "v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %13\n v_fma_f32 %2, %2, %12, %13\n v_fma_f32 %3, %3, %12, %13\n"
    "v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %7, %7, %12, %13\n", 8)
DEFINE_TEST(fma_dep,
    "v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %0, %0, %12, %13\n"
    "v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %0, %0, %12, %13\n", 8)
DEFINE_TEST(fma_dep2,
    "v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %13\n v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %13\n"
    "v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %13\n v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %13\n", 8)
DEFINE_TEST(fmac_indep,
    "v_fmac_f32 %0, %12, %13\n v_fmac_f32 %1, %12, %13\n v_fmac_f32 %2, %12, %13\n v_fmac_f32 %3, %12, %13\n"
    "v_fmac_f32 %4, %12, %13\n v_fmac_f32 %5, %12, %13\n v_fmac_f32 %6, %12, %13\n v_fmac_f32 %7, %12, %13\n", 8)
DEFINE_TEST(mul_indep,
    "v_mul_f32 %0, %0, %12\n v_mul_f32 %1, %1, %12\n v_mul_f32 %2, %2, %12\n v_mul_f32 %3, %3, %12\n"
    "v_mul_f32 %4, %4, %12\n v_mul_f32 %5, %5, %12\n v_mul_f32 %6, %6, %12\n v_mul_f32 %7, %7, %12\n", 8)
DEFINE_TEST(add_indep,
    "v_add_f32 %0, %0, %13\n v_add_f32 %1, %1, %13\n v_add_f32 %2, %2, %13\n v_add_f32 %3, %3, %13\n"
    "v_add_f32 %4, %4, %13\n v_add_f32 %5, %5, %13\n v_add_f32 %6, %6, %13\n v_add_f32 %7, %7, %13\n", 8)
DEFINE_TEST(pk_fma,
    "v_pk_fma_f32 %8, %8, %14, %15\n v_pk_fma_f32 %9, %9, %14, %15\n v_pk_fma_f32 %10, %10, %14, %15\n v_pk_fma_f32 %11, %11, %14, %15\n"
    "v_pk_fma_f32 %8, %8, %14, %15\n v_pk_fma_f32 %9, %9, %14, %15\n v_pk_fma_f32 %10, %10, %14, %15\n v_pk_fma_f32 %11, %11, %14, %15\n", 8)
DEFINE_TEST(pk_mul,
    "v_pk_mul_f32 %8, %8, %14\n v_pk_mul_f32 %9, %9, %14\n v_pk_mul_f32 %10, %10, %14\n v_pk_mul_f32 %11, %11, %14\n"
    "v_pk_mul_f32 %8, %8, %14\n v_pk_mul_f32 %9, %9, %14\n v_pk_mul_f32 %10, %10, %14\n v_pk_mul_f32 %11, %11, %14\n", 8)
DEFINE_TEST(pk_add,
    "v_pk_add_f32 %8, %8, %15\n v_pk_add_f32 %9, %9, %15\n v_pk_add_f32 %10, %10, %15\n v_pk_add_f32 %11, %11, %15\n"
    "v_pk_add_f32 %8, %8, %15\n v_pk_add_f32 %9, %9, %15\n v_pk_add_f32 %10, %10, %15\n v_pk_add_f32 %11, %11, %15\n", 8)
DEFINE_TEST(exp_indep,
    "v_exp_f32 %0, %12\n v_exp_f32 %1, %12\n v_exp_f32 %2, %12\n v_exp_f32 %3, %12\n"
    "v_exp_f32 %4, %12\n v_exp_f32 %5, %12\n v_exp_f32 %6, %12\n v_exp_f32 %7, %12\n", 8)
DEFINE_TEST(log_indep,
    "v_log_f32 %0, %12\n v_log_f32 %1, %12\n v_log_f32 %2, %12\n v_log_f32 %3, %12\n"
    "v_log_f32 %4, %12\n v_log_f32 %5, %12\n v_log_f32 %6, %12\n v_log_f32 %7, %12\n", 8)
DEFINE_TEST(rcp_indep,
    "v_rcp_f32 %0, %12\n v_rcp_f32 %1, %12\n v_rcp_f32 %2, %12\n v_rcp_f32 %3, %12\n"
    "v_rcp_f32 %4, %12\n v_rcp_f32 %5, %12\n v_rcp_f32 %6, %12\n v_rcp_f32 %7, %12\n", 8)
// the fold/replay mix of the scan: 1 mul + 1 exp + 1 mul + 2 fma (5 instr per element-state), independent chains
DEFINE_TEST(mix_exp1_of5,
    "v_mul_f32 %0, %12, %13\n v_exp_f32 %1, %0\n v_mul_f32 %2, %12, %13\n v_fma_f32 %3, %1, %3, %2\n v_fma_f32 %4, %1, %4, %2\n"
    "v_mul_f32 %5, %12, %13\n v_exp_f32 %6, %5\n v_mul_f32 %7, %12, %13\n", 8)
DEFINE_TEST(mix_exp1_of8,
    "v_exp_f32 %0, %12\n v_fma_f32 %1, %1, %12, %13\n v_fma_f32 %2, %2, %12, %13\n v_fma_f32 %3, %3, %12, %13\n"
    "v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %7, %7, %12, %13\n", 8)
DEFINE_TEST(add_dpp_shr1,
    "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 0\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 0\n"
    "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 0\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 0\n"
    "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 0\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 0\n"
    "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 0\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 0\n", 8)
DEFINE_TEST(fmac_dpp_shr1_src,
    "v_fmac_f32_dpp %0, %12, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fmac_f32_dpp %1, %12, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
    "v_fmac_f32_dpp %2, %12, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fmac_f32_dpp %3, %12, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
    "v_fmac_f32_dpp %4, %12, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fmac_f32_dpp %5, %12, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
    "v_fmac_f32_dpp %6, %12, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_fmac_f32_dpp %7, %12, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n", 8)
DEFINE_TEST(fmac_dpp_bcast31,
    "v_fmac_f32_dpp %0, %12, %13 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_fmac_f32_dpp %1, %12, %13 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
    "v_fmac_f32_dpp %2, %12, %13 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_fmac_f32_dpp %3, %12, %13 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
    "v_fmac_f32_dpp %4, %12, %13 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_fmac_f32_dpp %5, %12, %13 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
    "v_fmac_f32_dpp %6, %12, %13 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_fmac_f32_dpp %7, %12, %13 row_bcast:31 row_mask:0xc bank_mask:0xf\n", 8)
DEFINE_TEST(mov_dpp_wave_shr1,
    "v_mov_b32_dpp %0, %12 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %12 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
    "v_mov_b32_dpp %2, %12 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %12 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
    "v_mov_b32_dpp %4, %12 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %12 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
    "v_mov_b32_dpp %6, %12 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %12 wave_shr:1 row_mask:0xf bank_mask:0xf\n", 8)
DEFINE_TEST(cndmask,
    "v_cndmask_b32 %0, %12, %13, vcc\n v_cndmask_b32 %1, %12, %13, vcc\n v_cndmask_b32 %2, %12, %13, vcc\n v_cndmask_b32 %3, %12, %13, vcc\n"
    "v_cndmask_b32 %4, %12, %13, vcc\n v_cndmask_b32 %5, %12, %13, vcc\n v_cndmask_b32 %6, %12, %13, vcc\n v_cndmask_b32 %7, %12, %13, vcc\n", 8)
DEFINE_TEST(ds_read_b64,
    "ds_read_b64 %8, %16\n ds_read_b64 %9, %16 offset:8\n ds_read_b64 %10, %16 offset:16\n ds_read_b64 %11, %16 offset:24\n"
    "ds_read_b64 %8, %16 offset:32\n ds_read_b64 %9, %16 offset:40\n ds_read_b64 %10, %16 offset:48\n ds_read_b64 %11, %16 offset:56\n", 8)
DEFINE_TEST(ds_read_b32,
    "ds_read_b32 %0, %16\n ds_read_b32 %1, %16 offset:4\n ds_read_b32 %2, %16 offset:8\n ds_read_b32 %3, %16 offset:12\n"
    "ds_read_b32 %4, %16 offset:16\n ds_read_b32 %5, %16 offset:20\n ds_read_b32 %6, %16 offset:24\n ds_read_b32 %7, %16 offset:28\n", 8)
DEFINE_TEST(ds_write_b64,
    "ds_write_b64 %16, %8\n ds_write_b64 %16, %9 offset:8\n ds_write_b64 %16, %10 offset:16\n ds_write_b64 %16, %11 offset:24\n"
    "ds_write_b64 %16, %8 offset:32\n ds_write_b64 %16, %9 offset:40\n ds_write_b64 %16, %10 offset:48\n ds_write_b64 %16, %11 offset:56\n", 8)
// VALU beside LDS reads: 1 ds_read_b64 per 4 fma
DEFINE_TEST(mix_fma4_dsread1,
    "ds_read_b64 %8, %16\n v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %13\n v_fma_f32 %2, %2, %12, %13\n v_fma_f32 %3, %3, %12, %13\n"
    "ds_read_b64 %9, %16 offset:8\n v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %7, %7, %12, %13\n", 10)

// LDS float atomics (no return): conflict-free b32 layout (lane-major), the dB/dC accumulation candidate
DEFINE_TEST(ds_add_f32,
    "ds_add_f32 %17, %0\n ds_add_f32 %17, %1 offset:2048\n ds_add_f32 %17, %2 offset:4096\n ds_add_f32 %17, %3 offset:6144\n"
    "ds_add_f32 %17, %4 offset:8192\n ds_add_f32 %17, %5 offset:10240\n ds_add_f32 %17, %6 offset:12288\n ds_add_f32 %17, %7 offset:14336\n", 8)
DEFINE_TEST(ds_write_b32,
    "ds_write_b32 %17, %0\n ds_write_b32 %17, %1 offset:2048\n ds_write_b32 %17, %2 offset:4096\n ds_write_b32 %17, %3 offset:6144\n"
    "ds_write_b32 %17, %4 offset:8192\n ds_write_b32 %17, %5 offset:10240\n ds_write_b32 %17, %6 offset:12288\n ds_write_b32 %17, %7 offset:14336\n", 8)
DEFINE_TEST(mix_fma8_dsadd2,
    "ds_add_f32 %17, %0\n v_fma_f32 %1, %1, %12, %13\n v_fma_f32 %2, %2, %12, %13\n v_fma_f32 %3, %3, %12, %13\n v_fma_f32 %4, %4, %12, %13\n"
    "ds_add_f32 %17, %0 offset:2048\n v_fma_f32 %5, %5, %12, %13\n v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %7, %7, %12, %13\n v_fma_f32 %1, %1, %12, %13\n", 10)
DEFINE_TEST(readlane,
    "v_readlane_b32 s20, %0, 5\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 5\n v_readlane_b32 s23, %3, 5\n"
    "v_readlane_b32 s20, %4, 5\n v_readlane_b32 s21, %5, 5\n v_readlane_b32 s22, %6, 5\n v_readlane_b32 s23, %7, 5\n", 8)

struct Test { const char* name; void (*fn)(float*, long long*, float); int ninstr; };
#define T(NAME, N) {#NAME, k_##NAME, N}
static Test g_tests[] = {
    T(fma_indep, 8), T(fma_dep, 8), T(fma_dep2, 8), T(fmac_indep, 8), T(mul_indep, 8), T(add_indep, 8),
    T(pk_fma, 8), T(pk_mul, 8), T(pk_add, 8), T(exp_indep, 8), T(log_indep, 8), T(rcp_indep, 8),
    T(mix_exp1_of5, 8), T(mix_exp1_of8, 8), T(add_dpp_shr1, 8), T(fmac_dpp_shr1_src, 8), T(fmac_dpp_bcast31, 8),
    T(mov_dpp_wave_shr1, 8), T(cndmask, 8), T(ds_read_b64, 8), T(ds_read_b32, 8), T(ds_write_b64, 8),
    T(mix_fma4_dsread1, 10), T(ds_add_f32, 8), T(ds_write_b32, 8), T(mix_fma8_dsadd2, 10), T(readlane, 8),
};

// ---- semantics check: VOP2 DPP without bound_ctrl leaves lanes whose source is out of range untouched
__global__ void k_dpp_semantics(float* out) {
    const int lane = threadIdx.x;
    float p = 2.0f + lane;            // every lane distinct
    float q = p;
    // multiplicative scan step: lanes with no source must keep p (identity), others p[i] * p[i-1]
    asm volatile("s_nop 1\n v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1" : "+v"(q));
    out[lane] = q;
    float r = p;
    asm volatile("s_nop 1\n v_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1" : "+v"(r));
    out[64 + lane] = r;
    float s = p;
    asm volatile("s_nop 1\n v_mul_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n s_nop 1" : "+v"(s));
    out[128 + lane] = s;
    float t = p;
    asm volatile("s_nop 1\n v_mul_f32_dpp %0, %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n s_nop 1" : "+v"(t));
    out[192 + lane] = t;
}

int main(int argc, char** argv) {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    float* out; long long* cyc;
    CHECK(hipMalloc(&out, sizeof(float) * cus * 8 * 256));
    CHECK(hipMalloc(&cyc, sizeof(long long) * cus * 8 * 4));
    {
        k_dpp_semantics<<<1, 64>>>(out);
        CHECK(hipDeviceSynchronize());
        std::vector<float> h(256);
        CHECK(hipMemcpy(h.data(), out, 256 * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            const float p = 2.0f + l;
            const float e_shr = (l % 16 == 0) ? p : p * (p - 1.0f);
            const int row = l / 16;
            const float e_b15 = (row == 1 || row == 3) ? p * (2.0f + (row * 16 - 1)) : p;
            const float e_b31 = (row >= 2) ? p * (2.0f + 31) : p;
            const float e_shl = (l % 16 == 15) ? p : p * (p + 1.0f);
            if (h[l] != e_shr) { if (bad < 8) printf("# shr1 lane %d got %g want %g\n", l, h[l], e_shr); ++bad; }
            if (h[64 + l] != e_b15) { if (bad < 8) printf("# bcast15 lane %d got %g want %g\n", l, h[64 + l], e_b15); ++bad; }
            if (h[128 + l] != e_b31) { if (bad < 8) printf("# bcast31 lane %d got %g want %g\n", l, h[128 + l], e_b31); ++bad; }
            if (h[192 + l] != e_shl) { if (bad < 8) printf("# shl1 lane %d got %g want %g\n", l, h[192 + l], e_shl); ++bad; }
        }
        printf("{\"test\": \"dpp_no_bound_ctrl_keeps_lane\", \"mismatches\": %d}\n", bad);
    }
    for (const Test& t : g_tests) {
        for (int k : {1, 2, 4, 8}) {
            const int blocks = cus * k;       // 256-thread blocks: one wave per SIMD each
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            t.fn<<<blocks, 256>>>(out, cyc, 1.0f);   // warm
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            t.fn<<<blocks, 256>>>(out, cyc, 1.0f);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<long long> h(blocks * 4);
            CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            const double med = (double)h[h.size() / 2];
            const double ninstr = (double)ITER * 4 * t.ninstr;
            printf("{\"test\": \"%s\", \"waves_per_simd\": %d, \"cyc_per_instr_wave\": %.2f, \"cyc_per_instr_simd\": %.2f, \"kernel_us\": %.1f}\n",
                   t.name, k, med / ninstr, med / ninstr / k, ms * 1e3);
        }
    }
    return 0;
}
