// stateloop_ubench.hip -- what a SIMD of MI355X sustains on the STATE LOOP of the row-lane forward scan (round 6).
//
// The forward's recurrence per element-state is   a = exp2(dl * A2);  x = a * x + dlu * B;  y += C * x   = 4 plain VALU
// + 1 transcendental, with B / C as SGPR operands.  The kernels of csrc/scan_fwdr.hip run it at ~30 SIMD clocks per
// element-state where the per-class issue costs (profiles/r03_issue_ubench.jsonl) predict 17-20.  This bench isolates the
// loop: no memory traffic, no barriers, no LDS -- variants of the same arithmetic at a KNOWN number of resident waves.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o tools/ubench/bin/stateloop_ubench tools/ubench/stateloop_ubench.hip
//   ./stateloop_ubench   -> one JSON line per (variant, workgroups per CU)
//
// Occupancy: workgroups of 256 threads (one wave per SIMD), k workgroups per CU enforced by the dynamic LDS size
// (160 KB / k), grid = 16 rounds x k x CUs so that dispatch imbalance averages out; wall time by HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int T = 16;          // positions per tile
constexpr int NS = 4;          // states per wave
constexpr int TILES = 400;     // tiles per wave

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v16f_a16 __attribute__((ext_vector_type(16), aligned(16)));
typedef const __attribute__((address_space(4))) v16f_a16* cv16p_t;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// VARIANT 0: B / C by scalar loads per state, next state's requested a state ahead (the production scheme)
//         1: B / C in SGPRs, loaded once (no scalar loads in the loop)
//         2: B / C in VGPRs (same value in every lane)
//         3: as 1, exp replaced by a multiply (prices the transcendental)
//         4: as 1, the 16 exponentials of a state first, then the 16 recurrence steps (transcendental pipe back to back)
//         5: as 1, two states interleaved position by position (two independent chains)
template <int VARIANT>
__global__ void __launch_bounds__(256) k_loop(const float* __restrict__ bc, float* __restrict__ out, int tiles, float* __restrict__ big, long big_floats) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    float dl[T], dlu[T], y[T], A2[NS], x[NS];
#pragma unroll
    for (int k = 0; k < T; ++k) { dl[k] = 0.01f + 1e-4f * (lane + k); dlu[k] = 0.02f + 1e-4f * (lane - k); y[k] = 0.0f; }
#pragma unroll
    for (int s = 0; s < NS; ++s) { A2[s] = -(1.0f + s) * 1.4427f; x[s] = 0.0f; }
    const float* Bw = bc + (blockIdx.x & 7) * 4096;
    float Bn[T], Cn[T];
    if (VARIANT == 0) {
        const v16f b = *reinterpret_cast<cv16p_t>(reinterpret_cast<uintptr_t>(Bw)), c = *reinterpret_cast<cv16p_t>(reinterpret_cast<uintptr_t>(Bw + 16));
#pragma unroll
        for (int k = 0; k < T; ++k) { Bn[k] = b[k]; Cn[k] = c[k]; }
    }
    float Bf[T], Cf[T];
    if (VARIANT != 0) {
        if (VARIANT == 2) {
#pragma unroll
            for (int k = 0; k < T; ++k) { Bf[k] = Bw[k]; Cf[k] = Bw[16 + k]; asm volatile("" : "+v"(Bf[k]), "+v"(Cf[k])); }
        } else {
            const v16f b = *reinterpret_cast<cv16p_t>(reinterpret_cast<uintptr_t>(Bw)), c = *reinterpret_cast<cv16p_t>(reinterpret_cast<uintptr_t>(Bw + 16));
#pragma unroll
            for (int k = 0; k < T; ++k) { Bf[k] = b[k]; Cf[k] = c[k]; }
        }
    }
    // memory variants (arithmetic of variant 1): 9 = + 8 dword stores per tile (checkpoints as the kernels write them now: 256 B per
    // wave store), 10 = + 2 dwordx4 stores (the same bytes, states innermost), 11 = + 2 dwordx4 loads in the (row, chunk)
    // pattern (16 rows x 64 B per wave load, rows 4800 B apart), 12 = + 1 dwordx4 store in that pattern, 13 = 11 + 12 + 10
    constexpr bool MEMV = VARIANT >= 9;
    const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float* ckp = big + ((wave_id * 75 * 512) % (big_floats / 2 - 75 * 512 - 4096)) + lane;          // 2 KB per tile and wave
    const int rr = lane >> 2, cc = lane & 3;
    float* rowp = big + big_floats / 2 + ((wave_id * 16 * 1200) % (big_floats / 2 - 17 * 1200 - 4096)) + rr * 1200 + 4 * cc;
    typedef float v4 __attribute__((ext_vector_type(4)));
    v4 ld0 = {0, 0, 0, 0}, ld1 = {0, 0, 0, 0};
    for (int it = 0; it < tiles; ++it) {
        if (MEMV) {
            const int t75 = it % 75;
            if (VARIANT == 11 || VARIANT == 13) {
                dl[0] += 1e-30f * ld0[0]; dlu[0] += 1e-30f * ld1[1];
                ld0 = *reinterpret_cast<const v4*>(rowp + 16 * t75);
                ld1 = *reinterpret_cast<const v4*>(rowp + 16 * t75 + 600);
            }
            if (VARIANT == 12 || VARIANT == 13) { const v4 o = {y[0], y[1], y[2], y[3]}; *reinterpret_cast<v4*>(rowp + 16 * t75 + 8) = o; }
            if (VARIANT == 9) {
#pragma unroll
                for (int j = 0; j < 8; ++j) ckp[t75 * 512 + j * 64] = y[j];
            }
            if (VARIANT == 14 || VARIANT == 16) {
                // the same row bytes as 13, but 128 (256) contiguous bytes per row every 2nd (4th) tile: lane -> (row = lane / W, chunk = lane % W)
                constexpr int W = VARIANT == 14 ? 8 : 16, EV = VARIANT == 14 ? 2 : 4;
                if (it % EV == 0) {
                    float* rp = big + big_floats / 2 + ((wave_id * 16 * 1200) % (big_floats / 2 - 17 * 1200 - 4096)) + 4 * (lane % W) + 16 * (t75 / EV) * EV;
                    dl[0] += 1e-30f * ld0[0]; dlu[0] += 1e-30f * ld1[1];
#pragma unroll
                    for (int j = 0; j < EV; ++j) {
                        const int row = lane / W + j * (64 / W);
                        ld0 = *reinterpret_cast<const v4*>(rp + row * 1200);
                        dl[1 + j] += 1e-30f * ld0[0];
                        ld1 = *reinterpret_cast<const v4*>(rp + row * 1200 + 600);
                    }
#pragma unroll
                    for (int j = 0; j < EV / 2; ++j) {
                        const v4 o = {y[0], y[1], y[2], y[3 + j]};
                        const int row = lane / W + (2 * j) * (64 / W);
                        *reinterpret_cast<v4*>(rp + row * 1200 + 300) = o;
                        *reinterpret_cast<v4*>(rp + (row + 64 / W) * 1200 + 300) = o;
                    }
                }
            }
            if (VARIANT == 15) {
                dl[0] += 1e-30f * ld0[0]; dlu[0] += 1e-30f * ld1[1];
                ld0 = *reinterpret_cast<const v4*>(rowp + 16 * t75);
                ld1 = *reinterpret_cast<const v4*>(rowp + 16 * t75 + 600);
                const v4 o = {y[0], y[1], y[2], y[3]}; *reinterpret_cast<v4*>(rowp + 16 * t75 + 8) = o;
                const v4 c0 = {y[0], y[1], y[2], y[3]};
                *reinterpret_cast<v4*>(ckp - lane + t75 * 256 + lane * 4) = c0;
            }
            if (VARIANT == 10 || VARIANT == 13 || VARIANT == 14 || VARIANT == 16) {
                const v4 c0 = {y[0], y[1], y[2], y[3]}, c1 = {y[4], y[5], y[6], y[7]};
                *reinterpret_cast<v4*>(ckp - lane + t75 * 512 + lane * 4) = c0;
                *reinterpret_cast<v4*>(ckp - lane + t75 * 512 + 256 + lane * 4) = c1;
            }
        }
        // opaque to the optimiser: nothing of a tile is loop-invariant (the real kernel reads new delta / delta * u per tile)
        asm volatile("" : "+v"(dl[0]), "+v"(dl[1]), "+v"(dl[2]), "+v"(dl[3]), "+v"(dl[4]), "+v"(dl[5]), "+v"(dl[6]), "+v"(dl[7]),
                          "+v"(dl[8]), "+v"(dl[9]), "+v"(dl[10]), "+v"(dl[11]), "+v"(dl[12]), "+v"(dl[13]), "+v"(dl[14]), "+v"(dl[15]));
        asm volatile("" : "+v"(dlu[0]), "+v"(dlu[1]), "+v"(dlu[2]), "+v"(dlu[3]), "+v"(dlu[4]), "+v"(dlu[5]), "+v"(dlu[6]), "+v"(dlu[7]),
                          "+v"(dlu[8]), "+v"(dlu[9]), "+v"(dlu[10]), "+v"(dlu[11]), "+v"(dlu[12]), "+v"(dlu[13]), "+v"(dlu[14]), "+v"(dlu[15]));
        if (VARIANT == 6) {
            // exponentials one state ahead: a_nx[] of state s + 1 is produced next to the recurrence of state s, so that no
            // transcendental result is consumed within ~80 instructions of its issue
            float a_nx[T];
#pragma unroll
            for (int k = 0; k < T; ++k) a_nx[k] = fast_exp2(dl[k] * A2[0]);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float a_cur[T];
#pragma unroll
                for (int k = 0; k < T; ++k) a_cur[k] = a_nx[k];
                float xx = x[s];
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    if (s + 1 < NS) a_nx[k] = fast_exp2(dl[k] * A2[s + 1]);
                    xx = fmaf(a_cur[k], xx, dlu[k] * Bf[k]);
                    y[k] = fmaf(Cf[k], xx, y[k]);
                }
                x[s] = xx;
                asm volatile("" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]),
                                  "+v"(y[8]), "+v"(y[9]), "+v"(y[10]), "+v"(y[11]), "+v"(y[12]), "+v"(y[13]), "+v"(y[14]), "+v"(y[15]));
                asm volatile("" : "+v"(a_nx[0]), "+v"(a_nx[1]), "+v"(a_nx[2]), "+v"(a_nx[3]), "+v"(a_nx[4]), "+v"(a_nx[5]), "+v"(a_nx[6]), "+v"(a_nx[7]),
                                  "+v"(a_nx[8]), "+v"(a_nx[9]), "+v"(a_nx[10]), "+v"(a_nx[11]), "+v"(a_nx[12]), "+v"(a_nx[13]), "+v"(a_nx[14]), "+v"(a_nx[15]));
                __builtin_amdgcn_sched_barrier(0);
            }
            continue;
        }
        if (VARIANT == 7 || VARIANT == 8) {
            // 7: 64 independent exponentials per tile, nothing else;  8: the same + 4 independent multiplies each
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    y[k] += fast_exp2(dl[k] * A2[s]) ;
                    if (VARIANT == 8) { dlu[k] = dlu[k] * Bf[k]; dlu[k] = dlu[k] * Cf[k]; }
                }
            }
            continue;
        }
        if (VARIANT == 5) {
#pragma unroll
            for (int s = 0; s < NS; s += 2) {
                float x0 = x[s], x1 = x[s + 1];
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    const float a0 = fast_exp2(dl[k] * A2[s]), a1 = fast_exp2(dl[k] * A2[s + 1]);
                    x0 = fmaf(a0, x0, dlu[k] * Bf[k]);
                    x1 = fmaf(a1, x1, dlu[k] * Cf[k]);
                    y[k] = fmaf(Cf[k], x0, y[k]);
                    y[k] = fmaf(Bf[k], x1, y[k]);
                }
                x[s] = x0; x[s + 1] = x1;
                asm volatile("" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]),
                                  "+v"(y[8]), "+v"(y[9]), "+v"(y[10]), "+v"(y[11]), "+v"(y[12]), "+v"(y[13]), "+v"(y[14]), "+v"(y[15]));
                __builtin_amdgcn_sched_barrier(0);
            }
            continue;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float Bt[T], Ct[T];
            if (VARIANT == 0) {
#pragma unroll
                for (int k = 0; k < T; ++k) { Bt[k] = Bn[k]; Ct[k] = Cn[k]; }
                asm volatile("" : "+s"(Bt[0]), "+s"(Ct[0]));
                __builtin_amdgcn_sched_barrier(0);
                const float* nb = Bw + (((it * NS + s + 1) * 32) & 4095);
                const v16f b = *reinterpret_cast<cv16p_t>(reinterpret_cast<uintptr_t>(nb)), c = *reinterpret_cast<cv16p_t>(reinterpret_cast<uintptr_t>(nb + 16));
#pragma unroll
                for (int k = 0; k < T; ++k) { Bn[k] = b[k]; Cn[k] = c[k]; }
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int k = 0; k < T; ++k) { Bt[k] = Bf[k]; Ct[k] = Cf[k]; }
            }
            float xx = x[s];
            if (VARIANT == 4) {
                float a[T];
#pragma unroll
                for (int k = 0; k < T; ++k) a[k] = fast_exp2(dl[k] * A2[s]);
                asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                                  "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    xx = fmaf(a[k], xx, dlu[k] * Bt[k]);
                    y[k] = fmaf(Ct[k], xx, y[k]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    const float a = VARIANT == 3 ? (dl[k] * A2[s]) * 0.5f : fast_exp2(dl[k] * A2[s]);
                    xx = fmaf(a, xx, dlu[k] * Bt[k]);
                    y[k] = fmaf(Ct[k], xx, y[k]);
                }
            }
            x[s] = xx;
            asm volatile("" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]),
                              "+v"(y[8]), "+v"(y[9]), "+v"(y[10]), "+v"(y[11]), "+v"(y[12]), "+v"(y[13]), "+v"(y[14]), "+v"(y[15]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < T; ++k) acc += y[k];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc += x[s];
    if (acc == 1.2345f) out[blockIdx.x * 256 + threadIdx.x] = acc + lds[threadIdx.x];
}

typedef void (*kern_t)(const float*, float*, int, float*, long);

int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float *bc, *out;
    CHECK(hipMalloc(&bc, 8 * 4096 * sizeof(float) + 4096));
    CHECK(hipMalloc(&out, sizeof(float) * 256 * 65536));
    std::vector<float> h(8 * 4096 + 1024);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.5f + 1e-3f * (float)(i % 97);
    CHECK(hipMemcpy(bc, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    kern_t kerns[] = {k_loop<0>, k_loop<1>, k_loop<2>, k_loop<3>, k_loop<4>, k_loop<5>, k_loop<6>, k_loop<7>, k_loop<8>, k_loop<9>, k_loop<10>, k_loop<11>, k_loop<12>, k_loop<13>, k_loop<14>, k_loop<15>, k_loop<16>};
    const char* names[] = {"sgpr_sload", "sgpr_fixed", "vgpr", "sgpr_noexp", "sgpr_exp_first", "sgpr_two_chains", "exp_state_ahead", "exp_mul_add_only", "exp_mul_add_2mul",
                           "fixed+8_dword_stores", "fixed+2_x4_stores", "fixed+2_x4_loads", "fixed+1_x4_rowstore", "fixed+loads+rowstore+x4ckpt",
                           "as13_rows_128B_every_2nd", "as13_half_ckpt", "as13_rows_256B_every_4th"};
    float* big; const long big_floats = 3L << 28;    // 3 GiB
    CHECK(hipMalloc(&big, big_floats * sizeof(float)));
    CHECK(hipMemset(big, 0, big_floats * sizeof(float)));
    printf("{\"device\": \"%s\", \"cus\": %d}\n", prop.gcnArchName, cus);
    for (int v = (getenv("UB_FROM") ? atoi(getenv("UB_FROM")) : 0); v < 17; ++v) {
        for (int k = (v >= 13 ? 3 : 1); k <= (v >= 13 ? 3 : 4); ++k) {
            const size_t lds = (size_t)(160 * 1024 / k) - 1024 - (k == 1 ? 0 : 0);
            CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[v]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            int occ = 0;
            CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kerns[v], 256, lds));
            const int rounds = 8, grid = rounds * k * cus;
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(kerns[v], dim3(grid), dim3(256), lds, 0, bc, out, TILES, big, big_floats);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kerns[v], dim3(grid), dim3(256), lds, 0, bc, out, TILES, big, big_floats);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            // per SIMD: rounds * k waves, each TILES * NS * T element-states
            const double es = (double)rounds * k * TILES * NS * T;
            printf("{\"variant\": \"%s\", \"wg_per_cu\": %d, \"occupancy_api\": %d, \"us\": %.1f, \"ns_per_es_simd\": %.3f, \"clk_per_es_at_2p1GHz\": %.2f}\n",
                   names[v], k, occ, ms * 1e3, ms * 1e6 / es, ms * 1e6 / es * 2.1);
            fflush(stdout);
        }
    }
    return 0;
}
