// scan_fwd.hip -- selective-scan forward for gfx950 (MI355X), wave64.
//
// Replaces selective_scan_fwd_kernel (reference:
// models/encoders/selective_scan/csrc/selective_scan/selective_scan_fwd_kernel.cuh:62-206).
// Same mathematics, different machine mapping:
//
//   * one WAVE per channel row; a workgroup is `nwaves` rows of the SAME (batch, group),
//     so the group's B/C tile is fetched from HBM/L2 once per workgroup and shared via LDS
//     (the reference re-reads B/C from global for every row and every state);
//   * the sequence is walked in tiles of 64*T elements; lane i owns T consecutive
//     elements; per state n: serial fold over the lane's T elements (decay a, input b kept
//     in registers), one DPP wave scan of the 64 lane aggregates, serial replay with the
//     right incoming state.  The lane decay product is exp2(A * sum(delta)) -- one
//     transcendental instead of a T-long product;
//   * B/C are staged kStateBlock states at a time in a padded, lane-blocked LDS layout that
//     makes the ds_read_b128 of a lane's T consecutive values bank-conflict free;
//   * running state between tiles lives in LDS (one float per (row, state)), checkpoints
//     every 2048 elements go to x exactly as the reference lays them out
//     (selective_scan.cpp:225-228, fwd_kernel.cuh:181-184).
#include "scan_device.h"
#include "scan_launch.h"

namespace sigma {

template <typename io_t, int T>
__global__ void __launch_bounds__(1024)
scan_fwd_kernel(const FwdArgs p) {
    using G = TileGeom<T>;
    constexpr int NB = kStateBlock;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sB = smem;
    float* sC = smem + NB * G::ROW;
    float* sCA = smem + 2 * NB * G::ROW;              // [nwaves][N] float2 {A[r,n]*log2(e), running state x[n]}

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int nwaves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.N, L = p.L;

    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int b = lb / p.rowblocks;
    const int rb = lb - b * p.rowblocks;
    const int row0 = rb * nwaves;
    const int r = row0 + wave;
    const int g = row0 / p.rows_per_group;
    const bool vec = p.vec_ok != 0;

    const io_t* __restrict__ u_row = reinterpret_cast<const io_t*>(p.u) + (long)b * p.u_bs + (long)r * p.u_ds;
    const io_t* __restrict__ d_row = reinterpret_cast<const io_t*>(p.delta) + (long)b * p.dt_bs + (long)r * p.dt_ds;
    io_t* __restrict__ o_row = reinterpret_cast<io_t*>(p.out) + (long)b * p.o_bs + (long)r * p.o_ds;
    const io_t* __restrict__ Bg = reinterpret_cast<const io_t*>(p.B) + (long)b * p.B_bs + (long)g * p.B_gs;
    const io_t* __restrict__ Cg = reinterpret_cast<const io_t*>(p.C) + (long)b * p.C_bs + (long)g * p.C_gs;
    const float* __restrict__ A_row = p.A + (long)r * p.A_ds;
    const float bias = p.bias ? p.bias[r] : 0.0f;
    const float Dd = p.D ? p.D[r] : 0.0f;
    float* __restrict__ x_row = p.x ? p.x + ((long)b * p.dim + r) * (long)p.n_chunks * 2 * N : nullptr;

    for (int n = lane; n < N; n += 64) {
        sCA[(wave * N + n) * 2 + 0] = A_row[(long)n * p.A_ns] * kLog2e;
        sCA[(wave * N + n) * 2 + 1] = 0.0f;
    }

    float dtot = 0.0f;                                 // sum of delta from l = 0 (wave-uniform)
    const int ntiles = (L + G::TILE - 1) / G::TILE;
    for (int tile = 0; tile < ntiles; ++tile) {
        const int l0 = tile * G::TILE;
        const int lbase = l0 + lane * T;

        float dl[T], dlu[T], y[T];
        {
            float uv[T], dv[T];
            load_items<io_t, T>(u_row, lbase, L, vec, uv);
            load_items<io_t, T>(d_row, lbase, L, vec, dv);
#pragma unroll
            for (int k = 0; k < T; ++k) {
                float d = dv[k] + bias;
                if (p.softplus) { float sig; d = softplus_ref(d, sig); }
                d = (lbase + k < L) ? d : 0.0f;        // identity element past the end (a = 1, b = 0)
                dl[k] = d;
                dlu[k] = d * uv[k];
                y[k] = Dd * uv[k];
            }
        }
        float dsum = 0.0f;
#pragma unroll
        for (int k = 0; k < T; ++k) dsum += dl[k];
        dtot += wave_sum(dsum);
        const bool ckpt = x_row != nullptr && ((((l0 + G::TILE) & (2048 - 1)) == 0) || tile == ntiles - 1);
        const int chunk = l0 >> 11;

        for (int nb0 = 0; nb0 < N; nb0 += NB) {
            __syncthreads();                            // previous state block fully consumed
            // ---- stage B/C rows nb0 .. nb0+NB-1 of this tile into LDS (lane-blocked, padded)
            for (int idx = tid; idx < NB * (G::TILE / 4); idx += blockDim.x) {
                const int nn = idx / (G::TILE / 4);
                const int l4 = (idx - nn * (G::TILE / 4)) * 4;
                const int n = nb0 + nn;
                const int l = l0 + l4;
                float bv[4] = {0.f, 0.f, 0.f, 0.f}, cv[4] = {0.f, 0.f, 0.f, 0.f};
                if (n < N && l < L) {
                    load4<io_t>(Bg + (long)n * p.B_ns + l, vec, L - l, bv);
                    load4<io_t>(Cg + (long)n * p.C_ns + l, vec, L - l, cv);
                }
                const int off = nn * G::ROW + (l4 / T) * G::LSTR + (l4 % T);
                *reinterpret_cast<float4*>(sB + off) = make_float4(bv[0], bv[1], bv[2], bv[3]);
                *reinterpret_cast<float4*>(sC + off) = make_float4(cv[0], cv[1], cv[2], cv[3]);
            }
            __syncthreads();

            const int nend = (N - nb0 < NB) ? (N - nb0) : NB;
#pragma unroll 1
            for (int nn = 0; nn < nend; ++nn) {
                const int n = nb0 + nn;
                const float2 ca = *reinterpret_cast<const float2*>(sCA + (wave * N + n) * 2);
                const float A2 = ca.x;
                const float xin_tile = ca.y;
                const float4* __restrict__ pB = reinterpret_cast<const float4*>(sB + nn * G::ROW + lane * G::LSTR);
                const float4* __restrict__ pC = reinterpret_cast<const float4*>(sC + nn * G::ROW + lane * G::LSTR);

                // ---- pass A: lane-local fold with zero incoming state
                float a[T], bb[T];
                float xa = 0.0f;
#pragma unroll
                for (int q = 0; q < T / 4; ++q) {
                    const float4 bv = pB[q];
                    const float bq[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = 4 * q + j;
                        a[k] = fast_exp2(dl[k] * A2);
                        bb[k] = dlu[k] * bq[j];
                        xa = fmaf(a[k], xa, bb[k]);
                    }
                }
                // ---- wave scan of the lane aggregates (decay product, end state)
                float pa = fast_exp2(A2 * dsum);
                wave_scan_inclusive(pa, xa);
                const float pe = wave_prev_lane(pa, 1.0f);
                const float xe = wave_prev_lane(xa, 0.0f);
                float x = fmaf(pe, xin_tile, xe);       // state entering this lane's segment

                // ---- pass B: replay with the true incoming state, accumulate C.x
#pragma unroll
                for (int q = 0; q < T / 4; ++q) {
                    const float4 cv = pC[q];
                    const float cq[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = 4 * q + j;
                        x = fmaf(a[k], x, bb[k]);
                        y[k] = fmaf(cq[j], x, y[k]);
                    }
                }
                if (lane == 63) {
                    sCA[(wave * N + n) * 2 + 1] = x;    // running state after this tile
                    if (ckpt) {
                        float2 v;
                        v.x = fast_exp2(A2 * dtot);     // prod of a[n, 0..end(chunk)]
                        v.y = x;
                        *reinterpret_cast<float2*>(x_row + ((long)chunk * N + n) * 2) = v;
                    }
                }
            }
        }
        store_items<io_t, T>(o_row, lbase, L, vec, y);
    }
}

template <typename io_t, int T>
static hipError_t launch_fwd_t(const FwdArgs& a, int nwaves, hipStream_t stream) {
    using G = TileGeom<T>;
    const size_t lds = fwd_lds_bytes(T, nwaves, a.N);
    const int grid = a.rowblocks * a.batch;
    auto kern = scan_fwd_kernel<io_t, T>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nwaves * 64), lds, stream, a);
    (void)sizeof(G);
    return hipGetLastError();
}

template <typename io_t>
static hipError_t launch_fwd_io(const FwdArgs& a, int T, int nwaves, hipStream_t stream) {
    switch (T) {
        case 4: return launch_fwd_t<io_t, 4>(a, nwaves, stream);
        case 8: return launch_fwd_t<io_t, 8>(a, nwaves, stream);
        case 16: return launch_fwd_t<io_t, 16>(a, nwaves, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_scan_fwd(const FwdArgs& a, int dtype, int T, int nwaves, hipStream_t stream) {
    switch (dtype) {
        case 0: return launch_fwd_io<float>(a, T, nwaves, stream);
        case 1: return launch_fwd_io<f16_t>(a, T, nwaves, stream);
        case 2: return launch_fwd_io<bf16_t>(a, T, nwaves, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace sigma
