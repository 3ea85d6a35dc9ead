// scan_fwd4.hip -- selective-scan forward, "quad-row" mapping (gfx950 / MI355X, wave64, f32 IO, ckpt_pitch 160).
//
// Same operator as scan_fwd.hip (reference: models/encoders/selective_scan/csrc/selective_scan/
// selective_scan_fwd_kernel.cuh:62-238; mathematics SURVEY.md App. E.1), with the row-to-lane mapping of the
// quad-row backward (scan_bwd4.hip): a wave is FOUR channel rows x 16 lanes x 10 positions, one DPP row per
// channel row, tiles of 160 positions = the checkpoint pitch the quad-row backward reads.
//   * the scan of a state is 4 row_shr steps of v_fmac_f32_dpp + v_mul_f32_dpp inside a DPP row (the 64-lane
//     kernel: 6 steps incl. two row_bcast steps), the state entering a tile is folded into lane 0 of the row;
//   * a wave keeps its rows for the whole sequence, so the running state of all N states of its four rows is ONE
//     lane-vector register (lane 16*row + state) handed from tile to tile -- no LDS, no barrier per state block;
//     the checkpoint of a tile is that register stored as it is (64 contiguous bytes per row);
//   * the B/C image of a tile (all N states, 2*N*160 floats) is staged once per tile by LDS-DMA while the previous
//     tile is computed: ONE workgroup barrier per tile (the 64-lane kernel: one per block of 4 states);
//   * A and the incoming state of a state are broadcast inside their DPP row with ds_bpermute_b32, one state ahead.
// Needs what the quad-row backward needs (capi.hip: plan_fwd4); everything else runs scan_fwd.hip.
#include "scan_device.h"
#include "scan_launch.h"
#include "scan_quad.h"

#include <atomic>

namespace sigma {

template <bool REV>
__device__ __forceinline__ void scan_fwd4_body(const FwdArgs& p, float* smem, int b, int g, int chunk) {
    constexpr int T = kT4;
    const int W = blockDim.x >> 6;                    // waves; 4 rows each
    const int N = p.N, L = p.L;
    const int bufsz = 2 * N * kTile4;
    float* sBC = smem;                                // [2][2][N][160]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = lane & 15;
    const int qr = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool vec = p.vec_ok != 0;
    const bool li0 = li == 0, li15 = li == 15;
    const int rowbase4 = (lane & 48) << 2;            // ds_bpermute byte address of lane 0 of this DPP row
    const int vshift = 16 - N;                        // state n of a row sits in lane vshift + n of the state vector

    const int r = g * p.rows_per_group + (chunk * W + wave) * 4 + qr;          // channel row of this DPP row
    const int ur = r - ((g - (g >> p.u_gshift)) * p.rows_per_group);           // same row of group g >> u_gshift
    const float* __restrict__ u_row = reinterpret_cast<const float*>(p.u) + (long)b * p.u_bs + (long)ur * p.u_ds;
    const float* __restrict__ d_row = reinterpret_cast<const float*>(p.delta) + (long)b * p.dt_bs + (long)r * p.dt_ds;
    float* __restrict__ o_row = reinterpret_cast<float*>(p.out) + (long)b * p.o_bs + (long)r * p.o_ds;
    const float* __restrict__ Bg = reinterpret_cast<const float*>(p.B) + (long)b * p.B_bs + (long)g * p.B_gs;
    const float* __restrict__ Cg = reinterpret_cast<const float*>(p.C) + (long)b * p.C_bs + (long)g * p.C_gs;
    const int pr = param_row(r, g, p.rows_per_group, p.pswap);
    const float bias = p.bias ? p.bias[pr] : 0.0f;
    const float Dd = p.D ? p.D[pr] : 0.0f;
    float* __restrict__ x_row = p.x ? p.x + ((long)b * p.dim + r) * p.x_rs : nullptr;

    // lane vectors: A2v lane 16*row + n = A[row, n] * log2(e); Xv lane 16*row + vshift + n = running state n
    float A2v = 0.0f;
    if (li < N) A2v = p.A[(long)pr * p.A_ds + (long)li * p.A_ns] * kLog2e;
    float Xv = 0.0f;

    // never multiply uninitialised LDS bits (stale/NaN) into the padding of the last tile
    for (int i = tid; i < 2 * bufsz / 4; i += blockDim.x) reinterpret_cast<float4*>(sBC)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int ntiles = (L + kTile4 - 1) / kTile4;
    auto stage = [&](int buf, int tile) {
        stage_tile4<REV>(sBC + buf * bufsz, Bg, Cg, (int)p.B_ns, (int)p.C_ns, N, tile, L);
    };
    stage(0, 0);
    float uv[T], dv[T];
    load_items<float, T, REV>(u_row, li * T, L, vec, uv);
    load_items<float, T, REV>(d_row, li * T, L, vec, dv);
    lds_dma_wait();
    __syncthreads();

    int buf = 0;
    for (int j = 0; j < ntiles; ++j) {
        const int lbase = j * kTile4 + li * T;
        const float* cur = sBC + buf * bufsz;
        if (j + 1 < ntiles) stage(buf ^ 1, j + 1);    // lands while this tile is computed

        float dl[T], dlu[T], y[T];
        float dsum = 0.0f;
#pragma unroll
        for (int k = 0; k < T; ++k) {
            float d = dv[k] + bias;
            if (p.softplus) { float sig; d = softplus_ref(d, sig); }
            d = (lbase + k < L) ? d : 0.0f;            // identity element past the end (a = 1, b = 0)
            dl[k] = d;
            dlu[k] = d * uv[k];
            y[k] = Dd * uv[k];
            dsum += d;
        }
        if (j + 1 < ntiles) {                          // next tile's u / delta fly during the state loop
            load_items<float, T, REV>(u_row, lbase + kTile4, L, vec, uv);
            load_items<float, T, REV>(d_row, lbase + kTile4, L, vec, dv);
        }

        float Xn = 0.0f;                               // states leaving this tile, collected state by state
        float A2_nx = row_pick(A2v, rowbase4), x0_nx = row_pick(Xv, rowbase4 + 4 * vshift);
#pragma unroll 1
        for (int n = 0; n < N; ++n) {
            const float A2 = A2_nx, x0 = x0_nx;
            {
                const int nn = (n + 1 < N) ? n + 1 : n;
                A2_nx = row_pick(A2v, rowbase4 + 4 * nn);
                x0_nx = row_pick(Xv, rowbase4 + 4 * (vshift + nn));
            }
            const float* tB = cur + n * kTile4;
            const float* tC = tB + N * kTile4;
            float a[T], bb[T];
            // ---- in-lane fold (lane 0 of the row starts from the state entering the tile), row scan
            float xa = li0 ? x0 : 0.0f;
#pragma unroll
            for (int q = 0; q < T / 2; ++q) {
                float bq[2];
                lds_read_pair<REV>(tB, li, q, bq);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int k = 2 * q + jj;
                    a[k] = fast_exp2(dl[k] * A2);
                    bb[k] = dlu[k] * bq[jj];
                    xa = fmaf(a[k], xa, bb[k]);
                }
            }
            float pl = fast_exp2(A2 * dsum);           // the lane's decay product
            row_mscan_inclusive(pl, xa);
            float x = dpp_take<DPP_ROW_SHR1, 0xF>(x0, xa);       // state entering this lane's segment
            // ---- replay with the true incoming state, accumulate C.x
#pragma unroll
            for (int q = 0; q < T / 2; ++q) {
                float cq[2];
                lds_read_pair<REV>(tC, li, q, cq);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int k = 2 * q + jj;
                    x = fmaf(a[k], x, bb[k]);
                    y[k] = fmaf(cq[jj], x, y[k]);
                }
            }
            // lane 15 of the row holds the state after the tile (positions past L pass it through): rotate + insert,
            // so that after N states state n sits in lane vshift + n
            const float rot = row_rotate_left(Xn);
            Xn = li15 ? x : rot;
        }
        Xv = Xn;
        store_items<float, T, REV>(o_row, lbase, L, vec, y);
        if (x_row != nullptr && li >= vshift) x_row[(long)j * N + (li - vshift)] = Xv;   // checkpoint j = state after tile j
        lds_dma_wait();
        __syncthreads();                               // next image landed; this one consumed by every wave
        buf ^= 1;
    }
}

__global__ void __launch_bounds__(512)
scan_fwd4_kernel(const FwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int P = p.rowblocks;                        // workgroups per (batch, group)
    const int per_b = p.G * P;
    const int b = lb / per_b;
    const int rem = lb - b * per_b;
    const int g = rem / P;
    const int chunk = rem - g * P;
    if ((p.rev_mask >> g) & 1u) scan_fwd4_body<true>(p, smem, b, g, chunk);
    else scan_fwd4_body<false>(p, smem, b, g, chunk);
}

// a.R = waves per workgroup (4 rows each, <= 8), a.rowblocks = workgroups per (batch, group)
hipError_t launch_scan_fwd4(const FwdArgs& a, hipStream_t stream) {
    const size_t lds = fwd4_lds_bytes(a.N);
    const int grid = a.batch * a.G * a.rowblocks;
    auto kern = scan_fwd4_kernel;
    static std::atomic<size_t> lds_cap[kMaxDevices];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDevices) dev = 0;
    if (lds > 48 * 1024 && lds > lds_cap[dev].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_cap[dev].store(lds, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(a.R * 64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace sigma
