// scan_bwd3.hip -- selective-scan backward, state-parallel mapping (gfx950 / MI355X, wave64).
//
// Replaces selective_scan_bwd_kernel + reverse_scan.cuh (reference:
// models/encoders/selective_scan/csrc/selective_scan/selective_scan_bwd_kernel.cuh:66-308,
// reverse_scan.cuh:18-401) for dstate in {4, 8, 12, 16} with one state checkpoint per 320 elements.
// Mathematics: SURVEY.md App. E.2.
//
// Why a third mapping (profiles/r02_bwd2_phases.txt): with one wave per ROW the dB/dC terms of every
// state have to cross waves -- a slab write, a workgroup barrier and a column-sum phase PER STATE; the
// phase timers of scan_bwd2 put 49 % of a wave's time into barrier + column sum + flush and only 30 %
// into the recurrences.  Here a wave owns FOUR STATES of a row instead:
//   * workgroup = SLOTS row slots x Q waves (Q = dstate / 4); wave (slot, quad) handles states
//     4*quad .. 4*quad+3 of the slot's current row; the workgroup walks RB rows per slot for every
//     320-element tile (T = 5 elements per lane), tiles last to first;
//   * dB / dC are sums over ROWS: each wave keeps the terms of its four states in 40 registers and
//     adds row after row into them -- no LDS, no barrier; the SLOTS partial sums meet once per tile;
//   * du / ddelta are sums over STATES: the Q waves of a row exchange two 5-element partials per
//     lane through LDS (parity double buffer: ONE barrier per row step, i.e. per four states of
//     work), and the waves take turns to finish the row (softplus', stores, dD, ddelta_bias);
//     with dstate = 4 a wave has the whole row and the main loop has no barrier at all;
//   * B/C of a tile (all states) are staged in LDS once per tile and serve every row of the chunk;
//   * wave scans, carry handling, checkpoint use: as in scan_bwd2.hip (multiplicative DPP scans,
//     lane vectors for A / incoming state / reverse carry).
#include "scan_device.h"
#include "scan_launch.h"

#include <atomic>

namespace sigma {

namespace {

constexpr int kT3 = 5;                 // elements per lane
constexpr int kTile3 = 64 * kT3;       // 320

typedef const __attribute__((address_space(4))) BwdArgs* cold_args3_t;   // see scan_bwd2.hip: cold_args()
__device__ __forceinline__ cold_args3_t cold_args3() {
    cold_args3_t kp = (cold_args3_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return kp;
}

__device__ __forceinline__ float lane_put3(float uniform_val, int n, float old) {
    const int sval = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, uniform_val));
    int keep;
    asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                 : "+v"(old), "=&s"(keep) : "s"(sval), "s"(n));
    return old;
}

// register staging (16-bit IO, unaligned tensors): all N states of one tile -> dst [arr][N][TILE]
template <typename io_t>
__device__ __forceinline__ void stage_tile3(float* __restrict__ dst, const io_t* __restrict__ Bg, const io_t* __restrict__ Cg,
                                            long B_ns, long C_ns, int N, int tile, int L, bool rev, bool vec) {
    constexpr int CPR = kTile3 / 4;
    const int total = 2 * N * CPR;
    const int l0 = tile * kTile3;
    for (int ci = threadIdx.x; ci < total; ci += blockDim.x) {
        const int row = ci / CPR;                      // arr * N + n
        const int c4 = (ci - row * CPR) * 4;
        const int arr = row / N;
        const int n = row - arr * N;
        const int m = rev ? (L - l0 - kTile3 + c4) : (l0 + c4);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (m < L && m + 4 > 0) {
            const io_t* __restrict__ srow = arr == 0 ? Bg + (long)n * B_ns : Cg + (long)n * C_ns;
            load4_guard<io_t>(srow, m, L, vec, v);
        }
        *reinterpret_cast<float4*>(dst + (long)ci * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace

template <typename io_t, bool GLDS, bool REV>
__device__ __forceinline__ void scan_bwd3_body(const BwdArgs& q, float* smem, int b, int g, int chunk) {
    constexpr int T = kT3;
    constexpr int TILE = kTile3;
    const FwdArgs& p = q.f;
    const int N = p.N, L = p.L, RB = q.RB;
    const int Q = N >> 2;                                 // waves per row
    const int nw = blockDim.x >> 6;
    const int slots = nw / Q;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = wave / Q;
    const int quad = wave - slot * Q;
    const int nq0 = quad * 4;                             // first of this wave's four states
    const bool vec = p.vec_ok != 0;
    const bool lane0 = lane == 0, lane63 = lane == 63;
    const int row_c0 = g * p.rows_per_group + chunk * RB * slots;

    float* sBC = smem;                                    // [2][N][TILE]: B then C of the current tile
    float* sPart = sBC + 2 * N * TILE;                    // [2][nw][2][TILE] row partials; reused [nw][4][TILE] at tile end
    float* sRv = sPart + 4 * nw * TILE;                   // [RB*slots][N] reverse carries

    const io_t* __restrict__ Bg = reinterpret_cast<const io_t*>(p.B) + (long)b * p.B_bs + (long)g * p.B_gs;
    const io_t* __restrict__ Cg = reinterpret_cast<const io_t*>(p.C) + (long)b * p.C_bs + (long)g * p.C_gs;
    float* __restrict__ oB;
    float* __restrict__ oC;
    long o_nsB, o_nsC;
    if (q.P == 1) {
        oB = q.dB + (long)b * q.dB_bs + (long)g * q.dB_gs; o_nsB = q.dB_ns;
        oC = q.dC + (long)b * q.dC_bs + (long)g * q.dC_gs; o_nsC = q.dC_ns;
    } else {
        const long slab = (((long)chunk * p.batch + b) * p.G + g) * (long)N * L;
        oB = q.ws_dB + slab; oC = q.ws_dC + slab; o_nsB = L; o_nsC = L;
    }

    for (int i = tid; i < RB * slots * N; i += blockDim.x) sRv[i] = 0.0f;
    for (int i = tid; i < 2 * N * TILE / 4; i += blockDim.x) reinterpret_cast<float4*>(sBC)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    const int ntiles = (L + TILE - 1) / TILE;
    StagePlan<T, REV> plan;
    if constexpr (GLDS) plan.init(N, 1, L);

    float accB[4][T], accC[4][T];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int k = 0; k < T; ++k) { accB[s][k] = 0.0f; accC[s][k] = 0.0f; }

    int par = 0;                                          // parity of the row-partial buffer
    for (int j = ntiles - 1; j >= 0; --j) {
        const int l0 = j * TILE;
        const int lbase = l0 + lane * T;
        // ---- B/C of this tile, all states (every wave is past the previous tile: barrier at its end)
        if constexpr (GLDS) {
            plan.issue_async(sBC, reinterpret_cast<const float*>(Bg), reinterpret_cast<const float*>(Cg), (int)p.B_ns, (int)p.C_ns,
                             0, N, j, L, N * TILE);
            lds_dma_wait();
        } else {
            stage_tile3<io_t>(sBC, Bg, Cg, p.B_ns, p.C_ns, N, j, L, REV, vec);
        }
        __syncthreads();

        for (int step = 0; step < RB; ++step) {
            const int rl = step * slots + slot;           // row inside the chunk
            const int r = row_c0 + rl;
            cold_args3_t kq = cold_args3();
            const int rpg = kq->f.rows_per_group;
            const int ur = r - ((g - (g >> kq->f.u_gshift)) * rpg);
            const int gr = r - ((g - (g >> kq->g_gshift)) * rpg);
            const io_t* __restrict__ u_row = reinterpret_cast<const io_t*>(kq->f.u) + (long)b * kq->f.u_bs + (long)ur * kq->f.u_ds;
            const io_t* __restrict__ d_row = reinterpret_cast<const io_t*>(kq->f.delta) + (long)b * kq->f.dt_bs + (long)r * kq->f.dt_ds;
            const io_t* __restrict__ g_row = reinterpret_cast<const io_t*>(kq->dout) + (long)b * kq->g_bs + (long)gr * kq->g_ds;
            const int pr = param_row(r, g, rpg, kq->f.pswap);
            const float bias = kq->f.bias ? kq->f.bias[pr] : 0.0f;

            // lane vectors (lane s = state nq0 + s): A[r, :], state entering the tile, reverse carry
            float Av = 0.0f, X0v = 0.0f, Rvv = 0.0f, rvout_v = 0.0f, dA_v = 0.0f;
            if (lane < 4) {
                Av = kq->f.A[(long)pr * kq->f.A_ds + (long)(nq0 + lane) * kq->f.A_ns];
                if (j > 0) X0v = kq->f.x[((long)b * kq->f.dim + r) * kq->f.x_rs + (long)(j - 1) * N + nq0 + lane];
                Rvv = sRv[rl * N + nq0 + lane];
            }
            float dl[T], dlu[T], gg[T], psx[T], psa[T];
            {
                float dv[T], uu[T];
                load_items<io_t, T, REV>(u_row, lbase, L, vec, uu);
                load_items<io_t, T, REV>(d_row, lbase, L, vec, dv);
                load_items<io_t, T, REV>(g_row, lbase, L, vec, gg);
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    float d = dv[k] + bias;
                    if (p.softplus) { float sg; d = softplus_ref(d, sg); }
                    d = (lbase + k < L) ? d : 0.0f;       // identity element past the end (a = 1, b = 0)
                    dl[k] = d;
                    dlu[k] = d * uu[k];
                    psx[k] = 0.0f;
                    psa[k] = 0.0f;
                }
            }
            float dsum = 0.0f;
#pragma unroll
            for (int k = 0; k < T; ++k) dsum += dl[k];

#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float An = lane_bcast(Av, s);
                const float A2 = An * kLog2e;
                const float x0 = lane_bcast(X0v, s);
                const float carry = lane_bcast(Rvv, s);
                const float* tB = sBC + (nq0 + s) * TILE;
                const float* tC = tB + N * TILE;
                float a[T], xs[T], gc[T], bq[T];
                // ---- forward: in-lane fold (lane 0 starts from the checkpoint), wave scan, replay
                float xa = lane0 ? x0 : 0.0f;
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    float bv[1], cv[1];
                    lds_read_chunk<T, REV>(tB, lane, k, bv);
                    lds_read_chunk<T, REV>(tC, lane, k, cv);
                    bq[k] = bv[0];
                    a[k] = fast_exp2(dl[k] * A2);
                    xs[k] = dlu[k] * bv[0];
                    gc[k] = gg[k] * cv[0];
                    xa = fmaf(a[k], xa, xs[k]);
                }
                const float plane = fast_exp2(A2 * dsum);
                float pf = plane;
                wave_mscan_inclusive(pf, xa);
                const float xstart = wave_prev_lane(xa, x0);
                {
                    float x = xstart;
#pragma unroll
                    for (int k = 0; k < T; ++k) { x = fmaf(a[k], x, xs[k]); xs[k] = x; }
                }
                // ---- reverse: e_k = a_k * dx_k, dx_k = g_k C_k + e_{k+1}; lane 63 starts from the carry
                float e = lane63 ? carry : 0.0f;
#pragma unroll
                for (int k = T - 1; k >= 0; --k) e = a[k] * (gc[k] + e);
                float pr = plane;
                wave_mscan_inclusive_rev(pr, e);
                e = wave_next_lane(e, carry);
                float dAp = 0.0f;
#pragma unroll
                for (int k = T - 1; k >= 0; --k) {
                    const float dx = gc[k] + e;
                    e = a[k] * dx;
                    psx[k] = fmaf(dx, bq[k], psx[k]);
                    const float t = e * (k > 0 ? xs[k > 0 ? k - 1 : 0] : xstart);    // dx * a_k * x_{k-1}
                    psa[k] = fmaf(An, t, psa[k]);
                    dAp = fmaf(dl[k], t, dAp);
                    accB[s][k] = fmaf(dx, dlu[k], accB[s][k]);    // this row's term of dB[n, l]
                    accC[s][k] = fmaf(gg[k], xs[k], accC[s][k]);  // this row's term of dC[n, l]
                }
                rvout_v = lane_put3(e, s, rvout_v);
                dA_v = lane_put3(wave_sum(dAp), s, dA_v);
                // The four states are unrolled only for static accumulator indices.  Left alone, hipcc sinks the
                // psx / psa updates of all four states below the last one and keeps dx, B and t of every state
                // alive until then (+15 registers per state: 143 instead of 87 VGPRs).  Pinning the sums here
                // costs no instruction.
#pragma unroll
                for (int k = 0; k < T; ++k) asm volatile("" : "+v"(psx[k]), "+v"(psa[k]));
            }

            cold_args3_t ke = cold_args3();
            if (lane < 4) {
                sRv[rl * N + nq0 + lane] = rvout_v;
                atomicAdd(ke->dA + (long)pr * ke->dA_ds + (long)(nq0 + lane) * ke->dA_ns, dA_v);
            }
            // ---- sum over the row's Q waves, then one of them finishes the row
            bool duty = true;
            if (Q > 1) {
                float* mine = sPart + ((par * nw + wave) * 2) * TILE + lane * T;
#pragma unroll
                for (int k = 0; k < T; ++k) { mine[k] = psx[k]; mine[TILE + k] = psa[k]; }
                lds_barrier();
                duty = quad == (step % Q);
                if (duty) {
                    // all reads of the Q partials in flight before the first add (Q <= 4)
                    const float* src0 = sPart + ((par * nw + slot * Q) * 2) * TILE + lane * T;
                    float px[4][T], pa[4][T];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float* src = src0 + (w < Q ? w : 0) * 2 * TILE;
#pragma unroll
                        for (int k = 0; k < T; ++k) { px[w][k] = src[k]; pa[w][k] = src[TILE + k]; }
                    }
#pragma unroll
                    for (int k = 0; k < T; ++k) {
                        float sx = px[0][k], sa = pa[0][k];
#pragma unroll
                        for (int w = 1; w < 4; ++w) { sx += (w < Q) ? px[w][k] : 0.0f; sa += (w < Q) ? pa[w][k] : 0.0f; }
                        psx[k] = sx; psa[k] = sa;
                    }
                }
                par ^= 1;
            }
            if (duty) {
                const int rpg2 = ke->f.rows_per_group;
                const int ur2 = r - ((g - (g >> ke->f.u_gshift)) * rpg2);
                const io_t* __restrict__ u_row2 = reinterpret_cast<const io_t*>(ke->f.u) + (long)b * ke->f.u_bs + (long)ur2 * ke->f.u_ds;
                const io_t* __restrict__ d_row2 = reinterpret_cast<const io_t*>(ke->f.delta) + (long)b * ke->f.dt_bs + (long)r * ke->f.dt_ds;
                const float Dd = ke->f.D ? ke->f.D[pr] : 0.0f;
                const float bias2 = ke->f.bias ? ke->f.bias[pr] : 0.0f;
                float duv[T], ddv[T], dv2[T], uu[T];
                float dD_acc = 0.0f, dbias_acc = 0.0f;
                load_items<io_t, T, REV>(d_row2, lbase, L, vec, dv2);
                load_items<io_t, T, REV>(u_row2, lbase, L, vec, uu);
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    duv[k] = fmaf(Dd, gg[k], dl[k] * psx[k]);
                    float dd = fmaf(uu[k], psx[k], psa[k]);
                    if (p.softplus) {
                        const float raw = dv2[k] + bias2;
                        const float ez = fast_exp2(raw * kLog2e);
                        dd *= (raw > 20.0f) ? 1.0f : ez * fast_rcp(1.0f + ez);
                    }
                    ddv[k] = dd;
                    if (lbase + k < L) { dD_acc = fmaf(gg[k], uu[k], dD_acc); dbias_acc += dd; }
                }
                io_t* __restrict__ du_row = reinterpret_cast<io_t*>(ke->du) + (long)b * ke->du_bs + (long)r * ke->du_ds;
                io_t* __restrict__ dd_row = reinterpret_cast<io_t*>(ke->ddelta) + (long)b * ke->dd_bs + (long)r * ke->dd_ds;
                store_items<io_t, T, REV>(du_row, lbase, L, vec, duv);
                store_items<io_t, T, REV>(dd_row, lbase, L, vec, ddv);
                if (ke->dD) { dD_acc = wave_sum(dD_acc); if (lane0) atomicAdd(ke->dD + pr, dD_acc); }
                if (ke->dbias) { dbias_acc = wave_sum(dbias_acc); if (lane0) atomicAdd(ke->dbias + pr, dbias_acc); }
            }
        }

        // ---- tile end: the SLOTS partial dB/dC sums of every state meet in LDS (one array at a time)
#pragma unroll
        for (int arr = 0; arr < 2; ++arr) {
            __syncthreads();                                  // row partials / previous array consumed
            {
                float* mine = sPart + (wave * 4) * TILE + lane * T;          // [nw][4][TILE], position order
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int k = 0; k < T; ++k) mine[s * TILE + k] = arr == 0 ? accB[s][k] : accC[s][k];
            }
            __syncthreads();
            float* __restrict__ obase = arr == 0 ? oB : oC;
            const long o_ns = arr == 0 ? o_nsB : o_nsC;
            for (int idx = tid; idx < N * TILE; idx += blockDim.x) {
                const int n = idx / TILE;
                const int pp = idx - n * TILE;                // scan position inside the tile
                const int qd = n >> 2, s = n & 3;
                float sum = 0.0f;
                const float* colp = sPart + (qd * 4 + s) * TILE + pp;
                const int cstride = Q * 4 * TILE;
                int sl = 0;
                for (; sl + 4 <= slots; sl += 4) {
                    const float v0 = colp[sl * cstride], v1 = colp[(sl + 1) * cstride], v2 = colp[(sl + 2) * cstride],
                                v3 = colp[(sl + 3) * cstride];
                    sum += v0; sum += v1; sum += v2; sum += v3;
                }
                for (; sl < slots; ++sl) sum += colp[sl * cstride];
                const int m = REV ? (L - 1 - l0 - pp) : (l0 + pp);
                if (m >= 0 && m < L) obase[(long)n * o_ns + m] = sum;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int k = 0; k < T; ++k) { accB[s][k] = 0.0f; accC[s][k] = 0.0f; }
        __syncthreads();                                      // sPart and sBC free for the next tile
        par = 0;
    }
}

// MAXW = 16: 128-VGPR budget (4 waves per SIMD); MAXW = 12: 168 VGPRs (3 waves per SIMD, no spills)
template <typename io_t, bool GLDS, int MAXW>
__global__ void __launch_bounds__(64 * MAXW)
scan_bwd3_kernel(const BwdArgs q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int per_b = q.f.G * q.P;
    const int b = lb / per_b;
    const int rem = lb - b * per_b;
    const int g = rem / q.P;
    const int chunk = rem - g * q.P;
    if ((q.f.rev_mask >> g) & 1u) scan_bwd3_body<io_t, GLDS, true>(q, smem, b, g, chunk);
    else scan_bwd3_body<io_t, GLDS, false>(q, smem, b, g, chunk);
}

template <typename io_t, bool GLDS, int MAXW>
static hipError_t launch_bwd3_w(const BwdArgs& a, hipStream_t stream) {
    const int nw = a.f.R;                                  // waves per workgroup = slots * Q
    const size_t lds = bwd3_lds_bytes(nw, a.f.N, a.RB);
    const int grid = a.f.batch * a.f.G * a.P;
    auto kern = scan_bwd3_kernel<io_t, GLDS, MAXW>;
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
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nw * 64), lds, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || a.P == 1) return e;
    return launch_reduce_partials(a, stream);
}

template <typename io_t, bool GLDS>
static hipError_t launch_bwd3_t(const BwdArgs& a, hipStream_t stream) {
    return a.f.R > 12 ? launch_bwd3_w<io_t, GLDS, 16>(a, stream) : launch_bwd3_w<io_t, GLDS, 12>(a, stream);
}

hipError_t launch_scan_bwd3(const BwdArgs& a, int dtype, bool glds, hipStream_t stream) {
    switch (dtype) {
        case 0: return glds ? launch_bwd3_t<float, true>(a, stream) : launch_bwd3_t<float, false>(a, stream);
        case 1: return launch_bwd3_t<f16_t, false>(a, stream);
        case 2: return launch_bwd3_t<bf16_t, false>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace sigma
