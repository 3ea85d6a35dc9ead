// scan_bwd2.hip -- selective-scan backward, second generation (gfx950 / MI355X, wave64).
//
// Replaces selective_scan_bwd_kernel + reverse_scan.cuh (reference:
// models/encoders/selective_scan/csrc/selective_scan/selective_scan_bwd_kernel.cuh:66-308,
// reverse_scan.cuh:18-401) for the cases the fused model path produces: one state checkpoint per
// backward tile (ckpt_pitch == 64*T) and dstate <= 64.  Everything else stays on scan_bwd.hip.
// Mathematics: SURVEY.md App. E.2.
//
// What changed against scan_bwd.hip, and why (profiles/r01_final_pmc_enc_s2_b16.txt):
//   * ROW-BLOCK LOOP: a workgroup owns RB*R rows of one (batch, group) and walks them R at a time
//     for every tile, so the dB/dC partial sums of RB row blocks meet in LDS (one float2 per state
//     and column, private to the column-sum thread that owns the column) instead of in P = rows/R
//     global slabs.  The workspace shrinks by RB (64 -> 8 slabs on the dominant shape) or disappears;
//   * the wave scans carry the decay as a product (v_fmac_f32_dpp + v_mul_f32_dpp without
//     bound_ctrl): no v_exp_f32 inside a scan step (8 cycles each vs 2.25 for a plain VALU op,
//     tools/ubench), ONE exp2 per lane and state for the lane's decay product, shared by the
//     forward and the reverse scan;
//   * the incoming state of a tile (checkpoint) and the reverse carry of the tile to the right
//     are folded into the in-lane folds of lane 0 / lane 63, so no exclusive decay products and no
//     per-state LDS traffic for them: per row and tile they travel as lane vectors (lane n =
//     state n) and are picked with v_readlane / collected with v_writelane;
//   * softplus is evaluated once per element and tile (its derivative is rebuilt from the raw
//     delta in the epilogue), a_k * x_{k-1} is formed as e_k * x_{k-1} (one multiply less);
//   * dA / dD / ddelta_bias leave through one atomicAdd per (row, tile) and state.
// Kept: lane-blocked T-element segments, global_load_lds double-buffered B/C staging, per-wave
// LDS slabs + fixed-order column sums for the row reduction (deterministic; LDS float atomics
// measured 256 cycles per wave instruction: dead), reversed groups by addressing.
#include "scan_device.h"
#include "scan_launch.h"

#include <atomic>

// Phase timing for development builds (python -m sigma_amd.build --variant prof --flags=-DSIGMA_BWD2_PROF=1):
// every wave accumulates s_memtime deltas per phase and adds them to g_bwd2_prof at the end
// (sigma_scan_debug_read in capi.hip).  Costs ~10 % run time; compiled out of the product build.
#ifndef SIGMA_BWD2_PROF
#define SIGMA_BWD2_PROF 0
#endif
#if SIGMA_BWD2_PROF
__device__ unsigned long long g_bwd2_prof[16];
#define PROF_DECL long long prof_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long prof_last = __builtin_readcyclecounter();
#define PROF(i) { const long long t_ = __builtin_readcyclecounter(); prof_t[i] += t_ - prof_last; prof_last = t_; }
#define PROF_FLUSH if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 10; ++i_) atomicAdd(&g_bwd2_prof[i_], (unsigned long long)prof_t[i_]); atomicAdd(&g_bwd2_prof[15], 1ull); }
#else
#define PROF_DECL
#define PROF(i)
#define PROF_FLUSH
#endif

namespace sigma {

#if SIGMA_BWD2_PROF
hipError_t bwd2_prof_read(unsigned long long* out16) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_bwd2_prof), 16 * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bwd2_prof), z, sizeof(z));
}
#else
hipError_t bwd2_prof_read(unsigned long long* out16) { for (int i = 0; i < 16; ++i) out16[i] = 0; return hipSuccess; }
#endif

namespace {

// Register path of the B/C staging (16-bit IO types, unaligned tensors): states [n0, n0+nbn) of ONE
// tile into dst laid out [arr][NB][TILE]; the f32 / aligned case uses StagePlan (scan_device.h).
template <typename io_t, int T>
__device__ __forceinline__ void stage_tile2(float* __restrict__ dst, const io_t* __restrict__ Bg,
                                            const io_t* __restrict__ Cg, long B_ns, long C_ns, int n0, int nbn, int NB,
                                            int tile, int L, bool rev, bool vec) {
    constexpr int TILE = 64 * T;
    constexpr int CPR = TILE / 4;
    const int total = 2 * NB * CPR;
    const int l0 = tile * TILE;
    for (int ci = threadIdx.x; ci < total; ci += blockDim.x) {
        const int row = ci / CPR;                      // arr * NB + nn
        const int c4 = (ci - row * CPR) * 4;
        const int arr = row / NB;
        const int nn = row - arr * NB;
        const int m = rev ? (L - l0 - TILE + c4) : (l0 + c4);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (nn < nbn && m < L && m + 4 > 0) {
            const io_t* __restrict__ srow = arr == 0 ? Bg + (long)(n0 + nn) * B_ns : Cg + (long)(n0 + nn) * C_ns;
            load4_guard<io_t>(srow, m, L, vec, v);
        }
        *reinterpret_cast<float4*>(dst + (long)ci * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

__device__ __forceinline__ float lane_pick(float v, int n) {          // v_readlane with a uniform lane index
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), n));
}
// lane n of `old` <- a wave-uniform value (v_writelane_b32; no clang builtin in this toolchain).  The
// s_nop covers the "VALU writes SGPR -> v_writelane uses it" wait states for the readfirstlane result.
__device__ __forceinline__ float lane_put(float uniform_val, int n, float old) {
    const int sval = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, uniform_val));
    int keep;      // gfx9 VALU reads one SGPR: the lane select goes through M0 (saved: the compiler owns it)
    asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                 : "+v"(old), "=&s"(keep) : "s"(sval), "s"(n));
    return old;
}

// The parameter block is ~110 SGPRs wide; kept live across the state loop it spills into VGPR lanes
// (v1: 293 v_writelane / 1411 v_readlane of spill code).  Fields that are only needed once per
// (row, tile) -- row pointers, strides, the small per-row outputs -- are therefore read through the
// kernarg segment pointer, laundered so that the loads stay where they are used (s_load, ~15 per
// row and tile) instead of being hoisted out of every loop.
typedef const __attribute__((address_space(4))) BwdArgs* cold_args_t;   // constant address space: s_load
__device__ __forceinline__ cold_args_t cold_args() {
    cold_args_t kp = (cold_args_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return kp;
}

// L2 warm-up of the NEXT tile's u / delta / dout segments of a row (2.5 KB each at T = 10): lanes 0..nl-1 touch
// one 128-byte line each.  Issued as LDS-DMA into a 256-byte dummy area: no VGPR destination, so nothing the
// compiler could reuse while the load is in flight (an asm load INTO a register counts as written at once and
// its register was re-used as the next address: memory faults).  Untracked like the B/C stream; the caller's
// lds_dma_wait() at the end of the staging block retires them.  Turns the ~2 us HBM miss at the top of the next
// row step into an L2 hit.
__device__ __forceinline__ void touch_lines(const void* seg, int nbytes, int lane, unsigned lds_dummy) {
    const char* pa = reinterpret_cast<const char*>(seg) + lane * 128;
    if (lane * 128 < nbytes) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(pa), "s"(lds_dummy) : "memory");
    }
}

// Sum of one float2 column over the RR row slabs, fixed order w = 0 .. RR-1.  All RR reads are issued
// before the first add: with a run-time trip count hipcc emits read / s_waitcnt lgkmcnt(0) / add per
// slab, i.e. RR LDS latencies back to back (~2000 cycles per state at RR = 16, profiles/r02_bwd2_phases.txt).
template <int RR>
__device__ __forceinline__ float2 colsum_fixed(const float* __restrict__ colp, int stride) {
    float2 v[RR];
#pragma unroll
    for (int w = 0; w < RR; ++w) v[w] = *reinterpret_cast<const float2*>(colp + w * stride);
    float2 s = make_float2(0.f, 0.f);
#pragma unroll
    for (int w = 0; w < RR; ++w) { s.x += v[w].x; s.y += v[w].y; }
    return s;
}

__device__ __forceinline__ float2 colsum(const float* __restrict__ colp, int stride, int R) {
    // batches of 8 reads in flight (16 temporaries): more would push the 128-VGPR build into spills
    float2 s = make_float2(0.f, 0.f);
    int w = 0;
    for (; w + 8 <= R; w += 8) {
        const float2 t = colsum_fixed<8>(colp + w * stride, stride);
        s.x += t.x; s.y += t.y;
    }
    if (w + 4 <= R) {
        const float2 t = colsum_fixed<4>(colp + w * stride, stride);
        s.x += t.x; s.y += t.y;
        w += 4;
    }
    for (; w < R; ++w) {
        const float2 v = *reinterpret_cast<const float2*>(colp + w * stride);
        s.x += v.x; s.y += v.y;
    }
    return s;
}

}  // namespace

template <typename io_t, int T, bool GLDS, bool REV>
__device__ __forceinline__ void scan_bwd2_body(const BwdArgs& q, float* smem, int b, int g, int chunk) {
    constexpr int TILE = 64 * T;
    constexpr int VW = vec_width<T>::value;
    const FwdArgs& p = q.f;
    const int R = blockDim.x >> 6;
    const int N = p.N, L = p.L, NB = p.NB, RB = q.RB;
    const int nslab = q.slab2 ? 2 : 1;
    const int bufsz = 2 * NB * TILE;
    float* sBC = smem;                                // [2][2][NB][TILE]
    float* sRed = sBC + 2 * bufsz;                    // [nslab][R][2][TILE] per-row dB/dC terms of one state
    float* sRv = sRed + nslab * R * 2 * TILE;         // [RB*R][N] reverse carry a*dx of the tile to the right

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool vec = p.vec_ok != 0;
    const bool lane0 = lane == 0, lane63 = lane == 63;
    const int row_c0 = g * p.rows_per_group + chunk * RB * R;       // first row of this workgroup's chunk

    const io_t* __restrict__ Bg = reinterpret_cast<const io_t*>(p.B) + (long)b * p.B_bs + (long)g * p.B_gs;
    const io_t* __restrict__ Cg = reinterpret_cast<const io_t*>(p.C) + (long)b * p.C_bs + (long)g * p.C_gs;
    // where this workgroup's dB/dC sums go: the tensors themselves (it owns the whole group) or
    // its slab of the caller's workspace, summed over the P chunks by reduce_partials_kernel
    float* __restrict__ oB;
    float* __restrict__ oC;
    long o_nsB, o_nsC;
    bool o_vec;
    if (q.P == 1) {
        oB = q.dB + (long)b * q.dB_bs + (long)g * q.dB_gs; o_nsB = q.dB_ns;
        oC = q.dC + (long)b * q.dC_bs + (long)g * q.dC_gs; o_nsC = q.dC_ns;
        o_vec = q.out_vec_ok != 0 && (L & 1) == 0;
    } else {
        const long slab = (((long)chunk * p.batch + b) * p.G + g) * (long)N * L;
        oB = q.ws_dB + slab; oC = q.ws_dC + slab; o_nsB = L; o_nsC = L;
        o_vec = (L & 1) == 0;
    }

    for (int i = tid; i < RB * R * N; i += blockDim.x) sRv[i] = 0.0f;
    // never multiply uninitialised LDS bits (stale/NaN) into the padding of the last tile
    for (int i = tid; i < 2 * bufsz / 4; i += blockDim.x) reinterpret_cast<float4*>(sBC)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    const int nsb = (N + NB - 1) / NB;
    const int ntiles = (L + TILE - 1) / TILE;
    StagePlan<T, REV> plan;
    if constexpr (GLDS) plan.init(NB, 1, L);
    auto stage = [&](int step, int tile, int sb) {
        const int n0 = sb * NB;
        const int nbn = (N - n0 < NB) ? (N - n0) : NB;
        float* dst = sBC + (step & 1) * bufsz;
        if constexpr (GLDS) {
            plan.issue_async(dst, reinterpret_cast<const float*>(Bg), reinterpret_cast<const float*>(Cg), (int)p.B_ns,
                             (int)p.C_ns, n0, nbn, tile, L, NB * TILE);
        } else {
            stage_tile2<io_t, T>(dst, Bg, Cg, p.B_ns, p.C_ns, n0, nbn, NB, tile, L, REV, vec);
        }
    };

    // column-sum geometry: thread t < TILE owns the float2 at positions (pp, pp + 1) of array c
    // (0: dB, 1: dC) of every state, in scan-position order; its image (memory) offset follows
    const bool col_on = tid < TILE;
    const int col_c = tid / (TILE / 2);
    const int col_pp = (tid - col_c * (TILE / 2)) * 2;
    // RB > 1: the column sums of the row blocks of a tile meet in LDS, one float2 per state and column,
    // touched only by the thread that owns the column (no barrier, no atomics): the first row block
    // stores, the middle ones add, the last one adds and writes the total to memory.
    float* sAcc = sRv + RB * R * N + 64;               // [N][2][TILE] when RB > 1

    // write one float2 column of state n (positions pp, pp+1 of the tile starting at l0)
    auto put_col = [&](int n, int l0, float2 v) {
        const int m = REV ? (L - 2 - l0 - col_pp) : (l0 + col_pp);      // lower memory index of the pair
        if (REV) v = make_float2(v.y, v.x);
        float* __restrict__ dst = (col_c == 0 ? oB + (long)n * o_nsB : oC + (long)n * o_nsC) + m;
        if (o_vec && m >= 0 && m + 2 <= L) {
            *reinterpret_cast<float2*>(dst) = v;
        } else {
            if (m >= 0 && m < L) dst[0] = v.x;
            if (m + 1 >= 0 && m + 1 < L) dst[1] = v.y;
        }
    };

    int step = 0;                                      // staging buffer parity
    int cnt = 0;                                       // states processed (slab parity)
    PROF_DECL
    stage(0, ntiles - 1, 0);
    if constexpr (GLDS) lds_dma_wait();
    __syncthreads();

    PROF(0)
    for (int j = ntiles - 1; j >= 0; --j) {
        const int l0 = j * TILE;
        const int lbase = l0 + lane * T;
        for (int rb = 0; rb < RB; ++rb) {
            const int rl = rb * R + wave;              // row inside the chunk
            const int r = row_c0 + rl;
            cold_args_t kq = cold_args();
            const int rpg = kq->f.rows_per_group;
            const int ur = r - ((g - (g >> kq->f.u_gshift)) * rpg);   // same row of group g >> u_gshift
            const int gr = r - ((g - (g >> kq->g_gshift)) * rpg);
            const io_t* __restrict__ u_row = reinterpret_cast<const io_t*>(kq->f.u) + (long)b * kq->f.u_bs + (long)ur * kq->f.u_ds;
            const io_t* __restrict__ d_row = reinterpret_cast<const io_t*>(kq->f.delta) + (long)b * kq->f.dt_bs + (long)r * kq->f.dt_ds;
            const io_t* __restrict__ g_row = reinterpret_cast<const io_t*>(kq->dout) + (long)b * kq->g_bs + (long)gr * kq->g_ds;
            const int pr = param_row(r, g, rpg, kq->f.pswap);
            const float bias = kq->f.bias ? kq->f.bias[pr] : 0.0f;

            // lane vectors (lane n = state n): A[r, :], the state entering the tile, the reverse carry.
            // Loaded first: older than the row loads below, so complete once those have been consumed.
            float Av = 0.0f, X0v = 0.0f, Rvv = 0.0f, rvout_v = 0.0f, dA_v = 0.0f;
            if (lane < N) {
                Av = kq->f.A[(long)pr * kq->f.A_ds + (long)lane * kq->f.A_ns];
                if (j > 0) X0v = kq->f.x[((long)b * kq->f.dim + r) * kq->f.x_rs + (long)(j - 1) * N + lane];
                Rvv = sRv[rl * N + lane];
            }
            float dl[T], dlu[T], gg[T], sdxB[T], sAx[T];
            {
                float dv[T], uu[T];
                load_items<io_t, T, REV>(u_row, lbase, L, vec, uu);
                load_items<io_t, T, REV>(d_row, lbase, L, vec, dv);
                load_items<io_t, T, REV>(g_row, lbase, L, vec, gg);
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    float d = dv[k] + bias;
                    if (p.softplus) { float sg; d = softplus_ref(d, sg); }
                    d = (lbase + k < L) ? d : 0.0f;    // identity element past the end (a = 1, b = 0)
                    dl[k] = d;
                    dlu[k] = d * uu[k];
                    sdxB[k] = 0.0f;
                    sAx[k] = 0.0f;
                }
            }
            float dsum = 0.0f;
#pragma unroll
            for (int k = 0; k < T; ++k) dsum += dl[k];

            if (GLDS && !(q.flags & 1)) {
                const unsigned touch_sink = (unsigned)(uintptr_t)(lptr_t)(sRv + RB * R * N);      // 64 dummy floats
                // next step of THIS wave: same tile, next row block -- or the tile to the left of its first row
                const int jn = (rb + 1 < RB) ? j : j - 1;
                if (jn >= 0) {
                    const int rn = (rb + 1 < RB) ? r + R : row_c0 + wave;
                    const int urn = rn - ((g - (g >> kq->f.u_gshift)) * rpg);
                    const int grn = rn - ((g - (g >> kq->g_gshift)) * rpg);
                    const int l0n = jn * TILE;
                    const int m0 = REV ? (L - l0n - TILE < 0 ? 0 : L - l0n - TILE) : l0n;         // first memory element
                    const int m1 = REV ? L - l0n : (l0n + TILE < L ? l0n + TILE : L);
                    const int nb = (m1 - m0) * (int)sizeof(io_t);
                    touch_lines(reinterpret_cast<const io_t*>(kq->f.u) + (long)b * kq->f.u_bs + (long)urn * kq->f.u_ds + m0, nb, lane, touch_sink);
                    touch_lines(reinterpret_cast<const io_t*>(kq->f.delta) + (long)b * kq->f.dt_bs + (long)rn * kq->f.dt_ds + m0, nb, lane, touch_sink);
                    touch_lines(reinterpret_cast<const io_t*>(kq->dout) + (long)b * kq->g_bs + (long)grn * kq->g_ds + m0, nb, lane, touch_sink);
                }
            }
            PROF(1)                                            // row prologue: loads, softplus
            for (int sb = 0; sb < nsb; ++sb) {
                const float* cur = sBC + (step & 1) * bufsz;
                if (sb + 1 < nsb) stage(step + 1, j, sb + 1);
                else if (rb + 1 < RB) stage(step + 1, j, 0);
                else if (j > 0) stage(step + 1, j - 1, 0);
                PROF(2)                                        // B/C stage issue
                const int n0 = sb * NB;
                const int nend = (N - n0 < NB) ? (N - n0) : NB;
#pragma unroll 1
                for (int nn = 0; nn < nend; ++nn) {
                    const int n = n0 + nn;
                    const float An = lane_pick(Av, n);
                    const float A2 = An * kLog2e;
                    const float x0 = lane_pick(X0v, n);
                    const float carry = lane_pick(Rvv, n);
                    const float* tB = cur + nn * TILE;
                    const float* tC = tB + NB * TILE;
                    float a[T], xs[T], gc[T];
                    // ---- forward: in-lane fold (lane 0 starts from the checkpoint), wave scan, replay
                    float xa = lane0 ? x0 : 0.0f;
#pragma unroll
                    for (int qq = 0; qq < T / VW; ++qq) {
                        float bq[VW], cq[VW];
                        lds_read_chunk<T, REV>(tB, lane, qq, bq);
                        lds_read_chunk<T, REV>(tC, lane, qq, cq);
#pragma unroll
                        for (int jj = 0; jj < VW; ++jj) {
                            const int k = VW * qq + jj;
                            a[k] = fast_exp2(dl[k] * A2);
                            xs[k] = dlu[k] * bq[jj];
                            gc[k] = gg[k] * cq[jj];
                            xa = fmaf(a[k], xa, xs[k]);
                        }
                    }
                    const float plane = fast_exp2(A2 * dsum);     // this lane's decay product
                    float pf = plane;
                    wave_mscan_inclusive(pf, xa);
                    const float xstart = wave_prev_lane(xa, x0);   // state entering the lane
                    {
                        float x = xstart;
#pragma unroll
                        for (int k = 0; k < T; ++k) { x = fmaf(a[k], x, xs[k]); xs[k] = x; }
                    }
                    PROF(3)                                    // forward: LDS reads, fold, scan, replay
                    // ---- reverse: e_k = a_k * dx_k, dx_k = g_k C_k + e_{k+1}; lane 63 starts from the carry
                    float e = lane63 ? carry : 0.0f;
#pragma unroll
                    for (int k = T - 1; k >= 0; --k) e = a[k] * (gc[k] + e);
                    float pr = plane;
                    wave_mscan_inclusive_rev(pr, e);
                    e = wave_next_lane(e, carry);                  // e entering the lane from the right
                    float dAp = 0.0f;
                    float* sRedN = sRed + ((cnt & (nslab - 1)) ? R * 2 * TILE : 0);
                    PROF(4)                                    // reverse fold + scan
                    if (nslab == 1) lds_barrier();                 // slab free (previous state summed)
                    {
                        float* __restrict__ slab = sRedN + wave * 2 * TILE + lane * T;
#pragma unroll
                        for (int qq = T / VW - 1; qq >= 0; --qq) {
                            float bq[VW], vb[VW], vc[VW];
                            lds_read_chunk<T, REV>(tB, lane, qq, bq);    // B again: cheaper than T live registers
#pragma unroll
                            for (int jj = VW - 1; jj >= 0; --jj) {
                                const int k = VW * qq + jj;
                                const float dx = gc[k] + e;
                                e = a[k] * dx;
                                sdxB[k] = fmaf(dx, bq[jj], sdxB[k]);
                                const float t = e * (k > 0 ? xs[k > 0 ? k - 1 : 0] : xstart);   // dx * a_k * x_{k-1}
                                sAx[k] = fmaf(An, t, sAx[k]);
                                dAp = fmaf(dl[k], t, dAp);
                                vb[jj] = dx * dlu[k];                    // this row's term of dB[n, l]
                                vc[jj] = gg[k] * xs[k];                  // this row's term of dC[n, l]
                            }
                            if constexpr (VW == 4) {
                                *reinterpret_cast<float4*>(slab + 4 * qq) = make_float4(vb[0], vb[1], vb[2], vb[3]);
                                *reinterpret_cast<float4*>(slab + TILE + 4 * qq) = make_float4(vc[0], vc[1], vc[2], vc[3]);
                            } else if constexpr (VW == 2) {
                                *reinterpret_cast<float2*>(slab + 2 * qq) = make_float2(vb[0], vb[1]);
                                *reinterpret_cast<float2*>(slab + TILE + 2 * qq) = make_float2(vc[0], vc[1]);
                            } else {
                                slab[qq] = vb[0];
                                slab[TILE + qq] = vc[0];
                            }
                        }
                    }
                    rvout_v = lane_put(e, n, rvout_v);             // lane 0: a*dx of the tile's first element
                    dA_v = lane_put(wave_sum(dAp), n, dA_v);
                    PROF(5)                                    // reverse replay, slab writes, dA sum
                    // slabs complete; at the end of a staging block also "next B/C block landed"
                    if (nn == nend - 1) { if constexpr (GLDS) lds_dma_wait(); __syncthreads(); } else lds_barrier();
                    PROF(6)                                    // barrier wait
                    if (col_on) {
                        const float2 s = colsum(sRedN + col_c * TILE + col_pp, 2 * TILE, R);
                        if (RB > 1) {
                            float2* ap = reinterpret_cast<float2*>(sAcc + (n * 2 + col_c) * TILE + col_pp);
                            if (rb == 0) {
                                *ap = s;
                            } else {
                                float2 t = *ap;
                                t.x += s.x; t.y += s.y;
                                if (rb == RB - 1) put_col(n, l0, t); else *ap = t;
                            }
                        } else {
                            put_col(n, l0, s);
                        }
                    }
                    ++cnt;
                    PROF(7)                                    // column sums
                }
                ++step;
            }

            // ---- per-row results of this tile (cold parameters re-read here, see cold_args())
            cold_args_t ke = cold_args();
            const int rpg2 = ke->f.rows_per_group;
            const int ur2 = r - ((g - (g >> ke->f.u_gshift)) * rpg2);
            const io_t* __restrict__ u_row2 = reinterpret_cast<const io_t*>(ke->f.u) + (long)b * ke->f.u_bs + (long)ur2 * ke->f.u_ds;
            const io_t* __restrict__ d_row2 = reinterpret_cast<const io_t*>(ke->f.delta) + (long)b * ke->f.dt_bs + (long)r * ke->f.dt_ds;
            if (lane < N) {
                sRv[rl * N + lane] = rvout_v;
                atomicAdd(ke->dA + (long)pr * ke->dA_ds + (long)lane * ke->dA_ns, dA_v);
            }
            float duv[T], ddv[T];
            float dD_acc = 0.0f, dbias_acc = 0.0f;
            {
                // softplus' = sigmoid(raw) and the u factors: re-read delta and u (L2-resident) instead
                // of holding 2T registers across the whole state loop
                const float Dd = ke->f.D ? ke->f.D[pr] : 0.0f;
                const float bias2 = ke->f.bias ? ke->f.bias[pr] : 0.0f;
                float dv2[T], uu[T];
                load_items<io_t, T, REV>(d_row2, lbase, L, vec, dv2);
                load_items<io_t, T, REV>(u_row2, lbase, L, vec, uu);
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    duv[k] = fmaf(Dd, gg[k], dl[k] * sdxB[k]);
                    float dd = fmaf(uu[k], sdxB[k], sAx[k]);
                    if (p.softplus) {
                        const float raw = dv2[k] + bias2;
                        const float ez = fast_exp2(raw * kLog2e);
                        dd *= (raw > 20.0f) ? 1.0f : ez * fast_rcp(1.0f + ez);
                    }
                    ddv[k] = dd;
                    if (lbase + k < L) { dD_acc = fmaf(gg[k], uu[k], dD_acc); dbias_acc += dd; }
                }
            }
            io_t* __restrict__ du_row = reinterpret_cast<io_t*>(ke->du) + (long)b * ke->du_bs + (long)r * ke->du_ds;
            io_t* __restrict__ dd_row = reinterpret_cast<io_t*>(ke->ddelta) + (long)b * ke->dd_bs + (long)r * ke->dd_ds;
            store_items<io_t, T, REV>(du_row, lbase, L, vec, duv);
            store_items<io_t, T, REV>(dd_row, lbase, L, vec, ddv);
            if (ke->dD) { dD_acc = wave_sum(dD_acc); if (lane0) atomicAdd(ke->dD + pr, dD_acc); }
            if (ke->dbias) { dbias_acc = wave_sum(dbias_acc); if (lane0) atomicAdd(ke->dbias + pr, dbias_acc); }
            PROF(8)                                            // row epilogue
        }
    }
    PROF_FLUSH
}

template <typename io_t, int T, bool GLDS, int MAXW>
__global__ void __launch_bounds__(64 * MAXW)
scan_bwd2_kernel(const BwdArgs q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int per_b = q.f.G * q.P;                    // workgroups per batch entry
    const int b = lb / per_b;
    const int rem = lb - b * per_b;
    const int g = rem / q.P;
    const int chunk = rem - g * q.P;
    if ((q.f.rev_mask >> g) & 1u) scan_bwd2_body<io_t, T, GLDS, true>(q, smem, b, g, chunk);
    else scan_bwd2_body<io_t, T, GLDS, false>(q, smem, b, g, chunk);
}

template <typename io_t, int T, bool GLDS, int MAXW>
static hipError_t launch_bwd2_t(const BwdArgs& a, hipStream_t stream) {
    const size_t lds = bwd2_lds_bytes(T, a.f.R, a.f.NB, a.f.N, a.slab2 != 0, a.RB);
    const int grid = a.f.batch * a.f.G * a.P;
    auto kern = scan_bwd2_kernel<io_t, T, GLDS, MAXW>;
    // raise the dynamic-LDS cap per device and kernel (the attribute is per device)
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
    hipLaunchKernelGGL(kern, dim3(grid), dim3(a.f.R * 64), lds, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || a.P == 1) return e;
    return launch_reduce_partials(a, stream);
}

template <typename io_t, int T, bool GLDS>
static hipError_t launch_bwd2_acc(const BwdArgs& a, hipStream_t stream) {
    // 13..16 waves: 128-VGPR build; up to 12 waves: 168 VGPRs
    return a.f.R > 12 ? launch_bwd2_t<io_t, T, GLDS, 16>(a, stream) : launch_bwd2_t<io_t, T, GLDS, 12>(a, stream);
}

template <typename io_t, bool GLDS>
static hipError_t launch_bwd2_io(const BwdArgs& a, int T, hipStream_t stream) {
    switch (T) {
        case 5: return launch_bwd2_acc<io_t, 5, GLDS>(a, stream);
        case 10: return launch_bwd2_acc<io_t, 10, GLDS>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_scan_bwd2(const BwdArgs& a, int dtype, int T, bool glds, hipStream_t stream) {
    switch (dtype) {
        case 0: return glds ? launch_bwd2_io<float, true>(a, T, stream) : launch_bwd2_io<float, false>(a, T, stream);
        case 1: return launch_bwd2_io<f16_t, false>(a, T, stream);
        case 2: return launch_bwd2_io<bf16_t, false>(a, T, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace sigma
