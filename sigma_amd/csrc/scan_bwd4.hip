// scan_bwd4.hip -- selective-scan backward, "quad-row" mapping (gfx950 / MI355X, wave64, f32 IO).
//
// Same operator as scan_bwd2.hip (reference: models/encoders/selective_scan/csrc/selective_scan/
// selective_scan_bwd_kernel.cuh:66-308, reverse_scan.cuh:18-401; mathematics SURVEY.md App. E.2), other
// mapping of rows to lanes.  The round-2 counters of scan_bwd2 (profiles/r02_pmc_enc_s2_b16.txt) show a
// kernel bound by VALU issue plus LDS traffic, both dominated by overheads of the 64-lane tile:
//   * two 6-step wave scans per state and T elements (32 DPP instructions incl. the cross-row steps),
//   * 8 bytes of LDS written and 8 read per element-state for the dB/dC row reduction (every wave hands its
//     row's terms to the column-sum threads), the column sums themselves, one barrier per state.
// Here a wave is FOUR rows x 16 lanes x T = 10 positions (tile = 160 positions, one DPP row per channel row):
//   * the scans stay inside a DPP row: 4 steps (row_shr / row_shl), no row_bcast / readlane fix-ups;
//   * the dB/dC terms of the wave's four rows are summed IN REGISTERS with the gfx950 lane-swap instructions
//     (v_permlane32_swap: upper half of one register <-> lower half of another; v_permlane16_swap: odd DPP
//     rows <-> even DPP rows; semantics pinned by tools/ubench/lane_ops_probe.hip): 20 registers of per-row
//     terms become 5 registers of four-row sums in which every lane carries a distinct column, so a wave
//     writes 5 dwords per lane and state instead of 20, and the column sums read a quarter as much;
//   * the slabs are small (W x 320 floats per state), so several states share one barrier ("SB");
//   * the B/C image of a tile (all N states, 2*N*160 floats) is staged once per tile for all row blocks of
//     the workgroup, by LDS-DMA, while the previous tile is being processed;
//   * per-row scalars of a state (A, checkpoint, reverse carry) live in lane vectors (lane 16*row + n) and
//     are broadcast inside their DPP row with ds_bpermute_b32; the per-state results (outgoing carry, dA)
//     are collected with a select + row rotate.
// Needs: f32 IO, B/C eligible for global_load_lds, dstate in {4,8,16}, ckpt_pitch 160, rows per group
// divisible by 4*W.  Everything else stays on scan_bwd2.hip / scan_bwd.hip.
#include "scan_device.h"
#include "scan_launch.h"
#include "scan_quad.h"

#include <atomic>

// phase timing of development builds, as in scan_bwd2.hip (-DSIGMA_BWD2_PROF=1; tools/bwd2_prof.py)
#ifndef SIGMA_BWD2_PROF
#define SIGMA_BWD2_PROF 0
#endif
#if SIGMA_BWD2_PROF
__device__ unsigned long long g_bwd4_prof[16];
#define PROF_DECL long long prof_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long prof_last = __builtin_readcyclecounter();
#define PROF(i) { const long long t_ = __builtin_readcyclecounter(); prof_t[i] += t_ - prof_last; prof_last = t_; }
#define PROF_FLUSH if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 10; ++i_) atomicAdd(&g_bwd4_prof[i_], (unsigned long long)prof_t[i_]); atomicAdd(&g_bwd4_prof[15], 1ull); }
#else
#define PROF_DECL
#define PROF(i)
#define PROF_FLUSH
#endif

// Ablation builds (-DSIGMA_BWD4_ABL=<bits>; results are WRONG, timing only; profiles/r03_bwd4_ablation.txt):
//   1 no du / ddelta stores   2 no dA / dD / ddelta_bias atomics   4 no column sums (barriers stay)
//   8 no barriers and no column sums   16 no four-row folds / slab writes   32 no row prologue loads (constants)
#ifndef SIGMA_BWD4_ABL
#define SIGMA_BWD4_ABL 0
#endif

namespace sigma {

#if SIGMA_BWD2_PROF
hipError_t bwd4_prof_read(unsigned long long* out16) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_bwd4_prof), 16 * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bwd4_prof), z, sizeof(z));
}
#else
hipError_t bwd4_prof_read(unsigned long long* out16) { for (int i = 0; i < 16; ++i) out16[i] = 0; return hipSuccess; }
#endif

namespace {

typedef const __attribute__((address_space(4))) BwdArgs* cold4_t;   // kernarg segment: s_load (see scan_bwd2.hip)
__device__ __forceinline__ cold4_t cold_args4() {
    cold4_t kp = (cold4_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return kp;
}

// sum of one column over the W wave slabs, fixed order, reads batched ahead of the adds
template <int RR>
__device__ __forceinline__ float colsum1_fixed(const float* __restrict__ colp, int stride) {
    float v[RR];
#pragma unroll
    for (int w = 0; w < RR; ++w) v[w] = colp[w * stride];
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < RR; ++w) s += v[w];
    return s;
}
__device__ __forceinline__ float colsum1(const float* __restrict__ colp, int stride, int W) {
    float s = 0.0f;
    int w = 0;
    for (; w + 12 <= W; w += 12) s += colsum1_fixed<12>(colp + w * stride, stride);
    if (w + 8 <= W) { s += colsum1_fixed<8>(colp + w * stride, stride); w += 8; }
    if (w + 4 <= W) { s += colsum1_fixed<4>(colp + w * stride, stride); w += 4; }
    for (; w < W; ++w) s += colp[w * stride];
    return s;
}

}  // namespace

template <bool REV>
__device__ __forceinline__ void scan_bwd4_body(const BwdArgs& q, float* smem, int b, int g, int chunk, int seg) {
    constexpr int T = kT4;
    const FwdArgs& p = q.f;
    const int W = blockDim.x >> 6;                    // waves; 4 rows each
    const int N = p.N, L = p.L, RB = q.RB;
    const int SB = q.slab2;                           // states per barrier; 2*SB slab sets
    const int bufsz = 2 * N * kTile4;
    float* sBC = smem;                                // [2][2][N][160]
    const int nbuf = p.NB;                            // B/C images: 2 = the next tile streams in during this one,
                                                      // 1 = it streams in during the last column sums + row epilogue
    float* sRed = sBC + nbuf * bufsz;                 // [2*SB][W][320] four-row sums of the dB/dC terms
    float* sRv = sRed + 2 * SB * W * kCols4;          // [RB*4*W][N] reverse carry a*dx of the tile to the right
    float* sSink = sRv + RB * 4 * W * N;              // 64 floats: target of the L2 warm-up touches
    float* sAcc = sSink + 64;                         // [N][320] when RB > 1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = lane & 15;
    const int qr = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool vec = p.vec_ok != 0;
    const bool li0 = li == 0, li15 = li == 15;
    const int row_c0 = g * p.rows_per_group + chunk * RB * 4 * W;   // first row of this workgroup's chunk
    const int rowbase4 = (lane & 48) << 2;            // ds_bpermute byte address of lane 0 of this DPP row
    const int vshift = 16 - N;                        // collected lane vectors: state n sits in lane vshift + n

    const float* __restrict__ Bg = reinterpret_cast<const float*>(p.B) + (long)b * p.B_bs + (long)g * p.B_gs;
    const float* __restrict__ Cg = reinterpret_cast<const float*>(p.C) + (long)b * p.C_bs + (long)g * p.C_gs;
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

    // Sequence segments (few rows, long sequences): this workgroup owns tiles [t_lo, t_hi).  The reverse carry entering
    // its right end is the composition of the summaries (decay product P, value E with zero carry) of the segments to
    // its right, written by rev_summary4_kernel: carry <- E_t + P_t * carry for t = S-1 .. seg+1.
    const int ntiles_all = (L + kTile4 - 1) / kTile4;
    const int t_lo = q.S > 1 ? seg * q.seg_tiles : 0;
    const int t_hi = q.S > 1 ? (t_lo + q.seg_tiles < ntiles_all ? t_lo + q.seg_tiles : ntiles_all) : ntiles_all;
    for (int i = tid; i < RB * 4 * W * N; i += blockDim.x) {
        float carry = 0.0f;
        if (q.S > 1) {
            const int rl = i / N, n = i - rl * N;
            const long rn = ((long)b * p.dim + (row_c0 + rl)) * N + n;
            for (int t = q.S - 1; t > seg; --t) {
                const float2 pe = reinterpret_cast<const float2*>(q.summ)[(long)(t - 1) * p.batch * p.dim * N + rn];
                carry = fmaf(pe.x, carry, pe.y);
            }
        }
        sRv[i] = carry;
    }
    // never multiply uninitialised LDS bits (stale/NaN) into the padding of the last tile
    for (int i = tid; i < nbuf * bufsz / 4; i += blockDim.x) reinterpret_cast<float4*>(sBC)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    auto stage = [&](int buf, int tile) {
        stage_tile4<REV>(sBC + buf * bufsz, Bg, Cg, (int)p.B_ns, (int)p.C_ns, N, tile, L);
    };

    // Slab layout: column c of a state's 320 = array (c / 160: dB, dC) x tile position (c % 160).  After fold16
    // register m of a lane holds {dB pos 2m, dB pos 2m+1, dC pos 2m, dC pos 2m+1}[DPP row] of the 10 positions of
    // lane li, so the lane writes it to column slab_c0 + 2m (bank-conflict free: lane stride 10 dwords, the four
    // rows land 0 / 1 / 160 / 161 apart).  Column-sum threads: with S = threads / 320 >= 1, thread t < 320*S owns
    // column t % 320 of the states t / 320, t / 320 + S, ... of a state group -- consecutive threads own
    // consecutive memory positions, so the dB/dC stores of a wave are one contiguous 256-byte piece; smaller
    // workgroups loop over the columns as well.
    const int slab_c0 = (qr >> 1) * kTile4 + li * T + (qr & 1);
    const int colS = (int)blockDim.x / kCols4;
    const int col_c = tid % kCols4;
    const int col_s0 = tid / kCols4;
    auto col_pos = [&](int c, bool& is_c) {
        is_c = c >= kTile4;
        return is_c ? c - kTile4 : c;
    };
    bool my_is_c;
    const int my_pos = col_pos(col_c, my_is_c);
    auto put_at = [&](int n, int l0, int pos, bool is_c, float v) {
        const int sp = l0 + pos;                                      // scan position
        if (sp < L) {
            const int mi = REV ? (L - 1 - sp) : sp;
            float* __restrict__ dst = is_c ? oC + (long)n * o_nsC : oB + (long)n * o_nsB;
            dst[mi] = v;
        }
    };

    int buf = 0;
    int grp = 0;                                       // state groups processed (slab set parity)
    PROF_DECL
    stage(0, t_hi - 1);
    lds_dma_wait();
    __syncthreads();
    PROF(0)

    for (int j = t_hi - 1; j >= t_lo; --j) {
        const int l0 = j * kTile4;
        const int lbase_t = l0 + li * T;
        const float* cur = sBC + buf * bufsz;
        if (nbuf == 2 && j > t_lo) stage(buf ^ 1, j - 1); // lands while this tile is processed
        for (int rb = 0; rb < RB; ++rb) {
            // laundered per row step: the ten load offsets derived from lbase are row-step invariant, and hoisted out
            // of this loop they lived in scratch in the 128-VGPR build (30 reloads of 512 B per wave and row step)
            int lbase = lbase_t;
            asm volatile("" : "+v"(lbase));
            const int rl = (rb * W + wave) * 4 + qr;   // row inside the chunk (per DPP row)
            const int r = row_c0 + rl;
            cold4_t kq = cold_args4();
            const int rpg = kq->f.rows_per_group;
            const int ur = r - ((g - (g >> kq->f.u_gshift)) * rpg);   // same row of group g >> u_gshift
            const int gr = r - ((g - (g >> kq->g_gshift)) * rpg);
            const float* __restrict__ u_row = reinterpret_cast<const float*>(kq->f.u) + (long)b * kq->f.u_bs + (long)ur * kq->f.u_ds;
            const float* __restrict__ d_row = reinterpret_cast<const float*>(kq->f.delta) + (long)b * kq->f.dt_bs + (long)r * kq->f.dt_ds;
            const float* __restrict__ g_row = reinterpret_cast<const float*>(kq->dout) + (long)b * kq->g_bs + (long)gr * kq->g_ds;
            const int pr = param_row(r, g, rpg, kq->f.pswap);
            const float bias = kq->f.bias ? kq->f.bias[pr] : 0.0f;

            // lane vectors (lane 16*row + n = state n of that row)
            float Av = 0.0f, X0v = 0.0f, Rvv = 0.0f, rvout_v = 0.0f, dA_v = 0.0f;
            if (li < N) {
                Av = kq->f.A[(long)pr * kq->f.A_ds + (long)li * kq->f.A_ns];
                if (j > 0) X0v = kq->f.x[((long)b * kq->f.dim + r) * kq->f.x_rs + (long)(j - 1) * N + li];
                Rvv = sRv[rl * N + li];
            }
            float dl[T], dlu[T], gg[T], sdxB[T], sAx[T];
            float dD_acc = 0.0f;                       // sum of dout * u over the lane's positions (zero past the end)
            {
                float dv[T], uu[T];
#if SIGMA_BWD4_ABL & 32
#pragma unroll
                for (int k = 0; k < T; ++k) { uu[k] = 0.5f + 0.01f * (lane + k); dv[k] = 0.1f * k; gg[k] = 1.0f - 0.02f * k; }
                asm volatile("" : "+v"(uu[0]), "+v"(dv[0]), "+v"(gg[0]));
#else
                load_items<float, T, REV>(u_row, lbase, L, vec, uu);
                load_items<float, T, REV>(d_row, lbase, L, vec, dv);
                load_items<float, T, REV>(g_row, lbase, L, vec, gg);
#endif
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    float d = dv[k] + bias;
                    if (p.softplus) { float sg; d = softplus_ref(d, sg); }
                    d = (lbase + k < L) ? d : 0.0f;    // identity element past the end (a = 1, b = 0)
                    dl[k] = d;
                    dlu[k] = d * uu[k];
                    dD_acc = fmaf(gg[k], uu[k], dD_acc);
                    sdxB[k] = 0.0f;
                    sAx[k] = 0.0f;
                }
            }
            float dsum = 0.0f;
#pragma unroll
            for (int k = 0; k < T; ++k) dsum += dl[k];
            // dA / dD / ddelta_bias leave through one atomicAdd per (row, tile) and state.  Summing them over the
            // tiles in LDS first (measured: WRITE_SIZE -6 %) costs the second B/C image its LDS and 3 % run time.
#if !(SIGMA_BWD4_ABL & 2)
            if (kq->dD) { dD_acc = row_sum_to_lane0(dD_acc); if (li0) atomicAdd(kq->dD + pr, dD_acc); }
#endif

            PROF(1)                                            // row prologue: loads, softplus
            // row scalars of state 0; those of state n + 1 are fetched while state n is computed
            float An_nx = row_pick(Av, rowbase4), x0_nx = row_pick(X0v, rowbase4), cy_nx = row_pick(Rvv, rowbase4);
#pragma unroll 1
            for (int n = 0; n < N; ++n) {
                const float An = An_nx, x0 = x0_nx, carry = cy_nx;
                {
                    const int nn = (n + 1 < N) ? n + 1 : n;
                    const int ad = rowbase4 + 4 * nn;
                    An_nx = row_pick(Av, ad); x0_nx = row_pick(X0v, ad); cy_nx = row_pick(Rvv, ad);
                }
                if (n == N - 3 && !(q.flags & 1)) {
                    // next step of THIS wave: same tile, next row block -- or the tile to the left of its first rows
                    const int jn = (rb + 1 < RB) ? j : j - 1;
                    if (jn >= t_lo) {
                        cold4_t kt = cold_args4();
                        const int trow = lane / 6, tline = lane - trow * 6;          // 24 lanes: 4 rows x 6 lines
                        const int rn = row_c0 + (((rb + 1 < RB) ? rb + 1 : 0) * W + wave) * 4 + trow;
                        const int rpgt = kt->f.rows_per_group;
                        const int urn = rn - ((g - (g >> kt->f.u_gshift)) * rpgt);
                        const int grn = rn - ((g - (g >> kt->g_gshift)) * rpgt);
                        const int l0n = jn * kTile4;
                        const int m0 = REV ? (L - l0n - kTile4 < 0 ? 0 : L - l0n - kTile4) : l0n;      // first memory element
                        const int m1 = REV ? L - l0n : (l0n + kTile4 < L ? l0n + kTile4 : L);
                        const bool on = trow < 4 && tline * 32 < (m1 - m0) + 31;
                        const int me = m0 + tline * 32 < m1 ? m0 + tline * 32 : m1 - 1;                // element inside the line
                        const unsigned sink = (unsigned)(uintptr_t)(lptr_t)sSink;
                        touch_line4(reinterpret_cast<const char*>(reinterpret_cast<const float*>(kt->f.u) + (long)b * kt->f.u_bs + (long)urn * kt->f.u_ds + me), on, sink);
                        touch_line4(reinterpret_cast<const char*>(reinterpret_cast<const float*>(kt->f.delta) + (long)b * kt->f.dt_bs + (long)rn * kt->f.dt_ds + me), on, sink);
                        touch_line4(reinterpret_cast<const char*>(reinterpret_cast<const float*>(kt->dout) + (long)b * kt->g_bs + (long)grn * kt->g_ds + me), on, sink);
                    }
                }
                const float A2 = An * kLog2e;
                const float* tB = cur + n * kTile4;
                const float* tC = tB + N * kTile4;
                float a[T], xs[T], gc[T];
                // ---- forward: in-lane fold (lane 0 of the row starts from the checkpoint), row scan, replay
                float xa = li0 ? x0 : 0.0f;
#pragma unroll
                for (int qq = 0; qq < T / 2; ++qq) {
                    float bq[2], cq[2];
                    lds_read_pair<REV>(tB, li, qq, bq);
                    lds_read_pair<REV>(tC, li, qq, cq);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int k = 2 * qq + jj;
                        a[k] = fast_exp2(dl[k] * A2);
                        xs[k] = dlu[k] * bq[jj];
                        gc[k] = gg[k] * cq[jj];
                        xa = fmaf(a[k], xa, xs[k]);
                    }
                }
                const float plane = fast_exp2(A2 * dsum);       // this lane's decay product
                float pf = plane;
                PROF(2)                                        // B/C reads, exp, forward fold
                row_mscan_inclusive(pf, xa);
                const float xstart = dpp_take<DPP_ROW_SHR1, 0xF>(x0, xa);    // state entering the lane
                {
                    float x = xstart;
#pragma unroll
                    for (int k = 0; k < T; ++k) { x = fmaf(a[k], x, xs[k]); xs[k] = x; }
                }
                PROF(3)                                        // forward scan + replay
                // ---- reverse: e_k = a_k * dx_k, dx_k = g_k C_k + e_{k+1}; lane 15 of the row starts from the carry
                float e = li15 ? carry : 0.0f;
#pragma unroll
                for (int k = T - 1; k >= 0; --k) e = a[k] * (gc[k] + e);
                float prv = plane;
                row_mscan_inclusive_rev(prv, e);
                e = dpp_take<DPP_ROW_SHL1, 0xF>(carry, e);       // e entering the lane from the right
                float dAp = 0.0f;
                PROF(4)                                        // reverse fold + scan
                const int sidx = ((grp & 1) * SB + (n % SB)) * W + wave;
                float* __restrict__ slab = sRed + sidx * kCols4 + slab_c0;
#pragma unroll
                for (int qq = T / 2 - 1; qq >= 0; --qq) {
                    float bq[2], vb[2], vc[2];
                    lds_read_pair<REV>(tB, li, qq, bq);          // B again: cheaper than T live registers
#pragma unroll
                    for (int jj = 1; jj >= 0; --jj) {
                        const int k = 2 * qq + jj;
                        const float dx = gc[k] + e;
                        e = a[k] * dx;
                        sdxB[k] = fmaf(dx, bq[jj], sdxB[k]);
                        const float t = e * (k > 0 ? xs[k > 0 ? k - 1 : 0] : xstart);   // dx * a_k * x_{k-1}
                        sAx[k] = fmaf(An, t, sAx[k]);
                        dAp = fmaf(dl[k], t, dAp);
                        vb[jj] = dx * dlu[k];                    // this row's term of dB[n, l]
                        vc[jj] = gg[k] * xs[k];                  // this row's term of dC[n, l]
                    }
                    // four-row sums: rows of the result = {dB pos 2qq, dB pos 2qq+1, dC pos 2qq, dC pos 2qq+1}
#if SIGMA_BWD4_ABL & 16
                    asm volatile("" :: "v"(vb[0]), "v"(vc[0]), "v"(vb[1]), "v"(vc[1]));
#else
                    slab[2 * qq] = fold16(fold32(vb[0], vc[0]), fold32(vb[1], vc[1]));
#endif
                }
                // collect: lane 0 of each row holds the result of this state; select + rotate, so that after
                // N states the value of state n sits in lane 16 - N + n
                rvout_v = row_rotate_left(li0 ? e : rvout_v);
                const float dA_row = row_sum_to_lane0(dAp);      // all lanes take part: outside the select
                dA_v = row_rotate_left(li0 ? dA_row : dA_v);
                PROF(5)                                        // reverse replay, four-row sums, slab writes, collect
#if SIGMA_BWD4_ABL & 8
                if (n == N - 1 && rb == RB - 1) {
                    if (nbuf == 2) { lds_dma_wait(); __syncthreads(); }
                    else { lds_barrier(); if (j > t_lo) stage(0, j - 1); }
                }
                if (false) {
#else
                if ((n % SB) == SB - 1) {
#endif
                    // slabs of this state group complete; at the end of the tile also "next B/C image landed"
                    if (n == N - 1 && rb == RB - 1) {
                        if (nbuf == 2) { lds_dma_wait(); __syncthreads(); }
                        else { lds_barrier(); if (j > t_lo) stage(0, j - 1); }   // every wave is done with this tile's image
                    } else {
                        lds_barrier();
                    }
#if SIGMA_BWD4_ABL & 4
                    ++grp;
                    continue;
#endif
                    PROF(6)                                    // barrier wait
                    const float* sset = sRed + ((grp & 1) * SB) * W * kCols4;
                    auto column = [&](int s, int c, int pos, bool is_c) {
                        const int ns = n - (SB - 1) + s;
                        const float sum = colsum1(sset + s * W * kCols4 + c, kCols4, W);
                        if (RB > 1) {
                            float* ap = sAcc + ns * kCols4 + c;
                            if (rb == 0) {
                                *ap = sum;
                            } else {
                                const float t = *ap + sum;
                                if (rb == RB - 1) put_at(ns, l0, pos, is_c, t); else *ap = t;
                            }
                        } else {
                            put_at(ns, l0, pos, is_c, sum);
                        }
                    };
                    if (colS >= 1) {
                        if (col_s0 < colS)
                            for (int s = col_s0; s < SB; s += colS) column(s, col_c, my_pos, my_is_c);
                    } else {
                        for (int c2 = tid; c2 < SB * kCols4; c2 += blockDim.x) {
                            const int s = c2 / kCols4;
                            const int c = c2 - s * kCols4;
                            bool is_c;
                            const int pos = col_pos(c, is_c);
                            column(s, c, pos, is_c);
                        }
                    }
                    ++grp;
                    PROF(7)                                    // column sums
                }
            }

            // ---- per-row results of this tile (cold parameters re-read here)
            // Row indices are rebuilt from the (laundered) lane id: otherwise r, pr, lbase and the address parts the
            // compiler pre-computes from them stay live across the state loop -- in the 128-VGPR build as scratch spills.
            cold4_t ke = cold_args4();
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int li_e = lane_e & 15;
            const int rl_e = (rb * W + wave) * 4 + (lane_e >> 4);
            const int r_e = row_c0 + rl_e;
            const int pr_e = param_row(r_e, g, ke->f.rows_per_group, ke->f.pswap);
            const int lbase_e = l0 + li_e * T;
            if (li_e >= vshift) {
                const int st = li_e - vshift;
                sRv[rl_e * N + st] = rvout_v;
#if !(SIGMA_BWD4_ABL & 2)
                atomicAdd(ke->dA + (long)pr_e * ke->dA_ds + (long)st * ke->dA_ns, dA_v);
#endif
            }
            float dbias_acc = 0.0f;
            float* __restrict__ du_row = reinterpret_cast<float*>(ke->du) + (long)b * ke->du_bs + (long)r_e * ke->du_ds;
            float* __restrict__ dd_row = reinterpret_cast<float*>(ke->ddelta) + (long)b * ke->dd_bs + (long)r_e * ke->dd_ds;
            {
                const float Dd = ke->f.D ? ke->f.D[pr_e] : 0.0f;
                if (p.softplus) {
                    // Nothing is re-read here (the row epilogue was 19 % of the kernel waiting for u and delta again,
                    // profiles/r02_bwd4_phases.txt): with dl = softplus(raw) > 0,
                    //   u * sdxB            = (dl*u) * sdxB / dl
                    //   softplus'(raw)      = sigmoid(raw) = 1 - exp(-dl) = -expm1(-dl)
                    // -expm1(-dl): 1 - exp2 above 0.25 (relative error < 3e-7), a degree-7 series below (< 1e-7).
                    // dl == 0 only past the end of the row or where exp(raw) underflows, i.e. sigmoid(raw) = 0 too.
                    auto dd_of = [&](int k) {
                        const float d = dl[k];
                        const bool live = d >= 1.17549435e-38f;
                        const float usx = dlu[k] * (sdxB[k] * fast_rcp(live ? d : 1.0f));
                        const float big = 1.0f - fast_exp2(-d * kLog2e);
                        float ser = fmaf(d, -1.0f / 7.0f, 1.0f);
                        ser = fmaf(ser * d, -1.0f / 6.0f, 1.0f);
                        ser = fmaf(ser * d, -1.0f / 5.0f, 1.0f);
                        ser = fmaf(ser * d, -1.0f / 4.0f, 1.0f);
                        ser = fmaf(ser * d, -1.0f / 3.0f, 1.0f);
                        ser = fmaf(ser * d, -1.0f / 2.0f, 1.0f);
                        const float sig = d > 0.25f ? big : ser * d;
                        return live ? (usx + sAx[k]) * sig : 0.0f;
                    };
                    float duv[T], ddv[T];
#pragma unroll
                    for (int k = 0; k < T; ++k) {
                        duv[k] = fmaf(Dd, gg[k], dl[k] * sdxB[k]);
                        ddv[k] = dd_of(k);
                        dbias_acc += ddv[k];           // zero past the end (dl = 0 there)
                    }
#if SIGMA_BWD4_ABL & 1
#pragma unroll
                    for (int k = 0; k < T; ++k) asm volatile("" :: "v"(duv[k]), "v"(ddv[k]));
#else
                    store_items<float, T, REV>(du_row, lbase_e, L, vec, duv);
                    store_items<float, T, REV>(dd_row, lbase_e, L, vec, ddv);
#endif
                } else {
                    const int rpg2 = ke->f.rows_per_group;
                    const int ur2 = r_e - ((g - (g >> ke->f.u_gshift)) * rpg2);
                    const float* __restrict__ u_row2 = reinterpret_cast<const float*>(ke->f.u) + (long)b * ke->f.u_bs + (long)ur2 * ke->f.u_ds;
                    float uu[T];
                    load_items<float, T, REV>(u_row2, lbase_e, L, vec, uu);
                    float duv[T], ddv[T];
#pragma unroll
                    for (int k = 0; k < T; ++k) {
                        duv[k] = fmaf(Dd, gg[k], dl[k] * sdxB[k]);
                        ddv[k] = fmaf(uu[k], sdxB[k], sAx[k]);
                        dbias_acc += (lbase_e + k < L) ? ddv[k] : 0.0f;
                    }
                    store_items<float, T, REV>(du_row, lbase_e, L, vec, duv);
                    store_items<float, T, REV>(dd_row, lbase_e, L, vec, ddv);
                }
            }
#if !(SIGMA_BWD4_ABL & 2)
            if (ke->dbias) { dbias_acc = row_sum_to_lane0(dbias_acc); if (li_e == 0) atomicAdd(ke->dbias + pr_e, dbias_acc); }
#else
            asm volatile("" :: "v"(dbias_acc), "v"(dA_v));
#endif
            PROF(8)                                            // row epilogue
        }
        if (nbuf == 2) buf ^= 1;
        else if (j > t_lo) { lds_dma_wait(); __syncthreads(); }
    }
    PROF_FLUSH
}

template <int MAXW>
__global__ void __launch_bounds__(64 * MAXW)
scan_bwd4_kernel(const BwdArgs q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int PS = q.P * q.S;                         // workgroups per (batch, group): row chunks x sequence segments
    const int per_b = q.f.G * PS;
    const int b = lb / per_b;
    const int rem = lb - b * per_b;
    const int g = rem / PS;
    const int rem2 = rem - g * PS;
    const int chunk = rem2 / q.S;
    const int seg = rem2 - chunk * q.S;
    if ((q.f.rev_mask >> g) & 1u) scan_bwd4_body<true>(q, smem, b, g, chunk, seg);
    else scan_bwd4_body<false>(q, smem, b, g, chunk, seg);
}

// ---- reverse summaries of the sequence segments (S > 1) ----------------------------------------------------------
// For segment s = 1 .. S-1, row r and state n: (P, E) with  e(left end of s) = E + P * e(right end of s), i.e. the
// decay product of the segment and the reverse recurrence e_k = a_k (dout_k C_k + e_{k+1}) run over it with zero
// carry.  Same lane mapping as the main kernel; per tile the in-lane fold + 4-step row scan give the tile's E at lane
// 0 of the row, the tile's P is exp2(A2 * sum of delta over the tile); tiles compose right to left in two lane vectors.
template <bool REV>
__device__ __forceinline__ void rev_summary4_body(const BwdArgs& q, float* smem, int b, int g, int chunk, int seg) {
    constexpr int T = kT4;
    const FwdArgs& p = q.f;
    const int W = blockDim.x >> 6;
    const int N = p.N, L = p.L;
    const int bufsz = 2 * N * kTile4;
    float* sBC = smem;                                // [2][2][N][160] (only the C half is read)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int li = lane & 15;
    const int qr = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool vec = p.vec_ok != 0;
    const bool li0 = li == 0;
    const int rowbase4 = (lane & 48) << 2;
    const int vshift = 16 - N;

    const int r = g * p.rows_per_group + (chunk * W + wave) * 4 + qr;
    const int gr = r - ((g - (g >> q.g_gshift)) * p.rows_per_group);
    const float* __restrict__ d_row = reinterpret_cast<const float*>(p.delta) + (long)b * p.dt_bs + (long)r * p.dt_ds;
    const float* __restrict__ g_row = reinterpret_cast<const float*>(q.dout) + (long)b * q.g_bs + (long)gr * q.g_ds;
    const float* __restrict__ Bg = reinterpret_cast<const float*>(p.B) + (long)b * p.B_bs + (long)g * p.B_gs;
    const float* __restrict__ Cg = reinterpret_cast<const float*>(p.C) + (long)b * p.C_bs + (long)g * p.C_gs;
    const int pr = param_row(r, g, p.rows_per_group, p.pswap);
    const float bias = p.bias ? p.bias[pr] : 0.0f;
    float A2v = 0.0f;
    if (li < N) A2v = p.A[(long)pr * p.A_ds + (long)li * p.A_ns] * kLog2e;
    float PV = 1.0f, EV = 0.0f;                       // state n in lane vshift + n

    const int ntiles_all = (L + kTile4 - 1) / kTile4;
    const int t_lo = seg * q.seg_tiles;
    const int t_hi = t_lo + q.seg_tiles < ntiles_all ? t_lo + q.seg_tiles : ntiles_all;
    for (int i = tid; i < 2 * bufsz / 4; i += blockDim.x) reinterpret_cast<float4*>(sBC)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    auto stage = [&](int buf, int tile) {
        stage_tile4<REV>(sBC + buf * bufsz, Bg, Cg, (int)p.B_ns, (int)p.C_ns, N, tile, L);
    };
    stage(0, t_hi - 1);
    lds_dma_wait();
    __syncthreads();
    int buf = 0;
    for (int j = t_hi - 1; j >= t_lo; --j) {
        const float* cur = sBC + buf * bufsz;
        if (j > t_lo) stage(buf ^ 1, j - 1);
        const int lbase = j * kTile4 + li * T;
        float dl[T], gg[T];
        float dsum = 0.0f;
        {
            float dv[T];
            load_items<float, T, REV>(d_row, lbase, L, vec, dv);
            load_items<float, T, REV>(g_row, lbase, L, vec, gg);
#pragma unroll
            for (int k = 0; k < T; ++k) {
                float d = dv[k] + bias;
                if (p.softplus) { float sg; d = softplus_ref(d, sg); }
                d = (lbase + k < L) ? d : 0.0f;
                dl[k] = d;
                dsum += d;
            }
        }
        const float dsum_row = row_pick(row_sum_to_lane0(dsum), rowbase4);      // sum of delta over the row's tile
        float PVn = 0.0f, EVn = 0.0f;
#pragma unroll 1
        for (int n = 0; n < N; ++n) {
            const float A2 = row_pick(A2v, rowbase4 + 4 * n);
            const float Pold = row_pick(PV, rowbase4 + 4 * (vshift + n));
            const float Eold = row_pick(EV, rowbase4 + 4 * (vshift + n));
            const float* tC = cur + (N + n) * kTile4;
            float e = 0.0f;
#pragma unroll
            for (int qq = T / 2 - 1; qq >= 0; --qq) {
                float cq[2];
                lds_read_pair<REV>(tC, li, qq, cq);
#pragma unroll
                for (int jj = 1; jj >= 0; --jj) {
                    const int k = 2 * qq + jj;
                    e = fast_exp2(dl[k] * A2) * fmaf(gg[k], cq[jj], e);
                }
            }
            float prv = fast_exp2(A2 * dsum);
            row_mscan_inclusive_rev(prv, e);                                     // lane 0: the tile's E
            const float Ptile = fast_exp2(A2 * dsum_row);
            const float En = fmaf(Ptile, Eold, e);
            const float Pn = Ptile * Pold;
            EVn = row_rotate_left(li0 ? En : EVn);
            PVn = row_rotate_left(li0 ? Pn : PVn);
        }
        PV = PVn; EV = EVn;
        lds_dma_wait();
        __syncthreads();
        buf ^= 1;
    }
    if (li >= vshift)
        reinterpret_cast<float2*>(q.summ)[((long)(seg - 1) * p.batch * p.dim + (long)b * p.dim + r) * N + (li - vshift)] = make_float2(PV, EV);
}

__global__ void __launch_bounds__(512)
rev_summary4_kernel(const BwdArgs q, int Pq) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lb = blockIdx.x;
    const int PS = Pq * (q.S - 1);                    // row chunks x segments 1..S-1
    const int per_b = q.f.G * PS;
    const int b = lb / per_b;
    const int rem = lb - b * per_b;
    const int g = rem / PS;
    const int rem2 = rem - g * PS;
    const int chunk = rem2 / (q.S - 1);
    const int seg = 1 + rem2 - chunk * (q.S - 1);
    if ((q.f.rev_mask >> g) & 1u) rev_summary4_body<true>(q, smem, b, g, chunk, seg);
    else rev_summary4_body<false>(q, smem, b, g, chunk, seg);
}

// a.f.R = waves per workgroup (4 rows each), a.slab2 = states per barrier, a.RB = row blocks per workgroup
template <int MAXW>
static hipError_t launch_bwd4_t(const BwdArgs& a, hipStream_t stream) {
    const size_t lds = bwd4_lds_bytes(a.f.R, a.f.N, a.slab2, a.RB, a.f.NB);
    const int grid = a.f.batch * a.f.G * a.P * a.S;
    auto kern = scan_bwd4_kernel<MAXW>;
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

// the pre-pass of the sequence split: W4 = waves per workgroup (<= 8) over the same rows, one workgroup per segment 1..S-1
static hipError_t launch_rev_summary4(const BwdArgs& a, hipStream_t stream) {
    const int quads = a.f.rows_per_group / 4;
    int W4 = 8;
    while (W4 > 1 && quads % W4 != 0) --W4;
    const int Pq = quads / W4;
    const size_t lds = fwd4_lds_bytes(a.f.N);
    const int grid = a.f.batch * a.f.G * Pq * (a.S - 1);
    auto kern = rev_summary4_kernel;
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
    hipLaunchKernelGGL(kern, dim3(grid), dim3(W4 * 64), lds, stream, a, Pq);
    return hipGetLastError();
}

hipError_t launch_scan_bwd4(const BwdArgs& a, hipStream_t stream) {
    if (a.S > 1) {
        hipError_t e = launch_rev_summary4(a, stream);
        if (e != hipSuccess) return e;
    }
    // up to 12 waves: ~150 VGPRs, 3 waves per SIMD; 13..16 waves -- or two workgroups per CU (flags bit 1): the
    // 128-VGPR build
    return (a.f.R > 12 || (a.flags & 2)) ? launch_bwd4_t<16>(a, stream) : launch_bwd4_t<12>(a, stream);
}

}  // namespace sigma
