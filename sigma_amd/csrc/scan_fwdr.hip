// scan_fwdr.hip -- selective-scan forward, "row-lane" mapping (gfx950 / MI355X, wave64, f32 IO, ckpt_pitch 16).
//
// Same operator as scan_fwd.hip / scan_fwd4.hip (reference: models/encoders/selective_scan/csrc/selective_scan/
// selective_scan_fwd_kernel.cuh:62-238; mathematics SURVEY.md App. E.1); the mapping is described in scan_rowlane.h:
// a lane is a channel row, a wave is 64 rows x NS states, B / C are SGPR operands.  Per element-state the kernel issues
//     v_mul (delta * A2), v_exp, v_mul (delta*u * B), v_fma (x), v_fma (y)
// and nothing else: no fold / replay double pass, no DPP scan, no LDS read of B / C (the quad-row kernel: 7.8 VALU per
// element-state + one barrier per tile with the B/C image behind it).
//
// Few rows (one image per GPU): the sequence is cut into S segments owned by different workgroups.  A pre-pass
// (scan_fwdr_summary_kernel) writes per segment, row and state the pair (decay product P, end state X from zero); a
// segment's workgroup composes the summaries to its left into its incoming state (x <- X_t + P_t * x).  This replaces
// the reference's sequential chunk loop (selective_scan_fwd_kernel.cuh:109-110) for launches that cannot fill the chip.
#include "scan_device.h"
#include "scan_launch.h"
#include "scan_rowlane.h"

#include <atomic>
#include <type_traits>

#if SIGMA_RL_PROF
__device__ unsigned long long g_fwdr_prof[16];
#endif

namespace sigma {

#if SIGMA_RL_PROF
hipError_t fwdr_prof_read(unsigned long long* out16) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fwdr_prof), 16 * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fwdr_prof), z, sizeof(z));
}
#else
hipError_t fwdr_prof_read(unsigned long long* out16) { for (int i = 0; i < 16; ++i) out16[i] = 0; return hipSuccess; }
#endif

// MODE 0: out only (inference, x == NULL); 1: out + checkpoints; 2: the pre-pass of the sequence split -- recurrence only
// (no C, no out, no checkpoints), the segment's (P, X) pairs to p.fsumm.  A template parameter rather than a run-time
// test: the state loop of a tile must stay ONE basic block, or the compiler sinks the 64 C * x products of a tile behind
// the last state (and keeps every x alive until then).
// NW = state waves per row block (4, 8 or 16): a workgroup is 64 rows x NW waves x NS = dstate / NW states; the (row,
// chunk) duties (loads, softplus, the final sum and the stores) belong to waves 0..3.
template <int NS, int NW, bool REV, int MODE>
__device__ __forceinline__ void scan_fwdr_body(const FwdArgs& p, float* smem, int b, int g, int rbg, int seg) {
    constexpr int T = kRT;
    constexpr bool SUMMARY = MODE == 2;
    constexpr bool CK = MODE == 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int sw = __builtin_amdgcn_readfirstlane(tid >> 6);      // state wave 0..NW-1
    const bool rcw = sw < 4;                                      // this wave also works as (row, chunk) threads
    const int N = p.N, L = p.L;
    const int ntiles = (L + T - 1) / T;
    v4f* sProc = reinterpret_cast<v4f*>(smem);                    // [2][4][64] float4: delta, delta * u
    v4f* sEx = sProc + 2 * 256;                                   // [NW waves][4][64] float4: partial sums over the states

    // thread as (row, chunk): loads / pre-processes / stores 4 positions of one row
    const int rr = ((sw & 3) << 4) | (lane >> 2), cc = lane & 3;
    const int rpg = p.rows_per_group;
    const int row0 = g * rpg + rbg * kRRows;
    const int r_rc = row0 + rr;
    const int ur_rc = r_rc - ((g - (g >> p.u_gshift)) * rpg);     // same row of group g >> u_gshift
    const float* __restrict__ u_row = reinterpret_cast<const float*>(p.u) + (long)b * p.u_bs + (long)ur_rc * p.u_ds + 4 * cc;
    const float* __restrict__ d_row = reinterpret_cast<const float*>(p.delta) + (long)b * p.dt_bs + (long)r_rc * p.dt_ds + 4 * cc;
    float* __restrict__ o_row = reinterpret_cast<float*>(p.out) + (long)b * p.o_bs + (long)r_rc * p.o_ds + 4 * cc;
    const int pr_rc = param_row(r_rc, g, rpg, p.pswap);
    const float bias = p.bias ? p.bias[pr_rc] : 0.0f;
    const float Dd = p.D ? p.D[pr_rc] : 0.0f;

    // thread as row lane: states [sw * NS, sw * NS + NS) of row row0 + lane
    const int pr_ln = param_row(row0 + lane, g, rpg, p.pswap);
    const int n0 = sw * NS;
    float A2[NS], xst[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        A2[s] = p.A[(long)pr_ln * p.A_ds + (long)(n0 + s) * p.A_ns] * kLog2e;
        xst[s] = 0.0f;
    }
    const float* Bg = reinterpret_cast<const float*>(p.B) + (long)b * p.B_bs + (long)g * p.B_gs;
    const float* Cg = reinterpret_cast<const float*>(p.C) + (long)b * p.C_bs + (long)g * p.C_gs;
    const long rowblock = (long)b * (p.dim >> 6) + (row0 >> 6);
    // checkpoints (pitch 8): x[(((rowblock * ntiles + tile) * 2 + h) * N + n) * 64 + lane], h = 0: state after the first
    // scan half of the tile, h = 1: after the tile
#if SIGMA_BWDR_FULL
    float* __restrict__ ck = CK ? p.x + rowblock * ntiles * N * 64 + lane : nullptr;      // one checkpoint per tile (scan_rowlane.h)
#else
    float* __restrict__ ck = CK ? p.x + rowblock * ntiles * 2 * N * 64 + lane : nullptr;
#endif

    // tiles of this workgroup in scan order: steps it = 0 .. nst-1, memory tile m = REV ? hi - it : lo + it
    const int st_lo = p.segs > 1 ? seg * p.seg_tiles : 0;
    const int st_hi = p.segs > 1 ? (st_lo + p.seg_tiles < ntiles ? st_lo + p.seg_tiles : ntiles) : ntiles;   // scan steps [st_lo, st_hi)
    const int nst = st_hi - st_lo;
    float Pacc[NS];
    if (SUMMARY) {
#pragma unroll
        for (int s = 0; s < NS; ++s) Pacc[s] = 0.0f;              // log2 of the decay product
    } else if (p.segs > 1 && seg > 0) {
        // incoming state = composition of the summaries of the segments before this one (scan order)
        const float2* __restrict__ sm = reinterpret_cast<const float2*>(p.fsumm);
        // eight segments' summaries requested at once: one iteration per segment waited ~0.4 us for its own four loads, and the
        // LAST segment's workgroup entered its tile loop up to 47 of those late (round 6: (1,768,19200) runs 48 segments)
        const long sm_seg = (long)p.batch * (p.dim >> 6) * N * 64;
        const float2* __restrict__ sm0 = sm + ((long)rowblock * N + n0) * 64 + lane;
        int t = 0;
        for (; t + 8 <= seg; t += 8) {
            float2 pe[8][NS];
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int s = 0; s < NS; ++s) pe[k][s] = sm0[(long)(t + k) * sm_seg + s * 64];
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int s = 0; s < NS; ++s) xst[s] = fmaf(pe[k][s].x, xst[s], pe[k][s].y);
        }
        for (; t < seg; ++t) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float2 pe = sm0[(long)t * sm_seg + s * 64];
                xst[s] = fmaf(pe.x, xst[s], pe.y);
            }
        }
    }

    auto tile_of = [&](int it) { return REV ? (ntiles - 1 - (st_lo + it)) : (st_lo + it); };
    int m = tile_of(0);
    v4f u_nx = rl_load4(u_row + T * m, rcw && T * m + 4 * cc < L);
    v4f d_nx = rl_load4(d_row + T * m, rcw && T * m + 4 * cc < L);

    // wave-uniform bases of this wave's first state
    const float* Bw = Bg + (long)n0 * p.B_ns;
    const float* Cw = Cg + (long)n0 * p.C_ns;
    const int B_ns = (int)p.B_ns, C_ns = (int)p.C_ns;               // host: (N - 1) * stride + L fits 31 bits

    // one tile; TAIL = the partial last memory tile (chunk-wise conditional B / C loads), else B / C of state s + 1
    // are requested before state s is computed
    // B / C of the NEXT tile are pulled into L2 a tile ahead by one vector load per wave (lane 2s + arr: state s, array
    // arr): the scalar requests of the state loop run one state (~1000 clocks) ahead, which covers an L2 hit, not HBM --
    // and the row blocks of a (batch, group) walk the sequence together, so without this every tile would start with
    // HBM-latency scalar loads.
    const float* __restrict__ bc_touch = ((lane & 1) ? Cw : Bw) + (long)(lane >> 1) * ((lane & 1) ? C_ns : B_ns);
    const bool bc_touch_on = lane < 2 * NS;
    float touch_nx = 0.0f, touch_acc = 0.0f;
    RLPROF_DECL
    auto step = [&](int it, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        m = tile_of(it);
        const bool valid = rcw && T * m + 4 * cc < L;
        const v4f uu = u_nx, dd = d_nx;
        touch_acc += touch_nx;
        if (it + 1 < nst) {                                         // next tile's u / delta fly during this tile
            const int mn = tile_of(it + 1);
            touch_nx = bc_touch_on ? bc_touch[T * mn] : 0.0f;
            u_nx = rl_load4(u_row + T * mn, rcw && T * mn + 4 * cc < L);
            d_nx = rl_load4(d_row + T * mn, rcw && T * mn + 4 * cc < L);
        }
        const int nch = TAIL ? (L - T * m) >> 2 : 4;
        float Bn[T], Cn[T];
        rl_load_bc(Bw + T * m, nch, Bn);
        if (!SUMMARY) rl_load_bc(Cw + T * m, nch, Cn);
        v4f du4 = {0.0f, 0.0f, 0.0f, 0.0f};
        if (NW == 4 || rcw) {
            v4f dl4, dlu4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float raw = dd[j] + bias;
                float sig;
#if SIGMA_RL_ABL & 32
                const float sp = raw * raw; sig = 1.0f;
#else
                const float sp = softplus_ref(raw, sig);
#endif
                float d = p.softplus ? sp : raw;
                d = valid ? d : 0.0f;                               // identity element past the end (a = 1, b = 0)
                dl4[j] = d;
                dlu4[j] = d * uu[j];
                du4[j] = Dd * uu[j];
            }
            sProc[rl_unit(cc, rr)] = dl4;
            sProc[256 + rl_unit(cc, rr)] = dlu4;
        }
        RLPROF(0)                                                   // operand wait, softplus, LDS writes
        rl_barrier();
        RLPROF(1)                                                   // barrier 1

        float dl[T], dlu[T], y[T];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const v4f a = sProc[rl_unit(c, lane)], bq = sProc[256 + rl_unit(c, lane)];
#pragma unroll
            for (int j = 0; j < 4; ++j) { dl[4 * c + j] = a[j]; dlu[4 * c + j] = bq[j]; y[4 * c + j] = 0.0f; }
        }
        // LDS reads and scalar loads share one counter: retire the reads before the first scalar request of the state
        // loop, or its wait for them (lgkmcnt(0)) would also sit out that request
        asm volatile("" : "+v"(dl[0]), "+v"(dlu[0]), "+v"(dl[4]), "+v"(dlu[4]), "+v"(dl[8]), "+v"(dlu[8]), "+v"(dl[12]), "+v"(dlu[12]));
        __builtin_amdgcn_sched_barrier(0);
        RLPROF(2)                                                   // LDS reads
        float dsum = 0.0f;
        if (SUMMARY) {
#pragma unroll
            for (int k = 0; k < T; ++k) dsum += dl[k];
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float Bt[T], Ct[T];
#pragma unroll
            for (int k = 0; k < T; ++k) { Bt[k] = Bn[k]; Ct[k] = SUMMARY ? 0.0f : Cn[k]; }
            // Scalar loads return out of order, so every wait on them is lgkmcnt(0): the operands of THIS state are waited
            // for here, BEFORE the requests of the next state are issued -- which then have the whole state to arrive.
            asm volatile("" : "+s"(Bt[0]), "+s"(Ct[0]));
            __builtin_amdgcn_sched_barrier(0);
            RLPROF(3)                                               // scalar operand wait
            if (s + 1 < NS) {
                rl_load_bc(Bw + (s + 1) * B_ns + T * m, nch, Bn);
                if (!SUMMARY) rl_load_bc(Cw + (s + 1) * C_ns + T * m, nch, Cn);
            }
            __builtin_amdgcn_sched_barrier(0);                      // keep the requests of state s + 2 behind state s
            float x = xst[s];
#pragma unroll
            for (int kk = 0; kk < T; ++kk) {
                const int k = REV ? T - 1 - kk : kk;
                const float a = fast_exp2(dl[k] * A2[s]);
                x = fmaf(a, x, dlu[k] * Bt[k]);
                if (!SUMMARY) y[k] = fmaf(Ct[k], x, y[k]);
#if !(SIGMA_RL_ABL & 2) && !SIGMA_BWDR_FULL
                if (CK && kk == T / 2 - 1) ck[((long)(m * 2) * N + n0 + s) * 64] = x;
#endif
            }
            xst[s] = x;
            if (SUMMARY) Pacc[s] = fmaf(dsum, A2[s], Pacc[s]);
#if !(SIGMA_RL_ABL & 2)
#if !SIGMA_BWDR_FULL
            if (CK) ck[((long)(m * 2 + 1) * N + n0 + s) * 64] = x;  // state after memory tile m (scan order)
#endif
#endif
            __builtin_amdgcn_sched_barrier(0);
            RLPROF(4)                                               // state loop
        }
#if SIGMA_BWDR_FULL && !(SIGMA_RL_ABL & 2)
        if (CK) rl_store_ck<NS, NS * NW / 4>(ck - lane + (long)m * N * 64, n0, lane, xst);     // the states after memory tile m (scan order)
#endif
#if SIGMA_RL_ABL & 16
        asm volatile("" :: "v"(y[0]), "v"(y[5]), "v"(y[10]), "v"(y[15]));
#else
        if (!SUMMARY) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const v4f t = {y[4 * c], y[4 * c + 1], y[4 * c + 2], y[4 * c + 3]};
                sEx[sw * 256 + rl_unit(c, lane)] = t;
            }
        }
#endif
        RLPROF(5)                                                   // exchange writes
        rl_barrier();
        RLPROF(6)                                                   // barrier 2
        if (!SUMMARY && (NW == 4 || rcw)) {
            v4f acc = du4;
#if !(SIGMA_RL_ABL & 16)
#pragma unroll
            for (int w = 0; w < NW; ++w) acc += sEx[w * 256 + rl_unit(cc, rr)];
#endif
            if (valid) *reinterpret_cast<v4f*>(o_row + T * m) = acc;
        }
        RLPROF(7)                                                   // sum of the waves, store
    };
    // The partial tile (fewer than 16 positions: the memory-last one) is the last step of a forward group and the first
    // of a reversed one: it is peeled off the loop (two variants of the body inside one loop double its register need).
    const int tail_tile = (L % T) ? ntiles - 1 : -1;                // memory tile with fewer than 16 positions
    int it0 = 0, it1 = nst;
    if (REV && nst > 0 && tile_of(0) == tail_tile) { step(0, std::true_type{}); it0 = 1; }
    if (!REV && nst > 0 && tile_of(nst - 1) == tail_tile) it1 = nst - 1;
    for (int it = it0; it < it1; ++it) step(it, std::false_type{});
    if (it1 < nst) step(it1, std::true_type{});
    RLPROF_FLUSH(g_fwdr_prof)
    if (touch_acc == 1.2345678e-30f) o_row[0] = touch_acc;          // keeps the touches alive (never true in practice)
    if (SUMMARY) {
        float2* __restrict__ sm = reinterpret_cast<float2*>(p.fsumm);
#pragma unroll
        for (int s = 0; s < NS; ++s)
            sm[(((long)seg * p.batch * (p.dim >> 6) + rowblock) * N + n0 + s) * 64 + lane] = make_float2(fast_exp2(Pacc[s]), xst[s]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the same tile walk as a SOFTWARE PIPELINE with ONE workgroup barrier per tile.  The phase
// profile of the body above (profiles/r05_rowlane_phases.jsonl) has a wave in its state loop for only half of its life:
// the rest are per-tile phases in series -- operand wait + softplus + LDS hand-off, barrier, LDS read-back, the scalar
// operands of the first state, exchange, barrier, sum + store -- each an exposed latency.  Here the (row, chunk) duties
// of tile t + 1 (softplus, hand-off) and of tile t - 1 (sum of the state waves, store) are issued BESIDE the state loop
// of tile t, on double-buffered LDS blocks, so that the barrier at the end of step t publishes sProc[t + 1] and sEx[t]
// together; B / C of the first state of tile t + 1 are requested during the last state of tile t.  Four state waves only
// (NW = 4); the pre-pass of the sequence split (MODE 2) stays on the body above.
//   LDS: sProc [2 buffers][delta, delta * u][4][64] float4 (16 KB) + sEx [2 buffers][4 waves][4][64] float4 (32 KB) = 48 KB:
//   three workgroups per CU.
template <int NS, bool REV, bool CK>
__device__ __forceinline__ void scan_fwdp_body(const FwdArgs& p, float* smem, int b, int g, int rbg, int seg) {
    constexpr int T = kRT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int sw = __builtin_amdgcn_readfirstlane(tid >> 6);      // state wave 0..3
    const int N = p.N, L = p.L;
    const int ntiles = (L + T - 1) / T;
    v4f* sProc = reinterpret_cast<v4f*>(smem);                    // [2][2][4][64] float4
    v4f* sEx = sProc + 4 * 256;                                   // [2][4 waves][4][64] float4

    // thread as (row, chunk)
    const int rr = (sw << 4) | (lane >> 2), cc = lane & 3;
    const int rpg = p.rows_per_group;
    const int row0 = g * rpg + rbg * kRRows;
    const int r_rc = row0 + rr;
    const int ur_rc = r_rc - ((g - (g >> p.u_gshift)) * rpg);     // same row of group g >> u_gshift
    const float* __restrict__ u_row = reinterpret_cast<const float*>(p.u) + (long)b * p.u_bs + (long)ur_rc * p.u_ds + 4 * cc;
    const float* __restrict__ d_row = reinterpret_cast<const float*>(p.delta) + (long)b * p.dt_bs + (long)r_rc * p.dt_ds + 4 * cc;
    float* __restrict__ o_row = reinterpret_cast<float*>(p.out) + (long)b * p.o_bs + (long)r_rc * p.o_ds + 4 * cc;
    const int pr_rc = param_row(r_rc, g, rpg, p.pswap);
    const float bias = p.bias ? p.bias[pr_rc] : 0.0f;
    const float Dd = p.D ? p.D[pr_rc] : 0.0f;
    const bool use_softplus = p.softplus != 0;
    const int unit_rc = rl_unit(cc, rr);

    // thread as row lane: states [sw * NS, sw * NS + NS) of row row0 + lane
    const int pr_ln = param_row(row0 + lane, g, rpg, p.pswap);
    const int n0 = sw * NS;
    float A2[NS], xst[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        A2[s] = p.A[(long)pr_ln * p.A_ds + (long)(n0 + s) * p.A_ns] * kLog2e;
        xst[s] = 0.0f;
    }
    const float* Bw = reinterpret_cast<const float*>(p.B) + (long)b * p.B_bs + (long)g * p.B_gs + (long)n0 * p.B_ns;
    const float* Cw = reinterpret_cast<const float*>(p.C) + (long)b * p.C_bs + (long)g * p.C_gs + (long)n0 * p.C_ns;
    const int B_ns = (int)p.B_ns, C_ns = (int)p.C_ns;               // host: (N - 1) * stride + L fits 31 bits
    const long rowblock = (long)b * (p.dim >> 6) + (row0 >> 6);
#if SIGMA_BWDR_FULL
    float* __restrict__ ck = CK ? p.x + rowblock * ntiles * N * 64 : nullptr;      // one checkpoint block per tile (rl_store_ck)
#else
    float* __restrict__ ck = CK ? p.x + rowblock * ntiles * 2 * N * 64 + (long)n0 * 64 + lane : nullptr;
#endif

    const int st_lo = p.segs > 1 ? seg * p.seg_tiles : 0;
    const int st_hi = p.segs > 1 ? (st_lo + p.seg_tiles < ntiles ? st_lo + p.seg_tiles : ntiles) : ntiles;
    const int nst = st_hi - st_lo;
    if (nst <= 0) return;
    if (p.segs > 1 && seg > 0) {
        const float2* __restrict__ sm = reinterpret_cast<const float2*>(p.fsumm);
        // eight segments' summaries requested at once: one iteration per segment waited ~0.4 us for its own four loads, and the
        // LAST segment's workgroup entered its tile loop up to 47 of those late (round 6: (1,768,19200) runs 48 segments)
        const long sm_seg = (long)p.batch * (p.dim >> 6) * N * 64;
        const float2* __restrict__ sm0 = sm + ((long)rowblock * N + n0) * 64 + lane;
        int t = 0;
        for (; t + 8 <= seg; t += 8) {
            float2 pe[8][NS];
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int s = 0; s < NS; ++s) pe[k][s] = sm0[(long)(t + k) * sm_seg + s * 64];
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int s = 0; s < NS; ++s) xst[s] = fmaf(pe[k][s].x, xst[s], pe[k][s].y);
        }
        for (; t < seg; ++t) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float2 pe = sm0[(long)t * sm_seg + s * 64];
                xst[s] = fmaf(pe.x, xst[s], pe.y);
            }
        }
    }
    auto tile_of = [&](int it) { return REV ? (ntiles - 1 - (st_lo + it)) : (st_lo + it); };

    // (row, chunk) duty: raw u / delta of a tile -> delta, delta * u into sProc[buf]; returns D * u
    auto preprocess = [&](const v4f uu, const v4f dd, bool valid, int buf) {
        v4f dl4, dlu4, du4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float raw = dd[j] + bias;
            float sig;
#if SIGMA_RL_ABL & 32
            const float sp = raw * raw; sig = 1.0f;
#else
            const float sp = softplus_ref(raw, sig);
#endif
            float d = use_softplus ? sp : raw;
            d = valid ? d : 0.0f;                                   // identity element past the end (a = 1, b = 0)
            dl4[j] = d;
            dlu4[j] = d * uu[j];
            du4[j] = Dd * uu[j];
        }
        sProc[buf * 512 + unit_rc] = dl4;
        sProc[buf * 512 + 256 + unit_rc] = dlu4;
        return du4;
    };

    // ---- pipeline fill: tile 0 pre-processed, tile 1 in flight, B / C of the first state requested
    // loads of the (row, chunk) duty always run (from the row start when the chunk lies past the end) so that the vector-
    // memory operations of a step are the same on every path and the compiler's vmcnt bookkeeping stays exact
    auto load_rc = [&](const float* __restrict__ row, int mt, bool on) {
        const bool ok = on && T * mt + 4 * cc < L;
        const v4f z = {0.0f, 0.0f, 0.0f, 0.0f};
        const v4f t = *reinterpret_cast<const v4f*>(row + (ok ? T * mt : -4 * cc));
        return ok ? t : z;
    };
    int m = tile_of(0);
    v4f du_cur, du_prev = {0.0f, 0.0f, 0.0f, 0.0f};
    {
        const v4f u0 = load_rc(u_row, m, true), d0 = load_rc(d_row, m, true);
        du_cur = preprocess(u0, d0, T * m + 4 * cc < L, 0);
    }
    // u / delta travel PF tiles ahead of the state loop in PF register sets (set it & (PF - 1) feeds step it): with PF = 2
    // a set is consumed and refilled by every second step, so no copy ever touches a register with a load in flight
    constexpr int PF = SIGMA_FWDR_PF;
    v4f u_s0 = load_rc(u_row, tile_of(nst > 1 ? 1 : 0), nst > 1), d_s0 = load_rc(d_row, tile_of(nst > 1 ? 1 : 0), nst > 1);
    v4f u_s1 = u_s0, d_s1 = d_s0;
    if (PF == 2) { u_s1 = load_rc(u_row, tile_of(nst > 2 ? 2 : 0), nst > 2); d_s1 = load_rc(d_row, tile_of(nst > 2 ? 2 : 0), nst > 2); }
    float Bn[T], Cn[T];
    {
        const int nch0 = (L - T * m) >> 2;
        rl_load_bc(Bw + T * m, nch0 < 4 ? nch0 : 4, Bn);
        rl_load_bc(Cw + T * m, nch0 < 4 ? nch0 : 4, Cn);
    }
    rl_barrier();

    // One step.  FAST: the steady state -- a previous tile to finish (it > 0), two more tiles ahead (it + 2 < nst) and no
    // partial tile among tiles it - 1 .. it + 2: straight-line code, no masks, full-width scalar requests.  Otherwise every
    // condition is tested at run time (the first step, the last two, and the neighbours of the partial tile).
    RLPROF_DECL
    auto step = [&](int it, auto fast_tag, v4f& u_set, v4f& d_set) {
        constexpr bool FAST = decltype(fast_tag)::value;
        m = tile_of(it);
        const int buf = it & 1;
        const bool first = FAST ? false : it == 0;
        const bool more = FAST ? true : it + 1 < nst;
        const bool moreP = FAST ? true : it + 1 + PF < nst;
        const int mn = tile_of(more ? it + 1 : it);
        const int mp = tile_of(first ? it : it - 1);
        // ---- row lane: this tile's delta / delta * u
        float dl[T], dlu[T], y[T];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const v4f a = sProc[buf * 512 + rl_unit(c, lane)], bq = sProc[buf * 512 + 256 + rl_unit(c, lane)];
#pragma unroll
            for (int j = 0; j < 4; ++j) { dl[4 * c + j] = a[j]; dlu[4 * c + j] = bq[j]; y[4 * c + j] = 0.0f; }
        }
        // ---- (row, chunk): the partial sums of the previous tile (garbage on the first step: not stored)
#if SIGMA_RL_ABL & 16
        v4f e0 = du_prev, e1 = du_prev, e2 = du_cur, e3 = du_cur;
#else
        v4f e0 = sEx[((buf ^ 1) * 4 + 0) * 256 + unit_rc], e1 = sEx[((buf ^ 1) * 4 + 1) * 256 + unit_rc];
        v4f e2 = sEx[((buf ^ 1) * 4 + 2) * 256 + unit_rc], e3 = sEx[((buf ^ 1) * 4 + 3) * 256 + unit_rc];
#endif
        // ---- (row, chunk): next tile's operands -> sProc[buf ^ 1]
        v4f uu = u_set, dd = d_set;
#if SIGMA_RL_PROF
        RLPROF(1)                                                   // LDS read requests, addresses
        asm volatile("" : "+v"(uu), "+v"(dd));
        RLPROF(0)                                                   // wait for the next tile's u / delta (vmcnt)
#endif
        const v4f du_next = preprocess(uu, dd, FAST ? true : (more && T * mn + 4 * cc < L), buf ^ 1);
        // ... and tile it + 1 + PF starts flying into the registers just read (after their last use: no copy, no early wait)
        asm volatile("" :: "v"(du_next));
        __builtin_amdgcn_sched_barrier(0);
        {
            const int m2 = tile_of(moreP ? it + 1 + PF : it);
            if (FAST) {
#if SIGMA_RL_ABL & 4
                u_set = rl_load4(u_row + T * m2, true);
                d_set = rl_load4(d_row + T * m2, true);
#else
                u_set = *reinterpret_cast<const v4f*>(u_row + T * m2);
                d_set = *reinterpret_cast<const v4f*>(d_row + T * m2);
#endif
            } else {
                u_set = load_rc(u_row, m2, moreP);
                d_set = load_rc(d_row, m2, moreP);
            }
        }
        {
            const v4f acc = du_prev + ((e0 + e1) + (e2 + e3));
            if (FAST || (!first && T * mp + 4 * cc < L)) *reinterpret_cast<v4f*>(o_row + T * mp) = acc;
        }
        du_prev = du_cur;
        du_cur = du_next;
        RLPROF(7)                                                   // (row, chunk) duties: softplus, hand-off, sum, store
        // LDS reads and scalar loads share one counter: retire the reads before the first scalar request of the state loop,
        // or its wait for them (lgkmcnt(0)) would also sit out that request
        asm volatile("" : "+v"(dl[0]), "+v"(dlu[0]), "+v"(dl[4]), "+v"(dlu[4]), "+v"(dl[8]), "+v"(dlu[8]), "+v"(dl[12]), "+v"(dlu[12]));
        __builtin_amdgcn_sched_barrier(0);
        RLPROF(2)                                                   // LDS reads of this tile's operands
        // ---- state loop
        const int nch = FAST ? 4 : ((L - T * m) >> 2 < 4 ? (L - T * m) >> 2 : 4);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float Bt[T], Ct[T];
#pragma unroll
            for (int k = 0; k < T; ++k) { Bt[k] = Bn[k]; Ct[k] = Cn[k]; }
            // scalar loads return out of order: wait for THIS state's operands (lgkmcnt(0)), then request the next ones
            asm volatile("" : "+s"(Bt[0]), "+s"(Ct[0]));
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < NS) {
                rl_load_bc(Bw + (s + 1) * B_ns + T * m, nch, Bn);
                rl_load_bc(Cw + (s + 1) * C_ns + T * m, nch, Cn);
            } else if (more) {                                      // first state of the next tile
                const int nchn = FAST ? 4 : ((L - T * mn) >> 2 < 4 ? (L - T * mn) >> 2 : 4);
                rl_load_bc(Bw + T * mn, nchn, Bn);
                rl_load_bc(Cw + T * mn, nchn, Cn);
            }
            __builtin_amdgcn_sched_barrier(0);
            float x = xst[s];
#pragma unroll
            for (int kk = 0; kk < T; ++kk) {
                const int k = REV ? T - 1 - kk : kk;
                const float a = fast_exp2(dl[k] * A2[s]);
                x = fmaf(a, x, dlu[k] * Bt[k]);
                y[k] = fmaf(Ct[k], x, y[k]);
#if !(SIGMA_RL_ABL & 2) && !SIGMA_BWDR_FULL
                if (CK && kk == T / 2 - 1) ck[((long)(m * 2) * N + s) * 64] = x;
#endif
            }
            xst[s] = x;
#if !(SIGMA_RL_ABL & 2)
#if !SIGMA_BWDR_FULL
            if (CK) ck[((long)(m * 2 + 1) * N + s) * 64] = x;       // state after memory tile m (scan order)
#endif
#endif
            // the sums are complete HERE: without this the instruction selector parks the C * x products of a state behind
            // the last one (its scheduling barriers bind the machine scheduler only) and keeps every x and C alive until then
            asm volatile("" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]),
                              "+v"(y[8]), "+v"(y[9]), "+v"(y[10]), "+v"(y[11]), "+v"(y[12]), "+v"(y[13]), "+v"(y[14]), "+v"(y[15]));
            __builtin_amdgcn_sched_barrier(0);
        }
#if SIGMA_BWDR_FULL && !(SIGMA_RL_ABL & 2)
        if (CK) rl_store_ck<NS, NS>(ck + (long)m * N * 64, n0, lane, xst);            // the states after memory tile m (scan order)
#endif
        RLPROF(4)                                                   // state loop
#if SIGMA_RL_ABL & 16
        asm volatile("" :: "v"(y[0]), "v"(y[5]), "v"(y[10]), "v"(y[15]));
#else
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const v4f t = {y[4 * c], y[4 * c + 1], y[4 * c + 2], y[4 * c + 3]};
            sEx[(buf * 4 + sw) * 256 + rl_unit(c, lane)] = t;
        }
#endif
        RLPROF(5)                                                   // exchange writes
        rl_barrier();
        RLPROF(6)                                                   // barrier
    };
    // steady-state range [lo, hi]: it >= 1, it + 1 + PF <= nst - 1, and the partial tile (the first step of a reversed
    // group, the last of a forward one) not among tiles it - 1 .. it + 1 + PF
    const int tail_tile = (L % T) ? ntiles - 1 : -1;                // memory tile with fewer than 16 positions
    int lo = (REV && tile_of(0) == tail_tile) ? 2 : 1;
    int hi = nst - 1 - (PF + 1) - ((!REV && tile_of(nst - 1) == tail_tile) ? 1 : 0);
    if (hi < lo) { lo = nst; hi = nst - 1; }
    // steps outside the steady state use set 0 only (an odd step swaps the sets around itself: copies, and the waits they
    // imply, cost nothing there)
    auto slow = [&](int it) {
        const bool odd = PF == 2 && (it & 1);
        if (odd) { const v4f tu = u_s0, td = d_s0; u_s0 = u_s1; d_s0 = d_s1; u_s1 = tu; d_s1 = td; }
        step(it, std::false_type{}, u_s0, d_s0);
        if (odd) { const v4f tu = u_s0, td = d_s0; u_s0 = u_s1; d_s0 = d_s1; u_s1 = tu; d_s1 = td; }
    };
    int it = 0;
    const int lo2 = PF == 2 ? ((lo + 1) & ~1) : lo;                 // steady state starts on an even step (register set 0)
    for (; it < lo2 && it < nst; ++it) slow(it);
    if (PF == 2) {
        for (; it + 1 <= hi; it += 2) { step(it, std::true_type{}, u_s0, d_s0); step(it + 1, std::true_type{}, u_s1, d_s1); }
    } else {
        for (; it <= hi; ++it) step(it, std::true_type{}, u_s0, d_s0);
    }
    for (; it < nst; ++it) slow(it);
    RLPROF_FLUSH(g_fwdr_prof)
    // ---- drain: the sum and the store of the last tile
    {
        v4f acc = du_prev;
#pragma unroll
        for (int w = 0; w < 4; ++w) acc += sEx[((((nst - 1) & 1)) * 4 + w) * 256 + unit_rc];
        const int mp = tile_of(nst - 1);
        if (T * mp + 4 * cc < L) *reinterpret_cast<v4f*>(o_row + T * mp) = acc;
    }
}

template <int NS, int NW, int MODE>
__global__ void __launch_bounds__(64 * NW)
scan_fwdr_kernel(const FwdArgs p) {
    constexpr bool SUMMARY = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int S = SUMMARY ? p.segs - 1 : p.segs;              // the last segment needs no summary
    const int PS = p.rowblocks * S;                           // workgroups per (batch, group): row blocks x segments
    const int per_b = p.G * PS;
    const int b = lb / per_b;
    const int rem = lb - b * per_b;
    const int g = rem / PS;
    const int rem2 = rem - g * PS;
    const int rbg = rem2 / S;
    const int seg = rem2 - rbg * S;
#if SIGMA_FWDR_PIPE
    if constexpr (NW == 4 && MODE != 2) {
        if ((p.rev_mask >> g) & 1u) scan_fwdp_body<NS, true, MODE == 1>(p, smem, b, g, rbg, seg);
        else scan_fwdp_body<NS, false, MODE == 1>(p, smem, b, g, rbg, seg);
        return;
    }
#endif
    if ((p.rev_mask >> g) & 1u) scan_fwdr_body<NS, NW, true, MODE>(p, smem, b, g, rbg, seg);
    else scan_fwdr_body<NS, NW, false, MODE>(p, smem, b, g, rbg, seg);
}

template <int NS, int NW, int MODE>
static hipError_t launch_fwdr_t(const FwdArgs& a, hipStream_t stream) {
    const int S = MODE == 2 ? a.segs - 1 : a.segs;
    const int grid = a.batch * a.G * a.rowblocks * S;
    const size_t lds = fwdr_lds_bytes(NW);
    auto kern = scan_fwdr_kernel<NS, NW, MODE>;
    static std::atomic<size_t> lds_cap[kMaxDevices];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDevices) dev = 0;
    if (lds > 48 * 1024 && lds > lds_cap[dev].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_cap[dev].store(lds, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, stream, a);
    return hipGetLastError();
}

// a.R = state waves per row block: dstate / R states per wave
template <int MODE>
static hipError_t launch_fwdr_ns(const FwdArgs& a, hipStream_t stream) {
    switch (a.N * 100 + a.R) {
        case 1604: return launch_fwdr_t<4, 4, MODE>(a, stream);
        case 1608: return launch_fwdr_t<2, 8, MODE>(a, stream);
        case 1616: return launch_fwdr_t<1, 16, MODE>(a, stream);
        case 804: return launch_fwdr_t<2, 4, MODE>(a, stream);
        case 808: return launch_fwdr_t<1, 8, MODE>(a, stream);
        case 404: return launch_fwdr_t<1, 4, MODE>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

// a.rowblocks = 64-row blocks per (batch, group); a.R = state waves; a.segs segments of a.seg_tiles tiles (a.fsumm when segs > 1)
hipError_t launch_scan_fwdr(const FwdArgs& a, hipStream_t stream) {
    if (a.segs > 1) {
        hipError_t e = launch_fwdr_ns<2>(a, stream);
        if (e != hipSuccess) return e;
    }
    return a.x ? launch_fwdr_ns<1>(a, stream) : launch_fwdr_ns<0>(a, stream);
}

}  // namespace sigma
