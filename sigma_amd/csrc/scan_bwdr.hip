// scan_bwdr.hip -- selective-scan backward, "row-lane" mapping (gfx950 / MI355X, wave64, f32 IO, ckpt_pitch 16).
//
// Same operator as scan_bwd4.hip (reference: models/encoders/selective_scan/csrc/selective_scan/
// selective_scan_bwd_kernel.cuh:66-308, reverse_scan.cuh:18-401; mathematics SURVEY.md App. E.2); the mapping is the one
// of scan_fwdr.hip (scan_rowlane.h): a lane is a channel row, a wave is 64 rows x NS states, B / C are SGPR operands,
// tiles of 16 positions, the state entering a tile comes from the forward's per-tile checkpoint.  Per element-state:
//     forward replay   v_mul, v_exp, v_mul, v_fma                                   (a, x kept for the tile: 32 registers)
//     reverse          v_fma (dx = g C + e), v_mul (e = a dx), v_fma (sum dx B), v_mul (t = e x_prev),
//                      v_fma (sum A t), v_fma (dA), v_mul (dx delta u), v_mul (g x)
// = 11 plain VALU + 1 v_exp, against 17.5 + 1.1 of the quad-row kernel's mathematics plus its mapping overhead (row
// scans, broadcasts, slabs, column sums; profiles/r03_bwd4_issue_model.md: 1170 clocks per 640 element-states).  What
// this mapping pays instead: the dB / dC terms of a (state, position) are spread over the 64 LANES of a wave.  They are
// summed by a transpose-reduce network per state and tile: 32 registers (16 positions x {dB, dC}) -> 1 register in which
// every lane pair holds the total of one slot: 24 lane swaps (v_permlane32_swap, v_permlane16_swap) + 24 adds + 15 DPP
// adds + 1 select, ~20 clocks per 64 element-states.  The totals of a row block leave through the per-block slabs of the caller's
// workspace (reduce_partials_kernel adds the blocks of a group in a fixed order: deterministic).
//
// Few rows: the sequence is cut into S segments; rev_summary (MODE 1 of the same body) writes per segment, row and state
// (decay product P, reverse value E from zero), a segment composes the summaries to its right into its incoming carry.
#include "scan_device.h"
#include "scan_launch.h"
#include "scan_rowlane.h"

#include <atomic>
#include <type_traits>

// 1: u / delta / dout of the next tile travel by LDS-DMA (12 KB of LDS more: two workgroups per CU); 0: in registers
#ifndef SIGMA_RL_DMA
#define SIGMA_RL_DMA 1
#endif

// A/B build knob (round 5, measured and NOT taken: profiles/r05_bwdr_mfma_reduce_ab.txt).  1: the dB / dC terms of a
// (position, state) are summed over the 64 lanes (= rows) of a wave on the MATRIX pipe -- v_mfma_f32_16x16x4_f32 with the
// term as the A operand (A[i = lane & 15][k = lane >> 4]) and a one-hot B (column c = the position) adds the 4-lane sums
// sum_k term[i + 16 k] into column c of a 16 x 16 accumulator (exact fp32), 128 MFMAs per tile and wave instead of 96 lane
// swaps + 60 DPP adds.  Results identical to the oracle's tolerance, but (16,3072,1200,N16) runs in 1083 us against 748:
// the f32 MFMA holds its SIMD for ~80 clocks next to dependent VALU work, not the 32 of a bare MFMA stream.
// 0 (default): the transpose-reduce network on the vector ALU.
// 1: B / C of the next tile are pulled into L2 a tile ahead by one vector load per wave (round 4); 0: A/B builds without
// development only: a deliberately mis-counted wait (tests/test_isa_waits_cpu.py must refuse such a build)
#ifndef SIGMA_BWDR_WAIT_SKEW
#define SIGMA_BWDR_WAIT_SKEW 0
#endif
#ifndef SIGMA_BWDR_TOUCH
#define SIGMA_BWDR_TOUCH 1
#endif
#ifndef SIGMA_RL_MFMA
#define SIGMA_RL_MFMA 0
#endif

#if SIGMA_RL_PROF
__device__ unsigned long long g_bwdr_prof[16];
#endif
__device__ unsigned int g_bwdr_chain_timeouts;     // chained walk: hand-over waits that ran out (their row blocks were poisoned)

namespace sigma {

// waits of the chained walk that ran out since the last call (synchronises the device; resets the counter)
hipError_t bwdr_chain_timeouts_read(unsigned int* out) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwdr_chain_timeouts), sizeof(unsigned int));
    if (e != hipSuccess || *out == 0) return e;
    const unsigned int z = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bwdr_chain_timeouts), &z, sizeof(z));
}

#if SIGMA_RL_PROF
hipError_t bwdr_prof_read(unsigned long long* out16) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_bwdr_prof), 16 * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bwdr_prof), z, sizeof(z));
}
#else
hipError_t bwdr_prof_read(unsigned long long* out16) { for (int i = 0; i < 16; ++i) out16[i] = 0; return hipSuccess; }
#endif

namespace {

// a + b with the halves / rows regrouped (semantics pinned by tools/ubench/lane_ops_probe.hip, as scan_quad.h):
//   swap32: lanes 0-31 of the result = a[0:32] + a[32:64], lanes 32-63 = b[0:32] + b[32:64]
//   swap16: DPP rows of the result = {a.r0 + a.r1, b.r0 + b.r1, a.r2 + a.r3, b.r2 + b.r3}
__device__ __forceinline__ float rl_fold32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float rl_fold16(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

// Sum of the dB / dC terms of a tile over the 64 lanes.  The 32 slots of a tile are (memory half hm, position j in the
// half, array dB / dC).  rl_reduce_half takes the eight fold32(dB term, dC term) registers of ONE half (w[j]: lanes 0-31
// = dB of position j summed over lane pairs l, l + 32; lanes 32-63 = dC) to ONE register in which every lane holds the sum
// over 16 lanes of one slot; rl_reduce_finish merges the registers of the two halves: on return lane l holds the total of
//     array  l >> 5,   memory position  8 * bit1(l) + 2 * bit2(l) + bit3(l) + 4 * bit4(l)       (each total in lanes l, l ^ 1).
// Stages: halves of the wave (permlane32_swap: runs as soon as a position is done), row pairs (permlane16_swap: positions
// j and j + 4), inside a DPP row distance 8 and 4 by DPP adds under bank masks (a disabled lane keeps what the other add
// of the pair wrote), distance 2 by two quad permutes + one select, distance 1 by a quad permute.  The network was
// checked lane by lane with a symbolic model (tools/rowlane_reduce_model.py prints the table above).
// Hazard: a VALU write followed by a DPP read of the same VGPR needs two wait states.
__device__ __forceinline__ float rl_reduce_half(const float (&w)[8]) {
    const float z0 = rl_fold16(w[0], w[4]), z1 = rl_fold16(w[1], w[5]), z2 = rl_fold16(w[2], w[6]), z3 = rl_fold16(w[3], w[7]);
    float q0, q1, ph;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %2, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %2, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        : "=&v"(q0), "=&v"(q1), "=&v"(ph)
        : "v"(z0), "v"(z1), "v"(z2), "v"(z3));
    return ph;
}

// p0 = rl_reduce_half of memory half 0, p1 = of memory half 1
__device__ __forceinline__ float rl_reduce_finish(float p0, float p1) {
    float t0, t1, o;
    const unsigned long long lanes_bit1 = 0xCCCCCCCCCCCCCCCCull;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32 %2, %0, %1, %5\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        : "=&v"(t0), "=&v"(t1), "=&v"(o)
        : "v"(p0), "v"(p1), "s"(lanes_bit1));
    return o;
}

typedef float v8f_a16 __attribute__((ext_vector_type(8), aligned(16)));
typedef const __attribute__((address_space(4))) v8f_a16* cv8p_t;

// B or C of one (state, half tile): 8 consecutive floats at a wave-uniform address -> SGPRs; nch = valid 16-byte chunks
// of the half (2 except in a partial last tile), missing ones read as zero
__device__ __forceinline__ void rl_load_bc8(const float* base, int nch, float (&v)[8]) {
#if SIGMA_RL_ABL & 1
    float c = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((int)(reinterpret_cast<uintptr_t>(base) & 0xffff) | 0x3f000000));
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = c; asm volatile("" : "+s"(c)); }
    return;
#endif
    if (nch >= 2) {
        const auto t = *reinterpret_cast<cv8p_t>(reinterpret_cast<uintptr_t>(base));
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = t[k];
    } else {
        const v4f z = {0.0f, 0.0f, 0.0f, 0.0f};
        const v4f c0 = nch > 0 ? *reinterpret_cast<cv4p_t>(reinterpret_cast<uintptr_t>(base)) : z;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = c0[j]; v[4 + j] = 0.0f; }
    }
}

// 16 bytes of a row; `ok` false: zeros (the load itself always runs, from the row start, so that the VMEM operations of a
// tile are the same on every path and the compiler's vmcnt bookkeeping stays exact instead of falling back to vmcnt(0))
__device__ __forceinline__ v4f rl_load4u(const float* __restrict__ row, int off, bool ok) {
    const v4f z = {0.0f, 0.0f, 0.0f, 0.0f};
#if SIGMA_RL_ABL & 4
    const float c = 0.001f * (float)((reinterpret_cast<uintptr_t>(row) + off) & 0xff);
    const v4f tc = {c, 0.5f * c, 0.25f * c, -c};
    return ok ? tc : z;
#endif
    const v4f t = *reinterpret_cast<const v4f*>(row + (ok ? off : 0));
    return ok ? t : z;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// The backward proper.  A workgroup is one block of 64 rows x 4 state waves (NS = dstate / 4 states each).  Per tile of 16
// positions: the (row, chunk) threads pre-process u / delta / dout once (softplus, delta * u, softplus') into LDS; then
// the tile is walked as TWO halves of 8 positions, second half in scan order first.  Per half every wave runs, for each of
// its states: forward replay of the half from the forward's checkpoint (pitch 8), the reverse recurrence, the sums over
// the states of dx B and A2 dx a x_prev in 16 registers, and the lane-reduce of the dB / dC terms.  Halves rather than
// whole tiles keep the per-lane arrays at 8 entries (~120 VGPRs, 3-4 waves per SIMD) where whole tiles need > 200.
// One piece of work: scan steps [st_lo, st_hi) of one row block, walked from the last to the first.  The reverse carry
// entering at st_hi is zero (cin 0: the end of the sequence), the composition of the segment summaries (cin 1), or handed
// over by the workgroup that walked the steps above (cin 2: chained walk); cout: this piece hands its carry on.
struct RlPiece { int b, g, rbg, st_lo, st_hi, cin, cout, seg; };

template <int NS, bool REV>
__device__ __forceinline__ void scan_bwdr_body(const BwdArgs& q, float* smem, const RlPiece pc) {
    constexpr int T = kRT, H = kRT / 2;
    // wave-uniform by construction (block index arithmetic); said explicitly, because integer divisions are expanded on
    // the vector ALU and would otherwise make every address derived from them look divergent (no scalar loads)
    const int b = __builtin_amdgcn_readfirstlane(pc.b), g = __builtin_amdgcn_readfirstlane(pc.g);
    const int rbg = __builtin_amdgcn_readfirstlane(pc.rbg), seg = __builtin_amdgcn_readfirstlane(pc.seg);
    const FwdArgs& p = q.f;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int sw = __builtin_amdgcn_readfirstlane(tid >> 6);      // state wave 0..3
    const int N = p.N, L = p.L;
    const int ntiles = (L + T - 1) / T;
    v4f* sProc = reinterpret_cast<v4f*>(smem);                    // [5][4][64] float4: delta, delta * u, dout; u, softplus' (epilogue)
    v4f* sEx = sProc + 5 * 256;                                   // [4 waves][2][4][64] float4: sum dx B, sum A2 dx a x_prev
#if SIGMA_RL_ABL & 128
    v4f* sRaw = sEx + 4 * 256;
#else
    v4f* sRaw = sEx + 8 * 256;                                    // [3][256 threads] float4: u, delta, dout of the NEXT tile (LDS-DMA)
#endif
    const unsigned raw_base = (unsigned)(uintptr_t)(lptr_t)(sRaw + sw * 64);   // this wave's 1 KB of array 0

    // thread as (row, chunk)
    const int rr = (sw << 4) | (lane >> 2), cc = lane & 3;
    const int rpg = p.rows_per_group;
    const int row0 = g * rpg + rbg * kRRows;
    const int r_rc = row0 + rr;
    const int ur_rc = r_rc - ((g - (g >> p.u_gshift)) * rpg);     // same row of group g >> u_gshift
    const int gr_rc = r_rc - ((g - (g >> q.g_gshift)) * rpg);
    // rows of this thread as wave-uniform bases (the block's first row; SGPRs) + 32-bit element offsets (the row inside
    // the block and the chunk): five 64-bit per-thread pointers less across the state loops (host: 64 rows x stride < 2^29)
    const int ur0 = row0 - ((g - (g >> p.u_gshift)) * rpg), gr0 = row0 - ((g - (g >> q.g_gshift)) * rpg);
    const float* u_blk = reinterpret_cast<const float*>(p.u) + (long)b * p.u_bs + (long)ur0 * p.u_ds;
    const float* d_blk = reinterpret_cast<const float*>(p.delta) + (long)b * p.dt_bs + (long)row0 * p.dt_ds;
    const float* g_blk = reinterpret_cast<const float*>(q.dout) + (long)b * q.g_bs + (long)gr0 * q.g_ds;
    float* du_blk = reinterpret_cast<float*>(q.du) + (long)b * q.du_bs + (long)row0 * q.du_ds;
    float* dd_blk = reinterpret_cast<float*>(q.ddelta) + (long)b * q.dd_bs + (long)row0 * q.dd_ds;
    const int u_off = rr * (int)p.u_ds + 4 * cc, d_off = rr * (int)p.dt_ds + 4 * cc, g_off = rr * (int)q.g_ds + 4 * cc;
    const int du_off = rr * (int)q.du_ds + 4 * cc, dd_off = rr * (int)q.dd_ds + 4 * cc;
    (void)ur_rc; (void)gr_rc;
    const int pr_rc = param_row(r_rc, g, rpg, p.pswap);
    const float bias = p.bias ? p.bias[pr_rc] : 0.0f;
    const float Dd = p.D ? p.D[pr_rc] : 0.0f;
    float dD_acc = 0.0f, dbias_acc = 0.0f;

    // thread as row lane
    const int pr_ln = param_row(row0 + lane, g, rpg, p.pswap);
    const int n0 = sw * NS;
    float A2[NS], ecar[NS], dAacc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        A2[s] = p.A[(long)pr_ln * p.A_ds + (long)(n0 + s) * p.A_ns] * kLog2e;
        ecar[s] = 0.0f;
        dAacc[s] = 0.0f;
    }
    const float* Bw = reinterpret_cast<const float*>(p.B) + (long)b * p.B_bs + (long)g * p.B_gs + (long)n0 * p.B_ns;
    const float* Cw = reinterpret_cast<const float*>(p.C) + (long)b * p.C_bs + (long)g * p.C_gs + (long)n0 * p.C_ns;
    const int B_ns = (int)p.B_ns, C_ns = (int)p.C_ns;               // host: (N - 1) * stride + L fits 31 bits
    const long rowblock = (long)b * (p.dim >> 6) + (row0 >> 6);
    // checkpoints: x[((rowblock * ntiles + tile) * 2 + h) * N + n) * 64 + lane], h = 0 after the first scan half of the tile
#if SIGMA_BWDR_FULL
    // checkpoints (one per tile): x[((rowblock * ntiles + tile) * N + n) * 64 + lane] = state after memory tile `tile` in scan order
    const float* ck = p.x + rowblock * ntiles * N * 64;                         // wave-uniform; rl_load_ck adds the wave's slot
#else
    const float* ck = p.x + rowblock * ntiles * 2 * N * 64 + (long)n0 * 64;     // wave-uniform; + lane at the use
#endif

    // dB / dC of this row block: the caller's tensors when the block is the whole group, else its slab of the workspace
    float* __restrict__ oB;
    float* __restrict__ oC;
    long o_nsB, o_nsC;
    if (q.P == 1) {
        oB = q.dB + (long)b * q.dB_bs + (long)g * q.dB_gs; o_nsB = q.dB_ns;
        oC = q.dC + (long)b * q.dC_bs + (long)g * q.dC_gs; o_nsC = q.dC_ns;
    } else {
        const long slab = (((long)rbg * p.batch + b) * p.G + g) * (long)N * L;
        oB = q.ws_dB + slab; oC = q.ws_dC + slab; o_nsB = L; o_nsC = L;
    }
#if SIGMA_RL_MFMA
    // after the folds of the accumulators (below) lane l holds the total of array l >> 5 at memory position l & 15 (lanes l
    // and l ^ 16 hold the same total and store it twice)
    const int o_pos = lane & 15;
    float oh[T];                                                    // B operand of the reducing MFMA: one-hot column c
#pragma unroll
    for (int c = 0; c < T; ++c) oh[c] = (lane & 15) == c ? 1.0f : 0.0f;
#else
    // lane l of rl_reduce_finish: array l >> 5, memory position below (lanes l and l ^ 1 hold the same total and store it twice)
    const int o_pos = ((lane >> 1) & 1) * 8 + ((lane >> 2) & 1) * 2 + ((lane >> 3) & 1) + ((lane >> 4) & 1) * 4;
#endif
    float* __restrict__ o_lane = (lane < 32 ? oB + (long)n0 * o_nsB : oC + (long)n0 * o_nsC) + o_pos;
    const int o_ns_lane = (int)(lane < 32 ? o_nsB : o_nsC);

    // scan steps [st_lo, st_hi) of this workgroup, walked from the last to the first
    const int st_lo = __builtin_amdgcn_readfirstlane(pc.st_lo), st_hi = __builtin_amdgcn_readfirstlane(pc.st_hi);
    const int nst = st_hi - st_lo;
    const long rb_carry = (((long)b * p.G + g) * q.P + rbg) * N * 64 + (long)n0 * 64 + lane;   // this lane's slot of the hand-over
    if (pc.cin == 2) {
        // chained walk: wait for the workgroup that walked the steps above (one relaxed poll loop in one lane, then an
        // agent-scope acquire, then the barrier: MI355X_MICROARCH.md, inter-workgroup visibility).  The producer ran this
        // row block FIRST and this is our LAST piece, so the flag is normally up; the spin is bounded all the same.
        int* timed_out = reinterpret_cast<int*>(smem);                  // LDS is not in use yet
        if (tid == 0) {
            int* flag = q.chain_flag + ((b * p.G + g) * q.P + rbg);
            int spin = 0;
            for (; spin < (1 << 22) && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0; ++spin)
                __builtin_amdgcn_s_sleep(16);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            *timed_out = spin >= (1 << 22);
            if (spin >= (1 << 22)) atomicAdd(&g_bwdr_chain_timeouts, 1u);   // "rl_chain_timeouts": the host can tell why there are NaNs
        }
        __syncthreads();
        // A wait that ran out (the producer is not resident: another stream's kernel holds its slot) must not pass for a
        // result: the carry becomes NaN, which reaches du / ddelta / dA of this row block and every loss after it
        const float poison = *timed_out ? __builtin_nanf("") : 0.0f;
        __syncthreads();                                                // the flag word is free again before the tile loop writes LDS
#pragma unroll
        for (int s = 0; s < NS; ++s) ecar[s] = q.chain_carry[rb_carry + s * 64] + poison;
    }
    if (pc.cin == 1) {
        // reverse carry entering the right end = composition of the summaries of the segments after this one
        const float2* __restrict__ sm = reinterpret_cast<const float2*>(q.summ);
        // eight summaries requested at once (as in the forward: one iteration per segment waited for its own loads)
        const long sm_seg = (long)p.batch * (p.dim >> 6) * N * 64;
        const float2* __restrict__ sm0 = sm + ((long)rowblock * N + n0) * 64 + lane;
        int t = q.S - 1;
        for (; t - 8 >= seg; t -= 8) {
            float2 pe[8][NS];
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int s = 0; s < NS; ++s) pe[k][s] = sm0[(long)(t - 1 - k) * sm_seg + s * 64];
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int s = 0; s < NS; ++s) ecar[s] = fmaf(pe[k][s].x, ecar[s], pe[k][s].y);
        }
        for (; t > seg; --t) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float2 pe = sm0[(long)(t - 1) * sm_seg + s * 64];
                ecar[s] = fmaf(pe.x, ecar[s], pe.y);
            }
        }
    }
    auto tile_of = [&](int st) { return REV ? (ntiles - 1 - st) : st; };   // memory tile of scan step st
    // state entering scan half hs of scan step st: after the first half of the same tile / after the previous tile
    // (before the first scan step there is no checkpoint: the load is clamped to a valid one and the caller scales by zero
    // at the point of USE -- a select next to the load would make the compiler wait for it there)
    [[maybe_unused]] auto ck_in = [&](int st, int hs, int s) {
#if SIGMA_RL_ABL & 2
        return 0.001f * st;
#endif
#if SIGMA_BWDR_FULL
        (void)hs; (void)s;
        return 0.0f;                                                // (the whole-tile walk loads its NS states at once: ck_load)
#else
        const int tl = hs == 1 ? tile_of(st) : tile_of(st > 0 ? st - 1 : 0);
        return ck[(unsigned)(((tl * 2 + (hs == 1 ? 0 : 1)) * N + s) * 64 + lane)];      // host: floats per row block < 2^31
#endif
    };

    // u / delta / dout of a tile travel by LDS-DMA one tile ahead (with register targets the compiler, short of
    // registers, moved the requests next to their use: three exposed memory latencies per tile).  Chunks past the end of
    // the row are requested at the row start (finite values) and masked when they are read back.
    auto request = [&](int mt) {
        const int tl = T * mt + 4 * cc < L ? T * mt : -4 * cc;      // past the end: the row start
        rl_dma16s(u_blk, (unsigned)(u_off + tl) * 4u, raw_base);
        rl_dma16s(d_blk, (unsigned)(d_off + tl) * 4u, raw_base + 256 * 16);
        rl_dma16s(g_blk, (unsigned)(g_off + tl) * 4u, raw_base + 512 * 16);
    };
    int st = st_hi - 1;
    int m = tile_of(st);
#if SIGMA_RL_DMA
    request(m);
#else
    v4f u_nx = rl_load4u(u_blk + u_off - 4 * cc, T * m + 4 * cc, T * m + 4 * cc < L);
    v4f d_nx = rl_load4u(d_blk + d_off - 4 * cc, T * m + 4 * cc, T * m + 4 * cc < L);
    v4f g_nx = rl_load4u(g_blk + g_off - 4 * cc, T * m + 4 * cc, T * m + 4 * cc < L);
#endif
    float x1_nx[NS];                                                // entering the second scan half of the next step (FULL: entering the next step)
#if SIGMA_BWDR_FULL
    // the NS states entering scan step st_ = those after the previous step's tile: one access per lane (before the first
    // step there is none: the load is clamped to a valid block and the value scaled by zero where it is used)
    auto ck_load = [&](int st_, float (&xo)[NS]) {
#if SIGMA_RL_ABL & 2
        for (int s = 0; s < NS; ++s) xo[s] = 0.001f * st_;
        return;
#endif
        rl_load_ck<NS, NS>(ck + (unsigned)(tile_of(st_ > 0 ? st_ - 1 : 0) * N * 64), n0, lane, xo);      // host: floats per row block < 2^31
    };
    ck_load(st, x1_nx);
#else
#pragma unroll
    for (int s = 0; s < NS; ++s) x1_nx[s] = ck_in(st, 1, s);
#endif
    // B / C of the NEXT tile are pulled into L2 a tile ahead (see scan_fwdr.hip)
    const float* __restrict__ bc_touch = ((lane & 1) ? Cw : Bw) + (long)((lane >> 1) & (NS - 1)) * ((lane & 1) ? C_ns : B_ns);
    float touch_nx = 0.0f, touch_acc = 0.0f;
    RLPROF_DECL

    auto step = [&](int it, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        st = st_hi - 1 - it;
        m = tile_of(st);
        const bool valid = !TAIL || T * m + 4 * cc < L;
        const int nch = TAIL ? (L - T * m) >> 2 : 4;                // valid 16-byte chunks of this tile
        const int hm_first = REV ? 0 : 1;                           // memory half of the second scan half (walked first)
#if SIGMA_BWDR_FULL
        (void)hm_first;
        float Bn[T], Cn[T];
        rl_load_bc(Bw + T * m, nch, Bn);                            // first state's B / C: requested before anything waits
        rl_load_bc(Cw + T * m, nch, Cn);
#else
        float Bn[H], Cn[H];
        rl_load_bc8(Bw + T * m + H * hm_first, nch - 2 * hm_first, Bn);      // first step's B / C: requested before anything waits
        rl_load_bc8(Cw + T * m + H * hm_first, nch - 2 * hm_first, Cn);
#endif
        __builtin_amdgcn_sched_barrier(0);
#if SIGMA_RL_DMA
        // This tile's u / delta / dout have landed.  A full step of the loop issues 3 + 2 NS vector-memory operations
        // after its requests (the touch, NS checkpoint loads, NS dB/dC stores, du, ddelta), which stay in
        // flight; the partial tile may skip some of them, so it (and the step after it: the caller waits) drains the counter.
        // (whole-tile walk: the touch, ONE checkpoint load, NS dB/dC stores, du, ddelta)
        if (TAIL) rl_dma_wait(); else rl_dma_wait_keep<(SIGMA_BWDR_FULL ? 3 + NS : 2 + 2 * NS) + (SIGMA_BWDR_TOUCH ? 1 : 0) + SIGMA_BWDR_WAIT_SKEW>();
        RLPROF(8)                                                   // wait for this tile's u / delta / dout (LDS-DMA)
        const v4f uu = sRaw[tid], dd = sRaw[256 + tid];
        v4f g4 = sRaw[512 + tid];
        const v4f zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
        g4 = valid ? g4 : zero4;                                    // dout reads as zero past the end
#else
        const v4f uu = u_nx, dd = d_nx, g4 = g_nx;
#endif
        float x1[NS], x0[NS];
        const float x0_scale = st > 0 ? 1.0f : 0.0f; (void)x0_scale;
        touch_acc += touch_nx;
#if SIGMA_BWDR_FULL
#pragma unroll
        for (int s = 0; s < NS; ++s) { x1[s] = x1_nx[s]; x0[s] = 0.0f; }               // x1: the state entering this tile
#else
#pragma unroll
        for (int s = 0; s < NS; ++s) { x1[s] = x1_nx[s]; x0[s] = ck_in(st, 0, s); }     // x0: used half a tile from now
#endif
        {                                                           // next tile's operands fly during this tile
            const bool more = it + 1 < nst;
            const int mn = tile_of(more ? st - 1 : st);
#if SIGMA_RL_DMA
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the reads of this tile's bytes are done
            RLPROF(9)                                               // raw operand reads back from LDS (+ B / C of the first state)
            request(mn);
#else
            u_nx = rl_load4u(u_blk + u_off - 4 * cc, T * mn + 4 * cc, more && T * mn + 4 * cc < L);
            d_nx = rl_load4u(d_blk + d_off - 4 * cc, T * mn + 4 * cc, more && T * mn + 4 * cc < L);
            g_nx = rl_load4u(g_blk + g_off - 4 * cc, T * mn + 4 * cc, more && T * mn + 4 * cc < L);
#endif
#if SIGMA_BWDR_TOUCH
            touch_nx = bc_touch[T * mn];
#endif
        }
        RLPROF(10)                                                  // requests of the next tile
        {
            v4f dl4, dlu4, sg4;
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
                sg4[j] = p.softplus ? sig : 1.0f;
                d = valid ? d : 0.0f;                               // identity element past the end (a = 1, b = 0)
                dl4[j] = d;
                dlu4[j] = d * uu[j];
                dD_acc = fmaf(g4[j], uu[j], dD_acc);
            }
            // the epilogue of this tile reads its operands back (they would cost 16 registers across the state loops)
            sProc[rl_unit(cc, rr)] = dl4;
            sProc[256 + rl_unit(cc, rr)] = dlu4;
            sProc[512 + rl_unit(cc, rr)] = g4;
            sProc[768 + rl_unit(cc, rr)] = uu;
            sProc[1024 + rl_unit(cc, rr)] = sg4;
        }
        RLPROF(0)                                                   // operand wait, softplus, LDS writes
        rl_barrier();
        RLPROF(1)                                                   // barrier 1

#if SIGMA_BWDR_FULL
        {
            float dl[T], dlu[T], gg[T], sdxB[T], sAx[T];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const v4f a4 = sProc[rl_unit(c, lane)], b4 = sProc[256 + rl_unit(c, lane)], g4r = sProc[512 + rl_unit(c, lane)];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dl[4 * c + j] = a4[j]; dlu[4 * c + j] = b4[j]; gg[4 * c + j] = g4r[j];
                    sdxB[4 * c + j] = 0.0f; sAx[4 * c + j] = 0.0f;
                }
            }
            // LDS reads and scalar loads share one counter: retire the reads before the first scalar request of the state loop
            asm volatile("" : "+v"(dl[0]), "+v"(dlu[0]), "+v"(gg[0]), "+v"(dl[4]), "+v"(dlu[4]), "+v"(gg[4]),
                              "+v"(dl[8]), "+v"(dlu[8]), "+v"(gg[8]), "+v"(dl[12]), "+v"(dlu[12]), "+v"(gg[12]));
            __builtin_amdgcn_sched_barrier(0);
            RLPROF(2)                                               // LDS reads
            const float xin_scale = st > 0 ? 1.0f : 0.0f;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float Bt[T], Ct[T];
#pragma unroll
                for (int k = 0; k < T; ++k) { Bt[k] = Bn[k]; Ct[k] = Cn[k]; }
                // scalar loads return out of order: wait for THIS state's operands (lgkmcnt(0)), then request the next state's
                asm volatile("" : "+s"(Bt[0]), "+s"(Ct[0]));
                __builtin_amdgcn_sched_barrier(0);
                RLPROF(3)                                           // scalar operand wait
                if (s + 1 < NS) {
                    rl_load_bc(Bw + (s + 1) * B_ns + T * m, nch, Bn);
                    rl_load_bc(Cw + (s + 1) * C_ns + T * m, nch, Cn);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float xin = x1[s] * xin_scale;
                float a[T], xs[T];
                // ---- forward replay of the tile from the state entering it
                {
                    float x = xin;
#pragma unroll
                    for (int kk = 0; kk < T; ++kk) {
                        const int k = REV ? T - 1 - kk : kk;
                        a[k] = fast_exp2(dl[k] * A2[s]);
                        x = fmaf(a[k], x, dlu[k] * Bt[k]);
                        xs[k] = x;
                    }
                }
                // ---- reverse: dx_k = g_k C_k + e_{k+1}, e_k = a_k dx_k
                float e = ecar[s];
                float w[H], ph_first = 0.0f;
#pragma unroll
                for (int kk = T - 1; kk >= 0; --kk) {
                    const int k = REV ? T - 1 - kk : kk;
                    const float dx = fmaf(gg[k], Ct[k], e);
                    e = a[k] * dx;
                    const int kp = REV ? k + 1 : k - 1;             // previous position in scan order
                    const float xprev = kk > 0 ? xs[kk > 0 ? kp : k] : xin;
                    sdxB[k] = fmaf(dx, Bt[k], sdxB[k]);
                    const float t = e * xprev;                      // dx * a_k * x_{k-1}
                    sAx[k] = fmaf(A2[s], t, sAx[k]);                // x ln 2 in the epilogue
                    dAacc[s] = fmaf(dl[k], t, dAacc[s]);
#if SIGMA_RL_ABL & 64
                    w[k & 7] = dx * dlu[k] + gg[k] * xs[k];
#else
                    w[k & 7] = rl_fold32(dx * dlu[k], gg[k] * xs[k]);  // this row's terms of dB[n, l] and dC[n, l], wave halves summed
#endif
                    if (kk == H) {                                  // the half walked first is complete: memory half REV ? 0 : 1
#if SIGMA_RL_ABL & 64
                        ph_first = ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
#else
                        ph_first = rl_reduce_half(w);
#endif
                    }
                }
                ecar[s] = e;
#if SIGMA_RL_ABL & 64
                const float tot = ph_first + (((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7])));
#else
                const float ph_second = rl_reduce_half(w);          // memory half REV ? 1 : 0
                const float tot = REV ? rl_reduce_finish(ph_first, ph_second) : rl_reduce_finish(ph_second, ph_first);
#endif
                if (!TAIL || T * m + o_pos < L) o_lane[s * o_ns_lane + T * m] = tot;
                __builtin_amdgcn_sched_barrier(0);
                RLPROF(4)                                           // state loop
            }
            // ---- the sums over the states of the four waves meet in LDS
#if SIGMA_RL_ABL & 16
            asm volatile("" :: "v"(sdxB[0]), "v"(sdxB[5]), "v"(sAx[2]), "v"(sAx[7]), "v"(sdxB[9]), "v"(sAx[14]));
#else
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const v4f t1 = {sdxB[4 * c], sdxB[4 * c + 1], sdxB[4 * c + 2], sdxB[4 * c + 3]};
                const v4f t2 = {sAx[4 * c], sAx[4 * c + 1], sAx[4 * c + 2], sAx[4 * c + 3]};
                sEx[(sw * 2) * 256 + rl_unit(c, lane)] = t1;
                sEx[(sw * 2 + 1) * 256 + rl_unit(c, lane)] = t2;
            }
#endif
            ck_load(it + 1 < nst ? st - 1 : st, x1_nx);              // the states entering the NEXT step
            RLPROF(5)                                               // exchange writes
        }
#else
#if SIGMA_RL_MFMA
        float psB[NS], psC[NS];                                     // 4-lane sums of the half walked first, per state
#else
        float psave[NS];
#endif
#pragma unroll
        for (int hs = 1; hs >= 0; --hs) {
            const int hm = REV ? 1 - hs : hs;                       // memory half
            float dl[H], dlu[H], gg[H], sdxB[H], sAx[H];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const v4f a4 = sProc[rl_unit(2 * hm + c, lane)], b4 = sProc[256 + rl_unit(2 * hm + c, lane)];
                const v4f g4r = sProc[512 + rl_unit(2 * hm + c, lane)];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dl[4 * c + j] = a4[j]; dlu[4 * c + j] = b4[j]; gg[4 * c + j] = g4r[j];
                    sdxB[4 * c + j] = 0.0f; sAx[4 * c + j] = 0.0f;
                }
            }
            // LDS reads and scalar loads share one counter: retire the reads before the first scalar request of the state
            // loop, or its wait for them (lgkmcnt(0)) would also sit out that request
            asm volatile("" : "+v"(dl[0]), "+v"(dlu[0]), "+v"(gg[0]), "+v"(dl[4]), "+v"(dlu[4]), "+v"(gg[4]));
            __builtin_amdgcn_sched_barrier(0);
            RLPROF(2)                                               // LDS reads
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float Bt[H], Ct[H];
#pragma unroll
                for (int k = 0; k < H; ++k) { Bt[k] = Bn[k]; Ct[k] = Cn[k]; }
                // Scalar loads return out of order, so every wait on them is lgkmcnt(0): the operands of THIS step are
                // waited for here, BEFORE the requests of the next step are issued -- which then have a whole step to arrive.
                asm volatile("" : "+s"(Bt[0]), "+s"(Ct[0]));
                __builtin_amdgcn_sched_barrier(0);
                RLPROF(3)                                           // scalar operand wait
                if (s + 1 < NS) {
                    rl_load_bc8(Bw + (s + 1) * B_ns + T * m + H * hm, nch - 2 * hm, Bn);
                    rl_load_bc8(Cw + (s + 1) * C_ns + T * m + H * hm, nch - 2 * hm, Cn);
                } else if (hs == 1) {                               // first state of the other half
                    rl_load_bc8(Bw + T * m + H * (1 - hm), nch - 2 * (1 - hm), Bn);
                    rl_load_bc8(Cw + T * m + H * (1 - hm), nch - 2 * (1 - hm), Cn);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float xin = hs == 1 ? x1[s] : x0[s] * x0_scale;
                float a[H], xs[H];
                // ---- forward replay from the checkpoint
                {
                    float x = xin;
#pragma unroll
                    for (int kk = 0; kk < H; ++kk) {
                        const int k = REV ? H - 1 - kk : kk;
                        a[k] = fast_exp2(dl[k] * A2[s]);
                        x = fmaf(a[k], x, dlu[k] * Bt[k]);
                        xs[k] = x;
                    }
                }
                // ---- reverse: dx_k = g_k C_k + e_{k+1}, e_k = a_k dx_k
                float e = ecar[s];
#if SIGMA_RL_MFMA
                v4f accB = {0.0f, 0.0f, 0.0f, 0.0f}, accC = {0.0f, 0.0f, 0.0f, 0.0f};
#else
                float w[H];
#endif
#pragma unroll
                for (int kk = H - 1; kk >= 0; --kk) {
                    const int k = REV ? H - 1 - kk : kk;
                    const float dx = fmaf(gg[k], Ct[k], e);
                    e = a[k] * dx;
                    const int kp = REV ? k + 1 : k - 1;             // previous position in scan order
                    const float xprev = kk > 0 ? xs[kk > 0 ? kp : k] : xin;
                    sdxB[k] = fmaf(dx, Bt[k], sdxB[k]);
                    const float t = e * xprev;                      // dx * a_k * x_{k-1}
                    sAx[k] = fmaf(A2[s], t, sAx[k]);                // x ln 2 in the epilogue
                    dAacc[s] = fmaf(dl[k], t, dAacc[s]);
                    // this row's terms of dB[n, l] and dC[n, l]
#if SIGMA_RL_MFMA
#if SIGMA_RL_ABL & 64
                    accB[0] += dx * dlu[k]; accC[0] += gg[k] * xs[k];
#else
                    accB = __builtin_amdgcn_mfma_f32_16x16x4f32(dx * dlu[k], oh[H * hm + k], accB, 0, 0, 0);
                    accC = __builtin_amdgcn_mfma_f32_16x16x4f32(gg[k] * xs[k], oh[H * hm + k], accC, 0, 0, 0);
#endif
                }
                ecar[s] = e;
                // accumulator of lane l: column l & 15 (position), rows 4 (l >> 4) + r of the 16 four-lane sums
                const float tB = (accB[0] + accB[1]) + (accB[2] + accB[3]), tC = (accC[0] + accC[1]) + (accC[2] + accC[3]);
                if (hs == 1) {
                    psB[s] = tB; psC[s] = tC;                       // columns of the other half are zero in these
                } else {
#if SIGMA_RL_ABL & 64
                    const float tot = (tB + psB[s]) + (tC + psC[s]);
#else
                    // lanes l, l + 16, l + 32, l + 48 hold the four quarters of position l & 15: wave halves (dB to lanes
                    // 0-31, dC to 32-63), then DPP row pairs
                    const float f = rl_fold32(tB + psB[s], tC + psC[s]);
                    const float tot = rl_fold16(f, f);
#endif
                    if (!TAIL || T * m + o_pos < L) o_lane[s * o_ns_lane + T * m] = tot;
                }
#else
                    // ... the two wave halves summed at once
#if SIGMA_RL_ABL & 64
                    w[k] = dx * dlu[k] + gg[k] * xs[k];
#else
                    w[k] = rl_fold32(dx * dlu[k], gg[k] * xs[k]);
#endif
                }
                ecar[s] = e;
#if SIGMA_RL_ABL & 64
                const float ph = ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
#else
                const float ph = rl_reduce_half(w);
#endif
                if (hs == 1) {
                    psave[s] = ph;
                } else {
#if SIGMA_RL_ABL & 64
                    const float tot = ph + psave[s];
#else
                    const float tot = hm == 0 ? rl_reduce_finish(ph, psave[s]) : rl_reduce_finish(psave[s], ph);
#endif
                    if (!TAIL || T * m + o_pos < L) o_lane[s * o_ns_lane + T * m] = tot;
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
                RLPROF(4)                                           // state loop
            }
            // ---- the sums over the states of the four waves meet in LDS
#if SIGMA_RL_ABL & 16
            asm volatile("" :: "v"(sdxB[0]), "v"(sdxB[5]), "v"(sAx[2]), "v"(sAx[7]));
#else
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const v4f t1 = {sdxB[4 * c], sdxB[4 * c + 1], sdxB[4 * c + 2], sdxB[4 * c + 3]};
                const v4f t2 = {sAx[4 * c], sAx[4 * c + 1], sAx[4 * c + 2], sAx[4 * c + 3]};
#if SIGMA_RL_ABL & 128
                sEx[((sw & 1) * 2) * 256 + rl_unit(2 * hm + c, lane)] = t1;
                sEx[((sw & 1) * 2 + 1) * 256 + rl_unit(2 * hm + c, lane)] = t2;
#else
                sEx[(sw * 2) * 256 + rl_unit(2 * hm + c, lane)] = t1;
                sEx[(sw * 2 + 1) * 256 + rl_unit(2 * hm + c, lane)] = t2;
#endif
            }
#endif
            if (hs == 1) {                                          // entering the second scan half of the NEXT step
                const int stn = it + 1 < nst ? st - 1 : st;
#pragma unroll
                for (int s = 0; s < NS; ++s) x1_nx[s] = ck_in(stn, 1, s);
            }
            RLPROF(5)                                               // exchange writes
        }
#endif
        rl_barrier();
        RLPROF(6)                                                   // barrier 2
        {
            v4f S1 = {0.0f, 0.0f, 0.0f, 0.0f}, S2 = {0.0f, 0.0f, 0.0f, 0.0f};
#if !(SIGMA_RL_ABL & 16)
#pragma unroll
            for (int w = 0; w < 4; ++w) { S1 += sEx[((w & (SIGMA_RL_ABL & 128 ? 1 : 3)) * 2) * 256 + rl_unit(cc, rr)]; S2 += sEx[((w & (SIGMA_RL_ABL & 128 ? 1 : 3)) * 2 + 1) * 256 + rl_unit(cc, rr)]; }
#endif
            const v4f dle = sProc[rl_unit(cc, rr)], ge = sProc[512 + rl_unit(cc, rr)];
            const v4f ue = sProc[768 + rl_unit(cc, rr)], sge = sProc[1024 + rl_unit(cc, rr)];
#if SIGMA_RL_PROF
            asm volatile("" : "+v"(S1), "+v"(S2));
#endif
            RLPROF(11)                                              // epilogue: exchange reads
            v4f duv, ddv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                duv[j] = fmaf(Dd, ge[j], dle[j] * S1[j]);
                ddv[j] = fmaf(ue[j], S1[j], S2[j] * kLn2) * sge[j];
                dbias_acc += valid ? ddv[j] : 0.0f;
            }
            if (valid) {
                *reinterpret_cast<v4f*>(du_blk + (unsigned)(du_off + T * m)) = duv;
                *reinterpret_cast<v4f*>(dd_blk + (unsigned)(dd_off + T * m)) = ddv;
            }
        }
        RLPROF(7)                                                   // epilogue
    };
    // The partial tile (fewer than 16 positions: the memory-last one) is the first step of a forward group and the last
    // of a reversed one: it is peeled off the loop.  With both variants of the body inside ONE loop the register
    // allocation of the whole loop doubled (254 instead of 128 VGPRs at 16 states).
    const int tail_tile = (L % T) ? ntiles - 1 : -1;                // memory tile with fewer than 16 positions
    int it0 = 0, it1 = nst;
#if SIGMA_RL_DMA
    rl_dma_wait();                                                  // the first requests: nothing younger behind them yet
#endif
    if (!REV && nst > 0 && tile_of(st_hi - 1) == tail_tile) {
        step(0, std::true_type{});
        it0 = 1;
#if SIGMA_RL_DMA
        rl_dma_wait();
#endif
    }
    if (REV && nst > 0 && tile_of(st_lo) == tail_tile) it1 = nst - 1;
    for (int it = it0; it < it1; ++it) step(it, std::false_type{});
    if (it1 < nst) step(it1, std::true_type{});
    RLPROF_FLUSH(g_bwdr_prof)
    if (pc.cout) {
        // hand the reverse carry to the workgroup that walks the steps below: plain stores, barrier, one lane releases
        // at agent scope, drains its stores (the compiler may drop the wait: asm) and raises the flag
#pragma unroll
        for (int s = 0; s < NS; ++s) q.chain_carry[rb_carry + s * 64] = ecar[s];
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(q.chain_flag + ((b * p.G + g) * q.P + rbg), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (touch_acc == 1.2345678e-30f) du_blk[0] = touch_acc;        // keeps the touches alive (never true in practice)

    // per-row results: one atomicAdd per (row, state) / row and workgroup (several workgroups only with segments)
#pragma unroll
    for (int s = 0; s < NS; ++s) atomicAdd(q.dA + (long)pr_ln * q.dA_ds + (long)(n0 + s) * q.dA_ns, dAacc[s]);
    // the four chunk threads of a row are neighbouring lanes
    dD_acc += dpp_take<0xB1, 0xF>(0.0f, dD_acc);                    // quad_perm [1,0,3,2]
    dD_acc += dpp_take<0x4E, 0xF>(0.0f, dD_acc);                    // quad_perm [2,3,0,1]
    dbias_acc += dpp_take<0xB1, 0xF>(0.0f, dbias_acc);
    dbias_acc += dpp_take<0x4E, 0xF>(0.0f, dbias_acc);
    if (cc == 0) {
        if (q.dD) atomicAdd(q.dD + pr_rc, dD_acc);
        if (q.dbias) atomicAdd(q.dbias + pr_rc, dbias_acc);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Reverse summaries of the sequence split: for segment seg = 1 .. S-1, row and state the pair (decay product P, reverse
// value E with zero carry): e(left end of the segment) = E + P * e(right end).  The reverse recurrence only.
template <int NS, bool REV>
__device__ __forceinline__ void scan_bwdr_summary_body(const BwdArgs& q, float* smem, int b, int g, int rbg, int seg) {
    constexpr int T = kRT;
    const FwdArgs& p = q.f;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int sw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.N, L = p.L;
    const int ntiles = (L + T - 1) / T;
    v4f* sProc = reinterpret_cast<v4f*>(smem);                    // [2][4][64] float4: delta, dout
    const int rr = (sw << 4) | (lane >> 2), cc = lane & 3;
    const int rpg = p.rows_per_group;
    const int row0 = g * rpg + rbg * kRRows;
    const int r_rc = row0 + rr;
    const int gr_rc = r_rc - ((g - (g >> q.g_gshift)) * rpg);
    const float* __restrict__ d_row = reinterpret_cast<const float*>(p.delta) + (long)b * p.dt_bs + (long)r_rc * p.dt_ds;
    const float* __restrict__ g_row = reinterpret_cast<const float*>(q.dout) + (long)b * q.g_bs + (long)gr_rc * q.g_ds;
    const int pr_rc = param_row(r_rc, g, rpg, p.pswap);
    const float bias = p.bias ? p.bias[pr_rc] : 0.0f;
    const int pr_ln = param_row(row0 + lane, g, rpg, p.pswap);
    const int n0 = sw * NS;
    float A2[NS], ecar[NS], Pacc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        A2[s] = p.A[(long)pr_ln * p.A_ds + (long)(n0 + s) * p.A_ns] * kLog2e;
        ecar[s] = 0.0f;
        Pacc[s] = 0.0f;                                             // log2 of the decay product
    }
    const float* Cw = reinterpret_cast<const float*>(p.C) + (long)b * p.C_bs + (long)g * p.C_gs + (long)n0 * p.C_ns;
    const int C_ns = (int)p.C_ns;
    const long rowblock = (long)b * (p.dim >> 6) + (row0 >> 6);
    const int st_lo = seg * q.seg_tiles;
    const int st_hi = st_lo + q.seg_tiles < ntiles ? st_lo + q.seg_tiles : ntiles;
    const int nst = st_hi - st_lo;
    auto tile_of = [&](int st) { return REV ? (ntiles - 1 - st) : st; };
    int m = tile_of(st_hi - 1);
    v4f d_nx = rl_load4u(d_row, T * m + 4 * cc, T * m + 4 * cc < L);
    v4f g_nx = rl_load4u(g_row, T * m + 4 * cc, T * m + 4 * cc < L);
    for (int it = 0; it < nst; ++it) {
        const int st = st_hi - 1 - it;
        m = tile_of(st);
        const bool valid = T * m + 4 * cc < L;
        const v4f dd = d_nx, g4 = g_nx;
        {
            const bool more = it + 1 < nst;
            const int mn = tile_of(more ? st - 1 : st);
            d_nx = rl_load4u(d_row, T * mn + 4 * cc, more && T * mn + 4 * cc < L);
            g_nx = rl_load4u(g_row, T * mn + 4 * cc, more && T * mn + 4 * cc < L);
        }
        const int nch = (L - T * m) >= T ? 4 : (L - T * m) >> 2;
        v4f dl4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float raw = dd[j] + bias;
            float sig;
            const float sp = softplus_ref(raw, sig);
            const float d = p.softplus ? sp : raw;
            dl4[j] = valid ? d : 0.0f;
        }
        sProc[rl_unit(cc, rr)] = dl4;
        sProc[256 + rl_unit(cc, rr)] = g4;
        rl_barrier();
        float dl[T], gg[T];
        float dsum = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const v4f a4 = sProc[rl_unit(c, lane)], g4r = sProc[256 + rl_unit(c, lane)];
#pragma unroll
            for (int j = 0; j < 4; ++j) { dl[4 * c + j] = a4[j]; gg[4 * c + j] = g4r[j]; dsum += a4[j]; }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float Ct[T];
            rl_load_bc(Cw + s * C_ns + T * m, nch, Ct);
            float e = ecar[s];
#pragma unroll
            for (int kk = T - 1; kk >= 0; --kk) {
                const int k = REV ? T - 1 - kk : kk;
                e = fast_exp2(dl[k] * A2[s]) * fmaf(gg[k], Ct[k], e);
            }
            ecar[s] = e;
            Pacc[s] = fmaf(dsum, A2[s], Pacc[s]);
        }
        rl_barrier();
    }
    float2* __restrict__ sm = reinterpret_cast<float2*>(q.summ);
#pragma unroll
    for (int s = 0; s < NS; ++s)
        sm[(((long)(seg - 1) * p.batch * (p.dim >> 6) + rowblock) * N + n0 + s) * 64 + lane] = make_float2(fast_exp2(Pacc[s]), ecar[s]);
}

// MODE 0: the backward proper; MODE 1: the summaries of segments 1 .. S-1; MODE 2: the backward as a chained walk.
// Waves per SIMD the register allocation is sized for: the 64 KB of LDS of the backward proper (bwdr_lds_bytes) admit two
// workgroups per CU = 2 waves per SIMD = 256 VGPRs; asking for 3 (round 4) capped the 16-state build at 168 VGPRs with 81
// spilled to scratch for an occupancy the LDS never allowed.
#ifndef SIGMA_BWDR_WPS
#define SIGMA_BWDR_WPS 2
#endif
template <int NS, int MODE>
__global__ void __launch_bounds__(256, MODE == 1 ? 4 : SIGMA_BWDR_WPS)
scan_bwdr_kernel(const BwdArgs q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ntiles = (q.f.L + kRT - 1) / kRT;
    if (MODE == 2) {
        // Row blocks laid end to end, each walked from its last scan step to its first: position p = row block *
        // ntiles + (ntiles - 1 - step).  This workgroup owns [p0, p1).  A row block cut at p1 is STARTED here (no
        // incoming carry) and finished by the next workgroup: it is walked first, so that its carry is ready long
        // before the neighbour -- which walks its own cut-off head last -- asks for it.
        // (host: row blocks x tiles < 2^31.  Divisions are expanded on the vector ALU: the results are declared uniform,
        // or the whole piece loop would count as divergent control flow and lose its scalar loads.)
        const int W = q.chain_W, total = q.f.batch * q.f.G * q.P * ntiles;
        const int p0 = (int)blockIdx.x * W, p1 = p0 + W < total ? p0 + W : total;
        if (p0 >= p1) return;
        const int r_first = __builtin_amdgcn_readfirstlane(p0 / ntiles), o_first = p0 - r_first * ntiles;
        const int r_last = __builtin_amdgcn_readfirstlane((p1 - 1) / ntiles), o_end = p1 - r_last * ntiles;   // offsets [.., o_end) of r_last
        auto run = [&](int r, int oa, int ob) {                                                 // offsets [oa, ob) of row block r
            RlPiece pc;
            const int bg = __builtin_amdgcn_readfirstlane(r / q.P);
            pc.rbg = r - bg * q.P;
            pc.b = __builtin_amdgcn_readfirstlane(bg / q.f.G);
            pc.g = bg - pc.b * q.f.G;
            pc.st_hi = ntiles - oa; pc.st_lo = ntiles - ob;
            pc.cin = oa > 0 ? 2 : 0; pc.cout = ob < ntiles ? 1 : 0; pc.seg = 0;
            if ((q.f.rev_mask >> pc.g) & 1u) scan_bwdr_body<NS, true>(q, smem, pc);
            else scan_bwdr_body<NS, false>(q, smem, pc);
            __syncthreads();                                        // the next piece reuses the LDS
        };
        const int head = o_first > 0 ? 1 : 0;                       // r_first continues a walk begun by the previous workgroup
        const int tail = (o_end < ntiles && (r_last != r_first || !head)) ? 1 : 0;
        const int nwhole = (r_last - tail) - (r_first + head) + 1;  // whole row blocks in between (may be <= 0)
        const int npieces = tail + (nwhole > 0 ? nwhole : 0) + head;
        for (int i = 0; i < npieces; ++i) {                         // ONE call site: the cut-off tail first, the head last
            int r, oa = 0, ob = ntiles;
            if (tail && i == 0) { r = r_last; ob = o_end; }
            else if (head && i == npieces - 1) { r = r_first; oa = o_first; ob = r_last == r_first ? o_end : ntiles; }
            else r = r_first + head + (i - tail);
            run(r, oa, ob);
        }
        return;
    }
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int S = MODE == 1 ? q.S - 1 : q.S;
    const int PS = q.P * S;                                   // workgroups per (batch, group): row blocks x segments
    const int per_b = q.f.G * PS;
    const int b = __builtin_amdgcn_readfirstlane(lb / per_b);
    const int rem = lb - b * per_b;
    const int g = __builtin_amdgcn_readfirstlane(rem / PS);
    const int rem2 = rem - g * PS;
    const int rbg = __builtin_amdgcn_readfirstlane(rem2 / S);
    const int seg = rem2 - rbg * S + (MODE == 1 ? 1 : 0);
    const bool rev = (q.f.rev_mask >> g) & 1u;
    if (MODE == 1) {
        if (rev) scan_bwdr_summary_body<NS, true>(q, smem, b, g, rbg, seg);
        else scan_bwdr_summary_body<NS, false>(q, smem, b, g, rbg, seg);
    } else {
        RlPiece pc;
        pc.b = b; pc.g = g; pc.rbg = rbg; pc.seg = seg;
        pc.st_lo = q.S > 1 ? seg * q.seg_tiles : 0;
        pc.st_hi = q.S > 1 ? (pc.st_lo + q.seg_tiles < ntiles ? pc.st_lo + q.seg_tiles : ntiles) : ntiles;
        pc.cin = q.S > 1 ? 1 : 0; pc.cout = 0;
        if (rev) scan_bwdr_body<NS, true>(q, smem, pc);
        else scan_bwdr_body<NS, false>(q, smem, pc);
    }
}

template <int NS, int MODE>
static hipError_t launch_bwdr_t(const BwdArgs& a, hipStream_t stream) {
    const int S = MODE == 1 ? a.S - 1 : a.S;
    const int ntiles = (a.f.L + kRT - 1) / kRT;
    const long nrb = (long)a.f.batch * a.f.G * a.P;
    const int grid = MODE == 2 ? (int)((nrb * ntiles + a.chain_W - 1) / a.chain_W) : (int)(nrb * S);
    const size_t lds = bwdr_lds_bytes(4);
    auto kern = scan_bwdr_kernel<NS, MODE>;
    static std::atomic<size_t> lds_cap[kMaxDevices];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDevices) dev = 0;
    if (lds > 48 * 1024 && lds > lds_cap[dev].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_cap[dev].store(lds, std::memory_order_relaxed);
    }
    if (MODE == 2) {
        hipError_t e = hipMemsetAsync(a.chain_flag, 0, (size_t)nrb * sizeof(int), stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_bwdr_ns(const BwdArgs& a, hipStream_t stream) {
    switch (a.f.N) {
        case 16: return launch_bwdr_t<4, MODE>(a, stream);
        case 8: return launch_bwdr_t<2, MODE>(a, stream);
        case 4: return launch_bwdr_t<1, MODE>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

// workgroups of the backward proper one CU holds (registers, LDS): the chained walk needs every workgroup resident
int bwdr_resident_per_cu(int N) {
    static std::atomic<int> cache[kMaxDevices][3];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDevices) dev = 0;
    const int slot = N == 16 ? 0 : N == 8 ? 1 : 2;
    int v = cache[dev][slot].load(std::memory_order_relaxed);
    if (v > 0) return v;
    const size_t lds = bwdr_lds_bytes(4);
    int n = 0;
    hipError_t e = hipSuccess;
    auto probe = [&](auto kern) {
        if (lds > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 256, lds);
    };
    if (N == 16) probe(scan_bwdr_kernel<4, 2>); else if (N == 8) probe(scan_bwdr_kernel<2, 2>); else probe(scan_bwdr_kernel<1, 2>);
    if (e != hipSuccess || n < 1) n = 1;
    cache[dev][slot].store(n, std::memory_order_relaxed);
    return n;
}

// a.P = 64-row blocks per (batch, group); a.S segments of a.seg_tiles tiles (a.summ when S > 1); a.chain_W > 0: chained walk
hipError_t launch_scan_bwdr(const BwdArgs& a, hipStream_t stream) {
    hipError_t e;
    if (a.chain_W > 0) {
        e = launch_bwdr_ns<2>(a, stream);
    } else {
        if (a.S > 1) {
            e = launch_bwdr_ns<1>(a, stream);
            if (e != hipSuccess) return e;
        }
        e = launch_bwdr_ns<0>(a, stream);
    }
    if (e != hipSuccess || a.P == 1) return e;
    return launch_reduce_partials(a, stream);
}

}  // namespace sigma
