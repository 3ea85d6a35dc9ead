// scan_device.h -- device-side building blocks of the gfx950 selective-scan kernels.
//
// Wave64 / CDNA4 only.  No CUDA compatibility paths, no CUB/hipCUB: the scans are
// written directly on DPP (row_shr / row_bcast / wave_shr) data movement.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sigma {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr int kWave = 64;

// ------------------------------------------------------------------ io types
struct f16_t { uint16_t bits; };
struct bf16_t { uint16_t bits; };

template <typename T> struct io_traits;
template <> struct io_traits<float> {
    static __device__ __forceinline__ float to_float(float v) { return v; }
    static __device__ __forceinline__ float from_float(float v) { return v; }
};
template <> struct io_traits<f16_t> {
    static __device__ __forceinline__ float to_float(f16_t v) {
        return static_cast<float>(__builtin_bit_cast(_Float16, v.bits));
    }
    static __device__ __forceinline__ f16_t from_float(float v) {
        f16_t r; r.bits = __builtin_bit_cast(uint16_t, static_cast<_Float16>(v)); return r;
    }
};
template <> struct io_traits<bf16_t> {
    static __device__ __forceinline__ float to_float(bf16_t v) {
        return __builtin_bit_cast(float, static_cast<uint32_t>(v.bits) << 16);
    }
    static __device__ __forceinline__ bf16_t from_float(float v) {
        uint32_t u = __builtin_bit_cast(uint32_t, v);
        bf16_t r;
        if ((u & 0x7fffffffu) > 0x7f800000u) { r.bits = static_cast<uint16_t>((u >> 16) | 0x40u); return r; }
        u += 0x7fffu + ((u >> 16) & 1u);      // round to nearest even
        r.bits = static_cast<uint16_t>(u >> 16);
        return r;
    }
};

// ---- lane-blocked row access ------------------------------------------------------------
// A lane owns T consecutive SCAN POSITIONS starting at lbase.  In a forward group position p
// is memory index p; in a reversed group (rev) position p is memory index L-1-p, so the lane's
// T positions are the T consecutive memory elements [L-lbase-T, L-lbase) in reverse order.
// Fast path (whole segment in range, `vec`): vector loads of VW = 4 / 2 / 1 elements according
// to the divisibility of T (T = 20, 4 -> 16 B; T = 10 -> 8 B; T = 5 -> 4 B for f32).
template <int T> struct vec_width { static constexpr int value = (T % 4 == 0) ? 4 : ((T % 2 == 0) ? 2 : 1); };

template <typename io_t, int VW>
__device__ __forceinline__ void load_vw(const io_t* __restrict__ p, float* v) {
    if constexpr (VW == 1) {
        v[0] = io_traits<io_t>::to_float(p[0]);
    } else if constexpr (sizeof(io_t) == 4) {
        if constexpr (VW == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
        else { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
    } else {
        io_t e[VW];
        if constexpr (VW == 4) {
            const uint2 t = *reinterpret_cast<const uint2*>(p);
            e[0].bits = static_cast<uint16_t>(t.x & 0xffffu); e[1].bits = static_cast<uint16_t>(t.x >> 16);
            e[2].bits = static_cast<uint16_t>(t.y & 0xffffu); e[3].bits = static_cast<uint16_t>(t.y >> 16);
        } else {
            const uint32_t t = *reinterpret_cast<const uint32_t*>(p);
            e[0].bits = static_cast<uint16_t>(t & 0xffffu); e[1].bits = static_cast<uint16_t>(t >> 16);
        }
#pragma unroll
        for (int i = 0; i < VW; ++i) v[i] = io_traits<io_t>::to_float(e[i]);
    }
}

template <typename io_t, int VW>
__device__ __forceinline__ void store_vw(io_t* __restrict__ p, const float* v) {
    if constexpr (VW == 1) {
        p[0] = io_traits<io_t>::from_float(v[0]);
    } else if constexpr (sizeof(io_t) == 4) {
        if constexpr (VW == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        else *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
    } else {
        const uint32_t lo = static_cast<uint32_t>(io_traits<io_t>::from_float(v[0]).bits) |
                            (static_cast<uint32_t>(io_traits<io_t>::from_float(v[1]).bits) << 16);
        if constexpr (VW == 4) {
            const uint32_t hi = static_cast<uint32_t>(io_traits<io_t>::from_float(v[2]).bits) |
                                (static_cast<uint32_t>(io_traits<io_t>::from_float(v[3]).bits) << 16);
            *reinterpret_cast<uint2*>(p) = make_uint2(lo, hi);
        } else {
            *reinterpret_cast<uint32_t*>(p) = lo;
        }
    }
}

template <typename io_t, int T, bool REV>
__device__ __forceinline__ void load_items(const io_t* __restrict__ row, int lbase, int L, bool vec, float (&v)[T]) {
    constexpr int VW = vec_width<T>::value;
    if (vec && lbase + T <= L) {
        const io_t* __restrict__ src = row + (REV ? (L - lbase - T) : lbase);
#pragma unroll
        for (int q = 0; q < T / VW; ++q) {
            float t[VW];
            load_vw<io_t, VW>(src + VW * q, t);
#pragma unroll
            for (int j = 0; j < VW; ++j) v[REV ? (T - 1 - (VW * q + j)) : (VW * q + j)] = t[j];
        }
    } else {
#pragma unroll
        for (int k = 0; k < T; ++k) {
            const int pos = lbase + k;
            v[k] = (pos < L) ? io_traits<io_t>::to_float(row[REV ? (L - 1 - pos) : pos]) : 0.0f;
        }
    }
}

template <typename io_t, int T, bool REV>
__device__ __forceinline__ void store_items(io_t* __restrict__ row, int lbase, int L, bool vec, const float (&v)[T]) {
    constexpr int VW = vec_width<T>::value;
    if (vec && lbase + T <= L) {
        io_t* __restrict__ dst = row + (REV ? (L - lbase - T) : lbase);
#pragma unroll
        for (int q = 0; q < T / VW; ++q) {
            float t[VW];
#pragma unroll
            for (int j = 0; j < VW; ++j) t[j] = v[REV ? (T - 1 - (VW * q + j)) : (VW * q + j)];
            store_vw<io_t, VW>(dst + VW * q, t);
        }
    } else {
#pragma unroll
        for (int k = 0; k < T; ++k) {
            const int pos = lbase + k;
            if (pos < L) row[REV ? (L - 1 - pos) : pos] = io_traits<io_t>::from_float(v[k]);
        }
    }
}

// 4 consecutive memory elements starting at index m (may be < 0 or reach >= L); missing ones read 0
template <typename io_t>
__device__ __forceinline__ void load4_guard(const io_t* __restrict__ row, int m, int L, bool vec, float (&v)[4]) {
    if (vec && m >= 0 && m + 4 <= L) {
        load_vw<io_t, 4>(row + m, v);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (m + i >= 0 && m + i < L) ? io_traits<io_t>::to_float(row[m + i]) : 0.0f;
    }
}

// ---- staged B/C tiles in LDS -------------------------------------------------------------
// The LDS image of one (state, tile) is TILE = 64*T floats in MEMORY order (for a reversed
// group: memory [L-l0-TILE, L-l0)), unpadded -- which is what global_load_lds can produce
// (wave-uniform LDS base + lane*16 B).  A lane reads its T consecutive floats with
// ds_read_b128 / b64 / b32; for T in {4, 5, 10, 20} the lane stride (T dwords) makes every
// hardware lane group hit distinct banks, so no padding or swizzle is needed
// (MI355X_MICROARCH.md LDS table: b128 groups of 16 lanes over 64 banks, stride 20 -> 16
// distinct multiples of 4; b64 stride 10 -> 32 distinct even banks; b32 stride 5 -> 32 distinct).
// positions k = VW*q .. VW*q+VW-1 of the lane's segment (VW = vec_width<T>)
template <int T, bool REV>
__device__ __forceinline__ void lds_read_chunk(const float* __restrict__ tile, int lane, int q,
                                               float (&v)[vec_width<T>::value]) {
    constexpr int VW = vec_width<T>::value;
    const float* __restrict__ src = REV ? tile + (63 - lane) * T + (T - (q + 1) * VW) : tile + lane * T + q * VW;
    float t[VW];
    if constexpr (VW == 4) { const float4 x = *reinterpret_cast<const float4*>(src); t[0] = x.x; t[1] = x.y; t[2] = x.z; t[3] = x.w; }
    else if constexpr (VW == 2) { const float2 x = *reinterpret_cast<const float2*>(src); t[0] = x.x; t[1] = x.y; }
    else { t[0] = src[0]; }
#pragma unroll
    for (int j = 0; j < VW; ++j) v[j] = t[REV ? (VW - 1 - j) : j];
}

// ------------------------------------------------------------------ math
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }    // v_log_f32
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32

// softplus with the reference's threshold (selective_scan_fwd_kernel.cuh:131-134:
// delta <= 20 ? log1pf(expf(delta)) : delta) and its derivative sigmoid(raw)
// (selective_scan_bwd_kernel.cuh:234-247).  log1p via Kahan's correction so that
// tiny deltas (model init: softplus^-1 of 1e-3) keep full relative precision.
__device__ __forceinline__ float softplus_ref(float raw, float& sig) {
    // branch-free: both arms are cheap and selects keep the unrolled per-element code straight-line
    const float e = fast_exp2(raw * kLog2e);
    const float w = 1.0f + e;
    const float lw = fast_log2(w) * kLn2;
    const float wm1 = w - 1.0f;
    const float sp = (wm1 == 0.0f) ? e : lw * (e * fast_rcp(wm1));
    const bool big = raw > 20.0f;
    sig = big ? 1.0f : e * fast_rcp(w);
    return big ? raw : sp;
}

// ------------------------------------------------------------------ DPP plumbing
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float fallback, float src) {
    // lanes whose DPP source is out of range (bound_ctrl = 0) or that are masked
    // off by ROW_MASK keep `fallback`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
        __builtin_bit_cast(int, fallback), __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xF, false));
}

constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_SHL1 = 0x101, DPP_ROW_SHL2 = 0x102, DPP_ROW_SHL4 = 0x104, DPP_ROW_SHL8 = 0x108;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
constexpr int DPP_WAVE_SHR1 = 0x138, DPP_WAVE_SHL1 = 0x130;

// The scan element is the affine map  state -> p*state + x  ("decay p, injected state x"),
// compose(earlier, later) = (p_l*p_e, p_l*x_e + x_l)          (selective_scan_common.h:92-96).
// The decay is carried as s = log2(p) (a sum of delta*A*log2e terms): composition ADDS s, whose
// identity 0 is exactly what a DPP read with bound_ctrl returns for a missing source lane, and x's
// identity is 0 too -- so one scan step is three instructions with the lane shuffle folded into
// the arithmetic (v_exp_f32, v_add_f32_dpp, v_fmac_f32_dpp) instead of two v_mov_dpp + mul + fma
// with explicit identities.  Written in asm because hipcc does not fold DPP moves into VOP2 users
// here; wait states are inside the string (2 between a VALU write and a DPP read of the same
// VGPR, 1 between v_exp and a use of its result; the instruction order provides them).
#define SIGMA_SCAN_STEP_ASM(CTRL)                                  \
    "v_exp_f32 %2, %0\n\t"                                         \
    "v_add_f32_dpp %0, %0, %0 " CTRL "\n\t"                        \
    "v_fmac_f32_dpp %1, %1, %2 " CTRL "\n\t"

// inclusive scan over the 64 lanes in lane order (lane 0 earliest); s = log2 of the decay
__device__ __forceinline__ void wave_scan_inclusive(float& s, float& x) {
    float p;
    asm(
        "s_nop 1\n\t"
        SIGMA_SCAN_STEP_ASM("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
        SIGMA_SCAN_STEP_ASM("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0")
        SIGMA_SCAN_STEP_ASM("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0")
        SIGMA_SCAN_STEP_ASM("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0")
        SIGMA_SCAN_STEP_ASM("row_bcast:15 row_mask:0xa bank_mask:0xf")
        SIGMA_SCAN_STEP_ASM("row_bcast:31 row_mask:0xc bank_mask:0xf")
        "s_nop 1\n\t"
        : "+v"(s), "+v"(x), "=&v"(p));
}

// value of the previous lane (lane 0 gets `first`)
__device__ __forceinline__ float wave_prev_lane(float v, float first) {
    return dpp_take<DPP_WAVE_SHR1, 0xF>(first, v);
}
// value of the next lane (lane 63 gets `last`)
__device__ __forceinline__ float wave_next_lane(float v, float last) {
    return dpp_take<DPP_WAVE_SHL1, 0xF>(last, v);
}

__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// Suffix ("reverse") inclusive scan: lane 63 is the EARLIEST element of the scan
// order, lane 0 the last.  Element i composes with everything at higher lanes:
//   (s_i, x_i) <- (s_i + S_{i+1}, x_i + 2^{s_i} * X_{i+1})
// Rows are handled with row_shl; the cross-row part uses readlane of the row
// heads (lanes 16/32/48), all wave-uniform.
__device__ __forceinline__ void wave_scan_inclusive_rev(float& s, float& x) {
    float p;
    asm(
        "s_nop 1\n\t"
        SIGMA_SCAN_STEP_ASM("row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
        SIGMA_SCAN_STEP_ASM("row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:0")
        SIGMA_SCAN_STEP_ASM("row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0")
        SIGMA_SCAN_STEP_ASM("row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:0")
        "s_nop 1\n\t"
        : "+v"(s), "+v"(x), "=&v"(p));
    // row aggregates sit at the first lane of each row (lanes 0,16,32,48)
    const int lane = threadIdx.x & 63;
    const float s3 = lane_bcast(s, 48), x3 = lane_bcast(x, 48);
    const float s2 = lane_bcast(s, 32), x2 = lane_bcast(x, 32);
    const float s1 = lane_bcast(s, 16), x1 = lane_bcast(x, 16);
    // suffix aggregates of whole rows: S3 = row3, S2 = row2 o S3, S1 = row1 o S2
    const float sx2 = fmaf(fast_exp2(s2), x3, x2), ss2 = s2 + s3;
    const float sx1 = fmaf(fast_exp2(s1), sx2, x1), ss1 = s1 + ss2;
    const int row = lane >> 4;
    const float tx = row == 0 ? sx1 : (row == 1 ? sx2 : (row == 2 ? x3 : 0.0f));
    const float ts = row == 0 ? ss1 : (row == 1 ? ss2 : (row == 2 ? s3 : 0.0f));
    x = fmaf(fast_exp2(s), tx, x);
    s = s + ts;
}
#undef SIGMA_SCAN_STEP_ASM

// ---- multiplicative scans (scan_bwd2) ---------------------------------------------------------
// Same affine composition, but the decay travels as the product p itself: a step is
//     x += x[src] * p ;  p *= p[src]
// as v_fmac_f32_dpp + v_mul_f32_dpp WITHOUT bound_ctrl -- a lane whose DPP source does not exist is
// simply not written, which is the identity of both updates -- so a step has no transcendental
// (v_exp_f32 issues at a quarter of the plain rate: tools/ubench/valu_ubench.hip, profiles/r02_valu_ubench.txt).
// Hazard: a VALU write followed by a DPP read of the same VGPR needs 2 wait states; inside a step
// the other instruction of the pair provides one, the s_nop 0 the second.
#define SIGMA_MSTEP(CTRL)                                          \
    "v_fmac_f32_dpp %1, %1, %0 " CTRL "\n\t"                       \
    "v_mul_f32_dpp %0, %0, %0 " CTRL "\n\t"                        \
    "s_nop 0\n\t"
#define SIGMA_MSTEP_LAST(CTRL)                                     \
    "v_fmac_f32_dpp %1, %1, %0 " CTRL "\n\t"

// inclusive scan over the 64 lanes, lane 0 earliest.  On return x is the scanned state; p is
// scratch (its last update is skipped: nobody needs the full decay product).
__device__ __forceinline__ void wave_mscan_inclusive(float& p, float& x) {
    asm volatile(
        "s_nop 1\n\t"
        SIGMA_MSTEP("row_shr:1 row_mask:0xf bank_mask:0xf")
        SIGMA_MSTEP("row_shr:2 row_mask:0xf bank_mask:0xf")
        SIGMA_MSTEP("row_shr:4 row_mask:0xf bank_mask:0xf")
        SIGMA_MSTEP("row_shr:8 row_mask:0xf bank_mask:0xf")
        SIGMA_MSTEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
        SIGMA_MSTEP_LAST("row_bcast:31 row_mask:0xc bank_mask:0xf")
        "s_nop 1\n\t"
        : "+v"(p), "+v"(x));
}

// inclusive SUFFIX scan: lane 63 is the earliest element of the scan order; lane i ends up with
// the composition of lanes 63 .. i.  Rows by row_shl; then rows 0 and 2 absorb the head of the
// row to their right (wave_shl:1 parks that head in lane 15 of the row, row_newbcast:15 hands it
// to the whole row), then rows 0 and 1 absorb lane 32 (= rows 2-3 composed) through a readlane.
__device__ __forceinline__ void wave_mscan_inclusive_rev(float& p, float& x) {
    float te, tp;
    asm volatile(
        "s_nop 1\n\t"
        SIGMA_MSTEP("row_shl:1 row_mask:0xf bank_mask:0xf")
        SIGMA_MSTEP("row_shl:2 row_mask:0xf bank_mask:0xf")
        SIGMA_MSTEP("row_shl:4 row_mask:0xf bank_mask:0xf")
        SIGMA_MSTEP("row_shl:8 row_mask:0xf bank_mask:0xf")
        "s_nop 0\n\t"
        "v_mov_b32_dpp %2, %1 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %3, %0 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %1, %2, %0 row_newbcast:15 row_mask:0x5 bank_mask:0xf\n\t"
        "v_mul_f32_dpp %0, %3, %0 row_newbcast:15 row_mask:0x5 bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        : "+v"(p), "+v"(x), "=&v"(te), "=&v"(tp));
    const float xr = lane_bcast(x, 32);                  // rows 2-3 composed (uniform)
    float xv = xr;
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,1,2,3] row_mask:0x3 bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        : "+v"(x) : "v"(p), "v"(xv));
}
#undef SIGMA_MSTEP
#undef SIGMA_MSTEP_LAST

// plain sum over the wave, result in every lane
__device__ __forceinline__ float wave_sum(float v) {
#define SIGMA_ADD(CTRL, MASK) v += dpp_take<CTRL, MASK>(0.0f, v);
    SIGMA_ADD(DPP_ROW_SHR1, 0xF)
    SIGMA_ADD(DPP_ROW_SHR2, 0xF)
    SIGMA_ADD(DPP_ROW_SHR4, 0xF)
    SIGMA_ADD(DPP_ROW_SHR8, 0xF)
    SIGMA_ADD(DPP_ROW_BCAST15, 0xA)
    SIGMA_ADD(DPP_ROW_BCAST31, 0xC)
#undef SIGMA_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// XCD-aware, bijective remap of the hardware workgroup id: consecutive LOGICAL
// ids (which share a (batch, group) B/C tile) land on the same XCD's L2.
// Hardware places workgroup w on XCD w % 8 (speed only, never correctness).
__device__ __forceinline__ int xcd_logical_block(int hw, int nblk) {
    const int q = nblk >> 3, rem = nblk & 7;
    const int xcd = hw & 7, slot = hw >> 3;
    const int base = (xcd < rem) ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    return base + slot;
}

// workgroup barrier that orders LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier.  Unlike
// __syncthreads() it does not wait for outstanding global loads / LDS-DMA (vmcnt), so a
// global_load_lds stream issued earlier keeps flying across it.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- B/C staging with global_load_lds ----------------------------------------------------
// The LDS image of one staging block is [arr = B, C][rows][TILE] floats (rows = NB states x W
// tiles), filled in units of 64 chunks of 16 B: wave v issues units v, v + nwaves, ... and lane i
// of a unit owns chunk ci = unit*64 + i.  Which (array, state, tile, offset) a chunk is does not
// change from block to block, so the decomposition (integer divisions by run-time values) is
// done ONCE per kernel into a few registers; issuing a block is then ~10 VALU per 16-byte load.
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kStageMaxIt = 2;     // units of 64 chunks a wave issues per array and block (host-checked)

template <int T, bool REV>
struct StagePlan {
    static constexpr int TILE = 64 * T;
    static constexpr int CPR = TILE / 4;
    int moff[kStageMaxIt];     // memory index of the chunk for tile 0
    int nn[kStageMaxIt];       // state index inside the block; 0xffff = unused entry
    int nit;                   // entries in use (wave-uniform, <= kStageMaxIt: checked on the host)

    __device__ __forceinline__ void init(int NB, int W, int L) {
        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int nwaves = blockDim.x >> 6;
        const int total = NB * W * CPR;                // chunks of ONE array (B and C share the pattern)
        const int units = (total + 63) / 64;
        nit = (units + nwaves - 1) / nwaves;
#pragma unroll
        for (int i = 0; i < kStageMaxIt; ++i) {
            const int ci = (wave + i * nwaves) * 64 + lane;
            const int row = ci / CPR;                  // nn * W + w
            const int c4 = (ci - row * CPR) * 4;
            const int n = row / W;
            const int w = row - n * W;
            moff[i] = REV ? (L - w * TILE - TILE + c4) : (w * TILE + c4);
            nn[i] = (ci < total) ? n : 0xffff;
        }
    }

    // states [n0, n0 + nbn) of the W tiles starting at tile0 -> dst ([arr][NB][W][TILE] image);
    // Bg / Cg are wave-uniform (batch, group) bases, offsets inside one slice fit 31 bits (host-checked)
    __device__ __forceinline__ void issue(float* dst, const float* Bg, const float* Cg, int B_ns, int C_ns, int n0,
                                          int nbn, int tile0, int L, int arr_stride, bool with_c) const {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int nwaves = blockDim.x >> 6;
        const int t0e = REV ? -(tile0 * TILE) : tile0 * TILE;
#pragma unroll
        for (int i = 0; i < kStageMaxIt; ++i) {
            if (i < nit) {
                const int m = moff[i] + t0e;
                const bool ok = nn[i] < nbn && m >= 0 && m < L;           // L % 4 == 0: whole chunk in range
                float* d = dst + (wave + i * nwaves) * 256;
                if (ok) {
                    const unsigned ob = (unsigned)((n0 + nn[i]) * B_ns + m) * 4u;
                    __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(Bg) + ob), (lptr_t)d, 16, 0, 0);
                    if (with_c) {
                        const unsigned oc = (unsigned)((n0 + nn[i]) * C_ns + m) * 4u;
                        __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(Cg) + oc),
                                                         (lptr_t)(d + arr_stride), 16, 0, 0);
                    }
                }
            }
        }
    }

    // Same, but issued through inline asm so that hipcc does not count the loads: with the builtin it
    // drains them (s_waitcnt vmcnt(0)) before the first VMEM-dependent instruction after the issue,
    // i.e. at the top of the block they were meant to overlap with.  The CALLER owns completion:
    // lds_dma_wait() before the barrier that publishes the buffer (cdna_hip_programming.md 5.7).
    // Untracked loads can only make the compiler's own vmcnt(N) waits stricter, never too weak.
    __device__ __forceinline__ void issue_async(float* dst, const float* Bg, const float* Cg, int B_ns, int C_ns, int n0,
                                                int nbn, int tile0, int L, int arr_stride) const {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int nwaves = blockDim.x >> 6;
        const int t0e = REV ? -(tile0 * TILE) : tile0 * TILE;
#pragma unroll
        for (int i = 0; i < kStageMaxIt; ++i) {
            if (i < nit) {
                const int m = moff[i] + t0e;
                const bool ok = nn[i] < nbn && m >= 0 && m < L;           // L % 4 == 0: whole chunk in range
                const unsigned ldsB = (unsigned)(uintptr_t)(lptr_t)(dst + (wave + i * nwaves) * 256);
                const unsigned ldsC = ldsB + (unsigned)arr_stride * 4u;
                if (ok) {
                    const char* gb = reinterpret_cast<const char*>(Bg) + (unsigned)((n0 + nn[i]) * B_ns + m) * 4u;
                    const char* gc = reinterpret_cast<const char*>(Cg) + (unsigned)((n0 + nn[i]) * C_ns + m) * 4u;
                    unsigned keep;
                    asm volatile(
                        "s_mov_b32 %0, m0\n\t"
                        "s_mov_b32 m0, %3\n\t"
                        "s_nop 0\n\t"
                        "global_load_lds_dwordx4 %1, off\n\t"
                        "s_mov_b32 m0, %4\n\t"
                        "s_nop 0\n\t"
                        "global_load_lds_dwordx4 %2, off\n\t"
                        "s_mov_b32 m0, %0"
                        : "=&s"(keep) : "v"(gb), "v"(gc), "s"(ldsB), "s"(ldsC) : "memory");
                }
            }
        }
    }
};

// completion of issue_async() loads of THIS wave; a workgroup barrier must follow before other waves read
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// parameter row of channel row r of group g (FwdArgs::pswap): CrossScan direction k = j + 2i is kernel group
// g = 2j + i, so with pswap the A / D / bias rows of group g are those of group ((g & 1) << 1) | (g >> 1)
__device__ __forceinline__ int param_row(int r, int g, int rows_per_group, int pswap) {
    return pswap ? r + ((((g & 1) << 1) | (g >> 1)) - g) * rows_per_group : r;
}

// ------------------------------------------------------------------ kernel args
constexpr int kCkptPitch = 1280;   // default elements between state checkpoints in x (see include/sigma_scan.h)
constexpr int kCkptPitchFine = 640; // fine pitch: one checkpoint per 640-tile, no forward sweep in bwd

struct FwdArgs {
    const void* u; const void* delta; const float* A; const void* B; const void* C;
    const float* D; const float* bias; void* out; float* x;
    int batch, dim, L, N, G, n_chunks, rows_per_group, softplus, vec_ok, rowblocks;
    int R;                // rows per workgroup
    int W;                // consecutive tiles per workgroup ("super-tile"); waves = R * W
    int NB;               // states staged per step
    unsigned rev_mask;    // bit g set: group g runs over the sequence in reverse memory order
    int u_gshift;         // u rows of group g are those of group (g >> u_gshift): directions share copies of x
    int pswap;            // 1: the per-row parameters (A, D, delta_bias and their gradients) of group g live in the
                          // rows of group swap(g) (middle two of four groups exchanged): reference direction order
    int ckpt_pitch;       // elements between checkpoints (1280 or 640)
    long x_rs;            // floats per row of x: checkpoint j of row (b, r) at x[(b*dim + r)*x_rs + j*N + n]
    long u_bs, u_ds, dt_bs, dt_ds, A_ds, A_ns;
    long B_bs, B_gs, B_ns, C_bs, C_gs, C_ns, o_bs, o_ds;
    // row-lane forward (scan_fwdr.hip), few rows: the sequence in `segs` segments of `seg_tiles` 16-position tiles;
    // fsumm = [(segs-1)][batch * dim/64][N][64 lanes][2] forward summaries (decay product, end state from zero)
    int segs, seg_tiles;
    float* fsumm;
};

struct BwdArgs {
    FwdArgs f;
    const void* dout; void* du; void* ddelta;
    float* dA; float* dB; float* dC; float* dD; float* dbias;
    float* ws_dB; float* ws_dC;     // [P][batch][G][N][L] per-workgroup partials (P > 1)
    int P;                          // workgroups per (batch, group) = rows_per_group / R
    int out_vec_ok;                 // dB/dC rows are 16-byte aligned
    int g_gshift;                   // dout rows of group g are those of group (g >> g_gshift)
    int slab2;                      // two dB/dC slab sets (by state parity): one barrier per state instead of two
    int flags;                      // bit 0: no L2 warm-up touches (scan_bwd2; cleared by option "bwd_touch")
    int RB;                         // scan_bwd2: row blocks (of R rows) a workgroup walks per tile; P = rows_per_group / (R * RB)
    int S;                          // scan_bwd4: sequence segments (1 = whole sequence per workgroup)
    int seg_tiles;                  // scan_bwd4: 160-tiles per segment
    float* summ;                    // scan_bwd4, S > 1: [(S-1)][batch][dim][N][2] reverse summaries (decay product, e) of segments 1..S-1
    // row-lane backward, chained walk (scan_bwdr.hip): workgroup w owns the tiles [w * chain_W, (w + 1) * chain_W) of the
    // row blocks laid end to end; a row block cut between two workgroups hands its reverse carry over through
    // chain_carry [row blocks][N][64] behind chain_flag [row blocks] (zeroed by the launcher); 0 = one piece per workgroup
    int chain_W;
    float* chain_carry;
    int* chain_flag;
    long g_bs, g_ds, du_bs, du_ds, dd_bs, dd_ds, dA_ds, dA_ns;
    long dB_bs, dB_gs, dB_ns, dC_bs, dC_gs, dC_ns;
};

}  // namespace sigma
