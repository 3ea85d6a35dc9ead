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
constexpr int kStateBlock = 4;   // states whose B/C rows are staged in LDS at a time

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

// 4 consecutive, fully valid, vector-aligned elements (16 B for f32, 8 B for 16-bit types)
template <typename io_t>
__device__ __forceinline__ void load4_vec(const io_t* __restrict__ p, float (&v)[4]) {
    if constexpr (sizeof(io_t) == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        io_t e0, e1, e2, e3;
        e0.bits = static_cast<uint16_t>(t.x & 0xffffu); e1.bits = static_cast<uint16_t>(t.x >> 16);
        e2.bits = static_cast<uint16_t>(t.y & 0xffffu); e3.bits = static_cast<uint16_t>(t.y >> 16);
        v[0] = io_traits<io_t>::to_float(e0); v[1] = io_traits<io_t>::to_float(e1);
        v[2] = io_traits<io_t>::to_float(e2); v[3] = io_traits<io_t>::to_float(e3);
    }
}

template <typename io_t>
__device__ __forceinline__ void store4_vec(io_t* __restrict__ p, const float (&v)[4]) {
    if constexpr (sizeof(io_t) == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        const uint32_t lo = static_cast<uint32_t>(io_traits<io_t>::from_float(v[0]).bits) |
                            (static_cast<uint32_t>(io_traits<io_t>::from_float(v[1]).bits) << 16);
        const uint32_t hi = static_cast<uint32_t>(io_traits<io_t>::from_float(v[2]).bits) |
                            (static_cast<uint32_t>(io_traits<io_t>::from_float(v[3]).bits) << 16);
        *reinterpret_cast<uint2*>(p) = make_uint2(lo, hi);
    }
}

// 4 consecutive elements of which the first nvalid (may be <= 0 or > 4) exist; rest read as 0
template <typename io_t>
__device__ __forceinline__ void load4(const io_t* __restrict__ p, bool vec, int nvalid, float (&v)[4]) {
    if (vec && nvalid >= 4) {
        load4_vec<io_t>(p, v);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (i < nvalid) ? io_traits<io_t>::to_float(p[i]) : 0.0f;
    }
}

// lane-blocked row access: lane owns T consecutive elements starting at lbase.  The fast
// path (whole segment in range, rows vector-aligned) is branch-free vector traffic; the
// guarded scalar path only runs in the ragged last tile or for unaligned tensors.
template <typename io_t, int T>
__device__ __forceinline__ void load_items(const io_t* __restrict__ row, int lbase, int L, bool vec, float (&v)[T]) {
    if (vec && lbase + T <= L) {
#pragma unroll
        for (int q = 0; q < T / 4; ++q) {
            float t[4];
            load4_vec<io_t>(row + lbase + 4 * q, t);
            v[4 * q + 0] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
        }
    } else {
#pragma unroll
        for (int k = 0; k < T; ++k) v[k] = (lbase + k < L) ? io_traits<io_t>::to_float(row[lbase + k]) : 0.0f;
    }
}

template <typename io_t, int T>
__device__ __forceinline__ void store_items(io_t* __restrict__ row, int lbase, int L, bool vec, const float (&v)[T]) {
    if (vec && lbase + T <= L) {
#pragma unroll
        for (int q = 0; q < T / 4; ++q) {
            const float t[4] = {v[4 * q + 0], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            store4_vec<io_t>(row + lbase + 4 * q, t);
        }
    } else {
#pragma unroll
        for (int k = 0; k < T; ++k) if (lbase + k < L) row[lbase + k] = io_traits<io_t>::from_float(v[k]);
    }
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
    if (raw > 20.0f) { sig = 1.0f; return raw; }
    const float e = fast_exp2(raw * kLog2e);
    const float w = 1.0f + e;
    const float lw = fast_log2(w) * kLn2;
    const float wm1 = w - 1.0f;
    sig = e * fast_rcp(w);
    return (wm1 == 0.0f) ? e : lw * (e * fast_rcp(wm1));
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

// The scan element is the affine map  s -> p*s + x  ("decay p, injected state x").
// compose(earlier, later) = (p_l*p_e, p_l*x_e + x_l)          (selective_scan_common.h:92-96)
#define SIGMA_SCAN_STEP(CTRL, MASK)                                   \
    {                                                                 \
        const float pe_ = dpp_take<CTRL, MASK>(1.0f, p);              \
        const float xe_ = dpp_take<CTRL, MASK>(0.0f, x);              \
        x = fmaf(p, xe_, x);                                          \
        p = p * pe_;                                                  \
    }

// inclusive scan over the 64 lanes in lane order (lane 0 earliest)
__device__ __forceinline__ void wave_scan_inclusive(float& p, float& x) {
    SIGMA_SCAN_STEP(DPP_ROW_SHR1, 0xF)
    SIGMA_SCAN_STEP(DPP_ROW_SHR2, 0xF)
    SIGMA_SCAN_STEP(DPP_ROW_SHR4, 0xF)
    SIGMA_SCAN_STEP(DPP_ROW_SHR8, 0xF)
    SIGMA_SCAN_STEP(DPP_ROW_BCAST15, 0xA)
    SIGMA_SCAN_STEP(DPP_ROW_BCAST31, 0xC)
}
#undef SIGMA_SCAN_STEP

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
//   (p_i, x_i) <- x_i + p_i * X_{i+1},  p_i * P_{i+1}
// Rows are handled with row_shl; the cross-row part uses readlane of the row
// heads (lanes 16/32/48), all wave-uniform.
__device__ __forceinline__ void wave_scan_inclusive_rev(float& p, float& x) {
#define SIGMA_RSTEP(CTRL)                                             \
    {                                                                 \
        const float pe_ = dpp_take<CTRL, 0xF>(1.0f, p);               \
        const float xe_ = dpp_take<CTRL, 0xF>(0.0f, x);               \
        x = fmaf(p, xe_, x);                                          \
        p = p * pe_;                                                  \
    }
    SIGMA_RSTEP(DPP_ROW_SHL1)
    SIGMA_RSTEP(DPP_ROW_SHL2)
    SIGMA_RSTEP(DPP_ROW_SHL4)
    SIGMA_RSTEP(DPP_ROW_SHL8)
#undef SIGMA_RSTEP
    // row aggregates sit at the first lane of each row (lanes 0,16,32,48)
    const int lane = threadIdx.x & 63;
    const float p3 = lane_bcast(p, 48), x3 = lane_bcast(x, 48);
    const float p2 = lane_bcast(p, 32), x2 = lane_bcast(x, 32);
    const float p1 = lane_bcast(p, 16), x1 = lane_bcast(x, 16);
    // suffix aggregates of whole rows: S3 = row3, S2 = row2 o S3, S1 = row1 o S2
    const float sx3 = x3, sp3 = p3;
    const float sx2 = fmaf(p2, sx3, x2), sp2 = p2 * sp3;
    const float sx1 = fmaf(p1, sx2, x1), sp1 = p1 * sp2;
    const int row = lane >> 4;
    const float tx = row == 0 ? sx1 : (row == 1 ? sx2 : (row == 2 ? sx3 : 0.0f));
    const float tp = row == 0 ? sp1 : (row == 1 ? sp2 : (row == 2 ? sp3 : 1.0f));
    x = fmaf(p, tx, x);
    p = p * tp;
}

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

// ------------------------------------------------------------------ kernel args
struct FwdArgs {
    const void* u; const void* delta; const float* A; const void* B; const void* C;
    const float* D; const float* bias; void* out; float* x;
    int batch, dim, L, N, G, n_chunks, rows_per_group, softplus, vec_ok, rowblocks;
    long u_bs, u_ds, dt_bs, dt_ds, A_ds, A_ns;
    long B_bs, B_gs, B_ns, C_bs, C_gs, C_ns, o_bs, o_ds;
};

struct BwdArgs {
    FwdArgs f;
    const void* dout; void* du; void* ddelta;
    float* dA; float* dB; float* dC; float* dD; float* dbias;
    float* ws_dB; float* ws_dC;     // [P][batch][G][N][L] per-workgroup partials (P > 1)
    int P;                          // workgroups per (batch, group) = rows_per_group / nwaves
    int out_vec_ok;                 // dB/dC rows are 16-byte aligned (float4 stores legal when P == 1)
    long g_bs, g_ds, du_bs, du_ds, dd_bs, dd_ds, dA_ds, dA_ns;
    long dB_bs, dB_gs, dB_ns, dC_bs, dC_gs, dC_ns;
};

template <int T> struct TileGeom {
    static constexpr int PAD = (T >= 8) ? 4 : 0;     // keeps ds_read_b128 conflict-free
    static constexpr int LSTR = T + PAD;             // floats per lane slot
    static constexpr int ROW = kWave * LSTR;         // floats per staged state row
    static constexpr int TILE = kWave * T;           // sequence elements per tile
};

}  // namespace sigma
