// scan_quad.h -- device helpers shared by the quad-row kernels (scan_fwd4.hip, scan_bwd4.hip): one DPP row of 16
// lanes per channel row, 10 positions per lane, 160-position tiles.  gfx950 only.
#pragma once
#include "scan_device.h"

namespace sigma {
namespace {

constexpr int kT4 = 10;            // positions per lane
constexpr int kTile4 = 160;        // positions per DPP row and tile
constexpr int kCols4 = 320;        // dB + dC columns of a tile = 5 registers x 64 lanes per wave and state

// LDS-DMA staging of one tile: image [arr = B, C][N][160] floats in memory order, in units of 64 chunks of
// 16 bytes (wave v issues units v, v + nwaves, ...).  Issued once per tile and wave, so the chunk -> (state,
// offset) decomposition is simply recomputed.  Untracked issue (see StagePlan::issue_async in scan_device.h):
// the caller retires it with lds_dma_wait() + a barrier.
template <bool REV>
__device__ __forceinline__ void stage_tile4(float* dst, const float* Bg, const float* Cg, int B_ns, int C_ns, int N,
                                            int tile, int L) {
    constexpr int CPR = kTile4 / 4;                // 16-byte chunks per (state, tile)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int total = N * CPR;
    const int units = (total + 63) / 64;
    const int t0 = REV ? (L - kTile4 - tile * kTile4) : tile * kTile4;     // memory index of image element 0
    for (int unit = wave; unit < units; unit += nwaves) {
        const int ci = unit * 64 + lane;
        const int n = ci / CPR;
        const int m = t0 + (ci - n * CPR) * 4;
        const bool ok = ci < total && m >= 0 && m < L;                     // L % 4 == 0: whole chunk in range
        const unsigned ldsB = (unsigned)(uintptr_t)(lptr_t)(dst + unit * 256);
        const unsigned ldsC = ldsB + (unsigned)(N * kTile4) * 4u;
        if (ok) {
            const char* gb = reinterpret_cast<const char*>(Bg) + (unsigned)(n * B_ns + m) * 4u;
            const char* gc = reinterpret_cast<const char*>(Cg) + (unsigned)(n * C_ns + m) * 4u;
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

// L2 warm-up of the u / delta / dout segments (640 B per row) the wave loads in its NEXT row step: lane
// 6*row + line touches one 128-byte line through LDS-DMA into a dummy area (no VGPR destination; untracked like
// the B/C stream, retired by the next vmcnt wait).  Issued a few states before the end of the current row step,
// so the lines are still in L2 when the real loads arrive: the ~2 us HBM miss that all waves of the workgroup
// would otherwise sit out together at the top of a row step becomes an L2 hit.
__device__ __forceinline__ void touch_line4(const char* pa, bool on, unsigned lds_dummy) {
    if (on) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(pa), "s"(lds_dummy) : "memory");
    }
}

// positions 2q, 2q+1 of the lane's 10 (li = lane inside its DPP row); the image is in memory order
template <bool REV>
__device__ __forceinline__ void lds_read_pair(const float* __restrict__ tile, int li, int q, float (&v)[2]) {
    const float* __restrict__ src = REV ? tile + (15 - li) * kT4 + (kT4 - (q + 1) * 2) : tile + li * kT4 + q * 2;
    const float2 x = *reinterpret_cast<const float2*>(src);
    v[0] = REV ? x.y : x.x;
    v[1] = REV ? x.x : x.y;
}

// store positions 2q, 2q+1 of the lane's 10 (same addressing as store_items<float, 10, REV>)
template <bool REV>
__device__ __forceinline__ void store_pair(float* __restrict__ row, int lbase, int L, bool vec, int q, float v0, float v1) {
    if (vec && lbase + kT4 <= L) {
        float* __restrict__ dst = row + (REV ? (L - lbase - kT4) + (kT4 - 2 - 2 * q) : lbase + 2 * q);
        *reinterpret_cast<float2*>(dst) = REV ? make_float2(v1, v0) : make_float2(v0, v1);
    } else {
        const int p0 = lbase + 2 * q;
        if (p0 < L) row[REV ? (L - 1 - p0) : p0] = v0;
        if (p0 + 1 < L) row[REV ? (L - 2 - p0) : p0 + 1] = v1;
    }
}

// ---- scans inside one DPP row (16 lanes): the first four steps of wave_mscan_inclusive{,_rev}
#define SIGMA_MSTEP(CTRL)                                          \
    "v_fmac_f32_dpp %1, %1, %0 " CTRL "\n\t"                       \
    "v_mul_f32_dpp %0, %0, %0 " CTRL "\n\t"                        \
    "s_nop 0\n\t"
#define SIGMA_MSTEP_LAST(CTRL)                                     \
    "v_fmac_f32_dpp %1, %1, %0 " CTRL "\n\t"
__device__ __forceinline__ void row_mscan_inclusive(float& p, float& x) {          // lane 0 of the row earliest
    asm volatile(
        "s_nop 1\n\t"
        SIGMA_MSTEP("row_shr:1 row_mask:0xf bank_mask:0xf")
        SIGMA_MSTEP("row_shr:2 row_mask:0xf bank_mask:0xf")
        SIGMA_MSTEP("row_shr:4 row_mask:0xf bank_mask:0xf")
        SIGMA_MSTEP_LAST("row_shr:8 row_mask:0xf bank_mask:0xf")
        "s_nop 1\n\t"
        : "+v"(p), "+v"(x));
}
__device__ __forceinline__ void row_mscan_inclusive_rev(float& p, float& x) {      // lane 15 of the row earliest
    asm volatile(
        "s_nop 1\n\t"
        SIGMA_MSTEP("row_shl:1 row_mask:0xf bank_mask:0xf")
        SIGMA_MSTEP("row_shl:2 row_mask:0xf bank_mask:0xf")
        SIGMA_MSTEP("row_shl:4 row_mask:0xf bank_mask:0xf")
        SIGMA_MSTEP_LAST("row_shl:8 row_mask:0xf bank_mask:0xf")
        "s_nop 1\n\t"
        : "+v"(p), "+v"(x));
}
#undef SIGMA_MSTEP
#undef SIGMA_MSTEP_LAST

// sum over the 16 lanes of a DPP row, total in lane 0 of the row (other lanes: partial sums)
__device__ __forceinline__ float row_sum_to_lane0(float v) {
    v += dpp_take<DPP_ROW_SHL1, 0xF>(0.0f, v);
    v += dpp_take<DPP_ROW_SHL2, 0xF>(0.0f, v);
    v += dpp_take<DPP_ROW_SHL4, 0xF>(0.0f, v);
    v += dpp_take<DPP_ROW_SHL8, 0xF>(0.0f, v);
    return v;
}
// lane i of a row <- lane i+1 (lane 15 <- lane 0): row_ror:15
__device__ __forceinline__ float row_rotate_left(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x12F, 0xF, 0xF, false));
}
// every lane <- lane (16*row + n) of its own DPP row (n wave-uniform)
__device__ __forceinline__ float row_pick(float v, int addr4) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr4, __builtin_bit_cast(int, v)));
}

// a + b with the halves / rows regrouped (tools/ubench/lane_ops_probe.hip):
//   swap32: lanes 0-31 of the result = a[0:32] + a[32:64], lanes 32-63 = b[0:32] + b[32:64]
//   swap16: DPP rows of the result = {a.r0 + a.r1, b.r0 + b.r1, a.r2 + a.r3, b.r2 + b.r3}
__device__ __forceinline__ float fold32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float fold16(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

}  // namespace
}  // namespace sigma
