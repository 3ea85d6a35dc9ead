// scan_rowlane.h -- device helpers of the "row-lane" selective-scan kernels (scan_fwdr.hip, scan_bwdr.hip).  gfx950 only.
//
// Mapping (round 4): a LANE is a channel row.  A workgroup owns one block of 64 consecutive rows of one (batch, group)
// and has four waves; wave sw walks the states [sw * NS, (sw + 1) * NS) of all 64 rows (NS = dstate / 4).  The sequence
// is walked in tiles of 16 positions in scan order; inside a tile a lane runs the recurrence of its row serially.
//   * no cross-lane scan at all (the quad-row kernels: two 4-step DPP row scans + a fold/replay double pass per state);
//   * B and C of a (state, tile) are the same for the 64 rows of a wave: they are fetched with SCALAR loads
//     (s_load_dwordx16) and enter the VALU instructions as SGPR operands -- no LDS image, no ds_read per element;
//   * u / delta / dout are loaded coalesced (four lanes x 16 bytes per row), pre-processed once per element by the
//     thread that loaded them (softplus, delta * u) and handed to the row lanes through a 12 KB LDS block whose rows
//     are rotated by 4 * chunk so that both access patterns are bank-conflict free;
//   * the sums over the states (out; du, ddelta) of the four waves meet in LDS once per tile (two barriers per tile);
//   * the state entering a tile comes from a checkpoint per tile (pitch 16) in a layout private to these kernels:
//         x[(((b * dim/64 + rowblock) * ntiles + tile) * N + n) * 64 + lane]
//     (256 contiguous bytes per (tile, state) for the wave that owns the row block, in both kernels).
// Tiles are taken in MEMORY order (tile m = elements [16m, 16m + 16)); a reversed group walks them from the last to the
// first and the positions of a tile from 15 down to 0, so a partial last tile is always the memory-last one.
#pragma once
#include "scan_device.h"

// phase timing of development builds (-DSIGMA_RL_PROF=1; tools/rowlane_prof.py): cycles per phase, summed over the waves
#ifndef SIGMA_RL_PROF
#define SIGMA_RL_PROF 0
#endif
#if SIGMA_RL_PROF
#define RLPROF_DECL long long prof_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long prof_last = __builtin_readcyclecounter();
#define RLPROF(i) { const long long t_ = __builtin_readcyclecounter(); prof_t[i] += t_ - prof_last; prof_last = t_; }
#define RLPROF_FLUSH(arr) if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&arr[i_], (unsigned long long)prof_t[i_]); atomicAdd(&arr[15], 1ull); }
#else
#define RLPROF_DECL
#define RLPROF(i)
#define RLPROF_FLUSH(arr)
#endif

// Ablation builds (-DSIGMA_RL_ABL=<bits>; results are WRONG, timing only; tools/gpu_r4.sh):
//   1 B / C from constants (no scalar loads)   2 no checkpoint stores / loads   4 u / delta / dout from constants
//   8 no workgroup barriers   16 no exchange through LDS   32 no softplus (prologue arithmetic)   64 no dB/dC reduce network
#ifndef SIGMA_RL_ABL
#define SIGMA_RL_ABL 0
#endif

// Round 6: 1 = ONE checkpoint per tile -- the state after the tile, a block of N * 64 floats per (row block, tile) (rl_store_ck below) -- so
// the forward writes half as many checkpoint bytes (it is bound by HBM traffic: 0.5 of its 1.3 GB per launch were
// checkpoints) and the backward reads half as many, walking the tile WHOLE: per state a forward replay of its 16 positions
// from the state entering the tile, then the reverse recurrence over the 16 positions.  Same arithmetic per element-state as
// the half-tile walk (every position is replayed exactly once either way), 16-entry per-lane arrays (~250 VGPRs: the two
// waves per SIMD this kernel runs with have them), B / C of a (state, tile) in ONE 16-dword scalar request each.
// 0: the round-4 scheme (two checkpoints per tile, the backward in halves of 8 positions).
#ifndef SIGMA_BWDR_FULL
#define SIGMA_BWDR_FULL 1
#endif

namespace sigma {
namespace {

constexpr int kRT = 16;            // positions per tile = checkpoint pitch of the row-lane kernels
constexpr int kRRows = 64;         // rows per workgroup (one per lane)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f_a16 __attribute__((ext_vector_type(4), aligned(16)));
typedef float v16f_a16 __attribute__((ext_vector_type(16), aligned(16)));
typedef const __attribute__((address_space(4))) v4f_a16* cv4p_t;      // constant address space: scalar loads
typedef const __attribute__((address_space(4))) v16f_a16* cv16p_t;

// float4 unit of element (chunk c, row r) inside a [4][64] block: rows rotated by 4c.  Conflict-free for the
// (row = tid / 4, chunk = tid % 4) threads as well as for the (row = lane, chunk fixed) waves (ds_*_b128 lane groups).
__device__ __forceinline__ int rl_unit(int c, int r) { return c * 64 + ((r + 4 * c) & 63); }

// B or C of one (state, tile): 16 consecutive floats at a wave-uniform address -> SGPRs.  nch = valid 16-byte chunks
// (4 except in a partial last tile; L % 4 == 0 is a precondition of the kernels), missing ones read as zero.
__device__ __forceinline__ void rl_load_bc(const float* base, int nch, float (&v)[kRT]) {
#if SIGMA_RL_ABL & 1
    float c = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((int)(reinterpret_cast<uintptr_t>(base) & 0xffff) | 0x3f000000));
#pragma unroll
    for (int k = 0; k < kRT; ++k) { v[k] = c; asm volatile("" : "+s"(c)); }
    return;
#endif
    if (nch >= 4) {
        const v16f t = *reinterpret_cast<cv16p_t>(reinterpret_cast<uintptr_t>(base));
#pragma unroll
        for (int k = 0; k < kRT; ++k) v[k] = t[k];
    } else {
        const v4f z = {0.0f, 0.0f, 0.0f, 0.0f};
        const v4f c0 = *reinterpret_cast<cv4p_t>(reinterpret_cast<uintptr_t>(base));
        const v4f c1 = nch > 1 ? *reinterpret_cast<cv4p_t>(reinterpret_cast<uintptr_t>(base + 4)) : z;
        const v4f c2 = nch > 2 ? *reinterpret_cast<cv4p_t>(reinterpret_cast<uintptr_t>(base + 8)) : z;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = c0[j]; v[4 + j] = c1[j]; v[8 + j] = c2[j]; v[12 + j] = 0.0f; }
    }
}

__device__ __forceinline__ v4f rl_load4(const float* __restrict__ p, bool ok) {
    const v4f z = {0.0f, 0.0f, 0.0f, 0.0f};
#if SIGMA_RL_ABL & 4
    const float c = 0.001f * (float)(reinterpret_cast<uintptr_t>(p) & 0xff);
    const v4f t = {c, 0.5f * c, 0.25f * c, -c};
    return ok ? t : z;
#endif
    return ok ? *reinterpret_cast<const v4f*>(p) : z;
}

// 16 bytes per lane from global memory straight into LDS (no VGPR destination, so the compiler can neither delay the
// request to its use nor spill its target): lane i of the wave lands at lds_wave_base + 16 * i.  Untracked by the
// compiler's vmcnt bookkeeping (which can only become stricter by it): the issuing wave retires it with rl_dma_wait()
// before it reads the bytes back.
__device__ __forceinline__ void rl_dma16(const float* gptr, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(lds_wave_base) : "memory");
}
// the same with a wave-uniform 64-bit base (SGPR pair) + a 32-bit byte offset per lane: no 64-bit address registers
__device__ __forceinline__ void rl_dma16s(const void* sbase, unsigned voff_bytes, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff_bytes), "s"(sbase), "s"(lds_wave_base) : "memory");
}
__device__ __forceinline__ void rl_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// ... or, when at least YOUNGER vector-memory operations were issued after the requests (they retire in order), with
// those left in flight -- so that the wait does not sit out the stores issued just before it
template <int YOUNGER>
__device__ __forceinline__ void rl_dma_wait_keep() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(YOUNGER) : "memory"); }

// Checkpoints of one tile (SIGMA_BWDR_FULL): a block of N * 64 floats in which the states go in groups of G = N / 4 (the
// states of one wave of the backward, which always runs four state waves) and a group is innermost -- slot of state n of
// lane l = ((n / G) * 64 + l) * G + n % G -- so that the states of a wave are ONE 16- / 8- / 4-byte access per lane and
// 1 KB / 512 B / 256 B contiguous per wave (a vector-memory instruction costs the issuing SIMD ~25 ns whatever its
// width: tools/ubench/stateloop_ubench.hip).  n0 = first state of the wave (a multiple of NS); NS in {1, 2, 4}, NS <= G.
template <int NS, int G>
__device__ __forceinline__ int rl_ck_slot(int n0, int lane) { return ((n0 / G) * 64 + lane) * G + (n0 % G); }
template <int NS, int G>
__device__ __forceinline__ void rl_store_ck(float* __restrict__ tile_block, int n0, int lane, const float (&x)[NS]) {
    static_assert((NS == 1 || NS == 2 || NS == 4) && NS <= G && G % NS == 0, "states per wave");
    float* p = tile_block + rl_ck_slot<NS, G>(n0, lane);
    if constexpr (NS == 4) { const v4f v = {x[0], x[1], x[2], x[3]}; *reinterpret_cast<v4f*>(p) = v; }
    else if constexpr (NS == 2) { *reinterpret_cast<float2*>(p) = make_float2(x[0], x[1]); }
    else { p[0] = x[0]; }
}
template <int NS, int G>
__device__ __forceinline__ void rl_load_ck(const float* __restrict__ tile_block, int n0, int lane, float (&x)[NS]) {
    static_assert((NS == 1 || NS == 2 || NS == 4) && NS <= G && G % NS == 0, "states per wave");
    const float* p = tile_block + rl_ck_slot<NS, G>(n0, lane);
    if constexpr (NS == 4) { const v4f v = *reinterpret_cast<const v4f*>(p); x[0] = v[0]; x[1] = v[1]; x[2] = v[2]; x[3] = v[3]; }
    else if constexpr (NS == 2) { const float2 v = *reinterpret_cast<const float2*>(p); x[0] = v.x; x[1] = v.y; }
    else { x[0] = p[0]; }
}

__device__ __forceinline__ void rl_barrier() {
#if !(SIGMA_RL_ABL & 8)
    lds_barrier();
#endif
}

}  // namespace
}  // namespace sigma
