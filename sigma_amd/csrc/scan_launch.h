// scan_launch.h -- host-side launch helpers shared by the kernel TUs and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include "scan_device.h"

// rows (waves) per workgroup the backward kernel may use with T = 10 (640-element tiles):
// 12 = ~164 VGPRs, 3 waves per SIMD; 16 needs the kernel to fit 128 VGPRs (build knob for A/B runs)
#ifndef SIGMA_BWD_MAXW_T10
#define SIGMA_BWD_MAXW_T10 12
#endif

namespace sigma {

constexpr int kBwdMaxWavesT10 = SIGMA_BWD_MAXW_T10;

// forward: double-buffered B/C stage [2][2][NB][W][TILE] + tile aggregates + A, running state
inline size_t fwd_lds_bytes(int T, int R, int W, int NB, int N) {
    const size_t tile = (size_t)kWave * T;
    return sizeof(float) * (2 * 2 * (size_t)NB * W * tile + 2 * 2 * (size_t)R * W + 3 * (size_t)R * N);
}

// backward: double-buffered B/C stage [2][2][NB][TILE] + per-wave dB/dC term slabs [R][2][TILE]
// + per-wave scratch: tile-start states of one checkpoint span, reverse carry, dA partials, A
inline size_t bwd_lds_bytes(int T, int R, int NB, int N, bool slab2 = false) {
    const size_t tile = (size_t)kWave * T;
    const int tps = kCkptPitch / (kWave * T);
    return sizeof(float) * (2 * 2 * (size_t)NB * tile + (slab2 ? 2 : 1) * 2 * (size_t)R * tile + (size_t)R * N * (tps + 3));
}

// scan_bwd2: double-buffered B/C stage + nslab slab sets + reverse carries of the chunk's RB*R rows
inline size_t bwd2_lds_bytes(int T, int R, int NB, int N, bool slab2, int RB) {
    const size_t tile = (size_t)kWave * T;
    return sizeof(float) * (2 * 2 * (size_t)NB * tile + (slab2 ? 2 : 1) * 2 * (size_t)R * tile + (size_t)RB * R * N + 64 +   // + touch sink
                            (RB > 1 ? 2 * (size_t)N * tile : 0));                                                            // + dB/dC accumulators
}

// scan_bwd3 (state-parallel): B/C of one 320-tile for all states + row partials (two parity sets, reused at
// the tile end for the dB/dC slot sums) + reverse carries of the chunk's rows; nw = waves per workgroup
inline size_t bwd3_lds_bytes(int nw, int N, int RB) {
    const size_t tile = 320;
    const int Q = N / 4 > 0 ? N / 4 : 1;
    return sizeof(float) * (2 * (size_t)N * tile + 4 * (size_t)nw * tile + (size_t)RB * (nw / Q) * N);
}

// scan_bwd4 (quad-row): nbuf B/C images of one 160-tile for all states [nbuf][2][N][160] + 2*SB slab sets [W][320]
// + reverse carries [RB*4*W][N] + touch sink + the dB/dC accumulators [N][320] when RB > 1
inline size_t bwd4_lds_bytes(int W, int N, int SB, int RB, int nbuf = 2) {
    return sizeof(float) * ((size_t)nbuf * 2 * (size_t)N * 160 + 2 * (size_t)SB * W * 320 + (size_t)RB * 4 * W * N + 64 + (RB > 1 ? (size_t)N * 320 : 0));
}

// scan_fwd4 (quad-row forward): two B/C images of one 160-tile for all states [2][2][N][160]
inline size_t fwd4_lds_bytes(int N) { return sizeof(float) * (2 * 2 * (size_t)N * 160); }

// row-lane kernels (scan_fwdr.hip / scan_bwdr.hip): [arrays][4 chunks][64 rows] float4 of pre-processed operands +
// [state waves][...] partial sums over the states
// round 6 (SIGMA_FWDR_PIPE): four state waves run the pipelined body on double-buffered blocks (48 KB: three workgroups per CU)
#ifndef SIGMA_FWDR_PIPE
#define SIGMA_FWDR_PIPE 1
#endif
// tiles the (row, chunk) loads of u / delta run ahead of the state loop (1 or 2)
#ifndef SIGMA_FWDR_PF
#define SIGMA_FWDR_PF 1
#endif
inline size_t fwdr_lds_bytes(int NW) { return SIGMA_FWDR_PIPE && NW == 4 ? 16 * (size_t)(4 * 256 + 2 * 4 * 256) : 16 * (size_t)(2 * 256 + NW * 256); }
#if defined(SIGMA_RL_ABL) && (SIGMA_RL_ABL & 128)
inline size_t bwdr_lds_bytes(int NW) { return 16 * (size_t)(5 * 256 + NW * 256 + 3 * 256); }        // timing probe: half the exchange area
#else
inline size_t bwdr_lds_bytes(int NW) { return 16 * (size_t)(5 * 256 + NW * 2 * 256 + 3 * 256); }
#endif

constexpr int kMaxDevices = 16;    // per-device cache of the raised dynamic-LDS cap (hipFuncSetAttribute is per device)

hipError_t launch_scan_bwd2(const BwdArgs& a, int dtype, int T, bool glds, hipStream_t stream);
hipError_t launch_scan_bwd3(const BwdArgs& a, int dtype, bool glds, hipStream_t stream);   // a.f.R = waves per workgroup
hipError_t launch_scan_fwd4(const FwdArgs& a, hipStream_t stream);   // a.R = waves (4 rows each), a.rowblocks = workgroups per (batch, group)
hipError_t launch_scan_bwd4(const BwdArgs& a, hipStream_t stream);   // a.f.R = waves (4 rows each), a.slab2 = states per barrier
hipError_t launch_scan_fwdr(const FwdArgs& a, hipStream_t stream);   // a.rowblocks = 64-row blocks per (batch, group); a.segs / a.seg_tiles / a.fsumm
hipError_t launch_scan_bwdr(const BwdArgs& a, hipStream_t stream);
int bwdr_resident_per_cu(int N);                                     // workgroups of scan_bwdr_kernel a CU holds (occupancy query, cached)   // a.P = 64-row blocks per (batch, group); a.S / a.seg_tiles / a.summ
hipError_t launch_reduce_partials(const BwdArgs& a, hipStream_t stream);
hipError_t bwd4_prof_read(unsigned long long* out16);
hipError_t fwdr_prof_read(unsigned long long* out16);     // development builds (SIGMA_RL_PROF), zeros otherwise
hipError_t bwdr_prof_read(unsigned long long* out16);
hipError_t bwdr_chain_timeouts_read(unsigned int* out);   // chained walk: hand-over waits that ran out since the last call
hipError_t gemm_prof_read(unsigned long long* out16);     // development builds (SIGMA_GEMM_PROF), zeros otherwise
hipError_t bwd2_prof_read(unsigned long long* out16);     // development builds (SIGMA_BWD2_PROF), zeros otherwise
hipError_t launch_scan_fwd(const FwdArgs& a, int dtype, int T, bool glds, bool prefetch, hipStream_t stream);
hipError_t launch_scan_bwd(const BwdArgs& a, int dtype, int T, bool glds, hipStream_t stream);
hipError_t launch_selftest(float* out, hipStream_t stream);

}  // namespace sigma
