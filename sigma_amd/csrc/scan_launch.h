// scan_launch.h -- host-side launch helpers shared by the kernel TUs and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include "scan_device.h"

namespace sigma {

inline size_t fwd_lds_bytes(int T, int nwaves, int N) {
    const int pad = (T >= 8) ? 4 : 0;
    const size_t row = (size_t)kWave * (T + pad);
    return sizeof(float) * (2 * kStateBlock * row + 2 * (size_t)nwaves * N);
}

// backward: B, C tiles (padded, lane-blocked) + per-wave dB/dC term slabs ([wave][2][TILE])
// + per-wave scratch: tile-start states for every tile of a 2048 chunk, reverse carry, dA partials
inline size_t bwd_lds_bytes(int T, int nwaves, int N) {
    const int pad = (T >= 8) ? 4 : 0;
    const size_t row = (size_t)kWave * (T + pad);
    const int tiles_per_chunk = 2048 / (kWave * T);
    return sizeof(float) * (2 * kStateBlock * row + 2 * (size_t)nwaves * kWave * T +
                            (size_t)nwaves * N * (tiles_per_chunk + 3));
}

hipError_t launch_scan_fwd(const FwdArgs& a, int dtype, int T, int nwaves, hipStream_t stream);
hipError_t launch_scan_bwd(const BwdArgs& a, int dtype, int T, int nwaves, hipStream_t stream);
hipError_t launch_selftest(float* out, hipStream_t stream);

}  // namespace sigma
