// selftest.hip -- on-device check of the wave64 DPP scan primitives in scan_device.h
// against serial loops over LDS.  One wave, a few hundred instructions.
#include "scan_device.h"
#include "scan_launch.h"

namespace sigma {

__global__ void __launch_bounds__(64) selftest_kernel(float* out) {
    __shared__ float sp[64], sx[64];
    const int lane = threadIdx.x;
    // deterministic, lane-asymmetric data: p in (0.5, 1], x in [-1, 1]
    const float p0 = 0.5f + 0.5f * (float)((lane * 37 + 11) % 64) / 64.0f;
    const float x0 = (float)((lane * 53 + 7) % 64) / 32.0f - 1.0f;
    sp[lane] = p0; sx[lane] = x0;
    __syncthreads();

    // forward inclusive
    float p = __builtin_amdgcn_logf(p0), x = x0;        // the scans carry log2 of the decay
    wave_scan_inclusive(p, x);
    p = fast_exp2(p);
    float rp = 1.0f, rx = 0.0f;
    for (int i = 0; i <= lane; ++i) { rx = fmaf(sp[i], rx, sx[i]); rp *= sp[i]; }
    const float e_fwd = fmaxf(fabsf(p - rp) / fabsf(rp), fabsf(x - rx));

    // reverse inclusive (suffix)
    float q = __builtin_amdgcn_logf(p0), y = x0;
    wave_scan_inclusive_rev(q, y);
    q = fast_exp2(q);
    float sq = 1.0f, sy = 0.0f;
    for (int i = 63; i >= lane; --i) { sy = fmaf(sp[i], sy, sx[i]); sq *= sp[i]; }
    const float e_rev = fmaxf(fabsf(q - sq) / fabsf(sq), fabsf(y - sy));

    const float pv = wave_prev_lane(x0, -7.0f);
    const float e_prev = fabsf(pv - (lane == 0 ? -7.0f : sx[lane - 1]));
    const float nx = wave_next_lane(x0, 9.0f);
    const float e_next = fabsf(nx - (lane == 63 ? 9.0f : sx[lane + 1]));

    const float s = wave_sum(x0);
    float ss = 0.0f;
    for (int i = 0; i < 64; ++i) ss += sx[i];
    const float e_sum = fabsf(s - ss);

    // multiplicative scans of scan_bwd2 (decay carried as the product itself)
    float mp = p0, mx = x0;
    wave_mscan_inclusive(mp, mx);
    const float e_mfwd = fabsf(mx - rx);
    float mq = p0, my = x0;
    wave_mscan_inclusive_rev(mq, my);
    const float e_mrev = fabsf(my - sy);

    // max over lanes through LDS
    __shared__ float red[7][64];
    red[0][lane] = e_fwd; red[1][lane] = e_rev; red[2][lane] = e_prev; red[3][lane] = e_next; red[4][lane] = e_sum;
    red[5][lane] = e_mfwd; red[6][lane] = e_mrev;
    __syncthreads();
    if (lane == 0) {
        float m[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < 7; ++k) for (int i = 0; i < 64; ++i) m[k] = fmaxf(m[k], red[k][i]);
        const bool bad = m[0] > 2e-5f || m[1] > 2e-5f || m[2] != 0.0f || m[3] != 0.0f || m[4] > 1e-4f ||
                         m[5] > 2e-5f || m[6] > 2e-5f;
        out[0] = bad ? 1.0f : 0.0f;
        for (int k = 0; k < 7; ++k) out[1 + k] = m[k];
    }
}

hipError_t launch_selftest(float* out, hipStream_t stream) {
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, stream, out);
    return hipGetLastError();
}

}  // namespace sigma
