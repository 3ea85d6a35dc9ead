// hbm_mix_ubench.hip -- what HBM sustains on MI355X for the READ : WRITE mixes of the scan kernels, with plain streaming
// kernels (float4 per lane, grid-stride, every CU busy): the ceiling the `roofline.frac` of bench.py should be read against.
//   copy  1 read : 1 write        triad 2 reads : 1 write (scan forward without checkpoints)
//   fwd   2 reads : 2 writes      (u, delta -> out, checkpoints at one per 16 positions and 16 states)
//   bwd   3 reads : 2 writes      (u, delta, dout -> du, ddelta)         read  reads only
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/hbm_mix_ubench tools/ubench/hbm_mix_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NR, int NW>
__global__ void __launch_bounds__(256) k_mix(const float4* __restrict__ a, float4* __restrict__ o, long n4, long stride4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < NR; ++r) { const float4 v = a[i + r * stride4]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
#pragma unroll
        for (int w = 0; w < NW; ++w) { float4 t = acc; t.x += w; o[i + w * stride4] = t; }
        if (NW == 0 && acc.x == 1.2345e-30f) o[i] = acc;
    }
}

template <int NR, int NW>
void run(const char* name, float4* a, float4* o, long n4, long stride4) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    hipLaunchKernelGGL((k_mix<NR, NW>), dim3(grid), dim3(256), 0, 0, a, o, n4, stride4);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k_mix<NR, NW>), dim3(grid), dim3(256), 0, 0, a, o, n4, stride4);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 5.0 * (NR + NW) * n4 * 16.0;
    printf("{\"mix\": \"%s\", \"reads\": %d, \"writes\": %d, \"MB_per_stream\": %.0f, \"TBps\": %.3f}\n", name, NR, NW, n4 * 16.0 / 1e6, bytes / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    CHECK(hipSetDevice(0));
    const long n4 = (236L << 20) / 16;            // 236 MB per stream = one (16,3072,1200) fp32 activation
    float4 *a, *o;
    CHECK(hipMalloc(&a, 3 * n4 * 16)); CHECK(hipMalloc(&o, 3 * n4 * 16));
    CHECK(hipMemset(a, 0, 3 * n4 * 16));
    run<1, 0>("read", a, o, n4, n4);
    run<1, 1>("copy", a, o, n4, n4);
    run<2, 1>("triad", a, o, n4, n4);
    run<2, 2>("fwd_2r2w", a, o, n4, n4);
    run<3, 2>("bwd_3r2w", a, o, n4, n4);
    run<0, 1>("write", a, o, n4, n4);
    return 0;
}
