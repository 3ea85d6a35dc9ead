// upsample.hip -- bilinear x2 up-sampling of channels-last activations and its adjoint (gfx950).
//
// Reference: F.interpolate(scale_factor=2, mode='bilinear', align_corners=False) in the decoder's UpsampleExpand and
// FinalUpsample_X4 (models/decoders/MambaDecoder.py:33-51, 76-97).  ATen's channels-last kernels take 302 us forward and
// 990 us backward on the (8, 240, 320, 96) -> (8, 480, 640, 96) step (profiles/r03_bench_split3_kernel_stats.txt: 4 ms
// of the training step for 8 launches), 6x / 20x the time of moving the bytes once.  Both directions are written
// here as gathers (the adjoint too: an input pixel collects its <= 4 x 4 output pixels, no atomics), one float4 of
// channels per thread, every access a 16-byte piece of a contiguous channel run.
//
// Index arithmetic = ATen's (upsample_bilinear2d, area_pixel_compute_source_index with align_corners = false):
//   src = max((dst + 0.5) * 0.5 - 0.5, 0),  i0 = floor(src),  i1 = min(i0 + 1, n - 1),  w1 = src - i0,  w0 = 1 - w1.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sigma_ops.h"

namespace sigma {
namespace {

struct Tap { int i0, i1; float w0, w1; };

__device__ __forceinline__ Tap tap_of(int dst, int n) {
    float src = (dst + 0.5f) * 0.5f - 0.5f;
    src = src < 0.0f ? 0.0f : src;
    Tap t;
    t.i0 = (int)src;
    t.i1 = t.i0 + 1 < n ? t.i0 + 1 : n - 1;
    t.w1 = src - (float)t.i0;
    t.w0 = 1.0f - t.w1;
    return t;
}

// weight of input index i in output index o (0 when o does not read i)
__device__ __forceinline__ float weight_of(int o, int i, int n) {
    const Tap t = tap_of(o, n);
    return (t.i0 == i ? t.w0 : 0.0f) + (t.i1 == i ? t.w1 : 0.0f);
}

__global__ void __launch_bounds__(256)
up2x_fwd_kernel(const float4* __restrict__ in, float4* __restrict__ out, int B, int H, int W, int C4) {
    const long total = (long)B * 2 * H * 2 * W * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4);
        long r = idx / C4;
        const int ox = (int)(r % (2 * W)); r /= 2 * W;
        const int oy = (int)(r % (2 * H));
        const int b = (int)(r / (2 * H));
        const Tap ty = tap_of(oy, H), tx = tap_of(ox, W);
        const float4* __restrict__ base = in + (long)b * H * W * C4 + c;
        const float4 a = base[((long)ty.i0 * W + tx.i0) * C4], bq = base[((long)ty.i0 * W + tx.i1) * C4];
        const float4 cq = base[((long)ty.i1 * W + tx.i0) * C4], d = base[((long)ty.i1 * W + tx.i1) * C4];
        float4 o;
        o.x = ty.w0 * (tx.w0 * a.x + tx.w1 * bq.x) + ty.w1 * (tx.w0 * cq.x + tx.w1 * d.x);
        o.y = ty.w0 * (tx.w0 * a.y + tx.w1 * bq.y) + ty.w1 * (tx.w0 * cq.y + tx.w1 * d.y);
        o.z = ty.w0 * (tx.w0 * a.z + tx.w1 * bq.z) + ty.w1 * (tx.w0 * cq.z + tx.w1 * d.z);
        o.w = ty.w0 * (tx.w0 * a.w + tx.w1 * bq.w) + ty.w1 * (tx.w0 * cq.w + tx.w1 * d.w);
        out[idx] = o;
    }
}

// din[b, y, x, :] = sum over the output rows 2y-1 .. 2y+2 and columns 2x-1 .. 2x+2 that exist of wy * wx * g
__global__ void __launch_bounds__(256)
up2x_bwd_kernel(const float4* __restrict__ g, float4* __restrict__ din, int B, int H, int W, int C4) {
    const long total = (long)B * H * W * C4;
    const int OH = 2 * H, OW = 2 * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4);
        long r = idx / C4;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        float wy[4], wx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oy = 2 * y - 1 + k, ox = 2 * x - 1 + k;
            wy[k] = (oy >= 0 && oy < OH) ? weight_of(oy, y, H) : 0.0f;
            wx[k] = (ox >= 0 && ox < OW) ? weight_of(ox, x, W) : 0.0f;
        }
        const float4* __restrict__ base = g + (long)b * OH * OW * C4 + c;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            int oy = 2 * y - 1 + ky;
            oy = oy < 0 ? 0 : (oy >= OH ? OH - 1 : oy);               // clamped address, zero weight
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                int ox = 2 * x - 1 + kx;
                ox = ox < 0 ? 0 : (ox >= OW ? OW - 1 : ox);
                const float w = wy[ky] * wx[kx];
                const float4 v = base[((long)oy * OW + ox) * C4];
                acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
            }
        }
        din[idx] = acc;
    }
}

}  // namespace
}  // namespace sigma

extern "C" int sigma_upsample2x_nhwc(const float* in, float* out, int32_t batch, int32_t height, int32_t width, int32_t channels,
                                     int32_t backward, void* stream) {
    if (!in || !out || batch < 0 || height <= 0 || width <= 0 || channels <= 0 || channels % 4 != 0) return SIGMA_OPS_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) return SIGMA_OPS_ERR_ARG;
    if (batch == 0) return SIGMA_OPS_OK;
    const int C4 = channels / 4;
    const long work = backward ? (long)batch * height * width * C4 : (long)batch * 4 * height * width * C4;
    long blocks = (work + 255) / 256;
    if (blocks > 256L * 64) blocks = 256L * 64;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (backward)
        hipLaunchKernelGGL(sigma::up2x_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(in),
                           reinterpret_cast<float4*>(out), batch, height, width, C4);
    else
        hipLaunchKernelGGL(sigma::up2x_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(in),
                           reinterpret_cast<float4*>(out), batch, height, width, C4);
    return hipGetLastError() == hipSuccess ? SIGMA_OPS_OK : SIGMA_OPS_ERR_LAUNCH;
}
