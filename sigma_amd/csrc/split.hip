// split.hip -- fp32 -> (bf16 hi, bf16 lo) operand splitting for the split-operand MFMA GEMMs (gfx950).
//
// The linears of the model (in_proj / out_proj / PatchMerging / fusion and decoder linears; reference:
// nn.Linear calls of models/encoders/vmamba.py, e.g. :1067-1089) are bound by the fp32 MFMA rate (100-134 of
// 157 TFLOP/s, profiles/r02_step_profile_q160.txt).  bf16 MFMA is 16x faster, but plain bf16 operands miss the
// 1e-3 logit bar by 7x (profiles/r02_gemm_precision.jsonl).  With x = hi + lo (hi = bf16(x), lo = bf16(x - hi):
// 16 significant bits) a product is a_hi*b_hi + a_hi*b_lo + a_lo*b_hi (error 4e-6 rms, 5x the fp32 GEMM's own,
// profiles/r02_split_gemm_probe.jsonl), and the three terms become ONE bf16 GEMM with fp32 accumulation when the
// operands are concatenated along the reduction dimension: [a_hi | a_hi | a_lo] x [b_hi ; b_lo ; b_hi].
// This kernel writes those concatenated images in one pass (4 B read, 6 B written per element).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sigma_ops.h"

namespace sigma {
namespace {

__device__ __forceinline__ uint16_t bf16_rne(float v) {
    uint32_t u = __builtin_bit_cast(uint32_t, v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40u);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return static_cast<uint16_t>(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __builtin_bit_cast(float, static_cast<uint32_t>(b) << 16); }

// row r of src (rows x cols, row stride src_rs) -> hi at dst + r*dst_rs [+ hi2_off], lo at dst + r*dst_rs + lo_off
__global__ void __launch_bounds__(256)
split_bf16_kernel(const float* __restrict__ src, long rows, long cols, long src_rs, uint16_t* __restrict__ dst, long dst_rs,
                  long hi2_off, long lo_off, int vec) {
    const long stride = (long)gridDim.x * blockDim.x;
    if (vec) {
        const long c4 = cols >> 2;
        const long total = rows * c4;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const long r = i / c4;
            const long c = (i - r * c4) << 2;
            const float4 v = *reinterpret_cast<const float4*>(src + r * src_rs + c);
            const float f[4] = {v.x, v.y, v.z, v.w};
            uint16_t h[4], l[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { h[k] = bf16_rne(f[k]); l[k] = bf16_rne(f[k] - bf16_to_f32(h[k])); }
            const uint2 hv = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
            const uint2 lv = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
            uint16_t* __restrict__ d = dst + r * dst_rs + c;
            *reinterpret_cast<uint2*>(d) = hv;
            if (hi2_off >= 0) *reinterpret_cast<uint2*>(d + hi2_off) = hv;
            *reinterpret_cast<uint2*>(d + lo_off) = lv;
        }
    } else {
        const long total = rows * cols;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const long r = i / cols;
            const long c = i - r * cols;
            const float f = src[r * src_rs + c];
            const uint16_t h = bf16_rne(f);
            const uint16_t l = bf16_rne(f - bf16_to_f32(h));
            uint16_t* __restrict__ d = dst + r * dst_rs + c;
            d[0] = h;
            if (hi2_off >= 0) d[hi2_off] = h;
            d[lo_off] = l;
        }
    }
}

}  // namespace
}  // namespace sigma

extern "C" int sigma_split_bf16(const float* src, int64_t rows, int64_t cols, int64_t src_row_stride, void* dst, int64_t dst_row_stride,
                                int64_t hi2_offset, int64_t lo_offset, void* stream) {
    if (!src || !dst || rows < 0 || cols < 0 || src_row_stride < cols || lo_offset < 0) return SIGMA_OPS_ERR_ARG;
    if (rows == 0 || cols == 0) return SIGMA_OPS_OK;
    const bool al = (reinterpret_cast<uintptr_t>(src) % 16 == 0) && (reinterpret_cast<uintptr_t>(dst) % 8 == 0);
    const bool vec = al && cols % 4 == 0 && src_row_stride % 4 == 0 && dst_row_stride % 4 == 0 && lo_offset % 4 == 0 &&
                     (hi2_offset < 0 || hi2_offset % 4 == 0);
    const long work = vec ? rows * (cols / 4) : rows * cols;
    long blocks = (work + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(sigma::split_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src, (long)rows,
                       (long)cols, (long)src_row_stride, static_cast<uint16_t*>(dst), (long)dst_row_stride, (long)hi2_offset, (long)lo_offset,
                       vec ? 1 : 0);
    return hipGetLastError() == hipSuccess ? SIGMA_OPS_OK : SIGMA_OPS_ERR_LAUNCH;
}
