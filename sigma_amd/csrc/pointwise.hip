// pointwise.hip -- the squeeze/excite gate of the decoder's conv branch and the segmentation loss as HBM streams.
//
// Reference: ChannelAttention (models/encoders/vmamba.py:1725-1741): y = x * sigmoid(fc(avgpool(x)) + fc(maxpool(x)))
// on (B, C, H, W) activations, 12 calls per step in the Mamba decoder (:1744-1757, :1800-1805); and the loss of
// models/builder.py:146-166, nn.CrossEntropyLoss(reduction='mean', ignore_index) on the (B, classes, H, W) logits.
// ATen runs the gate as two plane reductions + a broadcast multiply and mirrors it in backward with ~10 elementwise /
// reduce launches over the activation (AmaxBackward0, MeanBackward1, MulBackward0: 4.5 ms per step,
// profiles/r03_aten_tail_by_node.txt); the loss as a transposing copy of the channels-last logits + log_softmax + nll
// and their three backward kernels (2.3 ms per step on 8 x 40 x 480 x 640).  Here:
//
//   plane_pool      one pass over x: mean, max and the number of elements equal to the max of every (b, c) plane
//   plane_scale     y = x * s[plane]
//   plane_dot       sum over the plane of g * x                       (gradient of the gate's pre-sigmoid input)
//   plane_gate_bwd  dx = g * s + dmean / HW + (x == max ? dmax / count : 0)   (amax backward: ties share the gradient)
//   softmax_ce_fwd  per pixel (a row of `classes` contiguous logits): log-sum-exp (kept for the backward) and
//                   lse - logit[label]; per-workgroup partial (sum, count) in a fixed layout -> deterministic mean
//   softmax_ce_bwd  dlogit = (exp(logit - lse) - [class == label]) * scale, zero rows for ignored pixels
//
// All of them are bound by HBM: bytes per element 4 (pool), 8 (scale, dot), 12 (gate_bwd), 4 / 8 (loss fwd / bwd).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/sigma_ops.h"
#include "scan_device.h"

namespace sigma {
namespace {

constexpr float kNegInf = -INFINITY;

__device__ __forceinline__ float wave_max_all(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__device__ __forceinline__ void max_count(float v, float& m, float& c) {
    if (v > m) { m = v; c = 1.0f; }
    else if (v == m) c += 1.0f;
}

// block-wide sum of `s` (result valid in thread 0)
__device__ __forceinline__ float block_sum(float s, float* sh) {
    s = wave_sum(s);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = s;
    __syncthreads();
    float t = 0.0f;
    if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    __syncthreads();
    return t;
}

__global__ void __launch_bounds__(256)
plane_pool_kernel(const float* __restrict__ x, long hw, int vec, float* __restrict__ mean, float* __restrict__ mx, float* __restrict__ cnt) {
    __shared__ float sh[12];
    const long plane = blockIdx.x;
    const float* __restrict__ xp = x + plane * hw;
    float s = 0.0f, m = kNegInf, c = 0.0f;
    if (vec) {
        const long n4 = hw >> 2;
        for (long i = threadIdx.x; i < n4; i += blockDim.x) {
            const float4 v = reinterpret_cast<const float4*>(xp)[i];
            s += (v.x + v.y) + (v.z + v.w);
            max_count(v.x, m, c); max_count(v.y, m, c); max_count(v.z, m, c); max_count(v.w, m, c);
        }
    } else {
        for (long i = threadIdx.x; i < hw; i += blockDim.x) { const float v = xp[i]; s += v; max_count(v, m, c); }
    }
    const float wm = wave_max_all(m);
    c = wave_sum(m == wm ? c : 0.0f);
    s = wave_sum(s);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[w] = s; sh[4 + w] = wm; sh[8 + w] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ts = 0.0f, tm = kNegInf, tc = 0.0f;
        for (int i = 0; i < 4; ++i) {
            ts += sh[i];
            if (sh[4 + i] > tm) { tm = sh[4 + i]; tc = sh[8 + i]; }
            else if (sh[4 + i] == tm) tc += sh[8 + i];
        }
        mean[plane] = ts / (float)hw;
        mx[plane] = tm;
        cnt[plane] = tc;
    }
}

__global__ void __launch_bounds__(256)
plane_dot_kernel(const float* __restrict__ a, const float* __restrict__ b, long hw, int vec, float* __restrict__ out) {
    __shared__ float sh[4];
    const long plane = blockIdx.x;
    const float* __restrict__ ap = a + plane * hw;
    const float* __restrict__ bp = b + plane * hw;
    float s = 0.0f;
    if (vec) {
        const long n4 = hw >> 2;
        for (long i = threadIdx.x; i < n4; i += blockDim.x) {
            const float4 u = reinterpret_cast<const float4*>(ap)[i];
            const float4 v = reinterpret_cast<const float4*>(bp)[i];
            s += (u.x * v.x + u.y * v.y) + (u.z * v.z + u.w * v.w);
        }
    } else {
        for (long i = threadIdx.x; i < hw; i += blockDim.x) s += ap[i] * bp[i];
    }
    const float t = block_sum(s, sh);
    if (threadIdx.x == 0) out[plane] = t;
}

__global__ void __launch_bounds__(256)
plane_scale_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ out, long planes, long hw, int vec) {
    const long stride = (long)gridDim.x * blockDim.x;
    if (vec) {
        const long n4 = hw >> 2, total = planes * n4;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const float f = s[i / n4];
            float4 v = reinterpret_cast<const float4*>(x)[i];
            v.x *= f; v.y *= f; v.z *= f; v.w *= f;
            reinterpret_cast<float4*>(out)[i] = v;
        }
    } else {
        const long total = planes * hw;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) out[i] = x[i] * s[i / hw];
    }
}

struct GateBwdArgs {
    const float* g; const float* x; const float* s; const float* dmean; const float* dmax; const float* mx; const float* cnt;
    float* dx; long planes, hw; int vec;
};

__global__ void __launch_bounds__(256) plane_gate_bwd_kernel(const GateBwdArgs a) {
    const long stride = (long)gridDim.x * blockDim.x;
    const float inv = 1.0f / (float)a.hw;
    if (a.vec) {
        const long n4 = a.hw >> 2, total = a.planes * n4;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const long p = i / n4;
            const float f = a.s[p], dm = a.dmean[p] * inv, m = a.mx[p], dx_ = a.dmax[p] / a.cnt[p];
            const float4 g = reinterpret_cast<const float4*>(a.g)[i];
            const float4 v = reinterpret_cast<const float4*>(a.x)[i];
            float4 o;
            o.x = g.x * f + dm + (v.x == m ? dx_ : 0.0f);
            o.y = g.y * f + dm + (v.y == m ? dx_ : 0.0f);
            o.z = g.z * f + dm + (v.z == m ? dx_ : 0.0f);
            o.w = g.w * f + dm + (v.w == m ? dx_ : 0.0f);
            reinterpret_cast<float4*>(a.dx)[i] = o;
        }
    } else {
        const long total = a.planes * a.hw;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const long p = i / a.hw;
            a.dx[i] = a.g[i] * a.s[p] + a.dmean[p] * inv + (a.x[i] == a.mx[p] ? a.dmax[p] / a.cnt[p] : 0.0f);
        }
    }
}

// ---- softmax cross entropy over rows of `nc` contiguous logits (nc % 4 == 0), one thread per row
__global__ void __launch_bounds__(256)
softmax_ce_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, long rows, int nc, long ignore,
                      float* __restrict__ lse, float* __restrict__ partial) {
    __shared__ float sh[4];
    float loss = 0.0f, count = 0.0f;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
        const float* __restrict__ xr = logits + r * nc;
        float m = kNegInf, s = 0.0f;                   // running max and sum of exp(x - m)
        for (int c = 0; c < nc; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            const float vm = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
            if (vm > m) { s *= __expf(m - vm); m = vm; }
            s += (__expf(v.x - m) + __expf(v.y - m)) + (__expf(v.z - m) + __expf(v.w - m));
        }
        const float l = m + __logf(s);
        lse[r] = l;
        const long y = labels[r];
        if (y != ignore && y >= 0 && y < nc) { loss += l - xr[y]; count += 1.0f; }
    }
    const float tl = block_sum(loss, sh);
    const float tc = block_sum(count, sh);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = tl; partial[2 * blockIdx.x + 1] = tc; }
}

__global__ void __launch_bounds__(256)
softmax_ce_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, const float* __restrict__ lse,
                      const float* __restrict__ scale, long rows, int nc, long ignore, float* __restrict__ dlogits) {
    const float sc = scale[0];
    const long stride = (long)gridDim.x * blockDim.x;
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
        const float* __restrict__ xr = logits + r * nc;
        float* __restrict__ dr = dlogits + r * nc;
        const long y = labels[r];
        const bool on = y != ignore && y >= 0 && y < nc;
        const float l = lse[r];
        const float f = on ? sc : 0.0f;
        for (int c = 0; c < nc; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            float4 o;
            o.x = (__expf(v.x - l) - (y == c ? 1.0f : 0.0f)) * f;
            o.y = (__expf(v.y - l) - (y == c + 1 ? 1.0f : 0.0f)) * f;
            o.z = (__expf(v.z - l) - (y == c + 2 ? 1.0f : 0.0f)) * f;
            o.w = (__expf(v.w - l) - (y == c + 3 ? 1.0f : 0.0f)) * f;
            *reinterpret_cast<float4*>(dr + c) = o;
        }
    }
}

// Same kernels with the row (NC4 float4) held in registers: every load of a row is issued before the first use.  The
// generic kernels above walk the row with one load in flight per thread (1.2 TB/s on 8 x 480 x 640 x 40).
template <int NC4>
__global__ void __launch_bounds__(256)
softmax_ce_fwd_reg_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, long rows, long ignore,
                          float* __restrict__ lse, float* __restrict__ partial) {
    __shared__ float sh[4];
    constexpr int nc = NC4 * 4;
    float loss = 0.0f, count = 0.0f;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
        const float4* __restrict__ xr = reinterpret_cast<const float4*>(logits + r * nc);
        float4 v[NC4];
#pragma unroll
        for (int c = 0; c < NC4; ++c) v[c] = xr[c];
        const long y = labels[r];
        float m = kNegInf;
#pragma unroll
        for (int c = 0; c < NC4; ++c) m = fmaxf(m, fmaxf(fmaxf(v[c].x, v[c].y), fmaxf(v[c].z, v[c].w)));
        float s = 0.0f, xy = 0.0f;
#pragma unroll
        for (int c = 0; c < NC4; ++c) {
            s += (__expf(v[c].x - m) + __expf(v[c].y - m)) + (__expf(v[c].z - m) + __expf(v[c].w - m));
            xy = (y == 4 * c) ? v[c].x : (y == 4 * c + 1) ? v[c].y : (y == 4 * c + 2) ? v[c].z : (y == 4 * c + 3) ? v[c].w : xy;
        }
        const float l = m + __logf(s);
        lse[r] = l;
        if (y != ignore && y >= 0 && y < nc) { loss += l - xy; count += 1.0f; }
    }
    const float tl = block_sum(loss, sh);
    const float tc = block_sum(count, sh);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = tl; partial[2 * blockIdx.x + 1] = tc; }
}

template <int NC4>
__global__ void __launch_bounds__(256)
softmax_ce_bwd_reg_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, const float* __restrict__ lse,
                          const float* __restrict__ scale, long rows, long ignore, float* __restrict__ dlogits) {
    constexpr int nc = NC4 * 4;
    const float sc = scale[0];
    const long stride = (long)gridDim.x * blockDim.x;
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
        const float4* __restrict__ xr = reinterpret_cast<const float4*>(logits + r * nc);
        float4* __restrict__ dr = reinterpret_cast<float4*>(dlogits + r * nc);
        float4 v[NC4];
#pragma unroll
        for (int c = 0; c < NC4; ++c) v[c] = xr[c];
        const long y = labels[r];
        const float l = lse[r];
        const float f = (y != ignore && y >= 0 && y < nc) ? sc : 0.0f;
#pragma unroll
        for (int c = 0; c < NC4; ++c) {
            float4 o;
            o.x = (__expf(v[c].x - l) - (y == 4 * c ? 1.0f : 0.0f)) * f;
            o.y = (__expf(v[c].y - l) - (y == 4 * c + 1 ? 1.0f : 0.0f)) * f;
            o.z = (__expf(v[c].z - l) - (y == 4 * c + 2 ? 1.0f : 0.0f)) * f;
            o.w = (__expf(v[c].w - l) - (y == 4 * c + 3 ? 1.0f : 0.0f)) * f;
            dr[c] = o;
        }
    }
}

// NC4 = classes / 4 held in registers up to 64 classes
template <typename F>
bool dispatch_nc4(int nc4, F&& f) {
    switch (nc4) {
#define SIGMA_NC4(N) case N: f(std::integral_constant<int, N>()); return true;
        SIGMA_NC4(1) SIGMA_NC4(2) SIGMA_NC4(3) SIGMA_NC4(4) SIGMA_NC4(5) SIGMA_NC4(6) SIGMA_NC4(7) SIGMA_NC4(8)
        SIGMA_NC4(9) SIGMA_NC4(10) SIGMA_NC4(11) SIGMA_NC4(12) SIGMA_NC4(13) SIGMA_NC4(14) SIGMA_NC4(15) SIGMA_NC4(16)
#undef SIGMA_NC4
        default: return false;
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- backward of  y = a + x * s  with a per-channel s on channels-last rows (CVSSDecoderBlock, vmamba.py:1800-1805) ------
// dx = dy * s and ds += sum over rows of dy * x, one pass over dy and x.  A thread keeps ONE 16-byte column chunk for the
// whole kernel (chunk = tid % (C / 4), row slot = tid / (C / 4); 256 / (C / 4) rows per block iteration), so its part of
// ds lives in four registers; the row slots of a block meet in LDS and one thread per chunk adds the block's sum to ds
// (float atomics: grid x C / 4 of them).  C % 4 == 0, C <= 1024.
__global__ void __launch_bounds__(256) colscale_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ s, float* __restrict__ dx,
                                                           float* __restrict__ ds, long rows, int C) {
    extern __shared__ float part[];                       // [slots][C]
    const int chunks = C >> 2;
    const int slots = 256 / chunks;
    const int slot = threadIdx.x / chunks, ch = threadIdx.x - slot * chunks;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (slot < slots) {
        const float4 sv = *reinterpret_cast<const float4*>(s + 4 * ch);
        for (long r = (long)blockIdx.x * slots + slot; r < rows; r += (long)gridDim.x * slots) {
            const float4 g = *reinterpret_cast<const float4*>(dy + r * C + 4 * ch);
            const float4 xv = *reinterpret_cast<const float4*>(x + r * C + 4 * ch);
            *reinterpret_cast<float4*>(dx + r * C + 4 * ch) = make_float4(g.x * sv.x, g.y * sv.y, g.z * sv.z, g.w * sv.w);
            acc.x = fmaf(g.x, xv.x, acc.x); acc.y = fmaf(g.y, xv.y, acc.y); acc.z = fmaf(g.z, xv.z, acc.z); acc.w = fmaf(g.w, xv.w, acc.w);
        }
        *reinterpret_cast<float4*>(part + slot * C + 4 * ch) = acc;
    }
    __syncthreads();
    if (slot == 0) {
        float4 t = acc;
        for (int k = 1; k < slots; ++k) {
            const float4 o = *reinterpret_cast<const float4*>(part + k * C + 4 * ch);
            t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        atomicAdd(ds + 4 * ch, t.x); atomicAdd(ds + 4 * ch + 1, t.y); atomicAdd(ds + 4 * ch + 2, t.z); atomicAdd(ds + 4 * ch + 3, t.w);
    }
}

unsigned stream_grid(long work_items) {
    long b = (work_items + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

int done() { return hipGetLastError() == hipSuccess ? SIGMA_OPS_OK : SIGMA_OPS_ERR_LAUNCH; }

}  // namespace
}  // namespace sigma

extern "C" {

int sigma_plane_pool(const float* x, int64_t planes, int64_t hw, float* mean, float* max, float* count, void* stream) {
    if (planes < 0 || hw <= 0 || planes > 2147483647L) return SIGMA_OPS_ERR_ARG;
    if (planes == 0) return SIGMA_OPS_OK;
    if (!x || !mean || !max || !count) return SIGMA_OPS_ERR_ARG;
    const int vec = (hw % 4 == 0 && sigma::al16(x)) ? 1 : 0;
    hipLaunchKernelGGL(sigma::plane_pool_kernel, dim3((unsigned)planes), dim3(256), 0, static_cast<hipStream_t>(stream), x, (long)hw, vec,
                       mean, max, count);
    return sigma::done();
}

int sigma_plane_dot(const float* a, const float* b, float* out, int64_t planes, int64_t hw, void* stream) {
    if (planes < 0 || hw <= 0 || planes > 2147483647L) return SIGMA_OPS_ERR_ARG;
    if (planes == 0) return SIGMA_OPS_OK;
    if (!a || !b || !out) return SIGMA_OPS_ERR_ARG;
    const int vec = (hw % 4 == 0 && sigma::al16(a) && sigma::al16(b)) ? 1 : 0;
    hipLaunchKernelGGL(sigma::plane_dot_kernel, dim3((unsigned)planes), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, (long)hw, vec, out);
    return sigma::done();
}

int sigma_plane_scale(const float* x, const float* scale, float* out, int64_t planes, int64_t hw, void* stream) {
    if (planes < 0 || hw <= 0) return SIGMA_OPS_ERR_ARG;
    if (planes == 0) return SIGMA_OPS_OK;
    if (!x || !scale || !out) return SIGMA_OPS_ERR_ARG;
    const int vec = (hw % 4 == 0 && sigma::al16(x) && sigma::al16(out)) ? 1 : 0;
    const long work = vec ? planes * (hw / 4) : planes * hw;
    hipLaunchKernelGGL(sigma::plane_scale_kernel, dim3(sigma::stream_grid(work)), dim3(256), 0, static_cast<hipStream_t>(stream), x, scale, out,
                       (long)planes, (long)hw, vec);
    return sigma::done();
}

int sigma_colscale_bwd(const float* dy, const float* x, const float* scale, float* dx, float* dscale, int64_t rows, int32_t channels,
                       void* stream) {
    if (rows < 0 || channels <= 0 || channels % 4 != 0 || channels > 1024) return SIGMA_OPS_ERR_ARG;
    if (rows == 0) return SIGMA_OPS_OK;
    if (!dy || !x || !scale || !dx || !dscale) return SIGMA_OPS_ERR_ARG;
    if (!sigma::al16(dy) || !sigma::al16(x) || !sigma::al16(scale) || !sigma::al16(dx)) return SIGMA_OPS_ERR_ARG;
    const int slots = 256 / (channels / 4);
    long grid = (rows + slots - 1) / slots;
    if (grid > 512) grid = 512;                       // two workgroups per CU: every block ends with C atomics on the same C addresses
    hipLaunchKernelGGL(sigma::colscale_bwd_kernel, dim3((unsigned)grid), dim3(256), (size_t)slots * channels * sizeof(float),
                       static_cast<hipStream_t>(stream), dy, x, scale, dx, dscale, (long)rows, (int)channels);
    return sigma::done();
}

int sigma_plane_gate_bwd(const sigma_gate_bwd_params* p, void* stream) {
    if (!p || p->planes < 0 || p->hw <= 0) return SIGMA_OPS_ERR_ARG;
    if (p->planes == 0) return SIGMA_OPS_OK;
    if (!p->g || !p->x || !p->scale || !p->dmean || !p->dmax || !p->max || !p->count || !p->dx) return SIGMA_OPS_ERR_ARG;
    sigma::GateBwdArgs a{p->g, p->x, p->scale, p->dmean, p->dmax, p->max, p->count, p->dx, (long)p->planes, (long)p->hw, 0};
    a.vec = (p->hw % 4 == 0 && sigma::al16(p->g) && sigma::al16(p->x) && sigma::al16(p->dx)) ? 1 : 0;
    const long work = a.vec ? a.planes * (a.hw / 4) : a.planes * a.hw;
    hipLaunchKernelGGL(sigma::plane_gate_bwd_kernel, dim3(sigma::stream_grid(work)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return sigma::done();
}

int sigma_softmax_ce_fwd(const float* logits, const int64_t* labels, int64_t rows, int32_t classes, int64_t ignore_index, float* lse,
                         float* partial, void* stream) {
    if (rows < 0 || classes <= 0 || classes % 4 != 0) return SIGMA_OPS_ERR_ARG;
    if (!partial) return SIGMA_OPS_ERR_ARG;
    if (rows > 0 && (!logits || !labels || !lse || !sigma::al16(logits))) return SIGMA_OPS_ERR_ARG;
    // every one of the SIGMA_CE_BLOCKS workgroups writes its (sum, count) pair, rows or not: the caller adds them up
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool reg = sigma::dispatch_nc4(classes / 4, [&](auto n) {
        hipLaunchKernelGGL(sigma::softmax_ce_fwd_reg_kernel<decltype(n)::value>, dim3(SIGMA_CE_BLOCKS), dim3(256), 0, s, logits, labels,
                           (long)rows, (long)ignore_index, lse, partial);
    });
    if (!reg)
        hipLaunchKernelGGL(sigma::softmax_ce_fwd_kernel, dim3(SIGMA_CE_BLOCKS), dim3(256), 0, s, logits, labels, (long)rows, (int)classes,
                           (long)ignore_index, lse, partial);
    return sigma::done();
}

int sigma_softmax_ce_bwd(const float* logits, const int64_t* labels, const float* lse, const float* scale, int64_t rows, int32_t classes,
                         int64_t ignore_index, float* dlogits, void* stream) {
    if (rows < 0 || classes <= 0 || classes % 4 != 0) return SIGMA_OPS_ERR_ARG;
    if (rows == 0) return SIGMA_OPS_OK;
    if (!logits || !labels || !lse || !scale || !dlogits || !sigma::al16(logits) || !sigma::al16(dlogits)) return SIGMA_OPS_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool reg = sigma::dispatch_nc4(classes / 4, [&](auto n) {
        hipLaunchKernelGGL(sigma::softmax_ce_bwd_reg_kernel<decltype(n)::value>, dim3(sigma::stream_grid(rows)), dim3(256), 0, s, logits,
                           labels, lse, scale, (long)rows, (long)ignore_index, dlogits);
    });
    if (!reg)
        hipLaunchKernelGGL(sigma::softmax_ce_bwd_kernel, dim3(sigma::stream_grid(rows)), dim3(256), 0, s, logits, labels, lse, scale,
                           (long)rows, (int)classes, (long)ignore_index, dlogits);
    return sigma::done();
}

}  // extern "C"
