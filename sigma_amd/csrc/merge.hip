// merge.hip -- CrossMerge and its adjoint as single passes between the scan's (B, G, d, L) planes and the
// channels-last (B, H, W, d) activations the rest of SS2D works in.
//
// Reference: CrossMerge.forward (models/encoders/vmamba.py:100-108) followed by
//   y = y.transpose(dim0=1, dim1=2).contiguous(); y = out_norm(y).view(B, H, W, -1)     (vmamba.py:221-224)
// and, in backward, CrossMerge.backward = CrossScan (vmamba.py:110-121) applied to the transposed gradient.
// With the flipped directions already written in natural order by the scan kernels (sigma_scan.h,
// rev_group_mask), the merge is
//   y[b, h, w, c] = ys[b,0,c,hW+w] + ys[b,1,c,hW+w] + ys[b,2,c,wH+h] + ys[b,3,c,wH+h]
// and its adjoint writes the gradient once in each memory order:
//   g2[b,0,c,hW+w] = g2[b,1,c,wH+h] = dy[b, h, w, c].
// Both are 3-D transposes (channel <-> position, and h <-> w for the column-major planes).  A workgroup
// owns a 16 x 16 pixel tile x 32 channels staged in LDS, so that every global access is a run of 16
// consecutive positions (64 B) of one plane or 32 consecutive channels (128 B) of one pixel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sigma_ops.h"
#include "scan_device.h"

namespace sigma {

namespace {

constexpr int kP = 16;      // pixels per tile side
constexpr int kC = 32;      // channels per tile

struct MergeArgs {
    const float* ys;   // (B, 4, d, L)      merge input
    float* y;          // (B, H, W, d)      merge output
    const float* dy;   // (B, H, W, d)      split input
    float* g2;         // (B, 2, d, L)      split output
    int B, d, H, W;
};

// LDS tile [c][h][w]: w-rows padded to 17 floats and c-slabs to 273 (both odd), so that walking along w
// (stride 1), along h (stride 17) and along c (stride 273 = 17 mod 32) all touch distinct banks.
constexpr int kSlab = kP * (kP + 1) + 1;
__device__ __forceinline__ int lds_idx(int c, int h, int w) { return c * kSlab + h * (kP + 1) + w; }

__global__ void __launch_bounds__(256) cross_merge_kernel(const MergeArgs a) {
    __shared__ float t[kC * kSlab];
    const int H = a.H, W = a.W, d = a.d;
    const long L = (long)H * W;
    const int tw = (W + kP - 1) / kP, th = (H + kP - 1) / kP, tc = (d + kC - 1) / kC;
    // The tiles of one (batch, channel block) share cache lines (a tile row is a 64-byte run, half a line; on the small
    // planes of the late stages a line even spans image rows).  Hardware deals consecutive workgroup ids round-robin to
    // the 8 XCDs, each with its own L2: neighbouring tiles on different XCDs fetch every shared line twice and cannot
    // merge their half-line writes.  Logical ids that are consecutive on ONE XCD (scan_device.h), pixel tiles fastest.
    int bid = xcd_logical_block((int)blockIdx.x, (int)gridDim.x);
    const int wi = bid % tw; bid /= tw;
    const int hi = bid % th; bid /= th;
    const int ci = bid % tc; bid /= tc;
    const int b = bid;
    const int c0 = ci * kC, w0 = wi * kP, h0 = hi * kP;
    const int tid = threadIdx.x;
    const float* __restrict__ p0 = a.ys + ((long)(b * 4 + 0) * d) * L;
    const float* __restrict__ p1 = a.ys + ((long)(b * 4 + 1) * d) * L;
    const float* __restrict__ p2 = a.ys + ((long)(b * 4 + 2) * d) * L;
    const float* __restrict__ p3 = a.ys + ((long)(b * 4 + 3) * d) * L;
    // Loads in batches of kU elements per thread, ALL issued before the first use (unconditional loads from a clamped
    // address, then a select): with one or two loads in flight per thread the kernel moved 2.4 TB/s -- 16 waves x 2 x 256 B
    // per CU in flight is ~1 TB/s worth of outstanding bytes at HBM latency.
    constexpr int kU = 8;
    // row-major planes: consecutive threads -> consecutive w
    for (int e0 = tid; e0 < kC * kP * kP; e0 += 256 * kU) {
        float a0[kU], a1[kU];
        bool in[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int e = e0 + 256 * u;
            const int wl = e & (kP - 1), hl = (e >> 4) & (kP - 1), cl = e >> 8;
            const int c = c0 + cl, h = h0 + hl, w = w0 + wl;
            in[u] = c < d && h < H && w < W;
            const long o = in[u] ? (long)c * L + (long)h * W + w : 0;
            a0[u] = p0[o]; a1[u] = p1[o];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int e = e0 + 256 * u;
            t[lds_idx(e >> 8, (e >> 4) & (kP - 1), e & (kP - 1))] = in[u] ? a0[u] + a1[u] : 0.0f;
        }
    }
    __syncthreads();
    // column-major planes: consecutive threads -> consecutive h
    for (int e0 = tid; e0 < kC * kP * kP; e0 += 256 * kU) {
        float a2[kU], a3[kU];
        bool in[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int e = e0 + 256 * u;
            const int hl = e & (kP - 1), wl = (e >> 4) & (kP - 1), cl = e >> 8;
            const int c = c0 + cl, h = h0 + hl, w = w0 + wl;
            in[u] = c < d && h < H && w < W;
            const long o = in[u] ? (long)c * L + (long)w * H + h : 0;
            a2[u] = p2[o]; a3[u] = p3[o];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int e = e0 + 256 * u;
            if (in[u]) t[lds_idx(e >> 8, e & (kP - 1), (e >> 4) & (kP - 1))] += a2[u] + a3[u];
        }
    }
    __syncthreads();
    // channels-last output: consecutive threads -> consecutive c
    float* __restrict__ yb = a.y + (long)b * L * d;
    for (int e = tid; e < kC * kP * kP; e += 256) {
        const int cl = e & (kC - 1), wl = (e >> 5) & (kP - 1), hl = e >> 9;
        const int c = c0 + cl, h = h0 + hl, w = w0 + wl;
        if (c < d && h < H && w < W) yb[((long)h * W + w) * d + c] = t[lds_idx(cl, hl, wl)];
    }
}

__global__ void __launch_bounds__(256) cross_split_kernel(const MergeArgs a) {
    __shared__ float t[kC * kSlab];
    const int H = a.H, W = a.W, d = a.d;
    const long L = (long)H * W;
    const int tw = (W + kP - 1) / kP, th = (H + kP - 1) / kP, tc = (d + kC - 1) / kC;
    // The tiles of one (batch, channel block) share cache lines (a tile row is a 64-byte run, half a line; on the small
    // planes of the late stages a line even spans image rows).  Hardware deals consecutive workgroup ids round-robin to
    // the 8 XCDs, each with its own L2: neighbouring tiles on different XCDs fetch every shared line twice and cannot
    // merge their half-line writes.  Logical ids that are consecutive on ONE XCD (scan_device.h), pixel tiles fastest.
    int bid = xcd_logical_block((int)blockIdx.x, (int)gridDim.x);
    const int wi = bid % tw; bid /= tw;
    const int hi = bid % th; bid /= th;
    const int ci = bid % tc; bid /= tc;
    const int b = bid;
    const int c0 = ci * kC, w0 = wi * kP, h0 = hi * kP;
    const int tid = threadIdx.x;
    const float* __restrict__ db = a.dy + (long)b * L * d;
    constexpr int kU = 8;                                  // loads of a batch all in flight together (see cross_merge_kernel)
    for (int e0 = tid; e0 < kC * kP * kP; e0 += 256 * kU) {
        float v[kU];
        bool in[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int e = e0 + 256 * u;
            const int cl = e & (kC - 1), wl = (e >> 5) & (kP - 1), hl = e >> 9;
            const int c = c0 + cl, h = h0 + hl, w = w0 + wl;
            in[u] = c < d && h < H && w < W;
            v[u] = db[in[u] ? ((long)h * W + w) * d + c : 0];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int e = e0 + 256 * u;
            t[lds_idx(e & (kC - 1), e >> 9, (e >> 5) & (kP - 1))] = in[u] ? v[u] : 0.0f;
        }
    }
    __syncthreads();
    float* __restrict__ g_rm = a.g2 + ((long)(b * 2 + 0) * d) * L;
    float* __restrict__ g_cm = a.g2 + ((long)(b * 2 + 1) * d) * L;
    for (int e = tid; e < kC * kP * kP; e += 256) {
        const int wl = e & (kP - 1), hl = (e >> 4) & (kP - 1), cl = e >> 8;
        const int c = c0 + cl, h = h0 + hl, w = w0 + wl;
        if (c < d && h < H && w < W) g_rm[(long)c * L + (long)h * W + w] = t[lds_idx(cl, hl, wl)];
    }
    for (int e = tid; e < kC * kP * kP; e += 256) {
        const int hl = e & (kP - 1), wl = (e >> 4) & (kP - 1), cl = e >> 8;
        const int c = c0 + cl, h = h0 + hl, w = w0 + wl;
        if (c < d && h < H && w < W) g_cm[(long)c * L + (long)w * H + h] = t[lds_idx(cl, hl, wl)];
    }
}

// dst[b][c][r] = src[b][r][c] for r < R, c < C with independent row / batch strides (in floats): the
// channels-last <-> channels-first copies around the depthwise conv (vmamba.py:1070-1071 permute +
// contiguous, and the chunk() whose backward concatenates the two gradient halves).
struct TrArgs { const float* src; float* dst; int R, C; long src_bs, src_rs, dst_bs, dst_rs; int tiles_r, tiles_c; };

__global__ void __launch_bounds__(256) transpose2d_kernel(const TrArgs a) {
    __shared__ float t[32][33];
    int bid = blockIdx.x;
    const int tc = bid % a.tiles_c; bid /= a.tiles_c;
    const int tr = bid % a.tiles_r; bid /= a.tiles_r;
    const int b = bid;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* __restrict__ s = a.src + (long)b * a.src_bs;
    float* __restrict__ d = a.dst + (long)b * a.dst_bs;
    const int r0 = tr * 32, c0 = tc * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        t[ty + 8 * i][tx] = (r < a.R && c < a.C) ? s[(long)r * a.src_rs + c] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < a.R && c < a.C) d[(long)c * a.dst_rs + r] = t[tx][ty + 8 * i];
    }
}

bool grid_for(const sigma_merge_params* p, unsigned* grid) {
    if (!p || p->batch < 0 || p->channels <= 0 || p->height <= 0 || p->width <= 0) return false;
    const long n = (long)p->batch * ((p->height + kP - 1) / kP) * ((p->width + kP - 1) / kP) * ((p->channels + kC - 1) / kC);
    if (n > 2147483647L) return false;
    *grid = (unsigned)n;
    return true;
}

}  // namespace

}  // namespace sigma

namespace sigma {
namespace {
// acc[o][i] += src[2o][i] + src[2o+1][i]; 16 bytes per lane, grid-stride over (o, i / 4)
__global__ void __launch_bounds__(256)
pair_sum_add_kernel(const float* __restrict__ src, float* __restrict__ acc, long n_outer, long inner, int vec) {
    // blockIdx.y walks the outer index (no division per element); four elements per thread and batch with all twelve
    // loads issued before the first add (memory-level parallelism, see cross_merge_kernel)
    constexpr int kU = 4;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long o = blockIdx.y; o < n_outer; o += gridDim.y) {
        if (vec) {
            const long inner4 = inner >> 2;
            const float4* __restrict__ s0 = reinterpret_cast<const float4*>(src + (2 * o) * inner);
            const float4* __restrict__ s1 = reinterpret_cast<const float4*>(src + (2 * o + 1) * inner);
            float4* __restrict__ d = reinterpret_cast<float4*>(acc + o * inner);
            for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < inner4; i0 += stride * kU) {
                float4 a[kU], b[kU], c[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const long i = i0 + u * stride;
                    const long ii = i < inner4 ? i : i0;
                    a[u] = s0[ii]; b[u] = s1[ii]; c[u] = d[ii];
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const long i = i0 + u * stride;
                    if (i < inner4) {
                        c[u].x += a[u].x + b[u].x; c[u].y += a[u].y + b[u].y; c[u].z += a[u].z + b[u].z; c[u].w += a[u].w + b[u].w;
                        d[i] = c[u];
                    }
                }
            }
        } else {
            for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < inner; i += stride)
                acc[o * inner + i] += src[(2 * o) * inner + i] + src[(2 * o + 1) * inner + i];
        }
    }
}
}  // namespace
}  // namespace sigma

extern "C" {

int sigma_pair_sum_add(const float* src, float* acc, int64_t n_outer, int64_t inner, void* stream) {
    if (!src || !acc || n_outer < 0 || inner < 0) return SIGMA_OPS_ERR_ARG;
    if (n_outer == 0 || inner == 0) return SIGMA_OPS_OK;
    const int vec = (inner % 4 == 0) && (reinterpret_cast<uintptr_t>(src) % 16 == 0) && (reinterpret_cast<uintptr_t>(acc) % 16 == 0);
    const long per = vec ? inner / 4 : inner;                        // elements (of 16 or 4 bytes) per outer index
    const long gy = n_outer < 65535 ? n_outer : 65535;
    long gx = (per + 256 * 4 - 1) / (256 * 4);                       // four elements per thread
    const long cap = (256 * 16 + gy - 1) / gy;                       // ~16 workgroups per CU in total
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(sigma::pair_sum_add_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, static_cast<hipStream_t>(stream), src, acc,
                       (long)n_outer, (long)inner, vec);
    return hipGetLastError() == hipSuccess ? SIGMA_OPS_OK : SIGMA_OPS_ERR_LAUNCH;
}

int sigma_cross_merge_nhwc(const sigma_merge_params* p, void* stream) {
    unsigned grid = 0;
    if (!sigma::grid_for(p, &grid)) return SIGMA_OPS_ERR_ARG;
    if (grid == 0) return SIGMA_OPS_OK;
    if (!p->planes4 || !p->nhwc) return SIGMA_OPS_ERR_ARG;
    sigma::MergeArgs a{};
    a.ys = p->planes4; a.y = p->nhwc; a.B = p->batch; a.d = p->channels; a.H = p->height; a.W = p->width;
    hipLaunchKernelGGL(sigma::cross_merge_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SIGMA_OPS_OK : SIGMA_OPS_ERR_LAUNCH;
}

int sigma_cross_split_nhwc(const sigma_merge_params* p, void* stream) {
    unsigned grid = 0;
    if (!sigma::grid_for(p, &grid)) return SIGMA_OPS_ERR_ARG;
    if (grid == 0) return SIGMA_OPS_OK;
    if (!p->planes2 || !p->nhwc) return SIGMA_OPS_ERR_ARG;
    sigma::MergeArgs a{};
    a.dy = p->nhwc; a.g2 = p->planes2; a.B = p->batch; a.d = p->channels; a.H = p->height; a.W = p->width;
    hipLaunchKernelGGL(sigma::cross_split_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SIGMA_OPS_OK : SIGMA_OPS_ERR_LAUNCH;
}

int sigma_transpose2d(const sigma_transpose_params* p, void* stream) {
    if (!p || p->batch < 0 || p->rows <= 0 || p->cols <= 0) return SIGMA_OPS_ERR_ARG;
    if (p->batch == 0) return SIGMA_OPS_OK;
    if (!p->src || !p->dst) return SIGMA_OPS_ERR_ARG;
    sigma::TrArgs a{};
    a.src = p->src; a.dst = p->dst; a.R = p->rows; a.C = p->cols;
    a.src_bs = p->src_batch_stride; a.src_rs = p->src_row_stride; a.dst_bs = p->dst_batch_stride; a.dst_rs = p->dst_row_stride;
    a.tiles_r = (p->rows + 31) / 32; a.tiles_c = (p->cols + 31) / 32;
    const long n = (long)p->batch * a.tiles_r * a.tiles_c;
    if (n > 2147483647L) return SIGMA_OPS_ERR_ARG;
    hipLaunchKernelGGL(sigma::transpose2d_kernel, dim3((unsigned)n), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SIGMA_OPS_OK : SIGMA_OPS_ERR_LAUNCH;
}

}  // extern "C"
