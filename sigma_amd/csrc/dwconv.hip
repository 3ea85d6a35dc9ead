// dwconv.hip -- depthwise 3x3 convolution + SiLU of SS2D, fused with the CrossScan layout step.
//
// Reference (models/encoders/vmamba.py:1071-1072, 683-691):
//     x = x.permute(0, 3, 1, 2).contiguous();  x = self.act(self.conv2d(x))       # Conv2d(d, d, 3, pad 1, groups=d) + SiLU
// followed by CrossScan's four permuted copies (vmamba.py:80-98).  On MI355X all of it is HBM-bound
// stencil / transpose work, so it is one pass: read the (B, d, H, W) plane once, write the activation in
// BOTH memory orders the scan kernels consume (row-major, and column-major through a padded LDS tile so
// that both stores are coalesced).  MIOpen serves this shape with its "naive_conv" fallbacks.
//
//   fwd :  y = silu(conv3x3(x) + bias)                 -> out[b, 0, c, h*W + w],  out[b, 1, c, w*H + h]
//   bwd1:  gpre = (g[b,0,c,h*W+w] + g[b,1,c,w*H+h]) * silu'(conv3x3(x) + bias)    (pre-activation recomputed)
//          dW[c, :, :] += sum gpre * x(shifted),  dbias[c] += sum gpre             (block reduce + one atomic each)
//   bwd2:  dx = correlate(gpre, W)                                                  (transposed stencil)
//
// One workgroup = one 32x32 spatial tile of one (batch, channel) plane, 256 threads as 32 x 8, four rows
// per thread.  Neighbour taps come straight from global memory (each value is re-used 9x out of L1).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sigma_ops.h"
#include "scan_device.h"

namespace sigma {

namespace {

constexpr int kTile = 32;

struct DwArgs {
    const float* x; const float* w; const float* bias;
    float* out2;            // fwd: (B, 2, d, L)
    const float* g2;        // bwd1: (B, 2, d, L)
    float* gpre;            // bwd1 out / bwd2 in: (B, d, H, W)
    float* dw; float* dbias;   // (d, 9), (d): accumulated
    float* dx;              // bwd2 out: (B, d, H, W)
    int B, d, H, W, orders;
    long x_bs, x_cs;        // plane (b, c) of x / dx at b * x_bs + c * x_cs floats (packed: d * L, L)
};

// first element of plane `plane_id` = b * d + c of x / dx (the other tensors are packed)
__device__ __forceinline__ long x_plane_offset(const DwArgs& a, int plane_id) {
    const int b = plane_id / a.d, c = plane_id - b * a.d;
    return (long)b * a.x_bs + (long)c * a.x_cs;
}

__device__ __forceinline__ float sigmoidf_fast(float v) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-v * 1.4426950408889634f));
}

// DPP row / bcast adds (scan_device.h wave_sum) instead of six ds_bpermute per sum: ten sums per tile
__device__ __forceinline__ float wave_sum_shfl(float v) { return wave_sum(v); }

// taps t[0..8] = x[h-1..h+1][w-1..w+1] (0 outside the plane)
__device__ __forceinline__ void load_taps(const float* __restrict__ plane, int h, int w, int H, int W, float (&t)[9]) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int hh = h + dy - 1;
        const bool hin = hh >= 0 && hh < H;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ww = w + dx - 1;
            t[dy * 3 + dx] = (hin && ww >= 0 && ww < W) ? plane[(long)hh * W + ww] : 0.0f;
        }
    }
}

__global__ void __launch_bounds__(256) dwconv_silu_fwd_kernel(const DwArgs a) {
    __shared__ float tile[kTile][kTile + 1];
    const int H = a.H, W = a.W;
    const long L = (long)H * W;
    const int tw = (W + kTile - 1) / kTile, th = (H + kTile - 1) / kTile;
    const int lbk = xcd_logical_block((int)blockIdx.x, (int)gridDim.x);   // tiles of a plane share lines: same XCD (L2)
    const int plane_id = lbk / (tw * th);           // b * d + c
    const int tile_id = lbk - plane_id * (tw * th);
    const int c = plane_id % a.d;
    const int b = plane_id / a.d;
    const int w0 = (tile_id % tw) * kTile, h0 = (tile_id / tw) * kTile;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* __restrict__ plane = a.x + x_plane_offset(a, plane_id);
    float* __restrict__ o_rm = a.out2 + ((long)(b * a.orders + 0) * a.d + c) * L;
    float* __restrict__ o_cm = a.out2 + ((long)(b * a.orders + 1) * a.d + c) * L;
    float wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = a.w[c * 9 + i];
    const float bias = a.bias ? a.bias[c] : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int hl = ty + 8 * i;
        const int h = h0 + hl, w = w0 + tx;
        float y = 0.0f;
        if (h < H && w < W) {
            float t[9];
            load_taps(plane, h, w, H, W, t);
            float acc = bias;
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = fmaf(wk[k], t[k], acc);
            y = acc * sigmoidf_fast(acc);
            o_rm[(long)h * W + w] = y;
        }
        tile[hl][tx] = y;
    }
    if (a.orders < 2) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int wl = ty + 8 * i;
        const int w = w0 + wl, h = h0 + tx;
        if (h < H && w < W) o_cm[(long)w * H + h] = tile[tx][wl];
    }
}

__global__ void __launch_bounds__(256) dwconv_silu_bwd1_kernel(const DwArgs a) {
    __shared__ float tile[kTile][kTile + 1];
    __shared__ float red[4][10];
    const int H = a.H, W = a.W;
    const long L = (long)H * W;
    const int tw = (W + kTile - 1) / kTile, th = (H + kTile - 1) / kTile;
    const int lbk = xcd_logical_block((int)blockIdx.x, (int)gridDim.x);
    const int plane_id = lbk / (tw * th);
    const int tile_id = lbk - plane_id * (tw * th);
    const int c = plane_id % a.d;
    const int b = plane_id / a.d;
    const int w0 = (tile_id % tw) * kTile, h0 = (tile_id / tw) * kTile;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* __restrict__ plane = a.x + x_plane_offset(a, plane_id);
    const float* __restrict__ g_rm = a.g2 + ((long)(b * a.orders + 0) * a.d + c) * L;
    const float* __restrict__ g_cm = a.g2 + ((long)(b * a.orders + 1) * a.d + c) * L;
    float* __restrict__ gp = a.gpre + (long)plane_id * L;
    // column-major gradient tile -> LDS, coalesced along h
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int wl = ty + 8 * i;
        const int w = w0 + wl, h = h0 + tx;
        tile[tx][wl] = (a.orders > 1 && h < H && w < W) ? g_cm[(long)w * H + h] : 0.0f;
    }
    __syncthreads();
    float wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = a.w[c * 9 + i];
    const float bias = a.bias ? a.bias[c] : 0.0f;
    float acc_w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    float acc_b = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int hl = ty + 8 * i;
        const int h = h0 + hl, w = w0 + tx;
        if (h < H && w < W) {
            float t[9];
            load_taps(plane, h, w, H, W, t);
            float pre = bias;
#pragma unroll
            for (int k = 0; k < 9; ++k) pre = fmaf(wk[k], t[k], pre);
            const float sg = sigmoidf_fast(pre);
            const float dsilu = sg * fmaf(pre, 1.0f - sg, 1.0f);           // d/dp [p * sigmoid(p)]
            const float g = (g_rm[(long)h * W + w] + tile[hl][tx]) * dsilu;
            gp[(long)h * W + w] = g;
            acc_b += g;
#pragma unroll
            for (int k = 0; k < 9; ++k) acc_w[k] = fmaf(g, t[k], acc_w[k]);
        }
    }
    // block reduction of the 10 per-channel sums, then one atomic each
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float s = wave_sum_shfl(acc_w[k]);
        if (lane == 0) red[wave][k] = s;
    }
    {
        const float s = wave_sum_shfl(acc_b);
        if (lane == 0) red[wave][9] = s;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (threadIdx.x < 9) atomicAdd(a.dw + c * 9 + threadIdx.x, s);
        else if (a.dbias) atomicAdd(a.dbias + c, s);
    }
}

__global__ void __launch_bounds__(256) dwconv_bwd2_kernel(const DwArgs a) {
    const int H = a.H, W = a.W;
    const long L = (long)H * W;
    const int tw = (W + kTile - 1) / kTile, th = (H + kTile - 1) / kTile;
    const int lbk = xcd_logical_block((int)blockIdx.x, (int)gridDim.x);
    const int plane_id = lbk / (tw * th);
    const int tile_id = lbk - plane_id * (tw * th);
    const int c = plane_id % a.d;
    const int w0 = (tile_id % tw) * kTile, h0 = (tile_id / tw) * kTile;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* __restrict__ gp = a.gpre + (long)plane_id * L;
    float* __restrict__ dx = a.dx + x_plane_offset(a, plane_id);
    float wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = a.w[c * 9 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int h = h0 + ty + 8 * i, w = w0 + tx;
        if (h < H && w < W) {
            float t[9];
            load_taps(gp, h, w, H, W, t);      // t[dy*3+dx] = gpre[h+dy-1][w+dx-1]
            // y[h'][w'] used x[h][w] with tap (h - h' + 1, w - w' + 1): dx[h][w] = sum_k W[8-k] * gpre tap k
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = fmaf(wk[8 - k], t[k], acc);
            dx[(long)h * W + w] = acc;
        }
    }
}

// ---- whole-plane variants --------------------------------------------------------------------------------------------
// Planes that fit LDS (everything below the 120 x 160 first stage): ONE workgroup owns a whole (batch, channel) plane.
// The 32 x 32 tiles above leave 41 % of their lanes idle on a 30 x 40 plane (two tiles, the second 8 columns wide) and fetch
// every tap with its own 4-byte global load; here the plane is read once with 16-byte loads into a zero-bordered LDS image,
// taps are LDS reads, and the backward keeps the pre-activation gradient in LDS between its two stencils (no gpre round
// trip through HBM, one launch instead of two).  Row pitch odd: walking down a column (the column-major order of the
// scan) touches every bank once.
__device__ __forceinline__ int plane_pitch(int W) { return (W + 2) | 1; }

__global__ void __launch_bounds__(256) dwconv_silu_fwd_plane_kernel(const DwArgs a) {
    extern __shared__ float smem[];
    const int H = a.H, W = a.W, L = H * W, pitch = plane_pitch(W);
    float* __restrict__ sIn = smem;                       // (H + 2) x pitch, zero border
    float* __restrict__ sOut = smem + (H + 2) * pitch;    // H x pitch
    const int plane_id = blockIdx.x;
    const int c = plane_id % a.d, b = plane_id / a.d;
    const int tid = threadIdx.x;
    const float* __restrict__ plane = a.x + x_plane_offset(a, plane_id);
    float* __restrict__ o_rm = a.out2 + ((long)(b * a.orders + 0) * a.d + c) * L;
    float* __restrict__ o_cm = a.out2 + ((long)(b * a.orders + 1) * a.d + c) * L;
    for (int i = tid; i < (H + 2) * pitch; i += 256) sIn[i] = 0.0f;
    __syncthreads();
    if ((W & 3) == 0 && (reinterpret_cast<uintptr_t>(plane) & 15u) == 0) {
        for (int i4 = tid; i4 < (L >> 2); i4 += 256) {
            const float4 v = reinterpret_cast<const float4*>(plane)[i4];
            const int idx = i4 << 2, h = idx / W, w = idx - h * W;
            float* __restrict__ o = sIn + (h + 1) * pitch + (w + 1);
            o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        }
    } else {
        for (int idx = tid; idx < L; idx += 256) { const int h = idx / W, w = idx - h * W; sIn[(h + 1) * pitch + (w + 1)] = plane[idx]; }
    }
    __syncthreads();
    float wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = a.w[c * 9 + i];
    const float bias = a.bias ? a.bias[c] : 0.0f;
    for (int idx = tid; idx < L; idx += 256) {
        const int h = idx / W, w = idx - h * W;
        const float* __restrict__ p = sIn + h * pitch + w;             // tap (h - 1, w - 1)
        float acc = bias;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) acc = fmaf(wk[dy * 3 + dx], p[dy * pitch + dx], acc);
        const float y = acc * sigmoidf_fast(acc);
        o_rm[idx] = y;
        sOut[h * pitch + w] = y;
    }
    if (a.orders < 2) return;
    __syncthreads();
    for (int idx = tid; idx < L; idx += 256) { const int w = idx / H, h = idx - w * H; o_cm[idx] = sOut[h * pitch + w]; }
}

__global__ void __launch_bounds__(256) dwconv_silu_bwd_plane_kernel(const DwArgs a) {
    extern __shared__ float smem[];
    __shared__ float red[4][10];
    const int H = a.H, W = a.W, L = H * W, pitch = plane_pitch(W);
    float* __restrict__ sIn = smem;                       // x, zero border
    float* __restrict__ sG = smem + (H + 2) * pitch;      // column-major gradient, then gpre; zero border
    const int plane_id = blockIdx.x;
    const int c = plane_id % a.d, b = plane_id / a.d;
    const int tid = threadIdx.x;
    const float* __restrict__ plane = a.x + x_plane_offset(a, plane_id);
    const float* __restrict__ g_rm = a.g2 + ((long)(b * a.orders + 0) * a.d + c) * L;
    const float* __restrict__ g_cm = a.g2 + ((long)(b * a.orders + 1) * a.d + c) * L;
    float* __restrict__ dx = a.dx + x_plane_offset(a, plane_id);
    for (int i = tid; i < 2 * (H + 2) * pitch; i += 256) smem[i] = 0.0f;
    __syncthreads();
    if ((W & 3) == 0 && (reinterpret_cast<uintptr_t>(plane) & 15u) == 0) {
        for (int i4 = tid; i4 < (L >> 2); i4 += 256) {
            const float4 v = reinterpret_cast<const float4*>(plane)[i4];
            const int idx = i4 << 2, h = idx / W, w = idx - h * W;
            float* __restrict__ o = sIn + (h + 1) * pitch + (w + 1);
            o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        }
    } else {
        for (int idx = tid; idx < L; idx += 256) { const int h = idx / W, w = idx - h * W; sIn[(h + 1) * pitch + (w + 1)] = plane[idx]; }
    }
    if (a.orders > 1)
        for (int idx = tid; idx < L; idx += 256) { const int w = idx / H, h = idx - w * H; sG[(h + 1) * pitch + (w + 1)] = g_cm[idx]; }
    __syncthreads();
    float wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = a.w[c * 9 + i];
    const float bias = a.bias ? a.bias[c] : 0.0f;
    float acc_w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    float acc_b = 0.0f;
    for (int idx = tid; idx < L; idx += 256) {
        const int h = idx / W, w = idx - h * W;
        const float* __restrict__ p = sIn + h * pitch + w;
        float t[9];
        float pre = bias;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx_ = 0; dx_ < 3; ++dx_) { t[dy * 3 + dx_] = p[dy * pitch + dx_]; pre = fmaf(wk[dy * 3 + dx_], t[dy * 3 + dx_], pre); }
        const float sg = sigmoidf_fast(pre);
        const float dsilu = sg * fmaf(pre, 1.0f - sg, 1.0f);
        float* __restrict__ own = sG + (h + 1) * pitch + (w + 1);       // read and written by this thread only
        const float g = (g_rm[idx] + *own) * dsilu;
        *own = g;
        acc_b += g;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc_w[k] = fmaf(g, t[k], acc_w[k]);
    }
    __syncthreads();
    for (int idx = tid; idx < L; idx += 256) {
        const int h = idx / W, w = idx - h * W;
        const float* __restrict__ q = sG + h * pitch + w;              // gpre[h - 1][w - 1]
        float acc = 0.0f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx_ = 0; dx_ < 3; ++dx_) acc = fmaf(wk[8 - (dy * 3 + dx_)], q[dy * pitch + dx_], acc);
        dx[idx] = acc;
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float s = wave_sum_shfl(acc_w[k]);
        if (lane == 0) red[wave][k] = s;
    }
    {
        const float s = wave_sum_shfl(acc_b);
        if (lane == 0) red[wave][9] = s;
    }
    __syncthreads();
    if (tid < 10) {
        const float s = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        if (tid < 9) atomicAdd(a.dw + c * 9 + tid, s);
        else if (a.dbias) atomicAdd(a.dbias + c, s);
    }
}

// LDS bytes of the whole-plane kernels (two (H + 2) x pitch images), 0 = the plane does not fit: tiled kernels
size_t plane_lds_bytes(const sigma_dwconv_params* p) {
    const long pitch = (p->width + 2) | 1;
    const long bytes = 2L * (p->height + 2) * pitch * (long)sizeof(float);
    return bytes <= 48 * 1024 ? (size_t)bytes : 0;
}

int check(const sigma_dwconv_params* p) {
    if (!p) return SIGMA_OPS_ERR_ARG;
    if (p->batch < 0 || p->channels <= 0 || p->height <= 0 || p->width <= 0) return SIGMA_OPS_ERR_ARG;
    if (p->n_orders != 1 && p->n_orders != 2) return SIGMA_OPS_ERR_ARG;
    const long tiles = (long)((p->width + kTile - 1) / kTile) * ((p->height + kTile - 1) / kTile);
    if ((long)p->batch * p->channels * tiles > 2147483647L) return SIGMA_OPS_ERR_ARG;
    return SIGMA_OPS_OK;
}

dim3 grid_for(const sigma_dwconv_params* p) {
    const long tiles = (long)((p->width + kTile - 1) / kTile) * ((p->height + kTile - 1) / kTile);
    return dim3((unsigned)(tiles * p->batch * p->channels));
}

}  // namespace

}  // namespace sigma

namespace sigma {
// plane strides of x / dx: 0 / 0 = packed (B, d, H, W); otherwise both given, planes contiguous and non-overlapping
bool plane_strides(const sigma_dwconv_params* p, DwArgs& a) {
    const long L = (long)p->height * p->width;
    if (p->x_batch_stride == 0 && p->x_channel_stride == 0) { a.x_bs = (long)p->channels * L; a.x_cs = L; return true; }
    if (p->x_batch_stride < L || p->x_channel_stride < L) return false;
    a.x_bs = p->x_batch_stride; a.x_cs = p->x_channel_stride;
    return true;
}
}  // namespace sigma

extern "C" {

int sigma_dwconv3x3_silu_fwd(const sigma_dwconv_params* p, void* stream) {
    int rc = sigma::check(p);
    if (rc) return rc;
    if (p->batch == 0) return SIGMA_OPS_OK;
    if (!p->x || !p->weight || !p->out2) return SIGMA_OPS_ERR_ARG;
    sigma::DwArgs a{};
    a.x = p->x; a.w = p->weight; a.bias = p->bias; a.out2 = p->out2;
    a.B = p->batch; a.d = p->channels; a.H = p->height; a.W = p->width; a.orders = p->n_orders;
    if (!sigma::plane_strides(p, a)) return SIGMA_OPS_ERR_ARG;
    if (const size_t lds = sigma::plane_lds_bytes(p))
        hipLaunchKernelGGL(sigma::dwconv_silu_fwd_plane_kernel, dim3((unsigned)(p->batch * p->channels)), dim3(256), lds,
                           static_cast<hipStream_t>(stream), a);
    else
        hipLaunchKernelGGL(sigma::dwconv_silu_fwd_kernel, sigma::grid_for(p), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? SIGMA_OPS_OK : SIGMA_OPS_ERR_LAUNCH;
}

int sigma_dwconv3x3_silu_bwd(const sigma_dwconv_params* p, void* stream) {
    int rc = sigma::check(p);
    if (rc) return rc;
    if (p->batch == 0) return SIGMA_OPS_OK;
    if (!p->x || !p->weight || !p->g2 || !p->gpre || !p->dweight || !p->dx) return SIGMA_OPS_ERR_ARG;
    sigma::DwArgs a{};
    a.x = p->x; a.w = p->weight; a.bias = p->bias; a.g2 = p->g2; a.gpre = p->gpre;
    a.dw = p->dweight; a.dbias = p->dbias; a.dx = p->dx;
    a.B = p->batch; a.d = p->channels; a.H = p->height; a.W = p->width; a.orders = p->n_orders;
    if (!sigma::plane_strides(p, a)) return SIGMA_OPS_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (const size_t lds = sigma::plane_lds_bytes(p)) {      // one launch, gpre stays in LDS (p->gpre is not written)
        hipLaunchKernelGGL(sigma::dwconv_silu_bwd_plane_kernel, dim3((unsigned)(p->batch * p->channels)), dim3(256), lds, s, a);
        return hipGetLastError() == hipSuccess ? SIGMA_OPS_OK : SIGMA_OPS_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(sigma::dwconv_silu_bwd1_kernel, sigma::grid_for(p), dim3(256), 0, s, a);
    if (hipGetLastError() != hipSuccess) return SIGMA_OPS_ERR_LAUNCH;
    hipLaunchKernelGGL(sigma::dwconv_bwd2_kernel, sigma::grid_for(p), dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? SIGMA_OPS_OK : SIGMA_OPS_ERR_LAUNCH;
}

}  // extern "C"
