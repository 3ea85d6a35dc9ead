// layernorm.hip -- LayerNorm over the channel (last) dimension, forward and backward.
//
// Reference: every nn.LayerNorm on the hot path -- VSSBlock.norm (vmamba.py:1693), SS2D.out_norm
// (vmamba.py:724), PatchMerging2D.norm (:617), the fusion blocks' out_norm_{1,2} (:1448-1449, 1183-1184)
// and the decoder norms (MambaDecoder.py:18,41,85, vmamba.py:1783,1797); eps 1e-5, affine.
// ATen's ROCm kernels spend 240 us on the backward of a (19200 x 384) call (three kernels, profile
// r01); the op is a pure HBM stream: forward 1 read + 1 write, backward 2 reads + 1 write.
//
// One WAVE per row: lane i owns the float4 columns {4*i + 256*j}; mean / variance by two wave
// reductions over register-resident values (no LDS, no barriers).  Backward: dx needs two row sums
// (sum g*gamma, sum g*gamma*xhat); dgamma / dbeta are column sums over all rows -- every wave keeps
// per-lane partials for its (fixed) columns across the rows it walks and writes ONE partial row per
// wave; a second tiny kernel adds the partial rows in a fixed order (deterministic, no atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sigma_ops.h"
#include "scan_device.h"

namespace sigma {

namespace {

// Sum over the 64 lanes, result in every lane: six DPP adds (row_shr 1/2/4/8, row_bcast 15/31) + v_readlane 63
// (scan_device.h wave_sum).  Round 2 used __shfl_xor here, i.e. six ds_bpermute_b32 per sum through the LDS crossbar
// (24 clocks each, tools/ubench/issue_ubench.hip) in a dependent chain: 12-24 of them per row made the backward
// latency-bound at 0.17-0.37 of the copy ceiling (profiles/r03_aux_roofline.jsonl).
__device__ __forceinline__ float wave_allsum(float v) { return wave_sum(v); }

struct LnArgs {
    const float* x; const float* gamma; const float* beta; const float* dy;
    float* y; float* mean; float* rstd; float* dx; float* ws;   // ws: [nwaves][2][C]
    const float* z; float* dz; long z_stride, dz_stride;          // optional SiLU gate: y = LN(x) * silu(z)
    const float* rsc; long rows_per_scale;                        // optional per-sample output factor (stochastic depth)
    const float* dx_add;                                          // optional addend of dx (the gradient of the residual stream)
    long M; int C; float eps;
};

__device__ __forceinline__ float sigmoid_f(float v) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-v * 1.4426950408889634f));
}

template <int NV>
__global__ void __launch_bounds__(256) ln_fwd_kernel(const LnArgs a) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    const int C = a.C;
    float4 g[NV], b[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane * 4 + 256 * j;
        g[j] = c < C ? *reinterpret_cast<const float4*>(a.gamma + c) : make_float4(0, 0, 0, 0);
        b[j] = (c < C && a.beta) ? *reinterpret_cast<const float4*>(a.beta + c) : make_float4(0, 0, 0, 0);
    }
    const float inv = 1.0f / (float)C;
    // sample of the row for the per-sample factor: ONE division per wave, then carried along (a 64-bit division per row
    // cost more than the row itself at C <= 256: 69 -> 106 us on the 153600 x 192 backward)
    long sb = a.rsc ? wave / a.rows_per_scale : 0, srem = a.rsc ? wave - sb * a.rows_per_scale : 0;
    // The operands of the NEXT row are requested before the reductions of the current one (registers, two rows in flight
    // per wave): a wave that loads, reduces twice and stores one row at a time keeps ~3 KB in flight, and the kernels
    // then sit at 0.35-0.4 of the copy ceiling whatever the occupancy (profiles/r03_aux_roofline_v4.jsonl).
    const bool gatedf = a.z != nullptr;
    auto load_row = [&](long r, float4 (&xv)[NV], float4 (&zv)[NV]) {
        const float* __restrict__ xr = a.x + r * C;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane * 4 + 256 * j;
            xv[j] = c < C ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0, 0, 0, 0);
            zv[j] = (gatedf && c < C) ? *reinterpret_cast<const float4*>(a.z + r * a.z_stride + c) : make_float4(0, 0, 0, 0);
        }
    };
    constexpr bool kPrefetch = NV <= 2;        // wider rows: the second row in flight costs the occupancy it buys
    float4 v[NV], zc[NV];
    if (kPrefetch && wave < a.M) load_row(wave, v, zc);
    for (long r = wave; r < a.M; r += nwaves) {
        if (!kPrefetch) load_row(r, v, zc);
        float4 vn[NV], zn[NV];
        const bool more = kPrefetch && r + nwaves < a.M;
        if (more) load_row(r + nwaves, vn, zn);
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < NV; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        const float mu = wave_allsum(s) * inv;
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane * 4 + 256 * j;
            if (c < C) {
                const float dx0 = v[j].x - mu, dx1 = v[j].y - mu, dx2 = v[j].z - mu, dx3 = v[j].w - mu;
                q += (dx0 * dx0 + dx1 * dx1) + (dx2 * dx2 + dx3 * dx3);
            }
        }
        const float rs = rsqrtf(wave_allsum(q) * inv + a.eps);
        float* __restrict__ yr = a.y + r * C;
        float sc = 1.0f;
        if (a.rsc) {
            sc = a.rsc[sb];
            srem += nwaves;
            while (srem >= a.rows_per_scale) { srem -= a.rows_per_scale; ++sb; }
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane * 4 + 256 * j;
            if (c < C) {
                float4 o;
                o.x = (v[j].x - mu) * rs * g[j].x + b[j].x;
                o.y = (v[j].y - mu) * rs * g[j].y + b[j].y;
                o.z = (v[j].z - mu) * rs * g[j].z + b[j].z;
                o.w = (v[j].w - mu) * rs * g[j].w + b[j].w;
                if (gatedf) {
                    const float4 zv = zc[j];
                    o.x *= zv.x * sigmoid_f(zv.x); o.y *= zv.y * sigmoid_f(zv.y);
                    o.z *= zv.z * sigmoid_f(zv.z); o.w *= zv.w * sigmoid_f(zv.w);
                }
                if (a.rsc) { o.x *= sc; o.y *= sc; o.z *= sc; o.w *= sc; }
                *reinterpret_cast<float4*>(yr + c) = o;
            }
        }
        if (lane == 0 && a.mean) { a.mean[r] = mu; a.rstd[r] = rs; }
        if (more) {
#pragma unroll
            for (int j = 0; j < NV; ++j) { v[j] = vn[j]; zc[j] = zn[j]; }
        }
    }
}

template <int NV>
__global__ void __launch_bounds__(256) ln_bwd_kernel(const LnArgs a) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    const int C = a.C;
    float4 g[NV], dg[NV], db[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane * 4 + 256 * j;
        g[j] = c < C ? *reinterpret_cast<const float4*>(a.gamma + c) : make_float4(0, 0, 0, 0);
        dg[j] = make_float4(0, 0, 0, 0);
        db[j] = make_float4(0, 0, 0, 0);
    }
    const bool gated = a.z != nullptr;
    const float inv = 1.0f / (float)C;
    long sb = a.rsc ? wave / a.rows_per_scale : 0, srem = a.rsc ? wave - sb * a.rows_per_scale : 0;
    // next row's operands requested before this row's reductions (see ln_fwd_kernel)
    auto load_row = [&](long r, float4 (&xv)[NV], float4 (&gv)[NV], float4 (&zv)[NV], float& mu, float& rs) {
        const float* __restrict__ xr = a.x + r * C;
        const float* __restrict__ gr = a.dy + r * C;
        mu = a.mean[r]; rs = a.rstd[r];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane * 4 + 256 * j;
            const bool in = c < C;
            xv[j] = in ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0, 0, 0, 0);
            gv[j] = in ? *reinterpret_cast<const float4*>(gr + c) : make_float4(0, 0, 0, 0);
            zv[j] = (gated && in) ? *reinterpret_cast<const float4*>(a.z + r * a.z_stride + c) : make_float4(0, 0, 0, 0);
        }
    };
    float4 bv_[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane * 4 + 256 * j;
        bv_[j] = (gated && a.beta && c < C) ? *reinterpret_cast<const float4*>(a.beta + c) : make_float4(0, 0, 0, 0);
    }
    constexpr bool kPrefetch = NV <= 2;
    float4 xc[NV], gc[NV], zc[NV];
    float mu = 0.0f, rs = 0.0f;
    if (kPrefetch && wave < a.M) load_row(wave, xc, gc, zc, mu, rs);
    for (long r = wave; r < a.M; r += nwaves) {
        if (!kPrefetch) load_row(r, xc, gc, zc, mu, rs);
        float4 xn[NV], gn[NV], zn[NV];
        float mun = 0.0f, rsn = 0.0f;
        const bool more = kPrefetch && r + nwaves < a.M;
        if (more) load_row(r + nwaves, xn, gn, zn, mun, rsn);
        float sc = 1.0f;
        if (a.rsc) {
            sc = a.rsc[sb];
            srem += nwaves;
            while (srem >= a.rows_per_scale) { srem -= a.rows_per_scale; ++sb; }
        }
        float4 xh[NV], t[NV];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane * 4 + 256 * j;
            if (c < C) {
                const float4 xv = xc[j];
                float4 gv = gc[j];
                if (a.rsc) { gv.x *= sc; gv.y *= sc; gv.z *= sc; gv.w *= sc; }
                xh[j] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                if (gated) {
                    // out = n * silu(z), n = xhat * gamma + beta:  dn = dout * silu(z),  dz = dout * n * silu'(z)
                    const float4 zv = zc[j];
                    const float4 bv = bv_[j];
                    const float s0 = sigmoid_f(zv.x), s1_ = sigmoid_f(zv.y), s2_ = sigmoid_f(zv.z), s3 = sigmoid_f(zv.w);
                    float4 dzv;
                    dzv.x = gv.x * fmaf(xh[j].x, g[j].x, bv.x) * s0 * fmaf(zv.x, 1.0f - s0, 1.0f);
                    dzv.y = gv.y * fmaf(xh[j].y, g[j].y, bv.y) * s1_ * fmaf(zv.y, 1.0f - s1_, 1.0f);
                    dzv.z = gv.z * fmaf(xh[j].z, g[j].z, bv.z) * s2_ * fmaf(zv.z, 1.0f - s2_, 1.0f);
                    dzv.w = gv.w * fmaf(xh[j].w, g[j].w, bv.w) * s3 * fmaf(zv.w, 1.0f - s3, 1.0f);
                    *reinterpret_cast<float4*>(a.dz + r * a.dz_stride + c) = dzv;
                    gv.x *= zv.x * s0; gv.y *= zv.y * s1_; gv.z *= zv.z * s2_; gv.w *= zv.w * s3;
                }
                t[j] = make_float4(gv.x * g[j].x, gv.y * g[j].y, gv.z * g[j].z, gv.w * g[j].w);
                s1 += (t[j].x + t[j].y) + (t[j].z + t[j].w);
                s2 += (t[j].x * xh[j].x + t[j].y * xh[j].y) + (t[j].z * xh[j].z + t[j].w * xh[j].w);
                dg[j].x += gv.x * xh[j].x; dg[j].y += gv.y * xh[j].y; dg[j].z += gv.z * xh[j].z; dg[j].w += gv.w * xh[j].w;
                db[j].x += gv.x; db[j].y += gv.y; db[j].z += gv.z; db[j].w += gv.w;
            } else {
                xh[j] = make_float4(0, 0, 0, 0);
                t[j] = make_float4(0, 0, 0, 0);
            }
        }
        const float m1 = wave_allsum(s1) * inv, m2 = wave_allsum(s2) * inv;
        float* __restrict__ dr = a.dx + r * C;
        const float* __restrict__ ar = a.dx_add ? a.dx_add + r * C : nullptr;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane * 4 + 256 * j;
            if (c < C) {
                float4 o;
                o.x = rs * (t[j].x - m1 - xh[j].x * m2);
                o.y = rs * (t[j].y - m1 - xh[j].y * m2);
                o.z = rs * (t[j].z - m1 - xh[j].z * m2);
                o.w = rs * (t[j].w - m1 - xh[j].w * m2);
                if (ar) {                                       // + the gradient that reached x past the LayerNorm
                    const float4 e = *reinterpret_cast<const float4*>(ar + c);
                    o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
                }
                *reinterpret_cast<float4*>(dr + c) = o;
            }
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < NV; ++j) { xc[j] = xn[j]; gc[j] = gn[j]; zc[j] = zn[j]; }
            mu = mun; rs = rsn;
        }
    }
    // the four waves of the workgroup meet in LDS: ONE partial (dgamma, dbeta) row per workgroup for ln_reduce_kernel
    // (a row per wave was 4x the partial traffic and made the 48-96 workgroups of the reduction 20 us long)
    extern __shared__ float part[];                           // [4][2][C]
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane * 4 + 256 * j;
        if (c < C) {
            *reinterpret_cast<float4*>(part + (wv * 2 + 0) * C + c) = dg[j];
            *reinterpret_cast<float4*>(part + (wv * 2 + 1) * C + c) = db[j];
        }
    }
    __syncthreads();
    float* __restrict__ wrow = a.ws + (long)blockIdx.x * 2 * C;
    for (int i = threadIdx.x * 4; i < 2 * C; i += 1024) {
        const float4 p0 = *reinterpret_cast<const float4*>(part + i), p1 = *reinterpret_cast<const float4*>(part + 2 * C + i);
        const float4 p2 = *reinterpret_cast<const float4*>(part + 4 * C + i), p3 = *reinterpret_cast<const float4*>(part + 6 * C + i);
        *reinterpret_cast<float4*>(wrow + i) = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y),
                                                          (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w));
    }
}

// dgamma[c] = sum_w ws[w][0][c], dbeta[c] = sum_w ws[w][1][c]   (fixed order)
// 16 columns x 16 row groups per workgroup: a thread adds every 16th partial row with four independent
// accumulators (the loads of 4 x 16 rows are in flight together), then the 16 groups meet in LDS.
__global__ void __launch_bounds__(256) ln_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, int nwaves, int C) {
    __shared__ float red[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + tx;                       // over 2*C
    const bool live = i < 2 * C;
    const int which = live ? i / C : 0, c = live ? i - which * C : 0;
    const float* __restrict__ col = ws + (long)which * C + c;
    const long stride = 2L * C;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    if (live) {
        int w = ty;
        for (; w + 48 < nwaves; w += 64) {
            a0 += col[(long)w * stride];
            a1 += col[(long)(w + 16) * stride];
            a2 += col[(long)(w + 32) * stride];
            a3 += col[(long)(w + 48) * stride];
        }
        for (; w < nwaves; w += 16) a0 += col[(long)w * stride];
    }
    red[ty][tx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ty == 0 && live) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += red[k][tx];
        if (which == 0) dgamma[c] = acc;
        else if (dbeta) dbeta[c] = acc;
    }
}

template <typename F>
bool dispatch_nv(int C, F&& f) {
    const int nv = (C + 255) / 256;
    switch (nv) {
        case 1: f(std::integral_constant<int, 1>()); return true;
        case 2: f(std::integral_constant<int, 2>()); return true;
        case 3: f(std::integral_constant<int, 3>()); return true;
        case 4: f(std::integral_constant<int, 4>()); return true;
        case 5: case 6: f(std::integral_constant<int, 6>()); return true;
        case 7: case 8: f(std::integral_constant<int, 8>()); return true;
        default: return false;
    }
}

int grid_blocks(long M, long cap = 2048) {   // 2048 blocks = 8192 waves: 8 per SIMD on 256 CUs
    long b = (M + 3) / 4;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}
constexpr long kBwdBlocks = 512;             // floor: 2048 waves
// workgroups of the backward: up to 8 waves per SIMD while the partial rows (2 x C floats per workgroup) stay within 2 MB
long bwd_blocks(long rows, int C) {
    long cap = (2L << 20) / (8L * C);
    if (cap > 2048) cap = 2048;
    if (cap < kBwdBlocks) cap = kBwdBlocks;
    return cap;
}

// Resident workgroups per CU of one instantiation (registers / LDS decide; measured once).  The grids are sized to ONE
// resident round: with the next-row prefetch ln_bwd_kernel<1> went from 8 to 5 waves per SIMD, and a grid still sized for
// 8 ran 1.6 rounds (69 -> 83 us on the 153600-row launches).
template <int NV, bool BWD>
int resident_blocks() {
    static int cached = 0;
    if (cached == 0) {
        int n = 0;
        hipError_t e = BWD ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ln_bwd_kernel<NV>, 256, 8 * 256 * NV * sizeof(float))
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, ln_fwd_kernel<NV>, 256, 0);
        cached = (e == hipSuccess && n > 0) ? (n > 8 ? 8 : n) : 4;
    }
    return cached;
}

int fwd_grid(long rows, int C) {
    int occ = 8;
    dispatch_nv(C, [&](auto nv) { occ = resident_blocks<decltype(nv)::value, false>(); });
    return grid_blocks(rows, 256L * occ);
}

int bwd_grid(long rows, int C) {
    int occ = 8;
    dispatch_nv(C, [&](auto nv) { occ = resident_blocks<decltype(nv)::value, true>(); });
    long cap = bwd_blocks(rows, C);
    if (cap > 256L * occ) cap = 256L * occ;
    return grid_blocks(rows, cap);
}

bool check(const sigma_layernorm_params* p) {
    return p && p->rows >= 0 && p->channels > 0 && p->channels % 4 == 0 && p->channels <= 2048;
}

}  // namespace

}  // namespace sigma

extern "C" {

int sigma_layernorm_bwd_partial_rows(int64_t rows, int32_t channels) {
    return sigma::bwd_grid(rows, channels > 0 ? channels : 4);
}

int sigma_layernorm_fwd(const sigma_layernorm_params* p, void* stream) {
    if (!sigma::check(p)) return SIGMA_OPS_ERR_ARG;
    if (p->rows == 0) return SIGMA_OPS_OK;
    if (!p->x || !p->gamma || !p->y) return SIGMA_OPS_ERR_ARG;
    sigma::LnArgs a{};
    a.x = p->x; a.gamma = p->gamma; a.beta = p->beta; a.y = p->y; a.mean = p->mean; a.rstd = p->rstd;
    a.z = p->gate; a.z_stride = p->gate_row_stride;
    if (p->gate && (p->gate_row_stride % 4 != 0 || p->gate_row_stride < p->channels)) return SIGMA_OPS_ERR_ARG;
    a.rsc = p->row_scale; a.rows_per_scale = p->rows_per_scale;
    if (p->row_scale && p->rows_per_scale <= 0) return SIGMA_OPS_ERR_ARG;
    a.M = p->rows; a.C = p->channels; a.eps = p->eps;
    const int grid = sigma::fwd_grid(p->rows, p->channels);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool ok = sigma::dispatch_nv(p->channels, [&](auto nv) {
        hipLaunchKernelGGL(sigma::ln_fwd_kernel<decltype(nv)::value>, dim3(grid), dim3(256), 0, s, a);
    });
    if (!ok) return SIGMA_OPS_ERR_ARG;
    return hipGetLastError() == hipSuccess ? SIGMA_OPS_OK : SIGMA_OPS_ERR_LAUNCH;
}

int sigma_layernorm_bwd(const sigma_layernorm_params* p, void* stream) {
    if (!sigma::check(p)) return SIGMA_OPS_ERR_ARG;
    if (!p->dgamma) return SIGMA_OPS_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = sigma::bwd_grid(p->rows, p->channels);
    if (p->rows > 0) {
        if (!p->x || !p->gamma || !p->dy || !p->mean || !p->rstd || !p->dx || !p->workspace) return SIGMA_OPS_ERR_ARG;
        sigma::LnArgs a{};
        a.x = p->x; a.gamma = p->gamma; a.beta = p->beta; a.dy = p->dy; a.mean = p->mean; a.rstd = p->rstd; a.dx = p->dx;
        a.ws = p->workspace; a.z = p->gate; a.z_stride = p->gate_row_stride; a.dz = p->dgate;
        a.dz_stride = p->dgate_row_stride > 0 ? p->dgate_row_stride : p->channels;
        a.rsc = p->row_scale; a.rows_per_scale = p->rows_per_scale;
        a.dx_add = p->dx_add;
        if (p->dx_add && (reinterpret_cast<uintptr_t>(p->dx_add) & 15)) return SIGMA_OPS_ERR_ARG;
        if (p->row_scale && p->rows_per_scale <= 0) return SIGMA_OPS_ERR_ARG;
        if (p->gate && (!p->dgate || p->gate_row_stride % 4 != 0 || p->gate_row_stride < p->channels || a.dz_stride % 4 != 0 ||
                        a.dz_stride < p->channels))
            return SIGMA_OPS_ERR_ARG;
        a.M = p->rows; a.C = p->channels; a.eps = p->eps;
        const bool ok = sigma::dispatch_nv(p->channels, [&](auto nv) {
            hipLaunchKernelGGL(sigma::ln_bwd_kernel<decltype(nv)::value>, dim3(grid), dim3(256), 8 * p->channels * sizeof(float), s, a);
        });
        if (!ok) return SIGMA_OPS_ERR_ARG;
        if (hipGetLastError() != hipSuccess) return SIGMA_OPS_ERR_LAUNCH;
    }
    const int nw = p->rows > 0 ? grid : 0;                   // partial rows: one per workgroup
    hipLaunchKernelGGL(sigma::ln_reduce_kernel, dim3((2 * p->channels + 15) / 16), dim3(256), 0, s, p->workspace, p->dgamma,
                       p->dbeta, nw, p->channels);
    return hipGetLastError() == hipSuccess ? SIGMA_OPS_OK : SIGMA_OPS_ERR_LAUNCH;
}

}  // extern "C"
