// scan_bwd.hip -- selective-scan backward for gfx950 (MI355X), wave64.
//
// Replaces selective_scan_bwd_kernel + reverse_scan.cuh (reference:
// models/encoders/selective_scan/csrc/selective_scan/selective_scan_bwd_kernel.cuh:66-308,
// reverse_scan.cuh:18-401).  Mathematics: SURVEY.md App. E.2.
//
// Machine mapping (differs from the reference on purpose):
//   * one wave per channel row, `nwaves` rows of one (batch, group) per workgroup;
//   * the sequence is walked chunk by chunk (2048 = the checkpoint pitch of x) from the
//     end; inside a chunk a cheap forward sweep (fold + wave scan only) rebuilds the state
//     at every tile start from the chunk checkpoint, then the tiles are processed last to
//     first: forward replay (x kept in registers) + reverse affine scan of
//     e = a * dx ("what flows to the element on the left");
//   * the reverse scan across lanes is a DPP row_shl scan + uniform row-head fix-up;
//   * dB / dC are reduced over the workgroup's rows in LDS (ds_add_f32 into a swizzled
//     [state][k][lane] tile); each workgroup then writes ONE plain coalesced partial per
//     (n, l) into its slab of a caller-provided workspace and reduce_partials_kernel sums the
//     P = rows_per_group / rows_per_workgroup slabs in a fixed order -- deterministic, and no
//     device-scope float atomics (the reference issues one global atomicAdd per row: 192
//     per element at stage 0).  dA / dD / ddelta_bias are wave-reduced with DPP and hit
//     memory once per row.
#include "scan_device.h"
#include "scan_launch.h"

namespace sigma {

template <typename io_t, int T>
__global__ void __launch_bounds__(1024)
scan_bwd_kernel(const BwdArgs q) {
    using G = TileGeom<T>;
    constexpr int NB = kStateBlock;
    constexpr int TPC = 2048 / G::TILE;               // tiles per checkpoint chunk
    const FwdArgs& p = q.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nwaves = blockDim.x >> 6;
    const int N = p.N, L = p.L;
    float* sB = smem;
    float* sC = sB + NB * G::ROW;
    float* sRed = sC + NB * G::ROW;                   // [nwaves][2][TILE] per-row dB/dC terms of one state
    float* sX0 = sRed + nwaves * 2 * G::TILE;         // [nwaves][TPC][N] state at tile start
    float* sR = sX0 + nwaves * TPC * N;               // [nwaves][N] reverse carry a*dx of the tile to the right
    float* sdA = sR + nwaves * N;                     // [nwaves][N]
    float* sA = sdA + nwaves * N;                     // [nwaves][N] A[r, n]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int b = lb / p.rowblocks;
    const int rb = lb - b * p.rowblocks;
    const int row0 = rb * nwaves;
    const int r = row0 + wave;
    const int g = row0 / p.rows_per_group;
    const bool vec = p.vec_ok != 0;

    const io_t* __restrict__ u_row = reinterpret_cast<const io_t*>(p.u) + (long)b * p.u_bs + (long)r * p.u_ds;
    const io_t* __restrict__ d_row = reinterpret_cast<const io_t*>(p.delta) + (long)b * p.dt_bs + (long)r * p.dt_ds;
    const io_t* __restrict__ g_row = reinterpret_cast<const io_t*>(q.dout) + (long)b * q.g_bs + (long)r * q.g_ds;
    io_t* __restrict__ du_row = reinterpret_cast<io_t*>(q.du) + (long)b * q.du_bs + (long)r * q.du_ds;
    io_t* __restrict__ dd_row = reinterpret_cast<io_t*>(q.ddelta) + (long)b * q.dd_bs + (long)r * q.dd_ds;
    const io_t* __restrict__ Bg = reinterpret_cast<const io_t*>(p.B) + (long)b * p.B_bs + (long)g * p.B_gs;
    const io_t* __restrict__ Cg = reinterpret_cast<const io_t*>(p.C) + (long)b * p.C_bs + (long)g * p.C_gs;
    float* __restrict__ dBg = q.dB + (long)b * q.dB_bs + (long)g * q.dB_gs;
    float* __restrict__ dCg = q.dC + (long)b * q.dC_bs + (long)g * q.dC_gs;
    const float* __restrict__ A_row = p.A + (long)r * p.A_ds;
    const float bias = p.bias ? p.bias[r] : 0.0f;
    const float Dd = p.D ? p.D[r] : 0.0f;
    const float* __restrict__ x_row = p.x ? p.x + ((long)b * p.dim + r) * (long)p.n_chunks * 2 * N : nullptr;

    const long ws_slab = (q.P > 1)
        ? ((((long)((row0 - g * p.rows_per_group) / nwaves) * p.batch + b) * p.G + g) * (long)N * L) : 0;
    for (int n = lane; n < N; n += 64) {
        sR[wave * N + n] = 0.0f; sdA[wave * N + n] = 0.0f; sA[wave * N + n] = A_row[(long)n * p.A_ns];
    }
    float dD_acc = 0.0f, dbias_acc = 0.0f;

    const int n_chunks = (L + 2047) >> 11;
    for (int c = n_chunks - 1; c >= 0; --c) {
        const int cl0 = c << 11;
        const int rem = L - cl0;
        const int ntc = (rem >= 2048) ? TPC : (rem + G::TILE - 1) / G::TILE;

        // ================= phase F: state at the start of every tile of this chunk
        for (int n = lane; n < N; n += 64)
            sX0[(wave * TPC + 0) * N + n] = (c > 0 && x_row) ? x_row[((long)(c - 1) * N + n) * 2 + 1] : 0.0f;
        for (int j = 0; j + 1 < ntc; ++j) {
            const int l0 = cl0 + j * G::TILE;
            const int lbase = l0 + lane * T;
            float dl[T], dlu[T];
            {
                float uv[T], dv[T];
                load_items<io_t, T>(u_row, lbase, L, vec, uv);
                load_items<io_t, T>(d_row, lbase, L, vec, dv);
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    float d = dv[k] + bias;
                    if (p.softplus) { float sg; d = softplus_ref(d, sg); }
                    d = (lbase + k < L) ? d : 0.0f;
                    dl[k] = d;
                    dlu[k] = d * uv[k];
                }
            }
            float dsum = 0.0f;
#pragma unroll
            for (int k = 0; k < T; ++k) dsum += dl[k];
            for (int nb0 = 0; nb0 < N; nb0 += NB) {
                __syncthreads();
                for (int idx = tid; idx < NB * (G::TILE / 4); idx += blockDim.x) {
                    const int nn = idx / (G::TILE / 4);
                    const int l4 = (idx - nn * (G::TILE / 4)) * 4;
                    const int n = nb0 + nn;
                    const int l = l0 + l4;
                    float bv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (n < N && l < L) load4<io_t>(Bg + (long)n * p.B_ns + l, vec, L - l, bv);
                    *reinterpret_cast<float4*>(sB + nn * G::ROW + (l4 / T) * G::LSTR + (l4 % T)) =
                        make_float4(bv[0], bv[1], bv[2], bv[3]);
                }
                __syncthreads();
                const int nend = (N - nb0 < NB) ? (N - nb0) : NB;
#pragma unroll 1
                for (int nn = 0; nn < nend; ++nn) {
                    const int n = nb0 + nn;
                    const float A2 = sA[wave * N + n] * kLog2e;
                    const float4* __restrict__ pB = reinterpret_cast<const float4*>(sB + nn * G::ROW + lane * G::LSTR);
                    float xa = 0.0f;
#pragma unroll
                    for (int qq = 0; qq < T / 4; ++qq) {
                        const float4 bv = pB[qq];
                        const float bq[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int k = 4 * qq + jj;
                            xa = fmaf(fast_exp2(dl[k] * A2), xa, dlu[k] * bq[jj]);
                        }
                    }
                    float pa = fast_exp2(A2 * dsum);
                    wave_scan_inclusive(pa, xa);
                    if (lane == 63) {
                        const float x0 = sX0[(wave * TPC + j) * N + n];
                        sX0[(wave * TPC + j + 1) * N + n] = fmaf(pa, x0, xa);
                    }
                }
            }
        }

        // ================= phase R: tiles of this chunk, last to first
        for (int j = ntc - 1; j >= 0; --j) {
            const int l0 = cl0 + j * G::TILE;
            const int lbase = l0 + lane * T;
            float dl[T], dlu[T], gg[T], sdxB[T], sAx[T];
            {
                float dv[T], uu[T];
                load_items<io_t, T>(u_row, lbase, L, vec, uu);
                load_items<io_t, T>(d_row, lbase, L, vec, dv);
                load_items<io_t, T>(g_row, lbase, L, vec, gg);
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    float d = dv[k] + bias;
                    if (p.softplus) { float sg; d = softplus_ref(d, sg); }
                    d = (lbase + k < L) ? d : 0.0f;
                    dl[k] = d;
                    dlu[k] = d * uu[k];
                    sdxB[k] = 0.0f;
                    sAx[k] = 0.0f;
                }
            }
            float dsum = 0.0f;
#pragma unroll
            for (int k = 0; k < T; ++k) dsum += dl[k];

            for (int nb0 = 0; nb0 < N; nb0 += NB) {
                __syncthreads();
                for (int idx = tid; idx < NB * (G::TILE / 4); idx += blockDim.x) {
                    const int nn = idx / (G::TILE / 4);
                    const int l4 = (idx - nn * (G::TILE / 4)) * 4;
                    const int n = nb0 + nn;
                    const int l = l0 + l4;
                    float bv[4] = {0.f, 0.f, 0.f, 0.f}, cv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (n < N && l < L) {
                        load4<io_t>(Bg + (long)n * p.B_ns + l, vec, L - l, bv);
                        load4<io_t>(Cg + (long)n * p.C_ns + l, vec, L - l, cv);
                    }
                    const int off = nn * G::ROW + (l4 / T) * G::LSTR + (l4 % T);
                    *reinterpret_cast<float4*>(sB + off) = make_float4(bv[0], bv[1], bv[2], bv[3]);
                    *reinterpret_cast<float4*>(sC + off) = make_float4(cv[0], cv[1], cv[2], cv[3]);
                }
                __syncthreads();

                const int nend = (N - nb0 < NB) ? (N - nb0) : NB;
#pragma unroll 1
                for (int nn = 0; nn < nend; ++nn) {
                    const int n = nb0 + nn;
                    const float An = sA[wave * N + n];
                    const float A2 = An * kLog2e;
                    const float4* __restrict__ pB = reinterpret_cast<const float4*>(sB + nn * G::ROW + lane * G::LSTR);
                    const float4* __restrict__ pC = reinterpret_cast<const float4*>(sC + nn * G::ROW + lane * G::LSTR);
                    float a[T], bb[T], xs[T], gc[T];
                    // ---- forward replay: x at every element
                    float xa = 0.0f;
#pragma unroll
                    for (int qq = 0; qq < T / 4; ++qq) {
                        const float4 bv = pB[qq];
                        const float4 cv = pC[qq];
                        const float bq[4] = {bv.x, bv.y, bv.z, bv.w};
                        const float cq[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int k = 4 * qq + jj;
                            a[k] = fast_exp2(dl[k] * A2);
                            bb[k] = dlu[k] * bq[jj];
                            gc[k] = gg[k] * cq[jj];
                            xa = fmaf(a[k], xa, bb[k]);
                        }
                    }
                    const float plane = fast_exp2(A2 * dsum);   // this lane's decay product
                    float pa = plane;
                    wave_scan_inclusive(pa, xa);
                    const float pe = wave_prev_lane(pa, 1.0f);
                    const float xe = wave_prev_lane(xa, 0.0f);
                    float x = fmaf(pe, sX0[(wave * TPC + j) * N + n], xe);
#pragma unroll
                    for (int k = 0; k < T; ++k) { x = fmaf(a[k], x, bb[k]); xs[k] = x; }

                    // ---- reverse: e_k = a_k * dx_k,  dx_k = g_k C_k + e_{k+1}
                    float e = 0.0f;
#pragma unroll
                    for (int k = T - 1; k >= 0; --k) e = a[k] * (gc[k] + e);
                    float pr = plane;
                    wave_scan_inclusive_rev(pr, e);              // suffix over lanes >= this one
                    const float pn = wave_next_lane(pr, 1.0f);
                    const float en = wave_next_lane(e, 0.0f);
                    e = fmaf(pn, sR[wave * N + n], en);          // e entering from the right
                    float dAp = 0.0f;
                    // Reverse replay.  Each lane's per-row terms of dB[n, l] / dC[n, l] go straight to
                    // this wave's LDS slab (natural l order), four at a time, to keep registers free.
                    // Then the workgroup's rows are reduced: the first 2*TILE/4 threads add the slabs
                    // in a fixed order and write ONE coalesced float4 per 4 elements -- straight into
                    // dB/dC when this workgroup owns the whole group, else into its slab of the
                    // partial workspace (summed by reduce_partials_kernel).  ds_add_f32 was measured
                    // at ~1000 cycles per wave instruction on gfx950 and global float atomics at
                    // ~0.16 TB/s; plain LDS traffic + two barriers per state is far cheaper.
                    float* __restrict__ slab = sRed + wave * 2 * G::TILE + lane * T;
#pragma unroll
                    for (int qq = T / 4 - 1; qq >= 0; --qq) {
                        const float4 bv = pB[qq];                // B again: cheaper than T live registers
                        const float bq[4] = {bv.x, bv.y, bv.z, bv.w};
                        float vb[4], vc[4];
#pragma unroll
                        for (int jj = 3; jj >= 0; --jj) {
                            const int k = 4 * qq + jj;
                            const float dx = gc[k] + e;
                            sdxB[k] = fmaf(dx, bq[jj], sdxB[k]);
                            const float t = dx * (xs[k] - bb[k]);    // dx * a_k * x_{k-1}
                            sAx[k] = fmaf(An, t, sAx[k]);
                            dAp = fmaf(dl[k], t, dAp);
                            vb[jj] = dx * dlu[k];                    // this row's term of dB[n, l]
                            vc[jj] = gg[k] * xs[k];                  // this row's term of dC[n, l]
                            e = a[k] * dx;
                        }
                        *reinterpret_cast<float4*>(slab + 4 * qq) = make_float4(vb[0], vb[1], vb[2], vb[3]);
                        *reinterpret_cast<float4*>(slab + G::TILE + 4 * qq) = make_float4(vc[0], vc[1], vc[2], vc[3]);
                    }
                    dAp = wave_sum(dAp);
                    if (lane == 0) {
                        sR[wave * N + n] = e;                    // a*dx of the tile's first element
                        sdA[wave * N + n] += dAp;
                    }
                    __syncthreads();
                    for (int t4 = tid; t4 < 2 * G::TILE / 4; t4 += blockDim.x) {
                        const int c = t4 / (G::TILE / 4);           // 0: dB, 1: dC
                        const int l4 = (t4 - c * (G::TILE / 4)) * 4;
                        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                        for (int w = 0; w < nwaves; ++w) {
                            const float4 v = *reinterpret_cast<const float4*>(sRed + w * 2 * G::TILE + c * G::TILE + l4);
                            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                        }
                        const int l = l0 + l4;
                        if (l < L) {
                            float* __restrict__ dst;
                            bool vst;
                            if (q.P == 1) {
                                dst = (c == 0 ? dBg + (long)n * q.dB_ns : dCg + (long)n * q.dC_ns) + l;
                                vst = q.out_vec_ok != 0;
                            } else {
                                dst = (c == 0 ? q.ws_dB : q.ws_dC) + ws_slab + (long)n * L + l;
                                vst = (L & 3) == 0;
                            }
                            if (vst && l + 3 < L) {
                                *reinterpret_cast<float4*>(dst) = acc;
                            } else {
                                const float av[4] = {acc.x, acc.y, acc.z, acc.w};
                                for (int i = 0; i < 4; ++i) if (l + i < L) dst[i] = av[i];
                            }
                        }
                    }
                    __syncthreads();
                }
            }
            // ---- per-element results
            float duv[T], ddv[T], dv2[T], uu[T];
            // softplus' = sigmoid(raw) and the u factors: re-read delta and u (L2-resident) instead
            // of holding 2T registers across the whole state loop
            load_items<io_t, T>(d_row, lbase, L, vec, dv2);
            load_items<io_t, T>(u_row, lbase, L, vec, uu);
#pragma unroll
            for (int k = 0; k < T; ++k) {
                duv[k] = fmaf(Dd, gg[k], dl[k] * sdxB[k]);
                float dd = fmaf(uu[k], sdxB[k], sAx[k]);
                if (p.softplus) { float sg; (void)softplus_ref(dv2[k] + bias, sg); dd *= sg; }
                ddv[k] = dd;
                if (lbase + k < L) { dD_acc = fmaf(gg[k], uu[k], dD_acc); dbias_acc += dd; }
            }
            store_items<io_t, T>(du_row, lbase, L, vec, duv);
            store_items<io_t, T>(dd_row, lbase, L, vec, ddv);
        }
    }

    dD_acc = wave_sum(dD_acc);
    dbias_acc = wave_sum(dbias_acc);
    if (lane == 0) {
        if (q.dD) atomicAdd(q.dD + r, dD_acc);
        if (q.dbias) atomicAdd(q.dbias + r, dbias_acc);
    }
    for (int n = lane; n < N; n += 64)
        atomicAdd(q.dA + (long)r * q.dA_ds + (long)n * q.dA_ns, sdA[wave * N + n]);
}

// out[b, g, n, l] = sum_p ws[p][b][g][n][l]   (deterministic order; 4 elements per thread)
__global__ void __launch_bounds__(256)
reduce_partials_kernel(const float* __restrict__ wsB, const float* __restrict__ wsC, float* __restrict__ dB,
                       float* __restrict__ dC, int P, int batch, int G, int N, int L, long dB_bs, long dB_gs,
                       long dB_ns, long dC_bs, long dC_gs, long dC_ns) {
    const long per = (long)batch * G * N * L;
    const long nvec = (per + 3) / 4;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.x * blockDim.x) {
        const long e0 = v * 4;
        float accB[4] = {0.f, 0.f, 0.f, 0.f}, accC[4] = {0.f, 0.f, 0.f, 0.f};
        const bool full = (e0 + 3 < per) && ((L & 3) == 0);
        if (full) {
            for (int pp = 0; pp < P; ++pp) {
                const float4 tb = *reinterpret_cast<const float4*>(wsB + pp * per + e0);
                const float4 tc = *reinterpret_cast<const float4*>(wsC + pp * per + e0);
                accB[0] += tb.x; accB[1] += tb.y; accB[2] += tb.z; accB[3] += tb.w;
                accC[0] += tc.x; accC[1] += tc.y; accC[2] += tc.z; accC[3] += tc.w;
            }
        } else {
            for (int pp = 0; pp < P; ++pp)
                for (int i = 0; i < 4; ++i)
                    if (e0 + i < per) { accB[i] += wsB[pp * per + e0 + i]; accC[i] += wsC[pp * per + e0 + i]; }
        }
        for (int i = 0; i < 4; ++i) {
            const long e = e0 + i;
            if (e >= per) break;
            const int l = (int)(e % L);
            const long t = e / L;
            const int n = (int)(t % N);
            const long t2 = t / N;
            const int g = (int)(t2 % G);
            const int b = (int)(t2 / G);
            dB[b * dB_bs + g * dB_gs + n * dB_ns + l] = accB[i];
            dC[b * dC_bs + g * dC_gs + n * dC_ns + l] = accC[i];
        }
    }
}

template <typename io_t, int T>
static hipError_t launch_bwd_t(const BwdArgs& a, int nwaves, hipStream_t stream) {
    const size_t lds = bwd_lds_bytes(T, nwaves, a.f.N);
    const int grid = a.f.rowblocks * a.f.batch;
    auto kern = scan_bwd_kernel<io_t, T>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nwaves * 64), lds, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || a.P == 1) return e;
    const long per = (long)a.f.batch * a.f.G * a.f.N * a.f.L;
    long blocks = (per / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a.ws_dB, a.ws_dC, a.dB,
                       a.dC, a.P, a.f.batch, a.f.G, a.f.N, a.f.L, a.dB_bs, a.dB_gs, a.dB_ns, a.dC_bs, a.dC_gs,
                       a.dC_ns);
    return hipGetLastError();
}

template <typename io_t>
static hipError_t launch_bwd_io(const BwdArgs& a, int T, int nwaves, hipStream_t stream) {
    switch (T) {
        case 4: return launch_bwd_t<io_t, 4>(a, nwaves, stream);
        case 8: return launch_bwd_t<io_t, 8>(a, nwaves, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_scan_bwd(const BwdArgs& a, int dtype, int T, int nwaves, hipStream_t stream) {
    switch (dtype) {
        case 0: return launch_bwd_io<float>(a, T, nwaves, stream);
        case 1: return launch_bwd_io<f16_t>(a, T, nwaves, stream);
        case 2: return launch_bwd_io<bf16_t>(a, T, nwaves, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace sigma
