// scan_bwd.hip -- selective-scan backward for gfx950 (MI355X), wave64.
//
// Replaces selective_scan_bwd_kernel + reverse_scan.cuh (reference:
// models/encoders/selective_scan/csrc/selective_scan/selective_scan_bwd_kernel.cuh:66-308,
// reverse_scan.cuh:18-401).  Mathematics: SURVEY.md App. E.2.
//
// Machine mapping (differs from the reference on purpose):
//   * one wave per channel row, R rows of one (batch, group) per workgroup; lane i owns T
//     consecutive elements of the current tile (T = 10 / 5 / 4: tiles of 640 / 320 / 256);
//   * the sequence is walked from the end in spans of 1280 elements (the pitch of the state
//     checkpoints the forward kernel leaves in x).  Inside a span a cheap forward sweep (fold +
//     wave scan only) rebuilds the state at every tile start, then the tiles are processed last
//     to first: forward replay (x kept in registers) + reverse affine scan of e = a * dx
//     ("what flows to the element on the left");
//   * the reverse scan across lanes is a DPP row_shl scan + uniform row-head fix-up;
//   * B/C tiles are streamed into LDS with global_load_lds, double buffered over blocks of NB
//     states (the next block lands while the current one is computed);
//   * dB / dC are reduced over the workgroup's rows through per-wave LDS slabs: every thread then
//     sums one 16-byte column of the R slabs in a fixed order and writes ONE coalesced partial
//     per (n, l) -- straight into dB/dC when the workgroup owns the whole group, else into its
//     slab of a caller-provided workspace that reduce_partials_kernel sums in a fixed order.
//     Deterministic, and no device-scope float atomics (the reference issues one global
//     atomicAdd per row: 192 per element at stage 0).  The two barriers per state only order
//     LDS traffic (s_waitcnt lgkmcnt + s_barrier), so they do not drain the B/C stream;
//   * reversed groups read/write every sequence operand at L-1-l (see scan_fwd.hip).
#include "scan_device.h"
#include "scan_launch.h"

#include <atomic>

namespace sigma {

namespace {

// Register path of the B/C staging (16-bit IO types, unaligned tensors): rows of states [n0, n0+nbn)
// of ONE tile into dst laid out [arr][NB][TILE]; the f32 / aligned case uses StagePlan (scan_device.h).
template <typename io_t, int T>
__device__ __forceinline__ void stage_tile(float* __restrict__ dst, const io_t* __restrict__ Bg,
                                           const io_t* __restrict__ Cg, long B_ns, long C_ns, int n0, int nbn, int NB,
                                           int tile, int L, bool rev, bool vec, bool with_c) {
    constexpr int TILE = 64 * T;
    constexpr int CPR = TILE / 4;
    const int total = (with_c ? 2 : 1) * NB * CPR;
    const int l0 = tile * TILE;
    for (int ci = threadIdx.x; ci < total; ci += blockDim.x) {
        const int row = ci / CPR;                      // arr * NB + nn
        const int c4 = (ci - row * CPR) * 4;
        const int arr = row / NB;
        const int nn = row - arr * NB;
        const int m = rev ? (L - l0 - TILE + c4) : (l0 + c4);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (nn < nbn && m < L && m + 4 > 0) {
            const io_t* __restrict__ srow = arr == 0 ? Bg + (long)(n0 + nn) * B_ns : Cg + (long)(n0 + nn) * C_ns;
            load4_guard<io_t>(srow, m, L, vec, v);
        }
        *reinterpret_cast<float4*>(dst + (long)ci * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace

template <typename io_t, int T, bool GLDS, bool REV>
__device__ __forceinline__ void scan_bwd_body(const BwdArgs& q, float* smem, int b, int row0, int g) {
    constexpr int TILE = 64 * T;
    constexpr int TPSMAX = kCkptPitch / TILE;         // LDS sizing: tiles per span at the default pitch
    const int TPS = q.f.ckpt_pitch / TILE;            // tiles per checkpoint span (1 with the fine pitch at T = 10)
    const int pitch = q.f.ckpt_pitch;
    constexpr int VW = vec_width<T>::value;
    const FwdArgs& p = q.f;
    const int R = blockDim.x >> 6;
    const int N = p.N, L = p.L, NB = p.NB;
    const int bufsz = 2 * NB * TILE;
    float* sBC = smem;                                // [2][2][NB][TILE]
    float* sRed = sBC + 2 * bufsz;                    // [R][2][TILE] per-row dB/dC terms of one state
    float* sX0 = sRed + (q.slab2 ? 2 : 1) * R * 2 * TILE;   // [R][TPS][N] state at tile start
    float* sRv = sX0 + R * TPSMAX * N;                // [R][N] reverse carry a*dx of the tile to the right
    float* sdA = sRv + R * N;                         // [R][N]
    float* sA = sdA + R * N;                          // [R][N] A[r, n]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = row0 + wave;
    const bool vec = p.vec_ok != 0;
    const int ur = r - ((g - (g >> p.u_gshift)) * p.rows_per_group);   // same row of group g >> u_gshift

    const io_t* __restrict__ u_row = reinterpret_cast<const io_t*>(p.u) + (long)b * p.u_bs + (long)ur * p.u_ds;
    const io_t* __restrict__ d_row = reinterpret_cast<const io_t*>(p.delta) + (long)b * p.dt_bs + (long)r * p.dt_ds;
    const int gr = r - ((g - (g >> q.g_gshift)) * p.rows_per_group);
    const io_t* __restrict__ g_row = reinterpret_cast<const io_t*>(q.dout) + (long)b * q.g_bs + (long)gr * q.g_ds;
    io_t* __restrict__ du_row = reinterpret_cast<io_t*>(q.du) + (long)b * q.du_bs + (long)r * q.du_ds;
    io_t* __restrict__ dd_row = reinterpret_cast<io_t*>(q.ddelta) + (long)b * q.dd_bs + (long)r * q.dd_ds;
    const io_t* __restrict__ Bg = reinterpret_cast<const io_t*>(p.B) + (long)b * p.B_bs + (long)g * p.B_gs;
    const io_t* __restrict__ Cg = reinterpret_cast<const io_t*>(p.C) + (long)b * p.C_bs + (long)g * p.C_gs;
    float* __restrict__ dBg = q.dB + (long)b * q.dB_bs + (long)g * q.dB_gs;
    float* __restrict__ dCg = q.dC + (long)b * q.dC_bs + (long)g * q.dC_gs;
    const int pr = param_row(r, g, p.rows_per_group, p.pswap);
    const float* __restrict__ A_row = p.A + (long)pr * p.A_ds;
    const float bias = p.bias ? p.bias[pr] : 0.0f;
    const float Dd = p.D ? p.D[pr] : 0.0f;
    const float* __restrict__ x_row = p.x ? p.x + ((long)b * p.dim + r) * p.x_rs : nullptr;

    const long ws_slab = (q.P > 1)
        ? ((((long)((row0 - g * p.rows_per_group) / R) * p.batch + b) * p.G + g) * (long)N * L) : 0;
    for (int n = lane; n < N; n += 64) {
        sRv[wave * N + n] = 0.0f; sdA[wave * N + n] = 0.0f; sA[wave * N + n] = A_row[(long)n * p.A_ns];
    }
    // never multiply uninitialised LDS bits (stale/NaN) into the padding of the last tile
    for (int i = tid; i < 2 * bufsz / 4; i += blockDim.x) reinterpret_cast<float4*>(sBC)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    float dD_acc = 0.0f, dbias_acc = 0.0f;

    const int nsb = (N + NB - 1) / NB;
    const int nspans = (L + pitch - 1) / pitch;
    auto tiles_in_span = [&](int s) {
        const int rem = L - s * pitch;
        return rem >= pitch ? TPS : (rem + TILE - 1) / TILE;
    };
    StagePlan<T, REV> plan;
    if constexpr (GLDS) plan.init(NB, 1, L);
    auto stage = [&](int step, int tile, int sb, bool with_c) {
        const int n0 = sb * NB;
        const int nbn = (N - n0 < NB) ? (N - n0) : NB;
        float* dst = sBC + (step & 1) * bufsz;
        if constexpr (GLDS) {
            plan.issue(dst, reinterpret_cast<const float*>(Bg), reinterpret_cast<const float*>(Cg), (int)p.B_ns, (int)p.C_ns,
                       n0, nbn, tile, L, NB * TILE, with_c);
        } else {
            stage_tile<io_t, T>(dst, Bg, Cg, p.B_ns, p.C_ns, n0, nbn, NB, tile, L, REV, vec, with_c);
        }
    };

    int step = 0;                                      // staging buffer parity
    {
        const int nt = tiles_in_span(nspans - 1);      // first step of the whole walk
        stage(0, (nspans - 1) * TPS + (nt > 1 ? 0 : nt - 1), 0, nt <= 1);
        __syncthreads();
    }

    for (int s = nspans - 1; s >= 0; --s) {
        const int ntc = tiles_in_span(s);
        const int tile_base = s * TPS;

        // ================= phase F: state at the start of every tile of this span
        for (int n = lane; n < N; n += 64) {
            float x0 = 0.0f;
            if (s > 0 && x_row) x0 = x_row[(long)(s - 1) * N + n];
            sX0[(wave * TPS + 0) * N + n] = x0;
        }
        for (int j = 0; j + 1 < ntc; ++j) {
            const int l0 = (tile_base + j) * TILE;
            const int lbase = l0 + lane * T;
            float dl[T], dlu[T];
            {
                float uv[T], dv[T];
                load_items<io_t, T, REV>(u_row, lbase, L, vec, uv);
                load_items<io_t, T, REV>(d_row, lbase, L, vec, dv);
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
            for (int sb = 0; sb < nsb; ++sb) {
                const float* cur = sBC + (step & 1) * bufsz;
                if (sb + 1 < nsb) stage(step + 1, tile_base + j, sb + 1, false);
                else if (j + 2 < ntc) stage(step + 1, tile_base + j + 1, 0, false);
                else stage(step + 1, tile_base + ntc - 1, 0, true);          // first R step
                const int n0 = sb * NB;
                const int nend = (N - n0 < NB) ? (N - n0) : NB;
#pragma unroll 1
                for (int nn = 0; nn < nend; ++nn) {
                    const int n = n0 + nn;
                    const float A2 = sA[wave * N + n] * kLog2e;
                    float xa = 0.0f;
#pragma unroll
                    for (int qq = 0; qq < T / VW; ++qq) {
                        float bq[VW];
                        lds_read_chunk<T, REV>(cur + nn * TILE, lane, qq, bq);
#pragma unroll
                        for (int jj = 0; jj < VW; ++jj) {
                            const int k = VW * qq + jj;
                            xa = fmaf(fast_exp2(dl[k] * A2), xa, dlu[k] * bq[jj]);
                        }
                    }
                    float sa = A2 * dsum;
                    wave_scan_inclusive(sa, xa);
                    if (lane == 63) {
                        const float x0 = sX0[(wave * TPS + j) * N + n];
                        sX0[(wave * TPS + j + 1) * N + n] = fmaf(fast_exp2(sa), x0, xa);
                    }
                }
                __syncthreads();
                ++step;
            }
        }

        // ================= phase R: tiles of this span, last to first
        for (int j = ntc - 1; j >= 0; --j) {
            const int l0 = (tile_base + j) * TILE;
            const int lbase = l0 + lane * T;
            float dl[T], dlu[T], gg[T], sdxB[T], sAx[T];
            {
                float dv[T], uu[T];
                load_items<io_t, T, REV>(u_row, lbase, L, vec, uu);
                load_items<io_t, T, REV>(d_row, lbase, L, vec, dv);
                load_items<io_t, T, REV>(g_row, lbase, L, vec, gg);
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

            // image (memory-order) window of this tile for the dB/dC outputs
            const int img0 = REV ? (L - l0 - TILE) : l0;

            for (int sb = 0; sb < nsb; ++sb) {
                const float* cur = sBC + (step & 1) * bufsz;
                if (sb + 1 < nsb) {
                    stage(step + 1, tile_base + j, sb + 1, true);
                } else if (j > 0) {
                    stage(step + 1, tile_base + j - 1, 0, true);
                } else if (s > 0) {                                          // next span: TPS tiles, F first
                    stage(step + 1, (s - 1) * TPS, 0, TPS <= 1);
                }
                const int n0 = sb * NB;
                const int nend = (N - n0 < NB) ? (N - n0) : NB;
#pragma unroll 1
                for (int nn = 0; nn < nend; ++nn) {
                    const int n = n0 + nn;
                    const float An = sA[wave * N + n];
                    const float A2 = An * kLog2e;
                    const float* tB = cur + nn * TILE;
                    const float* tC = tB + NB * TILE;
                    float a[T], xs[T], gc[T];
                    // ---- forward replay: x at every element (xs first holds b, then x)
                    float xa = 0.0f;
#pragma unroll
                    for (int qq = 0; qq < T / VW; ++qq) {
                        float bq[VW], cq[VW];
                        lds_read_chunk<T, REV>(tB, lane, qq, bq);
                        lds_read_chunk<T, REV>(tC, lane, qq, cq);
#pragma unroll
                        for (int jj = 0; jj < VW; ++jj) {
                            const int k = VW * qq + jj;
                            a[k] = fast_exp2(dl[k] * A2);
                            xs[k] = dlu[k] * bq[jj];
                            gc[k] = gg[k] * cq[jj];
                            xa = fmaf(a[k], xa, xs[k]);
                        }
                    }
                    const float slane = A2 * dsum;              // log2 of this lane's decay product
                    float sa = slane;
                    wave_scan_inclusive(sa, xa);
                    const float pe = fast_exp2(wave_prev_lane(sa, 0.0f));
                    const float xe = wave_prev_lane(xa, 0.0f);
                    const float xstart = fmaf(pe, sX0[(wave * TPS + j) * N + n], xe);   // state entering the lane
                    {
                        float x = xstart;
#pragma unroll
                        for (int k = 0; k < T; ++k) { x = fmaf(a[k], x, xs[k]); xs[k] = x; }
                    }

                    // ---- reverse: e_k = a_k * dx_k,  dx_k = g_k C_k + e_{k+1}
                    float e = 0.0f;
#pragma unroll
                    for (int k = T - 1; k >= 0; --k) e = a[k] * (gc[k] + e);
                    float sr = slane;
                    wave_scan_inclusive_rev(sr, e);              // suffix over lanes >= this one
                    const float pn = fast_exp2(wave_next_lane(sr, 0.0f));
                    const float en = wave_next_lane(e, 0.0f);
                    e = fmaf(pn, sRv[wave * N + n], en);         // e entering from the right
                    float dAp = 0.0f;
                    // Reverse replay.  The per-row terms of dB[n, l] / dC[n, l] go straight to this
                    // wave's LDS slab (position order), VW at a time, to keep registers free; then
                    // all threads add the R slabs column-wise.
                    float* sRedN = sRed + ((q.slab2 && (n & 1)) ? R * 2 * TILE : 0);
                    if (!q.slab2) lds_barrier();                 // slab free (previous state reduced)
                    {
                        float* __restrict__ slab = sRedN + wave * 2 * TILE + lane * T;
#pragma unroll
                        for (int qq = T / VW - 1; qq >= 0; --qq) {
                            float bq[VW], vb[VW], vc[VW];
                            lds_read_chunk<T, REV>(tB, lane, qq, bq);    // B again: cheaper than T live registers
#pragma unroll
                            for (int jj = VW - 1; jj >= 0; --jj) {
                                const int k = VW * qq + jj;
                                const float dx = gc[k] + e;
                                sdxB[k] = fmaf(dx, bq[jj], sdxB[k]);
                                const float ax = a[k] * (k > 0 ? xs[k > 0 ? k - 1 : 0] : xstart);   // a_k * x_{k-1}
                                const float t = dx * ax;
                                sAx[k] = fmaf(An, t, sAx[k]);
                                dAp = fmaf(dl[k], t, dAp);
                                vb[jj] = dx * dlu[k];                    // this row's term of dB[n, l]
                                vc[jj] = gg[k] * xs[k];                  // this row's term of dC[n, l]
                                e = a[k] * dx;
                            }
                            if constexpr (VW == 4) {
                                *reinterpret_cast<float4*>(slab + 4 * qq) = make_float4(vb[0], vb[1], vb[2], vb[3]);
                                *reinterpret_cast<float4*>(slab + TILE + 4 * qq) = make_float4(vc[0], vc[1], vc[2], vc[3]);
                            } else if constexpr (VW == 2) {
                                *reinterpret_cast<float2*>(slab + 2 * qq) = make_float2(vb[0], vb[1]);
                                *reinterpret_cast<float2*>(slab + TILE + 2 * qq) = make_float2(vc[0], vc[1]);
                            } else {
                                slab[qq] = vb[0];
                                slab[TILE + qq] = vc[0];
                            }
                        }
                    }
                    dAp = wave_sum(dAp);
                    if (lane == 0) {
                        sRv[wave * N + n] = e;                   // a*dx of the tile's first element
                        sdA[wave * N + n] += dAp;
                    }
                    lds_barrier();                               // slabs complete
                    for (int t4 = tid; t4 < 2 * TILE / 4; t4 += blockDim.x) {
                        const int c = t4 / (TILE / 4);           // 0: dB, 1: dC
                        const int q4 = (t4 - c * (TILE / 4)) * 4;   // image offset (memory order)
                        const int p4 = REV ? (TILE - 4 - q4) : q4;  // lowest slab position of the 4
                        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                        for (int w = 0; w < R; ++w) {
                            const float4 v = *reinterpret_cast<const float4*>(sRedN + w * 2 * TILE + c * TILE + p4);
                            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                        }
                        if (REV) { const float4 t = acc; acc = make_float4(t.w, t.z, t.y, t.x); }
                        const int m = img0 + q4;
                        if (m < L && m + 4 > 0) {
                            float* __restrict__ dst;
                            bool vst;
                            if (q.P == 1) {
                                dst = (c == 0 ? dBg + (long)n * q.dB_ns : dCg + (long)n * q.dC_ns) + m;
                                vst = q.out_vec_ok != 0 && (m & 3) == 0;
                            } else {
                                dst = (c == 0 ? q.ws_dB : q.ws_dC) + ws_slab + (long)n * L + m;
                                vst = ((L | m) & 3) == 0;
                            }
                            if (vst && m >= 0 && m + 4 <= L) {
                                *reinterpret_cast<float4*>(dst) = acc;
                            } else {
                                const float av[4] = {acc.x, acc.y, acc.z, acc.w};
                                for (int i = 0; i < 4; ++i) if (m + i >= 0 && m + i < L) dst[i] = av[i];
                            }
                        }
                    }
                }
                __syncthreads();                                 // next B/C block landed
                ++step;
            }
            // ---- per-element results
            float duv[T], ddv[T];
            {
                // softplus' = sigmoid(raw) and the u factors: re-read delta and u (L2-resident) instead
                // of holding 2T registers across the whole state loop
                float dv2[T], uu[T];
                load_items<io_t, T, REV>(d_row, lbase, L, vec, dv2);
                load_items<io_t, T, REV>(u_row, lbase, L, vec, uu);
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    duv[k] = fmaf(Dd, gg[k], dl[k] * sdxB[k]);
                    float dd = fmaf(uu[k], sdxB[k], sAx[k]);
                    if (p.softplus) { float sg; (void)softplus_ref(dv2[k] + bias, sg); dd *= sg; }
                    ddv[k] = dd;
                    if (lbase + k < L) { dD_acc = fmaf(gg[k], uu[k], dD_acc); dbias_acc += dd; }
                }
            }
            store_items<io_t, T, REV>(du_row, lbase, L, vec, duv);
            store_items<io_t, T, REV>(dd_row, lbase, L, vec, ddv);
        }
    }

    dD_acc = wave_sum(dD_acc);
    dbias_acc = wave_sum(dbias_acc);
    if (lane == 0) {
        if (q.dD) atomicAdd(q.dD + pr, dD_acc);
        if (q.dbias) atomicAdd(q.dbias + pr, dbias_acc);
    }
    for (int n = lane; n < N; n += 64)
        atomicAdd(q.dA + (long)pr * q.dA_ds + (long)n * q.dA_ns, sdA[wave * N + n]);
}

// T = 10 keeps ~90 values per lane live (5 per-element accumulators + a, x, g*C per state): it
// needs ~150 VGPRs, i.e. 3 waves per SIMD = 12 rows per workgroup; T = 4 / 5 fit 16.
template <int T> struct bwd_max_waves { static constexpr int value = (T >= 10) ? kBwdMaxWavesT10 : 16; };

template <typename io_t, int T, bool GLDS>
__global__ void __launch_bounds__(64 * bwd_max_waves<T>::value)
scan_bwd_kernel(const BwdArgs q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int R = blockDim.x >> 6;
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int b = lb / q.f.rowblocks;
    const int rb = lb - b * q.f.rowblocks;
    const int row0 = rb * R;
    const int g = row0 / q.f.rows_per_group;
    if ((q.f.rev_mask >> g) & 1u) scan_bwd_body<io_t, T, GLDS, true>(q, smem, b, row0, g);
    else scan_bwd_body<io_t, T, GLDS, false>(q, smem, b, row0, g);
}

// out[b, g, n, l] = sum_p ws[p][b][g][n][l]   (deterministic order; 4 elements per thread).
// L % 4 == 0 (VEC): the four elements share (b, g, n), so the index is decomposed once per float4 -- in 32-bit
// arithmetic, the 64-bit divisions of the first version cost more than the memory traffic (63 us for 98 MB at P = 4)
// -- and the sum leaves as one 16-byte store when the destination rows are 16-byte aligned (OVEC).
template <bool VEC, bool OVEC>
__global__ void __launch_bounds__(256)
reduce_partials_kernel(const float* __restrict__ wsB, const float* __restrict__ wsC, float* __restrict__ dB,
                       float* __restrict__ dC, int P, int batch, int G, int N, int L, long dB_bs, long dB_gs,
                       long dB_ns, long dC_bs, long dC_gs, long dC_ns) {
    const long per = (long)batch * G * N * L;
    const long nvec = (per + 3) / 4;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.x * blockDim.x) {
        const long e0 = v * 4;
        float accB[4] = {0.f, 0.f, 0.f, 0.f}, accC[4] = {0.f, 0.f, 0.f, 0.f};
        if (VEC) {
            for (int pp = 0; pp < P; ++pp) {
                const float4 tb = *reinterpret_cast<const float4*>(wsB + pp * per + e0);
                const float4 tc = *reinterpret_cast<const float4*>(wsC + pp * per + e0);
                accB[0] += tb.x; accB[1] += tb.y; accB[2] += tb.z; accB[3] += tb.w;
                accC[0] += tc.x; accC[1] += tc.y; accC[2] += tc.z; accC[3] += tc.w;
            }
            const unsigned row = (unsigned)(e0 / (unsigned)L);          // (b * G + g) * N + n; host: batch*G*N < 2^31
            const unsigned l = (unsigned)(e0 - (long)row * L);
            const unsigned n = row % (unsigned)N;
            const unsigned bg = row / (unsigned)N;
            const unsigned g = bg % (unsigned)G;
            const unsigned bb = bg / (unsigned)G;
            float* __restrict__ pb = dB + bb * dB_bs + g * dB_gs + n * dB_ns + l;
            float* __restrict__ pc = dC + bb * dC_bs + g * dC_gs + n * dC_ns + l;
            if (OVEC) {
                *reinterpret_cast<float4*>(pb) = make_float4(accB[0], accB[1], accB[2], accB[3]);
                *reinterpret_cast<float4*>(pc) = make_float4(accC[0], accC[1], accC[2], accC[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) { pb[i] = accB[i]; pc[i] = accC[i]; }
            }
        } else {
            for (int pp = 0; pp < P; ++pp)
                for (int i = 0; i < 4; ++i)
                    if (e0 + i < per) { accB[i] += wsB[pp * per + e0 + i]; accC[i] += wsC[pp * per + e0 + i]; }
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
}

hipError_t launch_reduce_partials(const BwdArgs& a, hipStream_t stream) {
    const long per = (long)a.f.batch * a.f.G * a.f.N * a.f.L;
    long blocks = (per / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    const bool vec = (a.f.L & 3) == 0 && (long)a.f.batch * a.f.G * a.f.N < (1L << 31);
    const bool ovec = vec && a.out_vec_ok != 0;
#define SIGMA_RP(V, O) hipLaunchKernelGGL((reduce_partials_kernel<V, O>), dim3((unsigned)blocks), dim3(256), 0, stream, a.ws_dB, a.ws_dC, \
                       a.dB, a.dC, a.P, a.f.batch, a.f.G, a.f.N, a.f.L, a.dB_bs, a.dB_gs, a.dB_ns, a.dC_bs, a.dC_gs, a.dC_ns)
    if (ovec) SIGMA_RP(true, true);
    else if (vec) SIGMA_RP(true, false);
    else SIGMA_RP(false, false);
#undef SIGMA_RP
    return hipGetLastError();
}

template <typename io_t, int T, bool GLDS>
static hipError_t launch_bwd_t(const BwdArgs& a, hipStream_t stream) {
    const size_t lds = bwd_lds_bytes(T, a.f.R, a.f.NB, a.f.N, a.slab2 != 0);
    const int grid = a.f.rowblocks * a.f.batch;
    auto kern = scan_bwd_kernel<io_t, T, GLDS>;
    // raise the dynamic-LDS cap once per device, kernel and size (not per launch: the call is host-expensive)
    static std::atomic<size_t> lds_cap[kMaxDevices];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDevices) dev = 0;
    if (lds > 48 * 1024 && lds > lds_cap[dev].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_cap[dev].store(lds, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(a.f.R * 64), lds, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || a.P == 1) return e;
    return launch_reduce_partials(a, stream);
}

template <typename io_t, bool GLDS>
static hipError_t launch_bwd_io(const BwdArgs& a, int T, hipStream_t stream) {
    switch (T) {
        case 4: return launch_bwd_t<io_t, 4, GLDS>(a, stream);
        case 5: return launch_bwd_t<io_t, 5, GLDS>(a, stream);
        case 10: return launch_bwd_t<io_t, 10, GLDS>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_scan_bwd(const BwdArgs& a, int dtype, int T, bool glds, hipStream_t stream) {
    switch (dtype) {
        case 0: return glds ? launch_bwd_io<float, true>(a, T, stream) : launch_bwd_io<float, false>(a, T, stream);
        case 1: return launch_bwd_io<f16_t, false>(a, T, stream);
        case 2: return launch_bwd_io<bf16_t, false>(a, T, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace sigma
