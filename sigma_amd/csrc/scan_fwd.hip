// scan_fwd.hip -- selective-scan forward for gfx950 (MI355X), wave64.
//
// Replaces selective_scan_fwd_kernel (reference:
// models/encoders/selective_scan/csrc/selective_scan/selective_scan_fwd_kernel.cuh:62-206).
// Same mathematics, different machine mapping:
//
//   * a workgroup is R channel rows of ONE (batch, group) times W consecutive sequence tiles
//     (R*W <= 16 waves); wave (wr, wt) owns row wr and tile wt of the current "super-tile" of
//     W*64*T elements and the workgroup walks the sequence super-tile by super-tile.  Many rows
//     (training batches): R = 16, W = 1.  Few rows (one image per GPU): W > 1 splits the sequence
//     inside the workgroup so that 256 CUs still get 15-16 waves each;
//   * lane i owns T consecutive elements of its tile.  Per state n: serial fold over the lane's
//     T elements (decay a and input b stay in registers), ONE DPP wave scan of the 64 lane
//     aggregates, serial replay with the true incoming state.  The lane decay product is
//     exp2(A * sum(delta)) -- one transcendental instead of a T-long product.  With W > 1 the
//     tile aggregates of a row meet in LDS (one extra barrier per state) and each wave composes
//     its predecessors' (decay, state) pairs before the replay;
//   * T is 20 / 10 / 5 / 4 (tiles of 1280 / 640 / 320 / 256): the model's sequence lengths are
//     multiples of 300 (15x20 .. 120x160), which power-of-two tiles pad by up to 40 %;
//   * the group's B/C tile is shared by the R rows through LDS: global_load_lds (async, no VGPR
//     round trip) into an unpadded memory-order image, double buffered over blocks of NB
//     states, so the next block streams in while the current one is computed and there is ONE
//     barrier per block.  u/delta of the next super-tile are prefetched into registers during
//     the last block of the current one;
//   * reversed groups (CrossScan directions 2, 3) read every sequence operand at L-1-l and write
//     out there, so no flipped copies of x / delta / B / C have to exist (vmamba.py:80-121);
//   * running state between super-tiles lives in LDS; states every ckpt_pitch (1280 / 640 / 320) elements go to x
//     (layout in include/sigma_scan.h) for the backward pass.
#include "scan_device.h"
#include "scan_launch.h"

#include <atomic>

namespace sigma {

namespace {

// Register path of the B/C staging (16-bit IO types, unaligned tensors): rows of states [n0, n0+nbn)
// for the W tiles starting at tile index tile0 into `dst` laid out [arr = B,C][NB][W][TILE] (floats,
// memory order).  The f32 / aligned case streams with global_load_lds instead (StagePlan, scan_device.h).
template <typename io_t, int T>
__device__ __forceinline__ void stage_bc(float* __restrict__ dst, const io_t* __restrict__ Bg,
                                         const io_t* __restrict__ Cg, long B_ns, long C_ns, int n0, int nbn, int NB,
                                         int W, int tile0, int L, bool rev, bool vec) {
    constexpr int TILE = 64 * T;
    constexpr int CPR = TILE / 4;                      // 16-byte chunks per (state, tile) row
    const int rows = NB * W;                           // rows per array in the LDS image
    const int total = 2 * rows * CPR;
    for (int ci = threadIdx.x; ci < total; ci += blockDim.x) {
        const int row = ci / CPR;                      // arr * rows + nn * W + w
        const int c4 = (ci - row * CPR) * 4;
        const int arr = row / rows;
        const int rr = row - arr * rows;
        const int nn = rr / W;
        const int w = rr - nn * W;
        const int l0 = (tile0 + w) * TILE;
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

// WSPLIT = false (one tile per workgroup step, W == 1): the state entering the tile is known before the
// scan, so it is folded into lane 0's in-lane fold and the wave scan runs in the multiplicative form
// (scan_device.h: no v_exp_f32 per scan step, no exclusive decay product) -- 1 + T instead of 8 + T
// transcendentals per state and tile.  WSPLIT = true keeps the log-domain scan: the tile aggregates
// (decay product, end state) of the W waves of a row have to meet in LDS before the incoming state is known.
template <typename io_t, int T, bool GLDS, bool PREFETCH, bool REV, bool WSPLIT>
__device__ __forceinline__ void scan_fwd_body(const FwdArgs& p, float* smem, int b, int row0, int g) {
    constexpr int TILE = 64 * T;
    constexpr int VW = vec_width<T>::value;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = p.N, L = p.L, W = p.W, R = p.R, NB = p.NB;
    const int wr = wave / W;
    const int wt = wave - wr * W;

    const int bufsz = 2 * NB * W * TILE;                 // floats per staging buffer
    float* sBC = smem;                                   // [2][2][NB][W][TILE]
    float2* sAgg = reinterpret_cast<float2*>(sBC + 2 * bufsz);   // [2][R][W] tile aggregates (W > 1)
    float* sA2 = reinterpret_cast<float*>(sAgg + 2 * R * W);     // [R][N]  A * log2(e)
    float* sRun = sA2 + R * N;                           // [2][R][N] state entering the super-tile

    const int r = row0 + wr;
    const bool vec = p.vec_ok != 0;
    const int ur = r - ((g - (g >> p.u_gshift)) * p.rows_per_group);   // same row of group g >> u_gshift

    const io_t* __restrict__ u_row = reinterpret_cast<const io_t*>(p.u) + (long)b * p.u_bs + (long)ur * p.u_ds;
    const io_t* __restrict__ d_row = reinterpret_cast<const io_t*>(p.delta) + (long)b * p.dt_bs + (long)r * p.dt_ds;
    io_t* __restrict__ o_row = reinterpret_cast<io_t*>(p.out) + (long)b * p.o_bs + (long)r * p.o_ds;
    const io_t* __restrict__ Bg = reinterpret_cast<const io_t*>(p.B) + (long)b * p.B_bs + (long)g * p.B_gs;
    const io_t* __restrict__ Cg = reinterpret_cast<const io_t*>(p.C) + (long)b * p.C_bs + (long)g * p.C_gs;
    const int pr = param_row(r, g, p.rows_per_group, p.pswap);
    const float bias = p.bias ? p.bias[pr] : 0.0f;
    const float Dd = p.D ? p.D[pr] : 0.0f;
    float* __restrict__ x_row = p.x ? p.x + ((long)b * p.dim + r) * p.x_rs : nullptr;

    if (wt == 0) {
        const float* __restrict__ A_row = p.A + (long)pr * p.A_ds;
        for (int n = lane; n < N; n += 64) {
            sA2[wr * N + n] = A_row[(long)n * p.A_ns] * kLog2e;
            sRun[wr * N + n] = 0.0f;
        }
    }

    const int ntiles = (L + TILE - 1) / TILE;
    const int nsuper = (ntiles + W - 1) / W;
    const int nsb = (N + NB - 1) / NB;

    // never multiply uninitialised LDS bits (stale/NaN) into the padding of the last tile
    for (int i = tid; i < 2 * bufsz / 4; i += blockDim.x) reinterpret_cast<float4*>(sBC)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // ---- B/C streaming: precomputed chunk plan + global_load_lds (f32, aligned), else registers
    StagePlan<T, REV> plan;
    if constexpr (GLDS) plan.init(NB, W, L);
    auto stage = [&](float* dst, int n0, int tile0) {
        const int nbn = (N - n0 < NB) ? (N - n0) : NB;
        if constexpr (GLDS) {
            // untracked LDS-DMA (scan_device.h: issue_async): hipcc would drain a builtin global_load_lds at the top of
            // the block it was meant to overlap with; completion = lds_dma_wait() before the block-end barrier
            plan.issue_async(dst, reinterpret_cast<const float*>(Bg), reinterpret_cast<const float*>(Cg), (int)p.B_ns,
                             (int)p.C_ns, n0, nbn, tile0, L, NB * W * TILE);
        } else {
            stage_bc<io_t, T>(dst, Bg, Cg, p.B_ns, p.C_ns, n0, nbn, NB, W, tile0, L, REV, vec);
        }
    };
    // ---- prologue: first B/C block and first u/delta segment
    stage(sBC, 0, 0);
    float uv[T], dv[T];
    load_items<io_t, T, REV>(u_row, wt * TILE + lane * T, L, vec, uv);
    load_items<io_t, T, REV>(d_row, wt * TILE + lane * T, L, vec, dv);
    if constexpr (GLDS) lds_dma_wait();
    __syncthreads();

    for (int st = 0; st < nsuper; ++st) {
        const int tile = st * W + wt;
        const int l0 = tile * TILE;
        const int lbase = l0 + lane * T;

        float dl[T], dlu[T], y[T];
#pragma unroll
        for (int k = 0; k < T; ++k) {
            float d = dv[k] + bias;
            if (p.softplus) { float sig; d = softplus_ref(d, sig); }
            d = (lbase + k < L) ? d : 0.0f;            // identity element past the end (a = 1, b = 0)
            dl[k] = d;
            dlu[k] = d * uv[k];
            y[k] = Dd * uv[k];
        }
        float dsum = 0.0f;
#pragma unroll
        for (int k = 0; k < T; ++k) dsum += dl[k];

        // state checkpoints: the lane whose segment ends on a multiple of the pitch (or holds the last
        // element) writes the state after its segment -- any pitch that T divides (320 / 640 / 1280)
        const int send = (lbase + T < L) ? (lbase + T) : L;
        const bool ckpt = x_row != nullptr && lbase < L && ((send % p.ckpt_pitch) == 0 || send == L);
        const int cidx = (send - 1) / p.ckpt_pitch;
        const float* sRunIn = sRun + (st & 1) * R * N + wr * N;
        float* sRunOut = sRun + ((st + 1) & 1) * R * N + wr * N;

        for (int sb = 0; sb < nsb; ++sb) {
            const int step = st * nsb + sb;
            const float* cur = sBC + (step & 1) * bufsz;
            // ---- stream the next block of states (or the first block of the next super-tile)
            {
                float* nxt = sBC + ((step + 1) & 1) * bufsz;
                if (sb + 1 < nsb) stage(nxt, (sb + 1) * NB, st * W);
                else if (st + 1 < nsuper) stage(nxt, 0, (st + 1) * W);
            }
            if (PREFETCH && sb == nsb - 1 && st + 1 < nsuper) {
                load_items<io_t, T, REV>(u_row, lbase + W * TILE, L, vec, uv);
                load_items<io_t, T, REV>(d_row, lbase + W * TILE, L, vec, dv);
            }

            const int n0 = sb * NB;
            const int nend = (N - n0 < NB) ? (N - n0) : NB;
#pragma unroll 1
            for (int nn = 0; nn < nend; ++nn) {
                const int n = n0 + nn;
                const float A2 = sA2[wr * N + n];
                const float* tB = cur + (nn * W + wt) * TILE;
                const float* tC = tB + NB * W * TILE;

                // ---- pass A: lane-local fold (zero incoming state; lane 0 of an unsplit tile starts from the running state)
                float a[T], bb[T];
                float xin = sRunIn[n];
                float xa = (!WSPLIT && lane == 0) ? xin : 0.0f;
#pragma unroll
                for (int q = 0; q < T / VW; ++q) {
                    float bq[VW];
                    lds_read_chunk<T, REV>(tB, lane, q, bq);
#pragma unroll
                    for (int j = 0; j < VW; ++j) {
                        const int k = VW * q + j;
                        a[k] = fast_exp2(dl[k] * A2);
                        bb[k] = dlu[k] * bq[j];
                        xa = fmaf(a[k], xa, bb[k]);
                    }
                }
                float x;
                if constexpr (!WSPLIT) {
                    // ---- wave scan, multiplicative form; the scanned value IS the state after each lane
                    float pl = fast_exp2(A2 * dsum);        // the lane's decay product
                    wave_mscan_inclusive(pl, xa);
                    x = wave_prev_lane(xa, xin);            // state entering this lane's segment
                } else {
                    // ---- wave scan of the lane aggregates (log2 of the decay product, end state)
                    float sa = A2 * dsum;
                    wave_scan_inclusive(sa, xa);
                    float2* agg = sAgg + ((n & 1) * R + wr) * W;
                    if (lane == 63) agg[wt] = make_float2(fast_exp2(sa), xa);
                    lds_barrier();
                    for (int w = 0; w < wt; ++w) {
                        const float2 t = agg[w];
                        xin = fmaf(t.x, xin, t.y);
                    }
                    const float pe = fast_exp2(wave_prev_lane(sa, 0.0f));
                    const float xe = wave_prev_lane(xa, 0.0f);
                    x = fmaf(pe, xin, xe);                  // state entering this lane's segment
                }

                // ---- pass B: replay with the true incoming state, accumulate C.x
#pragma unroll
                for (int q = 0; q < T / VW; ++q) {
                    float cq[VW];
                    lds_read_chunk<T, REV>(tC, lane, q, cq);
#pragma unroll
                    for (int j = 0; j < VW; ++j) {
                        const int k = VW * q + j;
                        x = fmaf(a[k], x, bb[k]);
                        y[k] = fmaf(cq[j], x, y[k]);
                    }
                }
                if (lane == 63 && wt == W - 1) sRunOut[n] = x;    // state after this super-tile
                if (ckpt) x_row[(long)cidx * N + n] = x;
            }
            if constexpr (GLDS) lds_dma_wait();
            __syncthreads();                            // staged block landed; current block consumed
        }
        if (!PREFETCH && st + 1 < nsuper) {
            load_items<io_t, T, REV>(u_row, lbase + W * TILE, L, vec, uv);
            load_items<io_t, T, REV>(d_row, lbase + W * TILE, L, vec, dv);
        }
        store_items<io_t, T, REV>(o_row, lbase, L, vec, y);
    }
}

// T = 20 keeps 100 values per lane live (delta, delta*u, y, a, b): it needs ~150 VGPRs, i.e. at
// most 3 waves per SIMD = 12 waves per workgroup; the shorter tiles fit 16 waves.
template <int T> struct fwd_max_waves { static constexpr int value = (T >= 20) ? 12 : 16; };

template <typename io_t, int T, bool GLDS, bool PREFETCH, bool WSPLIT>
__global__ void __launch_bounds__(64 * fwd_max_waves<T>::value)
scan_fwd_kernel(const FwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lb = xcd_logical_block(blockIdx.x, gridDim.x);
    const int b = lb / p.rowblocks;
    const int rb = lb - b * p.rowblocks;
    const int row0 = rb * p.R;
    const int g = row0 / p.rows_per_group;
    if ((p.rev_mask >> g) & 1u) scan_fwd_body<io_t, T, GLDS, PREFETCH, true, WSPLIT>(p, smem, b, row0, g);
    else scan_fwd_body<io_t, T, GLDS, PREFETCH, false, WSPLIT>(p, smem, b, row0, g);
}

template <typename io_t, int T, bool GLDS, bool PREFETCH, bool WSPLIT>
static hipError_t launch_fwd_w(const FwdArgs& a, hipStream_t stream) {
    const size_t lds = fwd_lds_bytes(T, a.R, a.W, a.NB, a.N);
    const int grid = a.rowblocks * a.batch;
    auto kern = scan_fwd_kernel<io_t, T, GLDS, PREFETCH, WSPLIT>;
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
    hipLaunchKernelGGL(kern, dim3(grid), dim3(a.R * a.W * 64), lds, stream, a);
    return hipGetLastError();
}

template <typename io_t, int T, bool GLDS, bool PREFETCH>
static hipError_t launch_fwd_t(const FwdArgs& a, hipStream_t stream) {
    return a.W > 1 ? launch_fwd_w<io_t, T, GLDS, PREFETCH, true>(a, stream) : launch_fwd_w<io_t, T, GLDS, PREFETCH, false>(a, stream);
}

template <typename io_t, bool GLDS>
static hipError_t launch_fwd_io(const FwdArgs& a, int T, bool prefetch, hipStream_t stream) {
    switch (T) {
        case 4: return launch_fwd_t<io_t, 4, GLDS, true>(a, stream);
        case 5: return launch_fwd_t<io_t, 5, GLDS, true>(a, stream);
        case 10: return prefetch ? launch_fwd_t<io_t, 10, GLDS, true>(a, stream) : launch_fwd_t<io_t, 10, GLDS, false>(a, stream);
        case 20: return launch_fwd_t<io_t, 20, GLDS, false>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_scan_fwd(const FwdArgs& a, int dtype, int T, bool glds, bool prefetch, hipStream_t stream) {
    switch (dtype) {
        case 0: return glds ? launch_fwd_io<float, true>(a, T, prefetch, stream) : launch_fwd_io<float, false>(a, T, prefetch, stream);
        case 1: return launch_fwd_io<f16_t, false>(a, T, false, stream);
        case 2: return launch_fwd_io<bf16_t, false>(a, T, false, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace sigma
