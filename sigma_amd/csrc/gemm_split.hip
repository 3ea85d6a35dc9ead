// gemm_split.hip -- fp32 GEMMs as split-operand bf16 MFMA GEMMs, the split done in registers (gfx950 / MI355X).
//
// Reference call sites: every nn.Linear of the hot path -- SS2D.in_proj / out_proj (models/encoders/vmamba.py:1067-1089),
// PatchMerging2D.reduction (:612-636), the CroMB / ConMB projections (:1588-1640, :1134-1284), the decoder linears
// (models/decoders/MambaDecoder.py:12-97) -- and the x_proj einsum of cross_selective_scan (vmamba.py:193-196).  The
// reference runs them as fp32 cuBLAS GEMMs; here (C ABI: include/sigma_gemm.h):
//
//   * every fp32 operand element is split on its way from global memory to LDS into two bf16 halves
//     (v_cvt_pk_bf16_f32: hi = bf16(x), lo = bf16(x - hi)), so the LDS images are bf16 [row][k] and cost the same
//     4 bytes per element as the fp32 tile would;
//   * a product is a_hi b_hi + a_hi b_lo + a_lo b_hi: three v_mfma_f32_32x32x16_bf16 per 32 x 32 x 16 block with one
//     fp32 accumulator (2.5 PFLOP/s bf16 / 3 = 830 TFLOP/s of fp32-equivalent peak against 157 TFLOP/s of fp32 MFMA);
//   * workgroup = 256 threads = WM x WN waves, tile BM x BN x 32; a wave owns TM x TN accumulators of 32 x 32;
//     the next k-step's global loads are in flight while the current one is multiplied (register prefetch, one LDS
//     image, two barriers per k-step; 2-3 workgroups per CU cover the barriers);
//   * the LDS row pitch is 80 bytes (32 bf16 + 8 pad): ds_read_b128 of 16 consecutive rows hits 16 distinct
//     4-bank groups (20 r mod 64, r = 0..15) -- conflict-free fragment reads without a swizzle;
//   * an operand whose reduction index is the SLOW memory index (the token dimension of the weight-gradient GEMM, the
//     feature dimension of the channels-first activations of x_proj) is transposed on the way into LDS: a thread
//     loads four k-rows of the same four columns and writes 4 x (hi, lo) ds_write_b64 of four consecutive k;
//   * workgroup ids are remapped so that the column tiles of one row tile run on the same XCD (its A tile is then
//     re-read from that XCD's L2).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sigma_gemm.h"
#include "../../include/sigma_ops.h"
#include "scan_device.h"

// Ablation builds (-DSIGMA_GEMM_ABL=<bits>; WRONG results, timing only): 1 no MFMA / fragment reads, 2 no global operand
// loads, 4 no C stores, 8 no fp32 -> bf16 split arithmetic (raw bits are stored), 16 no LDS stores and no barriers
#ifndef SIGMA_GEMM_ABL
#define SIGMA_GEMM_ABL 0
#endif

namespace sigma {
namespace {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int kBK = 32;            // reduction elements per k-step (two MFMA k-blocks of 16)
constexpr int kPitch = 40;         // bf16 elements per LDS row (80 bytes)

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    long M;                 // rows of C
    int N, K;               // columns of C, reduction length
    long lda, ldb, ldc;
    long sA, sB, sC;        // batch strides
    int ntn;                // column tiles
    int ntm;                // row tiles
    int slices;             // reduction slices (workgroups per output tile), each `slice_k` long (multiple of 32)
    int slice_k;
    int batch;              // problems
    int mode;               // 0: C = ..., 1: C += ... (plain), 2: atomicAdd (slices > 1)
    int a_mod;              // > 0: A of batch z is A + (z % a_mod) * sA  (weights shared by groups of problems)
};

// P bf16 pieces of two floats (packed pairs): piece[0] = bf16(x), piece[1] = bf16(x - piece[0]), piece[2] = bf16 of the
// next residual: 8 significant bits each, round to nearest even (v_cvt_pk_bf16_f32)
template <int P>
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned (&piece)[P]) {
#if SIGMA_GEMM_ABL & 8
#pragma unroll
    for (int q = 0; q < P; ++q) piece[q] = __builtin_bit_cast(unsigned, q & 1 ? x1 : x0);
    return;
#endif
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const f32x2_t f = {x0, x1};
        const bf16x2_t h = __builtin_convertvector(f, bf16x2_t);
        piece[q] = __builtin_bit_cast(unsigned, h);
        if (q + 1 < P) {
            x0 -= __builtin_bit_cast(float, piece[q] << 16);
            x1 -= __builtin_bit_cast(float, piece[q] & 0xffff0000u);
        }
    }
}

// ---- operand tile loaders: ROWS rows (output index) x 32 k, into registers, then split into the LDS images ----
// KS = false: memory [row][k] (k contiguous).  thread t: k-chunk t & 7 (4 floats), rows (t >> 3) + 32 i.
// KS = true : memory [k][row] (row contiguous).  thread t: k-group t & 7 (k = 4 (t & 7) + j), row chunk (t >> 3) (+ 32 i).
template <int ROWS, bool KS>
struct TileLoader {
    static constexpr int NV = KS ? ((ROWS + 127) / 128) * 4 : ROWS / 32;    // float4 registers per thread
    float4 v[NV];

    // src: first element of the operand (batch applied); row0: first row of the tile; nrows: valid rows of the operand;
    // k0: first reduction index of this step; kend: end of the reduction range.  FULL: the whole k-step is in range
    // (wave-uniform, the common case); otherwise out-of-range k are loaded from a clamped address and zeroed.  Every load
    // is unconditional (a per-lane "load or zero" makes hipcc branch around each load).
    template <bool FULL>
    __device__ __forceinline__ void load(const float* __restrict__ src, long ld, long row0, long nrows, int k0, int kend) {
        const int t = threadIdx.x;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#if SIGMA_GEMM_ABL & 2
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = make_float4(0.5f + t, 0.25f, 1.0f + i, 2.0f);
        asm volatile("" : "+v"(v[0].x));
        return;
#endif
        if constexpr (!KS) {
            const int kc = k0 + ((t & 7) << 2);
            const bool ok = FULL || kc < kend;                            // K % 4 == 0: a chunk is in or out as a whole
            const int kl = ok ? kc : kend - 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                long row = row0 + (t >> 3) + 32 * i;
                row = row < nrows ? row : nrows - 1;                      // tail rows: duplicates, masked at the C store
                const float4 x = *reinterpret_cast<const float4*>(src + row * ld + kl);
                v[i] = ok ? x : zero;
            }
        } else {
            const int kg = k0 + ((t & 7) << 2);
#pragma unroll
            for (int i = 0; i < NV / 4; ++i) {
                int ch = (t >> 3) + 32 * i;                               // chunk of 4 rows inside the tile
                ch = ch < ROWS / 4 ? ch : ROWS / 4 - 1;                   // lanes past the tile repeat its last chunk
                long row = row0 + 4L * ch;
                row = row + 4 <= nrows ? row : nrows - 4;                 // nrows % 4 == 0 (host-checked)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = FULL || kg + j < kend;
                    const int kl = ok ? kg + j : kend - 1;
                    const float4 x = *reinterpret_cast<const float4*>(src + (long)kl * ld + row);
                    v[4 * i + j] = ok ? x : zero;
                }
            }
        }
    }

    // split into P bf16 images [row][k], image q at img + q * img_stride
    template <int P>
    __device__ __forceinline__ void store(uint16_t* __restrict__ img, int img_stride) const {
        const int t = threadIdx.x;
        if constexpr (!KS) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int off = ((t >> 3) + 32 * i) * kPitch + ((t & 7) << 2);
                unsigned p01[P], p23[P];
                split_pair<P>(v[i].x, v[i].y, p01);
                split_pair<P>(v[i].z, v[i].w, p23);
#pragma unroll
                for (int q = 0; q < P; ++q) *reinterpret_cast<uint2*>(img + q * img_stride + off) = make_uint2(p01[q], p23[q]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV / 4; ++i) {
                int ch = (t >> 3) + 32 * i;
                ch = ch < ROWS / 4 ? ch : ROWS / 4 - 1;                   // repeated chunk: same values, same address
                const float* f0 = reinterpret_cast<const float*>(&v[4 * i + 0]);
                const float* f1 = reinterpret_cast<const float*>(&v[4 * i + 1]);
                const float* f2 = reinterpret_cast<const float*>(&v[4 * i + 2]);
                const float* f3 = reinterpret_cast<const float*>(&v[4 * i + 3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int off = (4 * ch + r) * kPitch + ((t & 7) << 2);
                    unsigned p01[P], p23[P];
                    split_pair<P>(f0[r], f1[r], p01);
                    split_pair<P>(f2[r], f3[r], p23);
#pragma unroll
                    for (int q = 0; q < P; ++q) *reinterpret_cast<uint2*>(img + q * img_stride + off) = make_uint2(p01[q], p23[q]);
                }
            }
        }
    }
};

#ifndef SIGMA_GEMM_DEPTH
#define SIGMA_GEMM_DEPTH 2
#endif
constexpr int kDepth = SIGMA_GEMM_DEPTH;   // k-steps of operands in flight per workgroup (register ring; 3 spills at 128 x 128)

// one output tile (x one reduction slice) of one problem of the batch
struct Item {
    long m0; int n0, kbeg, kend, sl;
    const float* Ab; const float* Bb; float* Cb;
};

// Persistent workgroups: workgroup b walks the work items b, b + gridDim.x, ... (an item = one BM x BN output tile
// x one reduction slice); its k-steps form ONE stream that crosses item boundaries, and the operands of step s + kDepth
// are requested from global memory when step s is consumed -- HBM latency (~2 us under load, i.e. several k-steps of
// MFMA time) is covered by the ring instead of by occupancy (174 registers: two workgroups per CU), and the epilogue
// of a tile overlaps the first loads of the next one.  Measured against the first version (operands one step ahead,
// one tile per workgroup): see profiles/r03_gemm_bench.jsonl.
template <int BM, int BN, int WM, int WN, bool A_KS, bool B_KS, int P>
__global__ void __launch_bounds__(256, 2)
gemm_split3_kernel(const GemmArgs g) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    static_assert(TM >= 1 && TN >= 1 && TM * 32 * WM == BM && TN * 32 * WN == BN, "tile shape");
    static_assert(P == 2 || P == 3, "two or three bf16 pieces per operand");
    __shared__ __attribute__((aligned(16))) uint16_t smem[P * (BM + BN) * kPitch];
    uint16_t* sA = smem;                               // [P][BM][kPitch]
    uint16_t* sB = smem + P * BM * kPitch;             // [P][BN][kPitch]

    const int per_z = g.ntm * g.ntn * g.slices;
    const int total = g.batch * per_z;
    // item id -> (batch, row tile, column tile, slice); ids that are consecutive after the XCD remap (column tiles and
    // slices of one row tile) run on the same XCD at about the same time: the A tile is re-read from that XCD's L2
    auto decode = [&](int id, Item& it) {
        const int lbk = xcd_logical_block(id, total);
        const int z = lbk / per_z;
        const int r0 = lbk - z * per_z;
        const int tm_i = r0 / (g.ntn * g.slices);
        const int rem = r0 - tm_i * (g.ntn * g.slices);
        const int tn_i = rem / g.slices;
        it.sl = rem - tn_i * g.slices;
        it.m0 = (long)tm_i * BM;
        it.n0 = tn_i * BN;
        it.kbeg = it.sl * g.slice_k;
        it.kend = (it.kbeg + g.slice_k < g.K) ? it.kbeg + g.slice_k : g.K;
        it.Ab = g.A + (long)(g.a_mod > 0 ? z % g.a_mod : z) * g.sA;
        it.Bb = g.B + (long)z * g.sB;
        it.Cb = g.C + (long)z * g.sC;
    };

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;

    // producer cursor (next k-step to request) and consumer cursor (next k-step to multiply); all wave-uniform
    int p_id = blockIdx.x, c_id = blockIdx.x;
    bool p_on = p_id < total, c_on = p_on;
    Item pit, cit;
    int pk = 0, ck = 0;
    if (p_on) { decode(p_id, pit); cit = pit; pk = pit.kbeg; ck = pk; }

    TileLoader<BM, A_KS> la[kDepth];
    TileLoader<BN, B_KS> lb_[kDepth];
    auto produce = [&](TileLoader<BM, A_KS>& LA, TileLoader<BN, B_KS>& LB) {
        if (!p_on) return;
        if (pk + kBK <= pit.kend) {
            LA.template load<true>(pit.Ab, g.lda, pit.m0, g.M, pk, pit.kend);
            LB.template load<true>(pit.Bb, g.ldb, pit.n0, g.N, pk, pit.kend);
        } else {
            LA.template load<false>(pit.Ab, g.lda, pit.m0, g.M, pk, pit.kend);
            LB.template load<false>(pit.Bb, g.ldb, pit.n0, g.N, pk, pit.kend);
        }
        pk += kBK;
        if (pk >= pit.kend) {
            p_id += gridDim.x;
            p_on = p_id < total;
            if (p_on) { decode(p_id, pit); pk = pit.kbeg; }
        }
    };
#pragma unroll
    for (int d = 0; d < kDepth; ++d) produce(la[d], lb_[d]);

    f32x16_t acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    };
    zero_acc();

    // fragment addresses: lane l -> row (l & 31) of the 32-row block, k-block 8 (l >> 5) of the 16
    const int frag = (lane & 31) * kPitch + ((lane >> 5) << 3);
    const uint16_t* fA = sA + (wm * TM * 32) * kPitch + frag;
    const uint16_t* fB = sB + (wn * TN * 32) * kPitch + frag;

    // epilogue: C/D layout of the 32x32 MFMA: register r of lane l = row (r & 3) + 8 (r >> 2) + 4 (l >> 5), col l & 31
    auto epilogue = [&](const Item& it, auto put) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = it.n0 + (wn * TN + j) * 32 + (lane & 31);
            const bool col_ok = col < g.N;
            const float bv = (g.bias != nullptr && col_ok && it.sl == 0) ? g.bias[col] : 0.0f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const long rbase = it.m0 + (wm * TM + i) * 32 + ((lane >> 5) << 2);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long row = rbase + (r & 3) + ((r >> 2) << 3);
                    const float val = acc[i][j][r] + bv;
#if SIGMA_GEMM_ABL & 4
                    asm volatile("" :: "v"(val), "v"(row), "v"(col_ok));
#else
                    if (col_ok && row < g.M) put(it.Cb + row * g.ldc + col, val);
#endif
                }
            }
        }
    };

    while (c_on) {
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
            if (!c_on) break;
#if SIGMA_GEMM_ABL & 16
            asm volatile("" :: "v"(la[u].v[0].x), "v"(lb_[u].v[0].x));
#else
            la[u].template store<P>(sA, BM * kPitch);
            lb_[u].template store<P>(sB, BN * kPitch);
            __syncthreads();
#endif
            produce(la[u], lb_[u]);                    // refill the slot just written to LDS: step s + kDepth
#pragma unroll
            for (int ks = 0; ks < (SIGMA_GEMM_ABL & 1 ? 0 : kBK / 16); ++ks) {
                bf16x8_t fa[P][TM], fb[P][TN];
#pragma unroll
                for (int q = 0; q < P; ++q) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[q][i] = *reinterpret_cast<const bf16x8_t*>(fA + q * BM * kPitch + i * 32 * kPitch + ks * 16);
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[q][j] = *reinterpret_cast<const bf16x8_t*>(fB + q * BN * kPitch + j * 32 * kPitch + ks * 16);
                }
                // piece products with qa + qb < P, smallest first: P = 2: lo*hi, hi*lo, hi*hi (dropped lo*lo ~ 2^-16);
                // P = 3: the six terms down to 2^-16 (dropped ~ 2^-24: fp32 GEMM accuracy)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int sum = P - 1; sum >= 0; --sum)
#pragma unroll
                            for (int qa = sum; qa >= 0; --qa)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][i], fb[sum - qa][j], acc[i][j], 0, 0, 0);
            }
#if !(SIGMA_GEMM_ABL & 16)
            __syncthreads();
#endif
            ck += kBK;
            if (ck >= cit.kend) {                      // tile (slice) complete
                if (g.mode == 0) epilogue(cit, [](float* dst, float v) { *dst = v; });
                else if (g.mode == 1) epilogue(cit, [](float* dst, float v) { *dst += v; });
                else epilogue(cit, [](float* dst, float v) { atomicAdd(dst, v); });
                zero_acc();
                c_id += gridDim.x;
                c_on = c_id < total;
                if (c_on) { decode(c_id, cit); ck = cit.kbeg; }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, bool A_KS, bool B_KS, int P>
hipError_t launch_cfg(const GemmArgs& g, int batch, hipStream_t stream) {
    const long items = (long)batch * g.ntm * g.ntn * g.slices;
    if (items <= 0 || items > 0x7fffffffL) return hipErrorInvalidValue;
    GemmArgs ga = g;
    ga.batch = batch;
    // persistent workgroups: as many as are resident at once (register-limited: 2 per CU at 128 x 128, 3-4 for the
    // narrower tiles); a multiple of the 8 XCDs so that the XCD remap of the item ids keeps its meaning
    static int per_cu = 0;                              // resident workgroups per CU of this instantiation
    if (per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_split3_kernel<BM, BN, WM, WN, A_KS, B_KS, P>, 256, 0) != hipSuccess || n < 1) n = 2;
        per_cu = n > 4 ? 4 : n;
    }
    long grid = 256L * per_cu;
    if (grid > items) grid = items;
    hipLaunchKernelGGL((gemm_split3_kernel<BM, BN, WM, WN, A_KS, B_KS, P>), dim3((unsigned)grid), dim3(256), 0, stream, ga);
    return hipGetLastError();
}

// tile width for N columns: 128 unless a narrower tile wastes less ((N = 96, 192: 96-wide tiles are exact)
int pick_bn(int N) {
    if (N % 128 == 0) return 128;
    if (N % 96 == 0) return 96;
    if (N <= 64) return 64;
    if (N <= 96) return 96;
    // padded columns of each candidate
    const int w128 = (N + 127) / 128 * 128 - N, w96 = (N + 95) / 96 * 96 - N;
    return w96 < w128 ? 96 : 128;
}

template <bool A_KS, bool B_KS, int P>
hipError_t launch_p(GemmArgs& g, int batch, hipStream_t stream) {
    const int bn = pick_bn(g.N);
    g.ntm = (int)((g.M + 127) / 128);
    g.ntn = (g.N + bn - 1) / bn;
    if (bn == 128) return launch_cfg<128, 128, 2, 2, A_KS, B_KS, P>(g, batch, stream);
    if (bn == 96) return launch_cfg<128, 96, 4, 1, A_KS, B_KS, P>(g, batch, stream);
    return launch_cfg<128, 64, 2, 2, A_KS, B_KS, P>(g, batch, stream);
}

template <bool A_KS, bool B_KS>
hipError_t launch_any(GemmArgs& g, int batch, int pieces, hipStream_t stream) {
    return pieces == 3 ? launch_p<A_KS, B_KS, 3>(g, batch, stream) : launch_p<A_KS, B_KS, 2>(g, batch, stream);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int fill_common(const sigma_gemm_params* p, GemmArgs& g) {
    if (!p || !p->A || !p->Bt || !p->C) return SIGMA_OPS_ERR_ARG;
    if (p->M < 0 || p->N < 0 || p->K < 0 || p->batch < 0) return SIGMA_OPS_ERR_ARG;
    if (p->pieces != 0 && p->pieces != 2 && p->pieces != 3) return SIGMA_OPS_ERR_ARG;
    if (!aligned16(p->A) || !aligned16(p->Bt) || p->lda % 4 != 0 || p->ldb % 4 != 0) return SIGMA_OPS_ERR_ARG;
    if (p->batch > 1 && (p->strideA % 4 != 0 || p->strideB % 4 != 0)) return SIGMA_OPS_ERR_ARG;
    g.A = p->A; g.B = p->Bt; g.C = p->C; g.bias = p->bias;
    g.lda = p->lda; g.ldb = p->ldb; g.ldc = p->ldc;
    g.sA = p->strideA; g.sB = p->strideB; g.sC = p->strideC;
    g.slices = 1; g.a_mod = 0;
    g.mode = p->accumulate ? 1 : 0;
    return SIGMA_OPS_OK;
}

}  // namespace
}  // namespace sigma

extern "C" int sigma_gemm_nt_split3(const sigma_gemm_params* p, void* stream) {
    sigma::GemmArgs g;
    int rc = sigma::fill_common(p, g);
    if (rc) return rc;
    if (p->K % 4 != 0) return SIGMA_OPS_ERR_ARG;
    const int batch = p->batch > 0 ? p->batch : 1;
    if (p->M == 0 || p->N == 0) return SIGMA_OPS_OK;
    g.M = p->M; g.N = p->N; g.K = p->K;
    if (p->K == 0) return SIGMA_OPS_ERR_ARG;
    g.slice_k = (p->K + 31) / 32 * 32;
    g.a_mod = p->a_mod;
    hipError_t e = sigma::launch_any<false, false>(g, batch, p->pieces, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? SIGMA_OPS_OK : SIGMA_OPS_ERR_LAUNCH;
}

extern "C" int sigma_gemm_nn_split3(const sigma_gemm_params* p, void* stream) {
    // C = A B with B = (K, N) row-major (params->Bt, row stride ldb): the B operand's reduction index is its slow index
    sigma::GemmArgs g;
    int rc = sigma::fill_common(p, g);
    if (rc) return rc;
    if (p->K % 4 != 0 || p->N % 4 != 0) return SIGMA_OPS_ERR_ARG;
    const int batch = p->batch > 0 ? p->batch : 1;
    if (p->M == 0 || p->N == 0) return SIGMA_OPS_OK;
    g.M = p->M; g.N = p->N; g.K = p->K;
    if (p->K == 0) return SIGMA_OPS_ERR_ARG;
    g.slice_k = (p->K + 31) / 32 * 32;
    g.a_mod = p->a_mod;
    hipError_t e = sigma::launch_any<false, true>(g, batch, p->pieces, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? SIGMA_OPS_OK : SIGMA_OPS_ERR_LAUNCH;
}

extern "C" int sigma_gemm_tn_split3(const sigma_gemm_params* p, void* stream) {
    // C (N_out x K_in) (+)= A^T B with A = (M, N_out), Bt = (M, K_in): both operands have the reduction (token) index
    // as the slow memory index.  Kernel view: rows of C = N_out, columns = K_in, reduction = M.
    sigma::GemmArgs g;
    int rc = sigma::fill_common(p, g);
    if (rc) return rc;
    if (p->N % 4 != 0 || p->K % 4 != 0) return SIGMA_OPS_ERR_ARG;
    const int batch = p->batch > 0 ? p->batch : 1;
    if (p->N == 0 || p->K == 0 || p->M == 0) return SIGMA_OPS_OK;
    g.M = p->N; g.N = p->K; g.K = (int)p->M;
    if (p->M > 0x7fffff00L) return SIGMA_OPS_ERR_ARG;
    // reduction slices: enough workgroups for ~3 per CU, each slice a multiple of 32 tokens, at least 8 k-steps long
    const int bn = sigma::pick_bn(g.N);
    const long tiles = (long)batch * ((g.M + 127) / 128) * ((g.N + bn - 1) / bn);
    const long steps = (p->M + 31) / 32;
    long want = (768 + tiles - 1) / tiles;
    if (want > steps / 8) want = steps / 8;
    if (want < 1) want = 1;
    const long per = (steps + want - 1) / want;
    g.slice_k = (int)(per * 32);
    g.slices = (int)((steps + per - 1) / per);
    if (g.slices > 1) g.mode = 2;                     // caller zero-filled C (or accumulates)
    hipError_t e = sigma::launch_any<true, true>(g, batch, p->pieces, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? SIGMA_OPS_OK : SIGMA_OPS_ERR_LAUNCH;
}
