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
#include <stdlib.h>

#include "../../include/sigma_gemm.h"
#include "../../include/sigma_ops.h"
#include "scan_device.h"

// Ablation builds (-DSIGMA_GEMM_ABL=<bits>; WRONG results, timing only): 1 no MFMA / fragment reads, 2 no global operand
// loads, 4 no C stores, 8 no fp32 -> bf16 split arithmetic (raw bits are stored), 16 no LDS stores and no barriers
#ifndef SIGMA_GEMM_ABL
#define SIGMA_GEMM_ABL 0
#endif

// Development builds (-DSIGMA_GEMM_PROF=1): clocks per phase of the k-step loop, summed over the waves (sigma_scan_debug_read;
// tools/diag/gemm_prof.py): 0 wait for operand loads, 1 split + LDS stores, 2 barrier, 3 load issue, 4 fragment reads + MFMA
// issue, 5 barrier, 6 epilogue + store drain, 7 item switch; 13 tiles, 14 all clocks of the loop, 15 k-steps.  The probes are
// scheduling barriers and each costs an s_memtime round trip: the instrumented loop runs ~30 % longer.
#ifndef SIGMA_GEMM_PROF
#define SIGMA_GEMM_PROF 0
#endif
#if SIGMA_GEMM_PROF
__device__ unsigned long long g_gemm_prof[16];
#define GPROF(slot) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); \
                      prof_[slot] += t_ - tp_; tp_ = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define GPROF(slot)
#endif

namespace sigma {

#if SIGMA_GEMM_PROF
hipError_t gemm_prof_read(unsigned long long* out16) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_gemm_prof), 16 * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_prof), z, sizeof(z));
}
#else
hipError_t gemm_prof_read(unsigned long long* out16) { for (int i = 0; i < 16; ++i) out16[i] = 0; return hipSuccess; }
#endif

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
    int c_mod;              // > 0: C of batch z is C + (z % c_mod) * sC  (mode 2: the problems sharing it are summed)
    const float* R; const float* R2;   // epilogue addends (or null), row stride ldr, batch stride sR
    long ldr, sR;
    // columns [0, t_cols) of the result go TRANSPOSED to Ct[col * ldct + row] (row-contiguous epilogue only, t_cols % 32 == 0);
    // columns >= t_cols to C[row * ldc + col - t_cols]
    float* Ct; long ldct; int t_cols;
    // nparts > 0: two-stage sums.  C is a scratch buffer of compact (M x N, row stride N) partial results: the item of
    // slice sl of problem z stores (plainly) into part (z / c_mod) * slices + sl of output z % c_mod (c_mod == 0: output z),
    // part_stride floats apart; reduce_parts_kernel then sums the parts of every output into the caller's C
    int nparts; long part_stride;
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

// ---- operand tile loaders: ROWS rows (output index) x 32 k ------------------------------------------------------------
// KS = false: memory [row][k] (k contiguous).  thread t: k-chunk t & 7 (4 floats), rows (t >> 3) + 32 i.
// KS = true : memory [k][row] (row contiguous).  thread t: k = 4 (t & 7) + j, row chunk (t >> 3) (+ 32 i): four k-rows of
//             the same four rows, packed along k on the way into LDS (the transposition).
// A load is global_load_dwordx4 vdst, voffset, s[base]: the per-thread 32-bit byte offsets depend only on the thread and
// on the tile's row clamp (recomputed when the workgroup moves to another tile), the 64-bit base is wave-uniform and
// moves with the k-step.  The loads are issued from inline asm, so hipcc does not know them: with loads of its own it
// drains vmcnt(0) at every loop-carried use (measured: operands 1, 2 or 3 steps ahead ran at the same speed,
// profiles/r03_gemm_depth.txt); the kernel counts them itself (LPS loads per stage, s_waitcnt vmcnt(newer * LPS)).
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int ROWS, bool KS>
struct TileLoader {
    static constexpr int NV = KS ? ((ROWS + 127) / 128) * 4 : ROWS / 32;    // 16-byte loads per thread and stage
    static constexpr int NOFF = KS ? NV / 4 : NV;                           // KS: the four k-rows of a chunk are ld apart
    unsigned off[NOFF];

    // row0: first row of the tile, nrows: rows of the operand (tail rows repeat the last valid row / chunk: their
    // products are masked at the C store)
    __device__ __forceinline__ void set_tile(long ld, long row0, long nrows) {
        const int t = threadIdx.x;
        if constexpr (!KS) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                long row = row0 + (t >> 3) + 32 * i;
                row = row < nrows ? row : nrows - 1;
                off[i] = (unsigned)(((row - row0) * ld + ((t & 7) << 2)) * 4);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV / 4; ++i) {
                int ch = (t >> 3) + 32 * i;
                ch = ch < ROWS / 4 ? ch : ROWS / 4 - 1;                   // lanes past the tile repeat its last chunk
                long row = row0 + 4L * ch;
                row = row + 4 <= nrows ? row : nrows - 4;                 // nrows % 4 == 0, row0 % 4 == 0 (host-checked)
                off[i] = (unsigned)(((long)((t & 7) << 2) * ld + (row - row0)) * 4);
            }
        }
    }

    // base: element (row0, k0) of the operand; rem = reduction elements left from k0 (>= 32: a full step)
    __device__ __forceinline__ void issue(f32x4_t (&v)[NV], const float* base, int rem, long ld) const {
        const int t = threadIdx.x;
#if SIGMA_GEMM_ABL & 2
#pragma unroll
        for (int i = 0; i < NV; ++i) { v[i] = f32x4_t{0.5f + t, 0.25f, 1.0f + i, 2.0f}; asm volatile("" : "+v"(v[i])); }
        return;
#endif
        const unsigned ldb4 = (unsigned)(ld * 4);
        // the base is wave-uniform by construction; readfirstlane folds away when the compiler knows it and keeps the
        // "s" operand of the asm legal when it does not (it then holds the pointer in VGPRs)
        const uintptr_t bq = reinterpret_cast<uintptr_t>(base);
        base = reinterpret_cast<const float*>(((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(bq >> 32)) << 32) |
                                              (unsigned)__builtin_amdgcn_readfirstlane((int)(bq & 0xffffffffu)));
        // The full step and the partial last step of a reduction are two code paths behind a wave-uniform BRANCH (both hold
        // volatile asm, so hipcc cannot turn the branch into selects): as selects the tail adjustment cost every full step
        // 5 + NV VALU here and 2 selects per element in store() -- 64 of the ~210 VALU of a 128 x 128 k-step (round 6, ISA).
        if (rem >= kBK) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const unsigned o = KS ? off[i >> 2] + (unsigned)(i & 3) * ldb4 : off[i];
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v[i]) : "v"(o), "s"(base) : "memory");
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                unsigned o = KS ? off[i >> 2] + (unsigned)(i & 3) * ldb4 : off[i];
                if constexpr (!KS) {
                    const int kc = (t & 7) << 2;
                    if (kc >= rem) o -= (unsigned)(kc - (rem - 4)) * 4u;            // K % 4 == 0: re-read the last chunk
                } else {
                    const int kk = ((t & 7) << 2) + (i & 3);
                    if (kk >= rem) o -= (unsigned)((long)(kk - (rem - 1)) * ld * 4);
                }
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v[i]) : "v"(o), "s"(base) : "memory");
            }
        }
    }

    // split into P bf16 images [row][k], image q at img + q * img_stride; elements past the reduction range are zero
    template <int P>
    __device__ __forceinline__ void store(f32x4_t (&xs)[NV], int rem, uint16_t* __restrict__ img, int img_stride) const {
        const int t = threadIdx.x;
        // only the last, partial k-step of a reduction masks (in place: the ring slot is refilled next): a wave-uniform
        // branch the compiler cannot flatten (the empty volatile asm inside it), so the full step carries no selects
        if (rem < kBK) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const bool ok = KS ? (((t & 7) << 2) + (i & 3) < rem) : (((t & 7) << 2) < rem);
                xs[i] = ok ? xs[i] : f32x4_t{0.f, 0.f, 0.f, 0.f};
                asm volatile("" : "+v"(xs[i]));
            }
        }
        if constexpr (!KS) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int o = ((t >> 3) + 32 * i) * kPitch + ((t & 7) << 2);
                const f32x4_t x = xs[i];
                unsigned p01[P], p23[P];
                split_pair<P>(x[0], x[1], p01);
                split_pair<P>(x[2], x[3], p23);
#pragma unroll
                for (int q = 0; q < P; ++q) *reinterpret_cast<uint2*>(img + q * img_stride + o) = make_uint2(p01[q], p23[q]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV / 4; ++i) {
                int ch = (t >> 3) + 32 * i;
                ch = ch < ROWS / 4 ? ch : ROWS / 4 - 1;                   // repeated chunk: same values, same address
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = (4 * ch + r) * kPitch + ((t & 7) << 2);
                    unsigned p01[P], p23[P];
                    split_pair<P>(xs[4 * i + 0][r], xs[4 * i + 1][r], p01);
                    split_pair<P>(xs[4 * i + 2][r], xs[4 * i + 3][r], p23);
#pragma unroll
                    for (int q = 0; q < P; ++q) *reinterpret_cast<uint2*>(img + q * img_stride + o) = make_uint2(p01[q], p23[q]);
                }
            }
        }
    }
};

template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// k-steps of operands in flight per workgroup (register ring).  Measured (profiles/r03_gemm_depth.txt): with both
// operands k-contiguous or one of them transposed, one step ahead is as fast as two (and leaves the registers to the
// compiler's schedule); the weight-gradient form (both operands transposed on the way in, reduction over the tokens
// straight from HBM) gains 10-20 % from two on the long-token shapes.
#ifndef SIGMA_GEMM_DEPTH
#define SIGMA_GEMM_DEPTH 0
#endif
#ifndef SIGMA_GEMM_SLICE_MAJOR
#define SIGMA_GEMM_SLICE_MAJOR 1           // 0: A/B builds with the tile-major item order for sliced reductions
#endif
#ifndef SIGMA_GEMM_FRAG_PIN
#define SIGMA_GEMM_FRAG_PIN 1              // 0: A/B builds that leave the placement of the fragment reads to the compiler
#endif
#ifndef SIGMA_GEMM_ROW_EPILOGUE
#define SIGMA_GEMM_ROW_EPILOGUE 1          // 0: A/B builds with the direct (dword) epilogue
#endif
template <bool A_KS, bool B_KS>
struct ring_depth { static constexpr int value = SIGMA_GEMM_DEPTH ? SIGMA_GEMM_DEPTH : ((A_KS && B_KS) ? 2 : 1); };

// one output tile (x one reduction slice) of one problem of the batch
struct Item {
    long m0; int n0, kbeg, kend, sl;
    const float* Ab; const float* Bb; float* Cb;
    long r_off;             // element offset of this problem's addends
};

// Persistent workgroups: workgroup b walks the work items b, b + gridDim.x, ... (an item = one BM x BN output tile
// x one reduction slice); its k-steps form ONE stream that crosses item boundaries, and the operands of step s + kDepth
// are requested from global memory when step s is consumed -- HBM latency (~2 us under load, i.e. several k-steps of
// MFMA time) is covered by the ring instead of by occupancy (two workgroups per CU), and the epilogue of a tile
// overlaps the first loads of the next one.
template <int BM, int BN, int WM, int WN, bool A_KS, bool B_KS, int P, bool RES>
__global__ void __launch_bounds__(256, 2)
gemm_split3_kernel(const GemmArgs g) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    static_assert(TM >= 1 && TN >= 1 && TM * 32 * WM == BM && TN * 32 * WN == BN, "tile shape");
    static_assert(P == 2 || P == 3, "two or three bf16 pieces per operand");
    constexpr int kDepth = ring_depth<A_KS, B_KS>::value;
    static_assert(kDepth >= 1 && kDepth <= 3, "ring depth");
    __shared__ __attribute__((aligned(16))) uint16_t smem[P * (BM + BN) * kPitch];
    uint16_t* sA = smem;                               // [P][BM][kPitch]
    uint16_t* sB = smem + P * BM * kPitch;             // [P][BN][kPitch]
    typedef TileLoader<BM, A_KS> LoaderA;
    typedef TileLoader<BN, B_KS> LoaderB;
    constexpr int LPS = LoaderA::NV + LoaderB::NV;     // loads per thread and stage
    static_assert((kDepth - 1) * LPS <= 63, "vmcnt is 6 bits");

    const int per_z = g.ntm * g.ntn * g.slices;
    const int total = g.batch * per_z;
    // item id -> (batch, row tile, column tile, slice); ids that are consecutive after the XCD remap (column tiles and
    // slices of one row tile) run on the same XCD at about the same time: the A tile is re-read from that XCD's L2
    auto decode = [&](int id, Item& it) {
        // the item index is wave-uniform by construction (workgroup id + k * grid); said explicitly, so that everything
        // derived from it (tile origin, operand bases -> the SGPR operands of the asm loads) stays on the scalar unit
        const int lbk = __builtin_amdgcn_readfirstlane(xcd_logical_block(id, total));
        const int z = lbk / per_z;
        const int r0 = lbk - z * per_z;
        // reduction slices (weight gradients): the tiles of ONE slice are consecutive (slice-major), so that the workgroups an
        // XCD runs at the same time stream the same k-range of both operands -- each slice of A and B then comes from HBM once
        // per XCD instead of once per row / column tile (the tile-major order shares only the A slice among the column
        // tiles).  One division chain for both orders (selects on wave-uniform values, no branch).
        const bool slice_major = SIGMA_GEMM_SLICE_MAJOR && B_KS && g.slices > 1;
        const int d1 = slice_major ? g.ntm * g.ntn : g.ntn * g.slices;
        const int q1 = r0 / d1;
        const int rem = r0 - q1 * d1;
        const int d2 = slice_major ? g.ntn : g.slices;
        const int q2 = rem / d2;
        const int q3 = rem - q2 * d2;
        const int tm_i = slice_major ? q2 : q1, tn_i = slice_major ? q3 : q2;
        it.sl = slice_major ? q1 : q3;
        it.m0 = (long)tm_i * BM;
        it.n0 = tn_i * BN;
        it.kbeg = it.sl * g.slice_k;
        it.kend = (it.kbeg + g.slice_k < g.K) ? it.kbeg + g.slice_k : g.K;
        it.Ab = g.A + (long)(g.a_mod > 0 ? z % g.a_mod : z) * g.sA;
        it.Bb = g.B + (long)z * g.sB;
        const int cidx = g.c_mod > 0 ? z % g.c_mod : z;
        const long part = (long)cidx * g.nparts + (long)(g.c_mod > 0 ? z / g.c_mod : 0) * g.slices + it.sl;
        it.Cb = g.C + (g.nparts > 0 ? part * g.part_stride : (long)cidx * g.sC);
        it.r_off = (long)z * g.sR;
    };

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;

    // producer cursor (next k-step to request) and consumer cursor (next k-step to multiply); all wave-uniform
    int p_id = blockIdx.x, c_id = blockIdx.x;
    bool p_on = p_id < total, c_on = p_on;
    Item pit, cit;
    int pk = 0, ck = 0;
    LoaderA la;
    LoaderB lb_;
    // operand bases of the producer's next k-step (wave-uniform; advanced by one k-step per request instead of being
    // rebuilt from the item every time: ~20 SALU of 64-bit multiplies per k-step)
    const float* pa = nullptr;
    const float* pb = nullptr;
    auto seek = [&]() {
        pa = pit.Ab + (A_KS ? (long)pk * g.lda + pit.m0 : pit.m0 * g.lda + pk);
        pb = pit.Bb + (B_KS ? (long)pk * g.ldb + pit.n0 : (long)pit.n0 * g.ldb + pk);
    };
    const long a_step = A_KS ? (long)kBK * g.lda : kBK, b_step = B_KS ? (long)kBK * g.ldb : kBK;
    if (p_on) {
        decode(p_id, pit); cit = pit; pk = pit.kbeg; ck = pk;
        la.set_tile(g.lda, pit.m0, g.M);
        lb_.set_tile(g.ldb, pit.n0, g.N);
        seek();
    }

    f32x4_t va[kDepth][LoaderA::NV], vb[kDepth][LoaderB::NV];     // the ring
    int rem_[kDepth];                                              // reduction elements of the stage in slot d (0: empty)
    auto produce = [&](int d_, f32x4_t (&VA)[LoaderA::NV], f32x4_t (&VB)[LoaderB::NV]) {
        rem_[d_] = 0;
        if (!p_on) return;
        const int rem = pit.kend - pk;
        la.issue(VA, pa, rem, g.lda);
        lb_.issue(VB, pb, rem, g.ldb);
        rem_[d_] = rem < kBK ? rem : kBK;
        pk += kBK;
        pa += a_step;
        pb += b_step;
        if (pk >= pit.kend) {
            p_id += gridDim.x;
            p_on = p_id < total;
            if (p_on) {
                decode(p_id, pit); pk = pit.kbeg;
                la.set_tile(g.lda, pit.m0, g.M);
                lb_.set_tile(g.ldb, pit.n0, g.N);
                seek();
            }
        }
    };
#pragma unroll
    for (int d = 0; d < kDepth; ++d) produce(d, va[d], vb[d]);

    f32x16_t acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    };
    // RES: the accumulators of a tile START as its epilogue addends (C = residual (+ residual2) + A B): the loads land
    // straight in the accumulator registers while the first k-steps are staged, no registers of their own and nothing
    // added to the epilogue.  Wave-uniform base + 32-bit element offsets (host: M * ldr < 2^31); rows / columns past the
    // matrix are clamped (their sums are never stored).
    const bool res_rows = RES && (g.N & 3) == 0 && (g.ldr & 3) == 0 && (g.sR & 3) == 0 && (reinterpret_cast<uintptr_t>(g.R) & 15) == 0 &&
                          (g.R2 == nullptr || (reinterpret_cast<uintptr_t>(g.R2) & 15) == 0) && SIGMA_GEMM_ROW_EPILOGUE;
    auto init_acc = [&](const Item& it) {
        if constexpr (!RES) { zero_acc(); return; }
        const float* __restrict__ r1 = g.R + it.r_off;
        const float* __restrict__ r2 = g.R2 ? g.R2 + it.r_off : nullptr;
        const unsigned ldr = (unsigned)g.ldr;
        if (res_rows) {
            // the mirror image of epilogue_rows: 16-byte loads of 8 x 128 contiguous bytes (both addends summed in
            // registers), one 32 x 32 block at a time through the wave's 4 KB of LDS into accumulator order -- the
            // direct form below is 64 (128 with two addends) dword loads per thread and tile.  The operand image is
            // free here (between two tiles); the barrier keeps the first operand stores of the tile off the staging areas.
            float* stage = reinterpret_cast<float*>(smem) + wave * 1024;
            const int rdw = ((lane >> 5) << 2) * 32 + (lane & 31);
            const int ld_row = lane >> 3, ld_col = (lane & 7) << 2;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = it.n0 + (wn * TN + j) * 32 + ld_col;
                const unsigned cc_ = (unsigned)(col < g.N ? col : g.N - 4);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const long rbase = it.m0 + (wm * TM + i) * 32 + ld_row;
                    f32x4_t v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const long row = rbase + 8 * q;
                        const unsigned o = (unsigned)(row < g.M ? row : g.M - 1) * ldr + cc_;
                        v[q] = *reinterpret_cast<const f32x4_t*>(r1 + o);
                        if (r2 != nullptr) v[q] += *reinterpret_cast<const f32x4_t*>(r2 + o);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4_t*>(stage + (ld_row + 8 * q) * 32 + ld_col) = v[q];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = stage[rdw + ((r & 3) + ((r >> 2) << 3)) * 32];
                }
            }
            lds_barrier();
            return;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = it.n0 + (wn * TN + j) * 32 + (lane & 31);
            const unsigned cc_ = (unsigned)(col < g.N ? col : g.N - 1);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const long rbase = it.m0 + (wm * TM + i) * 32 + ((lane >> 5) << 2);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long row = rbase + (r & 3) + ((r >> 2) << 3);
                    const unsigned o = (unsigned)(row < g.M ? row : g.M - 1) * ldr + cc_;
                    acc[i][j][r] = r1[o];
                    if (r2 != nullptr) acc[i][j][r] += r2[o];
                }
            }
        }
    };
    if (c_on) init_acc(cit); else zero_acc();

    // fragment addresses: lane l -> row (l & 31) of the 32-row block, k-block 8 (l >> 5) of the 16
    const int frag = (lane & 31) * kPitch + ((lane >> 5) << 3);
    const uint16_t* fA = sA + (wm * TM * 32) * kPitch + frag;
    const uint16_t* fB = sB + (wn * TN * 32) * kPitch + frag;

    // epilogue: C/D layout of the 32x32 MFMA: register r of lane l = row (r & 3) + 8 (r >> 2) + 4 (l >> 5), col l & 31.
    // No compiler-visible load may be pending while the stores are issued: with one (the bias) hipcc put
    // s_waitcnt vmcnt(0) in front of every element's store -- stores count in vmcnt on gfx9, so each of the 64 stores
    // of a thread waited for the previous one to reach L2 (the epilogue was 29 % of the kernel,
    // profiles/r03_gemm_ablation.txt).  The bias is therefore fetched first, waited for explicitly and laundered; tiles
    // that lie inside the matrix store without per-element bounds branches.
    auto epilogue = [&](const Item& it, auto put) {
        float bv[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[j] = 0.0f;
        if (g.bias != nullptr && it.sl == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = it.n0 + (wn * TN + j) * 32 + (lane & 31);
                bv[j] = g.bias[col < g.N ? col : g.N - 1];
            }
            vm_wait<0>();
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bv[j]));
        }
        const bool inside = it.m0 + BM <= g.M && it.n0 + BN <= g.N;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = it.n0 + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const long rbase = it.m0 + (wm * TM + i) * 32 + ((lane >> 5) << 2);
                float* __restrict__ p0 = it.Cb + rbase * g.ldc + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + ((r >> 2) << 3);
                    const float val = acc[i][j][r] + bv[j];
#if SIGMA_GEMM_ABL & 4
                    asm volatile("" :: "v"(val), "v"(p0));
#else
                    if (inside) put(p0 + (long)dr * g.ldc, val);
                    else if (col < g.N && rbase + dr < g.M) put(p0 + (long)dr * g.ldc, val);
#endif
                }
            }
        }
    };

    // Row-contiguous epilogue (round 4): the MFMA accumulator layout gives a lane four ROWS of one column, so the direct
    // epilogue above is 64 dword stores per thread and tile, each wave-instruction touching 2 x 128 bytes (the C stores
    // were 29 % of the in_proj kernel, profiles/r03_gemm_ablation_v2.txt).  Here every 32 x 32 block goes through 4 KB of
    // LDS private to the wave (the operand image is free between two tiles): 16 ds_write_b32 in accumulator order
    // ([row][32 floats]: conflict-free), 4 ds_read_b128 by (row = lane / 8 + 8 q, four columns 4 (lane % 8)), and 4
    // global_store_dwordx4 of 8 x 128 contiguous bytes each.  LDS operations of one wave execute in order, so the block
    // loop needs no wait of its own beyond the data dependency; the workgroup barrier after the tile keeps the next
    // tile's operand stores off the staging areas.  Taken when C rows are 16-byte aligned (N % 4 == 0, ldc % 4 == 0)
    // and the tile is stored or added to plainly (no atomics).
    auto epilogue_rows = [&](const Item& it, bool add) {
        float bv[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[j] = 0.0f;
        if (g.bias != nullptr && it.sl == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = it.n0 + (wn * TN + j) * 32 + (lane & 31);
                bv[j] = g.bias[col < g.N ? col : g.N - 1];
            }
            vm_wait<0>();
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bv[j]));
        }
        float* stage = reinterpret_cast<float*>(smem) + wave * 1056;          // 32 x 32 (33) floats of this wave
        const int wr = ((lane >> 5) << 2) * 32 + (lane & 31);                 // accumulator order: row 4 (l / 32) + ..., col l % 32
        const int rd_row = lane >> 3, rd_col = (lane & 7) << 2;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col0 = it.n0 + (wn * TN + j) * 32;
            const int col = col0 + rd_col;
            if (col0 < g.t_cols) {
                // Transposed blocks (round 6: the x half of SS2D.in_proj leaves the GEMM channel-major, so that the depthwise
                // conv reads it in place and the tiled transpose in front of it -- one read and one write of the
                // activation per block -- is gone).  Staged with a row pitch of 33 floats: the accumulator-order writes
                // stay conflict-free (32 consecutive floats per half wave) and the column reads below hit banks
                // 4 (l % 8) + (l / 8) + 33 k, distinct inside each half wave.  A lane stores four consecutive ROWS of one
                // column: 8 columns x 128 contiguous bytes per store instruction.
                const int wr33 = ((lane >> 5) << 2) * 33 + (lane & 31);
                const int rg = lane & 7, tc = lane >> 3;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) stage[wr33 + ((r & 3) + ((r >> 2) << 3)) * 33] = acc[i][j][r] + bv[j];
                    const long rbase = it.m0 + (wm * TM + i) * 32 + 4 * rg;
                    f32x4_t v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float* src = stage + (4 * rg) * 33 + tc + 8 * q;
                        v[q] = f32x4_t{src[0], src[33], src[66], src[99]};
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = col0 + tc + 8 * q;
                        if (c < g.t_cols && rbase < g.M) *reinterpret_cast<f32x4_t*>(g.Ct + (long)c * g.ldct + rbase) = v[q];
                    }
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) stage[wr + ((r & 3) + ((r >> 2) << 3)) * 32] = acc[i][j][r] + bv[j];
                const long rbase = it.m0 + (wm * TM + i) * 32 + rd_row;
                f32x4_t v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4_t*>(stage + (rd_row + 8 * q) * 32 + rd_col);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4_t* __restrict__ dst = reinterpret_cast<f32x4_t*>(it.Cb + (rbase + 8 * q) * g.ldc + (col - g.t_cols));
                    if (col < g.N && rbase + 8 * q < g.M) {
                        if (add) {
                            const f32x4_t old = *dst;
                            *dst = old + v[q];
                        } else {
                            *dst = v[q];
                        }
                    }
                }
            }
        }
    };
    const bool rows_ok = (g.N & 3) == 0 && (g.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 && (g.sC & 3) == 0 &&
                         !(SIGMA_GEMM_ABL & 4) && SIGMA_GEMM_ROW_EPILOGUE;

#if SIGMA_GEMM_PROF
    unsigned long long prof_[16] = {0};
    unsigned long long tp_ = __builtin_readcyclecounter();
    const unsigned long long t_begin_ = tp_;
#endif
    while (c_on) {
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
            if (!c_on) break;
            GPROF(7)
            // the loads of slot u are complete when at most the loads of the NEWER stages are outstanding (in-order return)
            {
                int newer = 0;
#pragma unroll
                for (int d = 0; d < kDepth; ++d) if (d != u && rem_[d] > 0) ++newer;
                if (kDepth >= 3 && newer >= 2) vm_wait<2 * LPS>();
                else if (kDepth >= 2 && newer == 1) vm_wait<LPS>();
                else vm_wait<0>();
#pragma unroll
                for (int i = 0; i < LoaderA::NV; ++i) asm volatile("" : "+v"(va[u][i]));      // uses stay behind the wait
#pragma unroll
                for (int i = 0; i < LoaderB::NV; ++i) asm volatile("" : "+v"(vb[u][i]));
            }
            GPROF(0)
            const int rem_u = rem_[u];
#if SIGMA_GEMM_ABL & 16
            asm volatile("" :: "v"(va[u][0]), "v"(vb[u][0]));
#else
            la.template store<P>(va[u], rem_u, sA, BM * kPitch);
            lb_.template store<P>(vb[u], rem_u, sB, BN * kPitch);
#if SIGMA_GEMM_PROF
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            GPROF(1)
#endif
            lds_barrier();
            GPROF(2)
#endif
            produce(u, va[u], vb[u]);                  // refill the slot just written to LDS: step s + kDepth
            GPROF(3)
            // ALL fragments of the k-step are requested up front (16 ds_read_b128 at 128 x 128, 64 VGPRs), pinned in front of
            // the MFMAs: left to its register heuristics hipcc re-used eight fragment registers and put s_waitcnt
            // lgkmcnt(0) straight after a read three times per k-block -- the MFMA phase ran at ~55 clocks per MFMA
            // against 32 (round 6, ISA + the no-memory ablation build).  LDS returns in order, so the compiler's counted
            // waits release each MFMA as soon as its own two fragments are there.
            // (three pieces per operand: one k-block at a time -- 24 fragments would not fit beside the accumulators)
            constexpr int KSN = (SIGMA_GEMM_ABL & 1) ? 0 : kBK / 16;
            constexpr int KH = P == 2 ? (KSN > 0 ? KSN : 1) : 1;          // k-blocks whose fragments are in registers at once
#pragma unroll
            for (int k0 = 0; k0 < KSN; k0 += KH) {
                bf16x8_t fa[KH][P][TM], fb[KH][P][TN];
#pragma unroll
                for (int kh = 0; kh < KH; ++kh) {
#pragma unroll
                    for (int q = 0; q < P; ++q) {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            fa[kh][q][i] = *reinterpret_cast<const bf16x8_t*>(fA + q * BM * kPitch + i * 32 * kPitch + (k0 + kh) * 16);
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            fb[kh][q][j] = *reinterpret_cast<const bf16x8_t*>(fB + q * BN * kPitch + j * 32 * kPitch + (k0 + kh) * 16);
                    }
                }
#if SIGMA_GEMM_FRAG_PIN
                if (P == 2) __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                for (int kh = 0; kh < KH; ++kh) {
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
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kh][qa][i], fb[kh][sum - qa][j], acc[i][j], 0, 0, 0);
                }
            }
            GPROF(4)
#if !(SIGMA_GEMM_ABL & 16)
            lds_barrier();
#endif
            GPROF(5)
#if SIGMA_GEMM_PROF
            prof_[15] += 1;
#endif
            ck += kBK;
            if (ck >= cit.kend) {                      // tile (slice) complete
                if (g.mode != 2 && rows_ok) {
                    epilogue_rows(cit, g.mode == 1);
                    lds_barrier();                     // staging areas free before the next tile's operands are written
                } else if (g.mode == 0) epilogue(cit, [](float* dst, float v) { *dst = v; });
                else if (g.mode == 1) epilogue(cit, [](float* dst, float v) { *dst += v; });
                else epilogue(cit, [](float* dst, float v) { atomicAdd(dst, v); });
#if SIGMA_GEMM_PROF
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // charge the store drain to the epilogue
                prof_[13] += 1;
#endif
                GPROF(6)
                c_id += gridDim.x;
                c_on = c_id < total;
                if (c_on) { decode(c_id, cit); ck = cit.kbeg; init_acc(cit); }
            }
        }
    }
#if SIGMA_GEMM_PROF
    prof_[14] = __builtin_readcyclecounter() - t_begin_;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) atomicAdd(&g_gemm_prof[i], prof_[i]);
    }
#endif
}

template <int BM, int BN, int WM, int WN, bool A_KS, bool B_KS, int P, bool RES>
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
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_split3_kernel<BM, BN, WM, WN, A_KS, B_KS, P, RES>, 256, 0) != hipSuccess || n < 1) n = 2;
        per_cu = n > 4 ? 4 : n;
#ifdef SIGMA_GEMM_MAX_PER_CU
        if (per_cu > SIGMA_GEMM_MAX_PER_CU) per_cu = SIGMA_GEMM_MAX_PER_CU;     // A/B builds: resident workgroups per CU capped
#endif
    }
    long grid = 256L * per_cu;
    if (grid > items) grid = items;
    hipLaunchKernelGGL((gemm_split3_kernel<BM, BN, WM, WN, A_KS, B_KS, P, RES>), dim3((unsigned)grid), dim3(256), 0, stream, ga);
    return hipGetLastError();
}

// tile width for N columns: 128 unless a narrower tile wastes less ((N = 96, 192: 96-wide tiles are exact)
int pick_bn(int N) {
#ifdef SIGMA_GEMM_FORCE_BN
    return SIGMA_GEMM_FORCE_BN;                         // A/B builds: one tile width for every problem
#endif
    if (N % 128 == 0) return 128;
    if (N % 96 == 0) return 96;
    if (N <= 64) return 64;
    if (N <= 96) return 96;
    // padded columns of each candidate
    const int w128 = (N + 127) / 128 * 128 - N, w96 = (N + 95) / 96 * 96 - N;
    return w96 < w128 ? 96 : 128;
}

template <bool A_KS, bool B_KS, int P, bool RES>
hipError_t launch_r(GemmArgs& g, int batch, hipStream_t stream) {
    const int bn = pick_bn(g.N);
    g.ntm = (int)((g.M + 127) / 128);
    g.ntn = (g.N + bn - 1) / bn;
    if (bn == 128) return launch_cfg<128, 128, 2, 2, A_KS, B_KS, P, RES>(g, batch, stream);
    if (bn == 96) return launch_cfg<128, 96, 4, 1, A_KS, B_KS, P, RES>(g, batch, stream);
    return launch_cfg<128, 64, 2, 2, A_KS, B_KS, P, RES>(g, batch, stream);
}

// epilogue addends exist for the two-piece nt / nn kernels (the forward and input-gradient GEMMs of the model)
template <bool A_KS, bool B_KS, int P>
hipError_t launch_p(GemmArgs& g, int batch, hipStream_t stream) {
    if constexpr (P == 2 && !(A_KS && B_KS)) {
        if (g.R != nullptr) return launch_r<A_KS, B_KS, P, true>(g, batch, stream);
    }
    if (g.R != nullptr) return hipErrorInvalidValue;
    return launch_r<A_KS, B_KS, P, false>(g, batch, stream);
}

template <bool A_KS, bool B_KS>
hipError_t launch_any(GemmArgs& g, int batch, int pieces, hipStream_t stream) {
    return pieces == 3 ? launch_p<A_KS, B_KS, 3>(g, batch, stream) : launch_p<A_KS, B_KS, 2>(g, batch, stream);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Second stage of a sliced / shared-output product: C[c] (+)= sum over its nparts compact (M x N) partial results, in a fixed
// order (deterministic, unlike the atomic sums it replaces).  VEC: four columns per thread (N % 4 == 0, aligned rows).
// Block = 64 elements x PG part groups: lanes run along the elements (coalesced 16-byte loads), part group y sums the parts
// y, y + PG, ... (two loads in flight), the groups meet in LDS -- a product with few output elements and hundreds of parts
// (the 96 x 96 weight gradient of the decoder's last linear: 2304 vectors x 512 parts) still fills the chip.
template <bool VEC>
__global__ void __launch_bounds__(1024)
reduce_parts_kernel(const float* __restrict__ ws, float* __restrict__ C, long M, int N, long ldc, long sC, int nparts, long part_stride, int add) {
    constexpr int W = VEC ? 4 : 1;
    typedef float vec_t __attribute__((ext_vector_type(W)));
    __shared__ vec_t red[16][64];
    const int PG = blockDim.y, y = threadIdx.y;
    const long n_el = M * N / W;
    const long e = (long)blockIdx.x * 64 + threadIdx.x;
    const bool on = e < n_el;
    const int c = blockIdx.y;
    const long flat = (on ? e : 0) * W;
    const long row = flat / N;
    const int col = (int)(flat - row * N);
    const float* __restrict__ src = ws + (long)c * nparts * part_stride + flat;
    vec_t acc = {};
    if (on) {
        int p = y;
        for (; p + PG < nparts; p += 2 * PG) {
            const vec_t a0 = *reinterpret_cast<const vec_t*>(src + (long)p * part_stride);
            const vec_t a1 = *reinterpret_cast<const vec_t*>(src + (long)(p + PG) * part_stride);
            acc += a0 + a1;
        }
        if (p < nparts) acc += *reinterpret_cast<const vec_t*>(src + (long)p * part_stride);
    }
    if (PG > 1) {
        red[y][threadIdx.x] = acc;
        __syncthreads();
        if (y == 0)
            for (int q = 1; q < PG; ++q) acc += red[q][threadIdx.x];
    }
    if (on && y == 0) {
        vec_t* __restrict__ dst = reinterpret_cast<vec_t*>(C + (long)c * sC + row * ldc + col);
        if (add) acc += *dst;
        *dst = acc;
    }
}

// bytes of scratch the two-stage sum of a launch needs (0: a single part per output -- no second stage)
int64_t parts_bytes(const GemmArgs& g, int batch) {
    const int per_out = g.slices * (g.c_mod > 0 ? batch / g.c_mod : 1);
    if (per_out <= 1) return 0;
    if (g.c_mod > 0 && batch % g.c_mod != 0) return 0;          // ragged groups keep the atomic path
    const long outs = g.c_mod > 0 ? g.c_mod : batch;
    return (int64_t)outs * per_out * g.M * g.N * (int64_t)sizeof(float);
}

// A launch whose items do not each own their output (reduction slices, problems sharing an output): with enough scratch
// (ws / ws_bytes from the caller) as plain partial stores + reduce_parts_kernel, otherwise with fp32 atomics into C (which
// the caller then zero-filled or accumulates into).  Round 6: the 64 dword atomics per thread and item were HALF of a
// weight-gradient launch (profiles/r06_gemm_phases.jsonl) and its shape-independent floor of ~50 us.
template <bool A_KS, bool B_KS>
hipError_t launch_summed(GemmArgs& g, int batch, int pieces, void* ws, int64_t ws_bytes, bool accumulate, hipStream_t stream) {
    const int64_t need = parts_bytes(g, batch);
    if (need == 0 || ws == nullptr || ws_bytes < need || !aligned16(ws)) {
        g.mode = 2;
        g.nparts = 0;
        return launch_any<A_KS, B_KS>(g, batch, pieces, stream);
    }
    float* const C = g.C;
    const long ldc = g.ldc, sC = g.sC;
    const int outs = g.c_mod > 0 ? g.c_mod : batch;
    g.nparts = g.slices * (g.c_mod > 0 ? batch / g.c_mod : 1);
    g.part_stride = g.M * g.N;
    g.C = static_cast<float*>(ws);
    g.ldc = g.N;
    g.sC = 0;
    g.mode = 0;
    hipError_t e = launch_any<A_KS, B_KS>(g, batch, pieces, stream);
    if (e != hipSuccess) return e;
    const bool vec = (g.N & 3) == 0 && (ldc & 3) == 0 && (sC & 3) == 0 && aligned16(C);
    const long n_el = g.M * g.N / (vec ? 4 : 1);
    const long blocks = (n_el + 63) / 64;
    int pg = 1;                                          // part groups per block: until ~2 waves per SIMD are in flight
    while (pg < 16 && 2 * pg <= g.nparts && blocks * outs * pg < 2048) pg *= 2;
    const dim3 grid((unsigned)blocks, (unsigned)outs), block(64, pg);
    if (vec) hipLaunchKernelGGL(reduce_parts_kernel<true>, grid, block, 0, stream, (const float*)g.C, C, g.M, g.N, ldc, sC, g.nparts, g.part_stride, accumulate ? 1 : 0);
    else hipLaunchKernelGGL(reduce_parts_kernel<false>, grid, block, 0, stream, (const float*)g.C, C, g.M, g.N, ldc, sC, g.nparts, g.part_stride, accumulate ? 1 : 0);
    return hipGetLastError();
}

int fill_common(const sigma_gemm_params* p, GemmArgs& g) {
    if (!p || !p->A || !p->Bt || !p->C) return SIGMA_OPS_ERR_ARG;
    if (p->M < 0 || p->N < 0 || p->K < 0 || p->batch < 0) return SIGMA_OPS_ERR_ARG;
    if (p->pieces != 0 && p->pieces != 2 && p->pieces != 3) return SIGMA_OPS_ERR_ARG;
    if (!aligned16(p->A) || !aligned16(p->Bt) || p->lda % 4 != 0 || p->ldb % 4 != 0) return SIGMA_OPS_ERR_ARG;
    // a tile is addressed with 32-bit byte offsets from a wave-uniform base: 127 rows x ld x 4 bytes must fit
    if (p->lda <= 0 || p->ldb <= 0 || p->lda > (1L << 22) || p->ldb > (1L << 22)) return SIGMA_OPS_ERR_ARG;
    if (p->batch > 1 && (p->strideA % 4 != 0 || p->strideB % 4 != 0)) return SIGMA_OPS_ERR_ARG;
    g.A = p->A; g.B = p->Bt; g.C = p->C; g.bias = p->bias;
    g.lda = p->lda; g.ldb = p->ldb; g.ldc = p->ldc;
    g.sA = p->strideA; g.sB = p->strideB; g.sC = p->strideC;
    g.slices = 1; g.a_mod = 0; g.c_mod = 0;
    g.mode = p->accumulate ? 1 : 0;
    g.R = p->residual; g.R2 = p->residual ? p->residual2 : nullptr; g.ldr = p->ldr; g.sR = p->strideR;
    g.Ct = nullptr; g.ldct = 0; g.t_cols = 0;
    g.nparts = 0; g.part_stride = 0;
    if (p->workspace_bytes < 0 || (p->workspace_bytes > 0 && !p->workspace)) return SIGMA_OPS_ERR_ARG;
    if (!p->residual && p->residual2) return SIGMA_OPS_ERR_ARG;
    if (p->residual && (p->ldr <= 0 || p->M * p->ldr >= 0x7fffffffL)) return SIGMA_OPS_ERR_ARG;
    if (p->c_mod < 0 || p->reserved != 0) return SIGMA_OPS_ERR_ARG;
    return SIGMA_OPS_OK;
}

}  // namespace
}  // namespace sigma

namespace sigma {
namespace {

// The three forms: argument checks and launch geometry (shared by the entry points and sigma_gemm_workspace_bytes).
// `summed`: the items of the launch do not each own their output (launch_summed); `empty`: nothing to do.
struct Planned { GemmArgs g; int batch; bool summed; bool empty; };

int plan_nt(const sigma_gemm_params* p, Planned& pl) {
    GemmArgs& g = pl.g;
    int rc = fill_common(p, g);
    if (rc) return rc;
    if (p->K % 4 != 0) return SIGMA_OPS_ERR_ARG;
    pl.batch = p->batch > 0 ? p->batch : 1;
    pl.summed = false;
    pl.empty = p->M == 0 || p->N == 0;
    if (pl.empty) return SIGMA_OPS_OK;
    g.M = p->M; g.N = p->N; g.K = p->K;
    if (p->K == 0) return SIGMA_OPS_ERR_ARG;
    g.slice_k = (p->K + 31) / 32 * 32;
    g.a_mod = p->a_mod;
    if (p->c_mod > 0 && pl.batch > p->c_mod) {       // several problems per output: summed
        if (p->residual || p->bias) return SIGMA_OPS_ERR_ARG;
        g.c_mod = p->c_mod; pl.summed = true;
    } else if (p->c_mod > 0) g.c_mod = p->c_mod;
    if (p->t_cols != 0) {                                // transposed column range: row-contiguous epilogue, plain stores only
        if (p->t_cols < 0 || p->t_cols > p->N || p->t_cols % 32 != 0 || !p->Ct || !aligned16(p->Ct) || p->ldct % 4 != 0 ||
            p->ldct < p->M || p->M % 4 != 0 || p->N % 4 != 0 || p->ldc % 4 != 0 || !aligned16(p->C) || pl.batch != 1 ||
            p->accumulate || p->residual || pl.summed || !SIGMA_GEMM_ROW_EPILOGUE)
            return SIGMA_OPS_ERR_ARG;
        g.Ct = p->Ct; g.ldct = p->ldct; g.t_cols = p->t_cols;
    }
    return SIGMA_OPS_OK;
}

int plan_nn(const sigma_gemm_params* p, Planned& pl) {
    // C = A B with B = (K, N) row-major (params->Bt, row stride ldb): the B operand's reduction index is its slow index
    GemmArgs& g = pl.g;
    int rc = fill_common(p, g);
    if (rc) return rc;
    if (p->K % 4 != 0 || p->N % 4 != 0) return SIGMA_OPS_ERR_ARG;
    pl.batch = p->batch > 0 ? p->batch : 1;
    pl.summed = false;
    pl.empty = p->M == 0 || p->N == 0;
    if (pl.empty) return SIGMA_OPS_OK;
    g.M = p->M; g.N = p->N; g.K = p->K;
    if (p->K == 0) return SIGMA_OPS_ERR_ARG;
    g.slice_k = (p->K + 31) / 32 * 32;
    g.a_mod = p->a_mod;
    if (p->c_mod > 0 && pl.batch > p->c_mod) {       // several problems per output: summed
        if (p->residual || p->bias) return SIGMA_OPS_ERR_ARG;
        g.c_mod = p->c_mod; pl.summed = true;
    } else if (p->c_mod > 0) g.c_mod = p->c_mod;
    if (p->k_slices == 1 && !pl.summed && !p->residual && !p->bias && pl.batch == 1) {
        // few output tiles, long reduction: reduction slices, one round of resident workgroups (as tn)
        const int bn = pick_bn(g.N);
        const long tiles = ((g.M + 127) / 128) * ((g.N + bn - 1) / bn);
        const long steps = (p->K + 31) / 32;
        long want = 512 / (tiles > 0 ? tiles : 1);
        if (want > steps / 8) want = steps / 8;
        if (want > 1) {
            const long per = (steps + want - 1) / want;
            g.slice_k = (int)(per * 32);
            g.slices = (int)((steps + per - 1) / per);
            if (g.slices > 1) pl.summed = true;
        }
    }
    return SIGMA_OPS_OK;
}

int plan_tn(const sigma_gemm_params* p, Planned& pl) {
    // C (N_out x K_in) (+)= A^T B with A = (M, N_out), Bt = (M, K_in): both operands have the reduction (token) index
    // as the slow memory index.  Kernel view: rows of C = N_out, columns = K_in, reduction = M.
    GemmArgs& g = pl.g;
    int rc = fill_common(p, g);
    if (rc) return rc;
    if (p->N % 4 != 0 || p->K % 4 != 0) return SIGMA_OPS_ERR_ARG;
    if (p->c_mod != 0 || p->residual) return SIGMA_OPS_ERR_ARG;      // nt / nn only
    pl.batch = p->batch > 0 ? p->batch : 1;
    pl.summed = false;
    pl.empty = p->N == 0 || p->K == 0 || p->M == 0;
    if (pl.empty) return SIGMA_OPS_OK;
    g.M = p->N; g.N = p->K; g.K = (int)p->M;
    if (p->M > 0x7fffff00L) return SIGMA_OPS_ERR_ARG;
    // reduction slices: each a multiple of 32 tokens and at least 8 k-steps long, and no more items than ONE round of
    // resident workgroups (2 per CU): 768 items on 512 slots ran 1.5 rounds (round 6: 144 -> 132 us at enc_s2_in_proj,
    // 92 -> 71 us at the small shapes)
    const int bn = pick_bn(g.N);
    const long tiles = (long)pl.batch * ((g.M + 127) / 128) * ((g.N + bn - 1) / bn);
    const long steps = (p->M + 31) / 32;
    static const long target_items = [] { const char* e = getenv("SIGMA_GEMM_TN_ITEMS"); const long v = e ? atol(e) : 0; return v > 0 ? v : 512L; }();
    long want = target_items / tiles;
    if (want > steps / 8) want = steps / 8;
    if (want < 1) want = 1;
    const long per = (steps + want - 1) / want;
    g.slice_k = (int)(per * 32);
    g.slices = (int)((steps + per - 1) / per);
    pl.summed = g.slices > 1;
    return SIGMA_OPS_OK;
}

template <bool A_KS, bool B_KS>
int run_planned(const sigma_gemm_params* p, Planned& pl, void* stream) {
    if (pl.empty) return SIGMA_OPS_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = pl.summed ? launch_summed<A_KS, B_KS>(pl.g, pl.batch, p->pieces, p->workspace, p->workspace_bytes, p->accumulate != 0, s)
                             : launch_any<A_KS, B_KS>(pl.g, pl.batch, p->pieces, s);
    return e == hipSuccess ? SIGMA_OPS_OK : SIGMA_OPS_ERR_LAUNCH;
}

}  // namespace
}  // namespace sigma

extern "C" int sigma_gemm_nt_split3(const sigma_gemm_params* p, void* stream) {
    sigma::Planned pl;
    const int rc = sigma::plan_nt(p, pl);
    return rc ? rc : sigma::run_planned<false, false>(p, pl, stream);
}

extern "C" int sigma_gemm_nn_split3(const sigma_gemm_params* p, void* stream) {
    sigma::Planned pl;
    const int rc = sigma::plan_nn(p, pl);
    return rc ? rc : sigma::run_planned<false, true>(p, pl, stream);
}

extern "C" int sigma_gemm_tn_split3(const sigma_gemm_params* p, void* stream) {
    sigma::Planned pl;
    const int rc = sigma::plan_tn(p, pl);
    return rc ? rc : sigma::run_planned<true, true>(p, pl, stream);
}

extern "C" int64_t sigma_gemm_workspace_bytes(const sigma_gemm_params* p, int form) {
    sigma::Planned pl;
    const int rc = form == 0 ? sigma::plan_nt(p, pl) : form == 1 ? sigma::plan_nn(p, pl) : form == 2 ? sigma::plan_tn(p, pl) : SIGMA_OPS_ERR_ARG;
    if (rc) return -1;
    return (pl.empty || !pl.summed) ? 0 : sigma::parts_bytes(pl.g, pl.batch);
}

// Self test: the three kernel forms on operands whose products and sums are exact in fp32 (integers of small magnitude:
// the bf16 split is then exact too), compared with host arithmetic; ragged sizes, so that partial tiles, the partial
// k-step and the row-contiguous epilogue all run.  Synchronises `stream`.  0 = pass; 1..3 = nt / nn / tn differ;
// negative = a HIP call failed.
extern "C" int sigma_gemm_selftest(void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int M = 200, N = 72, K = 44;
    const size_t nA = (size_t)M * K, nB = (size_t)N * K, nC = (size_t)M * N, nG = (size_t)M * N, nW = (size_t)N * K, nX = (size_t)M * K;
    float *hA = new float[nA], *hB = new float[nB], *hG = new float[nG], *hC = new float[nC], *hX = new float[nX], *hW = new float[nW];
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)((int)((st >> 16) % 17) - 8); };
    for (size_t i = 0; i < nA; ++i) hA[i] = rnd();
    for (size_t i = 0; i < nB; ++i) hB[i] = rnd();
    for (size_t i = 0; i < nG; ++i) hG[i] = rnd();
    float *dA = nullptr, *dB = nullptr, *dG = nullptr, *dC = nullptr, *dX = nullptr, *dW = nullptr;
    int rc = 0;
    auto ok = [&](hipError_t e) { if (e != hipSuccess && rc == 0) rc = -1; return e == hipSuccess; };
    if (ok(hipMalloc(&dA, nA * 4)) && ok(hipMalloc(&dB, nB * 4)) && ok(hipMalloc(&dG, nG * 4)) && ok(hipMalloc(&dC, nC * 4)) &&
        ok(hipMalloc(&dX, nX * 4)) && ok(hipMalloc(&dW, nW * 4)) &&
        ok(hipMemcpyAsync(dA, hA, nA * 4, hipMemcpyHostToDevice, s)) && ok(hipMemcpyAsync(dB, hB, nB * 4, hipMemcpyHostToDevice, s)) &&
        ok(hipMemcpyAsync(dG, hG, nG * 4, hipMemcpyHostToDevice, s)) && ok(hipMemsetAsync(dW, 0, nW * 4, s))) {
        sigma_gemm_params p{};
        p.batch = 1; p.pieces = 2;
        // nt: C = A B^T
        p.M = M; p.N = N; p.K = K; p.A = dA; p.Bt = dB; p.C = dC; p.lda = K; p.ldb = K; p.ldc = N;
        if (sigma_gemm_nt_split3(&p, stream) != 0) rc = -2;
        // nn: X = G B   (G (M, N), B (N, K) row-major)
        p.M = M; p.N = K; p.K = N; p.A = dG; p.Bt = dB; p.C = dX; p.lda = N; p.ldb = K; p.ldc = K;
        if (rc == 0 && sigma_gemm_nn_split3(&p, stream) != 0) rc = -2;
        // tn: W = G^T A  (reduction over the M rows)
        p.M = M; p.N = N; p.K = K; p.A = dG; p.Bt = dA; p.C = dW; p.lda = N; p.ldb = K; p.ldc = K; p.accumulate = 1;
        if (rc == 0 && sigma_gemm_tn_split3(&p, stream) != 0) rc = -2;
        if (rc == 0 && ok(hipMemcpyAsync(hC, dC, nC * 4, hipMemcpyDeviceToHost, s)) && ok(hipMemcpyAsync(hX, dX, nX * 4, hipMemcpyDeviceToHost, s)) &&
            ok(hipMemcpyAsync(hW, dW, nW * 4, hipMemcpyDeviceToHost, s)) && ok(hipStreamSynchronize(s))) {
            for (int m = 0; m < M && rc == 0; ++m)
                for (int n = 0; n < N; ++n) {
                    float acc = 0.0f;
                    for (int k = 0; k < K; ++k) acc += hA[(size_t)m * K + k] * hB[(size_t)n * K + k];
                    if (hC[(size_t)m * N + n] != acc) { rc = 1; break; }
                }
            for (int m = 0; m < M && rc == 0; ++m)
                for (int k = 0; k < K; ++k) {
                    float acc = 0.0f;
                    for (int n = 0; n < N; ++n) acc += hG[(size_t)m * N + n] * hB[(size_t)n * K + k];
                    if (hX[(size_t)m * K + k] != acc) { rc = 2; break; }
                }
            for (int n = 0; n < N && rc == 0; ++n)
                for (int k = 0; k < K; ++k) {
                    float acc = 0.0f;
                    for (int m = 0; m < M; ++m) acc += hG[(size_t)m * N + n] * hA[(size_t)m * K + k];
                    if (hW[(size_t)n * K + k] != acc) { rc = 3; break; }
                }
        }
    }
    (void)hipStreamSynchronize(s);      // every exit path: copies from / into the host buffers may still be enqueued
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dG); (void)hipFree(dC); (void)hipFree(dX); (void)hipFree(dW);
    delete[] hA; delete[] hB; delete[] hG; delete[] hC; delete[] hX; delete[] hW;
    return rc;
}
