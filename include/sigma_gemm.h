/*
 * sigma_gemm.h -- C ABI of the split-operand bf16 MFMA GEMMs in libsigma_hip.so (gfx950).
 *
 * The projections of the hot path are nn.Linear calls of the reference -- SS2D.in_proj / out_proj
 * (models/encoders/vmamba.py:1067-1089, parameters :679-726), PatchMerging2D.reduction (:612-636), the
 * CroMB / ConMB in / out projections (:1588-1640, :1134-1284) and the decoder linears
 * (models/decoders/MambaDecoder.py:12-97) -- i.e. fp32 GEMMs  y = x W^T (+ b)  with x (tokens, in) and
 * W (out, in) both contiguous along the reduction dimension.  On MI355X an fp32 GEMM is bound by the fp32 MFMA
 * rate (157 TFLOP/s, 1/16 of bf16); bf16 operands alone miss the 1e-3 logit tolerance.  These entry points
 * compute the fp32 GEMM from fp32 operands with
 *
 *     a = a_hi + a_lo,  a_hi = bf16(a), a_lo = bf16(a - a_hi)      (16 significant bits)
 *     a * b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi                   (dropped a_lo b_lo ~ 2^-16 relative)
 *
 * as three v_mfma_f32_32x32x16_bf16 per tile with fp32 accumulation; the split happens in registers on the way
 * from global memory to LDS (no operand images in HBM).  Error against an fp64 product ~4e-6 rms (the fp32
 * library GEMM: ~1e-6), see profiles/.
 *
 * Conventions as sigma_scan.h: device pointers, fp32, the callee enqueues on `stream` (hipStream_t as void*) of
 * the current device, never allocates, never synchronises; returns 0 or a SIGMA_OPS_ERR_* code (sigma_ops.h).
 */
#ifndef SIGMA_GEMM_H_
#define SIGMA_GEMM_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sigma_gemm_params {
    int64_t M;             /* rows of A and C (tokens)                                              */
    int32_t N;             /* columns of C = rows of Bt                                             */
    int32_t K;             /* reduction length; K % 4 == 0                                          */
    const float *A;        /* nt: (M, K), element (m, k) at A[m * lda + k]; 16-byte aligned rows    */
    const float *Bt;       /* nt: (N, K), element (n, k) at Bt[n * ldb + k]: C = A * Bt^T           */
    float *C;              /* (M, N), element (m, n) at C[m * ldc + n]                              */
    const float *bias;     /* (N) added to every row, or NULL                                       */
    int64_t lda, ldb, ldc; /* row strides in floats; lda % 4 == 0, ldb % 4 == 0, both <= 2^22 (32-bit tile
                              offsets inside the kernels)                                           */
    int32_t accumulate;    /* 1: C += A * Bt^T (+ bias), 0: C = ...                                 */
    int32_t batch;         /* >= 1: `batch` independent problems, operand b at base + b * stride    */
    int64_t strideA, strideB, strideC;   /* batch strides in floats (0: shared by every problem)    */
    int32_t a_mod;         /* > 0: problem z reads A + (z % a_mod) * strideA (nt / nn): a weight stack
                              shared by groups of problems, e.g. the 2 memory orders x B images of x_proj */
    int32_t pieces;        /* bf16 pieces per fp32 operand element: 0 or 2 = (hi, lo), three MFMAs per block, ~4e-6 rms
                              error; 3 = (hi, mid, lo), six MFMAs, ~1e-6 = the accuracy of an fp32 GEMM          */
    /* ABI 8: fused epilogue inputs and shared outputs (nt / nn only)                                            */
    int32_t c_mod;         /* > 0: problem z writes C + (z % c_mod) * strideC and the problems that share an output are
                              SUMMED into it with fp32 atomics (the caller zero-fills C, or passes accumulate = 1):
                              weight gradients of a stacked projection summed over the batch, e.g. d x_proj_weight =
                              sum_b dp[b] xs[b]^T (vmamba.py:193-196 under autograd)                                */
    int32_t reserved;      /* 0 */
    const float *residual; /* (M, N) per problem, element (m, n) at residual[z * strideR + m * ldr + n], or NULL:
                              C = A B (+ bias) + residual (+ residual2) -- the residual stream of a block added in the
                              GEMM epilogue (x + out_proj(y), vmamba.py:1716-1722) or the two per-direction input
                              gradients of the scan joined with the projection's (dxs = W^T dp + du[dir] + du[dir^1]) */
    const float *residual2;/* second addend with the same ldr / strideR, or NULL                                       */
    int64_t ldr, strideR;
    /* ABI 10 (round 6) */
    float *Ct;             /* nt only, with t_cols > 0: columns [0, t_cols) of the product are written TRANSPOSED,
                              element (m, n) at Ct[n * ldct + m], and C receives only the columns n >= t_cols, element
                              (m, n) at C[m * ldc + n - t_cols].  SS2D.in_proj (vmamba.py:1067-1071) thus hands its x half
                              to the depthwise convolution channel-major, with no transposing pass in between.
                              Needs t_cols % 32 == 0, M % 4 == 0, ldct % 4 == 0, 16-byte aligned Ct and C, N % 4 == 0,
                              ldc % 4 == 0, batch <= 1, accumulate = 0, no residual                                       */
    int64_t ldct;
    int32_t t_cols;
    int32_t k_slices;      /* nn only: 1 = the reduction may be cut into slices run by different workgroups and summed
                              into C with fp32 atomics, as tn does (the CALLER zero-fills C, or passes accumulate = 1):
                              products with few output tiles and a long reduction, e.g. the weight gradient
                              dW = dX^T X with dX held channel-major                                                       */
    void *workspace;       /* scratch for launches whose work items do not each own their output -- the reduction slices of
                              tn (and of nn with k_slices), the problems sharing an output under c_mod: with at least
                              sigma_gemm_workspace_bytes() bytes (16-byte aligned) every item stores its partial result
                              plainly and a second kernel sums the parts of each output in a fixed order into C (C = or
                              C += per `accumulate`; no zero fill, deterministic).  NULL / too small: the parts are added
                              into C with fp32 atomics instead, and the CALLER zero-fills C or passes accumulate = 1
                              (the atomics were half of a weight-gradient launch: DESIGN.md 4.5, round 6)                */
    int64_t workspace_bytes;
} sigma_gemm_params;

/*   sigma_gemm_nt_split3
 *       C[m][n] (+)= sum_k A[m][k] * Bt[n][k] (+ bias[n])      -- nn.Linear forward (A = x, Bt = weight) and its
 *       input gradient (A = dy, Bt = weight^T made contiguous by the caller: the weights are small).      */
int sigma_gemm_nt_split3(const sigma_gemm_params *params, void *stream);

/*   sigma_gemm_nn_split3
 *       C[m][n] (+)= sum_k A[m][k] * B[k][n] (+ bias[n])       -- B = params->Bt is (K, N) row-major with row stride
 *       ldb (N % 4 == 0): nn.Linear input gradient dx = dy W without a transposed weight copy (A = dy, B = W), and
 *       the x_proj einsum on channels-first activations, p = W_stack x (A = weights (rows, d), B = x (d, L);
 *       vmamba.py:193-196).                                                                              */
int sigma_gemm_nn_split3(const sigma_gemm_params *params, void *stream);

/*   sigma_gemm_tn_split3
 *       C[i][j] (+)= sum_m A[m][i] * Bt[m][j]                  -- nn.Linear weight gradient dW = dy^T x:
 *       A = dy (M, N_out) with lda, Bt = x (M, K_in) with ldb, C = dW (N_out, K_in); the reduction runs over the
 *       M tokens (params->M), params->N = N_out (rows of C), params->K = K_in (columns of C).  The token
 *       dimension is cut into slices run by different workgroups.  With params->workspace (see there) the slices'
 *       partial tiles are stored and summed by a second kernel (C = or C +=, deterministic); without it they are
 *       summed with fp32 atomics into C, which the CALLER then zero-fills unless accumulate = 1 (run-to-run
 *       differences at rounding level, like the reference's atomicAdd gradients,
 *       selective_scan_bwd_kernel.cuh:214-231).                                                             */
int sigma_gemm_tn_split3(const sigma_gemm_params *params, void *stream);

/*   sigma_gemm_workspace_bytes
 *       bytes of params->workspace with which the launch described by `params` (form 0 = nt, 1 = nn, 2 = tn) sums its
 *       partial results in two stages; 0 = every work item owns its output (no scratch needed); -1 = bad arguments.
 *       params->workspace / workspace_bytes themselves are ignored by the query.                                     */
int64_t sigma_gemm_workspace_bytes(const sigma_gemm_params *params, int form);

/*   sigma_gemm_selftest
 *       runs the three forms on a small ragged problem whose products are exact in fp32 and compares with host
 *       arithmetic (the operand loads of the kernels are hand-counted inline assembly: a toolchain that schedules them
 *       differently fails here instead of in a model).  Allocates and frees its own buffers, synchronises `stream`.
 *       0 = pass, 1 / 2 / 3 = nt / nn / tn differ, negative = a HIP call failed.  sigma_amd/gemm.py calls it once per
 *       process and device before the first GEMM.                                                              */
int sigma_gemm_selftest(void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SIGMA_GEMM_H_ */
