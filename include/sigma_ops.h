/*
 * sigma_ops.h -- C ABI of the fused layout / stencil operators around the scan in libsigma_hip.so.
 *
 * These replace torch op sequences of the reference's SS2D block (models/encoders/vmamba.py), they
 * have no counterpart in the reference's native code (which only has the selective-scan kernels):
 *
 *   sigma_dwconv3x3_silu_fwd / _bwd
 *       SS2D.forward, vmamba.py:1071-1072:  x = x.permute(0,3,1,2).contiguous(); x = act(conv2d(x))
 *       with conv2d = nn.Conv2d(d, d, 3, padding=1, groups=d, bias) (vmamba.py:683-691), fused with
 *       the layout half of CrossScan (vmamba.py:80-89): the activation is written once in row-major
 *       and once in column-major sequence order, which is all the scan kernels need (the two
 *       flipped directions are read backwards, see sigma_scan.h rev_group_mask).
 *
 * Same conventions as sigma_scan.h: device pointers, float32, contiguous tensors, the callee
 * enqueues on `stream` (hipStream_t as void*) of the current device, never allocates, never
 * synchronises; returns 0 or a SIGMA_OPS_ERR_* code.
 */
#ifndef SIGMA_OPS_H_
#define SIGMA_OPS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum sigma_ops_status {
    SIGMA_OPS_OK = 0,
    SIGMA_OPS_ERR_ARG = 1,
    SIGMA_OPS_ERR_LAUNCH = 5
};

typedef struct sigma_dwconv_params {
    int32_t batch, channels, height, width;
    int32_t n_orders;      /* 2: out2/g2 hold the row-major AND the column-major sequence (SS2D);
                              1: row-major only, i.e. plain conv + SiLU (CroMB / ConMB, vmamba.py:1629-1630,
                              1271-1272) */
    int32_t reserved_;
    const float *x;        /* (B, d, H, W)   input of the convolution                              */
    const float *weight;   /* (d, 1, 3, 3)                                                         */
    const float *bias;     /* (d) or NULL                                                          */
    /* forward */
    float *out2;           /* (B, n_orders, d, H*W): [:,0] = silu(conv(x)) row-major, [:,1] = the same
                              image in column-major order (index w*H + h)                          */
    /* backward */
    const float *g2;       /* (B, n_orders, d, H*W) gradient of out2                                      */
    float *gpre;           /* (B, d, H, W) scratch: gradient w.r.t. the pre-activation            */
    float *dweight;        /* (d, 1, 3, 3) ACCUMULATED into (caller zeroes)                        */
    float *dbias;          /* (d) ACCUMULATED into, or NULL                                        */
    float *dx;             /* (B, d, H, W) fully written                                           */
} sigma_dwconv_params;

int sigma_dwconv3x3_silu_fwd(const sigma_dwconv_params *params, void *stream);
int sigma_dwconv3x3_silu_bwd(const sigma_dwconv_params *params, void *stream);

/*   sigma_cross_merge_nhwc / sigma_cross_split_nhwc
 *       CrossMerge (vmamba.py:100-108) + the transpose to channels-last in front of out_norm
 *       (vmamba.py:221-224), and its adjoint (CrossMerge.backward = CrossScan, vmamba.py:110-121).
 *       The scan kernels already deliver the flipped directions in natural order, so with the group
 *       order g = 2 * memory_order + flipped:
 *           merge:  nhwc[b,h,w,c] = planes4[b,0,c,hW+w] + planes4[b,1,c,hW+w]
 *                                 + planes4[b,2,c,wH+h] + planes4[b,3,c,wH+h]
 *           split:  planes2[b,0,c,hW+w] = planes2[b,1,c,wH+h] = nhwc[b,h,w,c]                      */
typedef struct sigma_merge_params {
    int32_t batch, channels, height, width;
    const float *planes4;  /* merge in : (B, 4, d, H*W)                       */
    float *planes2;        /* split out: (B, 2, d, H*W)                       */
    float *nhwc;           /* merge out / split in: (B, H, W, d) contiguous   */
} sigma_merge_params;

int sigma_cross_merge_nhwc(const sigma_merge_params *params, void *stream);
int sigma_cross_split_nhwc(const sigma_merge_params *params, void *stream);

/*   sigma_transpose2d
 *       dst[b][c][r] = src[b][r][c] with free row / batch strides (floats): the channels-last ->
 *       channels-first copy in front of the depthwise conv (vmamba.py:1070-1071) reading the x half
 *       of the in_proj output in place, and the inverse copy that puts dx into the x half of the
 *       in_proj gradient (what autograd's chunk() backward does with a strided cat).               */
typedef struct sigma_transpose_params {
    int32_t batch, rows, cols, reserved_;
    const float *src;      /* element (b, r, c) at src[b*src_batch_stride + r*src_row_stride + c] */
    float *dst;            /* element (b, c, r) at dst[b*dst_batch_stride + c*dst_row_stride + r] */
    int64_t src_batch_stride, src_row_stride, dst_batch_stride, dst_row_stride;
} sigma_transpose_params;

int sigma_transpose2d(const sigma_transpose_params *params, void *stream);

/*   sigma_pair_sum_add
 *       acc[o][i] += src[2*o][i] + src[2*o + 1][i]  for o < n_outer, i < inner (contiguous fp32).
 *       Adjoint of reading ONE copy of x for the two directions of a memory order (u_group_shift = 1 in
 *       sigma_scan.h): du of the forward and of the flipped direction are added to the x_proj part of the
 *       gradient in one pass (autograd of the reference's CrossScan does it with flips and adds,
 *       vmamba.py:91-98); replaces two strided torch adds.                                        */
int sigma_pair_sum_add(const float *src, float *acc, int64_t n_outer, int64_t inner, void *stream);

/*   sigma_upsample2x_nhwc
 *       F.interpolate(scale_factor=2, mode="bilinear", align_corners=False) of a contiguous channels-last fp32 tensor
 *       (B, H, W, C) -> (B, 2H, 2W, C) (backward = 0), and its adjoint (backward = 1: `in` is the gradient
 *       (B, 2H, 2W, C), `out` the input gradient (B, H, W, C); height / width are ALWAYS those of the small tensor):
 *       the decoder's UpsampleExpand / FinalUpsample_X4 (models/decoders/MambaDecoder.py:33-51, 76-97).  C % 4 == 0,
 *       16-byte aligned pointers.  Gather formulation in both directions (deterministic, no atomics).           */
int sigma_upsample2x_nhwc(const float *in, float *out, int32_t batch, int32_t height, int32_t width, int32_t channels,
                          int32_t backward, void *stream);

/*   sigma_split_bf16
 *       Operand images of the split-operand bf16 GEMM (sigma_amd/split_linear.py; the nn.Linear calls of
 *       vmamba.py, e.g. SS2D.in_proj / out_proj :1067-1089): with hi = bf16(x) and lo = bf16(x - hi), row r of the
 *       fp32 source (rows x cols, row stride src_row_stride elements) is written as bf16 to
 *           dst + r * dst_row_stride              <- hi
 *           dst + r * dst_row_stride + hi2_offset <- hi again (skipped when hi2_offset < 0)
 *           dst + r * dst_row_stride + lo_offset  <- lo
 *       (offsets in bf16 elements), so that [hi | hi | lo], [hi | lo | hi] (concatenation along the columns) and
 *       [hi ; lo ; hi] (along the rows) come out of one pass.                                              */
int sigma_split_bf16(const float *src, int64_t rows, int64_t cols, int64_t src_row_stride, void *dst, int64_t dst_row_stride,
                     int64_t hi2_offset, int64_t lo_offset, void *stream);

/*   sigma_layernorm_fwd / sigma_layernorm_bwd
 *       nn.LayerNorm(C, eps=1e-5, affine) over the last dimension of a contiguous (rows, C) fp32
 *       tensor: every LayerNorm of the hot path (vmamba.py:617, 724, 1183-1184, 1448-1449, 1693,
 *       1783, 1797; MambaDecoder.py:18, 41, 85).  C % 4 == 0, C <= 2048.
 *       bwd: dx fully written; dgamma / dbeta fully written (deterministic two-stage column sums
 *       through `workspace` of sigma_layernorm_bwd_partial_rows(rows, C) * 2 * C floats).
 *       bwd with a gate needs beta as well (the normalised value is recomputed).                  */
typedef struct sigma_layernorm_params {
    int64_t rows;
    int32_t channels;
    float eps;
    const float *x;        /* (rows, C)                       */
    const float *gamma;    /* (C)                             */
    const float *beta;     /* (C) or NULL                     */
    float *y;              /* fwd out (rows, C)               */
    float *mean;           /* fwd out / bwd in (rows), or NULL in a forward that needs no backward */
    float *rstd;           /* fwd out / bwd in (rows)         */
    const float *dy;       /* bwd in  (rows, C)               */
    float *dx;             /* bwd out (rows, C)               */
    float *dgamma;         /* bwd out (C)                     */
    float *dbeta;          /* bwd out (C) or NULL             */
    float *workspace;      /* bwd scratch                     */
    /* optional fused gate of SS2D.forward (vmamba.py:1077: y = out_norm(y) * act(z)):
     * y = LayerNorm(x) * silu(gate);  gate rows are gate_row_stride floats apart (z is the second
     * half of the in_proj output), dgate is contiguous (rows, C).  NULL gate = plain LayerNorm.  */
    const float *gate;
    int64_t gate_row_stride;
    float *dgate;
} sigma_layernorm_params;

int sigma_layernorm_fwd(const sigma_layernorm_params *params, void *stream);
int sigma_layernorm_bwd(const sigma_layernorm_params *params, void *stream);
int sigma_layernorm_bwd_partial_rows(int64_t rows, int32_t channels);

#ifdef __cplusplus
}
#endif
#endif /* SIGMA_OPS_H_ */
