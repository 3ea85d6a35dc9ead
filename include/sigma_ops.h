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
    float *gpre;           /* (B, d, H, W) scratch: gradient w.r.t. the pre-activation (planes that fit
                              LDS keep it there: then not written)                                  */
    float *dweight;        /* (d, 1, 3, 3) ACCUMULATED into (caller zeroes)                        */
    float *dbias;          /* (d) ACCUMULATED into, or NULL                                        */
    float *dx;             /* (B, d, H, W) fully written                                           */
    /* ABI 10: planes of x and dx need not be packed as (B, d): plane (b, c) starts at b * x_batch_stride + c * x_channel_stride
     * floats (each plane H*W contiguous); 0 / 0 = the contiguous (B, d, H, W) tensor.  SS2D hands the x half of in_proj over
     * in CHANNEL-major order (d, B, H, W) -- written transposed by the GEMM epilogue (sigma_gemm.h, t_cols), read here in place. */
    int64_t x_batch_stride, x_channel_stride;
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

/*   sigma_plane_pool / sigma_plane_scale / sigma_plane_dot / sigma_plane_gate_bwd
 *       ChannelAttention of the decoder's conv branch (vmamba.py:1725-1741; called from ChannelAttentionBlock :1744-1757
 *       inside CVSSDecoderBlock :1800-1805):  y = x * sigmoid(fc(avg_pool(x)) + fc(max_pool(x)))  on contiguous
 *       (B, C, H, W) fp32 activations, seen here as `planes` = B * C planes of `hw` = H * W floats.
 *           pool     : mean[p], max[p], count[p] = number of elements of plane p equal to its max (one pass over x)
 *           scale    : out[p][i] = x[p][i] * scale[p]
 *           dot      : out[p] = sum_i a[p][i] * b[p][i]          (d/d scale of the product: a = dy, b = x)
 *           gate_bwd : dx[p][i] = g[p][i] * scale[p] + dmean[p] / hw + (x[p][i] == max[p] ? dmax[p] / count[p] : 0)
 *       -- the gradient of the max pool is shared by tied maxima, as torch.amax does (the reference's
 *       AdaptiveMaxPool2d routes it to one of them; the two agree wherever the maximum is unique).  The tiny (2B, C)
 *       squeeze/excite MLP between pool and scale stays with the caller.                                          */
int sigma_plane_pool(const float *x, int64_t planes, int64_t hw, float *mean, float *max, float *count, void *stream);
int sigma_plane_scale(const float *x, const float *scale, float *out, int64_t planes, int64_t hw, void *stream);
int sigma_plane_dot(const float *a, const float *b, float *out, int64_t planes, int64_t hw, void *stream);
typedef struct sigma_gate_bwd_params {
    int64_t planes, hw;
    const float *g;        /* (planes, hw) gradient of the gated output                     */
    const float *x;        /* (planes, hw) input of the gate                                */
    const float *scale;    /* (planes) sigmoid gate of the forward                          */
    const float *dmean;    /* (planes) gradient of the pooled means                         */
    const float *dmax;     /* (planes) gradient of the pooled maxima                        */
    const float *max;      /* (planes) pooled maxima of the forward                         */
    const float *count;    /* (planes) ties of the forward                                  */
    float *dx;             /* (planes, hw) fully written                                    */
} sigma_gate_bwd_params;
int sigma_plane_gate_bwd(const sigma_gate_bwd_params *params, void *stream);

/*   sigma_softmax_ce_fwd / sigma_softmax_ce_bwd
 *       nn.CrossEntropyLoss(reduction='mean', ignore_index) of models/builder.py:146-166 (criterion built in
 *       train.py:95) on channels-last logits: `rows` pixels of `classes` contiguous fp32 logits (classes % 4 == 0,
 *       16-byte aligned), int64 labels.
 *           fwd : lse[r] = log sum_c exp(logit[r][c]);  partial[2k], partial[2k+1] = (sum of lse[r] - logit[r][label[r]],
 *                 number of pixels) over the pixels workgroup k < SIGMA_CE_BLOCKS handled with label != ignore_index
 *                 (labels outside [0, classes) count as ignored; the reference asserts on them).  The caller adds the
 *                 SIGMA_CE_BLOCKS pairs: loss = sum / count, deterministic.
 *           bwd : dlogits[r][c] = (exp(logit[r][c] - lse[r]) - [c == label[r]]) * scale[0], zero for ignored pixels;
 *                 scale = upstream gradient / count, a DEVICE scalar (no host synchronisation).                   */
#define SIGMA_CE_BLOCKS 1024
int sigma_softmax_ce_fwd(const float *logits, const int64_t *labels, int64_t rows, int32_t classes, int64_t ignore_index,
                         float *lse, float *partial, void *stream);
int sigma_softmax_ce_bwd(const float *logits, const int64_t *labels, const float *lse, const float *scale, int64_t rows,
                         int32_t classes, int64_t ignore_index, float *dlogits, void *stream);

/*   sigma_colscale_bwd
 *       backward of  y = a + x * scale  with a per-channel `scale` on contiguous channels-last rows (rows, C): the residual
 *       of the decoder block, x * scale1 + op(norm1(x)) and x * scale2 + conv_blk(norm2(x)) (vmamba.py:1800-1805):
 *           dx[r][c] = dy[r][c] * scale[c]          dscale[c] += sum_r dy[r][c] * x[r][c]
 *       in one pass over dy and x (the autograd formulation: two multiplies and a column reduction).  `dscale` is
 *       ZERO-FILLED by the caller (float atomics, one per block and channel).  C % 4 == 0, C <= 1024, 16-byte aligned
 *       operands.                                                                                                   */
int sigma_colscale_bwd(const float *dy, const float *x, const float *scale, float *dx, float *dscale, int64_t rows,
                       int32_t channels, void *stream);

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
     * half of the in_proj output); dgate rows are dgate_row_stride floats apart (0 = C, contiguous):
     * the caller can have dz written straight into the z half of the in_proj output's gradient.
     * NULL gate = plain LayerNorm.  */
    const float *gate;
    int64_t gate_row_stride;
    float *dgate;
    int64_t dgate_row_stride;
    /* optional per-sample factor on the OUTPUT (stochastic depth, timm DropPath as used by VSSBlock vmamba.py:1716-1722
     * and CVSSDecoderBlock :1800-1805: the branch is multiplied by mask[b] / keep_prob before the residual add):
     * y = (LayerNorm(x) [* silu(gate)]) * row_scale[r / rows_per_scale]; the backward multiplies dy by the same factor
     * on load.  The out_proj that follows is linear and bias-free, so scaling its input scales the branch.  NULL = 1. */
    const float *row_scale;
    int64_t rows_per_scale;
    /* ABI 8: optional addend of dx, (rows, C) contiguous, 16-byte aligned (may be dx itself): the gradient that reached x
     * on the path AROUND the LayerNorm -- the residual stream of a block, x + op(norm(x)) (vmamba.py:1716-1722) --
     * joined here instead of in an add pass of its own.  Backward only; NULL = none. */
    const float *dx_add;
} sigma_layernorm_params;

int sigma_layernorm_fwd(const sigma_layernorm_params *params, void *stream);
int sigma_layernorm_bwd(const sigma_layernorm_params *params, void *stream);
/* rows of 2 * C floats the backward needs in `workspace`: one partial (dgamma, dbeta) row per WORKGROUP (its four waves
 * meet in LDS), added up in a fixed order by a second small kernel */
int sigma_layernorm_bwd_partial_rows(int64_t rows, int32_t channels);

#ifdef __cplusplus
}
#endif
#endif /* SIGMA_OPS_H_ */
