/*
 * sigma_scan.h -- C ABI of libsigma_hip.so: the MI355X (gfx950) selective-scan
 * operator that replaces the reference's CUDA extension `selective_scan_cuda_core`.
 *
 * Drop-in boundary (reference paths relative to /root/reference):
 *   sigma_selective_scan_fwd  <->  selective_scan_fwd()
 *       models/encoders/selective_scan/csrc/selective_scan/selective_scan.cpp:165-249
 *       (pybind `fwd`, :365)
 *   sigma_selective_scan_bwd  <->  selective_scan_bwd()
 *       models/encoders/selective_scan/csrc/selective_scan/selective_scan.cpp:251-362
 *       (pybind `bwd`, :366)
 *   the parameter blocks mirror SSMParamsBase / SSMParamsBwd
 *       models/encoders/selective_scan/csrc/selective_scan/selective_scan.h:26-90
 *       (sizes, raw device pointers, element strides).
 *
 * Plain pointers and sizes only -- no torch types.  The callee never allocates
 * and never synchronises: it enqueues kernels on `stream` (a hipStream_t passed
 * as void*; NULL = the null stream) of the CURRENT device and returns.
 * All pointers are device pointers.  Strides are in ELEMENTS (as in the reference,
 * selective_scan.cpp:87).  The innermost (sequence) stride of u, delta, B, C,
 * out, dout, du, ddelta, dB, dC must be 1 (selective_scan.cpp:189-190,206-208).
 *
 * Return value: 0 on success, a SIGMA_ERR_* code otherwise; the host wrapper
 * turns non-zero into RuntimeError like TORCH_CHECK does in the reference.
 * sigma_scan_last_error() returns a thread-local human-readable message.
 */
#ifndef SIGMA_SCAN_H_
#define SIGMA_SCAN_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIGMA_SCAN_ABI_VERSION 10

/* dtype of u, delta, B, C, out, dout, du, ddelta  (input_t of the reference,
 * selective_scan.cpp:174: float / half / bfloat16).  A, D, delta_bias, x, dA,
 * dD, ddelta_bias and the dB/dC accumulators are always float32 (weight_t). */
enum sigma_dtype {
    SIGMA_DTYPE_F32 = 0,
    SIGMA_DTYPE_F16 = 1,
    SIGMA_DTYPE_BF16 = 2
};

enum sigma_status {
    SIGMA_OK = 0,
    SIGMA_ERR_NULL_ARG = 1,      /* required pointer is NULL */
    SIGMA_ERR_BAD_SHAPE = 2,     /* sizes violate the reference's TORCH_CHECKs */
    SIGMA_ERR_BAD_DTYPE = 3,
    SIGMA_ERR_BAD_STRIDE = 4,
    SIGMA_ERR_LAUNCH = 5,        /* hip launch error (cf. C10_CUDA_KERNEL_LAUNCH_CHECK) */
    SIGMA_ERR_NO_DEVICE = 6,
    SIGMA_ERR_BAD_OPTION = 7
};

/* The SHAPE of the checkpoint tensor x follows the reference: (batch, dim, n_chunks, 2*dstate)
 * float32 with n_chunks = ceil(seqlen / 2048) (selective_scan.cpp:225-228).  Its CONTENTS are
 * private scratch between this library's fwd and bwd (the reference never documents or tests
 * them; its own bwd only accepts the x of its own fwd).  Layout used here -- the kernels tile the
 * sequence by 1280 / 640 / 320 / 256 elements, which 2048 is not a multiple of:
 *     checkpoint j = the N states of a row after element min(seqlen, (j+1)*pitch) - 1,
 *     for j < ceil(seqlen / pitch), stored at x[(b*dim + r) * x_row_stride + j*N + n].
 * Default pitch 1280 and x_row_stride = n_chunks*2*N (the reference shape; 2*ceil(seqlen/2048) >=
 * ceil(seqlen/1280), so it always fits).  Callers that allocate x themselves may ask for a fine
 * pitch 640 or 320 (ckpt_pitch field) with x_row_stride >= ceil(seqlen/pitch)*N: the backward then
 * needs no forward sweep at all (its tiles are 640 / 320 elements) and runs the second-generation
 * kernel, whose workgroups accumulate dB/dC over many rows before touching memory.  Pitch 160 selects
 * the quad-row kernels (scan_fwd4.hip / scan_bwd4.hip: a wave = 4 channel rows x 16 lanes x 10 positions; with few rows
 * the backward additionally splits the sequence into segments): f32 IO, dstate in {4, 8, 16}, rows per group divisible
 * by 4, seqlen % 4 == 0, 16-byte aligned operands -- the backward fails with SIGMA_ERR_BAD_SHAPE otherwise (the forward
 * falls back to the 64-lane kernel, which writes the same checkpoints).  Unused slots are not written.
 * Pitch 16 selects the row-lane kernels (scan_fwdr.hip / scan_bwdr.hip: a lane = a channel row, a wave = 64 rows x a
 * quarter / an eighth of the states, B and C as scalar operands): f32 IO, dstate in {4, 8, 16}, rows per group divisible
 * by 64, seqlen % 4 == 0, 16-byte aligned operands; BOTH entry points fail with SIGMA_ERR_BAD_SHAPE otherwise.  x then
 * holds ceil(seqlen / 16) * dstate floats per row (one checkpoint per 16-position tile: the backward replays a tile whole
 * from the state entering it; rounds 4-5 kept two per tile and walked halves -- the forward is bound by its HBM traffic, of
 * which those checkpoints were 40 %) in a layout private to the two kernels (x_row_stride is ignored):
 *     x[((b * dim/64 + r/64) * ceil(L/16) + tile) * N * 64 + ((n / G) * 64 + r % 64) * G + n % G], G = N / 4, = state n
 *     of row r after memory tile `tile` in scan order (a reversed group walks the tiles downwards).
 * With few rows both kernels cut the sequence into segments run by different workgroups (a pre-pass writes per-segment
 * summaries into the workspace): see sigma_scan_fwd_workspace_bytes / sigma_scan_bwd_workspace_bytes. */
#define SIGMA_SCAN_CHUNK 2048
#define SIGMA_SCAN_CKPT_PITCH 1280
#define SIGMA_SCAN_CKPT_PITCH_FINE 640
#define SIGMA_SCAN_CKPT_PITCH_160 160
#define SIGMA_SCAN_CKPT_PITCH_320 320
#define SIGMA_SCAN_CKPT_PITCH_16 16
/* dstate limit of the reference (selective_scan.cpp:10,201). */
#define SIGMA_SCAN_MAX_DSTATE 256

typedef struct sigma_scan_fwd_params {
    /* sizes */
    int32_t batch;      /* B            */
    int32_t dim;        /* K*d_inner    */
    int32_t seqlen;     /* L            */
    int32_t dstate;     /* N            */
    int32_t n_groups;   /* G; row r uses group r / (dim / G) */
    int32_t n_chunks;   /* ceil(L / 2048); checked */
    int32_t io_dtype;   /* enum sigma_dtype */
    int32_t delta_softplus;
    /* Extensions for the fused SS2D caller (both 0 = the reference operator):
     *  rev_group_mask: bit g set -> group g scans the sequence backwards: every sequence operand
     *      of its rows (u, delta, B, C, out, dout, du, ddelta, dB, dC) is read / written at index
     *      seqlen-1-l.  This is CrossScan's flip (vmamba.py:80-98, directions 2 and 3) done by
     *      addressing instead of by materialised copies.  Needs n_groups <= 32 when non-zero.
     *  u_group_shift: the rows of group g read the u rows of group (g >> u_group_shift), i.e. u
     *      has shape (B, dim >> u_group_shift, L): CrossScan directions that differ only by the
     *      flip share one physical copy of x. */
    uint32_t rev_group_mask;
    int32_t u_group_shift;
    int32_t ckpt_pitch;        /* 0 = SIGMA_SCAN_CKPT_PITCH (1280); 640 / 320 / 160 / 16 = fine checkpoints (see above) */
    int32_t param_group_swap;  /* 1 (needs n_groups == 4): A, D, delta_bias and dA, dD, ddelta_bias keep the REFERENCE's direction
                                  order k = [row, col, row reversed, col reversed] (vmamba.py:84-89) while the sequence
                                  operands use the kernel's group order g = 2*order + reversed: group g reads / writes the
                                  parameter rows of group ((g & 1) << 1) | (g >> 1).  Spares the caller six small
                                  permutation copies per call. */
    int64_t x_row_stride;      /* floats per (batch, row) of x; 0 = n_chunks * 2 * dstate */
    /* inputs */
    const void *u;            /* (B, dim, L)        io_dtype */
    const void *delta;        /* (B, dim, L)        io_dtype */
    const float *A;           /* (dim, N)           f32      */
    const void *B;            /* (B, G, N, L)       io_dtype */
    const void *C;            /* (B, G, N, L)       io_dtype */
    const float *D;           /* (dim) or NULL      f32      */
    const float *delta_bias;  /* (dim) or NULL      f32      */
    /* outputs */
    void *out;                /* (B, dim, L)        io_dtype */
    float *x;                 /* (B, dim, n_chunks, 2N) f32 (or x_row_stride floats per row), or NULL
                                 (inference): state checkpoints, layout above */
    /* element strides */
    int64_t u_batch_stride, u_d_stride;
    int64_t delta_batch_stride, delta_d_stride;
    int64_t A_d_stride, A_dstate_stride;
    int64_t B_batch_stride, B_group_stride, B_dstate_stride;
    int64_t C_batch_stride, C_group_stride, C_dstate_stride;
    int64_t out_batch_stride, out_d_stride;
    /* device scratch of >= sigma_scan_fwd_workspace_bytes() bytes, 16-byte aligned (ckpt_pitch 16 with few rows: the
     * forward summaries of the sequence segments); may be NULL when that function returns 0.  Unused by the backward. */
    void *workspace;
    int64_t workspace_bytes;
} sigma_scan_fwd_params;

typedef struct sigma_scan_bwd_params {
    sigma_scan_fwd_params fwd;   /* out is unused; x is the tensor saved by fwd (the reference
                                    requires it when n_chunks > 1, selective_scan.cpp:320; here it
                                    may be NULL only when seqlen <= SIGMA_SCAN_CKPT_PITCH) */
    const void *dout;            /* (B, dim >> dout_group_shift, L)   io_dtype */
    void *du;                    /* (B, dim, L)   io_dtype, fully written */
    void *ddelta;                /* (B, dim, L)   io_dtype, fully written */
    float *dA;                   /* (dim, N)      f32, ACCUMULATED into (caller zeroes, :331) */
    float *dB;                   /* (B, G, N, L)  f32, fully WRITTEN (the reference zero-fills and
                                    atomically accumulates, :332; here a deterministic 2-stage sum) */
    float *dC;                   /* (B, G, N, L)  f32, fully WRITTEN */
    float *dD;                   /* (dim) or NULL f32, ACCUMULATED into */
    float *ddelta_bias;          /* (dim) or NULL f32, ACCUMULATED into */
    void *workspace;             /* device scratch of >= sigma_scan_bwd_workspace_bytes() bytes,
                                    16-byte aligned; may be NULL when that function returns 0.
                                    Holds the per-workgroup dB/dC partials; contents are garbage
                                    after the call.  Provided by the caller because the callee
                                    never allocates. */
    int64_t workspace_bytes;
    int32_t dout_group_shift;    /* like u_group_shift for dout: CrossMerge's adjoint hands the same
                                    gradient to the directions that share a memory order */
    int32_t reserved_;
    int64_t dout_batch_stride, dout_d_stride;
    int64_t du_batch_stride, du_d_stride;
    int64_t ddelta_batch_stride, ddelta_d_stride;
    int64_t dA_d_stride, dA_dstate_stride;
    int64_t dB_batch_stride, dB_group_stride, dB_dstate_stride;
    int64_t dC_batch_stride, dC_group_stride, dC_dstate_stride;
} sigma_scan_bwd_params;

/* Forward: out, x <- scan(u, delta, A, B, C, D, delta_bias). */
int sigma_selective_scan_fwd(const sigma_scan_fwd_params *params, void *stream);

/* Backward: du, ddelta, dB, dC written; dA, dD, ddelta_bias accumulated (+=). */
int sigma_selective_scan_bwd(const sigma_scan_bwd_params *params, void *stream);

/* Scratch bytes sigma_selective_scan_fwd needs for this problem under the current options (non-zero only for
 * ckpt_pitch 16 when the sequence is cut into segments); negative = invalid params. */
int64_t sigma_scan_fwd_workspace_bytes(const sigma_scan_fwd_params *params);

/* Scratch bytes sigma_selective_scan_bwd needs for this problem under the current options: the per-workgroup dB/dC
 * partials (0 when one workgroup covers a whole (batch, group)) plus the segment summaries of a split sequence;
 * negative = invalid params. */
int64_t sigma_scan_bwd_workspace_bytes(const sigma_scan_bwd_params *params);

/* Thread-local description of the last non-zero status returned on this thread. */
const char *sigma_scan_last_error(void);

/* ABI version of the loaded library (== SIGMA_SCAN_ABI_VERSION it was built with). */
int sigma_scan_abi_version(void);

/* Tuning knobs for benchmarking; 0 restores the built-in heuristic.
 *   "fwd_items" / "bwd_items"  elements per lane: fwd {4,5,10,20}, bwd {4,5,10} (tile = 64 x items)
 *   "fwd_waves" / "bwd_waves"  channel rows per workgroup (1..16; must divide dim / n_groups)
 *   "fwd_tiles"                consecutive sequence tiles per workgroup, forward (rows x tiles <= 16)
 *   "fwd_nb" / "bwd_nb"        states per B/C staging block {1,2,4,8}
 *   "no_glds"                  1 = stage B/C through registers instead of global_load_lds
 *   "bwd_slab2"                1 = two dB/dC slab sets in LDS (one barrier per state) when they fit
 *   "fwd_prefetch"             2 = no register prefetch of the next tile's u/delta (T = 10)
 *   "bwd_gen"                  1 = first-generation backward (scan_bwd.hip) always, 2 = second generation
 *                              (scan_bwd2.hip: needs ckpt_pitch 640 / 320 and dstate <= 64) whenever legal
 *   "bwd_rb"                   second-generation backward: row blocks a workgroup accumulates over (0..256)
 *   "bwd_touch"                L2 warm-up of the next row step's u/delta/dout lines: 1 = on, 2 = off, 0 = on in the
 *                              quad-row backward (touches 3 states ahead: -3..-6 %), off in the second generation (there
 *                              a whole row step ahead: every line fetched twice for 1 %, profiles/r02_pmc_enc_s2_b16.txt)
 *   "fwd_gen"                  1 = never the quad-row forward (scan_fwd4.hip), which serves ckpt_pitch 160 with >= 8192
 *                              rows otherwise; 2 = also with fewer rows
 *   "bwd_sb"                   quad-row backward (ckpt_pitch 160): states per barrier {1, 2, 4, 8}; 0 = 2
 *   "bwd_seg"                  quad-row backward: sequence segments {2,3,4,6,8,12}; 1 = never split; 0 = cost model
 *   "bwd_wgs"                  quad-row backward: 2 = two workgroups of <= 8 waves per CU (A/B knob)
 *   "rl_waves"                 row-lane kernels (ckpt_pitch 16): state waves per 64-row block {4, 8, 16}; 0 = cost model
 *   "rl_segs"                  row-lane kernels: sequence segments (1 = never split, 2..64); 0 = cost model
 *   "rl_chain"                 row-lane backward: chained walk (row blocks laid end to end over exactly as many workgroups as
 *                              the chip holds, a cut row block hands its reverse carry to the neighbour): 2 = whenever
 *                              there are more row blocks than resident workgroups; 0 / 1 = never (measured: no gain)
 *   "rl_chain_timeouts"        READ-ONLY (get_option; set_option refuses it): hand-over waits of the chained walk that ran
 *                              out since the last read -- each one poisoned its row block's du / ddelta / dA with NaN (the
 *                              producer workgroup was not resident: another stream's kernel held its slot).  Reading
 *                              synchronises the device and resets the count; -1 on a device error.
 * Returns SIGMA_ERR_BAD_OPTION for unknown names / unsupported values. */
int sigma_scan_set_option(const char *name, int value);
int sigma_scan_get_option(const char *name);

/* Launch geometry the heuristic picks for a problem (for reports/tests): writes
 * {items_per_lane, rows_per_workgroup, workgroups, lds_bytes, tiles_per_workgroup, states_per_block} */
int sigma_scan_fwd_plan(const sigma_scan_fwd_params *params, int32_t plan[6]);
int sigma_scan_bwd_plan(const sigma_scan_bwd_params *params, int32_t plan[6]);

/* Development aid: per-phase cycle totals of the second-generation backward since the last call
 * (all zero unless the library was built with -DSIGMA_BWD2_PROF=1); synchronises the device. */
int sigma_scan_debug_read(uint64_t out16[16]);

/* On-device self test of the wave64 DPP scan primitives against a serial loop.
 * Returns 0 when every lane matches; enqueues on `stream` and synchronises it. */
int sigma_scan_selftest(void *stream);

/* ABI 9.  Self test of the row-lane kernels (ckpt_pitch 16; csrc/scan_fwdr.hip / scan_bwdr.hip): one small problem --
 * two groups of 64 rows, the second walked backwards, 16 states, 148 positions (a partial last tile), softplus, D and
 * delta_bias -- through sigma_selective_scan_fwd / _bwd, forward and all seven gradients compared with a host recurrence in
 * double precision (scaled max error < 2e-4).  These kernels retire their vector-memory requests with hand-counted
 * s_waitcnt values and pin their scalar-operand waits with scheduling barriers: a toolchain that schedules them
 * differently fails here, loudly.  Allocates, copies and synchronises `stream` (not for use inside a stream capture); the
 * host binding runs it once per process and device before the first ckpt_pitch-16 launch.  0 = pass. */
int sigma_scan_rowlane_selftest(void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SIGMA_SCAN_H_ */
