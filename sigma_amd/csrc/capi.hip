// capi.hip -- the C ABI of libsigma_hip.so (declared in include/sigma_scan.h).
//
// Host side of the drop-in for the reference's selective_scan.cpp: argument checks
// mirror its TORCH_CHECKs (selective_scan.cpp:173-223, 261-327), the launch ladder
// replaces its (threads, items) ladder (selective_scan_fwd_kernel.cuh:226-238) with a
// two-parameter choice: items per lane (tile length) x rows per workgroup.
#include "../../include/sigma_scan.h"
#include "scan_device.h"
#include "scan_launch.h"

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

namespace {

thread_local std::string g_err;

std::atomic<int> g_opt_fwd_items{0}, g_opt_fwd_waves{0}, g_opt_bwd_items{0}, g_opt_bwd_waves{0};

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int elem_size(int dtype) { return dtype == SIGMA_DTYPE_F32 ? 4 : 2; }

bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

int check_fwd(const sigma_scan_fwd_params* p, bool need_out, bool need_ptrs = true) {
    if (!p) return fail(SIGMA_ERR_NULL_ARG, "params is NULL");
    if (p->io_dtype < 0 || p->io_dtype > 2)
        return fail(SIGMA_ERR_BAD_DTYPE, "io_dtype %d not in {f32,f16,bf16}", p->io_dtype);
    if (p->batch < 0 || p->dim <= 0 || p->seqlen < 0 || p->dstate <= 0 || p->n_groups <= 0)
        return fail(SIGMA_ERR_BAD_SHAPE, "bad sizes batch=%d dim=%d seqlen=%d dstate=%d groups=%d", p->batch, p->dim,
                    p->seqlen, p->dstate, p->n_groups);
    if (p->dim % p->n_groups != 0)
        return fail(SIGMA_ERR_BAD_SHAPE, "dims should be dividable by n_groups (dim=%d, n_groups=%d)", p->dim,
                    p->n_groups);
    if (p->dstate > SIGMA_SCAN_MAX_DSTATE)
        return fail(SIGMA_ERR_BAD_SHAPE, "selective_scan only supports state dimension <= 256 (got %d)", p->dstate);
    if (p->n_chunks != (p->seqlen + SIGMA_SCAN_CHUNK - 1) / SIGMA_SCAN_CHUNK)
        return fail(SIGMA_ERR_BAD_SHAPE, "n_chunks must be ceil(seqlen/2048) (got %d for seqlen %d)", p->n_chunks,
                    p->seqlen);
    if (p->batch == 0 || p->seqlen == 0 || !need_ptrs) return SIGMA_OK;
    if (!p->u || !p->delta || !p->A || !p->B || !p->C || (need_out && !p->out))
        return fail(SIGMA_ERR_NULL_ARG, "u/delta/A/B/C/out must be non-NULL device pointers");
    return SIGMA_OK;
}

// vector (4-element) access is legal when every row start is 4-element aligned
bool vec_ok_fwd(const sigma_scan_fwd_params* p, bool with_out) {
    const size_t a = 4 * (size_t)elem_size(p->io_dtype);
    bool ok = aligned_to(p->u, a) && aligned_to(p->delta, a) && aligned_to(p->B, a) && aligned_to(p->C, a);
    if (with_out) ok = ok && aligned_to(p->out, a) && p->out_batch_stride % 4 == 0 && p->out_d_stride % 4 == 0;
    ok = ok && p->u_batch_stride % 4 == 0 && p->u_d_stride % 4 == 0;
    ok = ok && p->delta_batch_stride % 4 == 0 && p->delta_d_stride % 4 == 0;
    ok = ok && p->B_batch_stride % 4 == 0 && p->B_group_stride % 4 == 0 && p->B_dstate_stride % 4 == 0;
    ok = ok && p->C_batch_stride % 4 == 0 && p->C_group_stride % 4 == 0 && p->C_dstate_stride % 4 == 0;
    return ok;
}

sigma::FwdArgs make_fwd_args(const sigma_scan_fwd_params* p, int nwaves, bool vec) {
    sigma::FwdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.u = p->u; a.delta = p->delta; a.A = p->A; a.B = p->B; a.C = p->C; a.D = p->D; a.bias = p->delta_bias;
    a.out = p->out; a.x = p->x;
    a.batch = p->batch; a.dim = p->dim; a.L = p->seqlen; a.N = p->dstate; a.G = p->n_groups;
    a.n_chunks = p->n_chunks; a.rows_per_group = p->dim / p->n_groups; a.softplus = p->delta_softplus ? 1 : 0;
    a.vec_ok = vec ? 1 : 0; a.rowblocks = p->dim / nwaves;
    a.u_bs = p->u_batch_stride; a.u_ds = p->u_d_stride; a.dt_bs = p->delta_batch_stride; a.dt_ds = p->delta_d_stride;
    a.A_ds = p->A_d_stride; a.A_ns = p->A_dstate_stride;
    a.B_bs = p->B_batch_stride; a.B_gs = p->B_group_stride; a.B_ns = p->B_dstate_stride;
    a.C_bs = p->C_batch_stride; a.C_gs = p->C_group_stride; a.C_ns = p->C_dstate_stride;
    a.o_bs = p->out_batch_stride; a.o_ds = p->out_d_stride;
    return a;
}

// rows (waves) per workgroup: the largest power of two <= cap that divides the rows of a
// group (all rows of a workgroup must share B/C), shrunk while the grid would leave CUs idle.
int pick_waves(int rows_per_group, long total_rows, int cap, int forced) {
    if (forced > 0) {
        int w = forced;
        while (w > 1 && rows_per_group % w != 0) w >>= 1;
        return w;
    }
    int w = cap;
    while (w > 1 && rows_per_group % w != 0) w >>= 1;
    // keep >= 2 workgroups per CU in flight when the problem allows it (256 CUs)
    while (w > 4 && total_rows / w < 512) w >>= 1;
    return w;
}

int pick_items_fwd(int L, int N, int forced) {
    if (forced == 4 || forced == 8 || forced == 16) return forced;
    if (L <= 256) return 4;
    if (L <= 768) return 8;
    (void)N;
    return 16;
}

int pick_items_bwd(int L, int forced) {
    if (forced == 4 || forced == 8) return forced;
    return L <= 256 ? 4 : 8;
}

struct Plan { int items, waves, grid; size_t lds; };

Plan plan_fwd(const sigma_scan_fwd_params* p) {
    Plan pl;
    pl.items = pick_items_fwd(p->seqlen, p->dstate, g_opt_fwd_items.load());
    pl.waves = pick_waves(p->dim / p->n_groups, (long)p->batch * p->dim, 16, g_opt_fwd_waves.load());
    pl.grid = (p->dim / pl.waves) * p->batch;
    pl.lds = sigma::fwd_lds_bytes(pl.items, pl.waves, p->dstate);
    return pl;
}

Plan plan_bwd(const sigma_scan_fwd_params* p) {
    Plan pl;
    pl.items = pick_items_bwd(p->seqlen, g_opt_bwd_items.load());
    pl.waves = pick_waves(p->dim / p->n_groups, (long)p->batch * p->dim, 16, g_opt_bwd_waves.load());
    pl.lds = sigma::bwd_lds_bytes(pl.items, pl.waves, p->dstate);
    while (pl.lds > 160 * 1024 && pl.waves > 1) {   // very large dstate: fewer rows per workgroup
        pl.waves >>= 1;
        pl.lds = sigma::bwd_lds_bytes(pl.items, pl.waves, p->dstate);
    }
    pl.grid = (p->dim / pl.waves) * p->batch;
    return pl;
}

}  // namespace

extern "C" {

int sigma_scan_abi_version(void) { return SIGMA_SCAN_ABI_VERSION; }

const char* sigma_scan_last_error(void) { return g_err.c_str(); }

int sigma_scan_set_option(const char* name, int value) {
    if (!name) return fail(SIGMA_ERR_NULL_ARG, "option name is NULL");
    auto pow2 = [](int v) { return v == 0 || v == 1 || v == 2 || v == 4 || v == 8 || v == 16; };
    if (!std::strcmp(name, "fwd_items")) {
        if (!(value == 0 || value == 4 || value == 8 || value == 16)) return fail(SIGMA_ERR_BAD_OPTION, "fwd_items in {0,4,8,16}");
        g_opt_fwd_items = value; return SIGMA_OK;
    }
    if (!std::strcmp(name, "bwd_items")) {
        if (!(value == 0 || value == 4 || value == 8)) return fail(SIGMA_ERR_BAD_OPTION, "bwd_items in {0,4,8}");
        g_opt_bwd_items = value; return SIGMA_OK;
    }
    if (!std::strcmp(name, "fwd_waves")) {
        if (!pow2(value)) return fail(SIGMA_ERR_BAD_OPTION, "fwd_waves in {0,1,2,4,8,16}");
        g_opt_fwd_waves = value; return SIGMA_OK;
    }
    if (!std::strcmp(name, "bwd_waves")) {
        if (!pow2(value)) return fail(SIGMA_ERR_BAD_OPTION, "bwd_waves in {0,1,2,4,8,16}");
        g_opt_bwd_waves = value; return SIGMA_OK;
    }
    return fail(SIGMA_ERR_BAD_OPTION, "unknown option '%s'", name);
}

int sigma_scan_get_option(const char* name) {
    if (!name) return -1;
    if (!std::strcmp(name, "fwd_items")) return g_opt_fwd_items.load();
    if (!std::strcmp(name, "bwd_items")) return g_opt_bwd_items.load();
    if (!std::strcmp(name, "fwd_waves")) return g_opt_fwd_waves.load();
    if (!std::strcmp(name, "bwd_waves")) return g_opt_bwd_waves.load();
    return -1;
}

int sigma_scan_fwd_plan(const sigma_scan_fwd_params* p, int32_t plan[4]) {
    int rc = check_fwd(p, false, false);
    if (rc) return rc;
    if (!plan) return fail(SIGMA_ERR_NULL_ARG, "plan is NULL");
    Plan pl = plan_fwd(p);
    plan[0] = pl.items; plan[1] = pl.waves; plan[2] = pl.grid; plan[3] = (int32_t)pl.lds;
    return SIGMA_OK;
}

int sigma_scan_bwd_plan(const sigma_scan_bwd_params* p, int32_t plan[4]) {
    if (!p) return fail(SIGMA_ERR_NULL_ARG, "params is NULL");
    int rc = check_fwd(&p->fwd, false, false);
    if (rc) return rc;
    if (!plan) return fail(SIGMA_ERR_NULL_ARG, "plan is NULL");
    Plan pl = plan_bwd(&p->fwd);
    plan[0] = pl.items; plan[1] = pl.waves; plan[2] = pl.grid; plan[3] = (int32_t)pl.lds;
    return SIGMA_OK;
}

int sigma_selective_scan_fwd(const sigma_scan_fwd_params* p, void* stream) {
    int rc = check_fwd(p, true);
    if (rc) return rc;
    if (p->batch == 0 || p->seqlen == 0) return SIGMA_OK;
    const Plan pl = plan_fwd(p);
    if (pl.lds > 160 * 1024) return fail(SIGMA_ERR_BAD_SHAPE, "LDS budget exceeded (%zu B)", pl.lds);
    const sigma::FwdArgs a = make_fwd_args(p, pl.waves, vec_ok_fwd(p, true));
    hipError_t e = sigma::launch_scan_fwd(a, p->io_dtype, pl.items, pl.waves, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(SIGMA_ERR_LAUNCH, "scan_fwd launch failed: %s", hipGetErrorString(e));
    return SIGMA_OK;
}

int64_t sigma_scan_bwd_workspace_bytes(const sigma_scan_bwd_params* q) {
    if (!q) { fail(SIGMA_ERR_NULL_ARG, "params is NULL"); return -1; }
    const sigma_scan_fwd_params* p = &q->fwd;
    if (check_fwd(p, false, false)) return -1;
    if (p->batch == 0 || p->seqlen == 0) return 0;
    const Plan pl = plan_bwd(p);
    const int P = (p->dim / p->n_groups) / pl.waves;
    if (P <= 1) return 0;
    return (int64_t)2 * P * p->batch * p->n_groups * (int64_t)p->dstate * p->seqlen * (int64_t)sizeof(float);
}

int sigma_selective_scan_bwd(const sigma_scan_bwd_params* q, void* stream) {
    if (!q) return fail(SIGMA_ERR_NULL_ARG, "params is NULL");
    const sigma_scan_fwd_params* p = &q->fwd;
    int rc = check_fwd(p, false);
    if (rc) return rc;
    if (p->batch == 0 || p->seqlen == 0) return SIGMA_OK;
    if (!q->dout || !q->du || !q->ddelta || !q->dA || !q->dB || !q->dC)
        return fail(SIGMA_ERR_NULL_ARG, "dout/du/ddelta/dA/dB/dC must be non-NULL device pointers");
    if (p->n_chunks > 1 && !p->x)
        return fail(SIGMA_ERR_NULL_ARG, "x (forward checkpoints) is required when n_chunks > 1");
    if ((p->D == nullptr) != (q->dD == nullptr) || (p->delta_bias == nullptr) != (q->ddelta_bias == nullptr))
        return fail(SIGMA_ERR_NULL_ARG, "dD / ddelta_bias must be given exactly when D / delta_bias are");
    const Plan pl = plan_bwd(p);
    if (pl.lds > 160 * 1024) return fail(SIGMA_ERR_BAD_SHAPE, "LDS budget exceeded (%zu B)", pl.lds);
    const int P = (p->dim / p->n_groups) / pl.waves;
    const int64_t slab = (int64_t)P * p->batch * p->n_groups * (int64_t)p->dstate * p->seqlen;
    if (P > 1) {
        if (!q->workspace || q->workspace_bytes < 2 * slab * (int64_t)sizeof(float))
            return fail(SIGMA_ERR_NULL_ARG, "workspace of %lld bytes required (got %lld)",
                        (long long)(2 * slab * (int64_t)sizeof(float)), (long long)q->workspace_bytes);
        if (!aligned_to(q->workspace, 16)) return fail(SIGMA_ERR_BAD_STRIDE, "workspace must be 16-byte aligned");
    }
    const size_t al = 4 * (size_t)elem_size(p->io_dtype);
    bool vec = vec_ok_fwd(p, false) && aligned_to(q->dout, al) && aligned_to(q->du, al) && aligned_to(q->ddelta, al) &&
               q->dout_batch_stride % 4 == 0 && q->dout_d_stride % 4 == 0 && q->du_batch_stride % 4 == 0 &&
               q->du_d_stride % 4 == 0 && q->ddelta_batch_stride % 4 == 0 && q->ddelta_d_stride % 4 == 0;
    sigma::BwdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.f = make_fwd_args(p, pl.waves, vec);
    a.dout = q->dout; a.du = q->du; a.ddelta = q->ddelta;
    a.dA = q->dA; a.dB = q->dB; a.dC = q->dC; a.dD = q->dD; a.dbias = q->ddelta_bias;
    a.g_bs = q->dout_batch_stride; a.g_ds = q->dout_d_stride;
    a.du_bs = q->du_batch_stride; a.du_ds = q->du_d_stride;
    a.dd_bs = q->ddelta_batch_stride; a.dd_ds = q->ddelta_d_stride;
    a.dA_ds = q->dA_d_stride; a.dA_ns = q->dA_dstate_stride;
    a.dB_bs = q->dB_batch_stride; a.dB_gs = q->dB_group_stride; a.dB_ns = q->dB_dstate_stride;
    a.dC_bs = q->dC_batch_stride; a.dC_gs = q->dC_group_stride; a.dC_ns = q->dC_dstate_stride;
    a.P = P;
    a.out_vec_ok = (aligned_to(q->dB, 16) && aligned_to(q->dC, 16) && q->dB_batch_stride % 4 == 0 &&
                    q->dB_group_stride % 4 == 0 && q->dB_dstate_stride % 4 == 0 && q->dC_batch_stride % 4 == 0 &&
                    q->dC_group_stride % 4 == 0 && q->dC_dstate_stride % 4 == 0) ? 1 : 0;
    a.ws_dB = P > 1 ? static_cast<float*>(q->workspace) : nullptr;
    a.ws_dC = P > 1 ? static_cast<float*>(q->workspace) + slab : nullptr;
    hipError_t e = sigma::launch_scan_bwd(a, p->io_dtype, pl.items, pl.waves, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(SIGMA_ERR_LAUNCH, "scan_bwd launch failed: %s", hipGetErrorString(e));
    return SIGMA_OK;
}

int sigma_scan_selftest(void* stream) {
    float* d = nullptr;
    hipError_t e = hipMalloc(&d, 8 * sizeof(float));
    if (e != hipSuccess) return fail(SIGMA_ERR_NO_DEVICE, "hipMalloc failed: %s", hipGetErrorString(e));
    hipStream_t s = static_cast<hipStream_t>(stream);
    float h[8] = {0};
    e = sigma::launch_selftest(d, s);
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(SIGMA_ERR_LAUNCH, "selftest failed to run: %s", hipGetErrorString(e));
    if (h[0] != 0.0f)
        return fail(SIGMA_ERR_LAUNCH,
                    "wave-scan selftest mismatch: fwd=%g rev=%g prev=%g next=%g sum=%g (max abs errors)", h[1], h[2],
                    h[3], h[4], h[5]);
    return SIGMA_OK;
}

}  // extern "C"
