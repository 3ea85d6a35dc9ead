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
#include <cmath>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

std::atomic<int> g_opt_fwd_items{0}, g_opt_fwd_waves{0}, g_opt_fwd_tiles{0}, g_opt_fwd_nb{0};
std::atomic<int> g_opt_bwd_items{0}, g_opt_bwd_waves{0}, g_opt_bwd_nb{0}, g_opt_no_glds{0}, g_opt_bwd_slab2{0}, g_opt_fwd_prefetch{0};
std::atomic<int> g_opt_bwd_gen{0}, g_opt_bwd_rb{0}, g_opt_bwd_touch{0}, g_opt_bwd_sb{0}, g_opt_bwd_wgs{0}, g_opt_fwd_gen{0}, g_opt_bwd_seg{0};
std::atomic<int> g_opt_rl_waves{0}, g_opt_rl_segs{0}, g_opt_rl_chain{0};

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
    if (p->rev_group_mask != 0 && (p->n_groups > 32 || (p->n_groups < 32 && (p->rev_group_mask >> p->n_groups) != 0)))
        return fail(SIGMA_ERR_BAD_SHAPE, "rev_group_mask 0x%x names groups >= n_groups (%d)", p->rev_group_mask, p->n_groups);
    if (p->ckpt_pitch != 0 && p->ckpt_pitch != SIGMA_SCAN_CKPT_PITCH && p->ckpt_pitch != SIGMA_SCAN_CKPT_PITCH_FINE &&
        p->ckpt_pitch != SIGMA_SCAN_CKPT_PITCH_320 && p->ckpt_pitch != SIGMA_SCAN_CKPT_PITCH_160 && p->ckpt_pitch != SIGMA_SCAN_CKPT_PITCH_16)
        return fail(SIGMA_ERR_BAD_SHAPE, "ckpt_pitch must be 0, %d, %d, %d, %d or %d (got %d)", SIGMA_SCAN_CKPT_PITCH,
                    SIGMA_SCAN_CKPT_PITCH_FINE, SIGMA_SCAN_CKPT_PITCH_320, SIGMA_SCAN_CKPT_PITCH_160, SIGMA_SCAN_CKPT_PITCH_16, p->ckpt_pitch);
    if (p->ckpt_pitch != SIGMA_SCAN_CKPT_PITCH_16) {
        const int pitch = p->ckpt_pitch ? p->ckpt_pitch : SIGMA_SCAN_CKPT_PITCH;
        const int64_t need = (int64_t)((p->seqlen + pitch - 1) / pitch) * p->dstate;
        const int64_t have = p->x_row_stride ? p->x_row_stride : (int64_t)p->n_chunks * 2 * p->dstate;
        if (p->x && have < need)
            return fail(SIGMA_ERR_BAD_SHAPE, "x_row_stride %lld too small for %lld checkpoint floats per row",
                        (long long)have, (long long)need);
    }
    if (p->param_group_swap != 0 && (p->param_group_swap != 1 || p->n_groups != 4))
        return fail(SIGMA_ERR_BAD_SHAPE, "param_group_swap must be 0, or 1 with n_groups == 4 (got %d, %d groups)",
                    p->param_group_swap, p->n_groups);
    if (p->u_group_shift < 0 || p->u_group_shift > 5)
        return fail(SIGMA_ERR_BAD_SHAPE, "u_group_shift must be in [0, 5] (got %d)", p->u_group_shift);
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

sigma::FwdArgs make_fwd_args(const sigma_scan_fwd_params* p, int R, int W, int NB, bool vec) {
    sigma::FwdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.u = p->u; a.delta = p->delta; a.A = p->A; a.B = p->B; a.C = p->C; a.D = p->D; a.bias = p->delta_bias;
    a.out = p->out; a.x = p->x;
    a.batch = p->batch; a.dim = p->dim; a.L = p->seqlen; a.N = p->dstate; a.G = p->n_groups;
    a.n_chunks = p->n_chunks; a.rows_per_group = p->dim / p->n_groups; a.softplus = p->delta_softplus ? 1 : 0;
    a.vec_ok = vec ? 1 : 0; a.rowblocks = p->dim / R;
    a.R = R; a.W = W; a.NB = NB;
    a.rev_mask = p->rev_group_mask;
    a.u_gshift = p->u_group_shift;
    a.pswap = p->param_group_swap;
    a.ckpt_pitch = p->ckpt_pitch ? p->ckpt_pitch : sigma::kCkptPitch;
    a.x_rs = p->x_row_stride ? p->x_row_stride : (long)p->n_chunks * 2 * p->dstate;
    a.u_bs = p->u_batch_stride; a.u_ds = p->u_d_stride; a.dt_bs = p->delta_batch_stride; a.dt_ds = p->delta_d_stride;
    a.A_ds = p->A_d_stride; a.A_ns = p->A_dstate_stride;
    a.B_bs = p->B_batch_stride; a.B_gs = p->B_group_stride; a.B_ns = p->B_dstate_stride;
    a.C_bs = p->C_batch_stride; a.C_gs = p->C_group_stride; a.C_ns = p->C_dstate_stride;
    a.o_bs = p->out_batch_stride; a.o_ds = p->out_d_stride;
    return a;
}

// ---------------------------------------------------------------- launch planning
// Work per element-state in VALU issue slots (fold + replay + one wave scan per T elements):
// used only to rank tile lengths against the padding they cause.
double fwd_cost_per_element(int T) { return 7.0 + 24.0 / T; }
double bwd_cost_per_element(int T) { return 18.0 + 48.0 / T; }

int pick_items(int L, const int* cand, int ncand, double (*cost)(int), int forced) {
    for (int i = 0; i < ncand; ++i) if (forced == cand[i]) return forced;
    int best = cand[0];
    double best_score = 1e300;
    for (int i = 0; i < ncand; ++i) {
        const int T = cand[i];
        const long tile = 64L * T;
        const long padded = (L + tile - 1) / tile * tile;
        const double score = (double)padded * cost(T);
        if (score < best_score) { best_score = score; best = T; }
    }
    return best;
}

constexpr int kCUs = 256;
constexpr size_t kLdsLimit = 160 * 1024;

struct Plan { int items, rows, tiles, nb, grid; bool glds; size_t lds; bool slab2; };

// global_load_lds staging: f32, 16-byte aligned rows, whole chunks in range, 31-bit offsets inside
// one (batch, group) slice; the per-wave chunk plan holds kStageMaxIt units (scan_device.h).
bool glds_ok(const sigma_scan_fwd_params* p, bool vec) {
    const int64_t span = (int64_t)p->dstate * (p->B_dstate_stride > p->C_dstate_stride ? p->B_dstate_stride : p->C_dstate_stride) + p->seqlen;
    return vec && p->io_dtype == SIGMA_DTYPE_F32 && (p->seqlen % 4) == 0 && g_opt_no_glds.load() == 0 &&
           p->B_dstate_stride >= 0 && p->C_dstate_stride >= 0 && span * 4 < (int64_t(1) << 31);
}
bool glds_fits(int T, int NB, int W, int nwaves) {
    const int units = (NB * W * 16 * T + 63) / 64;
    return units <= sigma::kStageMaxIt * nwaves;
}

// Forward: rows R x tiles W per workgroup (R*W <= 16 waves, R divides the rows of a group).
// cost ~ (rounds of workgroups over the 256 CUs) x (super-tiles each walks) x (work per
// super-tile step, which grows with the B/C restaging share 1/R and the extra barrier of W > 1).
Plan plan_fwd(const sigma_scan_fwd_params* p, bool vec) {
    // T = 20 (one 1280 tile) exists but needs ~150 VGPRs -> 12-wave workgroups; it is only used
    // when forced ("fwd_items" = 20) until it measures faster than two 640 tiles.
    static const int cand[] = {10, 5, 4};
    Plan pl;
    const bool fine = p->ckpt_pitch == SIGMA_SCAN_CKPT_PITCH_FINE;    // tiles must divide 640: T in {10, 5}
    int forced_items = g_opt_fwd_items.load();
    if (fine && (forced_items == 20 || forced_items == 4)) forced_items = 0;
    // measured (tools/bwd_variants.py): with 16 states and long sequences the 1280-element tile wins
    // (half as many tile starts whose u/delta latency is exposed; 805 vs 961 us at (8,768,19200));
    // short sequences and few-state scans prefer 640-element tiles with the register prefetch
    const bool long_rows = (long)p->batch * p->dim >= 12L * kCUs && p->dstate > 8 && p->seqlen >= 10240;
    pl.items = forced_items == 20 ? 20
             : (forced_items == 0 && long_rows) ? 20
             : pick_items(p->seqlen, cand, fine ? 2 : 3, fwd_cost_per_element, forced_items);
    pl.glds = glds_ok(p, vec);
    const int rpg = p->dim / p->n_groups;
    const long total_rows = (long)p->batch * p->dim;
    const int tile = 64 * pl.items;
    const int ntiles = (p->seqlen + tile - 1) / tile;
    const int fr = g_opt_fwd_waves.load(), fw = g_opt_fwd_tiles.load(), fnb = g_opt_fwd_nb.load();
    double best = 1e300;
    pl.rows = 1; pl.tiles = 1; pl.nb = 1; pl.slab2 = false;
    const int maxw = pl.items >= 20 ? 12 : 16;                  // scan_fwd.hip: fwd_max_waves<T>
    for (int R = maxw; R >= 1; --R) {
        if (rpg % R != 0) continue;
        if (fr > 0 && R != fr && rpg % fr == 0 && fr <= maxw) continue;   // forced rows (when legal)
        int Wmax = maxw / R;
        if (Wmax > ntiles) Wmax = ntiles;
        if (Wmax < 1) Wmax = 1;
        for (int W = Wmax; W >= 1; --W) {
            if (fw > 0 && W != (fw < Wmax ? fw : Wmax)) continue;
            if (fw == 0 && W > 1 && total_rows / maxw >= kCUs) continue;   // enough rows: never split the sequence
            int NB = W == 1 ? 4 : (W == 2 ? 2 : 1);
            if (fnb > 0) NB = fnb;
            if (NB > p->dstate) NB = p->dstate;
            while (NB > 1 && sigma::fwd_lds_bytes(pl.items, R, W, NB, p->dstate) > kLdsLimit) NB >>= 1;
            if (sigma::fwd_lds_bytes(pl.items, R, W, NB, p->dstate) > kLdsLimit) continue;
            const long nwg = total_rows / R;
            const int nsuper = (ntiles + W - 1) / W;
            const int waves = R * W;
            // two 8-wave workgroups share a CU (<= 128 VGPRs, ~40 KB LDS each): their barriers and
            // staging waits interleave, measured 12-25 % faster than one 16-wave workgroup
            const size_t lds = sigma::fwd_lds_bytes(pl.items, R, W, NB, p->dstate);
            int wgpc = 16 / waves;
            if ((size_t)wgpc * lds > kLdsLimit) wgpc = (int)(kLdsLimit / lds);
            if (wgpc < 1) wgpc = 1;
            if (wgpc > 2) wgpc = 2;
            const long rounds = (nwg + (long)kCUs * wgpc - 1) / ((long)kCUs * wgpc);
            // a CU with few waves runs each of them faster, but not proportionally
            // (with <= 8 states the staging share per row matters more than the interleaving: 16 rows win)
            const double occ = p->dstate > 8 ? (0.35 + 0.65 * (double)(waves * wgpc) / 16.0) : 1.0;
            const double per_step = occ * (1.0 + 2.0 / R + (W > 1 ? 0.15 : 0.0)) * ((wgpc >= 2 && p->dstate > 8 && p->seqlen < 10240) ? 0.85 : 1.0);   // long rows: 16-row workgroups measured equal or better
            const double cost = (double)rounds * nsuper * per_step;
            if (cost < best) { best = cost; pl.rows = R; pl.tiles = W; pl.nb = NB; }
        }
    }
    pl.grid = (int)(total_rows / pl.rows);
    pl.lds = sigma::fwd_lds_bytes(pl.items, pl.rows, pl.tiles, pl.nb, p->dstate);
    pl.glds = pl.glds && glds_fits(pl.items, pl.nb, pl.tiles, pl.rows * pl.tiles);
    return pl;
}

Plan plan_bwd(const sigma_scan_fwd_params* p, bool vec) {
    static const int cand[] = {10, 5, 4};
    static const int cand320[] = {5};
    Plan pl;
    const bool fine = p->ckpt_pitch == SIGMA_SCAN_CKPT_PITCH_FINE;
    const bool p320 = p->ckpt_pitch == SIGMA_SCAN_CKPT_PITCH_320;     // tiles must divide the pitch
    int forced_bitems = g_opt_bwd_items.load();
    if (fine && forced_bitems == 4) forced_bitems = 0;
    if (p320) forced_bitems = 0;
    pl.items = p320 ? pick_items(p->seqlen, cand320, 1, bwd_cost_per_element, 0)
                    : pick_items(p->seqlen, cand, fine ? 2 : 3, bwd_cost_per_element, forced_bitems);
    pl.glds = glds_ok(p, vec);
    pl.tiles = 1;
    pl.slab2 = false;
    const int rpg = p->dim / p->n_groups;
    const long total_rows = (long)p->batch * p->dim;
    const int fr = g_opt_bwd_waves.load();
    int NB = g_opt_bwd_nb.load() > 0 ? g_opt_bwd_nb.load() : 4;
    if (NB > p->dstate) NB = p->dstate;
    double best = 1e300;
    pl.rows = 1; pl.nb = NB;
    const int maxw = pl.items >= 10 ? sigma::kBwdMaxWavesT10 : 16;   // scan_bwd.hip: bwd_max_waves<T>
    for (int R = maxw; R >= 1; --R) {
        if (rpg % R != 0) continue;
        if (fr > 0 && R != fr && rpg % fr == 0 && fr <= maxw) continue;   // forced rows (when legal)
        if (sigma::bwd_lds_bytes(pl.items, R, NB, p->dstate) > kLdsLimit) continue;
        // measured (tools/scan_bench.py --sweep): with enough rows the largest workgroup wins (B/C
        // staging and the dB/dC column sums are per-workgroup costs); with few rows 8 rows per
        // workgroup beat both 3 (fixed costs dominate) and 12 (too few workgroups)
        const long nwg = total_rows / R;
        const bool many = total_rows / maxw >= kCUs;
        const double cost = many ? (double)(maxw - R) : (R <= 8 ? (double)(8 - R) : 100.0 + R);
        (void)nwg;
        if (cost < best) { best = cost; pl.rows = R; }
    }
    pl.grid = (int)(total_rows / pl.rows);
    pl.slab2 = g_opt_bwd_slab2.load() == 1 && sigma::bwd_lds_bytes(pl.items, pl.rows, pl.nb, p->dstate, true) <= kLdsLimit;
    pl.lds = sigma::bwd_lds_bytes(pl.items, pl.rows, pl.nb, p->dstate, pl.slab2);
    pl.glds = pl.glds && glds_fits(pl.items, pl.nb, 1, pl.rows);
    return pl;
}

// scan_bwd2 (scan_bwd2.hip): one checkpoint per backward tile, dstate <= 64.  A workgroup is R rows x
// RB row blocks of one (batch, group); P = rows_per_group / (R * RB) workgroups share a group.
struct Plan2 { bool ok; int items, rows, nb, RB, P, nacc, grid; bool glds, slab2; size_t lds; };

Plan2 plan_bwd2(const sigma_scan_fwd_params* p, bool vec) {
    Plan2 pl;
    std::memset(&pl, 0, sizeof(pl));
    const int gen = g_opt_bwd_gen.load();
    if (gen == 1) return pl;                                         // forced: first-generation kernel
    const int pitch = p->ckpt_pitch ? p->ckpt_pitch : SIGMA_SCAN_CKPT_PITCH;
    const int T = pitch == SIGMA_SCAN_CKPT_PITCH_FINE ? 10 : (pitch == SIGMA_SCAN_CKPT_PITCH_320 ? 5 : 0);
    if (T == 0 || p->dstate > 64) return pl;
    const int forced_items = g_opt_bwd_items.load();
    if (forced_items != 0 && forced_items != T && gen != 2) return pl;
    const int tile = 64 * T;
    const int rpg = p->dim / p->n_groups;
    const int fr = g_opt_bwd_waves.load();
    // measured (tools/bwd2_check.py bench, profiles/r02_bwd2_variants.txt): 640-tiles run best with 12-row workgroups
    // (168 VGPRs, no spills; the LDS accumulators of the row-block loop do not fit beside 16 slabs), 320-tiles with 16
    static const int pref12[] = {12, 16, 8, 6, 10, 14, 15, 11, 13, 9, 7, 5};
    static const int pref16[] = {16, 12, 8, 6, 10, 14, 15, 11, 13, 9, 7, 5};
    const int* pref = T == 5 ? pref16 : pref12;
    int R = 0;
    if (fr > 0 && fr <= 16 && rpg % fr == 0 && fr * 64 >= tile) R = fr;
    for (int i = 0; R == 0 && i < 12; ++i)
        if (rpg % pref[i] == 0 && pref[i] * 64 >= tile) R = pref[i];
    if (R == 0) return pl;
    int NB = g_opt_bwd_nb.load() > 0 ? g_opt_bwd_nb.load() : 4;
    if (NB > p->dstate) NB = p->dstate;
    const int rowblocks = rpg / R;
    // Row blocks per workgroup ("bwd_rb"): the dB/dC sums of RB row blocks meet in LDS accumulators
    // (2 * N * tile floats) before they leave the workgroup: RB times fewer partial slabs in the workspace.
    // as many row blocks per workgroup as still leave one workgroup per CU (measured: -5 ... -15 % on the
    // 16-state shapes, and the reduce_partials pass shrinks with the slab count); "bwd_rb" forces a value
    int RB = 1;
    const int frb = g_opt_bwd_rb.load();
    if (frb > 0) {
        RB = frb;
        while (RB > 1 && rowblocks % RB != 0) --RB;
    } else {
        for (int d = 1; d <= rowblocks; ++d)
            if (rowblocks % d == 0 && (long)p->batch * p->n_groups * (rowblocks / d) >= kCUs) RB = d;
    }
    const int nacc = 0;
    const int sl = g_opt_bwd_slab2.load();
    bool slab2 = sl != 2;                                            // two slab sets when they fit beside NB = 4
    while (sigma::bwd2_lds_bytes(T, R, NB, p->dstate, slab2, RB) > kLdsLimit) {
        if (slab2) slab2 = false;
        else if (NB > 1) NB >>= 1;
        else if (RB > 1) { --RB; while (RB > 1 && rowblocks % RB != 0) --RB; }
        else return pl;
    }
    pl.ok = true;
    pl.items = T; pl.rows = R; pl.nb = NB; pl.RB = RB; pl.P = rowblocks / RB; pl.nacc = nacc;
    pl.grid = p->batch * p->n_groups * pl.P;
    pl.slab2 = slab2;
    pl.lds = sigma::bwd2_lds_bytes(T, R, NB, p->dstate, slab2, RB);
    pl.glds = glds_ok(p, vec) && glds_fits(T, NB, 1, R);
    return pl;
}

// scan_bwd3 (scan_bwd3.hip): state-parallel mapping, 320-element tiles; a workgroup is `slots` row slots x
// Q = dstate/4 waves and walks RB rows per slot; P = rows_per_group / (slots * RB) workgroups share a group.
struct Plan3 { bool ok; int nw, slots, RB, P, grid; bool glds; size_t lds; };

Plan3 plan_bwd3(const sigma_scan_fwd_params* p, bool vec) {
    Plan3 pl;
    std::memset(&pl, 0, sizeof(pl));
    const int gen = g_opt_bwd_gen.load();
    if (gen == 1 || gen == 2) return pl;
    if (p->ckpt_pitch != SIGMA_SCAN_CKPT_PITCH_320) return pl;
    const int N = p->dstate;
    if (N < 4 || N > 16 || (N & 3) != 0) return pl;
    // measured (profiles/r02_bwd3_variants.txt): the state-parallel mapping wins where a workgroup walks many
    // rows per tile and a wave holds the whole row -- 4 states, short sequences (dec (8,3072,1200,N4): 243 vs
    // 323 us) -- and loses with 16 states (1650 vs 1300 us) and on long rows; "bwd_gen" = 3 forces it
    const bool auto_ok = N <= 4 && p->seqlen <= 1280;
    if (gen != 3 && !auto_ok) return pl;
    const int Q = N / 4;
    const int rpg = p->dim / p->n_groups;
    int maxw = g_opt_bwd_waves.load();
    if (maxw <= 0 || maxw > 16) maxw = 12;                            // 12 waves: 168-VGPR build, no spills
    int slots = 0;
    for (int s = maxw / Q; s >= 1; --s) if (rpg % s == 0) { slots = s; break; }
    if (slots == 0) return pl;
    const int rowsteps = rpg / slots;
    int RB = 1;
    const int frb = g_opt_bwd_rb.load();
    if (frb > 0) {
        RB = frb;
        while (RB > 1 && rowsteps % RB != 0) --RB;
    } else {
        for (int d = 1; d <= rowsteps; ++d)
            if (rowsteps % d == 0 && (long)p->batch * p->n_groups * (rowsteps / d) >= kCUs) RB = d;
    }
    while (RB > 1 && sigma::bwd3_lds_bytes(slots * Q, N, RB) > kLdsLimit) { --RB; while (RB > 1 && rowsteps % RB != 0) --RB; }
    if (gen != 3 && RB < 4) return pl;                                // too few rows per workgroup to amortise a tile
    if (sigma::bwd3_lds_bytes(slots * Q, N, RB) > kLdsLimit) return pl;
    pl.ok = true;
    pl.nw = slots * Q; pl.slots = slots; pl.RB = RB; pl.P = rowsteps / RB;
    pl.grid = p->batch * p->n_groups * pl.P;
    pl.lds = sigma::bwd3_lds_bytes(pl.nw, N, RB);
    pl.glds = glds_ok(p, vec) && glds_fits(5, N, 1, pl.nw);
    return pl;
}

// scan_fwd4 (scan_fwd4.hip): quad-row forward for ckpt_pitch 160; a workgroup is W <= 8 waves x 4 rows of one
// (batch, group) sharing the B/C image of a tile.  Two 8-wave workgroups per CU (41 KB LDS each at N = 16).
struct PlanF4 { bool ok; int W, P, grid; size_t lds; };

PlanF4 plan_fwd4(const sigma_scan_fwd_params* p, bool vec) {
    PlanF4 pl;
    std::memset(&pl, 0, sizeof(pl));
    if (p->ckpt_pitch != SIGMA_SCAN_CKPT_PITCH_160 || g_opt_fwd_gen.load() == 1) return pl;
    const int N = p->dstate;
    if (N != 4 && N != 8 && N != 16) return pl;                       // the state counts the model uses (and the tests cover)
    if (!glds_ok(p, vec)) return pl;
    const int rpg = p->dim / p->n_groups;
    if (rpg % 4 != 0) return pl;
    // a wave keeps its four rows for the whole sequence: with few rows (the launches whose backward splits the sequence
    // instead) the 64-lane kernel with its in-workgroup sequence split is faster ((2,768,19200): 297 vs 715 us)
    if ((long)p->batch * p->dim < 8192 && g_opt_fwd_gen.load() != 2) return pl;
    const int quads = rpg / 4;
    const int fr = g_opt_fwd_waves.load();
    const long bg = (long)p->batch * p->n_groups;
    int W = 0;
    if (fr > 0 && fr <= 8 && quads % fr == 0) W = fr;
    for (int w = 8; W == 0 && w >= 1; --w) {
        if (quads % w != 0) continue;
        if (bg * (quads / w) >= 2 * kCUs || w == 1) W = w;               // two workgroups per CU wanted
    }
    if (W == 0) return pl;
    pl.ok = true;
    pl.W = W; pl.P = quads / W;
    pl.grid = (int)(bg * pl.P);
    pl.lds = sigma::fwd4_lds_bytes(N);
    return pl;
}

// scan_bwd4 (scan_bwd4.hip): quad-row mapping, 160-position tiles; a workgroup is W waves x 4 rows and walks RB
// row blocks per tile; P = rows_per_group / (4 * W * RB) workgroups share a group; SB states share a barrier.
struct Plan4 { bool ok; int W, RB, SB, P, grid, nbuf, wgs, S, seg_tiles; size_t lds; };

Plan4 plan_bwd4(const sigma_scan_fwd_params* p, bool vec) {
    Plan4 pl;
    std::memset(&pl, 0, sizeof(pl));
    if (p->ckpt_pitch != SIGMA_SCAN_CKPT_PITCH_160) return pl;
    const int N = p->dstate;
    if (N != 4 && N != 8 && N != 16) return pl;                       // the state counts the model uses (and the tests cover)
    if (!glds_ok(p, vec)) return pl;                                  // f32, aligned B/C: the kernel stages by LDS-DMA only
    const int rpg = p->dim / p->n_groups;
    if (rpg % 4 != 0) return pl;
    const int quads = rpg / 4;                                       // wave-sized row quads per group
    const int fr = g_opt_bwd_waves.load();
    // Geometry = (waves W per workgroup, row blocks RB it walks per tile).  Estimated time ~ rounds of workgroups
    // over the 256 CUs x rows a workgroup walks per tile x cost per row, where the per-row cost falls with the
    // workgroup size (B/C staging, barriers and column sums are per-workgroup costs; 16 waves = four per SIMD in
    // the 128-VGPR build): measured on (16,3072,1200,N16) 918 / 979 / 1185 us for W = 16 / 12 / 8 at equal grids
    // (profiles/r02_bwd4_shapes.txt).  Ties go to more row blocks (fewer partial dB/dC slabs).
    // "bwd_wgs" = 2: two workgroups of <= 8 waves per CU (<= 80 KB LDS each: one B/C image, one state per barrier).
    const int frb = g_opt_bwd_rb.load();
    const int wgs = g_opt_bwd_wgs.load() == 2 ? 2 : 1;
    const long slots = (long)kCUs * wgs;
    const long bg = (long)p->batch * p->n_groups;
    // Sequence segments ("bwd_seg": 0 = automatic, 1 = never, k = k segments): with few rows the workgroups cannot
    // fill the chip, so the sequence is cut into S segments run by different workgroups; a pre-pass
    // (rev_summary4_kernel, ~25 % of the work) provides the reverse carry entering each segment.
    const int ntiles = (p->seqlen + 159) / 160;
    int fseg = g_opt_bwd_seg.load();
    auto seg_legal = [&](int s) {
        return s == 1 || ((ntiles + s - 1) / s >= 2 && ((ntiles + s - 1) / s) * (s - 1) < ntiles);   // >= 2 tiles each, none empty
    };
    if (fseg > 1 && !seg_legal(fseg)) fseg = 1;                              // a forced count that cannot be served
    static const int seg_cand[] = {1, 2, 3, 4, 6, 8, 12, 16};
    int W = 0, RB = 1, S = 1;
    double best = 1e300;
    for (int si = 0; si < 8; ++si) {
        const int s = seg_cand[si];
        if (fseg == 1 && s != 1) continue;
        if (fseg > 1 && s != fseg) continue;
        if (!seg_legal(s)) continue;
        for (int w = wgs == 2 ? 8 : 16; w >= 1; --w) {
            if (quads % w != 0) continue;
            if (fr > 0 && fr <= 16 && quads % fr == 0 && w != fr) continue;      // forced waves (when legal)
            const int rowblocks_w = quads / w;
            for (int d = rowblocks_w; d >= 1; --d) {
                if (rowblocks_w % d != 0) continue;
                if (frb > 0) {                                                   // forced row blocks: largest divisor <= frb
                    int want = frb;
                    while (want > 1 && rowblocks_w % want != 0) --want;
                    if (d != want) continue;
                }
                if ((size_t)d * 4 * w * N * sizeof(float) > (wgs == 2 ? 8u : 24u) * 1024) continue;   // reverse carries of the chunk's rows
                const long grid = bg * (rowblocks_w / d) * s;
                const double rounds = (double)((grid + slots - 1) / slots);
                const double cost = rounds * d * w * (0.73 + 4.3 / w) / s * (s > 1 ? 1.25 : 1.0);
                if (cost < best * 0.999) { best = cost; W = w; RB = d; S = s; }
            }
        }
    }
    if (W == 0) return pl;
    const int rowblocks = quads / W;
    const size_t lds_limit = wgs == 2 ? kLdsLimit / 2 : kLdsLimit;
    int nbuf = (wgs == 2 || g_opt_bwd_nb.load() == 1) ? 1 : 2;       // "bwd_nb" = 1: one B/C image
    int SB = g_opt_bwd_sb.load() > 0 ? g_opt_bwd_sb.load() : (wgs == 2 ? 1 : 2);
    while (SB > 1 && N % SB != 0) SB >>= 1;
    while (sigma::bwd4_lds_bytes(W, N, SB, RB, nbuf) > lds_limit) {
        if (SB > 2) SB >>= 1;
        else if (nbuf > 1) nbuf = 1;                                 // measured: the second image buys 3 %, SB = 2 buys 5 %
        else if (SB > 1) SB >>= 1;
        else if (RB > 1) { --RB; while (RB > 1 && rowblocks % RB != 0) --RB; }
        else return pl;
    }
    pl.nbuf = nbuf; pl.wgs = wgs;
    pl.S = S; pl.seg_tiles = (ntiles + S - 1) / S;
    pl.ok = true;
    pl.W = W; pl.RB = RB; pl.SB = SB; pl.P = rowblocks / RB;
    pl.grid = p->batch * p->n_groups * pl.P * S;
    pl.lds = sigma::bwd4_lds_bytes(W, N, SB, RB, nbuf);
    return pl;
}

// Row-lane kernels (scan_fwdr.hip / scan_bwdr.hip, ckpt_pitch 16): a workgroup is one 64-row block of a (batch, group)
// x NW state waves (dstate / NW states each); with few rows the sequence is cut into S segments of seg_tiles 16-position
// tiles (a summary pre-pass provides the state / reverse carry entering a segment).
struct PlanR { bool ok; int NW, P, S, seg_tiles, grid; };

bool rowlane_legal(const sigma_scan_fwd_params* p, bool vec) {
    const int N = p->dstate;
    if (p->ckpt_pitch != SIGMA_SCAN_CKPT_PITCH_16) return false;
    if (N != 4 && N != 8 && N != 16) return false;
    if (p->io_dtype != SIGMA_DTYPE_F32 || !vec || (p->seqlen % 4) != 0) return false;
    if ((p->dim / p->n_groups) % 64 != 0) return false;
    const int64_t span = (int64_t)N * (p->B_dstate_stride > p->C_dstate_stride ? p->B_dstate_stride : p->C_dstate_stride) + p->seqlen;
    return p->B_dstate_stride >= 0 && p->C_dstate_stride >= 0 && span * 4 < (int64_t(1) << 31);
}

PlanR plan_rowlane(const sigma_scan_fwd_params* p, bool vec, bool backward) {
    PlanR pl;
    std::memset(&pl, 0, sizeof(pl));
    if (!rowlane_legal(p, vec)) return pl;
    const int N = p->dstate;
    const int P = (p->dim / p->n_groups) / 64;
    const long nwg0 = (long)p->batch * p->n_groups * P;
    const int ntiles = (p->seqlen + 15) / 16;
    // candidates (state waves, workgroups a CU holds: registers / LDS of the builds, see the kernel files)
    struct Cand { int nw, wgpc; };
    Cand cf16[] = {{4, 3}, {8, 2}, {16, 1}}, cf8[] = {{4, 4}, {8, 2}}, cf4[] = {{4, 4}};
    Cand cb16[] = {{4, 2}}, cb8[] = {{4, 2}}, cb4[] = {{4, 2}};
    const Cand* cand = backward ? (N == 16 ? cb16 : N == 8 ? cb8 : cb4) : (N == 16 ? cf16 : N == 8 ? cf8 : cf4);
    const int ncand = backward ? 1 : (N == 16 ? 3 : N == 8 ? 2 : 1);
    const int fw = g_opt_rl_waves.load(), fs = g_opt_rl_segs.load();
    // per-wave cost of one tile in issue clocks: NS states x 16 positions x clocks per element-state + fixed part
    const double ces = backward ? 55.0 : 17.7, fixed = backward ? 600.0 : 360.0, pre = backward ? 1.45 : 2.0;
    double best = 1e300;
    for (int ci = 0; ci < ncand; ++ci) {
        const int nw = cand[ci].nw;
        if (fw != 0 && fw != nw) {
            bool legal = false;
            for (int cj = 0; cj < ncand; ++cj) legal = legal || cand[cj].nw == fw;
            if (legal) continue;                                   // a forced value this build has
        }
        const long slots = (long)kCUs * cand[ci].wgpc;
        for (int S = 1; S <= 64; ++S) {
            if (fs != 0 && S != fs && !(fs > 1 && (ntiles + fs - 1) / fs < 2)) continue;
            const int st = (ntiles + S - 1) / S;
            if (S > 1 && (st < 2 || (long)st * (S - 1) >= ntiles)) continue;      // >= 2 tiles each, none empty
            const long nwg = nwg0 * S;
            const double rounds = (double)((nwg + slots - 1) / slots);
            // resident waves per SIMD ON THE BUSIEST CU: a launch that fits one round is as slow as the CUs that hold one
            // workgroup more than the others -- (1,768,19200) forward with 48 segments = 576 workgroups (three on 64 CUs,
            // two on the rest) ran 170 us, with 40 segments = 480 (two at most) 149 us (round 6, rl_segs sweeps:
            // profiles/r06_rowlane_segment_sweep.txt).  A third workgroup on a CU adds 67 % to a tile's time, not 50 %
            // (64 segments: 8.3 us per tile against 5.0 with two), and every further round costs its tail and refill
            // ((8,192,19200,N4) backward: 64 segments in three rounds 344 us, 21 in one 284).
            // The backward keeps the chip-average occupancy: its summary pre-pass costs more per segment, and on the small
            // one-image shapes fewer, longer segments measured better than an even fill ((1,768,9600,N4): 30 segments 82 us, 40: 88).
            const long per_cu = nwg <= slots ? (nwg + kCUs - 1) / kCUs : cand[ci].wgpc;
            double wres = backward ? (double)(nwg < slots ? nwg : slots) * nw / (4.0 * kCUs)
                                   : (double)per_cu * nw / 4.0 * (per_cu >= 3 && N >= 8 ? 1.1 : 1.0);   // 4 states: a lighter wave, the third / fourth workgroup scales
            if (wres < 1.9) wres = 1.9;                            // a lone wave issues every ~4.5 clocks
            const double t = rounds * (rounds > 1.0 ? 1.15 : 1.0) * st * wres * ((N / nw) * 16 * ces + fixed) * (S > 1 ? pre : 1.0);
            if (t < best * 0.999) { best = t; pl.NW = nw; pl.S = S; pl.seg_tiles = st; }
        }
    }
    if (pl.NW == 0) return pl;
    pl.ok = true;
    pl.P = P;
    pl.grid = (int)(nwg0 * pl.S);
    return pl;
}

int64_t rowlane_summary_floats(const sigma_scan_fwd_params* p, int S) {
    return S > 1 ? (int64_t)(S - 1) * p->batch * p->dim * (int64_t)p->dstate * 2 : 0;
}

// chained walk of the row-lane backward: hand-over slots [row blocks][N][64] floats + one flag per row block (reserved
// for every row-lane backward, so that the workspace size does not depend on the device's occupancy answer)
int64_t rowlane_chain_floats(const sigma_scan_fwd_params* p) {
    const int64_t nrb = (int64_t)p->batch * (p->dim / 64);
    return nrb * p->dstate * 64 + ((nrb + 3) / 4) * 4;
}

// tiles per workgroup of the chained walk, or 0: used when the row blocks do not fill whole rounds of resident
// workgroups (768 blocks on 512 slots = two rounds, the second half empty: 1.5 rounds' worth of work in the time of 2)
int rowlane_chain_tiles(const sigma_scan_fwd_params* p, const PlanR& pr) {
    const int mode = g_opt_rl_chain.load();
    // measured (profiles/r04_rowlane_chain.txt): (16,3072,1200,N16) 800 us chained against 779 us in 1.5 plain rounds --
    // the kernel is bound by the issue rate of a SIMD, and the workgroups of a half-empty last round simply run faster;
    // the walk is kept for launches that are latency-bound per wave, on request only
    if (mode != 2 || pr.S != 1) return 0;
    const long nrb = (long)p->batch * p->n_groups * pr.P;
    const int ntiles = (p->seqlen + 15) / 16;
    const long cap = (long)kCUs * sigma::bwdr_resident_per_cu(p->dstate);
    if (nrb <= cap || nrb * ntiles >= (1L << 31)) return 0;
    const long rounds = (nrb + cap - 1) / cap;
    (void)rounds;
    return (int)((nrb * ntiles + cap - 1) / cap);
}

}  // namespace

extern "C" {

int sigma_scan_abi_version(void) { return SIGMA_SCAN_ABI_VERSION; }

const char* sigma_scan_last_error(void) { return g_err.c_str(); }

namespace {
struct OptDesc { const char* name; std::atomic<int>* var; int allowed[8]; };
OptDesc g_opts[] = {
    {"fwd_items", &g_opt_fwd_items, {0, 4, 5, 10, 20, -1}},
    {"bwd_items", &g_opt_bwd_items, {0, 4, 5, 10, -1}},
    {"fwd_waves", &g_opt_fwd_waves, {-2}},     // rows per workgroup, 0..16
    {"bwd_waves", &g_opt_bwd_waves, {-2}},
    {"fwd_tiles", &g_opt_fwd_tiles, {-2}},     // sequence tiles per workgroup, 0..16
    {"fwd_nb", &g_opt_fwd_nb, {0, 1, 2, 4, 8, -1}},
    {"bwd_nb", &g_opt_bwd_nb, {0, 1, 2, 4, 8, -1}},
    {"no_glds", &g_opt_no_glds, {0, 1, -1}},
    {"bwd_slab2", &g_opt_bwd_slab2, {0, 1, 2, -1}},        // 1 = two dB/dC slab sets when they fit
    {"fwd_prefetch", &g_opt_fwd_prefetch, {0, 1, 2, -1}},  // 2 = no register prefetch of the next tile's u/delta (T = 10)
    {"bwd_gen", &g_opt_bwd_gen, {0, 1, 2, 3, -1}},         // backward kernel: 1 = scan_bwd.hip, 2 = scan_bwd2.hip, 3 / 0 = best legal
    {"bwd_seg", &g_opt_bwd_seg, {0, 1, 2, 3, 4, 6, 8, 12}},     // quad-row backward: sequence segments (0 auto, 1 never)
    {"fwd_gen", &g_opt_fwd_gen, {0, 1, 2, -1}},              // 1 = never the quad-row forward (scan_fwd4.hip), 2 = also with few rows
    {"bwd_wgs", &g_opt_bwd_wgs, {0, 1, 2, -1}},              // quad-row backward: 2 = two small workgroups per CU
    {"bwd_sb", &g_opt_bwd_sb, {0, 1, 2, 4, 8, -1}},          // quad-row backward: states per barrier (0 = 2)
    {"bwd_touch", &g_opt_bwd_touch, {0, 1, 2, -1}},   // L2 warm-up touches of the next row step: 1 = on, 2 = off, 0 = on in scan_bwd4 only
    {"bwd_rb", &g_opt_bwd_rb, {-3}},                       // scan_bwd2: row blocks per workgroup, 0..256
    {"rl_waves", &g_opt_rl_waves, {0, 4, 8, 16, -1}},       // row-lane kernels: state waves per 64-row block
    {"rl_segs", &g_opt_rl_segs, {-4}},                      // row-lane kernels: sequence segments, 0..64
    {"rl_chain", &g_opt_rl_chain, {0, 1, 2, -1}},           // row-lane backward: chained walk 1 = never, 2 = whenever legal
};
}  // namespace

int sigma_scan_set_option(const char* name, int value) {
    if (!name) return fail(SIGMA_ERR_NULL_ARG, "option name is NULL");
    for (auto& o : g_opts) {
        if (std::strcmp(name, o.name)) continue;
        bool ok = false;
        if (o.allowed[0] == -2) ok = value >= 0 && value <= 16;
        else if (o.allowed[0] == -3) ok = value >= 0 && value <= 256;
        else if (o.allowed[0] == -4) ok = value >= 0 && value <= 64;
        else for (int i = 0; i < 8 && o.allowed[i] != -1; ++i) ok = ok || o.allowed[i] == value;
        if (!ok) return fail(SIGMA_ERR_BAD_OPTION, "value %d not allowed for option '%s'", value, name);
        *o.var = value;
        return SIGMA_OK;
    }
    return fail(SIGMA_ERR_BAD_OPTION, "unknown option '%s'", name);
}

int sigma_scan_get_option(const char* name) {
    if (!name) return -1;
    for (auto& o : g_opts) if (!std::strcmp(name, o.name)) return o.var->load();
    if (!std::strcmp(name, "rl_chain_timeouts")) {      // read-only: see sigma_scan.h
        unsigned int n = 0;
        if (sigma::bwdr_chain_timeouts_read(&n) != hipSuccess) return -1;
        return (int)(n > 0x7fffffffu ? 0x7fffffffu : n);
    }
    return -1;
}

int sigma_scan_fwd_plan(const sigma_scan_fwd_params* p, int32_t plan[6]) {
    int rc = check_fwd(p, false, false);
    if (rc) return rc;
    if (!plan) return fail(SIGMA_ERR_NULL_ARG, "plan is NULL");
    if (p->ckpt_pitch == SIGMA_SCAN_CKPT_PITCH_16) {
        const PlanR pr = plan_rowlane(p, true, false);
        if (!pr.ok) return fail(SIGMA_ERR_BAD_SHAPE, "ckpt_pitch 16 (row-lane kernels) is not available for this problem");
        // items 16, rows slot = state waves, tiles slot = segments, states_per_block slot = -200
        plan[0] = 16; plan[1] = pr.NW; plan[2] = pr.grid; plan[3] = (int32_t)sigma::fwdr_lds_bytes(pr.NW); plan[4] = pr.S; plan[5] = -200;
        return SIGMA_OK;
    }
    const PlanF4 p4 = plan_fwd4(p, true);
    if (p4.ok) {      // quad-row forward: items 10, rows slot = waves (4 rows each), states_per_block slot = -100
        plan[0] = 10; plan[1] = p4.W; plan[2] = p4.grid; plan[3] = (int32_t)p4.lds; plan[4] = 1; plan[5] = -100;
        return SIGMA_OK;
    }
    Plan pl = plan_fwd(p, true);
    plan[0] = pl.items; plan[1] = pl.rows; plan[2] = pl.grid; plan[3] = (int32_t)pl.lds; plan[4] = pl.tiles; plan[5] = pl.nb;
    return SIGMA_OK;
}

int sigma_scan_bwd_plan(const sigma_scan_bwd_params* p, int32_t plan[6]) {
    if (!p) return fail(SIGMA_ERR_NULL_ARG, "params is NULL");
    int rc = check_fwd(&p->fwd, false, false);
    if (rc) return rc;
    if (!plan) return fail(SIGMA_ERR_NULL_ARG, "plan is NULL");
    if (p->fwd.ckpt_pitch == SIGMA_SCAN_CKPT_PITCH_16) {
        const PlanR pr = plan_rowlane(&p->fwd, true, true);
        if (!pr.ok) return fail(SIGMA_ERR_BAD_SHAPE, "ckpt_pitch 16 (row-lane kernels) is not available for this problem");
        plan[0] = 16; plan[1] = pr.NW; plan[2] = pr.grid; plan[3] = (int32_t)sigma::bwdr_lds_bytes(4); plan[4] = pr.S; plan[5] = -200;
        return SIGMA_OK;
    }
    if (p->fwd.ckpt_pitch == SIGMA_SCAN_CKPT_PITCH_160) {
        const Plan4 p4 = plan_bwd4(&p->fwd, true);
        if (!p4.ok) return fail(SIGMA_ERR_BAD_SHAPE, "ckpt_pitch 160 (quad-row backward) is not available for this problem");
        // items = 10, rows slot = waves (4 rows each), states_per_block slot = -(100 + states per barrier)
        // items slot: 10 (+ 1000 x sequence segments when the sequence is split)
        plan[0] = 10 + (p4.S > 1 ? 1000 * p4.S : 0); plan[1] = p4.W; plan[2] = p4.grid; plan[3] = (int32_t)p4.lds; plan[4] = -p4.RB; plan[5] = -(100 + p4.SB);
        return SIGMA_OK;
    }
    const Plan3 p3 = plan_bwd3(&p->fwd, true);
    if (p3.ok) {       // items = 5, states_per_block slot = -(waves per row) marks the state-parallel kernel
        plan[0] = 5; plan[1] = p3.nw; plan[2] = p3.grid; plan[3] = (int32_t)p3.lds; plan[4] = -p3.RB; plan[5] = -(p->fwd.dstate / 4);
        return SIGMA_OK;
    }
    const Plan2 p2 = plan_bwd2(&p->fwd, true);
    if (p2.ok) {       // tiles_per_workgroup slot: -(row blocks per workgroup) marks the second-generation kernel
        plan[0] = p2.items; plan[1] = p2.rows; plan[2] = p2.grid; plan[3] = (int32_t)p2.lds; plan[4] = -p2.RB; plan[5] = p2.nb;
        return SIGMA_OK;
    }
    Plan pl = plan_bwd(&p->fwd, true);
    plan[0] = pl.items; plan[1] = pl.rows; plan[2] = pl.grid; plan[3] = (int32_t)pl.lds; plan[4] = pl.tiles; plan[5] = pl.nb;
    return SIGMA_OK;
}

int sigma_selective_scan_fwd(const sigma_scan_fwd_params* p, void* stream) {
    int rc = check_fwd(p, true);
    if (rc) return rc;
    if (p->batch == 0 || p->seqlen == 0) return SIGMA_OK;
    const bool vec = vec_ok_fwd(p, true) && (p->rev_group_mask == 0 || p->seqlen % 4 == 0);
    if (p->ckpt_pitch == SIGMA_SCAN_CKPT_PITCH_16) {
        const PlanR pr = plan_rowlane(p, vec, false);
        if (!pr.ok)
            return fail(SIGMA_ERR_BAD_SHAPE, "ckpt_pitch 16 needs f32 IO, 16-byte aligned operands, dstate in {4,8,16}, seqlen %% 4 == 0 and rows per group divisible by 64");
        const int64_t need = rowlane_summary_floats(p, pr.S) * (int64_t)sizeof(float);
        if (need > 0) {
            if (!p->workspace || p->workspace_bytes < need)
                return fail(SIGMA_ERR_NULL_ARG, "forward workspace of %lld bytes required (got %lld)", (long long)need, (long long)p->workspace_bytes);
            if (!aligned_to(p->workspace, 16)) return fail(SIGMA_ERR_BAD_STRIDE, "workspace must be 16-byte aligned");
        }
        sigma::FwdArgs a = make_fwd_args(p, pr.NW, 1, p->dstate, vec);
        a.rowblocks = pr.P; a.segs = pr.S; a.seg_tiles = pr.seg_tiles;
        a.fsumm = need > 0 ? static_cast<float*>(p->workspace) : nullptr;
        hipError_t e = sigma::launch_scan_fwdr(a, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return fail(SIGMA_ERR_LAUNCH, "scan_fwdr launch failed: %s", hipGetErrorString(e));
        return SIGMA_OK;
    }
    const PlanF4 p4 = plan_fwd4(p, vec);
    if (p4.ok) {
        sigma::FwdArgs a = make_fwd_args(p, p4.W, 1, p->dstate, vec);
        a.rowblocks = p4.P;
        hipError_t e = sigma::launch_scan_fwd4(a, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return fail(SIGMA_ERR_LAUNCH, "scan_fwd4 launch failed: %s", hipGetErrorString(e));
        return SIGMA_OK;
    }
    const Plan pl = plan_fwd(p, vec);
    if (pl.lds > kLdsLimit) return fail(SIGMA_ERR_BAD_SHAPE, "LDS budget exceeded (%zu B)", pl.lds);
    const sigma::FwdArgs a = make_fwd_args(p, pl.rows, pl.tiles, pl.nb, vec);
    hipError_t e = sigma::launch_scan_fwd(a, p->io_dtype, pl.items, pl.glds, g_opt_fwd_prefetch.load() != 2,
                                          static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(SIGMA_ERR_LAUNCH, "scan_fwd launch failed: %s", hipGetErrorString(e));
    return SIGMA_OK;
}

namespace {
bool vec_ok_bwd(const sigma_scan_bwd_params* q) {
    const sigma_scan_fwd_params* p = &q->fwd;
    const size_t al = 4 * (size_t)elem_size(p->io_dtype);
    return vec_ok_fwd(p, false) && aligned_to(q->dout, al) && aligned_to(q->du, al) && aligned_to(q->ddelta, al) &&
           q->dout_batch_stride % 4 == 0 && q->dout_d_stride % 4 == 0 && q->du_batch_stride % 4 == 0 &&
           q->du_d_stride % 4 == 0 && q->ddelta_batch_stride % 4 == 0 && q->ddelta_d_stride % 4 == 0 &&
           (p->rev_group_mask == 0 || p->seqlen % 4 == 0);
}
}  // namespace

int64_t sigma_scan_fwd_workspace_bytes(const sigma_scan_fwd_params* p) {
    if (!p) { fail(SIGMA_ERR_NULL_ARG, "params is NULL"); return -1; }
    if (check_fwd(p, false, false)) return -1;
    if (p->batch == 0 || p->seqlen == 0 || p->ckpt_pitch != SIGMA_SCAN_CKPT_PITCH_16) return 0;
    const PlanR pr = plan_rowlane(p, true, false);
    if (!pr.ok) { fail(SIGMA_ERR_BAD_SHAPE, "ckpt_pitch 16 (row-lane kernels) is not available for this problem"); return -1; }
    return rowlane_summary_floats(p, pr.S) * (int64_t)sizeof(float);
}

int64_t sigma_scan_bwd_workspace_bytes(const sigma_scan_bwd_params* q) {
    if (!q) { fail(SIGMA_ERR_NULL_ARG, "params is NULL"); return -1; }
    const sigma_scan_fwd_params* p = &q->fwd;
    if (check_fwd(p, false, false)) return -1;
    if (p->batch == 0 || p->seqlen == 0) return 0;
    if (p->ckpt_pitch == SIGMA_SCAN_CKPT_PITCH_16) {
        const PlanR pr = plan_rowlane(p, true, true);
        if (!pr.ok) { fail(SIGMA_ERR_BAD_SHAPE, "ckpt_pitch 16 (row-lane kernels) is not available for this problem"); return -1; }
        const int64_t slabs = pr.P <= 1 ? 0 : (int64_t)2 * pr.P * p->batch * p->n_groups * (int64_t)p->dstate * p->seqlen;
        return (slabs + rowlane_summary_floats(p, pr.S) + rowlane_chain_floats(p)) * (int64_t)sizeof(float);
    }
    if (p->ckpt_pitch == SIGMA_SCAN_CKPT_PITCH_160) {
        const Plan4 p4 = plan_bwd4(p, true);
        if (!p4.ok) { fail(SIGMA_ERR_BAD_SHAPE, "ckpt_pitch 160 (quad-row backward) is not available for this problem"); return -1; }
        const int64_t slabs = p4.P <= 1 ? 0 : (int64_t)2 * p4.P * p->batch * p->n_groups * (int64_t)p->dstate * p->seqlen * (int64_t)sizeof(float);
        const int64_t summ = p4.S <= 1 ? 0 : (int64_t)(p4.S - 1) * p->batch * p->dim * (int64_t)p->dstate * 2 * (int64_t)sizeof(float);
        return slabs + summ;
    }
    const Plan3 p3 = plan_bwd3(p, true);     // no plan's workgroup count depends on alignment
    if (p3.ok)
        return p3.P <= 1 ? 0 : (int64_t)2 * p3.P * p->batch * p->n_groups * (int64_t)p->dstate * p->seqlen * (int64_t)sizeof(float);
    const Plan2 p2 = plan_bwd2(p, true);
    if (p2.ok)
        return p2.P <= 1 ? 0 : (int64_t)2 * p2.P * p->batch * p->n_groups * (int64_t)p->dstate * p->seqlen * (int64_t)sizeof(float);
    const Plan pl = plan_bwd(p, true);
    const int P = (p->dim / p->n_groups) / pl.rows;
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
    if (p->seqlen > (p->ckpt_pitch ? p->ckpt_pitch : SIGMA_SCAN_CKPT_PITCH) && !p->x)
        return fail(SIGMA_ERR_NULL_ARG, "x (forward checkpoints) is required when seqlen > the checkpoint pitch");
    if ((p->D == nullptr) != (q->dD == nullptr) || (p->delta_bias == nullptr) != (q->ddelta_bias == nullptr))
        return fail(SIGMA_ERR_NULL_ARG, "dD / ddelta_bias must be given exactly when D / delta_bias are");
    const bool vec = vec_ok_bwd(q);
    if (q->dout_group_shift < 0 || q->dout_group_shift > 5)
        return fail(SIGMA_ERR_BAD_SHAPE, "dout_group_shift must be in [0, 5] (got %d)", q->dout_group_shift);
    if (p->ckpt_pitch == SIGMA_SCAN_CKPT_PITCH_16) {
        const PlanR pr = plan_rowlane(p, vec, true);
        if (!pr.ok)
            return fail(SIGMA_ERR_BAD_SHAPE, "ckpt_pitch 16 needs f32 IO, 16-byte aligned operands, dstate in {4,8,16}, seqlen %% 4 == 0 and rows per group divisible by 64");
        if (p->seqlen > SIGMA_SCAN_CKPT_PITCH_16 && !p->x)
            return fail(SIGMA_ERR_NULL_ARG, "x (forward checkpoints) is required when seqlen > the checkpoint pitch");
        const int64_t slab = pr.P > 1 ? (int64_t)pr.P * p->batch * p->n_groups * (int64_t)p->dstate * p->seqlen : 0;
        const int64_t summ = rowlane_summary_floats(p, pr.S);
        const int64_t need = (2 * slab + summ + rowlane_chain_floats(p)) * (int64_t)sizeof(float);
        if (need > 0) {
            if (!q->workspace || q->workspace_bytes < need)
                return fail(SIGMA_ERR_NULL_ARG, "workspace of %lld bytes required (got %lld)", (long long)need, (long long)q->workspace_bytes);
            if (!aligned_to(q->workspace, 16)) return fail(SIGMA_ERR_BAD_STRIDE, "workspace must be 16-byte aligned");
        }
        sigma::BwdArgs a;
        std::memset(&a, 0, sizeof(a));
        a.f = make_fwd_args(p, pr.NW, 1, p->dstate, vec);
        a.dout = q->dout; a.du = q->du; a.ddelta = q->ddelta;
        a.dA = q->dA; a.dB = q->dB; a.dC = q->dC; a.dD = q->dD; a.dbias = q->ddelta_bias;
        a.g_bs = q->dout_batch_stride; a.g_ds = q->dout_d_stride;
        a.du_bs = q->du_batch_stride; a.du_ds = q->du_d_stride;
        a.dd_bs = q->ddelta_batch_stride; a.dd_ds = q->ddelta_d_stride;
        a.dA_ds = q->dA_d_stride; a.dA_ns = q->dA_dstate_stride;
        a.dB_bs = q->dB_batch_stride; a.dB_gs = q->dB_group_stride; a.dB_ns = q->dB_dstate_stride;
        a.dC_bs = q->dC_batch_stride; a.dC_gs = q->dC_group_stride; a.dC_ns = q->dC_dstate_stride;
        a.P = pr.P; a.S = pr.S; a.seg_tiles = pr.seg_tiles;
        a.g_gshift = q->dout_group_shift;
        a.out_vec_ok = (aligned_to(q->dB, 16) && aligned_to(q->dC, 16) && q->dB_batch_stride % 4 == 0 &&
                        q->dB_group_stride % 4 == 0 && q->dB_dstate_stride % 4 == 0 && q->dC_batch_stride % 4 == 0 &&
                        q->dC_group_stride % 4 == 0 && q->dC_dstate_stride % 4 == 0) ? 1 : 0;
        a.ws_dB = pr.P > 1 ? static_cast<float*>(q->workspace) : nullptr;
        a.ws_dC = pr.P > 1 ? static_cast<float*>(q->workspace) + slab : nullptr;
        a.summ = summ > 0 ? static_cast<float*>(q->workspace) + 2 * slab : nullptr;
        a.chain_W = rowlane_chain_tiles(p, pr);
        a.chain_carry = static_cast<float*>(q->workspace) + 2 * slab + summ;
        a.chain_flag = reinterpret_cast<int*>(a.chain_carry + (int64_t)p->batch * (p->dim / 64) * p->dstate * 64);
        hipError_t e = sigma::launch_scan_bwdr(a, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return fail(SIGMA_ERR_LAUNCH, "scan_bwdr launch failed: %s", hipGetErrorString(e));
        return SIGMA_OK;
    }
    Plan4 p4;
    std::memset(&p4, 0, sizeof(p4));
    if (p->ckpt_pitch == SIGMA_SCAN_CKPT_PITCH_160) {
        // B/C alignment is part of the plan: the caller chose the pitch at forward time with the same tensors
        p4 = plan_bwd4(p, vec_ok_fwd(p, false));
        if (!p4.ok)
            return fail(SIGMA_ERR_BAD_SHAPE, "ckpt_pitch 160 needs f32 IO, 16-byte aligned B/C, dstate in {4,8,16} and rows per group divisible by 4");
    }
    const Plan3 p3 = p4.ok ? Plan3{} : plan_bwd3(p, vec);
    Plan2 p2;
    std::memset(&p2, 0, sizeof(p2));
    if (!p3.ok && !p4.ok) p2 = plan_bwd2(p, vec);
    Plan pl;
    if (p4.ok) { pl.items = 10; pl.rows = p4.W; pl.tiles = 1; pl.nb = p->dstate; pl.grid = p4.grid; pl.glds = true; pl.lds = p4.lds; pl.slab2 = false; }
    else if (p3.ok) { pl.items = 5; pl.rows = p3.nw; pl.tiles = 1; pl.nb = p->dstate; pl.grid = p3.grid; pl.glds = p3.glds; pl.lds = p3.lds; pl.slab2 = false; }
    else if (p2.ok) { pl.items = p2.items; pl.rows = p2.rows; pl.tiles = 1; pl.nb = p2.nb; pl.grid = p2.grid; pl.glds = p2.glds; pl.lds = p2.lds; pl.slab2 = p2.slab2; }
    else pl = plan_bwd(p, vec);
    if (!p2.ok && !p3.ok && p->ckpt_pitch == SIGMA_SCAN_CKPT_PITCH_320 && g_opt_bwd_gen.load() == 1)
        return fail(SIGMA_ERR_BAD_OPTION, "ckpt_pitch 320 needs the second-generation backward (option bwd_gen != 1)");
    if (pl.lds > kLdsLimit) return fail(SIGMA_ERR_BAD_SHAPE, "LDS budget exceeded (%zu B)", pl.lds);
    const int P = p4.ok ? p4.P : p3.ok ? p3.P : (p2.ok ? p2.P : (p->dim / p->n_groups) / pl.rows);
    const int64_t slab = P > 1 ? (int64_t)P * p->batch * p->n_groups * (int64_t)p->dstate * p->seqlen : 0;
    const int S = p4.ok ? p4.S : 1;
    const int64_t summ_floats = S > 1 ? (int64_t)(S - 1) * p->batch * p->dim * (int64_t)p->dstate * 2 : 0;
    if (P > 1 || S > 1) {
        const int64_t need = (2 * slab + summ_floats) * (int64_t)sizeof(float);
        if (!q->workspace || q->workspace_bytes < need)
            return fail(SIGMA_ERR_NULL_ARG, "workspace of %lld bytes required (got %lld)", (long long)need, (long long)q->workspace_bytes);
        if (!aligned_to(q->workspace, 16)) return fail(SIGMA_ERR_BAD_STRIDE, "workspace must be 16-byte aligned");
    }
    sigma::BwdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.f = make_fwd_args(p, pl.rows, 1, pl.nb, vec);
    a.dout = q->dout; a.du = q->du; a.ddelta = q->ddelta;
    a.dA = q->dA; a.dB = q->dB; a.dC = q->dC; a.dD = q->dD; a.dbias = q->ddelta_bias;
    a.g_bs = q->dout_batch_stride; a.g_ds = q->dout_d_stride;
    a.du_bs = q->du_batch_stride; a.du_ds = q->du_d_stride;
    a.dd_bs = q->ddelta_batch_stride; a.dd_ds = q->ddelta_d_stride;
    a.dA_ds = q->dA_d_stride; a.dA_ns = q->dA_dstate_stride;
    a.dB_bs = q->dB_batch_stride; a.dB_gs = q->dB_group_stride; a.dB_ns = q->dB_dstate_stride;
    a.dC_bs = q->dC_batch_stride; a.dC_gs = q->dC_group_stride; a.dC_ns = q->dC_dstate_stride;
    if (q->dout_group_shift < 0 || q->dout_group_shift > 5)
        return fail(SIGMA_ERR_BAD_SHAPE, "dout_group_shift must be in [0, 5] (got %d)", q->dout_group_shift);
    a.P = P;
    a.slab2 = pl.slab2 ? 1 : 0;
    a.g_gshift = q->dout_group_shift;
    a.out_vec_ok = (aligned_to(q->dB, 16) && aligned_to(q->dC, 16) && q->dB_batch_stride % 4 == 0 &&
                    q->dB_group_stride % 4 == 0 && q->dB_dstate_stride % 4 == 0 && q->dC_batch_stride % 4 == 0 &&
                    q->dC_group_stride % 4 == 0 && q->dC_dstate_stride % 4 == 0) ? 1 : 0;
    a.ws_dB = P > 1 ? static_cast<float*>(q->workspace) : nullptr;
    a.ws_dC = P > 1 ? static_cast<float*>(q->workspace) + slab : nullptr;
    {
        const int t = g_opt_bwd_touch.load();        // bit 0 of flags = NO touches
        a.flags = (t == 1 || (t == 0 && p4.ok)) ? 0 : 1;
    }
    a.RB = p4.ok ? p4.RB : p3.ok ? p3.RB : (p2.ok ? p2.RB : 1);
    a.S = 1;
    if (p4.ok) {
        a.slab2 = p4.SB; a.f.NB = p4.nbuf; if (p4.wgs == 2) a.flags |= 2;
        a.S = p4.S; a.seg_tiles = p4.seg_tiles;
        a.summ = S > 1 ? static_cast<float*>(q->workspace) + 2 * slab : nullptr;
    }
    hipError_t e = p4.ok ? sigma::launch_scan_bwd4(a, static_cast<hipStream_t>(stream)) : p3.ok ? sigma::launch_scan_bwd3(a, p->io_dtype, pl.glds, static_cast<hipStream_t>(stream)) : p2.ok ? sigma::launch_scan_bwd2(a, p->io_dtype, pl.items, pl.glds, static_cast<hipStream_t>(stream))
                         : sigma::launch_scan_bwd(a, p->io_dtype, pl.items, pl.glds, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(SIGMA_ERR_LAUNCH, "scan_bwd launch failed: %s", hipGetErrorString(e));
    return SIGMA_OK;
}

int sigma_scan_debug_read(uint64_t out16[16]) {
    if (!out16) return fail(SIGMA_ERR_NULL_ARG, "out16 is NULL");
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = sigma::bwd2_prof_read(reinterpret_cast<unsigned long long*>(out16));
    if (e == hipSuccess && out16[15] == 0) e = sigma::bwd4_prof_read(reinterpret_cast<unsigned long long*>(out16));   // quad-row kernel ran
    if (e == hipSuccess && out16[15] == 0) e = sigma::bwdr_prof_read(reinterpret_cast<unsigned long long*>(out16));   // row-lane backward ran
    if (e == hipSuccess && out16[15] == 0) e = sigma::fwdr_prof_read(reinterpret_cast<unsigned long long*>(out16));   // row-lane forward ran
    if (e == hipSuccess && out16[15] == 0) e = sigma::gemm_prof_read(reinterpret_cast<unsigned long long*>(out16));   // split-operand GEMM ran
    if (e != hipSuccess) return fail(SIGMA_ERR_LAUNCH, "debug read failed: %s", hipGetErrorString(e));
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
                    "wave-scan selftest mismatch: fwd=%g rev=%g prev=%g next=%g sum=%g mfwd=%g mrev=%g (max abs errors)",
                    h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    return SIGMA_OK;
}

// Self test of the row-lane kernels (scan_fwdr.hip / scan_bwdr.hip), in the spirit of sigma_gemm_selftest: their vector
// memory waits are COUNTED by hand (LDS-DMA requests retired with s_waitcnt vmcnt(3 + 2 NS) while younger stores stay
// in flight), their scalar operand waits pinned with scheduling barriers -- a toolchain that schedules differently must
// fail HERE, loudly, not in a training run.  One small problem through the public entry points (two groups of 64 rows,
// the second walked backwards, 16 states, 9 full tiles + a partial one, softplus, D and bias): forward and all seven
// gradients against a host recurrence in double precision.  Synchronises `stream`.  0 = pass.
int sigma_scan_rowlane_selftest(void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int Bt = 1, G = 2, RPG = 64, KD = G * RPG, L = 148, N = 16, NT = (L + 15) / 16;
    const size_t nrow = (size_t)KD * L, nbc = (size_t)G * N * L, nx = (size_t)KD * NT * N;
    std::vector<float> u(nrow), dl(nrow), g(nrow), A((size_t)KD * N), Bm(nbc), Cm(nbc), D(KD), bias(KD);
    unsigned st = 2463534242u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)((st >> 8) & 0xffff) / 32768.0f - 1.0f; };   // [-1, 1)
    for (auto& v : u) v = rnd();
    for (auto& v : dl) v = 0.5f * rnd();
    for (auto& v : g) v = rnd();
    for (auto& v : Bm) v = rnd();
    for (auto& v : Cm) v = rnd();
    for (int r = 0; r < KD; ++r) {
        D[r] = rnd(); bias[r] = -1.0f + 0.5f * rnd();
        for (int n = 0; n < N; ++n) A[(size_t)r * N + n] = -0.1f * (float)(n + 1) * (1.0f + 0.1f * (float)(r % 3));
    }
    const unsigned rev_mask = 0b10;
    // ---- host reference (double)
    std::vector<double> r_out(nrow), r_du(nrow), r_dd(nrow), r_dA((size_t)KD * N, 0.0), r_dB(nbc, 0.0), r_dC(nbc, 0.0), r_dD(KD, 0.0), r_db(KD, 0.0);
    {
        std::vector<double> dsp(L), sg(L), xs((size_t)L * N), a((size_t)L * N);
        for (int r = 0; r < KD; ++r) {
            const int gi = r / RPG;
            const bool rev = (rev_mask >> gi) & 1u;
            auto pos = [&](int t) { return rev ? L - 1 - t : t; };          // memory position of scan step t
            for (int l = 0; l < L; ++l) {
                const double raw = (double)dl[(size_t)r * L + l] + bias[r];
                dsp[l] = raw > 20.0 ? raw : std::log1p(std::exp(raw));
                sg[l] = raw > 20.0 ? 1.0 : 1.0 / (1.0 + std::exp(-raw));
            }
            std::vector<double> x(N, 0.0);
            for (int t = 0; t < L; ++t) {
                const int l = pos(t);
                double y = (double)D[r] * u[(size_t)r * L + l];
                for (int n = 0; n < N; ++n) {
                    const double an = std::exp(dsp[l] * A[(size_t)r * N + n]);
                    x[n] = an * x[n] + dsp[l] * u[(size_t)r * L + l] * Bm[((size_t)gi * N + n) * L + l];
                    a[(size_t)t * N + n] = an; xs[(size_t)t * N + n] = x[n];
                    y += (double)Cm[((size_t)gi * N + n) * L + l] * x[n];
                }
                r_out[(size_t)r * L + l] = y;
            }
            std::vector<double> e(N, 0.0);
            for (int t = L - 1; t >= 0; --t) {
                const int l = pos(t);
                const double gl = g[(size_t)r * L + l], ul = u[(size_t)r * L + l];
                double s1 = 0.0, s2 = 0.0;
                for (int n = 0; n < N; ++n) {
                    const double dx = gl * Cm[((size_t)gi * N + n) * L + l] + e[n];
                    const double xprev = t > 0 ? xs[(size_t)(t - 1) * N + n] : 0.0;
                    const double an = a[(size_t)t * N + n], bn = Bm[((size_t)gi * N + n) * L + l];
                    e[n] = an * dx;
                    s1 += dx * bn;
                    s2 += dx * an * xprev * A[(size_t)r * N + n];
                    r_dA[(size_t)r * N + n] += dx * an * xprev * dsp[l];
                    r_dB[((size_t)gi * N + n) * L + l] += dx * dsp[l] * ul;
                    r_dC[((size_t)gi * N + n) * L + l] += gl * xs[(size_t)t * N + n];
                }
                r_du[(size_t)r * L + l] = (double)D[r] * gl + dsp[l] * s1;
                const double dd = (ul * s1 + s2) * sg[l];
                r_dd[(size_t)r * L + l] = dd;
                r_dD[r] += gl * ul;
                r_db[r] += dd;
            }
        }
    }
    // ---- device
    sigma_scan_bwd_params q;
    std::memset(&q, 0, sizeof(q));
    sigma_scan_fwd_params& p = q.fwd;
    p.batch = Bt; p.dim = KD; p.seqlen = L; p.dstate = N; p.n_groups = G; p.n_chunks = (L + SIGMA_SCAN_CHUNK - 1) / SIGMA_SCAN_CHUNK;
    p.io_dtype = SIGMA_DTYPE_F32; p.delta_softplus = 1; p.rev_group_mask = rev_mask; p.ckpt_pitch = SIGMA_SCAN_CKPT_PITCH_16;
    p.u_batch_stride = p.delta_batch_stride = p.out_batch_stride = (int64_t)nrow; p.u_d_stride = p.delta_d_stride = p.out_d_stride = L;
    p.A_d_stride = N; p.A_dstate_stride = 1;
    p.B_batch_stride = p.C_batch_stride = (int64_t)nbc; p.B_group_stride = p.C_group_stride = (int64_t)N * L; p.B_dstate_stride = p.C_dstate_stride = L;
    q.dout_batch_stride = q.du_batch_stride = q.ddelta_batch_stride = (int64_t)nrow; q.dout_d_stride = q.du_d_stride = q.ddelta_d_stride = L;
    q.dA_d_stride = N; q.dA_dstate_stride = 1;
    q.dB_batch_stride = q.dC_batch_stride = (int64_t)nbc; q.dB_group_stride = q.dC_group_stride = (int64_t)N * L; q.dB_dstate_stride = q.dC_dstate_stride = L;
    // one allocation, every tensor 256-byte aligned
    auto al = [](size_t n) { return (n + 63) / 64 * 64; };
    const size_t o_u = 0, o_dl = o_u + al(nrow), o_g = o_dl + al(nrow), o_A = o_g + al(nrow), o_B = o_A + al((size_t)KD * N), o_C = o_B + al(nbc),
                 o_D = o_C + al(nbc), o_bias = o_D + al(KD), o_out = o_bias + al(KD), o_x = o_out + al(nrow), o_du = o_x + al(nx),
                 o_dd = o_du + al(nrow), o_dA = o_dd + al(nrow), o_dB = o_dA + al((size_t)KD * N), o_dC = o_dB + al(nbc), o_dD = o_dC + al(nbc),
                 o_db = o_dD + al(KD), o_end = o_db + al(KD);
    float* dev = nullptr;
    hipError_t e = hipMalloc(&dev, o_end * sizeof(float));
    if (e != hipSuccess) return fail(SIGMA_ERR_NO_DEVICE, "row-lane selftest: hipMalloc failed: %s", hipGetErrorString(e));
    void* ws_f = nullptr; void* ws_b = nullptr;
    int rc = SIGMA_OK;
    std::vector<float> h_out(nrow), h_du(nrow), h_dd(nrow), h_dA((size_t)KD * N), h_dB(nbc), h_dC(nbc), h_dD(KD), h_db(KD);
    auto up = [&](size_t off, const std::vector<float>& v) { return hipMemcpyAsync(dev + off, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice, s); };
    auto down = [&](std::vector<float>& v, size_t off) { return hipMemcpyAsync(v.data(), dev + off, v.size() * sizeof(float), hipMemcpyDeviceToHost, s); };
    do {
        e = hipMemsetAsync(dev, 0, o_end * sizeof(float), s);
        if (e == hipSuccess) e = up(o_u, u);
        if (e == hipSuccess) e = up(o_dl, dl);
        if (e == hipSuccess) e = up(o_g, g);
        if (e == hipSuccess) e = up(o_A, A);
        if (e == hipSuccess) e = up(o_B, Bm);
        if (e == hipSuccess) e = up(o_C, Cm);
        if (e == hipSuccess) e = up(o_D, D);
        if (e == hipSuccess) e = up(o_bias, bias);
        if (e != hipSuccess) break;
        p.u = dev + o_u; p.delta = dev + o_dl; p.A = dev + o_A; p.B = dev + o_B; p.C = dev + o_C; p.D = dev + o_D; p.delta_bias = dev + o_bias;
        p.out = dev + o_out; p.x = dev + o_x;
        q.dout = dev + o_g; q.du = dev + o_du; q.ddelta = dev + o_dd; q.dA = dev + o_dA; q.dB = dev + o_dB; q.dC = dev + o_dC;
        q.dD = dev + o_dD; q.ddelta_bias = dev + o_db;
        const int64_t wf = sigma_scan_fwd_workspace_bytes(&p), wb = sigma_scan_bwd_workspace_bytes(&q);
        if (wf < 0 || wb < 0) { rc = SIGMA_ERR_BAD_SHAPE; break; }
        if (wf > 0) { e = hipMalloc(&ws_f, (size_t)wf); if (e != hipSuccess) break; p.workspace = ws_f; p.workspace_bytes = wf; }
        rc = sigma_selective_scan_fwd(&p, stream);
        if (rc != SIGMA_OK) break;
        if (wb > 0) { e = hipMalloc(&ws_b, (size_t)wb); if (e != hipSuccess) break; q.workspace = ws_b; q.workspace_bytes = wb; }
        rc = sigma_selective_scan_bwd(&q, stream);
        if (rc != SIGMA_OK) break;
        e = down(h_out, o_out);
        if (e == hipSuccess) e = down(h_du, o_du);
        if (e == hipSuccess) e = down(h_dd, o_dd);
        if (e == hipSuccess) e = down(h_dA, o_dA);
        if (e == hipSuccess) e = down(h_dB, o_dB);
        if (e == hipSuccess) e = down(h_dC, o_dC);
        if (e == hipSuccess) e = down(h_dD, o_dD);
        if (e == hipSuccess) e = down(h_db, o_db);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    } while (false);
    (void)hipStreamSynchronize(s);
    (void)hipFree(dev); (void)hipFree(ws_f); (void)hipFree(ws_b);
    if (e != hipSuccess) return fail(SIGMA_ERR_LAUNCH, "row-lane selftest failed to run: %s", hipGetErrorString(e));
    if (rc != SIGMA_OK) return rc;                                    // sigma_scan_last_error() holds the entry point's message
    auto worst = [](const std::vector<float>& got, const std::vector<double>& want) {
        double mx = 0.0, err = 0.0;
        for (size_t i = 0; i < got.size(); ++i) {
            mx = std::fmax(mx, std::fabs(want[i]));
            const double d = std::fabs((double)got[i] - want[i]);
            err = (d == d) ? std::fmax(err, d) : 1e30;               // NaN counts as a failure
        }
        return err / (mx + 1.0);
    };
    const double errs[8] = {worst(h_out, r_out), worst(h_du, r_du), worst(h_dd, r_dd), worst(h_dA, r_dA),
                            worst(h_dB, r_dB), worst(h_dC, r_dC), worst(h_dD, r_dD), worst(h_db, r_db)};
    for (int i = 0; i < 8; ++i)
        if (!(errs[i] < 2e-4))
            return fail(SIGMA_ERR_LAUNCH, "row-lane selftest mismatch (scaled max errors): out=%g du=%g ddelta=%g dA=%g dB=%g dC=%g dD=%g dbias=%g",
                        errs[0], errs[1], errs[2], errs[3], errs[4], errs[5], errs[6], errs[7]);
    return SIGMA_OK;
}

}  // extern "C"
