/*
 * scan_oracle.c -- CPU restatement of Sigma's selective-scan operator.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sigma_amd/ may import, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it (as the checker / the timed CPU baseline, never as the product).
 *
 * What it restates (paths relative to /root/reference):
 *   forward : models/encoders/selective_scan/selective_scan/selective_scan_interface.py:86-131
 *             (selective_scan_ref: fp32 upcast, delta += bias, softplus, exp(delta*A),
 *              grouped B/C, sequential x = dA*x + dB*u, y = <x, C>, out = y + u*D)
 *   backward: the analytic adjoint of that function, identical to what the CUDA
 *             kernel computes (csrc/selective_scan/selective_scan_bwd_kernel.cuh:141-273):
 *             dx[n,l]  = g_l*C[n,l] + a[n,l+1]*dx[n,l+1]
 *             du       = D*g + sum_n dx*B*delta
 *             ddelta   = sum_n dx*(B*u + A*(x - b)),  then softplus' and bias
 *             dA, dB, dC, dD, ddelta_bias reductions as in SURVEY.md App. E.2.
 *
 * Parity pinning: tests/test_oracle_scan.py checks both entry points against
 * golden vectors produced in the build container by the reference's own
 * selective_scan_ref + torch autograd (tests/golden/make_golden_scan.py).
 *
 * Two arithmetic modes:
 *   acc64 = 0 : float state/accumulators, same order of operations as the
 *               reference's fp32 torch code (closest to "what the reference prints")
 *   acc64 = 1 : double everywhere (inputs are still the fp32 values) -- the
 *               "truth" used when judging the HIP kernels' rounding.
 *
 * Layouts (all contiguous, float32):
 *   u, delta, out, dout, du, ddelta : (B, D, L)
 *   A, dA                           : (D, N)
 *   Bm, Cm, dB, dC                  : (B, G, N, L)   row d uses group d / (D/G)
 *   Dv, delta_bias, dD, ddelta_bias : (D) or NULL
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static inline double softplus_d(double x) {
    /* torch F.softplus, beta=1, threshold=20 (selective_scan_interface.py:105-106) */
    return x > 20.0 ? x : log1p(exp(x));
}
static inline float softplus_f(float x) {
    return x > 20.0f ? x : log1pf(expf(x));
}

/* ---------------------------------------------------------------- forward */
int scan_oracle_fwd(const float *u, const float *delta, const float *A,
                    const float *Bm, const float *Cm, const float *Dv,
                    const float *delta_bias, int delta_softplus,
                    int B, int D, int L, int N, int G, int acc64,
                    float *out, float *last_state /* (B,D,N) or NULL */)
{
    if (B < 0 || D <= 0 || L < 0 || N <= 0 || G <= 0 || D % G != 0) return 1;
    const int rows_per_group = D / G;
    const long rows = (long)B * D;
#pragma omp parallel for schedule(dynamic, 4)
    for (long br = 0; br < rows; ++br) {
        const int b = (int)(br / D), d = (int)(br % D);
        const int g = d / rows_per_group;
        const float *ur = u + br * (long)L;
        const float *dr = delta + br * (long)L;
        const float *Bg = Bm + ((long)b * G + g) * (long)N * L;
        const float *Cg = Cm + ((long)b * G + g) * (long)N * L;
        float *orow = out + br * (long)L;
        const float bias = delta_bias ? delta_bias[d] : 0.0f;
        const float Dd = Dv ? Dv[d] : 0.0f;
        if (acc64) {
            double x[256];
            for (int n = 0; n < N; ++n) x[n] = 0.0;
            for (int l = 0; l < L; ++l) {
                double dl = (double)dr[l] + (double)bias;
                if (delta_softplus) dl = softplus_d(dl);
                const double uu = ur[l];
                double y = 0.0;
                for (int n = 0; n < N; ++n) {
                    const double a = exp(dl * (double)A[(long)d * N + n]);
                    x[n] = a * x[n] + dl * (double)Bg[(long)n * L + l] * uu;
                    y += x[n] * (double)Cg[(long)n * L + l];
                }
                orow[l] = (float)(y + uu * (double)Dd);
            }
            if (last_state) for (int n = 0; n < N; ++n) last_state[br * N + n] = (float)x[n];
        } else {
            float x[256];
            for (int n = 0; n < N; ++n) x[n] = 0.0f;
            for (int l = 0; l < L; ++l) {
                float dl = delta_bias ? dr[l] + bias : dr[l];
                if (delta_softplus) dl = softplus_f(dl);
                const float uu = ur[l];
                float y = 0.0f;
                for (int n = 0; n < N; ++n) {
                    const float a = expf(dl * A[(long)d * N + n]);
                    /* einsum('bdl,bdnl,bdl->bdln'): (delta*B)*u */
                    x[n] = a * x[n] + (dl * Bg[(long)n * L + l]) * uu;
                    y += x[n] * Cg[(long)n * L + l];
                }
                orow[l] = Dv ? y + uu * Dd : y;
            }
            if (last_state) for (int n = 0; n < N; ++n) last_state[br * N + n] = x[n];
        }
    }
    return 0;
}

/* --------------------------------------------------------------- backward */
/* All accumulation in double when acc64, else float for the per-(n,l) terms and
 * double only for the long reductions (dA, dD, dbias) -- the reference's autograd
 * reduces those with torch.sum in fp32 pairwise order, which double brackets. */
int scan_oracle_bwd(const float *u, const float *delta, const float *A,
                    const float *Bm, const float *Cm, const float *Dv,
                    const float *delta_bias, const float *dout, int delta_softplus,
                    int B, int D, int L, int N, int G, int acc64,
                    float *du, float *ddelta, float *dA, float *dB, float *dC,
                    float *dD /* or NULL */, float *ddelta_bias /* or NULL */)
{
    (void)acc64; /* backward always runs in double: it is the checker's truth */
    if (B < 0 || D <= 0 || L < 0 || N <= 0 || G <= 0 || D % G != 0) return 1;
    const int rows_per_group = D / G;
    const long BG = (long)B * G;
    double *dA_acc = (double *)calloc((size_t)D * N, sizeof(double));
    double *dD_acc = (double *)calloc((size_t)D, sizeof(double));
    double *db_acc = (double *)calloc((size_t)D, sizeof(double));
    if (!dA_acc || !dD_acc || !db_acc) { free(dA_acc); free(dD_acc); free(db_acc); return 2; }
    int fail = 0;
    /* one task per (batch, group): dB/dC of a group are private to the task */
#pragma omp parallel
    {
        double *xs = (double *)malloc((size_t)N * (L > 0 ? L : 1) * sizeof(double));
        double *as = (double *)malloc((size_t)N * (L > 0 ? L : 1) * sizeof(double));
        double *dls = (double *)malloc((size_t)(L > 0 ? L : 1) * sizeof(double));
        double *dBg = (double *)malloc((size_t)N * (L > 0 ? L : 1) * sizeof(double));
        double *dCg = (double *)malloc((size_t)N * (L > 0 ? L : 1) * sizeof(double));
        double *dAl = (double *)calloc((size_t)D * N, sizeof(double));
        double *dDl = (double *)calloc((size_t)D, sizeof(double));
        double *dbl = (double *)calloc((size_t)D, sizeof(double));
        if (!xs || !as || !dls || !dBg || !dCg || !dAl || !dDl || !dbl) {
#pragma omp atomic write
            fail = 1;
        }
#pragma omp for schedule(dynamic, 1)
        for (long bg = 0; bg < BG; ++bg) {
            if (fail) continue;
            const int b = (int)(bg / G), g = (int)(bg % G);
            const float *Bg = Bm + bg * (long)N * L;
            const float *Cg = Cm + bg * (long)N * L;
            for (long i = 0; i < (long)N * L; ++i) { dBg[i] = 0.0; dCg[i] = 0.0; }
            for (int dd = 0; dd < rows_per_group; ++dd) {
                const int d = g * rows_per_group + dd;
                const long br = (long)b * D + d;
                const float *ur = u + br * (long)L;
                const float *dr = delta + br * (long)L;
                const float *gr = dout + br * (long)L;
                const double bias = delta_bias ? (double)delta_bias[d] : 0.0;
                const double Dd = Dv ? (double)Dv[d] : 0.0;
                /* forward recompute, keeping a and x */
                for (int n = 0; n < N; ++n) {
                    double x = 0.0;
                    const double An = A[(long)d * N + n];
                    for (int l = 0; l < L; ++l) {
                        if (n == 0) {
                            double dl = (double)dr[l] + bias;
                            if (delta_softplus) dl = softplus_d(dl);
                            dls[l] = dl;
                        }
                        const double a = exp(dls[l] * An);
                        x = a * x + dls[l] * (double)Bg[(long)n * L + l] * (double)ur[l];
                        as[(long)n * L + l] = a;
                        xs[(long)n * L + l] = x;
                    }
                }
                /* reverse sweep */
                for (int l = 0; l < L; ++l) { du[br * (long)L + l] = 0.0f; ddelta[br * (long)L + l] = 0.0f; }
                double dDsum = 0.0, dbsum = 0.0;
                /* per-l accumulators over n need a second array; reuse small stack via two passes */
                double *du_acc = (double *)calloc((size_t)(L > 0 ? L : 1), sizeof(double));
                double *dd_acc = (double *)calloc((size_t)(L > 0 ? L : 1), sizeof(double));
                if (!du_acc || !dd_acc) { free(du_acc); free(dd_acc); fail = 1; continue; }
                for (int n = 0; n < N; ++n) {
                    const double An = A[(long)d * N + n];
                    double dx = 0.0, dAsum = 0.0;
                    for (int l = L - 1; l >= 0; --l) {
                        const double gl = gr[l];
                        const double a_next = (l + 1 < L) ? as[(long)n * L + l + 1] : 1.0;
                        dx = gl * (double)Cg[(long)n * L + l] + a_next * dx;
                        const double x = xs[(long)n * L + l];
                        const double bterm = dls[l] * (double)Bg[(long)n * L + l] * (double)ur[l];
                        const double ax_prev = x - bterm; /* = a[l]*x[l-1] */
                        du_acc[l] += dx * (double)Bg[(long)n * L + l] * dls[l];
                        dd_acc[l] += dx * ((double)Bg[(long)n * L + l] * (double)ur[l] + An * ax_prev);
                        dAsum += dx * dls[l] * ax_prev;
                        dBg[(long)n * L + l] += dx * dls[l] * (double)ur[l];
                        dCg[(long)n * L + l] += gl * x;
                    }
                    dAl[(long)d * N + n] += dAsum;
                }
                for (int l = 0; l < L; ++l) {
                    const double gl = gr[l];
                    du[br * (long)L + l] = (float)(du_acc[l] + Dd * gl);
                    dDsum += gl * (double)ur[l];
                    double dd = dd_acc[l];
                    if (delta_softplus) {
                        const double raw = (double)dr[l] + bias;
                        /* d softplus / d raw = sigmoid(raw) below the threshold, 1 above */
                        if (raw <= 20.0) dd *= 1.0 / (1.0 + exp(-raw));
                    }
                    ddelta[br * (long)L + l] = (float)dd;
                    dbsum += dd;
                }
                free(du_acc); free(dd_acc);
                dDl[d] += dDsum;
                dbl[d] += dbsum;
            }
            for (long i = 0; i < (long)N * L; ++i) {
                dB[bg * (long)N * L + i] = (float)dBg[i];
                dC[bg * (long)N * L + i] = (float)dCg[i];
            }
        }
#pragma omp critical
        {
            if (!fail) {
                for (long i = 0; i < (long)D * N; ++i) dA_acc[i] += dAl[i];
                for (int d = 0; d < D; ++d) { dD_acc[d] += dDl[d]; db_acc[d] += dbl[d]; }
            }
        }
        free(xs); free(as); free(dls); free(dBg); free(dCg); free(dAl); free(dDl); free(dbl);
    }
    if (!fail) {
        for (long i = 0; i < (long)D * N; ++i) dA[i] = (float)dA_acc[i];
        if (dD) for (int d = 0; d < D; ++d) dD[d] = (float)dD_acc[d];
        if (ddelta_bias) for (int d = 0; d < D; ++d) ddelta_bias[d] = (float)db_acc[d];
    }
    free(dA_acc); free(dD_acc); free(db_acc);
    return fail ? 2 : 0;
}

int scan_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
