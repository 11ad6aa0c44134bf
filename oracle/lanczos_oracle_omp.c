/*
 * oracle/lanczos_oracle_omp.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The ALL-CORE companion of orc_lanczos (lanczos_oracle.c): the same loop -- serial/lib/lanczos.cc:9-56 over the spMV of
 * serial/lib/SPMV.cc:19-28 -- with "#pragma omp parallel for" over rows / elements, for the optional all-core CPU figure of
 * SURVEY.md 8(d) ("optionally an OpenMP all-core SpMV number labelled as such").  bench.py's cpu_baseline leg reports it as
 * kind "port-omp" beside the reference-faithful single-thread number; nothing else loads it.
 *
 * What differs from the single-thread restatement, and only that: a row of the spMV is still summed left to right by one
 * thread (same bits as orc_spmv), but inner products and norms are OpenMP reductions, i.e. per-thread partial sums combined
 * in an unspecified order -- a different rounding of alpha / beta.  It is a timing baseline, not a parity oracle; tests only
 * check it against orc_lanczos at a tolerance.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>

int orc_omp_threads(void) { return omp_get_max_threads(); }
/* (the OpenMP runtime may have been started by another library of the process long before: OMP_NUM_THREADS is read only then) */
void orc_omp_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

static void spmv_omp(uint64_t n, const uint64_t *row_offset, const uint32_t *col_idx, const double *in, double *out)
{
    /* rows of a degree-skewed graph: small dynamic chunks even the hubs out */
#pragma omp parallel for schedule(dynamic, 2048)
    for (uint64_t i = 0; i < n; ++i) {
        double acc = 0.0;
        for (uint64_t j = row_offset[i]; j < row_offset[i + 1]; ++j) acc += in[col_idx[j]];
        out[i] = acc;
    }
}

static double dot_omp(uint64_t n, const double *v, const double *w)
{
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (uint64_t i = 0; i < n; ++i) s += v[i] * w[i];
    return s;
}

/* same arguments as orc_lanczos without the basis and the external spMV; returns 0 / -1 */
int orc_lanczos_omp(uint64_t n, const uint64_t *row_offset, const uint32_t *col_idx, uint32_t k, const double *x,
                    double *alpha, double *beta, double *x_norm_out)
{
    double *v = (double *)malloc(sizeof(double) * n);
    double *Q_raw = (double *)malloc(sizeof(double) * 2 * n);
    if (!v || !Q_raw) { free(v); free(Q_raw); return -1; }
    double *Q_s[2] = { Q_raw, Q_raw + n };
    unsigned i = 0;
    const double x_norm = sqrt(dot_omp(n, x, x));
    if (x_norm_out) *x_norm_out = x_norm;
#pragma omp parallel for schedule(static)
    for (uint64_t r = 0; r < n; ++r) { Q_s[0][r] = x[r] / x_norm; Q_s[1][r] = 0.0; }
    for (uint32_t j = 0; j < k; ++j) {
        spmv_omp(n, row_offset, col_idx, Q_s[i], v);
        const double a = dot_omp(n, v, Q_s[i]);
        alpha[j] = a;
        const double b_prev = j > 0 ? beta[j - 1] : 0.0;
        const double *qi = Q_s[i], *qp = Q_s[1 - i];
        /* the reference's two passes (lanczos.cc:29-37), one after the other per element: same roundings */
#pragma omp parallel for schedule(static)
        for (uint64_t r = 0; r < n; ++r) {
            double t = v[r] - a * qi[r];
            if (j > 0) t -= b_prev * qp[r];
            v[r] = t;
        }
        if (j < k - 1) {
            const double b = sqrt(dot_omp(n, v, v));
            beta[j] = b;
            double *qn = Q_s[1 - i];
#pragma omp parallel for schedule(static)
            for (uint64_t r = 0; r < n; ++r) qn[r] = v[r] / b;
        }
        i = 1 - i;
    }
    free(v);
    free(Q_raw);
    return 0;
}
