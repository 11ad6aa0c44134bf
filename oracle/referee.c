/*
 * oracle/referee.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Accuracy referee for the Lanczos e^A x path: the loop of serial/lib/lanczos.cc:9-56 (the same recurrence the
 * oracle restates in lanczos_oracle.c) carried out in x87 extended precision (long double: 64-bit significand,
 * 2^-11 of fp64's rounding) for EVERYTHING -- the start vector, the SpMV sums, both inner products, the norms,
 * the stored basis, the small tridiagonal eigenproblem and the back-projection -- optionally with full
 * re-orthogonalisation (two modified Gram-Schmidt sweeps against every earlier vector in every iteration),
 * which makes it a stand-in for the exact-arithmetic k-step Lanczos approximation.
 *
 * What it is for: the parity tests at BASELINE's own k = 50 compare the engine with
 * the oracle, but fifty steps without re-orthogonalisation are not reproducible to 1e-10 by the fp64 algorithm
 * itself on the R-MAT graphs; the referee is something MORE accurate than both, so that the tests can assert
 *     err(engine vs referee) <= 1.5 * err(oracle vs referee)
 * instead of a multiple of the oracle's own noise.  It is not a restatement of a reference function (the
 * reference has no extended-precision path) and it pins nothing about operation order: sums are taken in fixed
 * chunks so that the result does not depend on the number of threads.
 *
 * Only tests/ may load this file's shared object (oracle/libreferee.so, `make -C oracle referee`).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef long double ld;

/* A GPU box hands a job a CPU share, not the machine: 256 hardware threads are visible, about 16 are ours, and an OpenMP
 * team of 256 spinning at every barrier then crawls.  The caller says how many threads to use (oracle.py: at most 16). */
void ref_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

#define REF_CHUNK 65536u

/* y = A x, pattern-only CSR (serial/lib/SPMV.cc:19-28 with a long double accumulator); rows in parallel */
static void ref_spmv(uint64_t n, const uint64_t *ro, const uint32_t *ci, const ld *in, ld *out)
{
#pragma omp parallel for schedule(dynamic, 4096)
    for (uint64_t i = 0; i < n; ++i) {
        ld acc = 0.0L;
        for (uint64_t j = ro[i]; j < ro[i + 1]; ++j) acc += in[ci[j]];
        out[i] = acc;
    }
}

/* <a, b>: chunks of REF_CHUNK summed left to right, chunk totals summed left to right (thread-count independent) */
static ld ref_dot(uint64_t n, const ld *a, const ld *b, ld *scratch)
{
    const uint64_t nc = (n + REF_CHUNK - 1) / REF_CHUNK;
#pragma omp parallel for schedule(static)
    for (uint64_t c = 0; c < nc; ++c) {
        const uint64_t lo = c * REF_CHUNK, hi = lo + REF_CHUNK < n ? lo + REF_CHUNK : n;
        ld s = 0.0L;
        for (uint64_t i = lo; i < hi; ++i) s += a[i] * b[i];
        scratch[c] = s;
    }
    ld s = 0.0L;
    for (uint64_t c = 0; c < nc; ++c) s += scratch[c];
    return s;
}

static void ref_axpy(uint64_t n, ld a, const ld *x, ld *y)   /* y -= a x */
{
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n; ++i) y[i] -= a * x[i];
}

/* Cyclic Jacobi on a dense symmetric k x k matrix (row-major a, destroyed): eigenvalues in lam, eigenvectors in
 * the COLUMNS of v (v[i*k + j] = component i of eigenvector j). */
static void ref_jacobi(uint32_t k, ld *a, ld *lam, ld *v)
{
    for (uint32_t i = 0; i < k; ++i)
        for (uint32_t j = 0; j < k; ++j) v[(size_t)i * k + j] = i == j ? 1.0L : 0.0L;
    for (int sweep = 0; sweep < 60; ++sweep) {
        ld off = 0.0L, diag = 0.0L;
        for (uint32_t i = 0; i < k; ++i) {
            diag += a[(size_t)i * k + i] * a[(size_t)i * k + i];
            for (uint32_t j = i + 1; j < k; ++j) off += a[(size_t)i * k + j] * a[(size_t)i * k + j];
        }
        if (off <= (ld)LDBL_EPSILON * (ld)LDBL_EPSILON * diag * 1e-6L || off == 0.0L) break;
        for (uint32_t p = 0; p + 1 < k; ++p)
            for (uint32_t q = p + 1; q < k; ++q) {
                const ld apq = a[(size_t)p * k + q];
                if (apq == 0.0L) continue;
                const ld app = a[(size_t)p * k + p], aqq = a[(size_t)q * k + q];
                const ld theta = (aqq - app) / (2.0L * apq);
                const ld t = (theta >= 0.0L ? 1.0L : -1.0L) / (fabsl(theta) + sqrtl(theta * theta + 1.0L));
                const ld c = 1.0L / sqrtl(t * t + 1.0L), s = t * c;
                for (uint32_t r = 0; r < k; ++r) {   /* columns p, q */
                    const ld arp = a[(size_t)r * k + p], arq = a[(size_t)r * k + q];
                    a[(size_t)r * k + p] = c * arp - s * arq;
                    a[(size_t)r * k + q] = s * arp + c * arq;
                }
                for (uint32_t r = 0; r < k; ++r) {   /* rows p, q */
                    const ld apr = a[(size_t)p * k + r], aqr = a[(size_t)q * k + r];
                    a[(size_t)p * k + r] = c * apr - s * aqr;
                    a[(size_t)q * k + r] = s * apr + c * aqr;
                }
                for (uint32_t r = 0; r < k; ++r) {
                    const ld vrp = v[(size_t)r * k + p], vrq = v[(size_t)r * k + q];
                    v[(size_t)r * k + p] = c * vrp - s * vrq;
                    v[(size_t)r * k + q] = s * vrp + c * vrq;
                }
            }
    }
    for (uint32_t i = 0; i < k; ++i) lam[i] = a[(size_t)i * k + i];
}

/*
 * The whole pipeline in extended precision.
 *   reorth: 0 = the plain three-term recurrence (serial/lib/lanczos.cc:9-56 at higher precision);
 *           1 = full re-orthogonalisation: v is orthogonalised twice (modified Gram-Schmidt) against q_0 .. q_j
 *               after the three-term update of every iteration;
 *           2 = the schedule of serial/lib/lanczos.cc:58-132 (decompose_with_arnoldi): when j % 2 == 0 and j > 2,
 *               A q_j is orthogonalised once against q_0 .. q_{j-2} before alpha_j is taken;
 *           100 + e = the same with `every` = e instead of the reference's 2.
 *   caps[ncaps]: one answer per entry, ans[c][n] = ||x|| Q V (exp(s (lam - lam_max)) .* V[0,:]) with
 *           s = min(1, caps[c] / lam_max) for caps[c] > 0, s = 1 for caps[c] == 0 (e^(A - theta_max) x), and for
 *           caps[c] < 0 the unshifted e^A x = ||x|| Q V (exp(lam) .* V[0,:]) (serial/lib/multiplyOut.cc:17-37).
 *   alpha_out[k], beta_out[k-1], lam_out[k] (any may be NULL): rounded to double.
 * Returns 0, -1 on allocation failure.
 */
int ref_expm_ld(uint64_t n, const uint64_t *ro, const uint32_t *ci, uint32_t k, const double *x, int reorth,
                uint32_t ncaps, const double *caps, double *ans, double *alpha_out, double *beta_out,
                double *lam_out, double *orth_loss_out)
{
    const uint64_t nc = (n + REF_CHUNK - 1) / REF_CHUNK;
    ld *Q = (ld *)malloc(sizeof(ld) * (size_t)k * n);
    ld *v = (ld *)malloc(sizeof(ld) * n);
    ld *scratch = (ld *)malloc(sizeof(ld) * (nc + 1));
    ld *alpha = (ld *)calloc(k, sizeof(ld)), *beta = (ld *)calloc(k, sizeof(ld));
    ld *T = (ld *)calloc((size_t)k * k, sizeof(ld)), *V = (ld *)malloc(sizeof(ld) * (size_t)k * k);
    ld *lam = (ld *)malloc(sizeof(ld) * k), *t = (ld *)malloc(sizeof(ld) * k);
    int rc = -1;
    if (!Q || !v || !scratch || !alpha || !beta || !T || !V || !lam || !t) goto out;

    {
        ld *q0 = Q;
#pragma omp parallel for schedule(static)
        for (uint64_t i = 0; i < n; ++i) q0[i] = (ld)x[i];
        const ld x_norm = sqrtl(ref_dot(n, q0, q0, scratch));
#pragma omp parallel for schedule(static)
        for (uint64_t i = 0; i < n; ++i) q0[i] /= x_norm;

        for (uint32_t j = 0; j < k; ++j) {
            const ld *qj = Q + (size_t)j * n;
            ref_spmv(n, ro, ci, qj, v);
            if (reorth >= 2 && j % (uint32_t)(reorth >= 100 ? reorth - 100 : 2) == 0 && j > 2)
                for (uint32_t m = 0; m + 1 < j; ++m) {
                    const ld d = ref_dot(n, v, Q + (size_t)m * n, scratch);
                    ref_axpy(n, d, Q + (size_t)m * n, v);
                }
            alpha[j] = ref_dot(n, v, qj, scratch);
            ref_axpy(n, alpha[j], qj, v);
            if (j > 0) ref_axpy(n, beta[j - 1], Q + (size_t)(j - 1) * n, v);
            if (reorth == 1)
                for (int pass = 0; pass < 2; ++pass)
                    for (uint32_t m = 0; m <= j; ++m) {
                        const ld d = ref_dot(n, v, Q + (size_t)m * n, scratch);
                        ref_axpy(n, d, Q + (size_t)m * n, v);
                    }
            if (j + 1 < k) {
                beta[j] = sqrtl(ref_dot(n, v, v, scratch));
                ld *qn = Q + (size_t)(j + 1) * n;
                const ld b = beta[j];
#pragma omp parallel for schedule(static)
                for (uint64_t i = 0; i < n; ++i) qn[i] = v[i] / b;
            }
        }

        if (orth_loss_out) {   /* max_j |q_0 . q_j|, j >= 2: how far the basis has drifted from orthogonality */
            ld worst = 0.0L;
            for (uint32_t j = 2; j < k; ++j) {
                const ld d = fabsl(ref_dot(n, Q, Q + (size_t)j * n, scratch));
                if (d > worst) worst = d;
            }
            *orth_loss_out = (double)worst;
        }

        for (uint32_t i = 0; i < k; ++i) {
            T[(size_t)i * k + i] = alpha[i];
            if (i + 1 < k) T[(size_t)i * k + i + 1] = T[(size_t)(i + 1) * k + i] = beta[i];
        }
        ref_jacobi(k, T, lam, V);
        ld lmax = lam[0];
        for (uint32_t i = 1; i < k; ++i) if (lam[i] > lmax) lmax = lam[i];

        for (uint32_t c = 0; c < ncaps; ++c) {
            const ld cap = (ld)caps[c];
            const ld s = cap > 0.0L && lmax > cap ? cap / lmax : 1.0L;
            const ld shift = cap < 0.0L ? 0.0L : lmax;
            for (uint32_t i = 0; i < k; ++i) {
                ld acc = 0.0L;
                for (uint32_t m = 0; m < k; ++m)
                    acc += V[(size_t)i * k + m] * (expl(s * (lam[m] - shift)) * (x_norm * V[m]));   /* V[0*k + m] */
                t[i] = acc;
            }
            double *a = ans + (size_t)c * n;
#pragma omp parallel for schedule(static)
            for (uint64_t r = 0; r < n; ++r) {
                ld acc = 0.0L;
                for (uint32_t jj = 0; jj < k; ++jj) acc += t[jj] * Q[(size_t)jj * n + r];
                a[r] = (double)acc;
            }
        }
        for (uint32_t i = 0; i < k; ++i) {
            if (alpha_out) alpha_out[i] = (double)alpha[i];
            if (beta_out && i + 1 < k) beta_out[i] = (double)beta[i];
            if (lam_out) lam_out[i] = (double)lam[i];
        }
        rc = 0;
    }
out:
    free(Q); free(v); free(scratch); free(alpha); free(beta); free(T); free(V); free(lam); free(t);
    return rc;
}

/* one SpMV with long double row sums, rounded once: the most accurate y = A x this box can form cheaply */
int ref_spmv_ld(uint64_t n, const uint64_t *ro, const uint32_t *ci, const double *x, double *y)
{
    ld *a = (ld *)malloc(sizeof(ld) * n), *b = (ld *)malloc(sizeof(ld) * n);
    if (!a || !b) { free(a); free(b); return -1; }
    for (uint64_t i = 0; i < n; ++i) a[i] = (ld)x[i];
    ref_spmv(n, ro, ci, a, b);
    for (uint64_t i = 0; i < n; ++i) y[i] = (double)b[i];
    free(a); free(b);
    return 0;
}
