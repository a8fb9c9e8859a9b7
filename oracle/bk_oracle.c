/*
 * CPU oracle for BK-Core in plain C (test infrastructure only: linked by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg -- never by the product path).
 *
 * One row per OpenMP thread, complex double inside, exactly the arithmetic order of the
 * reference's sequential log-scaled path:
 *   src/models/bk_core.py:27-210   get_tridiagonal_inverse_diagonal
 *   src/models/bk_core.py:239-300  BKCoreFunction.forward  (input clamp / sanitise, features)
 *   src/models/bk_core.py:303-385  BKCoreFunction.backward (reference-mode gradient)
 * Parity with the reference is pinned through tests/golden (tests/test_oracle_c.py).
 *
 * Build: see oracle/Makefile  (gcc -O2 -fopenmp -shared -fPIC).
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double complex cd;

static inline cd regularise(cd x) {              /* bk_core.py:78-94 */
    double m = cabs(x);
    cd dir = (m < 1e-12) ? 1.0 : x / (m + 1e-12);
    return x + 1e-3 * dir;
}

static inline double wall(double diff) {         /* bk_core.py:60-75 */
    return exp(88.0 * tanh(diff / 88.0));
}

/* One row.  a[N], b[N-1], c[N-1] already in double.  Writes G as complex double.
 * work: 2*(N+1) complex + 2*(N+1) double scratch supplied by the caller. */
static void row_get_diag(const double *a, const double *b, const double *c, cd z, long n,
                         int phi_exact, cd *G, cd *tth, double *lth, cd *tph, double *lph) {
    if (n == 0) return;
    /* theta sweep  bk_core.py:100-140 */
    tth[0] = 1.0; lth[0] = 0.0;
    cd v = regularise(a[0] - z);
    double s = cabs(v) + 1e-20;
    tth[1] = v / s; lth[1] = log(s);
    for (long i = 1; i < n; ++i) {
        double k = c[i - 1] * b[i - 1];
        v = regularise(a[i] - z) * tth[i] - k * tth[i - 1] * wall(lth[i - 1] - lth[i]);
        s = cabs(v) + 1e-20;
        tth[i + 1] = v / s; lth[i + 1] = lth[i] + log(s);
    }
    cd tdet = tth[n]; double ldet = lth[n];
    /* phi sweep  bk_core.py:150-184 */
    tph[n - 1] = 1.0; lph[n - 1] = 0.0;
    if (n > 1) {
        v = regularise(a[n - 1] - z);
        s = cabs(v) + 1e-20;
        tph[n - 2] = v / s; lph[n - 2] = log(s);
        for (long i = n - 3; i >= 0; --i) {
            double k = phi_exact ? c[i + 1] * b[i + 1] : c[i] * b[i];   /* reference uses k_i (bk_core.py:171) */
            v = regularise(a[i + 1] - z) * tph[i + 1] - k * tph[i + 2] * wall(lph[i + 2] - lph[i + 1]);
            s = cabs(v) + 1e-20;
            tph[i] = v / s; lph[i] = lph[i + 1] + log(s);
        }
    }
    /* combine  bk_core.py:190-208 */
    for (long i = 0; i < n; ++i) {
        double lm = 88.0 * tanh((lth[i] + lph[i] - ldet) / 88.0);
        cd g = (tth[i] * tph[i]) / (tdet + 1e-18) * exp(lm);
        if (!isfinite(creal(g)) || !isfinite(cimag(g))) g = 0.0;
        double mag = cabs(g);
        if (mag > 50.0) g *= 50.0 / (mag + 1e-9);
        G[i] = g;
    }
}

/* Batched get-diag (no input clamps): the vmapped path, bk_core.py:214-216.
 * a: B rows with leading dimension lda (0 = broadcast one row), same for b, c.
 * G_c64: B*N interleaved (re, im) floats (the reference's complex64 output), may be NULL.
 * G_c128: B*N interleaved doubles, may be NULL. */
int bk_oracle_get_diag(const double *a, long lda, const double *b, long ldb, const double *c, long ldc,
                       double z_re, double z_im, long B, long N, int phi_exact,
                       float *G_c64, double *G_c128) {
    if (B < 0 || N < 0) return -1;
    if (B == 0 || N == 0) return 0;
    cd z = z_re + z_im * I;
    int fail = 0;
#pragma omp parallel
    {
        cd *tth = (cd *)malloc(sizeof(cd) * (size_t)(N + 1));
        cd *tph = (cd *)malloc(sizeof(cd) * (size_t)(N + 1));
        double *lth = (double *)malloc(sizeof(double) * (size_t)(N + 1));
        double *lph = (double *)malloc(sizeof(double) * (size_t)(N + 1));
        cd *G = (cd *)malloc(sizeof(cd) * (size_t)N);
        if (!tth || !tph || !lth || !lph || !G) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (long r = 0; r < B; ++r) {
                row_get_diag(a + r * lda, b + r * ldb, c + r * ldc, z, N, phi_exact, G, tth, lth, tph, lph);
                for (long i = 0; i < N; ++i) {
                    if (G_c64) { G_c64[2 * (r * N + i)] = (float)creal(G[i]); G_c64[2 * (r * N + i) + 1] = (float)cimag(G[i]); }
                    if (G_c128) { G_c128[2 * (r * N + i)] = creal(G[i]); G_c128[2 * (r * N + i) + 1] = cimag(G[i]); }
                }
            }
        }
        free(tth); free(tph); free(lth); free(lph); free(G);
    }
    return fail ? -2 : 0;
}

static inline double clampd(double x, double lo, double hi) {   /* torch.clamp keeps NaN */
    if (isnan(x)) return x;
    return x < lo ? lo : (x > hi ? hi : x);
}

/* BKCoreFunction.forward on the sequential path, float32 inputs:
 * clamp a to +-100, b/c to +-10; if any non-finite value remains in a, nan_to_num all three
 * (bk_core.py:256-264; the trigger inspects he_diag only).  Writes G (c64 interleaved) and
 * features (same numbers, bk_core.py:296-298); either may be NULL. */
int bk_oracle_forward_f32(const float *a, long lda, const float *b, long ldb, const float *c, long ldc,
                          double z_re, double z_im, long B, long N, int phi_exact,
                          float *G_c64, float *features) {
    if (B < 0 || N < 0) return -1;
    if (B == 0 || N == 0) return 0;
    long nb = N > 1 ? N - 1 : 0;
    double *ad = (double *)malloc(sizeof(double) * (size_t)(B * N));
    double *bd = (double *)malloc(sizeof(double) * (size_t)(B * nb + 1));
    double *cd_ = (double *)malloc(sizeof(double) * (size_t)(B * nb + 1));
    float *Gtmp = G_c64 ? G_c64 : (float *)malloc(sizeof(float) * 2 * (size_t)(B * N));
    if (!ad || !bd || !cd_ || !Gtmp) { free(ad); free(bd); free(cd_); if (!G_c64) free(Gtmp); return -2; }
    int bad = 0;
    for (long r = 0; r < B; ++r) {
        for (long i = 0; i < N; ++i) {
            double v = clampd((double)a[r * lda + i], -100.0, 100.0);
            if (!isfinite(v)) bad = 1;
            ad[r * N + i] = v;
        }
        for (long i = 0; i < nb; ++i) {
            bd[r * nb + i] = clampd((double)b[r * ldb + i], -10.0, 10.0);
            cd_[r * nb + i] = clampd((double)c[r * ldc + i], -10.0, 10.0);
        }
    }
    if (bad) {
        for (long j = 0; j < B * N; ++j) {
            double v = ad[j];
            ad[j] = isnan(v) ? 0.0 : (isinf(v) ? (v > 0 ? 100.0 : -100.0) : v);
        }
        for (long j = 0; j < B * nb; ++j) {
            double v = bd[j];
            bd[j] = isnan(v) ? 1.0 : (isinf(v) ? (v > 0 ? 10.0 : -10.0) : v);
            v = cd_[j];
            cd_[j] = isnan(v) ? 1.0 : (isinf(v) ? (v > 0 ? 10.0 : -10.0) : v);
        }
    }
    int rc = bk_oracle_get_diag(ad, N, bd, nb, cd_, nb, z_re, z_im, B, N, phi_exact, Gtmp, NULL);
    if (rc == 0 && features) memcpy(features, Gtmp, sizeof(float) * 2 * (size_t)(B * N));
    free(ad); free(bd); free(cd_);
    if (!G_c64) free(Gtmp);
    return rc;
}

/* BKCoreFunction.backward, bk_core.py:303-385, in the reference's own precision (complex64 / float32).
 * G: n interleaved c64; gfeat: n*2 floats or NULL; gG: n interleaved c64 or NULL; grad: n floats. */
int bk_oracle_backward_ref(const float *G, const float *gfeat, const float *gG, double alpha, long n, float *grad) {
    if (n < 0) return -1;
    float al = (float)alpha;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        float complex g = 0.0f;
        if (gfeat) {
            float re = gfeat[2 * i], im = gfeat[2 * i + 1];
            /* torch.nan_to_num defaults (bk_core.py:325): nan->0, +-inf -> +-FLT_MAX.  The reference applies it to
             * the whole tensor when any entry is non-finite; elementwise application is identical. */
            if (isnan(re)) re = 0.0f; else if (isinf(re)) re = re > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
            if (isnan(im)) im = 0.0f; else if (isinf(im)) im = im > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
            g = re + im * I;
        }
        if (gG) g += gG[2 * i] + gG[2 * i + 1] * I;
        float complex gc = conjf(g);
        float complex Gi = G[2 * i] + G[2 * i + 1] * I;
        float complex g2 = Gi * Gi;
        float mag = cabsf(g2);
        float complex den = (mag < 1e-4f) ? g2 / (mag + 1e-9f) * 1e-4f : g2;
        float theo = -crealf(gc * g2);
        float h7 = -crealf(gc / (den + 1e-6f));
        float gv = (1.0f - al) * theo + al * h7;
        if (isnan(gv)) gv = 0.0f; else if (isinf(gv)) gv = gv > 0 ? 100.0f : -100.0f;
        gv = gv < -100.0f ? -100.0f : (gv > 100.0f ? 100.0f : gv);
        grad[i] = gv;
    }
    return 0;
}

int bk_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
