/*
 * TEST INFRASTRUCTURE ONLY -- plain-C restatement of the reference's GPR hot path
 * (cics-nd/gptorch v0.3.2), independent of PyTorch/MKL: textbook unblocked
 * algorithms in the reference's own operation order.  Second opinion for the
 * torch-CPU oracle (gp_oracle.py) and for the HIP path at small sizes.
 * Never linked into the product (gptorch_amd/).
 *
 * Citations are relative to /root/reference/.
 */
#include <math.h>
#include <stdlib.h>

/* util.squared_distance (gptorch/util.py:73-88): Gram trick
 * r2 = |a|^2 + |b|^2 - 2 a.b, negatives clamped to 0 -- on inputs already divided
 * by the length-scales (kernels.py:149-159). */
static void scaled_sqdist(const double* x, int n, const double* x2, int m, int d, const double* ls, int nls, double* r2) {
    double* xs = (double*)malloc(sizeof(double) * (size_t)n * d);
    double* ys = (double*)malloc(sizeof(double) * (size_t)m * d);
    for (int i = 0; i < n; ++i) for (int k = 0; k < d; ++k) xs[i * d + k] = x[i * d + k] / ls[nls == 1 ? 0 : k];
    for (int j = 0; j < m; ++j) for (int k = 0; k < d; ++k) ys[j * d + k] = x2[j * d + k] / ls[nls == 1 ? 0 : k];
    for (int i = 0; i < n; ++i) {
        double xi = 0.0;
        for (int k = 0; k < d; ++k) xi += xs[i * d + k] * xs[i * d + k];
        for (int j = 0; j < m; ++j) {
            double yj = 0.0, dot = 0.0;
            for (int k = 0; k < d; ++k) { yj += ys[j * d + k] * ys[j * d + k]; dot += xs[i * d + k] * ys[j * d + k]; }
            double v = xi + yj - 2.0 * dot;
            r2[(size_t)i * m + j] = v < 0.0 ? 0.0 : v;
        }
    }
    free(xs); free(ys);
}

/* kind: 0 Rbf (kernels.py:215-222), 1 Matern52 (204-212), 2 Matern32 (196-201), 3 Exp (182-190),
 * 5 Periodic (228-235) */
int gpo_kernel_matrix(int kind, const double* x, int n, const double* x2, int m, int d,
                      double variance, const double* ls, int nls, double* K) {
    scaled_sqdist(x, n, x2, m, d, ls, nls, K);
    for (size_t t = 0; t < (size_t)n * m; ++t) {
        double r2 = K[t];
        if (kind == 0) { K[t] = variance * exp(-r2 / 2.0); continue; }
        double r = sqrt(r2 < 1e-40 ? 1e-40 : r2);                       /* kernels.py:172 */
        if (kind == 1) { double s5 = sqrt(5.0); K[t] = variance * (1.0 + s5 * r + 5.0 / 3.0 * r * r) * exp(-s5 * r); }
        else if (kind == 2) { double r3 = sqrt(3.0) * r; K[t] = variance * (1.0 + r3) * exp(-r3); }
        else if (kind == 5) K[t] = variance * cos(r);
        else K[t] = variance * exp(-r);
    }
    return 0;
}

/* torch.cholesky (functions.py:46-47): lower, in place; returns LAPACK info (j>0: pivot j not positive) */
int gpo_cholesky(double* a, int n) {
    for (int j = 0; j < n; ++j) {
        double d = a[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= a[(size_t)j * n + k] * a[(size_t)j * n + k];
        if (!(d > 0.0)) return j + 1;
        d = sqrt(d);
        a[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = a[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= a[(size_t)i * n + k] * a[(size_t)j * n + k];
            a[(size_t)i * n + j] = s / d;
        }
        for (int c = j + 1; c < n; ++c) a[(size_t)j * n + c] = 0.0;
    }
    return 0;
}

/* functions.trtrs(b, L) (functions.py:71-76): forward substitution, b[n,k] in place */
void gpo_trtrs_lower(const double* L, int n, double* b, int k) {
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < k; ++c) {
            double s = b[(size_t)i * k + c];
            for (int j = 0; j < i; ++j) s -= L[(size_t)i * n + j] * b[(size_t)j * k + c];
            b[(size_t)i * k + c] = s / L[(size_t)i * n + i];
        }
}

/* GPR.log_likelihood (gpr.py:47-67) with the jitter ladder of functions.py:20-43.
 * resid = y - mean(x) [n,dy].  Returns the rung used (-1 plain) or -100 on "Max tries exceeded". */
int gpo_gpr_lml(int kind, const double* x, int n, int d, const double* resid, int dy,
                double variance, const double* ls, int nls, double noise, double* lml_out) {
    double* K = (double*)malloc(sizeof(double) * (size_t)n * n);
    double* A = (double*)malloc(sizeof(double) * (size_t)n * n);
    double* al = (double*)malloc(sizeof(double) * (size_t)n * dy);
    gpo_kernel_matrix(kind, x, n, x, n, d, variance, ls, nls, K);
    for (int i = 0; i < n; ++i) K[(size_t)i * n + i] += noise;            /* gpr.py:80-86 */
    int rung = -2;
    for (int t = -1; t < 10 && rung == -2; ++t) {
        double jit = t < 0 ? 0.0 : pow(10.0, -10 + t);
        for (size_t q = 0; q < (size_t)n * n; ++q) A[q] = K[q];
        for (int i = 0; i < n; ++i) A[(size_t)i * n + i] += jit;
        if (gpo_cholesky(A, n) == 0) rung = t;
    }
    if (rung == -2) { free(K); free(A); free(al); return -100; }
    for (size_t q = 0; q < (size_t)n * dy; ++q) al[q] = resid[q];
    gpo_trtrs_lower(A, n, al, dy);
    double quad = 0.0, logdet = 0.0;
    for (size_t q = 0; q < (size_t)n * dy; ++q) quad += al[q] * al[q];
    for (int i = 0; i < n; ++i) logdet += log(A[(size_t)i * n + i]);       /* functions.py:61-68 */
    *lml_out = -0.5 * quad - dy * logdet - 0.5 * dy * n * log(2.0 * M_PI);  /* gpr.py:63-67 */
    free(K); free(A); free(al);
    return rung;
}
