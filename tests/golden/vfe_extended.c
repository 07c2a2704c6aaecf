/* Extended-precision (x87 80-bit long double) value of Titsias' collapsed bound exactly as the reference evaluates it
 * (gptorch/models/sparse_gpr.py:108-153: A = L^-1 Kuf, B = I + A A^T / s2, c = LB^-1 A y / s2, and the six terms of the bound),
 * for an Rbf kernel, zero mean, one output column and a GIVEN jitter on K(Z) (the rung the reference's ladder,
 * functions.py:20-43, stops at for this matrix).  Test infrastructure (tests/golden/make_vfe_extended.py drives it): every
 * kernel entry, every sum and both Cholesky factorisations are long double, so the result carries ~1e-19 relative rounding
 * per operation where any fp64 evaluation carries 1e-16 -- the value that the reference's fp64 number, the CPU oracle's and
 * the GPU's are all approximating.  Nothing M x N is factored: with Phi = Kuf Kuf^T (M x M) and v = Kuf y,
 *     A A^T = L^-1 Phi L^-T,   A y = L^-1 v,
 * which are the same quantities in exact arithmetic.
 *
 * ROUND_ENTRIES=1 in the environment: every kernel entry is rounded to fp64 before it is used (the matrices an fp64
 * implementation starts from, to within an ulp) and everything after that stays long double -- separates the rounding of the
 * INPUTS, which every fp64 evaluation shares, from the rounding of the factorisations and sums.
 *
 * usage: vfe_extended <n> <m> <d> <variance> <length_scale> <noise> <jitter> <x.bin> <y.bin> <z.bin>   (fp64 row-major files)
 * prints one JSON object.  gcc -O2 -fopenmp vfe_extended.c -o vfe_extended -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef long double ld;

static double* read_f64(const char* path, size_t count) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  double* p = (double*)malloc(count * sizeof(double));
  if (fread(p, sizeof(double), count, f) != count) { fprintf(stderr, "short read %s\n", path); exit(2); }
  fclose(f);
  return p;
}

/* in-place lower Cholesky, row-major; returns 0 or the 1-based index of the first non-positive pivot */
static int chol(ld* a, int m) {
  for (int j = 0; j < m; ++j) {
    ld s = a[(size_t)j * m + j];
    for (int k = 0; k < j; ++k) s -= a[(size_t)j * m + k] * a[(size_t)j * m + k];
    if (!(s > 0)) return j + 1;
    const ld piv = sqrtl(s);
    a[(size_t)j * m + j] = piv;
#pragma omp parallel for schedule(static)
    for (int i = j + 1; i < m; ++i) {
      ld t = a[(size_t)i * m + j];
      const ld* ri = a + (size_t)i * m;
      const ld* rj = a + (size_t)j * m;
      for (int k = 0; k < j; ++k) t -= ri[k] * rj[k];
      a[(size_t)i * m + j] = t / piv;
    }
  }
  return 0;
}

/* X <- L^-1 X for X [m x k] row-major (forward substitution on every column) */
static void solve_lower(const ld* L, ld* X, int m, int k) {
#pragma omp parallel for schedule(static)
  for (int c = 0; c < k; ++c)
    for (int i = 0; i < m; ++i) {
      ld t = X[(size_t)i * k + c];
      for (int j = 0; j < i; ++j) t -= L[(size_t)i * m + j] * X[(size_t)j * k + c];
      X[(size_t)i * k + c] = t / L[(size_t)i * m + i];
    }
}

int main(int argc, char** argv) {
  if (argc != 11) { fprintf(stderr, "usage: see the header\n"); return 2; }
  const long n = atol(argv[1]);
  const int m = atoi(argv[2]), d = atoi(argv[3]);
  const ld variance = strtold(argv[4], NULL), ell = strtold(argv[5], NULL), s2 = strtold(argv[6], NULL), jitter = strtold(argv[7], NULL);
  double* X = read_f64(argv[8], (size_t)n * d);
  double* Y = read_f64(argv[9], (size_t)n);
  double* Z = read_f64(argv[10], (size_t)m * d);
  const int round_entries = getenv("ROUND_ENTRIES") && atoi(getenv("ROUND_ENTRIES")) != 0;
  /* Phi = Kuf Kuf^T (lower), v = Kuf y, in row chunks of the data: Kc [m x C] long double */
  const long C = 4096;
  ld* Kc = (ld*)malloc((size_t)m * C * sizeof(ld));
  ld* Phi = (ld*)calloc((size_t)m * m, sizeof(ld));
  ld* v = (ld*)calloc((size_t)m, sizeof(ld));
  ld yy = 0;
  for (long c0 = 0; c0 < n; c0 += C) {
    const long cn = (n - c0 < C) ? n - c0 : C;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < m; ++i)
      for (long j = 0; j < cn; ++j) {
        ld r2 = 0;
        for (int k = 0; k < d; ++k) {
          const ld df = ((ld)Z[(size_t)i * d + k] - (ld)X[(size_t)(c0 + j) * d + k]) / ell;
          r2 += df * df;
        }
        const ld kv = variance * expl(-r2 / 2);
        Kc[(size_t)i * C + j] = round_entries ? (ld)(double)kv : kv;
      }
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < m; ++i) {
      const ld* ki = Kc + (size_t)i * C;
      for (int j = 0; j <= i; ++j) {
        const ld* kj = Kc + (size_t)j * C;
        ld s0 = 0, s1 = 0, s2_ = 0, s3 = 0;
        long q = 0;
        for (; q + 4 <= cn; q += 4) { s0 += ki[q] * kj[q]; s1 += ki[q + 1] * kj[q + 1]; s2_ += ki[q + 2] * kj[q + 2]; s3 += ki[q + 3] * kj[q + 3]; }
        for (; q < cn; ++q) s0 += ki[q] * kj[q];
        Phi[(size_t)i * m + j] += (s0 + s1) + (s2_ + s3);
      }
      ld t = 0;
      for (long q = 0; q < cn; ++q) t += ki[q] * (ld)Y[c0 + q];
      v[i] += t;
    }
    for (long q = 0; q < cn; ++q) yy += (ld)Y[c0 + q] * (ld)Y[c0 + q];
  }
  for (int i = 0; i < m; ++i) for (int j = i + 1; j < m; ++j) Phi[(size_t)i * m + j] = Phi[(size_t)j * m + i];
  /* L = chol(K(Z) + jitter I) */
  ld* L = (ld*)calloc((size_t)m * m, sizeof(ld));
  for (int i = 0; i < m; ++i)
    for (int j = 0; j <= i; ++j) {
      ld r2 = 0;
      for (int k = 0; k < d; ++k) { const ld df = ((ld)Z[(size_t)i * d + k] - (ld)Z[(size_t)j * d + k]) / ell; r2 += df * df; }
      const ld kv = variance * expl(-r2 / 2);
      L[(size_t)i * m + j] = (round_entries ? (ld)(double)kv : kv) + (i == j ? jitter : 0);
    }
  int info = chol(L, m);
  if (info) { printf("{\"error\": \"K(Z) + jitter not positive definite in long double at pivot %d\"}\n", info); return 1; }
  ld logdetL = 0;
  for (int i = 0; i < m; ++i) logdetL += logl(L[(size_t)i * m + i]);
  /* AAT = L^-1 Phi L^-T / s2 : T = L^-1 Phi, then AAT^T = L^-1 T^T */
  solve_lower(L, Phi, m, m);                        /* Phi <- L^-1 Phi */
  ld* T = (ld*)malloc((size_t)m * m * sizeof(ld));
  for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) T[(size_t)i * m + j] = Phi[(size_t)j * m + i];
  solve_lower(L, T, m, m);                          /* T = L^-1 (L^-1 Phi)^T = A A^T (symmetric) */
  ld trAAT = 0;
  for (int i = 0; i < m; ++i) trAAT += T[(size_t)i * m + i] / s2;
  solve_lower(L, v, m, 1);                          /* v <- A y */
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) T[(size_t)i * m + j] = T[(size_t)i * m + j] / s2 + (i == j ? 1 : 0);      /* B */
  info = chol(T, m);
  if (info) { printf("{\"error\": \"B not positive definite at pivot %d\"}\n", info); return 1; }
  ld logdetLB = 0;
  for (int i = 0; i < m; ++i) logdetLB += logl(T[(size_t)i * m + i]);
  solve_lower(T, v, m, 1);                          /* LB^-1 A y */
  ld cc = 0;
  for (int i = 0; i < m; ++i) { const ld ci = v[i] / s2; cc += ci * ci; }
  const ld pi = 3.14159265358979323846264338327950288L;
  ld elbo = -0.5L * (ld)n * logl(2 * pi);
  elbo -= logdetLB;
  elbo -= 0.5L * (ld)n * logl(s2);
  elbo -= 0.5L * (yy + (ld)n * variance) / s2;
  elbo += 0.5L * cc;
  elbo += 0.5L * trAAT;
  printf("{\"elbo_extended\": %.21Lg, \"logdet_LB\": %.21Lg, \"logdet_L\": %.21Lg, \"c_sq\": %.21Lg, \"tr_AAT\": %.21Lg, \"yy\": %.21Lg}\n",
         elbo, logdetLB, logdetL, cc, trAAT, yy);
  return 0;
}
