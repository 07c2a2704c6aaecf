/* A consumer of libgpnative.so that is not Python: one GP log-marginal-likelihood evaluation,
 * its gradients and a prediction through the three whole-path entry points of
 * include/gpnative.h, with buffers from the HIP runtime's C API.
 *
 *   gcc -std=c99 examples/lml_consumer.c -Iinclude -I/opt/rocm/include -Lgptorch_amd/lib -lgpnative \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/gptorch_amd/lib -Wl,-rpath,/opt/rocm/lib -lm \
 *       -o build/lml_consumer && build/lml_consumer 2048 8
 *
 * Prints "lml=<value> grads=<variance ls noise> mean0=<..> var0=<..>"; the same inputs
 * (splitmix64 -> Box-Muller, gptorch_amd/rng.py) give the same numbers through the Python
 * shell (tests/test_gpu_parity.py::test_c_consumer_matches_the_shell). */
#define __HIP_PLATFORM_AMD__ 1
#define _GNU_SOURCE 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "gpnative.h"

#define CHECK(e) do { hipError_t _s = (e); if (_s != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_s)); return 2; } } while (0)
#define GPN(e) do { int _s = (e); if (_s != 0) { fprintf(stderr, "%s -> %d %s\n", #e, _s, gpn_last_hip_error()); return 3; } } while (0)

/* gptorch_amd/rng.py: splitmix64 stream -> uniforms -> Box-Muller normals (pairs) */
static uint64_t sm_state;
static uint64_t splitmix64(void) {
  uint64_t z = (sm_state += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static void normals(uint64_t seed, double* out, int64_t count) {
  sm_state = seed;                                        /* element i from uniforms (2i, 2i+1) */
  for (int64_t i = 0; i < count; ++i) {
    const double u1 = ((double)(splitmix64() >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)(splitmix64() >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    out[i] = sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
  }
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 2048;
  const int d = argc > 2 ? atoi(argv[2]) : 8;
  const int dy = 1, ns = 4;
  double *x = malloc(sizeof(double) * n * d), *eps = malloc(sizeof(double) * n), *y = malloc(sizeof(double) * n);
  double* xs = malloc(sizeof(double) * ns * d);
  normals(0, x, n * d);                                   /* rng.make_regression(n, d, 1, seed=0) */
  normals(1, eps, n);
  for (int64_t i = 0; i < n; ++i) {
    double s = 0.0;
    for (int c = 0; c < d; ++c) s += x[i * d + c];
    y[i] = sin(s) + 0.1 * eps[i];
  }
  normals(2, xs, (int64_t)ns * d);
  const double theta[3] = {1.0, sqrt((double)d), 1e-2};   /* variance, length scale, noise */

  const int64_t lda = gpn_factor_ld(n, dy), rows = gpn_factor_rows(n, dy);
  double *X, *Y, *Xs, *th, *A, *winv, *out3, *work, *grads, *pwork, *mean, *var;
  int32_t* info;
  CHECK(hipMalloc((void**)&X, sizeof(double) * n * d));
  CHECK(hipMalloc((void**)&Y, sizeof(double) * n));
  CHECK(hipMalloc((void**)&Xs, sizeof(double) * ns * d));
  CHECK(hipMalloc((void**)&th, sizeof(theta)));
  CHECK(hipMalloc((void**)&A, sizeof(double) * rows * lda));
  CHECK(hipMalloc((void**)&winv, gpn_winv_bytes(n)));
  CHECK(hipMalloc((void**)&info, sizeof(int32_t)));
  CHECK(hipMalloc((void**)&out3, 3 * sizeof(double)));
  CHECK(hipMalloc((void**)&work, gpn_lml_backward_work_bytes(n, dy, 1)));
  CHECK(hipMalloc((void**)&grads, 3 * sizeof(double)));
  CHECK(hipMalloc((void**)&pwork, gpn_predict_work_bytes(n, ns, dy)));
  CHECK(hipMalloc((void**)&mean, sizeof(double) * ns * dy));
  CHECK(hipMalloc((void**)&var, sizeof(double) * ns));
  CHECK(hipMemcpy(X, x, sizeof(double) * n * d, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(Y, y, sizeof(double) * n, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(Xs, xs, sizeof(double) * ns * d, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(th, theta, sizeof(theta), hipMemcpyHostToDevice));
  CHECK(hipMemset(A, 0, sizeof(double) * rows * lda));    /* factor buffers start zeroed, once */

  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  GPN(gpn_lml_forward(s, GPN_RBF, X, n, d, Y, NULL, dy, th, th + 1, 1, th + 2, A, lda, winv, info, out3));
  GPN(gpn_lml_backward(s, GPN_RBF, X, n, d, th, th + 1, 1, A, lda, winv, dy, work, grads, NULL));
  GPN(gpn_predict(s, GPN_RBF, X, n, d, Xs, ns, NULL /* zero mean function */, th, th + 1, 1, A, lda, winv, dy, 0, pwork, mean, var));
  CHECK(hipStreamSynchronize(s));

  int32_t hinfo;
  double o[3], g[3], m0, v0;
  CHECK(hipMemcpy(&hinfo, info, sizeof(hinfo), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(o, out3, sizeof(o), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(g, grads, sizeof(g), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(&m0, mean, sizeof(double), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(&v0, var, sizeof(double), hipMemcpyDeviceToHost));
  if (hinfo != 0) { fprintf(stderr, "not positive definite at pivot %d: replay with noise + 1e-10..1e-1\n", hinfo); return 4; }
  printf("lml=%.10f grads=%.10e %.10e %.10e mean0=%.12f var0=%.12f\n", o[2], g[0], g[1], g[2], m0, v0);
  return 0;
}
