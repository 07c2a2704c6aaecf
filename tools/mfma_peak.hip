// Micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate on MI355X (register operands only).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, int threads) {
  int iters = 20000;
  int grid = 256 * blocks_per_cu;
  double* out;
  (void)hipMalloc(&out, sizeof(double) * grid * threads);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(threads), 0, 0, out, 100, 1.0, 2.0);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.0, 2.0);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double waves = (double)grid * threads / 64;
  double flops = waves * iters * NACC * 2048.0;
  double cyc_per_mfma_per_simd = (ms * 1e-3 * 2.4e9) / ((double)iters * NACC * (waves / (256.0 * 4)));
  printf("nacc=%d blocks/cu=%d threads=%d: %.2f ms  %.2f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", NACC, blocks_per_cu,
         threads, ms, flops / (ms * 1e-3) / 1e12, cyc_per_mfma_per_simd);
  (void)hipFree(out);
}

int main() {
  run<1>(1, 256); run<4>(1, 256); run<16>(1, 256); run<16>(2, 256); run<8>(2, 512); run<16>(1, 512);
  return 0;
}
