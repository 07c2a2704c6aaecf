// Does s_setprio order fp64 MFMA issue between two waves of one SIMD?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/prio_bench tools/prio_bench.hip
// 512 threads: waves w and w + 4 share a SIMD.  Waves 0 and 4 each run 64 dependent v_mfma_f64_16x16x4_f64 (same accumulator);
// variant 1: wave 4 at s_setprio 3; variant 2: wave 4 at prio 3 and wave 0 inserts s_nop between its MFMAs.  Prints start/end cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned long long* out, double seed, int variant) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  d4 acc = {seed, seed + 1, seed + 2, seed + 3};
  double a = seed + lane * 1e-3, b = 1.0 + lane * 1e-4;
  __syncthreads();
  if (wave != 0 && wave != 4) return;
  if (variant >= 1 && wave == 4) __builtin_amdgcn_s_setprio(3);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    if (variant == 2 && wave == 0) __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" :: "v"(acc));
  unsigned long long t1;
  asm volatile("s_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(acc));
  if (lane == 0) { out[wave * 2] = t0; out[wave * 2 + 1] = t1; }
  if (acc[0] == 1.2345) out[63] = 1;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64 * 8);
  for (int v = 0; v < 3; ++v) {
    for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, d, 1.5, v); hipDeviceSynchronize(); }
    unsigned long long h[64]; hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
    unsigned long long b0 = h[0] < h[8] ? h[0] : h[8];
    printf("variant %d: wave 0 [%llu .. %llu], wave 4 [%llu .. %llu]  (64 dependent MFMAs each; alone = 4096 cycles)\n", v,
           h[0] - b0, h[1] - b0, h[8] - b0, h[9] - b0);
  }
  return 0;
}
