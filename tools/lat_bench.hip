// Single-wave instruction-cost probes for the leaf's pivot chain (gfx950).  Build:
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/lat_bench tools/lat_bench.hip
// Each probe runs REP copies of a pattern between two s_memtime stamps in one wave (nothing else on the CU) and prints cycles per copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define REP 64
// a stamp that the compiler cannot move past the values it is tied to
#define TIE(t, v0, v1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(v0), "+v"(v1) :: "memory")
__device__ __forceinline__ double bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
#define TIEA() asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(acc), "+v"(acc2), "+v"(aj), "+v"(an), "+v"(lo), "+v"(hi))
__global__ void probes(double* out, unsigned long long* cyc, double seed) {
  __shared__ double lds[64 * 20];
  const int lane = threadIdx.x;
  double x = seed + lane * 1e-3, y = 1.0 + lane * 1e-4, z = 0.5;
  unsigned long long t0, t1;
  int k = 0;
  double a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = x + j;
  d4 acc = {x, y, z, x}, acc2 = {y, y, z, x};
  double aj = fabs(x) + 2.0, an = fabs(y) + 3.0;
  int lo = __double2loint(x), hi = __double2hiint(x), lo2 = lane, hi2 = lane + 1;
  // 0: dependent v_fma_f64 chain
  TIE(t0, x, y); TIEA();
#pragma unroll
  for (int i = 0; i < REP; ++i) x = fma(x, y, z);
  asm volatile("" ::"v"(x));
  TIEA(); TIE(t1, x, y); cyc[k++] = t1 - t0;
  // 1: 8 independent fma chains (throughput)
  TIE(t0, x, y); TIEA();
#pragma unroll
  for (int i = 0; i < REP / 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = fma(a[j], y, z);
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(a[j]));
  TIEA(); TIE(t1, x, y); cyc[k++] = t1 - t0;
  // 2: dependent v_rsq_f64 chain
  TIE(t0, x, y); TIEA();
#pragma unroll
  for (int i = 0; i < REP; ++i) x = __builtin_amdgcn_rsq(x);
  asm volatile("" ::"v"(x));
  TIEA(); TIE(t1, x, y); cyc[k++] = t1 - t0;
  // 3: readlane pair -> fma using it -> readlane of the result ... (fully dependent: the pivot hop)
  TIE(t0, x, y); TIEA();
#pragma unroll
  for (int i = 0; i < REP; ++i) { const double s = bcast(x, i & 15); x = fma(y, s, z); }
  asm volatile("" ::"v"(x));
  TIEA(); TIE(t1, x, y); cyc[k++] = t1 - t0;
  // 4: independent (readlane pair + fma) groups, as the compiler schedules them (the lazy column updates)
  TIE(t0, x, y); TIEA();
#pragma unroll
  for (int i = 0; i < REP / 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = fma(-y, bcast(y, j + (i & 7)), a[j]);
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(a[j]));
  TIEA(); TIE(t1, x, y); cyc[k++] = t1 - t0;
  // 5: 8 readlane pairs first (distinct SGPRs), then 8 fmas
  TIE(t0, x, y); TIEA();
#pragma unroll
  for (int i = 0; i < REP / 8; ++i) {
    double s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = bcast(y, j + (i & 7));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = fma(-y, s[j], a[j]);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(a[j]));
  TIEA(); TIE(t1, x, y); cyc[k++] = t1 - t0;
  // 6: LDS broadcast: one ds_write_b64 + 4 ds_read_b128 (8 uniform doubles) + 8 fmas, waited
  TIE(t0, x, y); TIEA();
#pragma unroll
  for (int i = 0; i < REP / 8; ++i) {
    lds[lane] = y + i;
    double s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = lds[j + (i & 7)];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = fma(-y, s[j], a[j]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(a[j]));
  TIEA(); TIE(t1, x, y); cyc[k++] = t1 - t0;
  // 7: LDS write -> dependent read round trip (latency)
  TIE(t0, x, y); TIEA();
#pragma unroll
  for (int i = 0; i < REP; ++i) { lds[lane] = x; x = lds[(lane + 1) & 63] + 1.0; }
  asm volatile("" ::"v"(x));
  TIEA(); TIE(t1, x, y); cyc[k++] = t1 - t0;
  // 8: dependent MFMA f64 16x16x4 chain (same accumulator)
  TIE(t0, x, y); TIEA();
#pragma unroll
  for (int i = 0; i < REP; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0);
  asm volatile("" ::"v"(acc));
  TIEA(); TIE(t1, x, y); cyc[k++] = t1 - t0;
  // 9: MFMA whose A operand is the previous result (X -> X X^T hop)
  TIE(t0, x, y); TIEA();
#pragma unroll
  for (int i = 0; i < REP; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[0], z, acc, 0, 0, 0);
  asm volatile("" ::"v"(acc));
  TIEA(); TIE(t1, x, y); cyc[k++] = t1 - t0;
  // 10: two independent MFMA chains interleaved
  TIE(t0, x, y); TIEA();
#pragma unroll
  for (int i = 0; i < REP / 2; ++i) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc2, 0, 0, 0);
  }
  asm volatile("" ::"v"(acc), "v"(acc2));
  TIEA(); TIE(t1, x, y); cyc[k++] = t1 - t0;
  // 11: v_permlane32_swap pair (a double) dependent chain
  {
    TIE(t0, x, y); TIEA();
#pragma unroll
    for (int i = 0; i < REP; ++i) {
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(lo2));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(hi), "+v"(hi2));
    }
    TIEA(); TIE(t1, x, y); cyc[k++] = t1 - t0;
    x = __hiloint2double(hi ^ hi2, lo ^ lo2);
  }
  // 12: the pivot chain itself: readlane d -> rsq -> refine -> l -> readlane -> fma -> (next)
  {
    TIE(t0, x, y); TIEA();
#pragma unroll
    for (int i = 0; i < REP; ++i) {
      const double d = bcast(aj, i & 15);
      const double y0 = __builtin_amdgcn_rsq(d);
      const double ay0 = aj * y0;
      const double e = fma(-d * y0, y0, 1.0);
      const double pp = fma(e, 0.375, 0.5);
      const double l = fma(ay0 * e, pp, ay0);
      aj = fma(-l, bcast(l, (i + 1) & 15), an) + 4.0;
    }
    asm volatile("" ::"v"(aj));
    TIEA(); TIE(t1, x, y); cyc[k++] = t1 - t0;
    x += aj;
  }
  // 13: empty stamp pair
  TIE(t0, x, y); TIE(t1, x, y); cyc[k++] = t1 - t0;
  out[lane] = x + a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7] + acc[0] + acc2[1];
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 64 * 8);
  const char* names[] = {"dependent v_fma_f64", "independent v_fma_f64 (8 chains)", "dependent v_rsq_f64", "readlane pair -> fma -> readlane (dependent hop)",
                         "independent readlane pair + fma (compiler order)", "8 readlane pairs, then 8 fmas", "LDS broadcast: write + 8 uniform doubles + 8 fmas (per fma)",
                         "LDS write -> read round trip", "dependent MFMA f64 16x16x4 (same acc)", "MFMA with A = previous result", "two interleaved MFMA chains (per MFMA)",
                         "v_permlane32_swap pair", "pivot chain (readlane, rsq, refine, l, readlane, fma)", "empty stamp pair"};
  for (int it = 0; it < 3; ++it) {
    hipLaunchKernelGGL(probes, dim3(1), dim3(64), 0, 0, out, cyc, 1.25);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(16);
  hipMemcpy(h.data(), cyc, 14 * 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < 14; ++i) printf("%-64s %8.1f cycles each (total %llu)\n", names[i], (double)h[i] / (i == 13 ? 1 : REP), h[i]);
  return 0;
}
