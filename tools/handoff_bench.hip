// Micro-benchmark behind DESIGN 8-1f (ii): the leaf -> column-solve dependency of the factorisation's serial chain as
//   A. two dependent kernel launches per step (what the driver does): a 1-workgroup producer writes a 128 KB block (the inverted
//      leaf), a 255-workgroup consumer reads all of it in every workgroup (each solve workgroup loads W) -- plain stores / loads;
//   B. ONE persistent launch of 256 workgroups (one per CU) with the hand-off the micro-architecture guide documents
//      (MI355X_MICROARCH.md "Valid forms": every payload byte stored and loaded `sc1` = agent-scope relaxed atomics, every
//      storing wave drains its stores, one lane publishes an `sc1` flag; the consumers poll it with `sc1` loads, then a
//      workgroup barrier, then `sc1` loads of the payload; the way back is one agent-scope counter the producer polls).
// Both move the same bytes with the same two dependencies per step; the output is microseconds per step.  Every word is checked
// (a stale read shows up as a wrong sum), every spin is bounded.
// Build: hipcc --offload-arch=gfx950 -O3 tools/handoff_bench.hip -o tools/bin/handoff_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int PAYLOAD = 128 * 128;        // doubles = 128 KB
constexpr int THREADS = 256;
constexpr int PER_THREAD = PAYLOAD / THREADS;

__global__ __launch_bounds__(THREADS) void producer_kernel(double* buf, int step) {
  for (int i = 0; i < PER_THREAD; ++i) buf[i * THREADS + threadIdx.x] = (double)(step + i * THREADS + threadIdx.x);
}
__global__ __launch_bounds__(THREADS) void consumer_kernel(const double* buf, double* out, int step) {
  double s = 0.0;
  for (int i = 0; i < PER_THREAD; ++i) s += buf[i * THREADS + threadIdx.x] - (double)(step + i * THREADS + threadIdx.x);
  if (s != 0.0) out[0] = -1.0;            // stale or torn
  if (threadIdx.x == 0) out[1 + blockIdx.x] = (double)step;
}

__global__ __launch_bounds__(THREADS) void persistent_kernel(double* buf, int* flag, int* done, double* out, int steps, int nconsumers) {
  extern __shared__ char lds_pad[];       // (dynamic LDS only to keep the launch at one workgroup per CU)
  (void)lds_pad;
  const int tid = threadIdx.x;
  const int SPIN_MAX = 1 << 24;
  if (blockIdx.x == 0) {
    for (int step = 1; step <= steps; ++step) {
      for (int i = 0; i < PER_THREAD; ++i)
        __hip_atomic_store(&buf[i * THREADS + tid], (double)(step + i * THREADS + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_store(flag, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < step * nconsumers) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > SPIN_MAX) { out[0] = -2.0; break; }
        }
      }
      __syncthreads();
    }
    return;
  }
  for (int step = 1; step <= steps; ++step) {
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < step) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_MAX) { out[0] = -3.0; break; }
      }
    }
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < PER_THREAD; ++i)
      s += __hip_atomic_load(&buf[i * THREADS + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (double)(step + i * THREADS + tid);
    if (s != 0.0) out[0] = -1.0;
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); return 1; } } while (0)

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 2000;
  const int nconsumers = 255;
  double *buf, *out;
  int *flag, *done;
  CK(hipMalloc(&buf, PAYLOAD * sizeof(double)));
  CK(hipMalloc(&out, (2 + nconsumers) * sizeof(double)));
  CK(hipMalloc(&flag, 256));
  CK(hipMalloc(&done, 256));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  double host[2];
  // A: two dependent launches per step
  CK(hipMemsetAsync(out, 0, (2 + nconsumers) * sizeof(double), s));
  for (int warm = 0; warm < 2; ++warm) {
    CK(hipEventRecord(e0, s));
    for (int step = 1; step <= steps; ++step) {
      hipLaunchKernelGGL(producer_kernel, dim3(1), dim3(THREADS), 0, s, buf, step);
      hipLaunchKernelGGL(consumer_kernel, dim3(nconsumers), dim3(THREADS), 0, s, buf, out, step);
    }
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
  }
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(host, out, sizeof(host), hipMemcpyDeviceToHost));
  printf("A  kernel boundaries (producer 1 WG -> consumer %d WGs, 128 KB read by each): %.2f us per step%s\n", nconsumers, ms * 1e3 / steps,
         host[0] != 0.0 ? "  [CHECK FAILED]" : "");
  // B: one persistent launch, sc1 payload + flag, counter fan-in
  for (int warm = 0; warm < 2; ++warm) {
    CK(hipMemsetAsync(flag, 0, 256, s));
    CK(hipMemsetAsync(done, 0, 256, s));
    CK(hipMemsetAsync(out, 0, (2 + nconsumers) * sizeof(double), s));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(persistent_kernel, dim3(1 + nconsumers), dim3(THREADS), 96 * 1024, s, buf, flag, done, out, steps, nconsumers);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
  }
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(host, out, sizeof(host), hipMemcpyDeviceToHost));
  printf("B  persistent launch, sc1 hand-off (flag out, counter back), same bytes:        %.2f us per step%s\n", ms * 1e3 / steps,
         host[0] == 0.0 ? "" : (host[0] == -1.0 ? "  [STALE READ]" : "  [SPIN LIMIT]"));
  return 0;
}
