// dispatch_floor.hip — what does a DEPENDENT kernel boundary cost on this box, and what does that leave for the body of a kernel in a
// two-launch CGLS iteration at 512^2 (C2: 256 workgroups of 512 threads, one per CU)?
// A chain of K launches on one stream, each workgroup spinning for `body` shader cycles (s_memtime) before it exits: the time per
// launch minus the body is the boundary (drain + dispatch + ramp).  Also: the same body in ONE launch that loops K times with a
// grid-wide barrier between repetitions (one monotonic counter, agent-scope atomic add + sc1-load poll by one lane per workgroup —
// the simplest legal form, MI355X_MICROARCH.md "barrier-counter"), i.e. what a persistent iteration would pay per synchronisation point.
//   build: hipcc --offload-arch=gfx950 -O3 dispatch_floor.hip -o dispatch_floor
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__device__ __forceinline__ void spin(long long cycles) {
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(1);
}

__global__ __launch_bounds__(512) void k_body(float* sink, long long body) {
  spin(body);
  if (body < 0) sink[threadIdx.x] = 1.f;
}

__global__ __launch_bounds__(512) void k_persistent(float* sink, long long body, int reps, unsigned* counter) {
  for (int r = 0; r < reps; ++r) {
    spin(body);
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(r + 1) * gridDim.x;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
  if (body < 0) sink[threadIdx.x] = 1.f;
}

int main() {
  float* sink;
  unsigned* counter;
  CK(hipMalloc(&sink, 4096));
  CK(hipMalloc(&counter, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int K = 2000, G = 256;
  printf("# %d workgroups x 512 threads; K = %d launches / repetitions; body in shader cycles (2.4 GHz: 2400 = 1 us)\n", G, K);
  printf("%10s %22s %22s %26s\n", "body", "chain: us per launch", "boundary = that - body", "persistent: us per rep (- body)");
  for (long long body : {0LL, 2400LL, 7200LL, 12000LL}) {
    for (int w = 0; w < 50; ++w) hipLaunchKernelGGL(k_body, dim3(G), dim3(512), 0, 0, sink, body);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k_body, dim3(G), dim3(512), 0, 0, sink, body);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double per = ms * 1e3 / K, b_us = body / 2400.0;
    CK(hipMemset(counter, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_persistent, dim3(G), dim3(512), 0, 0, sink, body, K, counter);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms2;
    CK(hipEventElapsedTime(&ms2, e0, e1));
    const double per2 = ms2 * 1e3 / K;
    printf("%10lld %22.2f %22.2f %16.2f (%.2f)\n", body, per, per - b_us, per2, per2 - b_us);
  }
  return 0;
}
