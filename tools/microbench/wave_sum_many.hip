// block_sum_many (permlane / DPP butterfly) against block_sum_many_trees (ds_bpermute trees): the same bits, and what each costs.
#include "../../trips_py_amd/csrc/trk_internal.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace trk;
template <int NV, bool X>
__global__ __launch_bounds__(256) void k(const double* __restrict__ in, double* __restrict__ out, int reps) {
  __shared__ double lds[4 * NV];
  double v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = in[((size_t)blockIdx.x * NV + i) * 256 + threadIdx.x];
  double t = 0.0;
  for (int r = 0; r < reps; ++r) {
    t = X ? block_sum_many<256, NV>(v, lds) : block_sum_many_trees<256, NV>(v, lds);
    if (reps > 1) {
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] += t * 1e-30;
    }
  }
  if (threadIdx.x < NV) out[(size_t)blockIdx.x * NV + threadIdx.x] = t;
}
template <bool X>
__global__ __launch_bounds__(256) void k1(const double* __restrict__ in, double* __restrict__ out) {
  __shared__ double lds[4];
  double v = in[(size_t)blockIdx.x * 256 + threadIdx.x];
  double t;
  if (X) t = block_sum<256>(v, lds);
  else {
    v = wave_sum_trees(v);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    t = 0.0;
    if (threadIdx.x == 0) for (int w = 0; w < 4; ++w) t += lds[w];
  }
  if (threadIdx.x == 0) out[blockIdx.x] = t;
}
int run1() {
  const int nb = 4096;
  std::vector<double> h((size_t)nb * 256);
  for (auto& x : h) x = (double)rand() / RAND_MAX - 0.5 + 1e-9 * rand();
  double *in, *o0, *o1;
  hipMalloc(&in, h.size() * 8); hipMalloc(&o0, nb * 8); hipMalloc(&o1, nb * 8);
  hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k1<false>), dim3(nb), dim3(256), 0, 0, in, o0);
  hipLaunchKernelGGL((k1<true>), dim3(nb), dim3(256), 0, 0, in, o1);
  std::vector<double> a(nb), b(nb);
  hipMemcpy(a.data(), o0, nb * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), o1, nb * 8, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (int i = 0; i < nb; ++i) bad += a[i] != b[i];
  printf("block_sum<256>: wave_sum (permlane / DPP) against wave_sum_trees: %zu of %d differ (sample %.17g vs %.17g)\n", bad, nb, a[7], b[7]);
  return bad ? 1 : 0;
}
template <int NV>
int run() {
  const int nb = 2048;
  std::vector<double> h((size_t)nb * NV * 256);
  for (auto& x : h) x = (double)rand() / RAND_MAX - 0.5 + 1e-9 * rand();
  double *in, *o0, *o1;
  hipMalloc(&in, h.size() * 8); hipMalloc(&o0, (size_t)nb * NV * 8); hipMalloc(&o1, (size_t)nb * NV * 8);
  hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k<NV, false>), dim3(nb), dim3(256), 0, 0, in, o0, 1);
  hipLaunchKernelGGL((k<NV, true>), dim3(nb), dim3(256), 0, 0, in, o1, 1);
  std::vector<double> a((size_t)nb * NV), b((size_t)nb * NV);
  hipMemcpy(a.data(), o0, a.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), o1, b.size() * 8, hipMemcpyDeviceToHost);
  const int same = memcmp(a.data(), b.data(), a.size() * 8) == 0;
  size_t bad = 0;
  for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[2];
  for (int x = 0; x < 2; ++x) {
    hipEventRecord(e0);
    if (x) hipLaunchKernelGGL((k<NV, true>), dim3(nb), dim3(256), 0, 0, in, o1, 200);
    else hipLaunchKernelGGL((k<NV, false>), dim3(nb), dim3(256), 0, 0, in, o0, 200);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms[x], e0, e1);
  }
  printf("NV %2d: bit-identical %s (%zu of %zu differ; sample %.17g vs %.17g)  200 reductions x %d workgroups: trees %.3f ms, butterfly %.3f ms\n", NV,
         same ? "yes" : "NO", bad, a.size(), a[5], b[5], nb, ms[0], ms[1]);
  return same ? 0 : 1;
}
int main() { int rc = run1(); rc |= run<3>(); rc |= run<4>(); rc |= run<8>(); rc |= run<16>(); rc |= run<20>(); rc |= run<24>(); rc |= run<32>(); return rc; }
