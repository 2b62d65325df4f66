// wg_census.hip — where do the workgroups of a SMALL grid land, and when do they start?
//
// The projector kernels at 512^2 (1 024 … 1 440 workgroups of 256 threads: four to six per CU if dealt evenly) run at 40–55 % of the
// vector-issue time their own counters add up to (SQ_ACTIVE_INST_VALU against GRBM_GUI_ACTIVE, profiles/r04/radon_512_adj_pmc.txt),
// and a wave's average lifetime (SQ_WAVE_CYCLES / SQ_WAVES) is 58 % of the kernel's.  Either the dispatcher deals unevenly (a CU that
// holds 7 workgroups while another holds 2 finishes late) or it deals slowly (the last workgroup starts late).  This program
// launches G workgroups of 256 threads that each issue a FIXED number of vector instructions (so contention on a CU shows as a longer
// workgroup) and records, per workgroup: XCC, SE, CU, start and end on the constant 100 MHz clock (s_memrealtime).
//   build: hipcc --offload-arch=gfx950 -O3 wg_census.hip -o wg_census        run: ./wg_census [trips] [lds_bytes]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

struct Rec {
  unsigned xcc, hwid;
  unsigned long long r0, r1, c0, c1;
};

__global__ __launch_bounds__(256) void k_census(Rec* __restrict__ rec, float* __restrict__ sink, int trips) {
  extern __shared__ float dyn[];
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (float)(threadIdx.x + i);
  float m = 1.0000001f, b = 1e-9f;
  asm volatile("" : "+v"(m), "+v"(b));
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = fmaf(a[i], m, b);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.678f) sink[threadIdx.x] = s + dyn[0];
  __syncthreads();
  if (threadIdx.x == 0) {
    Rec r;
    r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    r.r0 = r0;
    r.c0 = c0;
    r.r1 = __builtin_amdgcn_s_memrealtime();
    r.c1 = __builtin_amdgcn_s_memtime();
    rec[blockIdx.x] = r;
  }
}

int main(int argc, char** argv) {
  const int trips = argc > 1 ? atoi(argv[1]) : 300;       // 32 VALU per trip and wave
  const int lds = argc > 2 ? atoi(argv[2]) : 0;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# device %s, %d CUs; %d trips x 32 v_fma per wave, %d bytes of dynamic LDS per workgroup\n", prop.gcnArchName, prop.multiProcessorCount, trips, lds);
  Rec* rec;
  float* sink;
  const int GMAX = 8192;
  CK(hipMalloc(&rec, sizeof(Rec) * GMAX));
  CK(hipMalloc(&sink, 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int Gs[] = {256, 512, 528, 1024, 1440, 2048, 4096};
  printf("%6s %9s | %28s | %34s | %s\n", "WGs", "wall us", "start spread us (p50 p90 max)", "WG duration us (min p50 max; alone)", "CUs holding k workgroups over the run  k:count");
  for (int G : Gs) {
    float alone_us = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(k_census, dim3(rep == 0 ? 1 : G), dim3(256), lds, 0, rec, sink, trips);
      CK(hipDeviceSynchronize());
      if (rep == 0) {
        Rec r;
        CK(hipMemcpy(&r, rec, sizeof(Rec), hipMemcpyDeviceToHost));
        alone_us = (float)(r.r1 - r.r0) * 0.01f;
      }
    }
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_census, dim3(G), dim3(256), lds, 0, rec, sink, trips);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Rec> h(G);
    CK(hipMemcpy(h.data(), rec, sizeof(Rec) * G, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (auto& r : h) t0 = std::min(t0, r.r0);
    std::vector<float> st, du;
    std::map<unsigned, int> per_cu;
    for (auto& r : h) {
      st.push_back((float)(r.r0 - t0) * 0.01f);
      du.push_back((float)(r.r1 - r.r0) * 0.01f);
      const unsigned cu = (r.hwid >> 8) & 15, sh = (r.hwid >> 12) & 1, se = (r.hwid >> 13) & 7;
      per_cu[(r.xcc & 15) << 16 | se << 8 | sh << 4 | cu]++;
    }
    std::sort(st.begin(), st.end());
    std::sort(du.begin(), du.end());
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second]++;
    hist[0] = prop.multiProcessorCount - (int)per_cu.size();
    printf("%6d %9.1f | %8.2f %8.2f %8.2f   | %8.2f %8.2f %8.2f ; %8.2f | ", G, ms * 1e3f, st[G / 2], st[G * 9 / 10], st[G - 1], du[0], du[G / 2], du[G - 1], alone_us);
    for (auto& kv : hist)
      if (kv.second) printf(" %d:%d", kv.first, kv.second);
    printf("\n");
  }
  return 0;
}
