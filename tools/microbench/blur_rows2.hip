// blur_rows2.hip — experiment (round 6): the separable 9 x 9 blur as a ROW-MARCHING kernel that fits two waves per SIMD.
//
// k_blur_slide (the product) runs ONE wave per SIMD in 231-258 registers (nine prefetched rows of three 16-byte loads each, nine
// rolling accumulator rows) and takes 24.0 us per 4096^2 matvec inside the CGLS loop where a plain copy of the same shape takes
// 20.4-21 us (profiles/r06/copy_sweep.txt): its ~13 us of vector work do not overlap with the memory time of its single wave.
// Here: ONE 16-byte load per lane and row (the neighbour columns by whole-wave DPP shifts, the four columns beyond either end of the
// 256-column span by one load that only lanes 0 and 63 execute), prefetch depth D, the weights in scalar registers: <= 128 registers,
// 8 waves per CU, a band of rows per wave sized so that the grid is ONE resident round.  Question: how far below 24 us does it get
// from cold operands?
//
// Build: hipcc --offload-arch=gfx950 -O3 blur_rows2.hip -o blur_rows2 ; run: ./blur_rows2 [N=4096] [sets=6]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e)                                                                          \
  do {                                                                                 \
    hipError_t r_ = (e);                                                               \
    if (r_ != hipSuccess) {                                                            \
      printf("%s -> %s (%d)\n", #e, hipGetErrorString(r_), __LINE__);                   \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int KH = 9, KW = 9, T = 4, L = 4;

__host__ __device__ constexpr int gcd_c(int a, int b) { return b == 0 ? a : gcd_c(b, a % b); }
__host__ __device__ constexpr int lcm_c(int a, int b) { return a / gcd_c(a, b) * b; }
__device__ __forceinline__ int reflect(int i, int n) {
  if ((unsigned)i < (unsigned)n) return i;
  const int p = 2 * n;
  i %= p;
  if (i < 0) i += p;
  return (i >= n) ? (p - 1 - i) : i;
}
// whole-wave shifts (gfx950: wave_shr / wave_shl act across all 64 lanes); the lane without a source keeps `old`
__device__ __forceinline__ float from_left(float old, float v) {   // lane l gets lane l - 1's v; lane 0 keeps old
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float from_right(float old, float v) {  // lane l gets lane l + 1's v; lane 63 keeps old
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
}

// One wave = a 256-column span x `rows_per_band` output rows, marching down; workgroup = 4 waves = 4 neighbouring spans of one band.
template <int D, bool SUMSQ>
__global__ __launch_bounds__(256, 2) void k_blur_rows2(const float* __restrict__ x, float* __restrict__ y, int nx, int ny,
                                                       const float* __restrict__ wts, int spans_x, int nbands, int rows_per_band,
                                                       double* __restrict__ partials, int nt_store) {
  __shared__ double red[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // XCD-aware placement of the (band, span group) tiles: XCD x takes a contiguous run of the row-major order
  const int groups_x = spans_x / 4;
  int band, sg;
  {
    const int id = blockIdx.x, total = nbands * groups_x;
    const int xcd = id & 7, local = id >> 3;
    const int t = xcd * (total >> 3) + min(xcd, total & 7) + local;
    band = t / groups_x;
    sg = t - band * groups_x;
  }
  const int span = sg * 4 + wave;
  const int i_begin = band * rows_per_band;
  const int i_end = min(i_begin + rows_per_band, nx);
  const int c0 = span * 256 + 4 * lane;
  const bool first_span = span == 0, last_span = span == spans_x - 1;
  const unsigned img_bytes = (unsigned)nx * (unsigned)ny * 4u;
  const auto rin = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, img_bytes, 0x00020000);
  const auto rout = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, img_bytes, 0x00020000);
  const int vc = c0 * 4;
  // the group beyond the span: lane 0 the four columns left of it, lane 63 the four right of it (inside the image; the image's own
  // borders are reflected from the lane's own columns below)
  const bool edge_lane = (lane == 0 && !first_span) || (lane == 63 && !last_span);
  const int ve = lane == 0 ? vc - 16 : vc + 16;
  float wr[KW], wc[KH];
#pragma unroll
  for (int b = 0; b < KW; ++b) wr[b] = wts[b];
#pragma unroll
  for (int a = 0; a < KH; ++a) wc[a] = wts[KW + a];
  const int band_rows = i_end - i_begin;
  const int total = band_rows + KH - 1;                 // staged rows: image rows i_begin - T .. i_end - 1 + (KH - 1 - T)
  const int rowbytes = ny * 4;
  const int first = i_begin - T;

  const bool interior = first >= 0 && first + total - 1 < nx;      // wave-uniform: no row reflection in this band
  f4 pC[D], pE[D];
  auto issue = [&](int t, int slot) {
    const int gi = interior ? first + t : reflect(first + t, nx);
    const int so = gi * rowbytes;
    pC[slot] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rin, vc, so, 0));
    if (edge_lane) pE[slot] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rin, ve, so, 0));
  };
#pragma unroll
  for (int d = 0; d < D; ++d) {
    pE[d] = (f4){0.f, 0.f, 0.f, 0.f};
    if (d < total) issue(d, d);
  }
  f2 acc[KH][2];
#pragma unroll
  for (int a = 0; a < KH; ++a) acc[a][0] = acc[a][1] = (f2){0.f, 0.f};
  double ss = 0.0;
  constexpr int U = lcm_c(KH, D);
  for (int t0 = 0; t0 < total; t0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u;
      if (t < total) {                                   // (wave-uniform; no `break`: the row loop must unroll or the rings go to scratch)
      const int slot = u % D;
      const f4 Cv = pC[slot], Ev = pE[slot];
      if (t + D < total) issue(t + D, slot);
      // neighbour groups: lane l - 1's / l + 1's columns; lanes 0 / 63 keep what they fetched, or the reflection of their own columns
      const f4 rev = (f4){Cv[3], Cv[2], Cv[1], Cv[0]};
      const f4 eL = first_span ? rev : Ev, eR = last_span ? rev : Ev;
      f4 Lv, Rv;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        Lv[k] = from_left(eL[k], Cv[k]);
        Rv[k] = from_right(eR[k], Cv[k]);
      }
      const float v[12] = {Lv[0], Lv[1], Lv[2], Lv[3], Cv[0], Cv[1], Cv[2], Cv[3], Rv[0], Rv[1], Rv[2], Rv[3]};
      float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f;
#pragma unroll
      for (int b = 0; b < KW; ++b) {
        h0 = fmaf(wr[b], v[b], h0);
        h1 = fmaf(wr[b], v[b + 1], h1);
        h2 = fmaf(wr[b], v[b + 2], h2);
        h3 = fmaf(wr[b], v[b + 3], h3);
      }
      const f2 hlo = {h0, h1}, hhi = {h2, h3};
      // staged row t contributes to outputs o = t - a (a = 0 .. KH - 1) with wc[a]; accumulator of output o sits in slot o % KH
#pragma unroll
      for (int a = 0; a < KH; ++a) {
        const int o_slot = ((u - a) % KH + KH) % KH;       // (t0 is a multiple of U, U a multiple of KH)
        const f2 w2 = {wc[a], wc[a]};
        acc[o_slot][0] = __builtin_elementwise_fma(w2, hlo, acc[o_slot][0]);
        acc[o_slot][1] = __builtin_elementwise_fma(w2, hhi, acc[o_slot][1]);
      }
      // output o = t - (KH - 1) is complete
      const int o = t - (KH - 1);
      const int os = ((u - (KH - 1)) % KH + KH) % KH;
      if (o >= 0) {
        const f4 out = {acc[os][0][0], acc[os][0][1], acc[os][1][0], acc[os][1][1]};
        const int off = vc + (i_begin + o) * rowbytes;
        if (nt_store) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, out), rout, off, 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, out), rout, off, 0, 0);
        if (SUMSQ) {
          const float q = fmaf(out[0], out[0], fmaf(out[1], out[1], fmaf(out[2], out[2], out[3] * out[3])));
          ss += (double)q;
        }
      }
      acc[os][0] = acc[os][1] = (f2){0.f, 0.f};
      }
    }
  }
  if (SUMSQ) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

__global__ void k_ref(const float* x, float* y, int nx, int ny, const float* wts) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)nx * ny) return;
  const int i = (int)(idx / ny), j = (int)(idx % ny);
  float acc = 0.f;
  for (int a = 0; a < KH; ++a) {
    const int gi = reflect(i - T + a, nx);
    float h = 0.f;
    for (int b = 0; b < KW; ++b) h = fmaf(wts[b], x[(int64_t)gi * ny + reflect(j - L + b, ny)], h);
    acc = fmaf(wts[KW + a], h, acc);
  }
  y[idx] = acc;
}

__global__ void k_copy_nt(const f4* __restrict__ x, f4* __restrict__ y, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) __builtin_nontemporal_store(x[i], y + i);
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 4096;
  const int NB = argc > 2 ? atoi(argv[2]) : 6;
  const int64_t n = (int64_t)N * N;
  std::vector<float> hx(n), w(KW + KH);
  for (int64_t i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  double s = 0;
  for (int k = 0; k < 9; ++k) {
    w[k] = expf(-0.5f * (k - 4) * (k - 4) / 9.f);
    s += w[k];
  }
  for (int k = 0; k < 9; ++k) w[k] = w[KW + k] = (float)(w[k] / s);
  float *dw, *dref;
  std::vector<float*> X(NB), Y(NB);
  CK(hipMalloc(&dw, sizeof(float) * w.size()));
  CK(hipMemcpy(dw, w.data(), sizeof(float) * w.size(), hipMemcpyHostToDevice));
  for (int b = 0; b < NB; ++b) {
    CK(hipMalloc(&X[b], sizeof(float) * n));
    CK(hipMalloc(&Y[b], sizeof(float) * n));
    CK(hipMemcpy(X[b], hx.data(), sizeof(float) * n, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&dref, sizeof(float) * n));
  double* part;
  CK(hipMalloc(&part, sizeof(double) * 65536));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int spans = N / 256;
  if (N % 1024) {
    printf("N must be a multiple of 1024\n");
    return 1;
  }
  hipLaunchKernelGGL(k_ref, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, X[0], dref, N, N, dw);
  CK(hipDeviceSynchronize());
  std::vector<float> a(n), b(n);
  CK(hipMemcpy(a.data(), dref, sizeof(float) * n, hipMemcpyDeviceToHost));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 60;
  auto check = [&](const char* tag) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(b.data(), Y[0], sizeof(float) * n, hipMemcpyDeviceToHost));
    double md = 0;
    int64_t nbad = 0;
    for (int64_t i = 0; i < n; ++i) {
      const double d = fabs((double)a[i] - b[i]);
      if (d > md) md = d;
      if (d > 1e-6) ++nbad;
    }
    printf("%s: max |kernel - reference| = %.3e (%lld pixels above 1e-6)\n", tag, md, (long long)nbad);
  };
  // rows per band: one resident round of 8 waves per CU (2 workgroups of four), or the given multiples of it
  for (int rounds : {1, 2, 4}) {
    const int waves = cus * 8 * rounds;
    int rpb = (int)(((int64_t)N * spans + waves - 1) / waves);
    const int nbands = (N + rpb - 1) / rpb;
    const int grid = nbands * (spans / 4);
#define RUN(DD)                                                                                                                         \
  do {                                                                                                                                  \
    hipLaunchKernelGGL((k_blur_rows2<DD, true>), dim3(grid), dim3(256), 0, 0, X[0], Y[0], N, N, dw, spans, nbands, rpb, part, 1);        \
    char tag[96];                                                                                                                       \
    snprintf(tag, sizeof tag, "rows/band %d (%d bands, grid %d) D=%d", rpb, nbands, grid, DD);                                          \
    check(tag);                                                                                                                         \
    for (int ss = 0; ss < 2; ++ss) {                                                                                                    \
      for (int r = 0; r < 6; ++r)                                                                                                       \
        hipLaunchKernelGGL((k_blur_rows2<DD, false>), dim3(grid), dim3(256), 0, 0, X[r % NB], Y[r % NB], N, N, dw, spans, nbands, rpb, part, 1); \
      CK(hipEventRecord(e0));                                                                                                           \
      for (int r = 0; r < reps; ++r) {                                                                                                  \
        if (ss) hipLaunchKernelGGL((k_blur_rows2<DD, true>), dim3(grid), dim3(256), 0, 0, X[r % NB], Y[r % NB], N, N, dw, spans, nbands, rpb, part, 1); \
        else hipLaunchKernelGGL((k_blur_rows2<DD, false>), dim3(grid), dim3(256), 0, 0, X[r % NB], Y[r % NB], N, N, dw, spans, nbands, rpb, part, 1); \
      }                                                                                                                                 \
      CK(hipEventRecord(e1));                                                                                                           \
      CK(hipEventSynchronize(e1));                                                                                                      \
      float ms;                                                                                                                         \
      CK(hipEventElapsedTime(&ms, e0, e1));                                                                                             \
      const double us = ms * 1e3 / reps;                                                                                                \
      printf("  rows2 D=%d rows/band=%d sumsq=%d: %.2f us per launch over %d rotating pairs = %.2f TB/s (%.3f of 8 TB/s)\n", DD, rpb, ss, us, \
             NB, 8.0 * n / us * 1e-6, 8.0 * n / us * 1e-6 / 8.0);                                                                       \
    }                                                                                                                                   \
  } while (0)
    RUN(3);
    RUN(6);
    RUN(9);
  }
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_copy_nt, dim3(cus * 4), dim3(256), 0, 0, (const f4*)X[r % NB], (f4*)Y[r % NB], n / 4);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("copy (plain loads, nt stores, 16 waves per CU): %.2f us = %.2f TB/s\n", ms * 1e3 / reps, 8.0 * n / (ms * 1e3 / reps) * 1e-6);
  return 0;
}
