// blur_tile.hip — experiment: the separable 9 x 9 blur as a TILE kernel whose waves share the halo rows of
// the vertical pass through LDS, so that every wave can be short (8 output rows) and many (8 waves per workgroup, 16 per CU) with
// all of its row loads in flight at once — the shape in which a plain copy streams at 6.7 TB/s (tools/experiments/copy_patterns.hip)
// — without re-reading KH - 1 halo rows per wave from memory.
//
//   workgroup = 8 waves x 64 lanes, tile = 64 output rows x 256 columns (4 per lane); wave w loads image rows 8 w .. 8 w + 7 of the
//   tile and ONE of the tile's 8 halo rows (4 above, 4 below), filters them horizontally in registers (neighbour columns by DPP wave
//   shifts, the two columns beyond the span by one predicated load) and parks the 9 filtered rows in LDS (72 rows x 1 KB); after ONE
//   barrier it reads the 16 filtered rows its 8 outputs need and runs the vertical pass with rolling accumulators.
//
// Build: hipcc --offload-arch=gfx950 -O3 blur_tile.hip -o blur_tile ; run: ./blur_tile [N=4096] [buffers=6]
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
constexpr int TR = 64, TC = 256, WAVES = 8, RW = TR / WAVES;   // tile rows / columns, waves, rows per wave
constexpr int HR = TR + KH - 1;                                  // filtered rows in LDS

__device__ __forceinline__ int reflect(int i, int n) {
  if ((unsigned)i < (unsigned)n) return i;
  const int p = 2 * n;
  i %= p;
  if (i < 0) i += p;
  return (i >= n) ? (p - 1 - i) : i;
}

// whole-wave shifts (gfx950: wave_shr / wave_shl act across all 64 lanes — tools/experiments/dpp_wave_shift.hip)
__device__ __forceinline__ float from_left(float v) {   // lane l gets lane l - 1's value
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float from_right(float v) {  // lane l gets lane l + 1's value
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
}

template <bool SUMSQ>
__global__ __launch_bounds__(64 * WAVES, 2) void k_blur_tile(const float* __restrict__ x, float* __restrict__ y, int nx, int ny,
                                                            const float* __restrict__ wts, int spans_x, int ntile_rows,
                                                            double* __restrict__ partials, int nt_store) {
  __shared__ f4 H[HR][64];
  __shared__ double red[WAVES];
  // XCD-aware placement: ids are dealt round-robin over 8 XCDs; each XCD takes a contiguous run of the row-major (tile row, span) order
  int trow, span;
  {
    const int id = blockIdx.x, total = ntile_rows * spans_x;
    const int xcd = id & 7, local = id >> 3;
    const int t = xcd * (total >> 3) + min(xcd, total & 7) + local;
    trow = t / spans_x;
    span = t - trow * spans_x;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i0 = trow * TR;
  const int c0 = span * TC + 4 * lane;
  const unsigned img_bytes = (unsigned)nx * (unsigned)ny * 4u;
  const auto rin = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, img_bytes, 0x00020000);
  const auto rout = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, img_bytes, 0x00020000);
  const int rowbytes = ny * 4;
  const bool ledge = (c0 == 0), redge = (c0 + 4 == ny);
  // the group beyond the span: lane 0 fetches the 4 columns left of it, lane 63 the 4 right of it (one predicated load per row)
  const bool has_edge = (lane == 0 && !ledge) || (lane == 63 && !redge);
  const int ve = (lane == 0 ? c0 - 4 : c0 + 4) * 4;
  float wr[KW], wc[KH];
#pragma unroll
  for (int b = 0; b < KW; ++b) wr[b] = wts[b];
#pragma unroll
  for (int a = 0; a < KH; ++a) wc[a] = wts[KW + a];

  // ---- phase 1: this wave's RW + 1 rows, all loads in flight together; LDS row index = image row - (i0 - T)
  // rows: own rows 8 w .. 8 w + 7 -> LDS rows T + 8 w + j ; halo row: wave w < 4: LDS row w (above), else LDS row TR + w (below: T + TR + (w - 4))
  f4 C[RW + 1], E[RW + 1];
  int hrow[RW + 1];
#pragma unroll
  for (int j = 0; j <= RW; ++j) {
    hrow[j] = j < RW ? T + RW * wave + j : (wave < 4 ? wave : TR + wave);
    const int gi = reflect(i0 - T + hrow[j], nx);
    C[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rin, c0 * 4, gi * rowbytes, 0));
    E[j] = (f4){0.f, 0.f, 0.f, 0.f};
    if (has_edge) E[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rin, ve, gi * rowbytes, 0));
  }
#pragma unroll
  for (int j = 0; j <= RW; ++j) {
    const f4 Cv = C[j];
    f4 Lv, Rv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      Lv[e] = from_left(Cv[e]);
      Rv[e] = from_right(Cv[e]);
    }
    const f4 rev = (f4){Cv[3], Cv[2], Cv[1], Cv[0]};
    if (lane == 0) Lv = ledge ? rev : E[j];
    if (lane == 63) Rv = redge ? rev : E[j];
    const float v[12] = {Lv[0], Lv[1], Lv[2], Lv[3], Cv[0], Cv[1], Cv[2], Cv[3], Rv[0], Rv[1], Rv[2], Rv[3]};
    float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f;
#pragma unroll
    for (int b = 0; b < KW; ++b) {
      h0 = fmaf(wr[b], v[b], h0);
      h1 = fmaf(wr[b], v[b + 1], h1);
      h2 = fmaf(wr[b], v[b + 2], h2);
      h3 = fmaf(wr[b], v[b + 3], h3);
    }
    H[hrow[j]][lane] = (f4){h0, h1, h2, h3};
  }
  __syncthreads();
  // ---- phase 2: outputs i0 + 8 w + o, o < 8, from filtered rows (LDS rows) 8 w + o .. 8 w + o + 8
  f2 acc[RW][2];
#pragma unroll
  for (int o = 0; o < RW; ++o) acc[o][0] = acc[o][1] = (f2){0.f, 0.f};
#pragma unroll
  for (int t = 0; t < RW + KH - 1; ++t) {
    const f4 h = H[RW * wave + t][lane];
    const f2 hlo = {h[0], h[1]}, hhi = {h[2], h[3]};
#pragma unroll
    for (int o = 0; o < RW; ++o) {
      const int a = t - o;
      if (a >= 0 && a < KH) {
        acc[o][0] = wc[a] * hlo + acc[o][0];
        acc[o][1] = wc[a] * hhi + acc[o][1];
      }
    }
  }
  double ss = 0.0;
  float q = 0.f;
#pragma unroll
  for (int o = 0; o < RW; ++o) {
    const int gi = i0 + RW * wave + o;
    const f4 out = (f4){acc[o][0][0], acc[o][0][1], acc[o][1][0], acc[o][1][1]};
    if (gi < nx) {
      if (nt_store)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, out), rout, c0 * 4 + gi * rowbytes, 0, 2);
      else
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, out), rout, c0 * 4 + gi * rowbytes, 0, 0);
      if (SUMSQ) q = fmaf(out[0], out[0], fmaf(out[1], out[1], fmaf(out[2], out[2], fmaf(out[3], out[3], q))));
    }
    if (SUMSQ && (o & 3) == 3) {
      ss += (double)q;
      q = 0.f;
    }
  }
  if (SUMSQ) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int w = 0; w < WAVES; ++w) t += red[w];
      partials[blockIdx.x] = t;
    }
  }
}

__global__ void k_ref(const float* x, float* y, int nx, int ny, const float* wts) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)nx * ny) return;
  const int i = (int)(idx / ny), j = (int)(idx % ny);
  float acc = 0.f;
  // the same order of operations as the tile kernel: horizontal sums per row (taps ascending), then the vertical taps ascending
  for (int a = 0; a < KH; ++a) {
    const int gi = reflect(i - T + a, nx);
    float h = 0.f;
    for (int b = 0; b < KW; ++b) h = fmaf(wts[b], x[(int64_t)gi * ny + reflect(j - L + b, ny)], h);
    acc = a == 0 ? wts[KW + a] * h : fmaf(wts[KW + a], h, acc);
  }
  y[idx] = acc;
}

__global__ void k_copy(const f4* __restrict__ x, f4* __restrict__ y, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) y[i] = x[i];
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
  const int spans = N / TC, trows = (N + TR - 1) / TR, grid = spans * trows;
  if (N % TC) {
    printf("N must be a multiple of %d\n", TC);
    return 1;
  }
  hipLaunchKernelGGL(k_ref, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, X[0], dref, N, N, dw);
  hipLaunchKernelGGL(k_blur_tile<true>, dim3(grid), dim3(64 * WAVES), 0, 0, X[0], Y[0], N, N, dw, spans, trows, part, 0);
  CK(hipDeviceSynchronize());
  std::vector<float> a(n), b(n);
  std::vector<double> hp(grid);
  CK(hipMemcpy(a.data(), dref, sizeof(float) * n, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), Y[0], sizeof(float) * n, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hp.data(), part, sizeof(double) * grid, hipMemcpyDeviceToHost));
  double md = 0, sq = 0, ps = 0;
  int64_t nbad = 0;
  for (int64_t i = 0; i < n; ++i) {
    const double d = fabs((double)a[i] - b[i]);
    if (d > md) md = d;
    if (d > 1e-6) ++nbad;
    sq += (double)b[i] * b[i];
  }
  for (int g = 0; g < grid; ++g) ps += hp[g];
  printf("N=%d: max |tile - reference| = %.3e (%lld pixels above 1e-6); sum of squares %.10e vs partials %.10e\n", N, md, (long long)nbad, sq, ps);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 60;
  for (int nt = 0; nt < 2; ++nt)
    for (int ss = 0; ss < 2; ++ss) {
      for (int r = 0; r < 6; ++r) hipLaunchKernelGGL(k_blur_tile<false>, dim3(grid), dim3(64 * WAVES), 0, 0, X[r % NB], Y[r % NB], N, N, dw, spans, trows, part, nt);
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) {
        if (ss)
          hipLaunchKernelGGL(k_blur_tile<true>, dim3(grid), dim3(64 * WAVES), 0, 0, X[r % NB], Y[r % NB], N, N, dw, spans, trows, part, nt);
        else
          hipLaunchKernelGGL(k_blur_tile<false>, dim3(grid), dim3(64 * WAVES), 0, 0, X[r % NB], Y[r % NB], N, N, dw, spans, trows, part, nt);
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps;
      printf("tile kernel  nt_store=%d sumsq=%d: %.2f us per launch over %d rotating buffer pairs = %.2f TB/s (%.3f of 8 TB/s)\n", nt, ss, us, NB,
             8.0 * n / us * 1e-6, 8.0 * n / us * 1e-6 / 8.0);
    }
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, 0, (const f4*)X[r % NB], (f4*)Y[r % NB], n / 4);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("plain copy: %.2f us = %.2f TB/s\n", ms * 1e3 / reps, 8.0 * n / (ms * 1e3 / reps) * 1e-6);
  return 0;
}
