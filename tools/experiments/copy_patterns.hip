// Probe: how much HBM bandwidth do different work decompositions of a plain 4096 x 4096 fp32 copy reach on MI355X?
// (What bounds the blur kernel's 5.5 TB/s: its arithmetic, or the shape in which its waves walk through memory?)
//   flat      : thread i copies float4 i, grid-stride
//   band RxC  : one wave (64 lanes x float4 = 256 columns) marches R rows of a band top to bottom, D loads in flight
//   band2     : the same, but odd bands march upward (the blur kernel's trick)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_flat(const f4* __restrict__ x, f4* __restrict__ y, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) y[i] = x[i];
}

template <int D, bool UPDOWN, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_band(const float* __restrict__ x, float* __restrict__ y, int N, int rows, int spans) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int id = blockIdx.x * WAVES + wave;
  const int band = id / spans, span = id % spans;
  const int r0 = band * rows;
  const bool up = UPDOWN && (band & 1);
  const size_t col = (size_t)span * 256 + 4 * lane;
  f4 buf[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const int r = up ? r0 + rows - 1 - d : r0 + d;
    buf[d] = *reinterpret_cast<const f4*>(x + (size_t)r * N + col);
  }
  for (int t = 0; t < rows; t += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int tt = t + d;
      const int r = up ? r0 + rows - 1 - tt : r0 + tt;
      const f4 v = buf[d];
      if (tt + D < rows) {
        const int rn = up ? r - D : r + D;
        buf[d] = *reinterpret_cast<const f4*>(x + (size_t)rn * N + col);
      }
      *reinterpret_cast<f4*>(y + (size_t)r * N + col) = v;
    }
  }
}

// Round 3: is the band pattern's deficit HBM CHANNEL ALIASING?  64-row bands of a 4096-float pitch start exactly 1 MiB apart, and
// all of them read their row t at the same time.  Two ways to break that without changing the work: (a) a padded pitch (the image
// itself cannot have one — contiguous vectors are the ABI — but the probe can); (b) STAGGER: band b starts its march `b * stag`
// rows into the band and wraps (a copy can; the blur's sliding window could only with re-priming).
template <int D>
__global__ __launch_bounds__(64) void k_band_stag(const float* __restrict__ x, float* __restrict__ y, int pitch, int rows, int spans,
                                                  int stag) {
  const int lane = threadIdx.x & 63;
  const int id = blockIdx.x;
  const int band = id / spans, span = id % spans;
  const int r0 = band * rows;
  const int ph = (band * stag) % rows;
  const size_t col = (size_t)span * 256 + 4 * lane;
  f4 buf[D];
#pragma unroll
  for (int d = 0; d < D; ++d) buf[d] = *reinterpret_cast<const f4*>(x + (size_t)(r0 + (ph + d) % rows) * pitch + col);
  for (int t = 0; t < rows; t += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int tt = t + d;
      const int r = r0 + (ph + tt) % rows;
      const f4 v = buf[d];
      if (tt + D < rows) buf[d] = *reinterpret_cast<const f4*>(x + (size_t)(r0 + (ph + tt + D) % rows) * pitch + col);
      *reinterpret_cast<f4*>(y + (size_t)r * pitch + col) = v;
    }
  }
}

template <typename F>
float timeit(F f, int reps = 30) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps * 1e3f;
}

int main() {
  const int N = 4096;
  const size_t n = (size_t)N * N;
  float *x, *y;
  hipMalloc(&x, n * 4);
  hipMalloc(&y, n * 4);
  hipMemset(x, 1, n * 4);
  auto report = [&](const char* name, float us) { printf("%-34s %7.2f us  %5.2f TB/s\n", name, us, 8.0 * n / us / 1e6); };
  report("flat, 2048 blocks", timeit([&] { hipLaunchKernelGGL(k_flat, dim3(2048), dim3(256), 0, 0, (const f4*)x, (f4*)y, n / 4); }));
  report("flat, 16384 blocks", timeit([&] { hipLaunchKernelGGL(k_flat, dim3(16384), dim3(256), 0, 0, (const f4*)x, (f4*)y, n / 4); }));
#define BAND(D, UD, W, ROWS)                                                                                             \
  report("band " #ROWS " rows D=" #D " updown=" #UD " waves/blk=" #W, timeit([&] {                                       \
           hipLaunchKernelGGL((k_band<D, UD, W>), dim3((N / ROWS) * (N / 256) / W), dim3(64 * W), 0, 0, x, y, N, ROWS, N / 256); \
         }))
  BAND(6, false, 1, 72);    // not a divisor of 4096: skip rows at the end (fine for timing): use 64 below
  BAND(4, false, 1, 64);
  BAND(8, false, 1, 64);
  BAND(16, false, 1, 64);
  BAND(8, true, 1, 64);
  BAND(8, false, 4, 64);
  BAND(8, false, 1, 32);
  BAND(8, false, 1, 16);
  BAND(16, false, 1, 128);
  BAND(8, false, 4, 16);
  {
    float *xp, *yp;
    const int P = N + 64;                      // padded pitch: band starts 1 MiB + 16 KiB apart
    hipMalloc(&xp, (size_t)P * N * 4);
    hipMalloc(&yp, (size_t)P * N * 4);
    hipMemset(xp, 1, (size_t)P * N * 4);
#define STAG(D, PITCH, XX, YY, STG)                                                                                         \
  report("band 64 rows D=" #D " pitch=" #PITCH " stagger=" #STG, timeit([&] {                                              \
           hipLaunchKernelGGL((k_band_stag<D>), dim3((N / 64) * (N / 256)), dim3(64), 0, 0, XX, YY, PITCH, 64, N / 256, STG); \
         }))
    STAG(8, N, x, y, 0);
    STAG(8, N, x, y, 1);
    STAG(8, N, x, y, 7);
    STAG(8, N, x, y, 13);
    STAG(8, P, xp, yp, 0);
    STAG(8, P, xp, yp, 7);
    STAG(16, N, x, y, 0);
    STAG(16, N, x, y, 7);
    STAG(16, P, xp, yp, 0);
  }
  return 0;
}
