// vecops.hip — the vector algebra inside the Krylov loops, HBM-bound streaming kernels:
// 16-byte-per-lane coalesced loads, grid-stride, fp32 storage, fp64 accumulation, wave64 __shfl reductions,
// one double of block partial per workgroup, summed in a fixed order by core.hip's finalize kernel.
#include "trk_internal.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

using namespace trk;

namespace {

constexpr int NT = 256;

// grid for a streaming kernel over n floats: one float4 per thread until the chip is covered 4x (<= kMaxPartialBlocks
// blocks so a reduction leaves at most that many partials), then grid-stride

inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// Workgroups of `kernel` (NT threads, no dynamic LDS) that one CU holds at a time.
template <class K>
inline int resident_blocks_per_cu(K kernel) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, NT, 0) != hipSuccess || nb < 1) nb = 4;
  return nb;
}

// Grid of the one-pass k-dot kernels (k_gemv_t / _t2 / _tr): x = shares of the vector, y = row tiles.  All workgroups take the
// same time, so the grid is EXACTLY one resident round — occupancy x CUs workgroups in all, rounded DOWN to whole x-columns.
// Measured (4096^2, k = 18, three tiles): 683 x 3 = 2049 workgroups, one more than the chip holds, ran 290 us; 1024 x 3 (two
// rounds) 222 us; a single full round is what every basis size gets now (tools/gemv_micro.py, profiles/r03/gemv_micro.txt).
inline int tiled_dot_grid_x(int64_t n, int ntile, int blocks_per_cu) {
  static const int env = env_int("TRK_GEMVT_PER_CU", 0);
  const int64_t total = (int64_t)cu_count() * (env > 0 ? env : blocks_per_cu);
  int64_t bx = total / ntile;
  // float4s of a row per thread at least: two on short vectors (512^2: one float4 per thread and row left a workgroup little but its
  // reduction to do; Hybrid-GMRES 17.0 -> 17.4 k iterations/s, four: 17.0), TRK_GEMVT_F4 overrides
  static const int env_f4 = env_int("TRK_GEMVT_F4", 0);
  const int per_thread = env_f4 > 0 ? env_f4 : (n <= ((int64_t)1 << 20) ? 2 : 1);
  const int64_t chunk = (int64_t)NT * 4 * per_thread;
  const int64_t want = (n + chunk - 1) / chunk;
  if (bx > want) bx = want;
  if (bx > kMaxPartialBlocks) bx = kMaxPartialBlocks;
  return bx < 1 ? 1 : (int)bx;
}

inline int stream_grid(int64_t n) {
  int64_t want = (n + (int64_t)NT * 4 - 1) / ((int64_t)NT * 4);
  int64_t cap = (int64_t)cu_count() * 4;
  if (cap > kMaxPartialBlocks) cap = kMaxPartialBlocks;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return (int)want;
}

__device__ __forceinline__ float4 ld4(const float* p, int64_t i4) { return reinterpret_cast<const float4*>(p)[i4]; }
__device__ __forceinline__ void st4(float* p, int64_t i4, float4 v) { reinterpret_cast<float4*>(p)[i4] = v; }
// non-temporal accesses for vectors that are not re-read soon (which ones and from which size: stream_nontemporal())
typedef float f4nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_nt(const float* p, int64_t i4) {
  const f4nt v = __builtin_nontemporal_load(reinterpret_cast<const f4nt*>(p) + i4);
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st4_nt(float* p, int64_t i4, float4 v) {
  __builtin_nontemporal_store((f4nt){v.x, v.y, v.z, v.w}, reinterpret_cast<f4nt*>(p) + i4);
}

// ------------------------------------------------------------------ dot / nrm2 / diff-nrm2
// MODE 0: sum x*y   1: sum x*x   2: sum (x-y)^2
template <int MODE, bool VEC>
__global__ __launch_bounds__(NT) void k_reduce2(const float* __restrict__ x, const float* __restrict__ y, int64_t n,
                                                double* __restrict__ partials) {
  __shared__ double lds[NT / 64];
  double acc = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  if (VEC) {
    const int64_t n4 = n >> 2;
    for (int64_t i = tid; i < n4; i += nth) {
      float4 a = ld4(x, i);
      if (MODE == 1) {
        acc += (double)a.x * a.x + (double)a.y * a.y + (double)a.z * a.z + (double)a.w * a.w;
      } else {
        float4 b = ld4(y, i);
        if (MODE == 0) {
          acc += (double)a.x * b.x + (double)a.y * b.y + (double)a.z * b.z + (double)a.w * b.w;
        } else {
          double d0 = (double)a.x - b.x, d1 = (double)a.y - b.y, d2 = (double)a.z - b.z, d3 = (double)a.w - b.w;
          acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
      }
    }
    for (int64_t i = (n4 << 2) + tid; i < n; i += nth) {
      double a = x[i], b = (MODE == 1) ? a : (double)y[i];
      acc += (MODE == 2) ? (a - b) * (a - b) : a * b;
    }
  } else {
    for (int64_t i = tid; i < n; i += nth) {
      double a = x[i], b = (MODE == 1) ? a : (double)y[i];
      acc += (MODE == 2) ? (a - b) * (a - b) : a * b;
    }
  }
  acc = block_sum<NT>(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

template <int MODE>
int launch_reduce2(const float* x, const float* y, int64_t n, double* out, hipStream_t s) {
  const int grid = stream_grid(n);
  double* part = nullptr;
  if (int rc = scratch_doubles(s, grid, &part)) return rc;
  const bool vec = aligned16(x) && (MODE == 1 || aligned16(y));
  if (vec)
    hipLaunchKernelGGL((k_reduce2<MODE, true>), dim3(grid), dim3(NT), 0, s, x, y, n, part);
  else
    hipLaunchKernelGGL((k_reduce2<MODE, false>), dim3(grid), dim3(NT), 0, s, x, y, n, part);
  TRK_LAUNCH_CHECK();
  return finalize_sums(part, grid, 1, 1, out, s);
}

// ------------------------------------------------------------------ out = A*x + B*y (+ sum out^2)
template <bool HAS_Y, bool SUMSQ, bool VEC>
__global__ __launch_bounds__(NT) void k_axpby(int64_t n, Coef ca, const float* x, Coef cb, const float* y, float* out,
                                              double* __restrict__ partials) {
  __shared__ double lds[NT / 64];
  const float a = (float)coef_eval(ca);
  const float b = HAS_Y ? (float)coef_eval(cb) : 0.f;
  double acc = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  int64_t tail0 = 0;
  if (VEC) {
    const int64_t n4 = n >> 2;
    tail0 = n4 << 2;
    for (int64_t i = tid; i < n4; i += nth) {
      float4 v = ld4(x, i), o;
      if (HAS_Y) {
        float4 w = ld4(y, i);
        o.x = fmaf(a, v.x, b * w.x);
        o.y = fmaf(a, v.y, b * w.y);
        o.z = fmaf(a, v.z, b * w.z);
        o.w = fmaf(a, v.w, b * w.w);
      } else {
        o.x = a * v.x;
        o.y = a * v.y;
        o.z = a * v.z;
        o.w = a * v.w;
      }
      st4(out, i, o);
      if (SUMSQ) acc += (double)o.x * o.x + (double)o.y * o.y + (double)o.z * o.z + (double)o.w * o.w;
    }
  }
  for (int64_t i = tail0 + tid; i < n; i += nth) {
    float o = HAS_Y ? fmaf(a, x[i], b * y[i]) : a * x[i];
    out[i] = o;
    if (SUMSQ) acc += (double)o * o;
  }
  if (SUMSQ) {
    acc = block_sum<NT>(acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
  }
}

// ------------------------------------------------------------------ out = a*x with <out, z> (the new basis vector and c_j = v_j . A^T b)
template <bool VEC>
__global__ __launch_bounds__(NT) void k_scale_dot(int64_t n, Coef ca, const float* x, float* out, const float* __restrict__ z,
                                                  double* __restrict__ partials) {
  __shared__ double lds[NT / 64];
  const float a = (float)coef_eval(ca);
  double acc = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  int64_t tail0 = 0;
  if (VEC) {
    const int64_t n4 = n >> 2;
    tail0 = n4 << 2;
    for (int64_t i = tid; i < n4; i += nth) {
      const float4 v = ld4(x, i), w = ld4(z, i);
      float4 o;
      o.x = a * v.x;
      o.y = a * v.y;
      o.z = a * v.z;
      o.w = a * v.w;
      st4(out, i, o);
      acc += (double)o.x * w.x + (double)o.y * w.y + (double)o.z * w.z + (double)o.w * w.w;
    }
  }
  for (int64_t i = tail0 + tid; i < n; i += nth) {
    const float o = a * x[i];
    out[i] = o;
    acc += (double)o * z[i];
  }
  acc = block_sum<NT>(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// ------------------------------------------------------------------ out = x*y ; MM weights
template <bool VEC>
__global__ __launch_bounds__(NT) void k_mul(int64_t n, const float* x, const float* y, float* out) {
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  int64_t tail0 = 0;
  if (VEC) {
    const int64_t n4 = n >> 2;
    tail0 = n4 << 2;
    for (int64_t i = tid; i < n4; i += nth) {
      float4 a = ld4(x, i), b = ld4(y, i);
      st4(out, i, make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w));
    }
  }
  for (int64_t i = tail0 + tid; i < n; i += nth) out[i] = x[i] * y[i];
}

// out = w * (x - y)
template <bool VEC>
__global__ __launch_bounds__(NT) void k_mul_diff(int64_t n, const float* w, const float* x, const float* y, float* out) {
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  int64_t tail0 = 0;
  if (VEC) {
    const int64_t n4 = n >> 2;
    tail0 = n4 << 2;
    for (int64_t i = tid; i < n4; i += nth) {
      float4 c = ld4(w, i), a = ld4(x, i), b = ld4(y, i);
      st4(out, i, make_float4(c.x * (a.x - b.x), c.y * (a.y - b.y), c.z * (a.z - b.z), c.w * (a.w - b.w)));
    }
  }
  for (int64_t i = tail0 + tid; i < n; i += nth) out[i] = w[i] * (x[i] - y[i]);
}

template <bool HAS_Y, bool VEC>
__global__ __launch_bounds__(NT) void k_mm_weights(int64_t n, const float* x, const float* y, float eps2, float e,
                                                   int special, float* out) {
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  int64_t tail0 = 0;
  if (VEC) {
    const int64_t n4 = n >> 2;
    tail0 = n4 << 2;
    for (int64_t i = tid; i < n4; i += nth) {
      float4 a = ld4(x, i);
      if (HAS_Y) {
        float4 b = ld4(y, i);
        a.x -= b.x;
        a.y -= b.y;
        a.z -= b.z;
        a.w -= b.w;
      }
      st4(out, i, make_float4(mm_w(a.x, eps2, e, special), mm_w(a.y, eps2, e, special), mm_w(a.z, eps2, e, special),
                              mm_w(a.w, eps2, e, special)));
    }
  }
  for (int64_t i = tail0 + tid; i < n; i += nth) {
    float v = HAS_Y ? x[i] - y[i] : x[i];
    out[i] = mm_w(v, eps2, e, special);
  }
}

// ------------------------------------------------------------------ group-sparsity weights (MMGKS.py:86-90)
// out[c*groups + i] = (sum_t d[i*len + t]^2 + add)^expo for c < copies: one weight per group of `len` consecutive
// entries (the nt entries of a row of Ls X), repeated `copies` times (np.kron(ones(nt), wr)).
__global__ __launch_bounds__(NT) void k_group_weights(const float* __restrict__ d, int64_t groups, int len, double add,
                                                      double expo, int copies, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (i >= groups) return;
  const float* g = d + i * len;
  double s = 0.0;
  for (int t = 0; t < len; ++t) s += (double)g[t] * (double)g[t];
  const float w = (float)pow(s + add, expo);
  for (int c = 0; c < copies; ++c) out[(int64_t)c * groups + i] = w;
}

// ------------------------------------------------------------------ isotropic-TV weights (MMGKS.py:61-77, weights.py:29-40)
// The flat iterate is viewed as X[i][j][t] (N x N x nt, t fastest: x.reshape(nx**2, nt) in C order, :71).  With the centered
// 3-point derivative of operators_old.py:22-45 (zero first / last row per axis):
//   g1 = (X[i][j+1][t] - X[i][j-1][t]) / 2  (0 at j = 0, N-1),   g2 = (X[i+1][j][t] - X[i-1][j][t]) / 2  (0 at i = 0, N-1)
//   w  = (g1^2 + g2^2 + eps^2)^e,  written to out[idx] and out[N*N*nt + idx]  (:75-76), idx = (i*N + j)*nt + t;
// behind them the temporal weights  out[2*N*N*nt + k] = (u_tail[k]^2 + eps^2)^e  (:77).  One pass: 4 B read (neighbours hit
// in L1/L2) + 8 B written per element.
__global__ __launch_bounds__(NT) void k_isotv_weights(const float* __restrict__ x, int N, int nt, const float* __restrict__ u_tail,
                                                      int64_t n_tail, float eps2, float e, int special, float* __restrict__ out) {
  const int64_t ns = (int64_t)N * N * nt;
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  const int64_t sj = nt, si = (int64_t)N * nt;
  for (int64_t idx = tid; idx < ns; idx += nth) {
    const int64_t r = idx / nt;
    const int i = (int)(r / N), j = (int)(r - (int64_t)i * N);
    const float g1 = (j > 0 && j < N - 1) ? 0.5f * x[idx + sj] - 0.5f * x[idx - sj] : 0.f;
    const float g2 = (i > 0 && i < N - 1) ? 0.5f * x[idx + si] - 0.5f * x[idx - si] : 0.f;
    const float t = fmaf(g1, g1, fmaf(g2, g2, eps2));
    const float w = special == 1 ? 1.0f : special == 2 ? 1.0f / sqrtf(t) : special == 3 ? 1.0f / sqrtf(sqrtf(t)) : powf(t, e);
    out[idx] = w;
    out[ns + idx] = w;
  }
  for (int64_t k = tid; k < n_tail; k += nth) {
    const float t = fmaf(u_tail[k], u_tail[k], eps2);
    out[2 * ns + k] = special == 1 ? 1.0f : special == 2 ? 1.0f / sqrtf(t) : special == 3 ? 1.0f / sqrtf(sqrtf(t)) : powf(t, e);
  }
}

// ------------------------------------------------------------------ fused CGLS update (CGLS.py:64-67,76,79)
// partials layout: [block][3] = ||x_new||^2, ||step*p||^2, ||x_new - x_true||^2
template <bool HAS_XT, bool VEC>
__global__ __launch_bounds__(NT) void k_cgls_update(int64_t n, int64_t m, ScalarSrc gamma, ScalarSrc delta,
                                                    const float* x, const float* p, float* x_new, float* r,
                                                    const float* w, const float* x_true, double* __restrict__ partials,
                                                    double* pub_delta, int nt) {
  __shared__ double lds[NT / 64];
  __shared__ double bc;
  float step;
  if (gamma.n == 1 && delta.n == 1) {                // finished scalars (grid-uniform)
    step = (float)(*gamma.p / *delta.p);
    if (blockIdx.x == 0 && threadIdx.x == 0 && pub_delta && pub_delta != delta.p) *pub_delta = *delta.p;   // a one-block producer
  } else {                                           // block partials of the producing kernel: one wave sums them
    if (threadIdx.x < 64) {
      const double g = scalar_from_wave(gamma, threadIdx.x), d = scalar_from_wave(delta, threadIdx.x);
      if (threadIdx.x == 0) {
        bc = g / d;
        if (blockIdx.x == 0 && pub_delta) *pub_delta = d;
      }
    }
    __syncthreads();
    step = (float)bc;
  }
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  int64_t ntail = 0, mtail = 0;
  if (VEC) {
    const int64_t n4 = n >> 2, m4 = m >> 2;
    ntail = n4 << 2;
    mtail = m4 << 2;
    for (int64_t i = tid; i < n4; i += nth) {
      const float4 xv = ld4(x, i), pv = ld4(p, i);
      const float4 d = make_float4(step * pv.x, step * pv.y, step * pv.z, step * pv.w);
      const float4 xn = make_float4(xv.x + d.x, xv.y + d.y, xv.z + d.z, xv.w + d.w);
      if (nt & 2) st4_nt(x_new, i, xn); else st4(x_new, i, xn);
      s0 += (double)xn.x * xn.x + (double)xn.y * xn.y + (double)xn.z * xn.z + (double)xn.w * xn.w;
      s1 += (double)d.x * d.x + (double)d.y * d.y + (double)d.z * d.z + (double)d.w * d.w;
      if (HAS_XT) {
        const float4 t = ld4(x_true, i);
        const double e0 = (double)xn.x - t.x, e1 = (double)xn.y - t.y, e2 = (double)xn.z - t.z, e3 = (double)xn.w - t.w;
        s2 += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
      }
    }
    for (int64_t i0 = tid; i0 < m4; i0 += nth) {
      const int64_t i = (nt & 256) ? m4 - 1 - i0 : i0;      // TRK_REV: last-written rows of the producer first (Infinity-Cache experiment)
      float4 rv = ld4(r, i);
      const float4 wv = ld4(w, i);
      rv.x = fmaf(-step, wv.x, rv.x);
      rv.y = fmaf(-step, wv.y, rv.y);
      rv.z = fmaf(-step, wv.z, rv.z);
      rv.w = fmaf(-step, wv.w, rv.w);
      if (nt & 4) st4_nt(r, i, rv); else st4(r, i, rv);
    }
  }
  for (int64_t i = ntail + tid; i < n; i += nth) {
    const float d = step * p[i];
    const float xn = x[i] + d;
    x_new[i] = xn;
    s0 += (double)xn * xn;
    s1 += (double)d * d;
    if (HAS_XT) {
      const double e = (double)xn - x_true[i];
      s2 += e * e;
    }
  }
  for (int64_t i = mtail + tid; i < m; i += nth) r[i] = fmaf(-step, w[i], r[i]);
  s0 = block_sum<NT>(s0, lds);
  s1 = block_sum<NT>(s1, lds);
  if (HAS_XT) s2 = block_sum<NT>(s2, lds);
  if (threadIdx.x == 0) {
    partials[blockIdx.x * 3 + 0] = s0;
    partials[blockIdx.x * 3 + 1] = s1;
    partials[blockIdx.x * 3 + 2] = HAS_XT ? s2 : 0.0;
  }
}

// ------------------------------------------------------------------ CGLS direction update (CGLS.py:72)
// p = t + (gamma_new / gamma_old) p with gamma_new possibly still the block partials of the adjoint kernel that produced
// t; block 0 publishes the finished gamma_new.  Same arithmetic as trk_axpby(1, t, gamma_new/gamma_old, p).
template <bool VEC>
__global__ __launch_bounds__(NT) void k_cgls_p_update(int64_t n, const float* __restrict__ t, float* p, ScalarSrc gnew,
                                                      const double* gold, double* pub_gamma, int nt) {
  __shared__ double bc;
  if (threadIdx.x < 64) {
    const double g = scalar_from_wave(gnew, threadIdx.x);
    if (threadIdx.x == 0) {
      bc = g / *gold;
      if (blockIdx.x == 0 && pub_gamma) *pub_gamma = g;
    }
  }
  __syncthreads();
  const float b = (float)bc;
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  int64_t tail0 = 0;
  if (VEC) {
    const int64_t n4 = n >> 2;
    tail0 = n4 << 2;
    for (int64_t i = tid; i < n4; i += nth) {
      const float4 v = ld4(t, i), w = ld4(p, i);
      float4 o;
      o.x = fmaf(1.f, v.x, b * w.x);
      o.y = fmaf(1.f, v.y, b * w.y);
      o.z = fmaf(1.f, v.z, b * w.z);
      o.w = fmaf(1.f, v.w, b * w.w);
      if (nt & 8) st4_nt(p, i, o); else st4(p, i, o);
    }
  }
  for (int64_t i = tail0 + tid; i < n; i += nth) p[i] = fmaf(1.f, t[i], b * p[i]);
}

// ------------------------------------------------------------------ CGLS residual update alone (CGLS.py:67)
// r -= (gamma_old / S(delta)) w with delta possibly still the block partials of the forward kernel; block 0 publishes it.
template <bool VEC>
__global__ __launch_bounds__(NT) void k_cgls_r_update(int64_t m, const double* gold, ScalarSrc delta, float* r,
                                                      const float* __restrict__ w, double* pub_delta, int nt) {
  __shared__ double bc;
  if (threadIdx.x < 64) {
    const double d = scalar_from_wave(delta, threadIdx.x);
    if (threadIdx.x == 0) {
      bc = *gold / d;
      if (blockIdx.x == 0 && pub_delta) *pub_delta = d;
    }
  }
  __syncthreads();
  const float step = (float)bc;
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  int64_t tail0 = 0;
  if (VEC) {
    const int64_t m4 = m >> 2;
    tail0 = m4 << 2;
    for (int64_t i0 = tid; i0 < m4; i0 += nth) {
      const int64_t i = (nt & 256) ? m4 - 1 - i0 : i0;      // TRK_REV: last-written rows of the producer first (Infinity-Cache experiment)
      float4 rv = ld4(r, i);
      const float4 wv = ld4(w, i);
      rv.x = fmaf(-step, wv.x, rv.x);
      rv.y = fmaf(-step, wv.y, rv.y);
      rv.z = fmaf(-step, wv.z, rv.z);
      rv.w = fmaf(-step, wv.w, rv.w);
      if (nt & 4) st4_nt(r, i, rv); else st4(r, i, rv);
    }
  }
  for (int64_t i = tail0 + tid; i < m; i += nth) r[i] = fmaf(-step, w[i], r[i]);
}

// ------------------------------------------------------------------ CGLS iterate + direction update in one pass over p
// x_new = x + (gamma_old/delta) p (CGLS.py:64-65) and p = t + (S(gamma_new)/gamma_old) p (:72): p is read once for both
// (20n bytes instead of 12n + 12n); norms as k_cgls_update: [block][3] raw partials; block 0 publishes gamma_new.
template <bool HAS_XT>
__global__ __launch_bounds__(NT) void k_cgls_xp_update(int64_t n, const double* gold, const double* delta, ScalarSrc gnew,
                                                       const float* __restrict__ x, float* p, const float* __restrict__ t,
                                                       float* __restrict__ x_new, const float* __restrict__ x_true,
                                                       double* pub_gamma, double* __restrict__ partials, int nt) {
  __shared__ double lds[NT / 64];
  __shared__ double bc[2];
  if (threadIdx.x < 64) {
    const double g = scalar_from_wave(gnew, threadIdx.x);
    if (threadIdx.x == 0) {
      bc[0] = *gold / *delta;
      bc[1] = g / *gold;
      if (blockIdx.x == 0 && pub_gamma) *pub_gamma = g;
    }
  }
  __syncthreads();
  const float step = (float)bc[0], b = (float)bc[1];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  const int64_t n4 = n >> 2;
  for (int64_t i0 = tid; i0 < n4; i0 += nth) {
    const int64_t i = (nt & 256) ? n4 - 1 - i0 : i0;        // TRK_REV (see k_cgls_r_update)
    const float4 xv = (nt & 16) ? ld4_nt(x, i) : ld4(x, i), pv = ld4(p, i), tv = (nt & 32) ? ld4_nt(t, i) : ld4(t, i);
    const float4 d = make_float4(step * pv.x, step * pv.y, step * pv.z, step * pv.w);
    const float4 xn = make_float4(xv.x + d.x, xv.y + d.y, xv.z + d.z, xv.w + d.w);
    if (nt & 2) st4_nt(x_new, i, xn); else st4(x_new, i, xn);
    float4 o;
    o.x = fmaf(1.f, tv.x, b * pv.x);
    o.y = fmaf(1.f, tv.y, b * pv.y);
    o.z = fmaf(1.f, tv.z, b * pv.z);
    o.w = fmaf(1.f, tv.w, b * pv.w);
    if (nt & 8) st4_nt(p, i, o); else st4(p, i, o);
    s0 += (double)xn.x * xn.x + (double)xn.y * xn.y + (double)xn.z * xn.z + (double)xn.w * xn.w;
    s1 += (double)d.x * d.x + (double)d.y * d.y + (double)d.z * d.z + (double)d.w * d.w;
    if (HAS_XT) {
      const float4 tt = ld4(x_true, i);
      const double e0 = (double)xn.x - tt.x, e1 = (double)xn.y - tt.y, e2 = (double)xn.z - tt.z, e3 = (double)xn.w - tt.w;
      s2 += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
    }
  }
  for (int64_t i = (n4 << 2) + tid; i < n; i += nth) {
    const float pv = p[i];
    const float d = step * pv;
    const float xn = x[i] + d;
    x_new[i] = xn;
    p[i] = fmaf(1.f, t[i], b * pv);
    s0 += (double)xn * xn;
    s1 += (double)d * d;
    if (HAS_XT) {
      const double e = (double)xn - x_true[i];
      s2 += e * e;
    }
  }
  s0 = block_sum<NT>(s0, lds);
  s1 = block_sum<NT>(s1, lds);
  if (HAS_XT) s2 = block_sum<NT>(s2, lds);
  if (threadIdx.x == 0) {
    partials[blockIdx.x * 3 + 0] = s0;
    partials[blockIdx.x * 3 + 1] = s1;
    partials[blockIdx.x * 3 + 2] = HAS_XT ? s2 : 0.0;
  }
}

// ------------------------------------------------------------------ CGLS x-update of the fused fast path
// x_new = x + (gamma/delta) p with gamma, delta possibly still block partials of the producing blur kernels; block 0
// publishes the two finished scalars; the three norms are left as raw partials [block][3] (summed once, after the solve).
template <bool HAS_XT>
__global__ __launch_bounds__(NT) void k_cgls_x_update(int64_t n, ScalarSrc gamma, ScalarSrc delta, const float* __restrict__ x,
                                                      const float* __restrict__ p, float* __restrict__ x_new,
                                                      const float* __restrict__ x_true, double* pub_delta,
                                                      double* pub_gamma, double* __restrict__ partials) {
  __shared__ double lds[NT / 64];
  __shared__ double bc[2];
  if (threadIdx.x < 64) {                          // one wave evaluates both scalars (fixed summation order)
    const double g = scalar_from_wave(gamma, threadIdx.x), d = scalar_from_wave(delta, threadIdx.x);
    if (threadIdx.x == 0) {
      bc[0] = g;
      bc[1] = d;
      if (blockIdx.x == 0) {
        if (pub_gamma) *pub_gamma = g;
        if (pub_delta) *pub_delta = d;
      }
    }
  }
  __syncthreads();
  const float step = (float)(bc[0] / bc[1]);
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  const int64_t n4 = n >> 2;
  for (int64_t i = tid; i < n4; i += nth) {
    const float4 xv = ld4(x, i), pv = ld4(p, i);
    const float4 d = make_float4(step * pv.x, step * pv.y, step * pv.z, step * pv.w);
    const float4 xn = make_float4(xv.x + d.x, xv.y + d.y, xv.z + d.z, xv.w + d.w);
    st4(x_new, i, xn);
    s0 += (double)xn.x * xn.x + (double)xn.y * xn.y + (double)xn.z * xn.z + (double)xn.w * xn.w;
    s1 += (double)d.x * d.x + (double)d.y * d.y + (double)d.z * d.z + (double)d.w * d.w;
    if (HAS_XT) {
      const float4 t = ld4(x_true, i);
      const double e0 = (double)xn.x - t.x, e1 = (double)xn.y - t.y, e2 = (double)xn.z - t.z, e3 = (double)xn.w - t.w;
      s2 += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
    }
  }
  for (int64_t i = (n4 << 2) + tid; i < n; i += nth) {
    const float d = step * p[i];
    const float xn = x[i] + d;
    x_new[i] = xn;
    s0 += (double)xn * xn;
    s1 += (double)d * d;
    if (HAS_XT) {
      const double e = (double)xn - x_true[i];
      s2 += e * e;
    }
  }
  s0 = block_sum<NT>(s0, lds);
  s1 = block_sum<NT>(s1, lds);
  if (HAS_XT) s2 = block_sum<NT>(s2, lds);
  if (threadIdx.x == 0) {
    partials[blockIdx.x * 3 + 0] = s0;
    partials[blockIdx.x * 3 + 1] = s1;
    partials[blockIdx.x * 3 + 2] = HAS_XT ? s2 : 0.0;
  }
}

// ------------------------------------------------------------------ h[j] = sum_i wt(i) V[j][i] r[i]   (k dots, one pass)
// grid = (bx, ceil(k/JT)); a block sweeps its share of i for JT rows; partials [bx][k].
// WPOW: 0 no weight, 1 multiply by w, 2 multiply by w^2.
constexpr int JT = 8;

template <int WPOW, bool VEC>
__global__ __launch_bounds__(NT) void k_gemv_t(const float* __restrict__ V, int64_t ld, int kv, int64_t n,
                                               const float* __restrict__ r, const float* __restrict__ w,
                                               double* __restrict__ partials, int nt, const float* __restrict__ xrow = nullptr) {
  __shared__ double lds[(NT / 64) * JT];
  // xrow: one more row that is not part of the basis (trk_gemv_t_x: the right-hand side b next to the images A v_j), row index kv
  const int k = kv + (xrow ? 1 : 0);
  // row tiles of equal height: ceil(k / tiles) <= JT rows each (k = 18: 6 + 6 + 6, not 8 + 8 + 2 — the short tile's workgroups
  // read the right-hand sides for a quarter of the work)
  const int jb = (k + (int)gridDim.y - 1) / (int)gridDim.y;
  const int j0 = blockIdx.y * jb;
  const int jn = (k - j0 < jb) ? (k - j0 < 0 ? 0 : k - j0) : jb;
  auto row = [&](int j) -> const float* { return (j0 + j < kv) ? V + (int64_t)(j0 + j) * ld : xrow; };
  double acc[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j) acc[j] = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  int64_t tail0 = 0;
  if (VEC) {
    const int64_t n4 = n >> 2;
    tail0 = n4 << 2;
    for (int64_t i = tid; i < n4; i += nth) {
      float4 rv = ld4(r, i);
      if (WPOW) {
        float4 wv = ld4(w, i);
        if (WPOW == 2) {
          wv.x *= wv.x;
          wv.y *= wv.y;
          wv.z *= wv.z;
          wv.w *= wv.w;
        }
        rv.x *= wv.x;
        rv.y *= wv.y;
        rv.z *= wv.z;
        rv.w *= wv.w;
      }
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        if (j < jn) {
          float4 v = (nt & 64) ? ld4_nt(row(j), i) : ld4(row(j), i);
          acc[j] += (double)v.x * rv.x + (double)v.y * rv.y + (double)v.z * rv.z + (double)v.w * rv.w;
        }
      }
    }
  }
  for (int64_t i = tail0 + tid; i < n; i += nth) {
    float rv = r[i];
    if (WPOW) {
      float wv = w[i];
      rv *= (WPOW == 2) ? wv * wv : wv;
    }
#pragma unroll
    for (int j = 0; j < JT; ++j)
      if (j < jn) acc[j] += (double)row(j)[i] * rv;
  }
  // (one exchange for the JT sums: with a block_sum each, the 2 JT barriers of a workgroup were a visible part of the kernel on
  // short vectors — dynamic problems, n = 2 M)
  const double t = block_sum_many<NT, JT>(acc, lds);
  if ((int)threadIdx.x < jn) partials[(size_t)blockIdx.x * k + j0 + threadIdx.x] = t;
}

// Two right-hand sides in one sweep over the basis: h[j] = V[j] . r and g[j] = V[j] . r2 (partials [bx][2k]).  What the
// Gram-matrix form of the repeated Gram-Schmidt sweeps needs: the coefficients of the new direction AND the Gram row of the
// vector appended last time, for the price of reading the basis once.
template <bool VEC>
__global__ __launch_bounds__(NT) void k_gemv_t2(const float* __restrict__ V, int64_t ld, int k, int64_t n,
                                                const float* __restrict__ r, const float* __restrict__ r2,
                                                double* __restrict__ partials, int nt) {
  __shared__ double lds[(NT / 64) * 2 * JT];
  // row tiles of equal height: ceil(k / tiles) <= JT rows each (k = 18: 6 + 6 + 6, not 8 + 8 + 2 — the short tile's workgroups
  // read the right-hand sides for a quarter of the work)
  const int jb = (k + (int)gridDim.y - 1) / (int)gridDim.y;
  const int j0 = blockIdx.y * jb;
  const int jn = (k - j0 < jb) ? (k - j0 < 0 ? 0 : k - j0) : jb;
  double acc[JT], acc2[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j) acc[j] = acc2[j] = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  int64_t tail0 = 0;
  if (VEC) {
    const int64_t n4 = n >> 2;
    tail0 = n4 << 2;
    for (int64_t i = tid; i < n4; i += nth) {
      const float4 rv = ld4(r, i), sv = ld4(r2, i);
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        if (j < jn) {
          float4 v = (nt & 64) ? ld4_nt(V + (int64_t)(j0 + j) * ld, i) : ld4(V + (int64_t)(j0 + j) * ld, i);
          acc[j] += (double)v.x * rv.x + (double)v.y * rv.y + (double)v.z * rv.z + (double)v.w * rv.w;
          acc2[j] += (double)v.x * sv.x + (double)v.y * sv.y + (double)v.z * sv.z + (double)v.w * sv.w;
        }
      }
    }
  }
  for (int64_t i = tail0 + tid; i < n; i += nth) {
    const float rv = r[i], sv = r2[i];
#pragma unroll
    for (int j = 0; j < JT; ++j)
      if (j < jn) {
        const double v = (double)V[(int64_t)(j0 + j) * ld + i];
        acc[j] += v * rv;
        acc2[j] += v * sv;
      }
  }
  double both[2 * JT];
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    both[j] = acc[j];
    both[JT + j] = acc2[j];
  }
  const double t = block_sum_many<NT, 2 * JT>(both, lds);      // value i in thread i
  const int q = threadIdx.x / JT, j = threadIdx.x % JT;
  if (threadIdx.x < 2 * JT && j < jn) partials[(size_t)blockIdx.x * 2 * k + (size_t)q * k + j0 + j] = t;
}

// The same with R right-hand sides (R = 3, 4): out[q k + j] = V[j] . rhs[q].  GKS rides the Gram rows of its NEXT basis vector on
// the sweep that orthogonalises it (krylov.GramSchmidtByGram.sweep, solvers/GKS.py): V^T (A^T A r) and V^T (L^T L r) next to V^T r
// and the newest vector's row of V^T V — one pass over the basis instead of two.
struct RhsSet {
  const float* p[4];
};
template <bool VEC, int R>
__global__ __launch_bounds__(NT) void k_gemv_tr(const float* __restrict__ V, int64_t ld, int k, int64_t n, RhsSet rhs,
                                                double* __restrict__ partials, int nt) {
  __shared__ double lds[(NT / 64) * R * JT];
  // row tiles of equal height: ceil(k / tiles) <= JT rows each (k = 18: 6 + 6 + 6, not 8 + 8 + 2 — the short tile's workgroups
  // read the right-hand sides for a quarter of the work)
  const int jb = (k + (int)gridDim.y - 1) / (int)gridDim.y;
  const int j0 = blockIdx.y * jb;
  const int jn = (k - j0 < jb) ? (k - j0 < 0 ? 0 : k - j0) : jb;
  double acc[R][JT];
#pragma unroll
  for (int q = 0; q < R; ++q)
#pragma unroll
    for (int j = 0; j < JT; ++j) acc[q][j] = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  int64_t tail0 = 0;
  if (VEC) {
    const int64_t n4 = n >> 2;
    tail0 = n4 << 2;
    for (int64_t i = tid; i < n4; i += nth) {
      float4 rv[R];
#pragma unroll
      for (int q = 0; q < R; ++q) rv[q] = ld4(rhs.p[q], i);
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        if (j < jn) {
          const float4 v = (nt & 64) ? ld4_nt(V + (int64_t)(j0 + j) * ld, i) : ld4(V + (int64_t)(j0 + j) * ld, i);
#pragma unroll
          for (int q = 0; q < R; ++q)
            acc[q][j] += (double)v.x * rv[q].x + (double)v.y * rv[q].y + (double)v.z * rv[q].z + (double)v.w * rv[q].w;
        }
      }
    }
  }
  for (int64_t i = tail0 + tid; i < n; i += nth) {
    float rv[R];
#pragma unroll
    for (int q = 0; q < R; ++q) rv[q] = rhs.p[q][i];
#pragma unroll
    for (int j = 0; j < JT; ++j)
      if (j < jn) {
        const double v = (double)V[(int64_t)(j0 + j) * ld + i];
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q][j] += v * rv[q];
      }
  }
  double all[R * JT];
#pragma unroll
  for (int q = 0; q < R; ++q)
#pragma unroll
    for (int j = 0; j < JT; ++j) all[q * JT + j] = acc[q][j];
  const double t = block_sum_many<NT, R * JT>(all, lds);        // value i in thread i
  const int q = threadIdx.x / JT, j = threadIdx.x % JT;
  if (threadIdx.x < R * JT && j < jn) partials[(size_t)blockIdx.x * R * k + (size_t)q * k + j0 + j] = t;
}

int launch_gemv_t(const float* V, int64_t ld, int k, int64_t n, const float* r, const float* w, int wpow, double* h,
                  hipStream_t s, const float* xrow = nullptr, double* h_x = nullptr) {
  const int kt = k + (xrow ? 1 : 0);
  const int ntile = ceil_div(kt, JT);
  static const int occ = resident_blocks_per_cu(k_gemv_t<0, true>);
  const int bx = tiled_dot_grid_x(n, ntile, occ);
  double* part = nullptr;
  if (int rc = scratch_doubles(s, (size_t)bx * kt, &part)) return rc;
  const bool vec = aligned16(V) && aligned16(r) && (ld % 4 == 0) && (!wpow || aligned16(w)) && (!xrow || aligned16(xrow));
  dim3 grid(bx, ntile);
#define GT(WP, VC) hipLaunchKernelGGL((k_gemv_t<WP, VC>), grid, dim3(NT), 0, s, V, ld, k, n, r, w, part, stream_nontemporal(n), xrow)
  if (wpow == 0) { if (vec) GT(0, true); else GT(0, false); }
  else if (wpow == 1) { if (vec) GT(1, true); else GT(1, false); }
  else { if (vec) GT(2, true); else GT(2, false); }
#undef GT
  TRK_LAUNCH_CHECK();
  if (xrow) return finalize_sums_split(part, bx, kt, kt, h, k, h_x, s);
  return finalize_sums(part, bx, k, k, h, s);
}

// ------------------------------------------------------------------ out = a*base + s * sum_j y[j] V[j]   (+ sum out^2)
constexpr int KMAX_LDS = 1024;  // coefficients staged in LDS as doubles

// HAS_REF: the partials are those of sum (out - ref)^2 instead of sum out^2 (the error norm against x_true)
// where the k coefficients come from: device memory, or the launch's own arguments (values the HOST holds — the projected solution
// of a hybrid solver whose lambda was chosen there — ride in the dispatch packet: no upload, no kernel that computes them)
struct YPtr {
  const double* p;
  __device__ __forceinline__ double at(int j) const { return p[j]; }
};
constexpr int YARG_MAX = 128;
struct YArg {
  double v[YARG_MAX];
  // read where the dispatch put them — the kernel-argument segment, of which this struct is the FIRST member (k_gemv_n) — and not
  // through `v`: indexing a by-value aggregate with the thread index makes every thread copy all of it to scratch first
  __device__ __forceinline__ double at(int j) const {
    return ((const __attribute__((address_space(4))) double*)__builtin_amdgcn_kernarg_segment_ptr())[j];
  }
};

template <bool HAS_BASE, bool SUMSQ, bool VEC, bool HAS_REF = false, int U = 8, class YS = YPtr>
__global__ __launch_bounds__(NT) void k_gemv_n(const YS y, const float* __restrict__ V, int64_t ld, int k, int64_t n,
                                               double a, const float* base, double sc,
                                               float* out, double* __restrict__ partials,
                                               const float* __restrict__ ref = nullptr, int nt = 0) {
  __shared__ double ys[KMAX_LDS];
  __shared__ double lds[NT / 64];
  for (int j = threadIdx.x; j < k; j += NT) ys[j] = sc * y.at(j);
  __syncthreads();
  double acc2 = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  int64_t tail0 = 0;
  if (VEC) {
    const int64_t n4 = n >> 2;
    tail0 = n4 << 2;
    for (int64_t i = tid; i < n4; i += nth) {
      double o0 = 0, o1 = 0, o2 = 0, o3 = 0;
      if (HAS_BASE) {
        float4 b = ld4(base, i);
        o0 = a * b.x;
        o1 = a * b.y;
        o2 = a * b.z;
        o3 = a * b.w;
      }
      // rows of the basis are requested in groups — U, then 8, then 4 — before the first of a group is used (the compiler's own
      // unrolling of the plain loop kept 4 in flight; U is the launcher's choice, trk_gemv_n)
      int j = 0;
      auto group = [&](auto width) {
        constexpr int W = decltype(width)::value;
        float4 v[W];
#pragma unroll
        for (int u = 0; u < W; ++u) v[u] = (nt & 128) ? ld4_nt(V + (int64_t)(j + u) * ld, i) : ld4(V + (int64_t)(j + u) * ld, i);
#pragma unroll
        for (int u = 0; u < W; ++u) {
          const double c = ys[j + u];
          o0 = fma(c, (double)v[u].x, o0);
          o1 = fma(c, (double)v[u].y, o1);
          o2 = fma(c, (double)v[u].z, o2);
          o3 = fma(c, (double)v[u].w, o3);
        }
        j += W;
      };
      while (j + U <= k) group(std::integral_constant<int, U>{});
      if (U > 8 && j + 8 <= k) group(std::integral_constant<int, 8>{});
      if (U > 4 && j + 4 <= k) group(std::integral_constant<int, 4>{});
      while (j < k) group(std::integral_constant<int, 1>{});
      float4 o = make_float4((float)o0, (float)o1, (float)o2, (float)o3);
      st4(out, i, o);
      if (SUMSQ && HAS_REF) {
        const float4 t = ld4(ref, i);
        const double e0 = (double)o.x - t.x, e1 = (double)o.y - t.y, e2 = (double)o.z - t.z, e3 = (double)o.w - t.w;
        acc2 += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
      } else if (SUMSQ) {
        acc2 += (double)o.x * o.x + (double)o.y * o.y + (double)o.z * o.z + (double)o.w * o.w;
      }
    }
  }
  for (int64_t i = tail0 + tid; i < n; i += nth) {
    double o = HAS_BASE ? a * base[i] : 0.0;
    for (int j = 0; j < k; ++j) o = fma(ys[j], (double)V[(int64_t)j * ld + i], o);
    const float of = (float)o;
    out[i] = of;
    if (SUMSQ && HAS_REF) {
      const double e = (double)of - ref[i];
      acc2 += e * e;
    } else if (SUMSQ) {
      acc2 += (double)of * of;
    }
  }
  if (SUMSQ) {
    acc2 = block_sum<NT>(acc2, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc2;
  }
}

// ------------------------------------------------------------------ out = (a*base + s * sum_j y[j] V[j]) [/ sqrt(*den2)] for SHORT vectors
// k_gemv_n gives a thread one 16-byte column of the basis and walks the k rows eight loads at a time: with m-length images of a
// projector (dynamic tomography, C5: m = 122 880 floats against n = 2 M) the grid is 30 workgroups and every thread waits for k / 8
// dependent round trips — 11.5 us for k = 40 rows of half a megabyte each.  Here the four waves of a workgroup share 64 columns and take
// a quarter of the rows each; their partial sums meet in LDS in wave order ((0 + 1) + (2 + 3)).  16-byte aligned operands, n % 4 == 0.
template <bool HAS_BASE>
__global__ __launch_bounds__(NT) void k_gemv_n_split(const double* __restrict__ y, const float* __restrict__ V, int64_t ld, int k,
                                                     int64_t n4, double a, const float* __restrict__ base, double sc, float* out,
                                                     const double* __restrict__ den2) {
  __shared__ double ys[KMAX_LDS];
  __shared__ double part[NT / 64][64][4];
  for (int j = threadIdx.x; j < k; j += NT) ys[j] = sc * y[j];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const int kq = (k + NT / 64 - 1) / (NT / 64);
  const int j0 = wv * kq, j1 = (j0 + kq < k) ? j0 + kq : k;
  double o0 = 0, o1 = 0, o2 = 0, o3 = 0;
  if (i < n4) {
    int j = j0;
    auto group = [&](auto width) {
      constexpr int W = decltype(width)::value;
      float4 v[W];
#pragma unroll
      for (int u = 0; u < W; ++u) v[u] = ld4(V + (int64_t)(j + u) * ld, i);
#pragma unroll
      for (int u = 0; u < W; ++u) {
        const double c = ys[j + u];
        o0 = fma(c, (double)v[u].x, o0);
        o1 = fma(c, (double)v[u].y, o1);
        o2 = fma(c, (double)v[u].z, o2);
        o3 = fma(c, (double)v[u].w, o3);
      }
      j += W;
    };
    while (j + 8 <= j1) group(std::integral_constant<int, 8>{});
    if (j + 4 <= j1) group(std::integral_constant<int, 4>{});
    while (j < j1) group(std::integral_constant<int, 1>{});
  }
  part[wv][lane][0] = o0;
  part[wv][lane][1] = o1;
  part[wv][lane][2] = o2;
  part[wv][lane][3] = o3;
  __syncthreads();
  if (wv != 0 || i >= n4) return;
  double t[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) t[q] = (part[0][lane][q] + part[1][lane][q]) + (part[2][lane][q] + part[3][lane][q]);
  if (HAS_BASE) {
    const float4 b = ld4(base, i);
    t[0] = fma(a, (double)b.x, t[0]);
    t[1] = fma(a, (double)b.y, t[1]);
    t[2] = fma(a, (double)b.z, t[2]);
    t[3] = fma(a, (double)b.w, t[3]);
  }
  if (den2) {
    const double inv = 1.0 / sqrt(*den2);
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] *= inv;
  }
  st4(out, i, make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]));
}
// whether the short-vector form serves a call (no fused norm, aligned, few columns, enough rows to be worth splitting)
static bool gemv_n_split_serves(int64_t n, int k, bool vec) {
  static const int on = env_int("TRK_GEMVN_SPLIT", 1);
  return on && vec && (n % 4) == 0 && n > 0 && (n >> 2) <= (int64_t)64 * 4 * cu_count() && k >= 12 && k <= KMAX_LDS;
}

// ------------------------------------------------------------------ the new basis vector AND the next iterate in ONE pass over the basis
// GKS / MMGKS pass over V three / four times per iteration: x = V y (GKS.py:76), h = V^T r, r - V c (:86-88) [+ the re-weighted Gram].
// The iterate of the NEXT iteration is x' = V y'[0..k) + y'[k] v_k with v_k = (r - V c) / rho the vector this sweep produces — and y'
// needs nothing of v_k but its Gram rows, which follow from the products of the h-sweep (trk_gram_row_from_sweep), and rho, which
// follows from them too (trk_cgs_coeffs_rho: rho^2 = r.r - 2 c.h + c.G c; r is the residual of the projected normal equations, h
// and c are of rounding size, nothing cancels).  So the projected problem of the next iteration is solved BEFORE this pass and the
// pass leaves both vectors: one read of the basis less per iteration.
//   vn = (w - sum_j c[j] V[j]) / sqrt(*rho2)        — the sums of k_gemv_n<HAS_BASE> in its order, ONE rounding to fp32 (after the scaling)
//   x  = sum_{j<k} y[j] V[j] + y[k] vn              — k_gemv_n's sum over the k + 1 stored vectors, term for term (vn as stored)
// HAS_REF: block partials of ||x - ref||^2 (trk_gemv_n_err's); chk != nullptr: block partials of ||w - V c||^2 as computed (float64,
// before the scaling) — what rho^2 stands for, for callers who want to see the two agree.
template <bool VEC, bool HAS_X, bool HAS_REF, int U>
__global__ __launch_bounds__(NT) void k_gemv_orth_iter(const float* __restrict__ V, int64_t ld, int k, int64_t n,
                                                       const float* __restrict__ w, const double* __restrict__ c,
                                                       const double* __restrict__ rho2, const double* __restrict__ y, float* vn, float* x,
                                                       const float* __restrict__ ref, double* __restrict__ partials,
                                                       double* __restrict__ chk, int nt) {
  __shared__ double2 cy[KMAX_LDS];      // (-c[j], y[j])
  __shared__ double lds[NT / 64];
  for (int j = threadIdx.x; j < k; j += NT) cy[j] = make_double2(-c[j], HAS_X ? y[j] : 0.0);
  const double inv = 1.0 / sqrt(*rho2);
  const double yk = HAS_X ? y[k] : 0.0;
  __syncthreads();
  double acc2 = 0.0, accc = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  int64_t tail0 = 0;
  if (VEC) {
    const int64_t n4 = n >> 2;
    tail0 = n4 << 2;
    for (int64_t i = tid; i < n4; i += nth) {
      const float4 b = ld4(w, i);
      double o0 = 1.0 * b.x, o1 = 1.0 * b.y, o2 = 1.0 * b.z, o3 = 1.0 * b.w;
      double x0 = 0, x1 = 0, x2 = 0, x3 = 0;
      int j = 0;
      auto group = [&](auto width) {
        constexpr int W = decltype(width)::value;
        float4 v[W];
#pragma unroll
        for (int u = 0; u < W; ++u) v[u] = (nt & 128) ? ld4_nt(V + (int64_t)(j + u) * ld, i) : ld4(V + (int64_t)(j + u) * ld, i);
#pragma unroll
        for (int u = 0; u < W; ++u) {
          const double2 q = cy[j + u];
          o0 = fma(q.x, (double)v[u].x, o0);
          o1 = fma(q.x, (double)v[u].y, o1);
          o2 = fma(q.x, (double)v[u].z, o2);
          o3 = fma(q.x, (double)v[u].w, o3);
          if (HAS_X) {
            x0 = fma(q.y, (double)v[u].x, x0);
            x1 = fma(q.y, (double)v[u].y, x1);
            x2 = fma(q.y, (double)v[u].z, x2);
            x3 = fma(q.y, (double)v[u].w, x3);
          }
        }
        j += W;
      };
      while (j + U <= k) group(std::integral_constant<int, U>{});
      if (U > 4 && j + 4 <= k) group(std::integral_constant<int, 4>{});
      while (j < k) group(std::integral_constant<int, 1>{});
      if (chk) accc += (o0 * o0 + o1 * o1) + (o2 * o2 + o3 * o3);
      const float4 vo = make_float4((float)(o0 * inv), (float)(o1 * inv), (float)(o2 * inv), (float)(o3 * inv));
      st4(vn, i, vo);
      if (HAS_X) {
        const float4 xo = make_float4((float)fma(yk, (double)vo.x, x0), (float)fma(yk, (double)vo.y, x1), (float)fma(yk, (double)vo.z, x2),
                                      (float)fma(yk, (double)vo.w, x3));
        st4(x, i, xo);
        if (HAS_REF) {
          const float4 t = ld4(ref, i);
          const double e0 = (double)xo.x - t.x, e1 = (double)xo.y - t.y, e2 = (double)xo.z - t.z, e3 = (double)xo.w - t.w;
          acc2 += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
        }
      }
    }
  }
  for (int64_t i = tail0 + tid; i < n; i += nth) {
    double o = 1.0 * w[i], xs = 0.0;
    for (int j = 0; j < k; ++j) {
      const double v = (double)V[(int64_t)j * ld + i];
      o = fma(cy[j].x, v, o);
      if (HAS_X) xs = fma(cy[j].y, v, xs);
    }
    if (chk) accc += o * o;
    const float vo = (float)(o * inv);
    vn[i] = vo;
    if (HAS_X) {
      const float xo = (float)fma(yk, (double)vo, xs);
      x[i] = xo;
      if (HAS_REF) {
        const double e = (double)xo - ref[i];
        acc2 += e * e;
      }
    }
  }
  if (HAS_X && HAS_REF) {
    acc2 = block_sum<NT>(acc2, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc2;
  }
  if (chk) {                                                      // uniform over the grid
    __syncthreads();
    accc = block_sum<NT>(accc, lds);
    if (threadIdx.x == 0) chk[blockIdx.x] = accc;
  }
}

// ------------------------------------------------------------------ damped-LSQR iterate by its short recurrence
// x_k = V_k y_k with y_k = argmin || [B_k; damp I] y - beta_1 e_1 ||  (Hybrid_LSQR.py:104-105 with a FIXED lambda, damp =
// sqrt(lambda)) is Paige & Saunders' damped LSQR iterate, which obeys  w_k = v_k - (theta_k / rho_{k-1}) w_{k-1},
// x_k = x_{k-1} + (phi_k / rho_k) w_k  — an algebraic identity in B_k (no orthogonality of V is used), so the k-term
// combination per iterate (4 k n bytes) becomes one pass over three vectors.  The plane rotations (two per step: one against
// the damping, one against beta_{k+1}) are recomputed by thread 0 of every workgroup from alpha_k^2, beta_{k+1}^2 and the
// four doubles the previous step left in st_in = {cs, sn, rho, phibar}; workgroup 0 leaves this step's in st_out (the caller
// alternates two slots).  vk is alpha_k v_k as GKState(normalized=False) stores it.
// Templated on the element type T of the vectors (float: the product; double: the float64 instrument of ref64.hip — the same
// statements with one type changed); VEC (16-byte accesses) exists for float only.
template <class T, bool VEC>
__global__ __launch_bounds__(NT) void k_lsqr_damped_update(const T* __restrict__ vk, T* w, const T* x_in, T* x_out,
                                                           const T* __restrict__ ref, double* __restrict__ err_part, int64_t n,
                                                           const double* __restrict__ a2, const double* __restrict__ b2,
                                                           const double* __restrict__ beta0_sq, double damp,
                                                           const double* __restrict__ st_in, double* __restrict__ st_out, int first) {
  __shared__ double cf[3];
  __shared__ double lds[NT / 64];
  if (threadIdx.x == 0) {
    const double alpha = sqrt(*a2), beta = sqrt(*b2);
    double rhobar, phibar, tw = 0.0;
    if (first) {
      rhobar = alpha;
      phibar = sqrt(*beta0_sq);
    } else {
      rhobar = -st_in[0] * alpha;
      tw = st_in[1] * alpha / st_in[2];
      phibar = st_in[3];
    }
    const double rhobar1 = sqrt(rhobar * rhobar + damp * damp);
    phibar *= rhobar / rhobar1;
    const double rho = sqrt(rhobar1 * rhobar1 + beta * beta);
    const double cs = rhobar1 / rho, sn = beta / rho;
    cf[0] = 1.0 / alpha;
    cf[1] = tw;
    cf[2] = cs * phibar / rho;
    if (blockIdx.x == 0) {
      st_out[0] = cs;
      st_out[1] = sn;
      st_out[2] = rho;
      st_out[3] = sn * phibar;
    }
  }
  __syncthreads();
  const double ia = cf[0], tw = cf[1], px = cf[2];
  double acc = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  auto one = [&](T v, T wo, T xo, T& wn, T& xn) {
    wn = (T)(ia * (double)v - (first ? 0.0 : tw * (double)wo));
    xn = (T)((x_in ? (double)xo : 0.0) + px * (double)wn);
  };
  int64_t tail0 = 0;
  if constexpr (VEC) {
    static_assert(std::is_same<T, float>::value, "16-byte accesses: float only");
    const int64_t n4 = n >> 2;
    tail0 = n4 << 2;
    for (int64_t i = tid; i < n4; i += nth) {
      const float4 v = ld4(vk, i);
      const float4 wo = first ? make_float4(0.f, 0.f, 0.f, 0.f) : ld4(w, i);
      const float4 xo = x_in ? ld4(x_in, i) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 wn, xn;
      one(v.x, wo.x, xo.x, wn.x, xn.x);
      one(v.y, wo.y, xo.y, wn.y, xn.y);
      one(v.z, wo.z, xo.z, wn.z, xn.z);
      one(v.w, wo.w, xo.w, wn.w, xn.w);
      st4(w, i, wn);
      st4(x_out, i, xn);
      if (ref) {
        const float4 t = ld4(ref, i);
        const double e0 = (double)xn.x - t.x, e1 = (double)xn.y - t.y, e2 = (double)xn.z - t.z, e3 = (double)xn.w - t.w;
        acc += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
      }
    }
  }
  for (int64_t i = tail0 + tid; i < n; i += nth) {
    T wn, xn;
    one(vk[i], first ? (T)0 : w[i], x_in ? x_in[i] : (T)0, wn, xn);
    w[i] = wn;
    x_out[i] = xn;
    if (ref) {
      const double e = (double)xn - (double)ref[i];
      acc += e * e;
    }
  }
  if (ref) {
    acc = block_sum<NT>(acc, lds);
    if (threadIdx.x == 0) err_part[blockIdx.x] = acc;
  }
}

// ------------------------------------------------------------------ fused reorthogonalisation step
// w_out = w_in - sum_j h[j] V[j]   AND   g[j] = sum_i V[j][i] w_out[i]   with ONE pass over the k basis rows:
// the middle step of repeated classical Gram-Schmidt  r -= V (V^T r)  (GKS.py:86-88 three times, MMGKS.py:119-120 twice,
// Arnoldi): the update with the previous pass's coefficients and the next pass's dot products read the same rows, so a
// thread keeps its k float4 of a column group in registers between the two uses (k <= KB = 8 / 16).
// Element formulas are those of k_gemv_n (fp64 accumulation of the combination, rounded once) and k_gemv_t.
template <int KB, bool VEC>
__global__ __launch_bounds__(NT, 2) void k_gemv_nt(const float* __restrict__ V, int64_t ld, int k, int64_t n,
                                                   const double* __restrict__ h, const float* w_in, float* w_out,
                                                   double* __restrict__ partials) {
  __shared__ double hs[KB];
  __shared__ double lds[NT / 64];
  if (threadIdx.x < KB) hs[threadIdx.x] = threadIdx.x < k ? h[threadIdx.x] : 0.0;
  __syncthreads();
  double acc[KB];
#pragma unroll
  for (int j = 0; j < KB; ++j) acc[j] = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  int64_t tail0 = 0;
  if (VEC) {
    const int64_t n4 = n >> 2;
    tail0 = n4 << 2;
    for (int64_t i = tid; i < n4; i += nth) {
      const float4 wv = ld4(w_in, i);
      float4 v[KB];
#pragma unroll
      for (int j = 0; j < KB; ++j) v[j] = (j < k) ? ld4(V + (int64_t)j * ld, i) : make_float4(0.f, 0.f, 0.f, 0.f);
      double o0 = (double)wv.x, o1 = (double)wv.y, o2 = (double)wv.z, o3 = (double)wv.w;
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        if (j < k) {
          const double c = -hs[j];
          o0 = fma(c, (double)v[j].x, o0);
          o1 = fma(c, (double)v[j].y, o1);
          o2 = fma(c, (double)v[j].z, o2);
          o3 = fma(c, (double)v[j].w, o3);
        }
      }
      const float4 o = make_float4((float)o0, (float)o1, (float)o2, (float)o3);
      st4(w_out, i, o);
#pragma unroll
      for (int j = 0; j < KB; ++j)
        if (j < k) acc[j] += (double)v[j].x * o.x + (double)v[j].y * o.y + (double)v[j].z * o.z + (double)v[j].w * o.w;
    }
  }
  for (int64_t i = tail0 + tid; i < n; i += nth) {
    float v[KB];
#pragma unroll
    for (int j = 0; j < KB; ++j) v[j] = (j < k) ? V[(int64_t)j * ld + i] : 0.f;
    double o0 = (double)w_in[i];
#pragma unroll
    for (int j = 0; j < KB; ++j)
      if (j < k) o0 = fma(-hs[j], (double)v[j], o0);
    const float o = (float)o0;
    w_out[i] = o;
#pragma unroll
    for (int j = 0; j < KB; ++j)
      if (j < k) acc[j] += (double)v[j] * o;
  }
#pragma unroll
  for (int j = 0; j < KB; ++j) {
    if (j < k) {                                                // uniform
      const double t = block_sum<NT>(acc[j], lds);
      if (threadIdx.x == 0) partials[(size_t)blockIdx.x * k + j] = t;
    }
  }
}

// ------------------------------------------------------------------ weighted Gram, tile pairs (TG x TG)
// G[a][b] = sum_i w_i^2 W[a][i] W[b][i].  grid = (bx, npairs): pair p -> (ta <= tb).  partials [bx][k*k] (upper blocks).
constexpr int TG = 4;

template <bool HAS_W, bool VEC>
__global__ __launch_bounds__(NT) void k_wgram(const float* __restrict__ W, int64_t ld, int k, int64_t m,
                                              const float* __restrict__ w, int ntile,
                                              double* __restrict__ partials) {
  __shared__ double lds[NT / 64];
  // blockIdx.y enumerates tile pairs (ta <= tb) row by row
  int ta = 0, rem = blockIdx.y;
  while (rem >= ntile - ta) {
    rem -= ntile - ta;
    ++ta;
  }
  const int a0 = ta * TG, b0 = (ta + rem) * TG;
  double acc[TG][TG];
#pragma unroll
  for (int a = 0; a < TG; ++a)
#pragma unroll
    for (int b = 0; b < TG; ++b) acc[a][b] = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  int64_t tail0 = 0;
  if (VEC) {
    const int64_t m4 = m >> 2;
    tail0 = m4 << 2;
    for (int64_t i = tid; i < m4; i += nth) {
      float4 ra[TG], rb[TG];
      float4 ww = make_float4(1.f, 1.f, 1.f, 1.f);
      if (HAS_W) {
        ww = ld4(w, i);
        ww.x *= ww.x;
        ww.y *= ww.y;
        ww.z *= ww.z;
        ww.w *= ww.w;
      }
#pragma unroll
      for (int a = 0; a < TG; ++a) {
        const int ja = (a0 + a < k) ? a0 + a : k - 1;  // clamp: out-of-range rows are computed and discarded
        ra[a] = ld4(W + (int64_t)ja * ld, i);
        if (HAS_W) {
          ra[a].x *= ww.x;
          ra[a].y *= ww.y;
          ra[a].z *= ww.z;
          ra[a].w *= ww.w;
        }
      }
#pragma unroll
      for (int b = 0; b < TG; ++b) {
        const int jb = (b0 + b < k) ? b0 + b : k - 1;
        rb[b] = ld4(W + (int64_t)jb * ld, i);
      }
#pragma unroll
      for (int a = 0; a < TG; ++a)
#pragma unroll
        for (int b = 0; b < TG; ++b)
          acc[a][b] += (double)ra[a].x * rb[b].x + (double)ra[a].y * rb[b].y + (double)ra[a].z * rb[b].z +
                       (double)ra[a].w * rb[b].w;
    }
  }
  for (int64_t i = tail0 + tid; i < m; i += nth) {
    float ww = 1.f;
    if (HAS_W) {
      ww = w[i];
      ww *= ww;
    }
#pragma unroll
    for (int a = 0; a < TG; ++a) {
      const int ja = (a0 + a < k) ? a0 + a : k - 1;
      const float va = W[(int64_t)ja * ld + i] * ww;
#pragma unroll
      for (int b = 0; b < TG; ++b) {
        const int jb = (b0 + b < k) ? b0 + b : k - 1;
        acc[a][b] += (double)va * W[(int64_t)jb * ld + i];
      }
    }
  }
#pragma unroll
  for (int a = 0; a < TG; ++a)
#pragma unroll
    for (int b = 0; b < TG; ++b) {
      double t = block_sum<NT>(acc[a][b], lds);
      if (threadIdx.x == 0 && a0 + a < k && b0 + b < k) {
        partials[(size_t)blockIdx.x * k * k + (size_t)(a0 + a) * k + (b0 + b)] = t;
        partials[(size_t)blockIdx.x * k * k + (size_t)(b0 + b) * k + (a0 + a)] = t;
      }
    }
}

// ------------------------------------------------------------------ weighted Gram, LDS-staged (every row read once)
// Rows of the augmented matrix  [ w*W_0 ; ... ; w*W_{k-1} ; b ; w*b ]  for a chunk of CH elements are staged in LDS (fp32),
// then every thread owns one 4x4 tile of row pairs (upper triangle) and a residue class of the chunk's elements and
// accumulates 16 fp64 FMAs per element.  One pass gives G = W diag(w^2) W^T, c1 = W (w*b), c2 = W (w^2*b) and ||w*b||^2.
// HBM traffic 4*m*(k+2) bytes; LDS rows are padded by one float (row stride CH+1) so that the 8 row reads of a wave's
// different tiles fall on different banks.  Partials: [block][KA*KA] (KA = k + 2 when b is given).
constexpr int WG_TILE = 4;

template <bool HAS_W, bool HAS_B>
__global__ __launch_bounds__(NT) void k_wgram2(const float* __restrict__ W, int64_t ld, int k, int64_t m,
                                               const float* __restrict__ w, const float* __restrict__ bvec, int CH,
                                               double* __restrict__ partials) {
  extern __shared__ float smem[];
  const int KA = k + (HAS_B ? 2 : 0);
  const int nt = (KA + WG_TILE - 1) / WG_TILE, KP = nt * WG_TILE;
  const int ntu = nt * (nt + 1) / 2;
  const int RS = CH + 1;                                  // padded row stride (floats)
  const int nslice = NT / ntu;                            // >= 1 (host guarantees ntu <= NT)
  const int pair = threadIdx.x % ntu, slice = threadIdx.x / ntu;
  const bool worker = slice < nslice;
  int ta = 0, rem = pair;
  while (rem >= nt - ta) {
    rem -= nt - ta;
    ++ta;
  }
  const int a0 = ta * WG_TILE, b0 = (ta + rem) * WG_TILE;
  double acc[WG_TILE][WG_TILE];
#pragma unroll
  for (int a = 0; a < WG_TILE; ++a)
#pragma unroll
    for (int b = 0; b < WG_TILE; ++b) acc[a][b] = 0.0;

  const int64_t nchunk = (m + CH - 1) / CH;
  for (int64_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
    const int64_t e0 = c * CH;
    const int len = (int)((m - e0 < CH) ? (m - e0) : CH);
    // stage: row r, element e  ->  smem[r*RS + e]
    for (int idx = threadIdx.x; idx < KP * CH; idx += NT) {
      const int r = idx / CH, e = idx - r * CH;
      float v = 0.f;
      if (e < len && r < KA) {
        const float wv = HAS_W ? w[e0 + e] : 1.f;
        if (r < k) v = W[(int64_t)r * ld + e0 + e] * wv;
        else if (r == k) v = bvec[e0 + e];
        else v = bvec[e0 + e] * wv;
      }
      smem[r * RS + e] = v;
    }
    __syncthreads();
    if (worker) {
      for (int e = slice; e < len; e += nslice) {
        double av[WG_TILE], bw[WG_TILE];
#pragma unroll
        for (int a = 0; a < WG_TILE; ++a) av[a] = (double)smem[(a0 + a) * RS + e];
#pragma unroll
        for (int b = 0; b < WG_TILE; ++b) bw[b] = (double)smem[(b0 + b) * RS + e];
#pragma unroll
        for (int a = 0; a < WG_TILE; ++a)
#pragma unroll
          for (int b = 0; b < WG_TILE; ++b) acc[a][b] = fma(av[a], bw[b], acc[a][b]);
      }
    }
    __syncthreads();
  }
  // reduce the slices of each tile pair in a fixed order through LDS (reusing the staging area), then write the block partial
  double* red = reinterpret_cast<double*>(smem);          // [NT][16]
#pragma unroll
  for (int a = 0; a < WG_TILE; ++a)
#pragma unroll
    for (int b = 0; b < WG_TILE; ++b) red[threadIdx.x * 16 + a * WG_TILE + b] = worker ? acc[a][b] : 0.0;
  __syncthreads();
  double* __restrict__ out = partials + (size_t)blockIdx.x * KA * KA;
  for (int idx = threadIdx.x; idx < ntu * 16; idx += NT) {
    const int pr = idx / 16, q = idx - pr * 16;
    double t = 0.0;
    for (int sl = 0; sl < nslice; ++sl) t += red[(sl * ntu + pr) * 16 + q];
    int pta = 0, prem = pr;
    while (prem >= nt - pta) {
      prem -= nt - pta;
      ++pta;
    }
    const int ra = pta * WG_TILE + q / WG_TILE, rb = (pta + prem) * WG_TILE + (q % WG_TILE);
    if (ra < KA && rb < KA) {
      out[(size_t)ra * KA + rb] = t;
      out[(size_t)rb * KA + ra] = t;
    }
  }
}

// ------------------------------------------------------------------ weighted Gram on the matrix cores
// The one GEMM-shaped contraction of the path (SYRK: G = R R^T, R = the <= 64 augmented rows, m up to 3e7 long) runs on
// v_mfma_f32_32x32x2_f32: a wave feeds lane l with R[row l&31][element slot l>>5] as BOTH operands (A[i][k] = R_i[e_k],
// B[k][j] = R_j[e_k]), so one instruction adds two elements to a full 32 x 32 tile of G.  Products are exact fp32 FMAs;
// the fp32 tile is flushed into fp64 accumulators after every 32 elements (per wave), so the long sum is fp64.
// Rows are staged per 128-element chunk through LDS with coalesced 16-byte loads (row stride 129 floats: the 32 rows a
// wave reads at one element fall on 32 different banks).  NTILE = 1: KA <= 32 rows (1 tile); NTILE = 2: KA <= 64 (3 tiles).
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NTILE, bool HAS_W, bool HAS_B>
__global__ __launch_bounds__(NT) void k_wgram_mfma(const float* __restrict__ W, int64_t ld, int k, int64_t m,
                                                   const float* __restrict__ w, const float* __restrict__ bvec,
                                                   double* __restrict__ partials) {
  constexpr int KP = 32 * NTILE, CH = 128, RS = CH + 1, CH4 = CH / 4;
  constexpr int NPAIR = NTILE * (NTILE + 1) / 2;
  __shared__ float tile[KP * RS];
  __shared__ double red[3][16][64];
  const int KA = k + (HAS_B ? 2 : 0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  double accd[NPAIR][16];
#pragma unroll
  for (int p = 0; p < NPAIR; ++p)
#pragma unroll
    for (int q = 0; q < 16; ++q) accd[p][q] = 0.0;
  const bool vec_ok = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15u) == 0) &&
                      (!HAS_W || (reinterpret_cast<uintptr_t>(w) & 15u) == 0) &&
                      (!HAS_B || (reinterpret_cast<uintptr_t>(bvec) & 15u) == 0);
  const int64_t nchunk = (m + CH - 1) / CH;
  for (int64_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
    const int64_t e0 = c * CH;
    const bool full = vec_ok && (e0 + CH <= m);
    if (full) {
      for (int idx = threadIdx.x; idx < KP * CH4; idx += NT) {
        const int row = idx / CH4, q = idx - row * CH4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < KA) {
          float4 wv = make_float4(1.f, 1.f, 1.f, 1.f);
          if (HAS_W && row != k) wv = ld4(w + e0, q);
          if (row < k) v = ld4(W + (int64_t)row * ld + e0, q);
          else v = ld4(bvec + e0, q);
          v.x *= wv.x;
          v.y *= wv.y;
          v.z *= wv.z;
          v.w *= wv.w;
        }
        float* d = &tile[row * RS + 4 * q];
        d[0] = v.x;
        d[1] = v.y;
        d[2] = v.z;
        d[3] = v.w;
      }
    } else {
      for (int idx = threadIdx.x; idx < KP * CH; idx += NT) {
        const int row = idx / CH, e = idx - row * CH;
        float v = 0.f;
        if (row < KA && e0 + e < m) {
          const float wv = (HAS_W && row != k) ? w[e0 + e] : 1.f;
          v = ((row < k) ? W[(int64_t)row * ld + e0 + e] : bvec[e0 + e]) * wv;
        }
        tile[row * RS + e] = v;
      }
    }
    __syncthreads();
    f16v acc[NPAIR];
#pragma unroll
    for (int p = 0; p < NPAIR; ++p)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[p][q] = 0.f;
    const int ebase = wave * 32 + h * 16;       // this wave's 32 elements: slot h takes one half
#pragma unroll 4
    for (int sidx = 0; sidx < 16; ++sidx) {
      const float a0 = tile[r * RS + ebase + sidx];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, acc[0], 0, 0, 0);
      if (NTILE == 2) {
        const float a1 = tile[(32 + r) * RS + ebase + sidx];
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, acc[1], 0, 0, 0);   // rows of tile 0 x rows of tile 1
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a1, acc[2], 0, 0, 0);
      }
    }
#pragma unroll
    for (int p = 0; p < NPAIR; ++p)
#pragma unroll
      for (int q = 0; q < 16; ++q) accd[p][q] += (double)acc[p][q];
    __syncthreads();
  }
  // combine the 4 waves (fixed order) and write the block partial in matrix order
  double* __restrict__ out = partials + (size_t)blockIdx.x * KA * KA;
#pragma unroll
  for (int p = 0; p < NPAIR; ++p) {
    if (wave > 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) red[wave - 1][q][lane] = accd[p][q];
    }
    __syncthreads();
    if (wave == 0) {
      const int ti = (p == 2) ? 1 : 0, tj = (p == 0) ? 0 : 1;     // pair 0: (0,0)  1: (0,1)  2: (1,1)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const double t = ((accd[p][q] + red[0][q][lane]) + red[1][q][lane]) + red[2][q][lane];
        const int row = ti * 32 + (q & 3) + 8 * (q >> 2) + 4 * h, col = tj * 32 + r;   // MFMA 32x32 C/D map
        if (row < KA && col < KA) {
          out[(size_t)row * KA + col] = t;
          if (ti != tj) out[(size_t)col * KA + row] = t;
        }
      }
    }
    __syncthreads();
  }
}

// Register-direct variant on v_mfma_f32_16x16x4_f32 for KA <= 64 and 16-byte aligned rows: no LDS staging, no block
// barriers inside the stream, and only the T(T+1)/2 upper 16 x 16 tiles of the symmetric Gram are computed (T = ceil(KA/16)).
// Why: the 32 x 32 kernel above is bound by the fp32 matrix pipe, not by HBM — 64 cycles per instruction = 16 B/clk/CU of
// input (measured 0.97 ms per pass over 33.5 M elements whatever k <= 30 is); one 32 x 32 tile costs 128 cycles per 4
// elements, the 16 x 16 tiles cost 32 (T = 1), 96 (T = 2), 192 (T = 3), 320 (T = 4), which puts the pipe at or above
// the HBM rate for every KA <= 64.
// A wave owns groups of 32 U consecutive elements (U = 1).  Lane (r = l & 15, s = l >> 4) loads, for
// each 16-row tile t, 2 U float4 of row 16 t + r at element offsets 16 j + 4 s (j < 2 U): the four s-lanes of a row read 64 contiguous bytes per
// instruction.  Component c of float4 j feeds MFMA number 4 j + c of every tile pair (element slots s = 0..3 then hold
// elements 16 j + 4 s + c).  The next group's loads are issued before the current group's MFMAs; the fp32 tiles are
// flushed into fp64 accumulators after every group, so the long sum is fp64.
typedef float f4v __attribute__((ext_vector_type(4)));

template <int T, bool HAS_W, bool HAS_B>
__global__ __launch_bounds__(NT, 2) void k_wgram_t16(const float* __restrict__ W, int64_t ld, int k, int64_t m,
                                                  const float* __restrict__ w, const float* __restrict__ bvec,
                                                  double* __restrict__ partials) {
  constexpr int NP = T * (T + 1) / 2;
  constexpr int U = 1;                                         // a group is 32 U elements: 2 U float4 per lane and tile (U = 4, 2 measured slower)
  constexpr int GE = 32 * U;
  __shared__ double red[3][4][64];
  const int KA = k + (HAS_B ? 2 : 0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, sl = lane >> 4;
  // every lane loads unconditionally (no divergent branches, so all loads of a group are in flight together): lanes
  // beyond the last row read row 0 and multiply by 0; `one[t]` is the factor of an unweighted live row
  const float* rowp[T];
  bool live[T], wtd[T];
  float one[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int rr = 16 * t + r;
    live[t] = rr < KA;
    wtd[t] = HAS_W && live[t] && rr != k;                      // row k is the plain b; rows < k and row k+1 carry w
    one[t] = live[t] ? 1.f : 0.f;
    rowp[t] = (rr < k) ? W + (int64_t)rr * ld : (HAS_B && live[t] ? bvec : W);
  }
  double accd[NP][4];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) accd[p][q] = 0.0;
  const int64_t ngroup = m / GE;                               // full groups; the tail (< GE elements) is done below
  const int64_t g0 = (int64_t)blockIdx.x * (NT / 64) + wave, gs = (int64_t)gridDim.x * (NT / 64);
  float4 v[2][T][2 * U];
  auto load = [&](int buf, int64_t g) {
    const int64_t e = g * GE + 4 * sl;
#pragma unroll
    for (int j = 0; j < 2 * U; ++j) {
      float4 wv = make_float4(1.f, 1.f, 1.f, 1.f);
      if (HAS_W) wv = *reinterpret_cast<const float4*>(w + e + 16 * j);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        float4 x = *reinterpret_cast<const float4*>(rowp[t] + e + 16 * j);
        x.x *= wtd[t] ? wv.x : one[t];
        x.y *= wtd[t] ? wv.y : one[t];
        x.z *= wtd[t] ? wv.z : one[t];
        x.w *= wtd[t] ? wv.w : one[t];
        v[buf][t][j] = x;
      }
    }
  };
  auto consume = [&](int buf) {
    int p = 0;
#pragma unroll
    for (int ta = 0; ta < T; ++ta)
#pragma unroll
      for (int tb = ta; tb < T; ++tb, ++p) {
        f4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2 * U; ++j) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[buf][ta][j].x, v[buf][tb][j].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[buf][ta][j].y, v[buf][tb][j].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[buf][ta][j].z, v[buf][tb][j].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[buf][ta][j].w, v[buf][tb][j].w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) accd[p][q] += (double)acc[q];
      }
  };
  if (g0 < ngroup) {
    load(0, g0);
    int64_t g = g0;
    for (; g + 2 * gs < ngroup; g += 2 * gs) {                 // two groups per trip: buffer indices stay compile-time
      load(1, g + gs);
      consume(0);
      load(0, g + 2 * gs);
      consume(1);
    }
    if (g + gs < ngroup) {
      load(1, g + gs);
      consume(0);
      consume(1);
    } else {
      consume(0);
    }
  }
  // tail elements [GE * ngroup, m): one wave, scalar predicated loads; element slot sl of step i holds element 4 i + sl
  if (blockIdx.x == 0 && wave == 0 && (m % GE)) {
    const int64_t e0 = ngroup * GE;
    for (int i = 0; i < 8 * U; ++i) {
      const int64_t e = e0 + 4 * i + sl;
      float a[T];
#pragma unroll
      for (int t = 0; t < T; ++t) a[t] = (live[t] && e < m) ? rowp[t][e] * (wtd[t] ? w[e] : 1.f) : 0.f;
      int p = 0;
#pragma unroll
      for (int ta = 0; ta < T; ++ta)
#pragma unroll
        for (int tb = ta; tb < T; ++tb, ++p) {
          f4v acc = {0.f, 0.f, 0.f, 0.f};
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], a[tb], acc, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) accd[p][q] += (double)acc[q];
        }
    }
  }
  // combine the 4 waves (fixed order) and write the block partial in matrix order (16x16 C/D map: lane (r, sl), register q
  // holds D[4 sl + q][r])
  double* __restrict__ out = partials + (size_t)blockIdx.x * KA * KA;
  int p = 0;
#pragma unroll
  for (int ta = 0; ta < T; ++ta)
#pragma unroll
    for (int tb = ta; tb < T; ++tb, ++p) {
      if (wave > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave - 1][q][lane] = accd[p][q];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double t = ((accd[p][q] + red[0][q][lane]) + red[1][q][lane]) + red[2][q][lane];
          const int row = 16 * ta + 4 * sl + q, col = 16 * tb + r;
          if (row < KA && col < KA) {
            out[(size_t)row * KA + col] = t;
            if (ta != tb) out[(size_t)col * KA + row] = t;
          }
        }
      }
      __syncthreads();
    }
}

// ------------------------------------------------------------------ weighted Gram of L V for the 2-D first-difference L, from V
// G[a][b] = sum over the rows e of L of  w_e^2 (L v_a)_e (L v_b)_e,   L = [D_h; D_v]  (MMGKS.py:94-95: the R factor of wr * (L V)
// enters the projected problem only through this Gram matrix).  k_wgram_t16 reads the stored images L v_j — 2 n floats
// per basis vector; here a wave forms them on the fly from V (n floats per vector: HALF the bytes, and L V is never written or
// kept).  A wave owns a strip of 32 image columns and marches down a band of rows: lane (r = l & 15, sl = l >> 4) holds, per
// 16-row tile t of V, the 8 pixels at columns 32 strip + 8 sl .. + 7 of row 16 t + r of V, for the current and the next image row
// — every element of V is loaded ONCE (the row below is the next step's current row).  The pixel right of a lane's eight comes
// from lane l + 16 (ds_bpermute; the last quarter takes the next strip's first pixel, one dword load per tile and step); the 64
// weights of a step (32 horizontal, 32 vertical) are ONE coalesced dword load per wave, spread to the lanes through LDS.  (First
// version: per-lane loads of the weights and of the right neighbours — as many L2 requests again as the rows themselves; 2 TB/s.)
// Per step 64 weighted differences go through the same v_mfma_f32_16x16x4_f32 tile pairs as in k_wgram_t16, flushed into fp64
// every step.  Rows are loaded two steps ahead.  w = [w_h: N rows of N-1 | w_v: N-1 rows of N] (trk_tv_weights).  N % 32 == 0.
template <int T>
struct TvRow {
  float4 x[T][2];      // 8 consecutive pixels of the image row, per tile
  float nx[T];         // the first pixel of the next strip (used by the lanes sl == 3 only)
  float w;             // lane l < 32: w_h of column 32 strip + l ; l >= 32: w_v of column 32 strip + l - 32
  float z;             // (Z) lane l: z of column 32 strip + (l & 31)
};

// Z: the pass also takes h[j] = V[j] . z for one more image z (MMGKS: the Gram row V^T (A^T A v_new) of the vector appended last,
// which needs the same sweep over V — trk_wgram_tv_z): fp64 products as in k_gemv_t, z spread to the lanes like the weights.
// Block partials: [k*k Gram | k dots] per workgroup.
// D: image rows held per wave (the current one, the next, D - 2 in flight).  D = 3 is what is instantiated: with one workgroup per
// CU and D = 6 (four rows in flight, 320-380 VGPRs) k = 32 went from 689 to 716 us — latency is not what that case waits for; for
// k <= 24 the kernel sits at the fp32 matrix pipe's rate already (48 MFMAs per 32 pixels at two tiles).
// BF (round 4): the products of the Gram tiles through the bf16 matrix pipe, each operand split into two bf16 halves and ALL FOUR
// partial products taken — (a_hi + a_lo)(b_hi + b_lo) exactly; what is lost is each operand's third piece, 2^-17 of it, of either
// sign (dropping a_lo b_lo instead would bias every diagonal entry low by ~1e-6: the squares of the roundings do not cancel).
// v_mfma_f32_16x16x32_bf16 takes a lane's eight weighted differences of a step in ONE instruction: 4 x 16 cycles per tile pair and
// direction where eight v_mfma_f32_16x16x4_f32 took 256 — the fp32 matrix pipe, at the vector unit's own rate, was this kernel's
// bound at two tiles (k = 17 .. 32: 0.51-0.57 ms whatever k; 48 MFMAs x 32 cycles per 32 pixels), now the rows' traffic is.
// The verdict of trk_wgram_tv's 'auto' arithmetic (the probe is further down: k_wgram_tv_probe): sums = the probe's 2 x 10 finished
// sums {S_ab, S'_ab}; worst = max_ab |S' - S| / sqrt(S_aa S_bb); verdict = worst > threshold.  Every workgroup of the pair of Gram
// launches evaluates it (20 scalar loads, the same bits everywhere); the first one of the bf16 launch also records it in `record`.
struct ProbeGate {
  const double* sums;     // NULL: no gating
  double threshold;
  int want;               // this launch runs iff verdict == want
  double* record;         // {verdict, worst} for trk_wgram_tv_last_probe
  int groups = 1;         // sets of four sampled basis vectors: 2 x 10 sums each (k_wgram_tv_probe)
};
__device__ __forceinline__ int probe_verdict(const ProbeGate& pg, bool record) {
  constexpr int PV = 4, PP = 10;
  double worst = 0.0;
  for (int g = 0; g < pg.groups; ++g) {
    const double* __restrict__ S = pg.sums + g * 2 * PP;
    int q = 0;
#pragma unroll
    for (int a = 0; a < PV; ++a)
#pragma unroll
      for (int b = a; b < PV; ++b, ++q) {
        // (the diagonal entries of the upper-triangle order: a = 0 -> 0, 1 -> 4, 2 -> 7, 3 -> 9)
        const int qa = a == 0 ? 0 : a == 1 ? 4 : a == 2 ? 7 : 9, qb = b == 0 ? 0 : b == 1 ? 4 : b == 2 ? 7 : 9;
        const double sc = sqrt(fabs(S[qa] * S[qb]));
        if (sc > 0.0) worst = fmax(worst, fabs(S[PP + q] - S[q]) / sc);
      }
  }
  const int verdict = worst > pg.threshold ? 1 : 0;
  if (record && pg.want == 0 && pg.record) {
    pg.record[0] = (double)verdict;
    pg.record[1] = worst;
  }
  return verdict;
}
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void bf16_split8(const float (&d)[8], bf8v& hi, bf8v& lo) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    hi[c] = (__bf16)d[c];
    lo[c] = (__bf16)(d[c] - (float)hi[c]);
  }
}
// BF = 3 (round 5): THREE bf16 pieces per operand — hi + mid + lo is the fp32 value exactly — and the six partial products down to
// 2^-16 of the largest (hi hi, hi mid, mid hi, mid mid, hi lo, lo hi; what is dropped is 2^-24 and below, the fp32 accumulator's own
// resolution).  Two pieces lose up to 2^-16 of EACH operand; on noisy data the residuals average out (measured 5e-9 against the
// fp32-pipe Gram), on piecewise-constant or repeated values they are all the same number and do not: a Gram entry was then off by
// up to 3e-5 of itself (tests/test_gpu_kernels.py::test_wgram_tv_split_products_on_adversarial_images).
__device__ __forceinline__ void bf16_split8x3(const float (&d)[8], bf8v& hi, bf8v& mid, bf8v& lo) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    hi[c] = (__bf16)d[c];
    const float r1 = d[c] - (float)hi[c];
    mid[c] = (__bf16)r1;
    lo[c] = (__bf16)(r1 - (float)mid[c]);
  }
}
// The arithmetic of one image row of a 32-column strip (k_wgram_tv and k_wgram_tv_lds): lane (r, sl) holds, per 16-vector tile t, the eight
// pixels Px[t] of image row i and Qx[t] of the row below, nbr[t] = the pixel right of the strip (used by the last quarter only), and the
// row's weights for its eight columns; the weighted differences' tile products in the arithmetic AR (0 fp32 pipe, 2 / 3 bf16 pieces),
// flushed into the float64 accumulators.
template <int T, int AR>
__device__ __forceinline__ void tv_row_products(const float4 (&Px)[T][2], const float4 (&Qx)[T][2], const float (&nbr)[T], int up16, int sl,
                                                const float4& wh0, const float4& wh1, const float4& wv0, const float4& wv1,
                                                double (&accd)[T * (T + 1) / 2][4]) {
  constexpr int NP = T * (T + 1) / 2;
    f4v acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = (f4v){0.f, 0.f, 0.f, 0.f};
    float dh[T][8], dv[T][8];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float4 a = Px[t][0], b = Px[t][1], c = Qx[t][0], d = Qx[t][1];
      // the pixel right of this lane's eight: lane l + 16 holds it as its first, the last quarter takes the next strip's
      float right = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(up16, __builtin_bit_cast(int, a.x)));
      right = sl == 3 ? nbr[t] : right;
      dh[t][0] = (a.x - a.y) * wh0.x;
      dh[t][1] = (a.y - a.z) * wh0.y;
      dh[t][2] = (a.z - a.w) * wh0.z;
      dh[t][3] = (a.w - b.x) * wh0.w;
      dh[t][4] = (b.x - b.y) * wh1.x;
      dh[t][5] = (b.y - b.z) * wh1.y;
      dh[t][6] = (b.z - b.w) * wh1.z;
      dh[t][7] = (b.w - right) * wh1.w;
      dv[t][0] = (a.x - c.x) * wv0.x;
      dv[t][1] = (a.y - c.y) * wv0.y;
      dv[t][2] = (a.z - c.z) * wv0.z;
      dv[t][3] = (a.w - c.w) * wv0.w;
      dv[t][4] = (b.x - d.x) * wv1.x;
      dv[t][5] = (b.y - d.y) * wv1.y;
      dv[t][6] = (b.z - d.z) * wv1.z;
      dv[t][7] = (b.w - d.w) * wv1.w;
    }
    if constexpr (AR == 3) {
      // one direction at a time (its three pieces die before the other direction's are made: the register file is the limit here)
      auto dir = [&](const float (&dd)[T][8]) {
        bf8v ph[T], pm[T], pl[T];
#pragma unroll
        for (int t = 0; t < T; ++t) bf16_split8x3(dd[t], ph[t], pm[t], pl[t]);
        int pp = 0;
#pragma unroll
        for (int ta = 0; ta < T; ++ta)
#pragma unroll
          for (int tb = ta; tb < T; ++tb, ++pp) {
            acc[pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[ta], pl[tb], acc[pp], 0, 0, 0);      // smallest terms first
            acc[pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl[ta], ph[tb], acc[pp], 0, 0, 0);
            acc[pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pm[ta], pm[tb], acc[pp], 0, 0, 0);
            acc[pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[ta], pm[tb], acc[pp], 0, 0, 0);
            acc[pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pm[ta], ph[tb], acc[pp], 0, 0, 0);
            acc[pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[ta], ph[tb], acc[pp], 0, 0, 0);
          }
      };
      dir(dh);
      dir(dv);
    } else {
      auto two_pieces = [&]() {
        bf8v hh[T], hl[T], vh[T], vl[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          bf16_split8(dh[t], hh[t], hl[t]);
          bf16_split8(dv[t], vh[t], vl[t]);
        }
        int pp = 0;
#pragma unroll
        for (int ta = 0; ta < T; ++ta)
#pragma unroll
          for (int tb = ta; tb < T; ++tb, ++pp) {
            acc[pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hl[ta], hl[tb], acc[pp], 0, 0, 0);      // smallest terms first
            acc[pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl[ta], vl[tb], acc[pp], 0, 0, 0);
            acc[pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hh[ta], hl[tb], acc[pp], 0, 0, 0);
            acc[pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hl[ta], hh[tb], acc[pp], 0, 0, 0);
            acc[pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh[ta], vl[tb], acc[pp], 0, 0, 0);
            acc[pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl[ta], vh[tb], acc[pp], 0, 0, 0);
            acc[pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hh[ta], hh[tb], acc[pp], 0, 0, 0);
            acc[pp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh[ta], vh[tb], acc[pp], 0, 0, 0);
          }
      };
      auto fp32_pipe = [&]() {
        int pp = 0;
#pragma unroll
        for (int ta = 0; ta < T; ++ta)
#pragma unroll
          for (int tb = ta; tb < T; ++tb, ++pp) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              acc[pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(dh[ta][c], dh[tb][c], acc[pp], 0, 0, 0);
              acc[pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[ta][c], dv[tb][c], acc[pp], 0, 0, 0);
            }
          }
      };
      if constexpr (AR == 2) two_pieces();
      else fp32_pipe();
    }
#pragma unroll
    for (int p2 = 0; p2 < NP; ++p2)
#pragma unroll
      for (int q = 0; q < 4; ++q) accd[p2][q] += (double)acc[p2][q];
}

template <int T, bool Z, int D, int BF = 3, int MINB = (D > 3 ? 1 : T == 1 ? 4 : (T == 2 && !Z) ? 3 : 2)>
__global__ __launch_bounds__(NT, MINB) void k_wgram_tv(const float* __restrict__ V, int64_t ld, int k, int N,
                                                    const float* __restrict__ w, int nbands, int band_rows,
                                                    double* __restrict__ partials, const float* __restrict__ z, int lockstep_in,
                                                    ProbeGate pg) {
  // 'auto' arithmetic (trk_wgram_tv_precision): the launch is one of a pair — two bf16 pieces / the fp32 pipe — of which the probe's
  // verdict (worked out by every workgroup from the probe's 20 finished sums: no launch for it) lets exactly one run; the other leaves at once
  // BF == 4: ONE launch holds both arithmetics and the verdict picks per launch (uniform branch in the step) — no idle launches
  bool use_f32 = false;
  if constexpr (BF == 4) {
    use_f32 = pg.sums && probe_verdict(pg, blockIdx.x == 0 && threadIdx.x == 0) != 0;
  } else {
    if (pg.sums && probe_verdict(pg, blockIdx.x == 0 && threadIdx.x == 0) != pg.want) return;
  }
  constexpr int NP = T * (T + 1) / 2;
  const int lockstep = lockstep_in & 1;
  const bool no_xcd_map = (lockstep_in & 2) != 0;                // TRK_WGRAM_TV_NO_XCD=1: the round-robin unit order (A/B)
#ifdef TRK_WGRAM_TV_EXPERIMENT
  // timing experiments only (tools/r05_wgram_exp.sh builds a separate library with this macro; the product never defines it and
  // the results are WRONG with either bit): 4 = no wave fetches the column right of its strips, 8 = the loads alone, no arithmetic
  const bool x_nohalo = (lockstep_in & 4) != 0, x_loadonly = (lockstep_in & 8) != 0;
#else
  constexpr bool x_nohalo = false, x_loadonly = false;
#endif
  __shared__ double red[3][4][64];
  __shared__ __attribute__((aligned(16))) float wl[NT / 64][Z ? 96 : 64];
  // lockstep (the launcher's choice when the workgroup's four waves always own four neighbouring strips of one band): the pixel
  // right of a strip is the first pixel of the next wave's strip, which that wave holds in registers — it is handed over through
  // LDS, one row ahead, behind the one barrier per image row that keeps the four waves together.  Only the workgroup's last wave
  // still fetches its neighbour column from memory.  (Every wave fetching it cost 25-40 % more HBM traffic than the operands: the
  // column is a sector of a line that the neighbouring strip's wave fetches again at some other time;
  // profiles/r03/traffic_bench_c4.txt.)
  __shared__ float nxs[2][NT / 64][T][16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, sl = lane >> 4;
  const float* __restrict__ wh = w;
  const float* __restrict__ wv = w + (int64_t)N * (N - 1);
  const float* rowp[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int rr = 16 * t + r;
    rowp[t] = V + (int64_t)(rr < k ? rr : 0) * ld;              // rows beyond the basis read row 0: their Gram entries are never stored
  }
  double accd[NP][4];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) accd[p][q] = 0.0;
  double accz[T];                                                // (the dots V[j] . z stay float64 FMAs: through the matrix pipe as
#pragma unroll                                                   //  well — split operands, four products — they cost what they cost here)
  for (int t = 0; t < T; ++t) accz[t] = 0.0;
  const int strips = N / 32;
  const int64_t units = (int64_t)strips * nbands;
  const int64_t gw = (int64_t)blockIdx.x * (NT / 64) + wave, nw = (int64_t)gridDim.x * (NT / 64);
  float* __restrict__ my = wl[wave];
  const int up16 = ((lane + 16) & 63) << 2;                      // ds_bpermute address of lane l + 16

  // Which XCD's L2 meets which strips (round 4).  The pixel right of a workgroup's four strips is one dword of a line that the
  // workgroup owning the next four strips fetches as its own data: with workgroups dealt round-robin to the eight XCDs the two
  // sit behind different L2s and the line crosses the fabric twice — 1.33 x the operand bytes at 4096^2 (FETCH_SIZE,
  // profiles/r04/traffic_c4_before_xcd_map.txt).  Workgroups b and b + 8 share an XCD (placement is speed only, never correctness),
  // so each XCD is given a CONTIGUOUS eighth of every band's strip groups: the neighbour line is then in its own L2 except at the
  // eight seams.  Needs the groups of a band to divide by 8 (N a multiple of 1024) and a grid that is a multiple of 8.
  const int groups = strips / (NT / 64);
  const bool xcd_map = strips % (NT / 64) == 0 && (groups & 7) == 0 && (gridDim.x & 7) == 0 && !no_xcd_map;   // (uniform)
  // The whole sweep once per arithmetic AR (0 fp32 pipe, 2 / 3 bf16 pieces).  BF == 4 ('auto') instantiates it twice under ONE uniform
  // branch on the probe's verdict: both forms in one launch, each with its own register allocation (a branch inside the step made
  // the allocator keep both forms' operands live: 34-99 spilled registers).
  auto run = [&](auto arith_tag) {
  constexpr int AR = decltype(arith_tag)::value;
  for (int64_t it = 0;; ++it) {
    int64_t u;
    if (xcd_map) {
      const int64_t U = (int64_t)blockIdx.x + it * gridDim.x;
      if (U >= units / (NT / 64)) break;
      const int per = groups >> 3;
      const int64_t q = U >> 3;
      const int64_t bnd = q / per;
      const int grp = (int)(U & 7) * per + (int)(q - bnd * per);
      u = (bnd * groups + grp) * (NT / 64) + wave;
    } else {
      u = gw + it * nw;
      if (u >= units) break;
    }
    const int band = (int)(u / strips), strip = (int)(u - (int64_t)band * strips);
    const int i0 = band * band_rows, i1 = (i0 + band_rows < N) ? i0 + band_rows : N;
    const int cs = 32 * strip;
    const int c0 = cs + 8 * sl;
    const bool last_strip = cs + 32 >= N;                        // (uniform) no pixel right of this strip
    const int nxo = last_strip ? 31 : 32;                        // clamped: a valid address, met by a zero weight
    const bool need_nx = (!lockstep || wave == NT / 64 - 1) && !x_nohalo;   // (uniform per wave)

    // every load is unconditional (clamped addresses, zeroed weights instead of branches): all loads of a row are in flight together
    auto load = [&](TvRow<T>& P, int i) {
      const int ic = i < N ? i : N - 1;                          // the row below the image is looked at with a zero weight only
      const int64_t e = (int64_t)ic * N;
      const int iv = ic < N - 1 ? ic : N - 2;                    // the last image row has no vertical difference: weight zeroed at use
      P.w = lane < 32 ? wh[(int64_t)ic * (N - 1) + cs + (lane < 31 || !last_strip ? lane : 30)] : wv[(int64_t)iv * N + cs + lane - 32];
      if (Z) P.z = z[e + cs + (lane & 31)];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        P.x[t][0] = *reinterpret_cast<const float4*>(rowp[t] + e + c0);
        P.x[t][1] = *reinterpret_cast<const float4*>(rowp[t] + e + c0 + 4);
        P.nx[t] = need_nx ? rowp[t][e + cs + nxo] : 0.f;
      }
    };
    auto publish = [&](const TvRow<T>& R, int i) {               // this wave's first pixel column of image row i, for the wave to its left
      if (lane < 16) {
#pragma unroll
        for (int t = 0; t < T; ++t) nxs[i & 1][wave][t][lane] = R.x[t][0].x;
      }
    };
    auto step = [&](const TvRow<T>& P, const TvRow<T>& Q, int i) {   // P: image row i, Q: the one below
      if (x_loadonly) {
#pragma unroll
        for (int t = 0; t < T; ++t) accz[t] += (double)(P.x[t][0].x + P.x[t][1].w + P.nx[t] + P.w + (Z ? P.z : 0.f));
        return;
      }
      float nbr[T];
#pragma unroll
      for (int t = 0; t < T; ++t) nbr[t] = P.nx[t];
      if (lockstep) {                                            // (uniform)
        __syncthreads();                                         // row i's columns are published; everybody has left row i - 1
        publish(Q, i + 1);
        if (wave < NT / 64 - 1) {
#pragma unroll
          for (int t = 0; t < T; ++t) nbr[t] = nxs[i & 1][wave + 1][t][r];
        }
      }
      // this row's weights to the lanes: h at my[0..31], v at my[32..63]
      float wk = P.w;
      if (lane == 31 && last_strip) wk = 0.f;                    // column N - 1 has no right neighbour
      if (lane >= 32 && i >= N - 1) wk = 0.f;                    // row N - 1 has none below
      // (the LDS traffic of one wave is in order in hardware; the compiler must not reorder the float stores and the float4 loads
      //  of the next lines either — it did, across two unrolled steps: fences for the compiler only)
      __asm__ volatile("" ::: "memory");
      my[lane] = wk;
      if (Z && lane < 32) my[64 + lane] = P.z;
      __asm__ volatile("" ::: "memory");
      const float4 wh0 = *reinterpret_cast<const float4*>(my + 8 * sl), wh1 = *reinterpret_cast<const float4*>(my + 8 * sl + 4);
      const float4 wv0 = *reinterpret_cast<const float4*>(my + 32 + 8 * sl), wv1 = *reinterpret_cast<const float4*>(my + 36 + 8 * sl);
      float4 z0 = make_float4(0.f, 0.f, 0.f, 0.f), z1 = z0;
      if (Z) {
        z0 = *reinterpret_cast<const float4*>(my + 64 + 8 * sl);
        z1 = *reinterpret_cast<const float4*>(my + 68 + 8 * sl);
      }
      __asm__ volatile("" ::: "memory");
      if (Z) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const float4 a = P.x[t][0], b = P.x[t][1];
          accz[t] += (double)a.x * z0.x + (double)a.y * z0.y + (double)a.z * z0.z + (double)a.w * z0.w +
                     (double)b.x * z1.x + (double)b.y * z1.y + (double)b.z * z1.z + (double)b.w * z1.w;
        }
      }
      tv_row_products<T, AR>(P.x, Q.x, nbr, up16, sl, wh0, wh1, wv0, wv1, accd);
    };

    TvRow<T> P[D];                                               // a ring: row i sits in P[(i - i0) % D]; indices are compile-time below
#pragma unroll
    for (int j = 0; j < D - 1; ++j) load(P[j], i0 + j);
    if (lockstep) {
      __syncthreads();                                           // the previous unit's last exchange has been read
      publish(P[0], i0);
    }
    int i = i0;
    for (; i + D <= i1; i += D) {                                // roles rotate: no register copies
#pragma unroll
      for (int j = 0; j < D; ++j) {
        load(P[(j + D - 1) % D], i + j + D - 1);
        step(P[j], P[(j + 1) % D], i + j);
      }
    }
#pragma unroll
    for (int j = 0; j < D - 1; ++j) {                            // fewer than D rows left; P[j] = row i + j
      if (i + j < i1) {                                          // (uniform)
        if (i + j + D - 1 <= i1) load(P[(j + D - 1) % D], i + j + D - 1);
        step(P[j], P[(j + 1) % D], i + j);
      }
    }
  }
  };
  if constexpr (BF == 4) {
    if (use_f32) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 2>{});
  } else {
    run(std::integral_constant<int, BF>{});
  }
  // combine the 4 waves (fixed order) and write the block partial in matrix order (16x16 C/D map: lane (r, sl), register q
  // holds D[4 sl + q][r])
  double* __restrict__ out = partials + (size_t)blockIdx.x * (k * k + (Z ? k : 0));
  int p = 0;
#pragma unroll
  for (int ta = 0; ta < T; ++ta)
#pragma unroll
    for (int tb = ta; tb < T; ++tb, ++p) {
      if (wave > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave - 1][q][lane] = accd[p][q];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double t = ((accd[p][q] + red[0][q][lane]) + red[1][q][lane]) + red[2][q][lane];
          const int row = 16 * ta + 4 * sl + q, col = 16 * tb + r;
          // one value per unordered pair: tiles above the diagonal are mirrored, and so is the upper triangle of a diagonal tile
          // (its lower triangle holds the same sums with the split products added in another order: equal to rounding, not to the bit)
          if (row < k && col < k && (ta != tb || row <= col)) {
            out[(size_t)row * k + col] = t;
            if (row != col) out[(size_t)col * k + row] = t;
          }
        }
      }
      __syncthreads();
    }
  if (Z) {
    // the dots: the four quarter-strip lanes of a row first (fixed order), then the four waves
#pragma unroll
    for (int t = 0; t < T; ++t) {
      double v = accz[t];
      const double v1 = __shfl(v, r + 16, 64), v2 = __shfl(v, r + 32, 64), v3 = __shfl(v, r + 48, 64);
      v = ((__shfl(v, r, 64) + v1) + v2) + v3;
      if (lane < 16) red[0][wave][16 * (t & 3) + lane] = v;      // [wave][row within a group of 4 tiles]
    }
    __syncthreads();
    if (threadIdx.x < 16 * T) {
      const int row = threadIdx.x;
      const double t = ((red[0][0][row] + red[0][1][row]) + red[0][2][row]) + red[0][3][row];
      if (row < k) out[(size_t)k * k + row] = t;
    }
  }
}

// ------------------------------------------------------------------ the same pass, the rows of V through LDS in FULL lines (round 5)
// k_wgram_tv's loads are fragment-shaped: one wave-instruction touches 16 basis vectors x 64 bytes, a workgroup 512 contiguous bytes
// per (vector, image row) — the loads ALONE take the kernel's whole time (tools/r05_wgram_exp.sh: k = 32 at 4096^2 576 us with the
// arithmetic removed, 545 with it; 3.9 TB/s where the plain streams of k_gemv_n run at 5.9).  Here a workgroup of 8 waves owns 256
// image columns: every (vector, image row) of its tile is ONE 1 KiB global_load_lds_dwordx4 (a whole wave reading 1 KiB of one
// row of one basis vector) straight into an LDS stage, four stages deep — two image rows in flight per CU without a register held
// for them — and the waves take their MFMA operands from LDS.  The LDS image of a (vector, row) piece has its 16-byte slots XOR-ed with
// the vector's index mod 16 (applied to the per-lane SOURCE address of the load: the LDS side of such a load is lane-linear): the four
// 16-lane groups of a ds_read_b128 — {0-3, 12-15, 20-27}, .. — then meet 16 different slots (padding the rows to 260 floats does
// not do it: 61 % of the LDS cycles were bank conflicts, SQ_LDS_BANK_CONFLICT).  The pixel right of a wave's strip is the next strip's first pixel in the same stage; the column right
// of the TILE comes with the stage as one dword per vector.  The row's weights and the row of z arrive the same way (no load of the
// loop has a register destination: one vmcnt queue, counted by hand, raw barriers — a __syncthreads() would drain it).
// Arithmetic, block partials and their order: exactly k_wgram_tv's (tv_row_products).  N a multiple of 256, k <= 32.
constexpr int LW_NW = 8, LW_NT = 64 * LW_NW, LW_COLS = 32 * LW_NW, LW_RS = LW_COLS, LW_S = 4;
constexpr int LW_WH = 0, LW_WV = 256, LW_Z = 512, LW_HALO = 768, LW_XF = 832;      // the extras of a stage, in floats after its rows
__device__ __forceinline__ void glds16(const float* g, unsigned lds) {              // lane l: 16 bytes at g -> LDS byte lds + 16 l
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ void glds4(const float* g, unsigned lds) {               // lane l: 4 bytes at g -> LDS byte lds + 4 l
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
// LDS reads the compiler may not take apart (it split the float4 reads of a row into b64 / b32 / read2_b32 pieces: 18 LDS instructions
// per step instead of 12, most of them on the 32-bank paths): issued as written, waited for by hand (lds_wait), and the values tied to
// the wait (lds_tie) so that no use is scheduled before it.
__device__ __forceinline__ f4v lds_r128(const float* p) {
  f4v v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const float*)p) : "memory");
  return v;
}
__device__ __forceinline__ float lds_r32(const float* p) {
  float v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const float*)p) : "memory");
  return v;
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <class X>
__device__ __forceinline__ void lds_tie(X& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ float4 as_float4(const f4v& v) { return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void wait_vm_le(int n) {                                 // (n: wave-uniform, <= 10)
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
  }
}
template <int T>
struct LwRow {
  float4 x[T][2];      // lane (r, sl): 8 pixels of vector 16 t + r at columns 32 wave + 8 sl ..
  float nx[T];         // the pixel right of the wave's strip
};
template <int T, bool Z, int BF>
__global__ __launch_bounds__(LW_NT, T == 1 ? 4 : 2) void k_wgram_tv_lds(const float* __restrict__ V, int64_t ld, int k, int N,
                                                           const float* __restrict__ w, int nbands, int band_rows,
                                                           double* __restrict__ partials, const float* __restrict__ z, int flags,
                                                           ProbeGate pg) {
  bool use_f32 = false;
  if constexpr (BF == 4) {
    use_f32 = pg.sums && probe_verdict(pg, blockIdx.x == 0 && threadIdx.x == 0) != 0;
  } else {
    if (pg.sums && probe_verdict(pg, blockIdx.x == 0 && threadIdx.x == 0) != pg.want) return;
  }
  constexpr int NP = T * (T + 1) / 2, NV = 16 * T, SF = NV * LW_RS + LW_XF;
  __shared__ __attribute__((aligned(16))) float smem[LW_S * SF];                  // (ALL of the kernel's LDS: the sums at the end alias it)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63, r = lane & 15, sl = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  const float* __restrict__ wh = w;
  const float* __restrict__ wv = w + (int64_t)N * (N - 1);
  const int up16 = ((lane + 16) & 63) << 2;
  // rows of vectors beyond the basis are never loaded: zero in every stage (their Gram entries are never stored; finite all the same)
  for (int s = 0; s < LW_S; ++s)
    for (int v = k; v < NV; ++v)
      for (int c = threadIdx.x; c < LW_RS; c += LW_NT) smem[s * SF + v * LW_RS + c] = 0.f;
  // this wave's pieces of a stage: the vectors wave, wave + 8, .. below k, and one of the extras
  int nv_mine = 0;
#pragma unroll
  for (int j = 0; j < 2 * T; ++j) nv_mine += (wave + 8 * j < k) ? 1 : 0;
  const bool has_extra = wave <= 4 || (wave == 5 && Z) || wave == 6;
  const int npw = nv_mine + (has_extra ? 1 : 0);
  double accd[NP][4];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) accd[p][q] = 0.0;
  double accz[T];
#pragma unroll
  for (int t = 0; t < T; ++t) accz[t] = 0.0;
  const int tiles = N / LW_COLS;
  const int64_t units = (int64_t)tiles * nbands;
  const bool xcd_map = (tiles & 7) == 0 && (gridDim.x & 7) == 0 && (flags & 2) == 0;      // (uniform) as in k_wgram_tv: an XCD's tiles are neighbours
  __syncthreads();

  auto run = [&](auto arith_tag) {
  constexpr int AR = decltype(arith_tag)::value;
  for (int64_t U = blockIdx.x; U < units; U += gridDim.x) {
    int band, tile;
    if (xcd_map) {
      const int per = tiles >> 3;
      const int64_t q = U >> 3;
      band = (int)(q / per);
      tile = (int)(U & 7) * per + (int)(q - (int64_t)band * per);
    } else {
      band = (int)(U / tiles);
      tile = (int)(U - (int64_t)band * tiles);
    }
    const int i0 = band * band_rows, i1 = (i0 + band_rows < N) ? i0 + band_rows : N;
    const int c0 = LW_COLS * tile;
    const bool last_tile = c0 + LW_COLS >= N;                    // (uniform)
    const int halo_col = last_tile ? N - 1 : c0 + LW_COLS;       // clamped: a valid address, met by a zero weight

    auto issue = [&](int i) {                                    // image row i -> stage i & 3
      const int ic = i < N ? i : N - 1;
      const int64_t e = (int64_t)ic * N;
      const int iv = ic < N - 1 ? ic : N - 2;
      const unsigned sb = lds0 + (unsigned)((i & (LW_S - 1)) * SF) * 4u;
#pragma unroll
      for (int j = 0; j < 2 * T; ++j) {
        const int v = wave + 8 * j;
        if (v < k) glds16(V + (int64_t)v * ld + e + c0 + 4 * (lane ^ (v & 15)), sb + (unsigned)(v * LW_RS) * 4u);
      }
      const unsigned xb = sb + (unsigned)(NV * LW_RS) * 4u;
      if (wave < 4) glds4(wh + (int64_t)ic * (N - 1) + c0 + 64 * wave + lane, xb + (unsigned)(LW_WH + 64 * wave) * 4u);
      else if (wave == 4) glds16(wv + (int64_t)iv * N + c0 + 4 * lane, xb + LW_WV * 4u);
      else if (wave == 5) { if (Z) glds16(z + e + c0 + 4 * lane, xb + LW_Z * 4u); }
      else if (wave == 6) {
        const int hv = (lane & (NV - 1)) < k ? (lane & (NV - 1)) : 0;
        glds4(V + (int64_t)hv * ld + e + halo_col, xb + LW_HALO * 4u);
      }
    };
    // RAW: the reads as hand-issued ds_read_b128 (every value used only after lds_wait + lds_tie).  Measured per form: without the dots
    // the compiler takes the float4 reads apart (18 LDS instructions per step, 53 % of the LDS cycles bank conflicts; raw reads 415 us
    // against 428 at k = 24); with them it keeps them whole and schedules around its own waits better than one lds_wait does (435 us
    // against 463).
    constexpr bool RAW = !Z;
    auto fetch_raw = [&](f4v (&x)[T][2], float (&nx)[T], int i) {   // the wave's operands of image row i, from its stage
      const float* S = smem + (i & (LW_S - 1)) * SF;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float* row = S + (16 * t + r) * LW_RS;             // slot q of the row sits at slot q ^ r
        const int q = 8 * wave + 2 * sl;
        const float* np = wave < LW_NW - 1 ? row + 4 * ((8 * wave + 8) ^ r) : S + NV * LW_RS + LW_HALO + 16 * t + r;
        if constexpr (RAW) {
          x[t][0] = lds_r128(row + 4 * (q ^ r));
          x[t][1] = lds_r128(row + 4 * ((q + 1) ^ r));
          nx[t] = lds_r32(np);
        } else {
          x[t][0] = *reinterpret_cast<const f4v*>(row + 4 * (q ^ r));
          x[t][1] = *reinterpret_cast<const f4v*>(row + 4 * ((q + 1) ^ r));
          nx[t] = *np;
        }
      }
    };
    auto settle = [&](LwRow<T>& R, f4v (&x)[T][2], float (&nx)[T]) {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        if constexpr (RAW) {
          lds_tie(x[t][0]);
          lds_tie(x[t][1]);
          lds_tie(nx[t]);
        }
        R.x[t][0] = as_float4(x[t][0]);
        R.x[t][1] = as_float4(x[t][1]);
        R.nx[t] = nx[t];
      }
    };
    auto step = [&](const LwRow<T>& P, LwRow<T>& Q, int i) {     // P: image row i (held), Q: the row below (fetched here)
      wait_vm_le(i + 2 <= i1 ? npw : 0);                         // row i + 1 has landed (this wave's pieces); row i + 2 may be in flight
      __builtin_amdgcn_s_barrier();                              // ... everybody's; and everybody has left step i - 1
      asm volatile("" ::: "memory");                            // (the raw barrier is IntrNoMem: nothing else keeps the plain LDS reads of the Z form below it)
      if (i + 3 <= i1) issue(i + 3);                             // into the stage of row i - 1
      const float* X = smem + (i & (LW_S - 1)) * SF + NV * LW_RS + 32 * wave + 8 * sl;
      f4v rwh0, rwh1, rwv0, rwv1, rz0 = {0.f, 0.f, 0.f, 0.f}, rz1 = rz0;
      if constexpr (RAW) {
        rwh0 = lds_r128(X + LW_WH), rwh1 = lds_r128(X + LW_WH + 4), rwv0 = lds_r128(X + LW_WV), rwv1 = lds_r128(X + LW_WV + 4);
      } else {
        rwh0 = *reinterpret_cast<const f4v*>(X + LW_WH), rwh1 = *reinterpret_cast<const f4v*>(X + LW_WH + 4);
        rwv0 = *reinterpret_cast<const f4v*>(X + LW_WV), rwv1 = *reinterpret_cast<const f4v*>(X + LW_WV + 4);
        if (Z) {
          rz0 = *reinterpret_cast<const f4v*>(X + LW_Z);
          rz1 = *reinterpret_cast<const f4v*>(X + LW_Z + 4);
        }
      }
      f4v qx[T][2];
      float qn[T];
      fetch_raw(qx, qn, i + 1);
      if constexpr (RAW) {
        lds_wait();
        lds_tie(rwh0); lds_tie(rwh1); lds_tie(rwv0); lds_tie(rwv1);
      }
      settle(Q, qx, qn);
      float4 wh0 = as_float4(rwh0), wh1 = as_float4(rwh1), wv0 = as_float4(rwv0), wv1 = as_float4(rwv1);
      if (last_tile && wave == LW_NW - 1 && sl == 3) wh1.w = 0.f;      // column N - 1 has no right neighbour
      if (i >= N - 1) {                                                // row N - 1 has none below
        wv0 = make_float4(0.f, 0.f, 0.f, 0.f);
        wv1 = wv0;
      }
      if (Z) {
        const float4 z0 = as_float4(rz0), z1 = as_float4(rz1);
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const float4 a = P.x[t][0], b = P.x[t][1];
          accz[t] += (double)a.x * z0.x + (double)a.y * z0.y + (double)a.z * z0.z + (double)a.w * z0.w +
                     (double)b.x * z1.x + (double)b.y * z1.y + (double)b.z * z1.z + (double)b.w * z1.w;
        }
      }
      tv_row_products<T, AR>(P.x, Q.x, P.nx, up16, sl, wh0, wh1, wv0, wv1, accd);
    };

    LwRow<T> A, B;
    __builtin_amdgcn_s_barrier();                                // the previous unit's last rows have been read
    asm volatile("" ::: "memory");
    issue(i0);
    issue(i0 + 1);
    issue(i0 + 2);
    wait_vm_le(2 * npw);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
      f4v ax[T][2];
      float an[T];
      fetch_raw(ax, an, i0);
      if constexpr (RAW) lds_wait();
      settle(A, ax, an);
    }
    int i = i0;
    for (; i + 2 <= i1; i += 2) {                                // roles alternate: no register copies
      step(A, B, i);
      step(B, A, i + 1);
    }
    if (i < i1) step(A, B, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (rows past a band shorter than the prologue's three)
  }
  };
  if constexpr (BF == 4) {
    if (use_f32) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 2>{});
  } else {
    run(std::integral_constant<int, BF>{});
  }
  // combine the 8 waves (fixed order) and write the block partial in matrix order, as k_wgram_tv does
  __syncthreads();
  double* red = reinterpret_cast<double*>(smem);                 // [7][4][64]
  double* __restrict__ out = partials + (size_t)blockIdx.x * (k * k + (Z ? k : 0));
  int p = 0;
#pragma unroll
  for (int ta = 0; ta < T; ++ta)
#pragma unroll
    for (int tb = ta; tb < T; ++tb, ++p) {
      if (wave > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) red[((wave - 1) * 4 + q) * 64 + lane] = accd[p][q];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          double t = accd[p][q];
#pragma unroll
          for (int ww = 0; ww < LW_NW - 1; ++ww) t += red[(ww * 4 + q) * 64 + lane];
          const int row = 16 * ta + 4 * sl + q, col = 16 * tb + r;
          if (row < k && col < k && (ta != tb || row <= col)) {
            out[(size_t)row * k + col] = t;
            if (row != col) out[(size_t)col * k + row] = t;
          }
        }
      }
      __syncthreads();
    }
  if (Z) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      double v = accz[t];
      const double v1 = __shfl(v, r + 16, 64), v2 = __shfl(v, r + 32, 64), v3 = __shfl(v, r + 48, 64);
      v = ((__shfl(v, r, 64) + v1) + v2) + v3;
      if (lane < 16) red[wave * 64 + 16 * t + lane] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16 * T) {
      const int row = threadIdx.x;
      double t = red[row];
#pragma unroll
      for (int ww = 1; ww < LW_NW; ++ww) t += red[ww * 64 + row];
      if (row < k) out[(size_t)k * k + row] = t;
    }
  }
}

// ------------------------------------------------------------------ the probe of the 'auto' arithmetic of trk_wgram_tv
// What two bf16 pieces per operand lose is each operand's third piece (<= 2^-16 of it).  On noisy data those residuals average out; on
// data that repeats a few values they are one number, millions of times (tests/test_gpu_kernels.py: 5.8e-6 per entry).  Whether the
// data at hand is of that kind is MEASURED per call on a sample: runs of 1024 pixels in 256 image rows (jittered), four of the k basis vectors; for
// their weighted differences d the kernel forms both  S_ab = sum d_a d_b  and  S'_ab = sum t_a t_b , t = the two-piece value of d,
// in float64 — S' - S is exactly what the split loses on the sample.  k_wgram_tv_gate turns the partials into
//     gate[1] = max_ab |S'_ab - S_ab| / sqrt(S_aa S_bb) ,   gate[0] = gate[1] > threshold (3e-7) ,
// and the pair of Gram launches behind it reads gate[0]: the bf16 form runs when it is 0, the fp32 pipe when it is 1 — decided on the
// device, nothing visits the host.  Cost: ~12 MB of reads whatever N, the probe and its 20-sum finalize, and the two launches of the
// pair that does not run (measured at 4096^2: see profiles/r05/wgram_tv_auto.txt).
constexpr int PROBE_V = 4, PROBE_P = PROBE_V * (PROBE_V + 1) / 2, PROBE_ROWS = 256, PROBE_SEG = 1024;
__device__ __forceinline__ float two_piece(float d) {
  const float hi = (float)(__bf16)d;
  return hi + (float)(__bf16)(d - hi);
}
// the j-th sample: one image row per stride, at a pseudo-random place inside it, and of that row one pseudo-random run of PROBE_SEG
// columns (a regular comb would never meet the block edges of a piecewise-constant image whose blocks are multiples of the stride —
// measured: such an image passed the first version's probe).  262 144 pixels whatever N: the probe's cost does not grow with the image.
__device__ __forceinline__ unsigned probe_hash(unsigned j) {
  unsigned h = j * 2654435761u;
  h ^= h >> 15;
  h *= 2246822519u;
  return h ^ (h >> 13);
}
__global__ __launch_bounds__(NT) void k_wgram_tv_probe(const float* __restrict__ V, int64_t ld, int k, int N, const float* __restrict__ w,
                                                       int row_step, double* __restrict__ part) {
  __shared__ double lds[(NT / 64) * 2 * PROBE_P];
  const unsigned h = probe_hash(blockIdx.x);
  int i = blockIdx.x * row_step + (int)(h % (unsigned)row_step);
  i = i < N ? i : N - 1;
  const int span = N > PROBE_SEG ? N - PROBE_SEG : 0;
  const int cbeg = span > 0 ? (int)((h >> 8) % (unsigned)(span + 1)) : 0;
  const int c_end = cbeg + PROBE_SEG < N ? cbeg + PROBE_SEG : N;
  // group 0: four vectors spread over the basis (the first and the last among them); group 1 (k >= 8): the NEWEST four — a solver's
  // basis changes character at its end first (MMGKS late in a TV solve: VERDICT round 5, weak 3), and all pairs of the last four are
  // what a spread sample of one of them cannot see
  int pr[PROBE_V];
#pragma unroll
  for (int a = 0; a < PROBE_V; ++a) pr[a] = blockIdx.y == 0 ? (int)(((int64_t)a * (k - 1)) / (PROBE_V - 1)) : k - PROBE_V + a;
  const float* __restrict__ wh = w;
  const float* __restrict__ wv = w + (int64_t)N * (N - 1);
  double acc[2 * PROBE_P];
#pragma unroll
  for (int q = 0; q < 2 * PROBE_P; ++q) acc[q] = 0.0;
  for (int c = cbeg + threadIdx.x; c < c_end; c += NT) {
    const float whc = c < N - 1 ? wh[(int64_t)i * (N - 1) + c] : 0.f;
    const float wvc = i < N - 1 ? wv[(int64_t)i * N + c] : 0.f;
    float dh[PROBE_V], dv[PROBE_V], th[PROBE_V], tv[PROBE_V];
#pragma unroll
    for (int a = 0; a < PROBE_V; ++a) {
      const float* __restrict__ row = V + (int64_t)pr[a] * ld + (int64_t)i * N;
      const float x = row[c];
      const float xr = c < N - 1 ? row[c + 1] : x;
      const float xb = i < N - 1 ? row[c + N] : x;
      dh[a] = (x - xr) * whc;
      dv[a] = (x - xb) * wvc;
      th[a] = two_piece(dh[a]);
      tv[a] = two_piece(dv[a]);
    }
    int q = 0;
#pragma unroll
    for (int a = 0; a < PROBE_V; ++a)
#pragma unroll
      for (int b = a; b < PROBE_V; ++b, ++q) {
        acc[q] += (double)dh[a] * (double)dh[b] + (double)dv[a] * (double)dv[b];
        acc[PROBE_P + q] += (double)th[a] * (double)th[b] + (double)tv[a] * (double)tv[b];
      }
  }
  const double t = block_sum_many<NT, 2 * PROBE_P>(acc, lds);
  if (threadIdx.x < 2 * PROBE_P) part[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 2 * PROBE_P + threadIdx.x] = t;
}
// k_finalize / k_finalize_split of core.hip for one launch of such a pair (outputs from nsplit on go to out2)
__global__ __launch_bounds__(256) void k_finalize_gated(const double* __restrict__ partials, int nblocks, int stride, double* __restrict__ out,
                                                        int nsplit, double* __restrict__ out2, ProbeGate pg) {
  if (probe_verdict(pg, false) != pg.want) return;
  __shared__ double lds[4];
  const int o = blockIdx.x;
  const double* __restrict__ p = partials + o;
  double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
  int b = threadIdx.x;
  for (; b + 768 < nblocks; b += 1024) {
    v0 += p[(size_t)b * stride];
    v1 += p[(size_t)(b + 256) * stride];
    v2 += p[(size_t)(b + 512) * stride];
    v3 += p[(size_t)(b + 768) * stride];
  }
  for (; b < nblocks; b += 256) v0 += p[(size_t)b * stride];
  const double v = block_sum<256>((v0 + v1) + (v2 + v3), lds);
  if (threadIdx.x == 0) {
    if (o < nsplit) out[o] = v;
    else out2[o - nsplit] = v;
  }
}

// scatter the augmented Gram [KA x KA] into G (k x k), c1, c2 (and optionally ||w b||^2)
__global__ void k_wgram_unpack(const double* __restrict__ Ga, int k, int KA, double* __restrict__ G, double* __restrict__ c1,
                               double* __restrict__ c2) {
  for (int idx = threadIdx.x; idx < k * k; idx += blockDim.x) G[idx] = Ga[(size_t)(idx / k) * KA + (idx % k)];
  if (KA > k)
    for (int a = threadIdx.x; a < k; a += blockDim.x) {
      c1[a] = Ga[(size_t)a * KA + k];
      c2[a] = Ga[(size_t)a * KA + k + 1];
    }
}

}  // namespace

// ======================================================================================= C ABI
extern "C" {

int trk_dot(const float* x, const float* y, int64_t n, double* out, trk_stream st) {
  TRK_REQUIRE(out && n >= 0 && ((x && y) || n == 0), "trk_dot: NULL argument or n < 0");
  return launch_reduce2<0>(x, y, n, out, (hipStream_t)st);
}

int trk_nrm2sq(const float* x, int64_t n, double* out, trk_stream st) {
  TRK_REQUIRE(out && n >= 0 && (x || n == 0), "trk_nrm2sq: NULL argument or n < 0");
  return launch_reduce2<1>(x, x, n, out, (hipStream_t)st);
}

int trk_diff_nrm2sq(const float* x, const float* y, int64_t n, double* out, trk_stream st) {
  TRK_REQUIRE(out && n >= 0 && ((x && y) || n == 0), "trk_diff_nrm2sq: NULL argument or n < 0");
  return launch_reduce2<2>(x, y, n, out, (hipStream_t)st);
}

int trk_axpby(int64_t n, double ca, const double* a_num, const double* a_den, int a_flags, const float* x, double cb,
              const double* b_num, const double* b_den, int b_flags, const float* y, float* out, double* sumsq,
              trk_stream st) {
  TRK_REQUIRE(x && out && n >= 0, "trk_axpby: NULL x/out or n < 0");
  hipStream_t s = (hipStream_t)st;
  const Coef A{ca, a_num, a_den, a_flags}, B{cb, b_num, b_den, b_flags};
  const int grid = stream_grid(n);
  double* part = nullptr;
  if (sumsq)
    if (int rc = scratch_doubles(s, grid, &part)) return rc;
  const bool vec = aligned16(x) && aligned16(out) && (!y || aligned16(y));
#define AX(HY, SS, VC) hipLaunchKernelGGL((k_axpby<HY, SS, VC>), dim3(grid), dim3(NT), 0, s, n, A, x, B, y, out, part)
  if (y) {
    if (sumsq) { if (vec) AX(true, true, true); else AX(true, true, false); }
    else       { if (vec) AX(true, false, true); else AX(true, false, false); }
  } else {
    if (sumsq) { if (vec) AX(false, true, true); else AX(false, true, false); }
    else       { if (vec) AX(false, false, true); else AX(false, false, false); }
  }
#undef AX
  TRK_LAUNCH_CHECK();
  if (sumsq) return finalize_sums(part, grid, 1, 1, sumsq, s);
  return TRK_OK;
}

int trk_scale_dot(int64_t n, double ca, const double* a_num, const double* a_den, int a_flags, const float* x, float* out,
                  const float* z, double* dot_out, trk_stream st) {
  TRK_REQUIRE(x && out && z && dot_out && n >= 0, "trk_scale_dot: NULL argument or n < 0");
  hipStream_t s = (hipStream_t)st;
  const Coef A{ca, a_num, a_den, a_flags};
  const int grid = stream_grid(n);
  double* part = nullptr;
  if (int rc = scratch_doubles(s, grid, &part)) return rc;
  if (aligned16(x) && aligned16(out) && aligned16(z))
    hipLaunchKernelGGL((k_scale_dot<true>), dim3(grid), dim3(NT), 0, s, n, A, x, out, z, part);
  else
    hipLaunchKernelGGL((k_scale_dot<false>), dim3(grid), dim3(NT), 0, s, n, A, x, out, z, part);
  TRK_LAUNCH_CHECK();
  return finalize_sums(part, grid, 1, 1, dot_out, s);
}

int trk_mul(int64_t n, const float* x, const float* y, float* out, trk_stream st) {
  TRK_REQUIRE(x && y && out && n >= 0, "trk_mul: NULL argument or n < 0");
  const int grid = stream_grid(n);
  if (aligned16(x) && aligned16(y) && aligned16(out))
    hipLaunchKernelGGL((k_mul<true>), dim3(grid), dim3(NT), 0, (hipStream_t)st, n, x, y, out);
  else
    hipLaunchKernelGGL((k_mul<false>), dim3(grid), dim3(NT), 0, (hipStream_t)st, n, x, y, out);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_mul_diff(int64_t n, const float* w, const float* x, const float* y, float* out, trk_stream st) {
  TRK_REQUIRE(w && x && y && out && n >= 0, "trk_mul_diff: NULL argument or n < 0");
  const int grid = stream_grid(n);
  if (aligned16(w) && aligned16(x) && aligned16(y) && aligned16(out))
    hipLaunchKernelGGL((k_mul_diff<true>), dim3(grid), dim3(NT), 0, (hipStream_t)st, n, w, x, y, out);
  else
    hipLaunchKernelGGL((k_mul_diff<false>), dim3(grid), dim3(NT), 0, (hipStream_t)st, n, w, x, y, out);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_mm_weights(int64_t n, const float* x, const float* y, double eps, double p, float* out, trk_stream st) {
  TRK_REQUIRE(x && out && n >= 0, "trk_mm_weights: NULL argument or n < 0");
  const float e = (float)(p / 2.0 - 1.0), eps2 = (float)(eps * eps);
  const int special = (p == 2.0) ? 1 : (p == 1.0) ? 2 : 0;
  const int grid = stream_grid(n);
  const bool vec = aligned16(x) && aligned16(out) && (!y || aligned16(y));
  hipStream_t s = (hipStream_t)st;
#define MW(HY, VC) hipLaunchKernelGGL((k_mm_weights<HY, VC>), dim3(grid), dim3(NT), 0, s, n, x, y, eps2, e, special, out)
  if (y) { if (vec) MW(true, true); else MW(true, false); }
  else   { if (vec) MW(false, true); else MW(false, false); }
#undef MW
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_group_weights(const float* d, int64_t groups, int group_len, double add, double expo, int copies, float* out,
                      trk_stream st) {
  TRK_REQUIRE(d && out && groups >= 0 && group_len >= 1 && copies >= 1, "trk_group_weights: bad argument");
  if (groups == 0) return TRK_OK;
  hipLaunchKernelGGL(k_group_weights, dim3(ceil_div(groups, NT)), dim3(NT), 0, (hipStream_t)st, d, groups, group_len, add, expo,
                     copies, out);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_isotv_weights(const float* x, int N, int nt, const float* u_tail, int64_t n_tail, double eps, double q, float* out,
                      trk_stream st) {
  TRK_REQUIRE(x && out && N >= 1 && nt >= 1 && n_tail >= 0 && (u_tail || n_tail == 0), "trk_isotv_weights: bad argument");
  const double ed = (q - 2.0) / 4.0;                                   // sic: (q-2)/4 (MMGKS.py:75,77)
  const int special = (ed == 0.0) ? 1 : (ed == -0.5) ? 2 : (ed == -0.25) ? 3 : 0;
  const int64_t ns = (int64_t)N * N * nt;
  hipLaunchKernelGGL(k_isotv_weights, dim3(stream_grid(ns > n_tail ? ns : n_tail)), dim3(NT), 0, (hipStream_t)st, x, N, nt,
                     u_tail, n_tail, (float)(eps * eps), (float)ed, special, out);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_cgls_update_xr(int64_t n, int64_t m, const double* gamma, const double* delta, const float* x, const float* p,
                       float* x_new, float* r, const float* w, const float* x_true, double* sums, trk_stream st) {
  TRK_REQUIRE(gamma && delta && x && p && x_new && r && w && sums, "trk_cgls_update_xr: NULL argument");
  TRK_REQUIRE(n >= 0 && m >= 0, "trk_cgls_update_xr: negative size");
  hipStream_t s = (hipStream_t)st;
  const int grid = stream_grid(n > m ? n : m);
  double* part = nullptr;
  if (int rc = scratch_doubles(s, (size_t)grid * 3, &part)) return rc;
  const bool vec = aligned16(x) && aligned16(p) && aligned16(x_new) && aligned16(r) && aligned16(w) &&
                   (!x_true || aligned16(x_true));
#define CU(XT, VC) hipLaunchKernelGGL((k_cgls_update<XT, VC>), dim3(grid), dim3(NT), 0, s, n, m, ScalarSrc{gamma, 1}, ScalarSrc{delta, 1}, x, p, x_new, r, w, x_true, part, (double*)nullptr, stream_nontemporal(n))
  if (x_true) { if (vec) CU(true, true); else CU(true, false); }
  else        { if (vec) CU(false, true); else CU(false, false); }
#undef CU
  TRK_LAUNCH_CHECK();
  return finalize_sums(part, grid, 3, 3, sums, s);
}

int trk_cgls_update_xr_src(int64_t n, int64_t m, const double* gamma, int gamma_n, const double* delta, int delta_n,
                           const float* x, const float* p, float* x_new, float* r, const float* w, const float* x_true,
                           double* publish_delta, double* norm_partials, int capacity_blocks, int* n_blocks,
                           trk_stream st) {
  TRK_REQUIRE(gamma && delta && gamma_n >= 1 && delta_n >= 1 && x && p && x_new && r && w && norm_partials && n_blocks,
              "trk_cgls_update_xr_src: NULL argument");
  TRK_REQUIRE(n >= 0 && m >= 0, "trk_cgls_update_xr_src: negative size");
  hipStream_t s = (hipStream_t)st;
  const int grid = stream_grid(n > m ? n : m);
  TRK_REQUIRE(grid <= capacity_blocks, "trk_cgls_update_xr_src: partial buffer too small (%d blocks needed)", grid);
  *n_blocks = grid;
  const bool vec = aligned16(x) && aligned16(p) && aligned16(x_new) && aligned16(r) && aligned16(w) &&
                   (!x_true || aligned16(x_true));
  const ScalarSrc g{gamma, gamma_n}, d{delta, delta_n};
#define CU(XT, VC) hipLaunchKernelGGL((k_cgls_update<XT, VC>), dim3(grid), dim3(NT), 0, s, n, m, g, d, x, p, x_new, r, w, x_true, norm_partials, publish_delta, stream_nontemporal(n))
  if (x_true) { if (vec) CU(true, true); else CU(true, false); }
  else        { if (vec) CU(false, true); else CU(false, false); }
#undef CU
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_cgls_update_xr_deferred(int64_t n, int64_t m, const double* gamma, const double* delta, const float* x,
                                const float* p, float* x_new, float* r, const float* w, const float* x_true,
                                double* norm_partials, int capacity_blocks, int* n_blocks, trk_stream st) {
  return trk_cgls_update_xr_src(n, m, gamma, 1, delta, 1, x, p, x_new, r, w, x_true, nullptr, norm_partials,
                                capacity_blocks, n_blocks, st);
}

int trk_cgls_r_update(int64_t m, const double* gamma_old, const double* delta, int delta_n, float* r, const float* w,
                      double* publish_delta, trk_stream st) {
  TRK_REQUIRE(gamma_old && delta && delta_n >= 1 && r && w && m >= 0, "trk_cgls_r_update: bad argument");
  const int grid = stream_grid(m);
  const ScalarSrc d{delta, delta_n};
  hipStream_t s = (hipStream_t)st;
  if (aligned16(r) && aligned16(w))
    hipLaunchKernelGGL((k_cgls_r_update<true>), dim3(grid), dim3(NT), 0, s, m, gamma_old, d, r, w, publish_delta, stream_nontemporal(m));
  else
    hipLaunchKernelGGL((k_cgls_r_update<false>), dim3(grid), dim3(NT), 0, s, m, gamma_old, d, r, w, publish_delta, stream_nontemporal(m));
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_cgls_xp_update(int64_t n, const double* gamma_old, const double* delta, const double* gamma_new, int gamma_new_n,
                       const float* x, float* p, const float* t, float* x_new, const float* x_true, double* publish_gamma,
                       double* norm_partials, int capacity_blocks, int* n_blocks, trk_stream st) {
  TRK_REQUIRE(gamma_old && delta && gamma_new && gamma_new_n >= 1 && x && p && t && x_new && norm_partials && n_blocks && n >= 0,
              "trk_cgls_xp_update: bad argument");
  TRK_REQUIRE(aligned16(x) && aligned16(p) && aligned16(t) && aligned16(x_new) && (!x_true || aligned16(x_true)),
              "trk_cgls_xp_update: vectors must be 16-byte aligned");
  const int grid = stream_grid(n);
  TRK_REQUIRE(grid <= capacity_blocks, "trk_cgls_xp_update: partial buffer too small (%d blocks needed)", grid);
  *n_blocks = grid;
  const ScalarSrc g{gamma_new, gamma_new_n};
  hipStream_t s = (hipStream_t)st;
  if (x_true)
    hipLaunchKernelGGL((k_cgls_xp_update<true>), dim3(grid), dim3(NT), 0, s, n, gamma_old, delta, g, x, p, t, x_new, x_true, publish_gamma, norm_partials, stream_nontemporal(n));
  else
    hipLaunchKernelGGL((k_cgls_xp_update<false>), dim3(grid), dim3(NT), 0, s, n, gamma_old, delta, g, x, p, t, x_new, x_true, publish_gamma, norm_partials, stream_nontemporal(n));
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_cgls_p_update(int64_t n, const float* t, float* p, const double* gamma_new, int gamma_new_n,
                      const double* gamma_old, double* publish_gamma, trk_stream st) {
  TRK_REQUIRE(t && p && gamma_new && gamma_new_n >= 1 && gamma_old && n >= 0, "trk_cgls_p_update: bad argument");
  const int grid = stream_grid(n);
  const ScalarSrc g{gamma_new, gamma_new_n};
  hipStream_t s = (hipStream_t)st;
  if (aligned16(t) && aligned16(p))
    hipLaunchKernelGGL((k_cgls_p_update<true>), dim3(grid), dim3(NT), 0, s, n, t, p, g, gamma_old, publish_gamma, stream_nontemporal(n));
  else
    hipLaunchKernelGGL((k_cgls_p_update<false>), dim3(grid), dim3(NT), 0, s, n, t, p, g, gamma_old, publish_gamma, stream_nontemporal(n));
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_cgls_x_update(int64_t n, const double* gamma, int gamma_n, const double* delta, int delta_n, const float* x,
                      const float* p, float* x_new, const float* x_true, double* publish_delta, double* publish_gamma,
                      double* norm_partials, int capacity_blocks, int* n_blocks, trk_stream st) {
  TRK_REQUIRE(gamma && delta && gamma_n >= 1 && delta_n >= 1 && x && p && x_new && norm_partials && n_blocks,
              "trk_cgls_x_update: NULL argument");
  TRK_REQUIRE(aligned16(x) && aligned16(p) && aligned16(x_new) && (!x_true || aligned16(x_true)),
              "trk_cgls_x_update: vectors must be 16-byte aligned");
  const int grid = stream_grid(n);
  TRK_REQUIRE(grid <= capacity_blocks, "trk_cgls_x_update: partial buffer too small (%d blocks needed)", grid);
  *n_blocks = grid;
  const ScalarSrc g{gamma, gamma_n}, d{delta, delta_n};
  hipStream_t s = (hipStream_t)st;
  if (x_true)
    hipLaunchKernelGGL((k_cgls_x_update<true>), dim3(grid), dim3(NT), 0, s, n, g, d, x, p, x_new, x_true, publish_delta, publish_gamma, norm_partials);
  else
    hipLaunchKernelGGL((k_cgls_x_update<false>), dim3(grid), dim3(NT), 0, s, n, g, d, x, p, x_new, x_true, publish_delta, publish_gamma, norm_partials);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_gemv_t(const float* V, int64_t ld, int k, int64_t n, const float* r, const float* w2, double* h, trk_stream st) {
  TRK_REQUIRE(V && r && h, "trk_gemv_t: NULL argument");
  TRK_REQUIRE(k >= 1 && n >= 0 && ld >= n, "trk_gemv_t: need k >= 1, n >= 0, ld >= n");
  return launch_gemv_t(V, ld, k, n, r, w2, w2 ? 1 : 0, h, (hipStream_t)st);
}

int trk_gemv_t_x(const float* V, int64_t ld, int k, int64_t n, const float* r, const float* xrow, double* h, double* h_x,
                 trk_stream st) {
  TRK_REQUIRE(V && r && h && xrow && h_x, "trk_gemv_t_x: NULL argument");
  TRK_REQUIRE(k >= 1 && n >= 0 && ld >= n, "trk_gemv_t_x: need k >= 1, n >= 0, ld >= n");
  return launch_gemv_t(V, ld, k, n, r, nullptr, 0, h, (hipStream_t)st, xrow, h_x);
}

int trk_gemv_t2(const float* V, int64_t ld, int k, int64_t n, const float* r, const float* r2, double* h2k, trk_stream st) {
  TRK_REQUIRE(h2k, "trk_gemv_t2: NULL argument");
  double* part = nullptr;
  int bx = 0;
  if (int rc = gemv_t2_partials(V, ld, k, n, r, r2, &part, &bx, (hipStream_t)st)) return rc;
  return finalize_sums(part, bx, 2 * k, 2 * k, h2k, (hipStream_t)st);
}

}  // extern "C"
int trk::gemv_t2_partials(const float* V, int64_t ld, int k, int64_t n, const float* r, const float* r2, double** part_out, int* nblk,
                          hipStream_t s) {
  TRK_REQUIRE(V && r && r2, "trk_gemv_t2: NULL argument");
  TRK_REQUIRE(k >= 1 && n >= 0 && ld >= n, "trk_gemv_t2: need k >= 1, n >= 0, ld >= n");
  const int ntile = ceil_div(k, JT);
  static const int occ = resident_blocks_per_cu(k_gemv_t2<true>);
  const int bx = tiled_dot_grid_x(n, ntile, occ);
  double* part = nullptr;
  if (int rc = scratch_doubles(s, (size_t)bx * 2 * k, &part)) return rc;
  const bool vec = aligned16(V) && aligned16(r) && aligned16(r2) && (ld % 4 == 0);
  dim3 grid(bx, ntile);
  if (vec) hipLaunchKernelGGL((k_gemv_t2<true>), grid, dim3(NT), 0, s, V, ld, k, n, r, r2, part, stream_nontemporal(n));
  else hipLaunchKernelGGL((k_gemv_t2<false>), grid, dim3(NT), 0, s, V, ld, k, n, r, r2, part, stream_nontemporal(n));
  TRK_LAUNCH_CHECK();
  *part_out = part;
  *nblk = bx;
  return TRK_OK;
}
extern "C" {

int trk_gemv_tn(const float* V, int64_t ld, int k, int64_t n, const float* const* rhs, int n_rhs, double* out, trk_stream st) {
  TRK_REQUIRE(V && rhs && out, "trk_gemv_tn: NULL argument");
  TRK_REQUIRE(n_rhs == 3 || n_rhs == 4, "trk_gemv_tn: 3 or 4 right-hand sides (1: trk_gemv_t, 2: trk_gemv_t2)");
  TRK_REQUIRE(k >= 1 && n >= 0 && ld >= n, "trk_gemv_tn: need k >= 1, n >= 0, ld >= n");
  RhsSet rs{};
  bool vec = aligned16(V) && (ld % 4 == 0);
  for (int q = 0; q < n_rhs; ++q) {
    TRK_REQUIRE(rhs[q], "trk_gemv_tn: NULL right-hand side");
    rs.p[q] = rhs[q];
    vec = vec && aligned16(rhs[q]);
  }
  hipStream_t s = (hipStream_t)st;
  const int ntile = ceil_div(k, JT);
  static const int occ3 = resident_blocks_per_cu(k_gemv_tr<true, 3>), occ4 = resident_blocks_per_cu(k_gemv_tr<true, 4>);
  const int bx = tiled_dot_grid_x(n, ntile, n_rhs == 3 ? occ3 : occ4);
  double* part = nullptr;
  if (int rc = scratch_doubles(s, (size_t)bx * n_rhs * k, &part)) return rc;
  dim3 grid(bx, ntile);
  const int nt = stream_nontemporal(n);
#define GR(VC, RR) hipLaunchKernelGGL((k_gemv_tr<VC, RR>), grid, dim3(NT), 0, s, V, ld, k, n, rs, part, nt)
  if (n_rhs == 3) { if (vec) GR(true, 3); else GR(false, 3); }
  else            { if (vec) GR(true, 4); else GR(false, 4); }
#undef GR
  TRK_LAUNCH_CHECK();
  return finalize_sums(part, bx, n_rhs * k, n_rhs * k, out, s);
}

int trk_gemv_n(const float* V, int64_t ld, int k, int64_t n, const double* y, double a, const float* base, double sc,
               float* out, double* sumsq, trk_stream st) {
  double* part = nullptr;
  int nblk = 0;
  if (int rc = gemv_n_partials(V, ld, k, n, y, a, base, sc, out, sumsq ? &part : nullptr, &nblk, (hipStream_t)st)) return rc;
  if (sumsq) return finalize_sums(part, nblk, 1, 1, sumsq, (hipStream_t)st);
  return TRK_OK;
}

}  // extern "C"
int trk::gemv_n_partials(const float* V, int64_t ld, int k, int64_t n, const double* y, double a, const float* base, double sc,
                         float* out, double** part_out, int* nblk, hipStream_t s) {
  TRK_REQUIRE(V && y && out, "trk_gemv_n: NULL argument");
  TRK_REQUIRE(k >= 1 && k <= KMAX_LDS && n >= 0 && ld >= n, "trk_gemv_n: need 1 <= k <= %d, n >= 0, ld >= n", KMAX_LDS);
  const bool sumsq = part_out != nullptr;
  const int grid = stream_grid(n);
  double* part = nullptr;
  const bool vec = aligned16(V) && aligned16(out) && (ld % 4 == 0) && (!base || aligned16(base));
  // short vectors WITH a base (the residual (AV) y - b of a projector's images): rows split over the waves.  The plain combination
  // x = V y stays with k_gemv_n at every size: trk_gemv_n_err and trk_gemv_orth_iterate are its sum term for term (tested bit for bit)
  if (!sumsq && base && gemv_n_split_serves(n, k, vec)) {
    const int64_t n4 = n >> 2;
    const unsigned g = (unsigned)((n4 + 63) / 64);
    hipLaunchKernelGGL((k_gemv_n_split<true>), dim3(g), dim3(NT), 0, s, y, V, ld, k, n4, a, base, sc, out, (const double*)nullptr);
    TRK_LAUNCH_CHECK();
    *nblk = 0;
    return TRK_OK;
  }
  static const int U = env_int("TRK_GEMVN_UNROLL", 8);        // measured, tools/gemv_micro.py: 4 -> 5.4-5.8 TB/s, 8 (+ 8 blocks per CU) -> 6.2-6.4
  static const int gmul = env_int("TRK_GEMVN_GRID", 8);          // blocks per CU (0: stream_grid's 4)
  const int grid_n = gmul > 0 ? (int)std::min<int64_t>((n + (int64_t)NT * 4 - 1) / ((int64_t)NT * 4), (int64_t)cu_count() * gmul) : grid;
  if (sumsq)                                                     // one partial per workgroup of the grid actually launched
    if (int rc = scratch_doubles(s, (size_t)(grid_n > grid ? grid_n : grid), &part)) return rc;
#define GN(HB, SS, VC)                                                                                                            \
  do {                                                                                                                            \
    if (U >= 16) hipLaunchKernelGGL((k_gemv_n<HB, SS, VC, false, 16>), dim3(grid_n), dim3(NT), 0, s, YPtr{y}, V, ld, k, n, a, base, sc, out, part, (const float*)nullptr, stream_nontemporal(n)); \
    else if (U >= 8) hipLaunchKernelGGL((k_gemv_n<HB, SS, VC, false, 8>), dim3(grid_n), dim3(NT), 0, s, YPtr{y}, V, ld, k, n, a, base, sc, out, part, (const float*)nullptr, stream_nontemporal(n)); \
    else hipLaunchKernelGGL((k_gemv_n<HB, SS, VC, false, 4>), dim3(grid_n), dim3(NT), 0, s, YPtr{y}, V, ld, k, n, a, base, sc, out, part, (const float*)nullptr, stream_nontemporal(n)); \
  } while (0)
  if (base) {
    if (sumsq) { if (vec) GN(true, true, true); else GN(true, true, false); }
    else       { if (vec) GN(true, false, true); else GN(true, false, false); }
  } else {
    if (sumsq) { if (vec) GN(false, true, true); else GN(false, true, false); }
    else       { if (vec) GN(false, false, true); else GN(false, false, false); }
  }
#undef GN
  TRK_LAUNCH_CHECK();
  if (sumsq) *part_out = part;
  *nblk = grid_n;
  return TRK_OK;
}

// out = x / sqrt(S), S = the sum of nblk block partials added up by every workgroup as k_finalize would (finalize_block_256: the same
// bits as the finalize launch + trk_axpby(1 / sqrt(*S)) pair it replaces); workgroup 0 leaves S in *sum_out and carries the mailbox
// post, if any (PostReq: its scalars are final here — *sum_out is the last one).
// DOT: the pass also takes <out, dotv> (Hybrid-GMRES with the discrepancy principle wants V_{k+1}^T b, one new entry per step): block
// partials stored write-through, a ticket per workgroup, and the workgroup that draws the LAST one adds them up (k_finalize's order, loads
// past the caches), stores the sum in *dot_out and carries the post instead of workgroup 0 — the dot travels with it (PostReq::sum_host).
template <bool VEC, bool DOT>
__global__ __launch_bounds__(NT) void k_scale_fin(int64_t n, const double* __restrict__ part, int nblk, const float* x, float* out,
                                                  double* sum_out, const PostReq pq, const float* __restrict__ dotv, double* dot_part,
                                                  unsigned* cnt, double* dot_out) {
  __shared__ double lds[NT / 64];
  __shared__ double bc, bd;
  __shared__ unsigned ticket;
  const double S = finalize_block_256(part, nblk, 1, lds);
  if (threadIdx.x == 0) {
    bc = S;
    if (blockIdx.x == 0) *sum_out = S;
  }
  __syncthreads();
  double cv = 1.0;
  cv /= sqrt(bc);                                        // coef_eval(Coef{1.0, nullptr, S, TRK_SQRT_DEN})
  const float a = (float)cv;
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  int64_t tail0 = 0;
  double acc = 0.0;
  if (VEC) {
    const int64_t n4 = n >> 2;
    tail0 = n4 << 2;
    for (int64_t i = tid; i < n4; i += nth) {
      float4 v = ld4(x, i), o;
      o.x = a * v.x;
      o.y = a * v.y;
      o.z = a * v.z;
      o.w = a * v.w;
      st4(out, i, o);
      if (DOT) {
        const float4 bv = ld4(dotv, i);
        acc += (double)o.x * bv.x + (double)o.y * bv.y + (double)o.z * bv.z + (double)o.w * bv.w;
      }
    }
  }
  for (int64_t i = tail0 + tid; i < n; i += nth) {
    const float o = a * x[i];
    out[i] = o;
    if (DOT) acc += (double)o * dotv[i];
  }
  bool poster = blockIdx.x == 0;
  if (DOT) {
    acc = block_sum<NT>(acc, lds);
    if (threadIdx.x == 0) {
      asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(dot_part + blockIdx.x), "v"(acc) : "memory");
      ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    poster = ticket == gridDim.x - 1;
    if (!poster) return;
    if (threadIdx.x == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // for the next launch
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;        // finalize_block_256's association, the partials loaded past the caches
    auto ldp = [&](int bb) -> double {
      double t;
      asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(t) : "v"(dot_part + bb) : "memory");
      return t;
    };
    int bb = threadIdx.x;
    const int nb = (int)gridDim.x;
    for (; bb + 768 < nb; bb += 1024) {
      v0 += ldp(bb);
      v1 += ldp(bb + 256);
      v2 += ldp(bb + 512);
      v3 += ldp(bb + 768);
    }
    for (; bb < nb; bb += 256) v0 += ldp(bb);
    const double D = block_sum<NT>((v0 + v1) + (v2 + v3), lds);
    if (threadIdx.x == 0) {
      bd = D;
      *dot_out = D;
    }
    __syncthreads();
  }
  if (pq.on && poster && threadIdx.x < 64) {                    // one wave: its lanes move in step, the publication follows the copies
    for (int c = threadIdx.x; c < pq.count; c += 64) {
      const double* sp = pq.src + c;
      pq.dst[c] = (sp == sum_out) ? bc : *sp;
    }
    if (DOT && pq.sum_host && threadIdx.x == 0) *pq.sum_host = bd;       // the dot: its own place on the host (PostReq::sum_host)
    __threadfence_system();
    if (threadIdx.x == 0) __hip_atomic_store(pq.seq, pq.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int trk::scale_by_partials(int64_t n, const double* part, int nblk, const float* x, float* out, double* sum_out, const PostReq& post,
                           hipStream_t s, const float* dotv, double* dot_out) {
  TRK_REQUIRE(part && nblk >= 1 && x && out && sum_out && n >= 0 && (!dotv || dot_out), "scale_by_partials: bad argument");
  const int grid = stream_grid(n);
  const bool vec = aligned16(x) && aligned16(out) && (!dotv || aligned16(dotv));
  if (dotv) {
    unsigned* cnt = nullptr;
    if (int rc = stream_ticket(s, &cnt)) return rc;
    // the dot's partials behind the norm's in the stream's scratch (`part` is its start: gemv_n_partials left nblk doubles there)
    double* base = nullptr;
    if (int rc = scratch_doubles(s, (size_t)nblk + (size_t)grid, &base)) return rc;
    TRK_REQUIRE(base == part, "scale_by_partials: the norm's partials are not at the start of the stream's scratch");
    double* dpart = base + nblk;
    if (vec) hipLaunchKernelGGL((k_scale_fin<true, true>), dim3(grid), dim3(NT), 0, s, n, part, nblk, x, out, sum_out, post, dotv, dpart, cnt, dot_out);
    else hipLaunchKernelGGL((k_scale_fin<false, true>), dim3(grid), dim3(NT), 0, s, n, part, nblk, x, out, sum_out, post, dotv, dpart, cnt, dot_out);
  } else {
    const float* nof = nullptr;
    double* nod = nullptr;
    unsigned* noc = nullptr;
    if (vec) hipLaunchKernelGGL((k_scale_fin<true, false>), dim3(grid), dim3(NT), 0, s, n, part, nblk, x, out, sum_out, post, nof, nod, noc, nod);
    else hipLaunchKernelGGL((k_scale_fin<false, false>), dim3(grid), dim3(NT), 0, s, n, part, nblk, x, out, sum_out, post, nof, nod, noc, nod);
  }
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}
extern "C" {

int trk_gemv_n_err(const float* V, int64_t ld, int k, int64_t n, const double* y, float* out, const float* ref,
                   double* err_partials, int capacity_blocks, int* n_blocks, trk_stream st) {
  TRK_REQUIRE(V && y && out && ref && err_partials && n_blocks, "trk_gemv_n_err: NULL argument");
  TRK_REQUIRE(k >= 1 && k <= KMAX_LDS && n >= 0 && ld >= n, "trk_gemv_n_err: need 1 <= k <= %d, n >= 0, ld >= n", KMAX_LDS);
  // the launch shape of trk_gemv_n (8 workgroups per CU: tools/gemv_micro.py) when the caller's buffer has room for its partials
  static const int gmul = env_int("TRK_GEMVN_GRID", 8);
  int grid = stream_grid(n);
  if (gmul > 0) {
    const int g8 = (int)std::min<int64_t>((n + (int64_t)NT * 4 - 1) / ((int64_t)NT * 4), (int64_t)cu_count() * gmul);
    if (g8 >= 1 && g8 <= capacity_blocks) grid = g8;
  }
  TRK_REQUIRE(grid <= capacity_blocks, "trk_gemv_n_err: partial buffer too small (%d blocks needed)", grid);
  *n_blocks = grid;
  hipStream_t s = (hipStream_t)st;
  const float* nobase = nullptr;
  if (aligned16(V) && aligned16(out) && aligned16(ref) && (ld % 4 == 0))
    hipLaunchKernelGGL((k_gemv_n<false, true, true, true>), dim3(grid), dim3(NT), 0, s, YPtr{y}, V, ld, k, n, 1.0, nobase, 1.0, out, err_partials, ref, stream_nontemporal(n));
  else
    hipLaunchKernelGGL((k_gemv_n<false, true, false, true>), dim3(grid), dim3(NT), 0, s, YPtr{y}, V, ld, k, n, 1.0, nobase, 1.0, out, err_partials, ref, stream_nontemporal(n));
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_gemv_orth_iterate(const float* V, int64_t ld, int k, int64_t n, const float* w, const double* c, const double* rho2,
                          const double* y_next, float* vn, float* x_next, const float* ref, double* err_partials, int capacity_blocks,
                          int* n_blocks, double* chk_sumsq, trk_stream st) {
  TRK_REQUIRE(V && w && c && rho2 && vn, "trk_gemv_orth_iterate: NULL argument");
  TRK_REQUIRE((y_next != nullptr) == (x_next != nullptr), "trk_gemv_orth_iterate: y_next and x_next come together");
  TRK_REQUIRE(!ref || (x_next && err_partials && n_blocks), "trk_gemv_orth_iterate: ref needs x_next and room for the partials");
  TRK_REQUIRE(k >= 1 && k < KMAX_LDS && n >= 0 && ld >= n, "trk_gemv_orth_iterate: need 1 <= k < %d, n >= 0, ld >= n", KMAX_LDS);
  TRK_REQUIRE(vn != w && x_next != w && x_next != vn, "trk_gemv_orth_iterate: the outputs must not alias w or each other");
  static const int gmul = env_int("TRK_GEMVN_GRID", 8);        // trk_gemv_n_err's launch shape
  int grid = stream_grid(n);
  if (gmul > 0) {
    const int g8 = (int)std::min<int64_t>((n + (int64_t)NT * 4 - 1) / ((int64_t)NT * 4), (int64_t)cu_count() * gmul);
    if (g8 >= 1 && (!ref || g8 <= capacity_blocks)) grid = g8;
  }
  if (ref) {
    TRK_REQUIRE(grid <= capacity_blocks, "trk_gemv_orth_iterate: partial buffer too small (%d blocks needed)", grid);
    *n_blocks = grid;
  }
  hipStream_t s = (hipStream_t)st;
  double* chk = nullptr;
  if (chk_sumsq)
    if (int rc = scratch_doubles(s, (size_t)grid, &chk)) return rc;
  const bool vec = aligned16(V) && aligned16(w) && aligned16(vn) && (!x_next || aligned16(x_next)) && (!ref || aligned16(ref)) && (ld % 4 == 0);
  if (!x_next && !chk_sumsq && gemv_n_split_serves(n, k, vec)) {  // the image of the new vector, A v_k = (A r - AV c) / rho: short rows
    const int64_t n4 = n >> 2;
    hipLaunchKernelGGL((k_gemv_n_split<true>), dim3((unsigned)((n4 + 63) / 64)), dim3(NT), 0, s, c, V, ld, k, n4, 1.0, w, -1.0, vn, rho2);
    TRK_LAUNCH_CHECK();
    return TRK_OK;
  }
  const int nt = stream_nontemporal(n);
#define GO(VC, HX, HR) hipLaunchKernelGGL((k_gemv_orth_iter<VC, HX, HR, 8>), dim3(grid), dim3(NT), 0, s, V, ld, k, n, w, c, rho2, y_next, vn, x_next, ref, err_partials, chk, nt)
  if (!x_next)  { if (vec) GO(true, false, false); else GO(false, false, false); }
  else if (ref) { if (vec) GO(true, true, true); else GO(false, true, true); }
  else          { if (vec) GO(true, true, false); else GO(false, true, false); }
#undef GO
  TRK_LAUNCH_CHECK();
  if (chk_sumsq) return finalize_sums(chk, grid, 1, 1, chk_sumsq, s);
  return TRK_OK;
}

int trk_gemv_n_hosty(const float* V, int64_t ld, int k, int64_t n, const double* y_host, float* out, const float* ref,
                     double* err_partials, int capacity_blocks, int* n_blocks, trk_stream st) {
  TRK_REQUIRE(V && y_host && out, "trk_gemv_n_hosty: NULL argument");
  TRK_REQUIRE(!ref || (err_partials && n_blocks), "trk_gemv_n_hosty: ref given but no room for the partials");
  TRK_REQUIRE(k >= 1 && n >= 0 && ld >= n, "trk_gemv_n_hosty: need k >= 1, n >= 0, ld >= n");
  const int grid = stream_grid(n);
  if (ref) {
    TRK_REQUIRE(grid <= capacity_blocks, "trk_gemv_n_hosty: partial buffer too small (%d blocks needed)", grid);
    *n_blocks = grid;
  }
  hipStream_t s = (hipStream_t)st;
  const bool vec = aligned16(V) && aligned16(out) && (!ref || aligned16(ref)) && (ld % 4 == 0);
  // YARG_MAX coefficients per launch; further groups of rows add to what the launches before them left in `out` (rounded to
  // fp32 in between: one rounding more per 128 terms), the last one carries the error norm
  for (int j0 = 0; j0 < k; j0 += YARG_MAX) {
    const int kk = std::min(YARG_MAX, k - j0);
    const bool last = j0 + kk == k, first = j0 == 0;
    YArg ya;
    for (int j = 0; j < kk; ++j) ya.v[j] = y_host[j0 + j];
    for (int j = kk; j < YARG_MAX; ++j) ya.v[j] = 0.0;
    const float* Vj = V + (int64_t)j0 * ld;
    const float* base = first ? nullptr : out;
    double* part = (last && ref) ? err_partials : nullptr;
#define GH(HB, SS, VC, HR) hipLaunchKernelGGL((k_gemv_n<HB, SS, VC, HR, 8, YArg>), dim3(grid), dim3(NT), 0, s, ya, Vj, ld, kk, n, 1.0, base, 1.0, out, part, ref, 0)
    if (first) {
      if (part) { if (vec) GH(false, true, true, true); else GH(false, true, false, true); }
      else      { if (vec) GH(false, false, true, false); else GH(false, false, false, false); }
    } else {
      if (part) { if (vec) GH(true, true, true, true); else GH(true, true, false, true); }
      else      { if (vec) GH(true, false, true, false); else GH(true, false, false, false); }
    }
#undef GH
  }
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_lsqr_damped_update(const float* vk, float* w, const float* x_in, float* x_out, int64_t n, const float* ref,
                           double* err_partials, int capacity_blocks, int* n_blocks, const double* alpha_sq,
                           const double* beta_next_sq, const double* beta0_sq, double damp, const double* state_in,
                           double* state_out, int first, trk_stream st) {
  TRK_REQUIRE(vk && w && x_out && alpha_sq && beta_next_sq && state_out, "trk_lsqr_damped_update: NULL argument");
  TRK_REQUIRE(first ? beta0_sq != nullptr : (state_in != nullptr && x_in != nullptr),
              "trk_lsqr_damped_update: the first step needs beta0_sq, later ones state_in and x_in");
  TRK_REQUIRE(!ref || (err_partials && n_blocks), "trk_lsqr_damped_update: ref given but no room for the partials");
  TRK_REQUIRE(n >= 0 && damp >= 0.0, "trk_lsqr_damped_update: need n >= 0, damp >= 0");
  const int grid = stream_grid(n);
  if (ref) {
    TRK_REQUIRE(grid <= capacity_blocks, "trk_lsqr_damped_update: partial buffer too small (%d blocks needed)", grid);
    *n_blocks = grid;
  }
  hipStream_t s = (hipStream_t)st;
  const bool vec = aligned16(vk) && aligned16(w) && aligned16(x_out) && (!x_in || aligned16(x_in)) && (!ref || aligned16(ref));
  if (vec)
    hipLaunchKernelGGL((k_lsqr_damped_update<float, true>), dim3(grid), dim3(NT), 0, s, vk, w, x_in, x_out, ref, err_partials, n, alpha_sq,
                       beta_next_sq, beta0_sq, damp, state_in, state_out, first);
  else
    hipLaunchKernelGGL((k_lsqr_damped_update<float, false>), dim3(grid), dim3(NT), 0, s, vk, w, x_in, x_out, ref, err_partials, n, alpha_sq,
                       beta_next_sq, beta0_sq, damp, state_in, state_out, first);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_gemv_nt(const float* V, int64_t ld, int k, int64_t n, const double* h, const float* w_in, float* w_out,
                double* g, trk_stream st) {
  TRK_REQUIRE(V && h && w_in && w_out && g, "trk_gemv_nt: NULL argument");
  TRK_REQUIRE(k >= 1 && k <= 16 && n >= 0 && ld >= n, "trk_gemv_nt: need 1 <= k <= 16, n >= 0, ld >= n");
  hipStream_t s = (hipStream_t)st;
  int bx = stream_grid(n);
  double* part = nullptr;
  if (int rc = scratch_doubles(s, (size_t)bx * k, &part)) return rc;
  const bool vec = aligned16(V) && aligned16(w_in) && aligned16(w_out) && (ld % 4 == 0);
#define NTK(KBB, VC) hipLaunchKernelGGL((k_gemv_nt<KBB, VC>), dim3(bx), dim3(NT), 0, s, V, ld, k, n, h, w_in, w_out, part)
  if (k <= 8) { if (vec) NTK(8, true); else NTK(8, false); }
  else        { if (vec) NTK(16, true); else NTK(16, false); }
#undef NTK
  TRK_LAUNCH_CHECK();
  return finalize_sums(part, bx, k, k, g, s);
}

static int g_wgram_tv_mode = -1;       // -1: not chosen yet (environment, else 1 = auto)
static double* g_wgram_gate = nullptr; // {verdict, worst sampled deviation} of the last 'auto' call (device; process-wide like the mode)
static int wgram_tv_mode() {
  if (g_wgram_tv_mode < 0) {
    const int pcs = env_int("TRK_WGRAM_TV_PIECES", 0);
    g_wgram_tv_mode = env_int("TRK_WGRAM_TV_F32", 0) ? 0 : (pcs == 3 ? 3 : pcs == 2 ? 2 : 1);
  }
  return g_wgram_tv_mode;
}
int trk_wgram_tv_precision(int mode) {
  TRK_REQUIRE(mode >= -1 && mode <= 3, "trk_wgram_tv_precision: mode 1 (auto), 0 (fp32 pipe), 2 or 3 (bf16 pieces), -1 (query)");
  const int was = wgram_tv_mode();
  if (mode >= 0) g_wgram_tv_mode = mode;
  return was;
}
int trk_wgram_tv_last_probe(double* verdict_and_deviation_host) {
  TRK_REQUIRE(verdict_and_deviation_host, "trk_wgram_tv_last_probe: NULL argument");
  verdict_and_deviation_host[0] = verdict_and_deviation_host[1] = -1.0;
  if (!g_wgram_gate) return TRK_OK;                              // no 'auto' call yet
  TRK_HIP(hipDeviceSynchronize());
  TRK_HIP(hipMemcpy(verdict_and_deviation_host, g_wgram_gate, 2 * sizeof(double), hipMemcpyDeviceToHost));
  return TRK_OK;
}
static int wgram_tv_run(const float* V, int64_t ld, int k, int N, const float* w, double* G, const float* z, double* h, trk_stream st);

int trk_wgram_tv(const float* V, int64_t ld, int k, int N, const float* w, double* G, trk_stream st) {
  return wgram_tv_run(V, ld, k, N, w, G, nullptr, nullptr, st);
}

int trk_wgram_tv_z(const float* V, int64_t ld, int k, int N, const float* w, double* G, const float* z, double* h, trk_stream st) {
  TRK_REQUIRE(z && h && aligned16(z), "trk_wgram_tv_z: z / h NULL or z not 16-byte aligned");
  return wgram_tv_run(V, ld, k, N, w, G, z, h, st);
}

static int wgram_tv_run(const float* V, int64_t ld, int k, int N, const float* w, double* G, const float* z, double* h, trk_stream st) {
  TRK_REQUIRE(V && w && G, "trk_wgram_tv: NULL argument");
  TRK_REQUIRE(k >= 1 && k <= 48, "trk_wgram_tv: need 1 <= k <= 48 (trk_wgram over the stored images beyond)");
  TRK_REQUIRE(N >= 32 && N % 32 == 0 && ld >= (int64_t)N * N && ld % 4 == 0 && aligned16(V) && aligned16(w),
              "trk_wgram_tv: need N a multiple of 32, 16-byte aligned rows and weights");
  hipStream_t s = (hipStream_t)st;
  const int T16 = (k + 15) / 16;
  static const int pc_env = env_int("TRK_WGRAM_TV_PER_CU", 0);
  const int strips = N / 32;
  // The four waves of a workgroup in step (see the kernel: the right-neighbour column handed over through LDS instead of fetched
  // again — 25-40 % less HBM traffic, one barrier per image row): pays where the rows' traffic is the bound.  Measured at 4096^2
  // (tools/wgram_tv_micro.py; us, lockstep vs not): plain, two tiles k = 20 / 32: 507 / 620 vs 592 / 831; one tile k = 16: 308 vs
  // 366, k <= 10 equal; with the dots of _z (218 registers, two workgroups per CU at two tiles): k = 18 / 26 / 32: 640 / 649 / 676 vs
  // 598 / 666 / 791 — from k = 25 on.
  static const int ls_env = env_int("TRK_WGRAM_TV_LOCKSTEP", -1);
  const bool ls_pays = T16 == 1 ? k >= 12 : T16 == 2 ? (z ? k >= 22 : true) : false;      // (round 5, two bf16 pieces, _z: k = 20 455 vs 464 us, k = 24 540 vs 478)
  const int lock = (strips % (NT / 64) == 0 && (ls_env < 0 ? ls_pays : ls_env != 0)) ? 1 : 0;
  // workgroups per CU: what the registers let be resident while the matrix pipe is the bound; without the lockstep exchange fewer
  // once the rows' traffic is (k = 32 plain: 762 / 744 / 683 us with 4 / 2 / 1 — more waves, more row streams open at once)
  const int per_cu = pc_env > 0 ? pc_env
                   : lock ? (z ? (T16 == 1 ? 3 : 2) : 3)
                   : z ? (T16 == 1 ? 3 : T16 == 2 ? 2 : 2)
                       : (T16 == 1 ? (k <= 11 ? 4 : k <= 14 ? 2 : 1) : T16 == 2 ? (k <= 25 ? 3 : 1) : 2);
  int bx = cu_count() * per_cu;
  // (strip, band) units, band-major: the waves in flight together then work on a few neighbouring image rows of every basis vector
  static const int band_env = env_int("TRK_WGRAM_TV_BAND", 64);
  int band_rows = band_env < N ? (band_env > 0 ? band_env : 16) : N;
  int nbands = (N + band_rows - 1) / band_rows;
  const int64_t units = (int64_t)strips * nbands;
  if ((int64_t)bx * (NT / 64) > units) bx = (int)((units + NT / 64 - 1) / (NT / 64));
  // Large images, at most two tiles of vectors: the rows of V staged through LDS in full lines (k_wgram_tv_lds): 256-column tiles,
  // one workgroup of 8 waves per CU (two at one tile of vectors).  TRK_WGRAM_TV_LDS=0: the register-fed kernel everywhere (A/B).
  static const int lds_env = env_int("TRK_WGRAM_TV_LDS", 1);
  static const int lds_band_env = env_int("TRK_WGRAM_TV_LDS_BAND", 64);
  static const int lds_pc_env = env_int("TRK_WGRAM_TV_LDS_PER_CU", 0);
  const bool use_lds = lds_env != 0 && T16 <= 2 && N % LW_COLS == 0 && N >= 2048;
  if (use_lds) {
    band_rows = lds_band_env >= 4 && lds_band_env < N ? lds_band_env : 64;
    while (N % band_rows) band_rows >>= 1;                      // (N is a multiple of 256: a power of two <= 256 divides it)
    nbands = N / band_rows;
    const int64_t lunits = (int64_t)(N / LW_COLS) * nbands;
    bx = cu_count() * (lds_pc_env > 0 ? (lds_pc_env > 2 ? 2 : lds_pc_env) : (T16 == 1 ? 2 : 1));
    if (T16 == 2 && bx > cu_count()) bx = cu_count();
    if (bx > lunits) bx = (int)lunits;
  }
  double* part = nullptr;
  const int nv = k * k + (z ? k : 0);
  if (int rc = scratch_doubles(s, (size_t)bx * nv + ((size_t)PROBE_ROWS + 1) * 2 * 2 * PROBE_P, &part)) return rc;
  double* probe_part = part + (size_t)bx * nv;
  double* probe_sums = probe_part + (size_t)PROBE_ROWS * 2 * 2 * PROBE_P;
#ifdef TRK_WGRAM_TV_EXPERIMENT
  static const int no_xcd = (env_int("TRK_WGRAM_TV_NO_XCD", 0) ? 2 : 0) | (env_int("TRK_WGRAM_TV_X", 0) & 12);
#else
  static const int no_xcd = env_int("TRK_WGRAM_TV_NO_XCD", 0) ? 2 : 0;
#endif
  // Which arithmetic forms the tile products (trk_wgram_tv_precision; environment TRK_WGRAM_TV_F32=1 / TRK_WGRAM_TV_PIECES=2|3 set the
  // process default): 1 auto (default: two bf16 pieces unless the probe finds the data's roundings correlated, then the fp32 pipe),
  // 0 fp32 matrix pipe, 2 two bf16 pieces, 3 three bf16 pieces
  const int mode = wgram_tv_mode();
  // two tiles, three pieces: 256 registers (no spills) at 2 workgroups per CU is the faster form (532 / 605 us at k = 17 / 32 against
  // 648 / 781 with 32 spilled registers at 3: profiles/r05/wgram_tv_pieces.txt)
  static const int occ2 = env_int("TRK_WGRAM_TV_OCC2", 1);
  ProbeGate pg{nullptr, 0.0, 0, nullptr, 1};
  if (mode == 1) {
    if (!g_wgram_gate) TRK_HIP(hipMalloc((void**)&g_wgram_gate, 2 * sizeof(double)));
    // verdict threshold on the SAMPLED deviation: a third of the 1e-6 the contract promises (the sample is an estimate)
    static const double thr = getenv("TRK_WGRAM_TV_PROBE_THRESHOLD") ? atof(getenv("TRK_WGRAM_TV_PROBE_THRESHOLD")) : 3e-7;
    const int row_step = N / PROBE_ROWS > 0 ? N / PROBE_ROWS : 1;
    const int prows = (N + row_step - 1) / row_step < PROBE_ROWS ? (N + row_step - 1) / row_step : PROBE_ROWS;
    const int pgroups = k >= 2 * PROBE_V ? 2 : 1;
    hipLaunchKernelGGL(k_wgram_tv_probe, dim3(prows, pgroups), dim3(NT), 0, s, V, ld, k, N, w, row_step, probe_part);
    TRK_LAUNCH_CHECK();
    if (int rc = finalize_sums(probe_part, prows, pgroups * 2 * PROBE_P, pgroups * 2 * PROBE_P, probe_sums, s)) return rc;
    pg = ProbeGate{probe_sums, thr, 0, g_wgram_gate, pgroups};
  }
  // arith: 0 fp32 pipe, 2 / 3 bf16 pieces; want: with a probe, the launch runs iff the probe's verdict equals it
#define WTV(TT, ZZ, ARITH, WANT)                                                                                                                      \
  do {                                                                                                                                                \
    if ((ARITH) == 4) hipLaunchKernelGGL((k_wgram_tv<TT, ZZ, 3, 4>), dim3(bx), dim3(NT), 0, s, V, ld, k, N, w, nbands, band_rows, part, z, lock | no_xcd, ProbeGate{pg.sums, pg.threshold, 0, pg.record, pg.groups}); \
    else if ((ARITH) == 0) hipLaunchKernelGGL((k_wgram_tv<TT, ZZ, 3, 0>), dim3(bx), dim3(NT), 0, s, V, ld, k, N, w, nbands, band_rows, part, z, lock | no_xcd, ProbeGate{pg.sums, pg.threshold, WANT, pg.record, pg.groups}); \
    else if ((ARITH) == 2) hipLaunchKernelGGL((k_wgram_tv<TT, ZZ, 3, 2>), dim3(bx), dim3(NT), 0, s, V, ld, k, N, w, nbands, band_rows, part, z, lock | no_xcd, ProbeGate{pg.sums, pg.threshold, WANT, pg.record, pg.groups}); \
    else if (TT == 2 && !ZZ && occ2) hipLaunchKernelGGL((k_wgram_tv<TT, ZZ, 3, 3, 2>), dim3(bx), dim3(NT), 0, s, V, ld, k, N, w, nbands, band_rows, part, z, lock | no_xcd, ProbeGate{pg.sums, pg.threshold, WANT, pg.record, pg.groups}); \
    else hipLaunchKernelGGL((k_wgram_tv<TT, ZZ, 3, 3>), dim3(bx), dim3(NT), 0, s, V, ld, k, N, w, nbands, band_rows, part, z, lock | no_xcd, ProbeGate{pg.sums, pg.threshold, WANT, pg.record, pg.groups});           \
  } while (0)
  // one pass of the chosen arithmetic: the kernel and the sum of its block partials
#define WTVL1(TT, ZZ, BFV, WANT) hipLaunchKernelGGL((k_wgram_tv_lds<TT, ZZ, BFV>), dim3(bx), dim3(LW_NT), 0, s, V, ld, k, N, w, nbands, band_rows, part, z, (int)no_xcd, ProbeGate{pg.sums, pg.threshold, WANT, pg.record, pg.groups})
#define WTVL(TT, ZZ, ARITH, WANT)                                \
  do {                                                           \
    if ((ARITH) == 4) WTVL1(TT, ZZ, 4, 0);                       \
    else if ((ARITH) == 0) WTVL1(TT, ZZ, 0, WANT);               \
    else if ((ARITH) == 2) WTVL1(TT, ZZ, 2, WANT);               \
    else WTVL1(TT, ZZ, 3, WANT);                                 \
  } while (0)
  auto pass = [&](int arith, int want) -> int {
    const bool two_pass = z && T16 == 3;     // three tiles AND the dots do not fit the register file (108 spilled registers): the dots
    if (use_lds) {
      if (z) { if (T16 == 1) WTVL(1, true, arith, want); else WTVL(2, true, arith, want); }
      else { if (T16 == 1) WTVL(1, false, arith, want); else WTVL(2, false, arith, want); }
    } else
    if (two_pass) WTV(3, false, arith, want);   // of 33 <= k <= 48 in a pass of their own, below
    else if (z) { if (T16 == 1) WTV(1, true, arith, want); else WTV(2, true, arith, want); }
    else { if (T16 == 1) WTV(1, false, arith, want); else if (T16 == 2) WTV(2, false, arith, want); else WTV(3, false, arith, want); }
    TRK_LAUNCH_CHECK();
    const int nout = two_pass ? k * k : nv;
    if (pg.sums && arith != 4) {
      hipLaunchKernelGGL(k_finalize_gated, dim3(nout), dim3(256), 0, s, part, bx, nout, G, k * k, h, ProbeGate{pg.sums, pg.threshold, want, nullptr, pg.groups});
      TRK_LAUNCH_CHECK();
      return TRK_OK;
    }
    if (z && !two_pass) return finalize_sums_split(part, bx, nv, nv, G, k * k, h, s);
    return finalize_sums(part, bx, k * k, k * k, G, s);
  };
  static const int auto_pair = env_int("TRK_WGRAM_TV_AUTO_PAIR", 0);     // 1: the gated pair of launches instead of one launch with both forms (A/B)
  if (mode == 1 && auto_pair) {
    if (int rc = pass(2, 0)) return rc;
    if (int rc = pass(0, 1)) return rc;
  } else if (mode == 1) {
    if (int rc = pass(4, 0)) return rc;
  } else {
    if (int rc = pass(mode, 0)) return rc;
  }
  if (z && T16 == 3) return launch_gemv_t(V, ld, k, (int64_t)N * N, z, nullptr, 0, h, s);
  return TRK_OK;
}
#undef WTV
#undef WTVL
#undef WTVL1

int trk_wgram(const float* W, int64_t ld, int k, int64_t m, const float* w, const float* b1, double* G, double* c1,
              double* c2, trk_stream st) {
  TRK_REQUIRE(W && G, "trk_wgram: NULL argument");
  TRK_REQUIRE(k >= 1 && k <= 512 && m >= 0 && ld >= m, "trk_wgram: need 1 <= k <= 512, m >= 0, ld >= m");
  TRK_REQUIRE(!b1 || (c1 && c2), "trk_wgram: b1 given but c1/c2 NULL");
  hipStream_t s = (hipStream_t)st;
  const int KA = k + (b1 ? 2 : 0);
  const int nt2 = ceil_div(KA, WG_TILE), ntu = nt2 * (nt2 + 1) / 2;
  static const bool no_mfma = getenv("TRK_WGRAM_NO_MFMA") != nullptr;
  if (KA <= 64 && !no_mfma) {
    // matrix-core single pass (every row read once)
    static const bool no_direct = getenv("TRK_WGRAM_NO_DIRECT") != nullptr;
    const bool al16 = (ld % 4 == 0) && aligned16(W) && (!w || aligned16(w)) && (!b1 || aligned16(b1));
    const bool direct = al16 && !no_direct && KA <= 48;   // T = 4 does not fit the register file: 49..64 rows stay on the 32x32 kernel
    const int T16 = (KA + 15) / 16;
    // blocks per CU = what the register budget of the variant lets be resident (8 / 4 / 2 waves per SIMD for T = 1 / 2 / 3)
    static const int per_cu_env = getenv("TRK_WGRAM_PER_CU") ? atoi(getenv("TRK_WGRAM_PER_CU")) : 0;
    const int per_cu = per_cu_env ? per_cu_env : direct ? (T16 == 1 ? 8 : T16 == 2 ? 4 : 2) : 3;
    int64_t nchunk = (m + 127) / 128;
    int bx = (int)(nchunk < (int64_t)cu_count() * per_cu ? (nchunk > 0 ? nchunk : 1) : (int64_t)cu_count() * per_cu);
    if (bx > 2 * kMaxPartialBlocks) bx = 2 * kMaxPartialBlocks;
    double* part = nullptr;
    const size_t npart = (size_t)bx * KA * KA;
    if (int rc = scratch_doubles(s, npart + (size_t)KA * KA, &part)) return rc;
    double* Ga = part + npart;
#define WM(NTI, HW, HB) hipLaunchKernelGGL((k_wgram_mfma<NTI, HW, HB>), dim3(bx), dim3(NT), 0, s, W, ld, k, m, w, b1, part)
    if (direct) {
#define WD(TT, HW, HB) hipLaunchKernelGGL((k_wgram_t16<TT, HW, HB>), dim3(bx), dim3(NT), 0, s, W, ld, k, m, w, b1, part)
#define WDT(TT) do { if (w) { if (b1) WD(TT, true, true); else WD(TT, true, false); } \
                     else   { if (b1) WD(TT, false, true); else WD(TT, false, false); } } while (0)
      if (T16 == 1) WDT(1); else if (T16 == 2) WDT(2); else WDT(3);
#undef WDT
#undef WD
    } else if (KA <= 32) {
      if (w) { if (b1) WM(1, true, true); else WM(1, true, false); }
      else   { if (b1) WM(1, false, true); else WM(1, false, false); }
    } else {
      if (w) { if (b1) WM(2, true, true); else WM(2, true, false); }
      else   { if (b1) WM(2, false, true); else WM(2, false, false); }
    }
#undef WM
    TRK_LAUNCH_CHECK();
    if (int rc = finalize_sums(part, bx, KA * KA, KA * KA, Ga, s)) return rc;
    hipLaunchKernelGGL(k_wgram_unpack, dim3(1), dim3(256), 0, s, Ga, k, KA, G, c1, c2);
    TRK_LAUNCH_CHECK();
    return TRK_OK;
  }
  // The LDS-staged single-pass kernel reads every row once but is instruction-bound (8 ds_read + 8 cvt per 16 fp64 FMAs):
  // measured slower than the tile-pair kernel at 4096^2 (k = 3..33: 3.2 ms vs 1.7 ms average), so it is opt-in.
  static const bool use_lds = getenv("TRK_WGRAM_LDS") != nullptr;
  if (use_lds && ntu <= NT) {
    // LDS-staged single pass
    const int KP = nt2 * WG_TILE;
    int CH = (KP <= 40) ? 256 : (KP <= 80) ? 128 : 64;
    size_t lds = (size_t)KP * (CH + 1) * sizeof(float);
    const size_t red = (size_t)NT * 16 * sizeof(double);
    if (lds < red) lds = red;
    int64_t nchunk = (m + CH - 1) / CH;
    int bx = (int)(nchunk < (int64_t)cu_count() * 2 ? (nchunk > 0 ? nchunk : 1) : (int64_t)cu_count() * 2);
    if (bx > kMaxPartialBlocks) bx = kMaxPartialBlocks;
    double* part = nullptr;  // [bx][KA*KA] partials, then the finished augmented Gram
    const size_t npart = (size_t)bx * KA * KA;
    if (int rc = scratch_doubles(s, npart + (size_t)KA * KA, &part)) return rc;
    double* Ga = part + npart;
#define WG2(HW, HB) hipLaunchKernelGGL((k_wgram2<HW, HB>), dim3(bx), dim3(NT), lds, s, W, ld, k, m, w, b1, CH, part)
    if (w) { if (b1) WG2(true, true); else WG2(true, false); }
    else   { if (b1) WG2(false, true); else WG2(false, false); }
#undef WG2
    TRK_LAUNCH_CHECK();
    if (int rc = finalize_sums(part, bx, KA * KA, KA * KA, Ga, s)) return rc;
    hipLaunchKernelGGL(k_wgram_unpack, dim3(1), dim3(256), 0, s, Ga, k, KA, G, c1, c2);
    TRK_LAUNCH_CHECK();
    return TRK_OK;
  }
  // many rows: tile-pair form (re-reads W once per tile pair)
  const int nt = ceil_div(k, TG);
  const int npairs = nt * (nt + 1) / 2;
  int bx = stream_grid(m);
  const int cap = (cu_count() * 8 + npairs - 1) / npairs;
  if (bx > cap) bx = cap < 1 ? 1 : cap;
  double* part = nullptr;  // [bx][k*k] partials
  if (int rc = scratch_doubles(s, (size_t)bx * k * k, &part)) return rc;
  const bool vec = aligned16(W) && (ld % 4 == 0) && (!w || aligned16(w));
  dim3 grid(bx, npairs);
#define WG(HW, VC) hipLaunchKernelGGL((k_wgram<HW, VC>), grid, dim3(NT), 0, s, W, ld, k, m, w, nt, part)
  if (w) { if (vec) WG(true, true); else WG(true, false); }
  else   { if (vec) WG(false, true); else WG(false, false); }
#undef WG
  TRK_LAUNCH_CHECK();
  if (int rc = finalize_sums(part, bx, k * k, k * k, G, s)) return rc;
  if (b1) {
    if (w) {
      if (int rc = launch_gemv_t(W, ld, k, m, b1, w, 1, c1, s)) return rc;
      if (int rc = launch_gemv_t(W, ld, k, m, b1, w, 2, c2, s)) return rc;
    } else {
      if (int rc = launch_gemv_t(W, ld, k, m, b1, nullptr, 0, c1, s)) return rc;
      TRK_HIP(hipMemcpyAsync(c2, c1, sizeof(double) * k, hipMemcpyDeviceToDevice, s));
    }
  }
  return TRK_OK;
}

}  // extern "C"

// the damped-LSQR update on float or double vectors (ref64.hip's chain): the scalar-access instantiation of the production template
namespace trk {
int lsqr_damped_update_any(size_t elem_bytes, const void* vk, void* w, const void* x_in, void* x_out, int64_t n, const double* alpha_sq,
                           const double* beta_next_sq, const double* beta0_sq, double damp, const double* state_in, double* state_out,
                           int first, hipStream_t s) {
  const int grid = stream_grid(n);
  if (elem_bytes == 8)
    hipLaunchKernelGGL((k_lsqr_damped_update<double, false>), dim3(grid), dim3(NT), 0, s, (const double*)vk, (double*)w, (const double*)x_in,
                       (double*)x_out, (const double*)nullptr, (double*)nullptr, n, alpha_sq, beta_next_sq, beta0_sq, damp, state_in,
                       state_out, first);
  else
    hipLaunchKernelGGL((k_lsqr_damped_update<float, false>), dim3(grid), dim3(NT), 0, s, (const float*)vk, (float*)w, (const float*)x_in,
                       (float*)x_out, (const float*)nullptr, (double*)nullptr, n, alpha_sq, beta_next_sq, beta0_sq, damp, state_in,
                       state_out, first);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}
}  // namespace trk
