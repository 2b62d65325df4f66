// ref64.hip — the float64 instantiation of the Golub-Kahan / damped-LSQR chain (SURVEY §7 hard part 2: "keep an fp64
// instantiation of every kernel for debugging").  An INSTRUMENT, not a fast path:
//
//   * a parallel-beam projector pair with float64 arithmetic, templated on the STORAGE type T of its vectors (float / double)
//     and on where the interpolation weights come from:
//         weights 0: the geometry evaluated in float64 (q = (d - sdh) inv + k0 + tt dq), the oracle's own numbers to 1e-13;
//         weights 1: the production kernels' fixed-point tables (A32 + B32, 24 fractional bits, radon2d.hip) — the production
//                    operator's weights bit for bit, summed in float64 instead of fp32.
//     One thread per ray (forward) / per pixel (adjoint, gather over the rays that touch it: matched to the forward tap by tap).
//   * the vector kernels of a Golub-Kahan half step on the element type: out = a Op(x) + b z (float64 coefficients and
//     products, ONE rounding to T — the arithmetic of the production projector's epilogue, radon2d.hip epi_combine) with the fused
//     sum of squares; the damped-LSQR update is the production template itself (vecops.hip, k_lsqr_damped_update<T, ..>).
//   * trk_gk_lsqr_chain: the engine's arrangement of Hybrid-LSQR at a fixed lambda (krylov.GKState(normalized=False): U[j] =
//     beta_j u_j, V[j] = alpha_j v_j, squared norms as device doubles, the divisions folded into the next step's coefficients;
//     x_k by Paige & Saunders' short recurrence) on vectors of type T, enqueued by one call.
//
// What it answers (tests/test_gpu_c3_float64.py, tools/r05_c3_instrument.py): with T = double and weights 0 every iterate of C3 is
// within 1e-9 of the float64 oracle — arrangement and kernels exact on the hardware; T = float tells what fp32 STORAGE alone costs
// at the semi-convergence transient; weights 1 tells what the 2^-24 weight grid adds; the production kernels what their fp32
// partial sums add on top.  Reference lines: trips/solvers/Hybrid_LSQR.py:73-110, trips/utilities/decompositions.py:230-255,
// trips/utilities/io.py:392-399.
#include "trk_internal.h"

#include <cmath>
#include <cstdint>
#include <type_traits>

using namespace trk;

namespace {

constexpr int NT = 256;
constexpr int QF = 24;
constexpr int A32_PAD = 2;

// ---------------------------------------------------------------------------------------------- one tap pair of one ray
// column (row) index c of the left tap and the fraction f of the right one, for ray d of angle row `a` at marching index tt
template <int WEIGHTS>
__device__ __forceinline__ void ref_tap(const RadonRefAngle& p, const RadonRefGeom& g, int a, int d, int tt, int& c, double& f) {
  const double q = ((double)d - 0.5 * (double)(g.nd - 1)) * p.inv + p.k0 + (double)tt * p.dq;
  const double fl = floor(q);
  if (WEIGHTS == 0) {
    c = (int)fl;
    f = q - fl;
  } else {
    const unsigned Q = g.A32[(int64_t)a * (g.nd + 2 * A32_PAD) + d + A32_PAD] + g.B32[(int64_t)a * g.npad + tt];
    const int ce = (int)fl, cm = (int)(Q >> QF);
    c = ce + (((cm - ce + 128) & 255) - 128);                       // the table knows the column mod 256 (radon_abs_col)
    f = (double)(Q & 0xFFFFFFu) * (1.0 / 16777216.0);
  }
}

template <class T, int WEIGHTS>
__global__ __launch_bounds__(NT) void k_ref_radon_fwd(const T* __restrict__ img, T* __restrict__ sino, RadonRefGeom g) {
  const int64_t ray = (int64_t)blockIdx.x * NT + threadIdx.x;
  const int64_t nrays = (int64_t)g.nt * g.na * g.nd;
  if (ray >= nrays) return;
  const int a = (int)(ray / g.nd), d = (int)(ray - (int64_t)a * g.nd);
  const RadonRefAngle p = g.ang[a];
  const T* __restrict__ I = img + (int64_t)(a / g.na) * g.N * g.N;
  const int N = g.N;
  double acc = 0.0;
  if (g.chunk_fwd > 0) {
    // the product kernels' arithmetic, emulated: fp32 FMAs into one accumulator per tap side, moved to the float64 total every
    // chunk_fwd steps; the angle's weight applied to the fp32-rounded total in fp32
    float a0 = 0.f, a1 = 0.f;
    for (int tt = 0; tt < N; ++tt) {
      int c;
      double f;
      ref_tap<WEIGHTS>(p, g, a, d, tt, c, f);
      float v0 = 0.f, v1 = 0.f;
      if ((unsigned)c < (unsigned)N) v0 = (float)(p.mode ? I[(int64_t)c * N + tt] : I[(int64_t)tt * N + c]);
      if ((unsigned)(c + 1) < (unsigned)N) v1 = (float)(p.mode ? I[(int64_t)(c + 1) * N + tt] : I[(int64_t)tt * N + c + 1]);
      a0 = fmaf((float)(1.0 - f), v0, a0);
      a1 = fmaf((float)f, v1, a1);
      if ((tt + 1) % g.chunk_fwd == 0 || tt == N - 1) {
        acc += (double)(a0 + a1);
        a0 = a1 = 0.f;
      }
    }
    sino[ray] = (T)((float)p.w * (float)acc);
    return;
  }
  for (int tt = 0; tt < N; ++tt) {
    int c;
    double f;
    ref_tap<WEIGHTS>(p, g, a, d, tt, c, f);
    double v0 = 0.0, v1 = 0.0;
    if ((unsigned)c < (unsigned)N) v0 = (double)(p.mode ? I[(int64_t)c * N + tt] : I[(int64_t)tt * N + c]);
    if ((unsigned)(c + 1) < (unsigned)N) v1 = (double)(p.mode ? I[(int64_t)(c + 1) * N + tt] : I[(int64_t)tt * N + c + 1]);
    acc += (1.0 - f) * v0 + f * v1;
  }
  sino[ray] = (T)(p.w * acc);
}

template <class T, int WEIGHTS>
__global__ __launch_bounds__(NT) void k_ref_radon_adj(const T* __restrict__ sino, T* __restrict__ img, RadonRefGeom g) {
  const int64_t pix = (int64_t)blockIdx.x * NT + threadIdx.x;
  const int N = g.N;
  if (pix >= (int64_t)N * N) return;
  const int frame = blockIdx.y;
  const int i = (int)(pix / N), j = (int)(pix - (int64_t)i * N);
  const double sdh = 0.5 * (double)(g.nd - 1);
  double acc = 0.0;
  float af32[4] = {0.f, 0.f, 0.f, 0.f};                      // chunk_adj > 0: fp32 accumulators by candidate position (the product: 3)
  for (int af = 0; af < g.na; ++af) {
    const int a = frame * g.na + af;
    const RadonRefAngle p = g.ang[a];
    const int tt = p.mode ? j : i, col = p.mode ? i : j;
    // rays whose coordinate at tt is within one column of `col`: |dq/dd| = |inv| >= 1, so they are among the four around d*
    const double dstar = ((double)col - p.k0 - (double)tt * p.dq) / p.inv + sdh;
    const int d0 = (int)floor(dstar);
    double s = 0.0;
    if (WEIGHTS == 1 && g.chunk_adj < 0) {
      // the product's rule (k_radon_adj_tile): the nearest ray dn by its own table entry (exact t0), its two neighbours at t0 -+ |inv|
      int dn = 0;
      bool found = false;
      long long tbest = 0;
      for (int d = d0 - 1; d <= d0 + 2; ++d) {
        if (d < -A32_PAD || d >= g.nd + A32_PAD) continue;
        const unsigned Q = g.A32[(int64_t)a * (g.nd + 2 * A32_PAD) + d + A32_PAD] + g.B32[(int64_t)a * g.npad + tt] - ((unsigned)col << QF);
        const long long t = (long long)(int)Q;
        if (!found || llabs(t) < llabs(tbest)) { dn = d; tbest = t; found = true; }
      }
      if (!found) continue;
      const double t0 = (double)tbest * (1.0 / 16777216.0);
      const double c1 = g.chunk_adj == -2 ? (double)(float)(1.0 - fabs(p.inv)) : 1.0 - fabs(p.inv);
      const int sg = p.inv < 0.0 ? -1 : 1;                           // the ray on the larger-q side is dn + sg
      auto smp = [&](int d) -> double { return (d >= 0 && d < g.nd) ? (double)sino[(int64_t)a * g.nd + d] : 0.0; };
      const double wm = fmin(fmax(c1 + t0, 0.0), 1.0), wp = fmin(fmax(c1 - t0, 0.0), 1.0);
      s = (1.0 - fabs(t0)) * smp(dn) + wm * smp(dn - sg) + wp * smp(dn + sg);
      acc += p.w * s;
      continue;
    }
    for (int d = d0 - 1; d <= d0 + 2; ++d) {
      if (d < 0 || d >= g.nd) continue;
      int c;
      double f;
      ref_tap<WEIGHTS>(p, g, a, d, tt, c, f);
      const double wt = (c == col) ? 1.0 - f : (c + 1 == col) ? f : 0.0;
      if (wt == 0.0) continue;
      if (g.chunk_adj > 0) {
        const float ws = (float)p.w * (float)sino[(int64_t)a * g.nd + d];        // the record's pre-weighted sample
        af32[d - d0 + 1] = fmaf((float)wt, ws, af32[d - d0 + 1]);
      } else {
        s += wt * (double)sino[(int64_t)a * g.nd + d];
      }
    }
    if (g.chunk_adj > 0) {
      if ((af + 1) % g.chunk_adj == 0 || af == g.na - 1) {
        acc += (double)((af32[0] + af32[1]) + (af32[2] + af32[3]));
        af32[0] = af32[1] = af32[2] = af32[3] = 0.f;
      }
    } else {
      acc += p.w * s;
    }
  }
  img[(int64_t)frame * N * N + pix] = (T)acc;
}

// ---------------------------------------------------------------------------------------------- vector kernels on T
// out = a x + b z: float64 coefficients and products, one rounding of the result (radon2d.hip epi_combine, on == 2), fused sum of
// the squares of the ROUNDED outputs in float64.  x may alias out.
template <class T>
__global__ __launch_bounds__(NT) void k_ref_axpby(int64_t n, Coef ca, const T* x, Coef cb, const T* z, T* out, double* __restrict__ partials) {
  __shared__ double lds[NT / 64];
  const double a = coef_eval(ca), b = z ? coef_eval(cb) : 0.0;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    const T o = (T)(z ? fma(a, (double)x[i], b * (double)z[i]) : a * (double)x[i]);
    out[i] = o;
    acc += (double)o * (double)o;
  }
  acc = block_sum<NT>(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

template <class T>
__global__ __launch_bounds__(NT) void k_ref_nrm2sq(int64_t n, const T* __restrict__ x, double* __restrict__ partials) {
  __shared__ double lds[NT / 64];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) acc += (double)x[i] * (double)x[i];
  acc = block_sum<NT>(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

inline int ref_grid(int64_t n) {
  int64_t want = (n + NT - 1) / NT;
  if (want > kMaxPartialBlocks) want = kMaxPartialBlocks;
  return want < 1 ? 1 : (int)want;
}

template <class T>
int ref_axpby(int64_t n, Coef a, const T* x, Coef b, const T* z, T* out, double* sumsq, hipStream_t s) {
  const int grid = ref_grid(n);
  double* part = nullptr;
  if (int rc = scratch_doubles(s, grid, &part)) return rc;
  hipLaunchKernelGGL((k_ref_axpby<T>), dim3(grid), dim3(NT), 0, s, n, a, x, b, z, out, part);
  TRK_LAUNCH_CHECK();
  return sumsq ? finalize_sums(part, grid, 1, 1, sumsq, s) : TRK_OK;
}

template <class T>
int ref_nrm2sq(int64_t n, const T* x, double* out, hipStream_t s) {
  const int grid = ref_grid(n);
  double* part = nullptr;
  if (int rc = scratch_doubles(s, grid, &part)) return rc;
  hipLaunchKernelGGL((k_ref_nrm2sq<T>), dim3(grid), dim3(NT), 0, s, n, x, part);
  TRK_LAUNCH_CHECK();
  return finalize_sums(part, grid, 1, 1, out, s);
}

template <class T, int WEIGHTS>
int ref_radon_launch(const RadonRefGeom& g, int tr, const T* x, T* y, hipStream_t s) {
  if (!tr) {
    const int64_t nrays = (int64_t)g.nt * g.na * g.nd;
    hipLaunchKernelGGL((k_ref_radon_fwd<T, WEIGHTS>), dim3((unsigned)((nrays + NT - 1) / NT)), dim3(NT), 0, s, x, y, g);
  } else {
    const int64_t npix = (int64_t)g.N * g.N;
    hipLaunchKernelGGL((k_ref_radon_adj<T, WEIGHTS>), dim3((unsigned)((npix + NT - 1) / NT), g.nt), dim3(NT), 0, s, x, y, g);
  }
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

template <class T>
int ref_radon(const RadonRefGeom& g, int tr, int weights, const T* x, T* y, hipStream_t s) {
  return weights ? ref_radon_launch<T, 1>(g, tr, x, y, s) : ref_radon_launch<T, 0>(g, tr, x, y, s);
}

// ---------------------------------------------------------------------------------------------- the chain on T
// Work vectors inside `work` (caller-owned): U0, U1 (m), V0, V1 (n), w (n).  x_hist: n_iter rows of n — row k is the damped-LSQR
// iterate after k + 1 Golub-Kahan steps (the reference reports rows 1 .. n_iter-1: it forms no iterate at the first step,
// Hybrid_LSQR.py:77-78).  AB: 2 n_iter + 1 doubles as krylov.GKState; st: 8 doubles (two rotation-state slots).
template <class T>
int gk_lsqr_chain(trk_op* op, const RadonRefGeom& g, int weights, const T* b, int n_iter, double lambda, T* x_hist, T* work,
                  double* AB, double* st, hipStream_t s) {
  const int64_t m = op->rows, n = op->cols;
  T* U[2] = {work, work + m};
  T* V[2] = {work + 2 * m, work + 2 * m + n};
  T* w = work + 2 * m + 2 * n;
  TRK_HIP(hipMemcpyAsync(U[0], b, sizeof(T) * (size_t)m, hipMemcpyDeviceToDevice, s));
  if (int rc = ref_nrm2sq<T>(m, U[0], AB, s)) return rc;
  const double damp = std::sqrt(lambda);
  for (int k = 0; k < n_iter; ++k) {
    T* u = U[k & 1];
    T* un = U[(k + 1) & 1];
    T* v = V[k & 1];
    T* vp = V[(k + 1) & 1];
    double* bk2 = AB + 2 * k;
    double* a2 = AB + 2 * k + 1;
    double* b2 = AB + 2 * k + 2;
    // V[k] = (1/beta_k) A^T U[k] - (beta_k/alpha_{k-1}) V[k-1]          (the coefficients of core.hip trk_gk_step)
    if (int rc = ref_radon<T>(g, 1, weights, u, v, s)) return rc;
    if (int rc = ref_axpby<T>(n, Coef{1.0, nullptr, bk2, TRK_SQRT_DEN}, v,
                              k == 0 ? Coef{0.0, nullptr, nullptr, 0} : Coef{-1.0, bk2, AB + 2 * k - 1, TRK_SQRT_NUM | TRK_SQRT_DEN},
                              k == 0 ? (const T*)nullptr : vp, v, a2, s))
      return rc;
    // U[k+1] = (1/alpha_k) A V[k] - (alpha_k/beta_k) U[k]
    if (int rc = ref_radon<T>(g, 0, weights, v, un, s)) return rc;
    if (int rc = ref_axpby<T>(m, Coef{1.0, nullptr, a2, TRK_SQRT_DEN}, un, Coef{-1.0, a2, bk2, TRK_SQRT_NUM | TRK_SQRT_DEN}, u, un, b2, s))
      return rc;
    // x_{k+1} from x_k by the short recurrence (the production template on T)
    T* x_out = x_hist + (int64_t)k * n;
    const T* x_in = k == 0 ? nullptr : x_hist + (int64_t)(k - 1) * n;
    if (int rc = lsqr_damped_update_any(sizeof(T), v, w, x_in, x_out, n, a2, b2, AB, damp, st + 4 * ((k + 1) & 1), st + 4 * (k & 1),
                                        k == 0 ? 1 : 0, s))
      return rc;
  }
  return TRK_OK;
}

}  // namespace

namespace trk {

int radon_ref_apply_f32(const RadonRefGeom& g, int tr, int weights, const float* x, float* y, hipStream_t s) {
  return ref_radon<float>(g, tr, weights, x, y, s);
}

int ref_axpby_f32(int64_t n, Coef a, const float* x, Coef b, const float* z, float* out, double* sumsq, hipStream_t s) {
  return ref_axpby<float>(n, a, x, b, z, out, sumsq, s);
}

}  // namespace trk

extern "C" {

int trk_radon2d_apply_ref(trk_op* op, int transpose, int elem_bytes, int weights, const void* x, void* y, trk_stream stream) {
  TRK_REQUIRE(op && x && y && x != y, "trk_radon2d_apply_ref: NULL / aliased argument");
  TRK_REQUIRE(elem_bytes == 4 || elem_bytes == 8, "trk_radon2d_apply_ref: elem_bytes must be 4 (float) or 8 (double)");
  TRK_REQUIRE(weights == 0 || weights == 1, "trk_radon2d_apply_ref: weights 0 (float64 geometry) or 1 (the fixed-point tables)");
  RadonRefGeom g;
  if (!radon_ref_geometry(op, &g)) return fail(TRK_EINVAL, "trk_radon2d_apply_ref: not a parallel-beam handle");
  hipStream_t s = (hipStream_t)stream;
  if (elem_bytes == 4) return ref_radon<float>(g, transpose ? 1 : 0, weights, (const float*)x, (float*)y, s);
  return ref_radon<double>(g, transpose ? 1 : 0, weights, (const double*)x, (double*)y, s);
}

int trk_ref_axpby(int elem_bytes, int64_t n, double ca, const double* a_num, const double* a_den, int a_flags, const void* x,
                  double cb, const double* b_num, const double* b_den, int b_flags, const void* z, void* out, double* sumsq,
                  trk_stream stream) {
  TRK_REQUIRE(x && out && n >= 0, "trk_ref_axpby: NULL argument");
  TRK_REQUIRE(elem_bytes == 4 || elem_bytes == 8, "trk_ref_axpby: elem_bytes must be 4 or 8");
  const Coef a{ca, a_num, a_den, a_flags}, b{cb, b_num, b_den, b_flags};
  hipStream_t s = (hipStream_t)stream;
  if (elem_bytes == 4) return ref_axpby<float>(n, a, (const float*)x, b, (const float*)z, (float*)out, sumsq, s);
  return ref_axpby<double>(n, a, (const double*)x, b, (const double*)z, (double*)out, sumsq, s);
}

int trk_gk_lsqr_chain(trk_op* op, int elem_bytes, int weights, const void* b, int n_iter, double lambda, void* x_hist, void* work,
                      double* AB, double* state8, trk_stream stream) {
  TRK_REQUIRE(op && b && x_hist && work && AB && state8, "trk_gk_lsqr_chain: NULL argument");
  TRK_REQUIRE(elem_bytes == 4 || elem_bytes == 8, "trk_gk_lsqr_chain: elem_bytes must be 4 (float) or 8 (double)");
  TRK_REQUIRE(weights == 0 || weights == 1, "trk_gk_lsqr_chain: weights 0 (float64 geometry) or 1 (the fixed-point tables)");
  TRK_REQUIRE(n_iter >= 1 && lambda >= 0.0, "trk_gk_lsqr_chain: need n_iter >= 1, lambda >= 0");
  RadonRefGeom g;
  if (!radon_ref_geometry(op, &g)) return fail(TRK_EINVAL, "trk_gk_lsqr_chain: not a parallel-beam handle");
  hipStream_t s = (hipStream_t)stream;
  if (elem_bytes == 4) return gk_lsqr_chain<float>(op, g, weights, (const float*)b, n_iter, lambda, (float*)x_hist, (float*)work, AB, state8, s);
  return gk_lsqr_chain<double>(op, g, weights, (const double*)b, n_iter, lambda, (double*)x_hist, (double*)work, AB, state8, s);
}

}  // extern "C"
