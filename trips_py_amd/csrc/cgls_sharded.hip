// cgls_sharded.hip — CGLS for problems whose unknowns are spread over ranks (frames of a dynamic problem, io.py:420), with
// ONE all-reduce per iteration instead of the two dependent ones of the recurrence as written (CGLS.py:61 ||A p||^2, :70 ||A^T r||^2).
//
// Per rank the work of a C5 iteration is a few microseconds of kernels; what separates them is the latency of the collectives, so
// the two reductions are merged (the Chronopoulos-Gear arrangement of the same recurrence): with q = A t_{k-1} formed explicitly,
//     w_k = A p_k = q + beta w_{k-1},      delta_k = ||w_k||^2 = ||q||^2 + 2 beta <q, w_{k-1}> + beta^2 ||w_{k-1}||^2,
// so gamma_{k-1} = ||t_{k-1}||^2, ||q||^2, <q, w_{k-1}> and ||w_{k-1}||^2 — all formed from vectors iteration k-1 left behind —
// travel in one all-reduce of four doubles, and everything after it is local (||w_{k-1}||^2 is summed from the vector itself, not
// carried over as delta_{k-1}: the expansion is then that of the very w_k the update forms, and no error compounds through beta^2):
//     beta = gamma_{k-1} / gamma_{k-2} (0 for k = 1);  p_k = t_{k-1} + beta p_{k-1};  w_k = q + beta w_{k-1};
//     alpha = gamma_{k-1} / delta_k;  x_k = x_{k-1} + alpha p_k;  r_k = r_{k-1} - alpha w_k;  t_k = A^T r_k.
// Same operator applies per iteration (one A, one A^T), same iterates in exact arithmetic; in floating point w_k carries the
// rounding of its recurrence instead of that of a fresh product (tests: iterates within 1e-5 of the two-reduction form).
// The three norms the reference records per iterate (CGLS.py:76-80) are only REPORTED with tol = 0, so they stay local block
// partials and are summed over blocks and ranks once, after the solve.
#include "trk_internal.h"

using namespace trk;

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float4 ld4(const float* p, int64_t i4) { return reinterpret_cast<const float4*>(p)[i4]; }
__device__ __forceinline__ void st4(float* p, int64_t i4, float4 v) { reinterpret_cast<float4*>(p)[i4] = v; }

inline int grid_for(int64_t n) {
  int64_t want = (n + (int64_t)NT * 4 - 1) / ((int64_t)NT * 4);
  int64_t cap = (int64_t)cu_count() * 4;
  if (cap > kMaxPartialBlocks) cap = kMaxPartialBlocks;
  if (want > cap) want = cap;
  return want < 1 ? 1 : (int)want;
}

// partials[block][3] = sum q*q, sum q*w, sum w*w  (w == NULL: the last two are 0)
template <bool VEC>
__global__ __launch_bounds__(NT) void k_dot_pair(const float* __restrict__ q, const float* __restrict__ w, int64_t n,
                                                 double* __restrict__ partials) {
  __shared__ double lds[(NT / 64) * 3];
  double acc[3] = {0.0, 0.0, 0.0};
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  int64_t tail = 0;
  if (VEC) {
    const int64_t n4 = n >> 2;
    tail = n4 << 2;
    for (int64_t i = tid; i < n4; i += nth) {
      const float4 a = ld4(q, i);
      acc[0] += (double)a.x * a.x + (double)a.y * a.y + (double)a.z * a.z + (double)a.w * a.w;
      if (w) {
        const float4 b = ld4(w, i);
        acc[1] += (double)a.x * b.x + (double)a.y * b.y + (double)a.z * b.z + (double)a.w * b.w;
        acc[2] += (double)b.x * b.x + (double)b.y * b.y + (double)b.z * b.z + (double)b.w * b.w;
      }
    }
  }
  for (int64_t i = tail + tid; i < n; i += nth) {
    const double a = q[i];
    acc[0] += a * a;
    if (w) {
      const double b = w[i];
      acc[1] += a * b;
      acc[2] += b * b;
    }
  }
  const double s = block_sum_many<NT, 3>(acc, lds);
  if (threadIdx.x < 3) partials[blockIdx.x * 3 + threadIdx.x] = s;
}

// Everything the iteration's one exchange carries: G[1..3] = sum q*q, sum q*w, sum w*w, and G[0] = the sum of the n_g raw block
// partials of ||t||^2 the adjoint kernel left (n_g = 0: G[0] is finished already and stays).  By ONE workgroup for vectors of up to
// kOneBlockMax floats (one launch instead of partials + finalize + the adjoint's finalize), else by k_dot_pair's partials and ONE
// finalize launch for all four (k_sharded_finalize).  Measured on the C5 shape (tools/c5_cgls_rate.py): a single workgroup over a
// rank's 122 880 sinogram samples took longer than the three launches it replaced (CGLS 15.4 k -> 12.0 k iterations/s) — one CU's
// loads in flight do not stream a megabyte — hence the small limit.  Fixed order: reproducible.
constexpr int NT1 = 1024;
constexpr int64_t kOneBlockMax = (int64_t)1 << 14;    // beyond: one workgroup's few loads in flight make it the slowest kernel of the iteration (measured, below)
template <bool VEC>
__global__ __launch_bounds__(NT1) void k_sharded_scalars(const float* __restrict__ q, const float* __restrict__ w, int64_t n,
                                                         const double* __restrict__ pg, int n_g, double* __restrict__ G) {
  __shared__ double lds[(NT1 / 64) * 4];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < n_g; i += NT1) acc[0] += pg[i];
  int64_t tail = 0;
  if (VEC) {
    const int64_t n4 = n >> 2;
    tail = n4 << 2;
    for (int64_t i = threadIdx.x; i < n4; i += NT1) {
      const float4 a = ld4(q, i);
      acc[1] += (double)a.x * a.x + (double)a.y * a.y + (double)a.z * a.z + (double)a.w * a.w;
      if (w) {
        const float4 b = ld4(w, i);
        acc[2] += (double)a.x * b.x + (double)a.y * b.y + (double)a.z * b.z + (double)a.w * b.w;
        acc[3] += (double)b.x * b.x + (double)b.y * b.y + (double)b.z * b.z + (double)b.w * b.w;
      }
    }
  }
  for (int64_t i = tail + threadIdx.x; i < n; i += NT1) {
    const double a = q[i];
    acc[1] += a * a;
    if (w) {
      const double b = w[i];
      acc[2] += a * b;
      acc[3] += b * b;
    }
  }
  const double s = block_sum_many<NT1, 4>(acc, lds);
  if (threadIdx.x < 4 && (threadIdx.x > 0 || n_g > 0)) G[threadIdx.x] = s;
}

// G[0] = sum of pg[0 .. n_g) (skipped for n_g = 0), G[1 + v] = sum over blocks of part[block][v]: one workgroup per output
__global__ __launch_bounds__(256) void k_sharded_finalize(const double* __restrict__ part, int nblocks, const double* __restrict__ pg,
                                                          int n_g, double* __restrict__ G) {
  __shared__ double lds[4];
  const int o = blockIdx.x;
  if (o == 0 && n_g == 0) return;
  double v = 0.0;
  if (o == 0)
    for (int i = threadIdx.x; i < n_g; i += 256) v += pg[i];
  else
    for (int b = threadIdx.x; b < nblocks; b += 256) v += part[(size_t)b * 3 + (o - 1)];
  v = block_sum<256>(v, lds);
  if (threadIdx.x == 0) G[o] = v;
}

// The local half of the merged iteration.  G = {gamma_{k-1}, ||q||^2, <q, w_{k-1}>, ||w_{k-1}||^2} summed over the ranks;
// gprev = gamma_{k-2} (unused when first).  Every thread evaluates the three scalars itself from the same five doubles (grid-
// uniform loads), block 0 publishes delta_k and gamma_{k-1}.  partials[block][3] = ||x_k||^2, ||alpha p_k||^2, ||x_k - x_true||^2.
template <bool HAS_XT, bool VEC>
__global__ __launch_bounds__(NT) void k_cgls_sharded_update(int64_t n, int64_t m, const double* __restrict__ G,
                                                            const double* gprev, int first,
                                                            const float* x, float* p, const float* __restrict__ t, float* x_new,
                                                            float* r, const float* __restrict__ q, float* w,
                                                            const float* __restrict__ x_true, double* pub_delta,
                                                            double* pub_gamma, double* __restrict__ partials) {
  __shared__ double lds[(NT / 64) * 3];
  const double g = G[0], qq = G[1], qw = G[2], ww = G[3];
  const double beta_d = first ? 0.0 : g / *gprev;
  const double delta = first ? qq : qq + 2.0 * beta_d * qw + beta_d * beta_d * ww;
  const float beta = (float)beta_d, alpha = (float)(g / delta);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *pub_delta = delta;
    *pub_gamma = g;
  }
  double acc[3] = {0.0, 0.0, 0.0};
  const int64_t tid = (int64_t)blockIdx.x * NT + threadIdx.x, nth = (int64_t)gridDim.x * NT;
  int64_t ntail = 0, mtail = 0;
  if (VEC) {
    const int64_t n4 = n >> 2, m4 = m >> 2;
    ntail = n4 << 2;
    mtail = m4 << 2;
    for (int64_t i = tid; i < n4; i += nth) {
      const float4 tv = ld4(t, i), xv = ld4(x, i);
      float4 pv = tv;
      if (!first) {
        const float4 po = ld4(p, i);
        pv = make_float4(fmaf(beta, po.x, tv.x), fmaf(beta, po.y, tv.y), fmaf(beta, po.z, tv.z), fmaf(beta, po.w, tv.w));
      }
      st4(p, i, pv);
      const float4 d = make_float4(alpha * pv.x, alpha * pv.y, alpha * pv.z, alpha * pv.w);
      const float4 xn = make_float4(xv.x + d.x, xv.y + d.y, xv.z + d.z, xv.w + d.w);
      st4(x_new, i, xn);
      acc[0] += (double)xn.x * xn.x + (double)xn.y * xn.y + (double)xn.z * xn.z + (double)xn.w * xn.w;
      acc[1] += (double)d.x * d.x + (double)d.y * d.y + (double)d.z * d.z + (double)d.w * d.w;
      if (HAS_XT) {
        const float4 e = ld4(x_true, i);
        const double e0 = (double)xn.x - e.x, e1 = (double)xn.y - e.y, e2 = (double)xn.z - e.z, e3 = (double)xn.w - e.w;
        acc[2] += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
      }
    }
    for (int64_t i = tid; i < m4; i += nth) {
      const float4 qv = ld4(q, i);
      float4 wv = qv;
      if (!first) {
        const float4 wo = ld4(w, i);
        wv = make_float4(fmaf(beta, wo.x, qv.x), fmaf(beta, wo.y, qv.y), fmaf(beta, wo.z, qv.z), fmaf(beta, wo.w, qv.w));
      }
      st4(w, i, wv);
      float4 rv = ld4(r, i);
      rv.x = fmaf(-alpha, wv.x, rv.x);
      rv.y = fmaf(-alpha, wv.y, rv.y);
      rv.z = fmaf(-alpha, wv.z, rv.z);
      rv.w = fmaf(-alpha, wv.w, rv.w);
      st4(r, i, rv);
    }
  }
  for (int64_t i = ntail + tid; i < n; i += nth) {
    const float pv = first ? t[i] : fmaf(beta, p[i], t[i]);
    p[i] = pv;
    const float d = alpha * pv, xn = x[i] + d;
    x_new[i] = xn;
    acc[0] += (double)xn * xn;
    acc[1] += (double)d * d;
    if (HAS_XT) {
      const double e = (double)xn - x_true[i];
      acc[2] += e * e;
    }
  }
  for (int64_t i = mtail + tid; i < m; i += nth) {
    const float wv = first ? q[i] : fmaf(beta, w[i], q[i]);
    w[i] = wv;
    r[i] = fmaf(-alpha, wv, r[i]);
  }
  const double s = block_sum_many<NT, 3>(acc, lds);
  if (threadIdx.x < 3) partials[blockIdx.x * 3 + threadIdx.x] = s;
}

}  // namespace

extern "C" {

int trk_dot_pair(const float* q, const float* w, int64_t n, double* out3, trk_stream st) {
  TRK_REQUIRE(q && out3 && n >= 0, "trk_dot_pair: bad argument");
  hipStream_t s = (hipStream_t)st;
  const int grid = grid_for(n);
  double* part = nullptr;
  if (int rc = scratch_doubles(s, (size_t)grid * 3, &part)) return rc;
  if (aligned16(q) && (!w || aligned16(w)))
    hipLaunchKernelGGL((k_dot_pair<true>), dim3(grid), dim3(NT), 0, s, q, w, n, part);
  else
    hipLaunchKernelGGL((k_dot_pair<false>), dim3(grid), dim3(NT), 0, s, q, w, n, part);
  TRK_LAUNCH_CHECK();
  return finalize_sums(part, grid, 3, 3, out3, s);
}

int trk_cgls_sharded_scalars(const float* q, const float* w, int64_t m, const double* gamma_partials, int n_gamma, double* G4,
                             trk_stream st) {
  TRK_REQUIRE(q && G4 && m >= 0 && n_gamma >= 0 && (n_gamma == 0 || gamma_partials), "trk_cgls_sharded_scalars: bad argument");
  hipStream_t s = (hipStream_t)st;
  if (m <= kOneBlockMax && n_gamma <= (1 << 16)) {
    if (aligned16(q) && (!w || aligned16(w)))
      hipLaunchKernelGGL((k_sharded_scalars<true>), dim3(1), dim3(NT1), 0, s, q, w, m, gamma_partials, n_gamma, G4);
    else
      hipLaunchKernelGGL((k_sharded_scalars<false>), dim3(1), dim3(NT1), 0, s, q, w, m, gamma_partials, n_gamma, G4);
    TRK_LAUNCH_CHECK();
    return TRK_OK;
  }
  const int grid = grid_for(m);
  double* part = nullptr;
  if (int rc = scratch_doubles(s, (size_t)grid * 3, &part)) return rc;
  if (aligned16(q) && (!w || aligned16(w)))
    hipLaunchKernelGGL((k_dot_pair<true>), dim3(grid), dim3(NT), 0, s, q, w, m, part);
  else
    hipLaunchKernelGGL((k_dot_pair<false>), dim3(grid), dim3(NT), 0, s, q, w, m, part);
  hipLaunchKernelGGL(k_sharded_finalize, dim3(4), dim3(256), 0, s, part, grid, gamma_partials, n_gamma, G4);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_cgls_sharded_update(int64_t n, int64_t m, const double* G4, const double* gamma_prev, int first, const float* x, float* p, const float* t, float* x_new, float* r, const float* q,
                            float* w, const float* x_true, double* publish_delta, double* publish_gamma,
                            double* norm_partials, int capacity_blocks, int* n_blocks, trk_stream st) {
  TRK_REQUIRE(G4 && x && p && t && x_new && r && q && w && publish_delta && publish_gamma && norm_partials && n_blocks,
              "trk_cgls_sharded_update: NULL argument");
  TRK_REQUIRE(first || gamma_prev, "trk_cgls_sharded_update: gamma_prev needed after the first step");
  TRK_REQUIRE(n >= 0 && m >= 0, "trk_cgls_sharded_update: negative size");
  TRK_REQUIRE(x_new != p && x_new != t, "trk_cgls_sharded_update: x_new must not alias p or t");
  hipStream_t s = (hipStream_t)st;
  const int grid = grid_for(n > m ? n : m);
  TRK_REQUIRE(grid <= capacity_blocks, "trk_cgls_sharded_update: partial buffer too small (%d blocks needed)", grid);
  *n_blocks = grid;
  const bool vec = aligned16(x) && aligned16(p) && aligned16(t) && aligned16(x_new) && aligned16(r) && aligned16(q) &&
                   aligned16(w) && (!x_true || aligned16(x_true));
#define SU(XT, VC)                                                                                                      \
  hipLaunchKernelGGL((k_cgls_sharded_update<XT, VC>), dim3(grid), dim3(NT), 0, s, n, m, G4, gamma_prev, first, x, \
                     p, t, x_new, r, q, w, x_true, publish_delta, publish_gamma, norm_partials)
  if (x_true) { if (vec) SU(true, true); else SU(true, false); }
  else        { if (vec) SU(false, true); else SU(false, false); }
#undef SU
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_cgls_iterate_sharded(trk_op* A, trk_comm* comm, int k_first, int n_iters, float* p, float* r, float* t, float* q,
                             float* w, float* X, int64_t x_ld, int keep_history, const float* x_prev, const float* x_true,
                             double* S, double* G4, double* NP, int np_capacity_blocks, int* n_np_inout, double* PG, int pcap,
                             int* n_g_inout, trk_stream stream) {
  TRK_REQUIRE(A && p && r && t && q && w && X && x_prev && S && G4 && NP && n_np_inout,
              "trk_cgls_iterate_sharded: NULL argument");
  TRK_REQUIRE(k_first >= 1 && n_iters >= 0, "trk_cgls_iterate_sharded: need k_first >= 1, n_iters >= 0");
  TRK_REQUIRE(!PG || (pcap > 0 && n_g_inout), "trk_cgls_iterate_sharded: PG needs its capacity and partial count");
  const int64_t m = A->rows, n = A->cols;
  int n_np = *n_np_inout;
  // ||t||^2 of the adjoint apply left as raw block partials for the scalars kernel to add up (operators with a fused apply)
  const bool raw = PG && A->apply_fused;
  int n_g = raw ? *n_g_inout : 0;
  for (int k = k_first; k < k_first + n_iters; ++k) {
    double* row = S + 5 * (int64_t)k;                          // [delta_k, gamma_k, ||x||^2, ||dx||^2, ||x-xt||^2]
    double* gpub = (k == 1) ? S : row - 4;                     // gamma_{k-1}
    const double* gprev = (k == 2) ? S : row - 9;              // gamma_{k-2} (k >= 2)
    float* x_new = X + (int64_t)(keep_history ? (k - 1) : ((k - 1) & 1)) * x_ld;
    int rc = trk_op_apply(A, 0, t, 0, q, 0, 1, nullptr, stream);                                   // q = A t_{k-1}
    if (rc) return rc;
    rc = trk_cgls_sharded_scalars(q, k == 1 ? nullptr : w, m, PG, n_g, G4, stream);      // gamma_{k-1}, ||q||^2, <q, w>, ||w||^2 (local)
    if (rc) return rc;
    if (comm) {
      rc = trk_allreduce_f64(comm, G4, 4, stream);                                                 // the iteration's ONE exchange
      if (rc) return rc;
    }
    rc = trk_cgls_sharded_update(n, m, G4, gprev, k == 1, x_prev, p, t, x_new, r, q, w, x_true, row, gpub,
                                 NP + 3 * (int64_t)n_np * (k - 1), np_capacity_blocks, &n_np, stream);
    if (rc) return rc;
    if (raw)                                                                                        // t_k = A^T r_k, local ||t_k||^2
      rc = trk_op_apply_fused(A, 1, r, nullptr, 0.0, nullptr, 0, nullptr, 0, nullptr, t, PG, pcap, &n_g, stream);
    else
      rc = trk_op_apply(A, 1, r, 0, t, 0, 1, G4, stream);
    if (rc) return rc;
    x_prev = x_new;
  }
  *n_np_inout = n_np;
  if (raw) *n_g_inout = n_g;
  return TRK_OK;
}

}  // extern "C"
