// tvops.hip — matrix-free first-difference regularisers, replacing the scipy.sparse CSR matrices of
// trips/utilities/operators.py:24-45.
//
//   2-D, N x N image x (row-major):   L x = [ H ; V ],   H[i][j] = x[i][j] - x[i][j+1]   (N rows x (N-1)),  i-major
//                                                         V[i][j] = x[i][j] - x[i+1][j]   ((N-1) rows x N)
//   space-time, nt frame-major frames:  [ L2 x_0 ; ... ; L2 x_{nt-1} ;  x_0 - x_1 ; ... ; x_{nt-2} - x_{nt-1} ]
//
// Both directions are written in GATHER form (one thread per output element, no atomics):
//   (L^T y)[i][j] = H[i][j] - H[i][j-1] + V[i][j] - V[i-1][j]      (terms outside the index range are zero)
// HBM-bound streaming stencils: 4n read + 8n write forward, 8n + 4n transposed; neighbours come from L1/L2.
// For a time-sharded dynamic problem each rank owns whole frames; the temporal rows need the first frame of the next
// rank (forward) and the last temporal block of the previous rank (transpose): trk_spacetime_set_halo.
#include "trk_internal.h"

using namespace trk;

namespace {

constexpr int NT = 256;

inline int grid_for(int64_t n) {
  int64_t want = (n + NT - 1) / NT;
  const int64_t cap = kMaxPartialBlocks;
  if (want > cap) want = cap;
  return (int)(want < 1 ? 1 : want);
}

// blockIdx.y = frame / batch vector
template <bool SUMSQ>
__global__ __launch_bounds__(NT) void k_d2_fwd(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                               int64_t ldy, int N, double* __restrict__ partials) {
  __shared__ double red[NT / 64];
  x += (int64_t)blockIdx.y * ldx;
  y += (int64_t)blockIdx.y * ldy;
  const int64_t npix = (int64_t)N * N;
  float* __restrict__ yh = y;
  float* __restrict__ yv = y + (int64_t)N * (N - 1);
  double ss = 0.0;
  for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < npix; idx += (int64_t)gridDim.x * NT) {
    const int i = (int)(idx / N), j = (int)(idx - (int64_t)i * N);
    const float c = x[idx];
    if (j < N - 1) {
      const float h = c - x[idx + 1];
      yh[(int64_t)i * (N - 1) + j] = h;
      if (SUMSQ) ss += (double)h * h;
    }
    if (i < N - 1) {
      const float v = c - x[idx + N];
      yv[idx] = v;
      if (SUMSQ) ss += (double)v * v;
    }
  }
  if (SUMSQ) {
    ss = block_sum<NT>(ss, red);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = ss;
  }
}

// out = L2^T y  (+ optional temporal terms of the space-time operator: + tcur - tprev)
template <bool SUMSQ, bool TEMPORAL>
__global__ __launch_bounds__(NT) void k_d2_adj(const float* __restrict__ y, int64_t ldy, float* __restrict__ out,
                                               int64_t ldo, int N, const float* __restrict__ tcur, int64_t ldt,
                                               const float* __restrict__ tprev, int first_has_prev, int last_has_cur,
                                               const float* __restrict__ halo_prev, double* __restrict__ partials) {
  __shared__ double red[NT / 64];
  const int f = blockIdx.y, nf = gridDim.y;
  y += (int64_t)f * ldy;
  out += (int64_t)f * ldo;
  const int64_t npix = (int64_t)N * N;
  const float* __restrict__ yh = y;
  const float* __restrict__ yv = y + (int64_t)N * (N - 1);
  // temporal rows: row t = x_t - x_{t+1};  (L^T y)_t += T_t (if row t exists) - T_{t-1} (if row t-1 exists)
  const float* tc = nullptr;
  const float* tp = nullptr;
  if (TEMPORAL) {
    if (f < nf - 1 || last_has_cur) tc = tcur + (int64_t)f * ldt;
    if (f > 0) tp = tprev + (int64_t)(f - 1) * ldt;
    else if (first_has_prev) tp = halo_prev;
  }
  double ss = 0.0;
  for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < npix; idx += (int64_t)gridDim.x * NT) {
    const int i = (int)(idx / N), j = (int)(idx - (int64_t)i * N);
    float acc = 0.f;
    const int64_t hb = (int64_t)i * (N - 1) + j;
    if (j < N - 1) acc += yh[hb];
    if (j > 0) acc -= yh[hb - 1];
    if (i < N - 1) acc += yv[idx];
    if (i > 0) acc -= yv[idx - N];
    if (TEMPORAL) {
      if (tc) acc += tc[idx];
      if (tp) acc -= tp[idx];
    }
    out[idx] = acc;
    if (SUMSQ) ss += (double)acc * acc;
  }
  if (SUMSQ) {
    ss = block_sum<NT>(ss, red);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = ss;
  }
}

// temporal rows, forward: T_t = x_t - x_{t+1}; blockIdx.y = t.  The last row may take x_{t+1} from the halo.
template <bool SUMSQ>
__global__ __launch_bounds__(NT) void k_time_fwd(const float* __restrict__ x, float* __restrict__ T, int64_t npix,
                                                 int nt_local, const float* __restrict__ halo_next,
                                                 double* __restrict__ partials) {
  __shared__ double red[NT / 64];
  const int t = blockIdx.y;
  const float* a = x + (int64_t)t * npix;
  const float* b = (t + 1 < nt_local) ? x + (int64_t)(t + 1) * npix : halo_next;
  float* o = T + (int64_t)t * npix;
  double ss = 0.0;
  for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < npix; idx += (int64_t)gridDim.x * NT) {
    const float v = a[idx] - b[idx];
    o[idx] = v;
    if (SUMSQ) ss += (double)v * v;
  }
  if (SUMSQ) {
    ss = block_sum<NT>(ss, red);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = ss;
  }
}

// ------------------------------------------------------------------------------------------------ 2-D operator
struct D2Impl {
  int N;
};

int d2_apply(trk_op* op, int tr, const float* x, int64_t ldx, float* y, int64_t ldy, int batch, double* sumsq,
             hipStream_t s) {
  const int N = static_cast<D2Impl*>(op->impl)->N;
  const int gx = grid_for((int64_t)N * N);
  double* part = nullptr;
  if (sumsq)
    if (int rc = scratch_doubles(s, (size_t)gx * batch, &part)) return rc;
  dim3 grid(gx, batch);
  if (!tr) {
    if (sumsq) hipLaunchKernelGGL((k_d2_fwd<true>), grid, dim3(NT), 0, s, x, ldx, y, ldy, N, part);
    else hipLaunchKernelGGL((k_d2_fwd<false>), grid, dim3(NT), 0, s, x, ldx, y, ldy, N, part);
  } else {
    if (sumsq) hipLaunchKernelGGL((k_d2_adj<true, false>), grid, dim3(NT), 0, s, x, ldx, y, ldy, N, nullptr, (int64_t)0, nullptr, 0, 0, nullptr, part);
    else hipLaunchKernelGGL((k_d2_adj<false, false>), grid, dim3(NT), 0, s, x, ldx, y, ldy, N, nullptr, (int64_t)0, nullptr, 0, 0, nullptr, part);
  }
  TRK_LAUNCH_CHECK();
  if (sumsq) return finalize_sums(part, gx * batch, 1, 1, sumsq, s);
  return TRK_OK;
}

void d2_destroy(trk_op* op) { delete static_cast<D2Impl*>(op->impl); }

// ------------------------------------------------------------------------------------------------ space-time operator
struct STImpl {
  int N, nt, has_next, has_prev;
  const float* halo_next;  // first frame of the next rank (forward)
  const float* halo_prev;  // last temporal block of the previous rank (transpose)
};

int st_apply(trk_op* op, int tr, const float* x, int64_t ldx, float* y, int64_t ldy, int batch, double* sumsq,
             hipStream_t s) {
  auto* im = static_cast<STImpl*>(op->impl);
  const int N = im->N, nt = im->nt;
  const int64_t npix = (int64_t)N * N, ps = 2 * (int64_t)N * (N - 1);
  const int ntemp = nt - 1 + (im->has_next ? 1 : 0);
  if (im->has_next && !tr && !im->halo_next) return fail(TRK_EINVAL, "spacetime forward: halo of the next rank not set");
  if (im->has_prev && tr && !im->halo_prev) return fail(TRK_EINVAL, "spacetime transpose: halo of the previous rank not set");
  const int gx = grid_for(npix);
  double* part = nullptr;
  // block partials per vector: forward = spatial (gx*nt) + temporal (gx*ntemp) ; transpose = gx*nt
  const int per_vec = tr ? gx * nt : gx * nt + gx * (ntemp > 0 ? ntemp : 0);
  if (sumsq)
    if (int rc = scratch_doubles(s, (size_t)per_vec * batch, &part)) return rc;
  for (int b = 0; b < batch; ++b) {
    const float* xb = x + (int64_t)b * ldx;
    float* yb = y + (int64_t)b * ldy;
    double* pb = part ? part + (size_t)b * per_vec : nullptr;
    if (!tr) {
      dim3 g1(gx, nt);
      if (sumsq) hipLaunchKernelGGL((k_d2_fwd<true>), g1, dim3(NT), 0, s, xb, npix, yb, ps, N, pb);
      else hipLaunchKernelGGL((k_d2_fwd<false>), g1, dim3(NT), 0, s, xb, npix, yb, ps, N, pb);
      if (ntemp > 0) {
        dim3 g2(gx, ntemp);
        float* T = yb + (int64_t)nt * ps;
        if (sumsq) hipLaunchKernelGGL((k_time_fwd<true>), g2, dim3(NT), 0, s, xb, T, npix, nt, im->halo_next, pb + (size_t)gx * nt);
        else hipLaunchKernelGGL((k_time_fwd<false>), g2, dim3(NT), 0, s, xb, T, npix, nt, im->halo_next, pb);
      }
    } else {
      dim3 g1(gx, nt);
      const float* T = xb + (int64_t)nt * ps;
      if (sumsq) hipLaunchKernelGGL((k_d2_adj<true, true>), g1, dim3(NT), 0, s, xb, ps, yb, npix, N, T, npix, T, im->has_prev, im->has_next, im->halo_prev, pb);
      else hipLaunchKernelGGL((k_d2_adj<false, true>), g1, dim3(NT), 0, s, xb, ps, yb, npix, N, T, npix, T, im->has_prev, im->has_next, im->halo_prev, pb);
    }
    TRK_LAUNCH_CHECK();
  }
  if (sumsq) return finalize_sums(part, per_vec * batch, 1, 1, sumsq, s);
  return TRK_OK;
}

void st_destroy(trk_op* op) { delete static_cast<STImpl*>(op->impl); }

}  // namespace

extern "C" {

int trk_deriv2d_create(int N, trk_op** out) {
  TRK_REQUIRE(out && N >= 2, "trk_deriv2d_create: need N >= 2");
  auto* im = new D2Impl{N};
  *out = new trk_op{3, 2 * (int64_t)N * (N - 1), (int64_t)N * N, im, d2_apply, d2_destroy, nullptr, 0};
  return TRK_OK;
}

int trk_spacetime_create(int N, int nt_local, int has_next, int has_prev, trk_op** out) {
  TRK_REQUIRE(out && N >= 2 && nt_local >= 1, "trk_spacetime_create: need N >= 2 and nt_local >= 1");
  auto* im = new STImpl{N, nt_local, has_next ? 1 : 0, has_prev ? 1 : 0, nullptr, nullptr};
  const int64_t npix = (int64_t)N * N, ps = 2 * (int64_t)N * (N - 1);
  const int ntemp = nt_local - 1 + (has_next ? 1 : 0);
  *out = new trk_op{4, (int64_t)nt_local * ps + (int64_t)ntemp * npix, (int64_t)nt_local * npix, im, st_apply, st_destroy, nullptr, 0};
  return TRK_OK;
}

int trk_spacetime_set_halo(trk_op* op, const float* x_next, const float* y_prev) {
  TRK_REQUIRE(op && op->kind == 4, "trk_spacetime_set_halo: not a space-time operator");
  auto* im = static_cast<STImpl*>(op->impl);
  im->halo_next = x_next;
  im->halo_prev = y_prev;
  return TRK_OK;
}

}  // extern "C"
