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
constexpr int RB = 4;              // image rows per thread and batch
constexpr int kTvMaxBlocks = 4096;  // cap on the blocks (= block partials) per frame of the kernels that reduce

inline int grid_for(int64_t n) {
  int64_t want = (n + NT - 1) / NT;
  const int64_t cap = kMaxPartialBlocks;
  if (want > cap) want = cap;
  return (int)(want < 1 ? 1 : want);
}

// The 2-D stencils run one thread per image COLUMN over batches of RB rows: every load of a batch is issued before the
// first use (the kernels are latency-, then HBM-bound: memory-level parallelism is what fills the pipe), the vertical
// neighbours of the batch are loaded once, row offsets are running sums (no integer division) and every load/store of a
// wavefront is one contiguous row segment.  grid = (ceil(N/NT), row batches (grid-strided when capped), frames).
struct Grid2 {
  dim3 g;
  int per_frame;  // blocks (= block partials) per frame
};
inline Grid2 grid2(int N, int frames, bool capped) {
  const int gx = (N + NT - 1) / NT;
  int gy = (N + RB - 1) / RB;
  if (capped) {
    const int cap = kTvMaxBlocks / gx;
    if (gy > cap) gy = cap < 1 ? 1 : cap;
  }
  return {dim3(gx, gy, frames), gx * gy};
}

// blockIdx.z = frame / batch vector.  WEIGHTS: y = ((L2 x)^2 + eps2)^e instead of L2 x (k_tv_weights below).
template <bool SUMSQ, bool WEIGHTS>
__device__ __forceinline__ void d2_fwd_body(const float* __restrict__ x, float* __restrict__ y, int N, float eps2, float e,
                                            int special, double& ss) {
  float* __restrict__ yh = y;
  float* __restrict__ yv = y + (int64_t)N * (N - 1);
  const int j = blockIdx.x * NT + threadIdx.x;
  if (j >= N) return;
  const bool hr = j < N - 1;
  for (int i0 = blockIdx.y * RB; i0 < N; i0 += gridDim.y * RB) {
    const int64_t o0 = (int64_t)i0 * N + j;
    const int64_t h0 = (int64_t)i0 * (N - 1) + j;
    float xc[RB + 1], xr[RB];
#pragma unroll
    for (int t = 0; t <= RB; ++t) xc[t] = (i0 + t < N) ? x[o0 + (int64_t)t * N] : 0.f;
#pragma unroll
    for (int t = 0; t < RB; ++t) xr[t] = (hr && i0 + t < N) ? x[o0 + (int64_t)t * N + 1] : 0.f;
#pragma unroll
    for (int t = 0; t < RB; ++t) {
      const int i = i0 + t;
      if (i < N) {
        if (hr) {
          const float h = xc[t] - xr[t];
          yh[h0 + (int64_t)t * (N - 1)] = WEIGHTS ? mm_w(h, eps2, e, special) : h;
          if (SUMSQ) ss += (double)h * h;
        }
        if (i < N - 1) {
          const float v = xc[t] - xc[t + 1];
          yv[o0 + (int64_t)t * N] = WEIGHTS ? mm_w(v, eps2, e, special) : v;
          if (SUMSQ) ss += (double)v * v;
        }
      }
    }
  }
}

template <bool SUMSQ>
__global__ __launch_bounds__(NT) void k_d2_fwd(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                               int64_t ldy, int N, double* __restrict__ partials) {
  __shared__ double red[NT / 64];
  double ss = 0.0;
  d2_fwd_body<SUMSQ, false>(x + (int64_t)blockIdx.z * ldx, y + (int64_t)blockIdx.z * ldy, N, 0.f, 0.f, 0, ss);
  if (SUMSQ) {
    ss = block_sum<NT>(ss, red);
    if (threadIdx.x == 0) partials[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = ss;
  }
}

// out = L2^T y  (+ optional temporal terms of the space-time operator: + tcur - tprev)
template <bool SUMSQ, bool TEMPORAL>
__global__ __launch_bounds__(NT) void k_d2_adj(const float* __restrict__ y, int64_t ldy, float* __restrict__ out,
                                               int64_t ldo, int N, const float* __restrict__ tcur, int64_t ldt,
                                               const float* __restrict__ tprev, int first_has_prev, int last_has_cur,
                                               const float* __restrict__ halo_prev, double* __restrict__ partials) {
  __shared__ double red[NT / 64];
  const int f = blockIdx.z, nf = gridDim.z;
  y += (int64_t)f * ldy;
  out += (int64_t)f * ldo;
  const float* __restrict__ yh = y;
  const float* __restrict__ yv = y + (int64_t)N * (N - 1);
  // temporal rows: row t = x_t - x_{t+1};  (L^T y)_t += T_t (if row t exists) - T_{t-1} (if row t-1 exists)
  const float* __restrict__ tc = nullptr;
  const float* __restrict__ tp = nullptr;
  if (TEMPORAL) {
    if (f < nf - 1 || last_has_cur) tc = tcur + (int64_t)f * ldt;
    if (f > 0) tp = tprev + (int64_t)(f - 1) * ldt;
    else if (first_has_prev) tp = halo_prev;
  }
  const int j = blockIdx.x * NT + threadIdx.x;
  double ss = 0.0;
  if (j < N) {
    const bool hr = j < N - 1, hl = j > 0;
    for (int i0 = blockIdx.y * RB; i0 < N; i0 += gridDim.y * RB) {
      const int64_t o0 = (int64_t)i0 * N + j;
      const int64_t h0 = (int64_t)i0 * (N - 1) + j;
      float hc[RB], hp[RB], v[RB + 1], a[RB], b[RB];
#pragma unroll
      for (int t = 0; t <= RB; ++t) {      // v[t] = yv[i0 + t - 1][j]
        const int i = i0 + t - 1;
        v[t] = (i >= 0 && i < N - 1) ? yv[o0 + (int64_t)(t - 1) * N] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < RB; ++t) {
        const bool in = i0 + t < N;
        hc[t] = (in && hr) ? yh[h0 + (int64_t)t * (N - 1)] : 0.f;
        hp[t] = (in && hl) ? yh[h0 + (int64_t)t * (N - 1) - 1] : 0.f;
        if (TEMPORAL) {
          a[t] = (in && tc) ? tc[o0 + (int64_t)t * N] : 0.f;
          b[t] = (in && tp) ? tp[o0 + (int64_t)t * N] : 0.f;
        }
      }
#pragma unroll
      for (int t = 0; t < RB; ++t) {
        const int i = i0 + t;
        if (i < N) {
          float acc = 0.f;
          if (hr) acc += hc[t];
          if (hl) acc -= hp[t];
          if (i < N - 1) acc += v[t + 1];
          if (i > 0) acc -= v[t];
          if (TEMPORAL) {
            if (tc) acc += a[t];
            if (tp) acc -= b[t];
          }
          out[o0 + (int64_t)t * N] = acc;
          if (SUMSQ) ss += (double)acc * acc;
        }
      }
    }
  }
  if (SUMSQ) {
    ss = block_sum<NT>(ss, red);
    if (threadIdx.x == 0) partials[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = ss;
  }
}

// Fused forms for the re-weighted (MM) solvers, nothing of length 2N(N-1) is written or read twice:
//   k_tv_weights:  w = ((L2 x)^2 + eps^2)^(q/2-1)                      (MMGKS.py:60,93)      4n read, 8n written
//   k_tv_grad:     out = r_in + lam * L2^T (w .* (L2 x))               (MMGKS.py:116-118)    x, w, r_in read: 16n, 4n written
// (W = false: unit weights, out = r_in + lam * L2^T L2 x, the GKS residual term GKS.py:81-84.)
__global__ __launch_bounds__(NT) void k_tv_weights(const float* __restrict__ x, int N, float eps2, float e, int special,
                                                   float* __restrict__ w) {
  // blockIdx.z = frame (space-time operator: frame-major images, per-frame spatial rows; the temporal rows: k_tvt_weights)
  const int64_t npix = (int64_t)N * N, ps = 2 * (int64_t)N * (N - 1);
  double unused = 0.0;
  d2_fwd_body<false, true>(x + blockIdx.z * npix, w + blockIdx.z * ps, N, eps2, e, special, unused);
}

// temporal rows of the space-time operator: w_t = ((x_t - x_{t+1})^2 + eps^2)^e (blockIdx.y = row).  Rows t < nt - 1 are inside the
// rank's frames; with a time-sharded vector row nt - 1 (has_next) takes x_{t+1} from the next rank's first frame, and ONE more row —
// the weight of the row the PREVIOUS rank owns, (its last frame) - x_0, which this rank's k_tv_grad needs for frame 0 — is
// recomputed here from the halo: the same expression on the same two floats, the same bits as on the rank that owns it.
__global__ __launch_bounds__(NT) void k_tvt_weights(const float* __restrict__ x, int64_t npix, int nt, float eps2, float e, int special,
                                                    float* __restrict__ wt, const float* __restrict__ xprev,
                                                    const float* __restrict__ xnext) {
  const int r = blockIdx.y;
  const float* a;
  const float* b;
  if (r < nt - 1) { a = x + (int64_t)r * npix; b = a + npix; }
  else if (r == nt - 1 && xnext) { a = x + (int64_t)r * npix; b = xnext; }
  else { a = xprev; b = x; }                                       // the previous rank's boundary row
  float* o = wt + (int64_t)r * npix;
  for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < npix; idx += (int64_t)gridDim.x * NT)
    o[idx] = mm_w(a[idx] - b[idx], eps2, e, special);
}

// nt > 1 or halos: the space-time operator.  blockIdx.z = frame; w = [nt x 2N(N-1) spatial | temporal rows x N^2] as the rows of L;
// the temporal part adds wt_t (x_t - x_{t+1}) - wt_{t-1} (x_{t-1} - x_t) to frame t.  A time-sharded vector brings its neighbours'
// boundary frames (xprev: the previous rank's last frame, xnext: the next rank's first; trk_tv_halo): frame 0 / nt - 1 then have
// their temporal neighbours like any inner frame, and the rank forms ITS pixels of L^T (w .* L x) completely — the same kernel as on
// one rank, no exchange of rows of L x.  Temporal weights: rows 0 .. nt-2 (+ row nt-1 with xnext), then the previous rank's
// boundary row (with xprev), as k_tvt_weights lays them out.
// DOT: also the block partial of <out, dotv> (GKS: r . L^T L r for the Gram row of the next basis vector, GKS.py:92-96 through the
// Gram form — no pass over the two vectors of its own); every thread then stays to the end (the block sum's barriers).
template <bool W, bool RIN, bool DOT = false>
__global__ __launch_bounds__(NT) void k_tv_grad(const float* __restrict__ x, const float* __restrict__ w,
                                                const float* __restrict__ rin, float lam, float* __restrict__ out, int N, int nt,
                                                const float* __restrict__ dotv = nullptr, double* __restrict__ dot_part = nullptr,
                                                const float* __restrict__ xprev = nullptr, const float* __restrict__ xnext = nullptr,
                                                int xsq = 0) {
#pragma clang fp contract(off)       // products and sums as written (HIP's __fmul_rn is a plain `*`): the same bits in every instantiation
  __shared__ double dred[DOT ? NT / 64 : 1];
  double dacc = 0.0, xacc = 0.0;       // xsq (DOT only): also the block partial of <x, x> over the rank's own pixels, behind the dot's
  const int f = blockIdx.z;
  const int64_t npix = (int64_t)N * N, ps = 2 * (int64_t)N * (N - 1);
  const float* __restrict__ xf = x + f * npix;
  const float* __restrict__ wh = W ? w + f * ps : nullptr;
  const float* __restrict__ wv = W ? wh + (int64_t)N * (N - 1) : nullptr;
  const float* __restrict__ wt = (W && (nt > 1 || xprev || xnext)) ? w + nt * ps : nullptr;      // temporal weights, row t at wt + t npix
  rin = RIN ? rin + f * npix : rin;
  out += f * npix;
  if (DOT) dotv += f * npix;
  const int jraw = blockIdx.x * NT + threadIdx.x;
  const bool live = jraw < N;
  if (!DOT && !live) return;
  const int j = live ? jraw : N - 1;                             // (DOT: a thread beyond the image works on the last column, stores nothing)
  const bool hr = j < N - 1, hl = j > 0;
  // the frames after / before this one: inside the rank's block, or the neighbour rank's boundary frame
  const float* __restrict__ xnf = (f < nt - 1) ? xf + npix : xnext;
  const float* __restrict__ xpf = (f > 0) ? xf - npix : xprev;
  const bool tnext = xnf != nullptr, tprev = xpf != nullptr;
  // weight rows: row f couples f with f + 1; the row before frame 0 is the previous rank's, kept behind the rank's own rows
  const int ntemp = nt - 1 + (xnext ? 1 : 0);
  const int64_t wrow_prev = (f > 0) ? (int64_t)(f - 1) * npix : (int64_t)ntemp * npix;
  for (int i0 = blockIdx.y * RB; i0 < N; i0 += gridDim.y * RB) {
    const int64_t o0 = (int64_t)i0 * N + j;
    const int64_t h0 = (int64_t)i0 * (N - 1) + j;
    float xc[RB + 2], xl[RB], xr[RB], wc[RB], wp[RB], v[RB + 1], rr[RB], xn[RB], xp[RB], wn[RB], wq[RB];
#pragma unroll
    for (int t = 0; t < RB + 2; ++t) {   // xc[t] = x[i0 + t - 1][j]
      const int i = i0 + t - 1;
      xc[t] = (i >= 0 && i < N) ? xf[o0 + (int64_t)(t - 1) * N] : 0.f;
    }
#pragma unroll
    for (int t = 0; t <= RB; ++t) {      // v[t] = wv[i0 + t - 1][j]
      const int i = i0 + t - 1;
      v[t] = W ? ((i >= 0 && i < N - 1) ? wv[o0 + (int64_t)(t - 1) * N] : 0.f) : 1.f;
    }
#pragma unroll
    for (int t = 0; t < RB; ++t) {
      const bool in = i0 + t < N;
      const int64_t o = o0 + (int64_t)t * N, hb = h0 + (int64_t)t * (N - 1);
      xr[t] = (in && hr) ? xf[o + 1] : 0.f;
      xl[t] = (in && hl) ? xf[o - 1] : 0.f;
      wc[t] = W ? ((in && hr) ? wh[hb] : 0.f) : 1.f;
      wp[t] = W ? ((in && hl) ? wh[hb - 1] : 0.f) : 1.f;
      rr[t] = (RIN && in) ? rin[o] : 0.f;
      xn[t] = (in && tnext) ? xnf[o] : 0.f;
      xp[t] = (in && tprev) ? xpf[o] : 0.f;
      wn[t] = W ? ((in && tnext) ? wt[(int64_t)f * npix + o] : 0.f) : 1.f;
      wq[t] = W ? ((in && tprev) ? wt[wrow_prev + o] : 0.f) : 1.f;
    }
#pragma unroll
    for (int t = 0; t < RB; ++t) {
      const int i = i0 + t;
      if (i < N) {
        const float c = xc[t + 1];
        float acc = 0.f;
        if (hr) acc += __fmul_rn(wc[t], c - xr[t]);
        if (hl) acc -= __fmul_rn(wp[t], xl[t] - c);
        if (i < N - 1) acc += __fmul_rn(v[t + 1], c - xc[t + 2]);
        if (i > 0) acc -= __fmul_rn(v[t], xc[t] - c);
        if (tnext) acc += __fmul_rn(wn[t], c - xn[t]);             // same order as k_d2_adj<TEMPORAL>: + T_t - T_{t-1}
        if (tprev) acc -= __fmul_rn(wq[t], xp[t] - c);
        const float ov = RIN ? __fadd_rn(rr[t], __fmul_rn(lam, acc)) : __fmul_rn(lam, acc);   // (no contraction: the same bits with and without DOT)
        if (!DOT || live) out[o0 + (int64_t)t * N] = ov;
        if (DOT && live) dacc += (double)ov * (double)dotv[o0 + (int64_t)t * N];
        if (DOT && live && xsq) xacc += (double)c * (double)c;
      }
    }
  }
  if (DOT) {
    const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    dacc = block_sum<NT>(dacc, dred);
    if (!xsq) {
      if (threadIdx.x == 0) dot_part[blk] = dacc;
    } else {
      __syncthreads();
      xacc = block_sum<NT>(xacc, dred);
      if (threadIdx.x == 0) {
        dot_part[2 * blk] = dacc;
        dot_part[2 * blk + 1] = xacc;
      }
    }
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
  const Grid2 g2 = grid2(N, batch, sumsq != nullptr);
  const int gx = g2.per_frame;
  double* part = nullptr;
  if (sumsq)
    if (int rc = scratch_doubles(s, (size_t)gx * batch, &part)) return rc;
  const dim3 grid = g2.g;
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
  const float* xh_prev;    // fused forms (trk_tv_halo): the previous rank's LAST frame of the vector the next fused call takes
  const float* xh_next;    // ... and the next rank's FIRST frame
};

int st_apply(trk_op* op, int tr, const float* x, int64_t ldx, float* y, int64_t ldy, int batch, double* sumsq,
             hipStream_t s) {
  auto* im = static_cast<STImpl*>(op->impl);
  const int N = im->N, nt = im->nt;
  const int64_t npix = (int64_t)N * N, ps = 2 * (int64_t)N * (N - 1);
  const int ntemp = nt - 1 + (im->has_next ? 1 : 0);
  if (im->has_next && !tr && !im->halo_next) return fail(TRK_EINVAL, "spacetime forward: halo of the next rank not set");
  if (im->has_prev && tr && !im->halo_prev) return fail(TRK_EINVAL, "spacetime transpose: halo of the previous rank not set");
  const int gt = grid_for(npix);        // temporal rows: flat streaming kernel
  const Grid2 gs = grid2(N, nt, sumsq != nullptr);  // spatial rows: column batches, blockIdx.z = frame
  const int gsp = gs.per_frame;
  double* part = nullptr;
  // block partials per vector: forward = spatial (gsp*nt) + temporal (gt*ntemp) ; transpose = gsp*nt
  const int per_vec = tr ? gsp * nt : gsp * nt + gt * (ntemp > 0 ? ntemp : 0);
  if (sumsq)
    if (int rc = scratch_doubles(s, (size_t)per_vec * batch, &part)) return rc;
  for (int b = 0; b < batch; ++b) {
    const float* xb = x + (int64_t)b * ldx;
    float* yb = y + (int64_t)b * ldy;
    double* pb = part ? part + (size_t)b * per_vec : nullptr;
    if (!tr) {
      if (sumsq) hipLaunchKernelGGL((k_d2_fwd<true>), gs.g, dim3(NT), 0, s, xb, npix, yb, ps, N, pb);
      else hipLaunchKernelGGL((k_d2_fwd<false>), gs.g, dim3(NT), 0, s, xb, npix, yb, ps, N, pb);
      if (ntemp > 0) {
        dim3 g2(gt, ntemp);
        float* T = yb + (int64_t)nt * ps;
        if (sumsq) hipLaunchKernelGGL((k_time_fwd<true>), g2, dim3(NT), 0, s, xb, T, npix, nt, im->halo_next, pb + (size_t)gsp * nt);
        else hipLaunchKernelGGL((k_time_fwd<false>), g2, dim3(NT), 0, s, xb, T, npix, nt, im->halo_next, pb);
      }
    } else {
      const float* T = xb + (int64_t)nt * ps;
      if (sumsq) hipLaunchKernelGGL((k_d2_adj<true, true>), gs.g, dim3(NT), 0, s, xb, ps, yb, npix, N, T, npix, T, im->has_prev, im->has_next, im->halo_prev, pb);
      else hipLaunchKernelGGL((k_d2_adj<false, true>), gs.g, dim3(NT), 0, s, xb, ps, yb, npix, N, T, npix, T, im->has_prev, im->has_next, im->halo_prev, pb);
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
  auto* im = new STImpl{N, nt_local, has_next ? 1 : 0, has_prev ? 1 : 0, nullptr, nullptr, nullptr, nullptr};
  const int64_t npix = (int64_t)N * N, ps = 2 * (int64_t)N * (N - 1);
  const int ntemp = nt_local - 1 + (has_next ? 1 : 0);
  *out = new trk_op{4, (int64_t)nt_local * ps + (int64_t)ntemp * npix, (int64_t)nt_local * npix, im, st_apply, st_destroy, nullptr, 0};
  return TRK_OK;
}

// N and the frame count of a first-difference regulariser the fused TV forms serve: the 2-D operator or the space-time operator.
// A time-sharded space-time handle needs the neighbours' boundary frames of the operand (trk_tv_halo, consumed by this call).
struct TvGeo {
  int N, nt;
  const float *xprev, *xnext;
};
static int tv_geometry(trk_op* L, const char* who, TvGeo* g) {
  g->xprev = g->xnext = nullptr;
  if (L->kind == 3) {
    g->N = static_cast<D2Impl*>(L->impl)->N;
    g->nt = 1;
    return TRK_OK;
  }
  if (L->kind == 4) {
    auto* im = static_cast<STImpl*>(L->impl);
    if ((im->has_prev && !im->xh_prev) || (im->has_next && !im->xh_next))
      return fail(TRK_EINVAL, "%s: the time axis is sharded over ranks: give the operand's boundary frames of the neighbour ranks with trk_tv_halo first", who);
    g->N = im->N;
    g->nt = im->nt;
    g->xprev = im->has_prev ? im->xh_prev : nullptr;
    g->xnext = im->has_next ? im->xh_next : nullptr;
    im->xh_prev = im->xh_next = nullptr;             // they belong to ONE operand
    return TRK_OK;
  }
  return fail(TRK_EINVAL, "%s: L must come from trk_deriv2d_create or trk_spacetime_create", who);
}

int trk_tv_halo(trk_op* L, const float* x_prev_last, const float* x_next_first) {
  TRK_REQUIRE(L && L->kind == 4, "trk_tv_halo: not a space-time operator");
  auto* im = static_cast<STImpl*>(L->impl);
  TRK_REQUIRE((!im->has_prev || x_prev_last) && (!im->has_next || x_next_first), "trk_tv_halo: NULL frame for an existing neighbour");
  im->xh_prev = x_prev_last;
  im->xh_next = x_next_first;
  return TRK_OK;
}

int trk_tv_weights(trk_op* L, const float* x, double eps, double q, float* w, trk_stream st) {
  TRK_REQUIRE(L, "trk_tv_weights: NULL operator");
  TvGeo g;
  const int rcg = tv_geometry(L, "trk_tv_weights", &g);      // first: the frames trk_tv_halo parked are consumed on EVERY exit path
  TRK_REQUIRE(x && w, "trk_tv_weights: NULL argument");
  if (rcg) return rcg;
  const int N = g.N, nt = g.nt;
  const float e = (float)(q / 2.0 - 1.0), eps2 = (float)(eps * eps);
  const int special = (q == 2.0) ? 1 : (q == 1.0) ? 2 : 0;
  hipLaunchKernelGGL(k_tv_weights, grid2(N, nt, false).g, dim3(NT), 0, (hipStream_t)st, x, N, eps2, e, special, w);
  const int rows = nt - 1 + (g.xnext ? 1 : 0) + (g.xprev ? 1 : 0);
  if (rows > 0) {
    const int64_t npix = (int64_t)N * N;
    hipLaunchKernelGGL(k_tvt_weights, dim3(grid_for(npix), rows), dim3(NT), 0, (hipStream_t)st, x, npix, nt, eps2, e, special,
                       w + (int64_t)nt * 2 * N * (N - 1), g.xprev, g.xnext);
  }
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_tv_grad(trk_op* L, const float* x, const float* w, const float* r_in, double lam, float* out, trk_stream st) {
  TRK_REQUIRE(L, "trk_tv_grad: NULL operator");
  TvGeo gg;
  const int rcg = tv_geometry(L, "trk_tv_grad", &gg);        // first: see trk_tv_weights
  TRK_REQUIRE(x && out, "trk_tv_grad: NULL argument");
  TRK_REQUIRE(out != x && out != r_in, "trk_tv_grad: out must not alias x or r_in");
  if (rcg) return rcg;
  const int N = gg.N, nt = gg.nt;
  const dim3 g = grid2(N, nt, false).g;
  hipStream_t s = (hipStream_t)st;
#define TG(W, R) hipLaunchKernelGGL((k_tv_grad<W, R>), g, dim3(NT), 0, s, x, w, r_in, (float)lam, out, N, nt, (const float*)nullptr, (double*)nullptr, gg.xprev, gg.xnext)
  if (w) { if (r_in) TG(true, true); else TG(true, false); }
  else   { if (r_in) TG(false, true); else TG(false, false); }
#undef TG
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_tv_grad_dot(trk_op* L, const float* x, const float* w, const float* r_in, double lam, float* out, const float* dotv,
                    double* dot_out, trk_stream st) {
  TRK_REQUIRE(L, "trk_tv_grad_dot: NULL operator");
  TvGeo gg;
  const int rcg = tv_geometry(L, "trk_tv_grad_dot", &gg);    // first: see trk_tv_weights
  TRK_REQUIRE(x && out && dotv && dot_out, "trk_tv_grad_dot: NULL argument");
  TRK_REQUIRE(out != x && out != r_in && out != dotv, "trk_tv_grad_dot: out must not alias x, r_in or dotv");
  if (rcg) return rcg;
  const int N = gg.N, nt = gg.nt;
  const Grid2 g2 = grid2(N, nt, true);
  const int nblk = g2.per_frame * nt;
  hipStream_t s = (hipStream_t)st;
  double* part = nullptr;
  if (int rc = scratch_doubles(s, (size_t)nblk, &part)) return rc;
#define TGD(W, R) hipLaunchKernelGGL((k_tv_grad<W, R, true>), g2.g, dim3(NT), 0, s, x, w, r_in, (float)lam, out, N, nt, dotv, part, gg.xprev, gg.xnext)
  if (w) { if (r_in) TGD(true, true); else TGD(true, false); }
  else   { if (r_in) TGD(false, true); else TGD(false, false); }
#undef TGD
  TRK_LAUNCH_CHECK();
  return finalize_sums(part, nblk, 1, 1, dot_out, s);
}

int trk_tv_grad_dot_xsq(trk_op* L, const float* x, const float* w, const float* r_in, double lam, float* out, const float* dotv,
                        double* dot_out, double* xsq_out, trk_stream st) {
  TRK_REQUIRE(L, "trk_tv_grad_dot_xsq: NULL operator");
  TvGeo gg;
  const int rcg = tv_geometry(L, "trk_tv_grad_dot_xsq", &gg);    // first: see trk_tv_weights
  TRK_REQUIRE(x && out && dotv && dot_out && xsq_out, "trk_tv_grad_dot_xsq: NULL argument");
  TRK_REQUIRE(out != x && out != r_in && out != dotv, "trk_tv_grad_dot_xsq: out must not alias x, r_in or dotv");
  if (rcg) return rcg;
  const int N = gg.N, nt = gg.nt;
  const Grid2 g2 = grid2(N, nt, true);
  const int nblk = g2.per_frame * nt;
  hipStream_t s = (hipStream_t)st;
  double* part = nullptr;
  if (int rc = scratch_doubles(s, 2 * (size_t)nblk, &part)) return rc;
#define TGD(W, R) hipLaunchKernelGGL((k_tv_grad<W, R, true>), g2.g, dim3(NT), 0, s, x, w, r_in, (float)lam, out, N, nt, dotv, part, gg.xprev, gg.xnext, 1)
  if (w) { if (r_in) TGD(true, true); else TGD(true, false); }
  else   { if (r_in) TGD(false, true); else TGD(false, false); }
#undef TGD
  TRK_LAUNCH_CHECK();
  return finalize_sums_split(part, nblk, 2, 2, dot_out, 1, xsq_out, s);
}

int trk_spacetime_set_halo(trk_op* op, const float* x_next, const float* y_prev) {
  TRK_REQUIRE(op && op->kind == 4, "trk_spacetime_set_halo: not a space-time operator");
  auto* im = static_cast<STImpl*>(op->impl);
  im->halo_next = x_next;
  im->halo_prev = y_prev;
  return TRK_OK;
}

}  // extern "C"
