// spmv.hip — general sparse operator (CSR SpMV), for operators given as matrices: the reference builds its
// regularisers as scipy.sparse matrices (trips/utilities/operators.py:24-45 derivative operators, :50-113 framelet
// analysis operators) and the real-data dynamic problems ship a precomputed sparse forward matrix sliced into per-frame
// blocks (trips/utilities/io.py:132-135,197-229).  y = A x uses the CSR of A; y = A^T x uses the CSR of A^T (built once
// by the host) — both directions are gathers (no atomics).
//   * short rows (derivative / framelet rows: 2..9 non-zeros): one thread per row;
//   * long rows  (tomography rays: ~N non-zeros): one wave per row, lanes stride the row's non-zeros (coalesced value /
//     index loads), wave64 shuffle reduction.
// fp32 values and vectors, fp64 row accumulation.  HBM-bound: 8 bytes per non-zero (value + column index) + the gathers of x.
#include "trk_internal.h"

using namespace trk;

namespace {

constexpr int NT = 256;

struct Csr {
  int64_t nrows, ncols, nnz;
  int64_t* indptr;
  int* indices;
  float* vals;
  bool long_rows;
};

struct SpImpl {
  Csr a, at;
};

template <bool SUMSQ>
__global__ __launch_bounds__(NT) void k_csr_thread(int64_t nrows, const int64_t* __restrict__ indptr,
                                                   const int* __restrict__ indices, const float* __restrict__ vals,
                                                   const float* __restrict__ x, float* __restrict__ y,
                                                   double* __restrict__ partials) {
  __shared__ double red[NT / 64];
  double ss = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * NT + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * NT) {
    const int64_t p0 = indptr[r], p1 = indptr[r + 1];
    double acc = 0.0;
    for (int64_t p = p0; p < p1; ++p) acc = fma((double)vals[p], (double)x[indices[p]], acc);
    const float o = (float)acc;
    y[r] = o;
    if (SUMSQ) ss += (double)o * o;
  }
  if (SUMSQ) {
    ss = block_sum<NT>(ss, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = ss;
  }
}

template <bool SUMSQ>
__global__ __launch_bounds__(NT) void k_csr_wave(int64_t nrows, const int64_t* __restrict__ indptr,
                                                 const int* __restrict__ indices, const float* __restrict__ vals,
                                                 const float* __restrict__ x, float* __restrict__ y,
                                                 double* __restrict__ partials) {
  __shared__ double red[NT / 64];
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * NT + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * NT) >> 6;
  double ss = 0.0;
  for (int64_t r = wave0; r < nrows; r += nwaves) {
    const int64_t p0 = indptr[r], p1 = indptr[r + 1];
    double acc = 0.0;
    for (int64_t p = p0 + lane; p < p1; p += 64) acc = fma((double)vals[p], (double)x[indices[p]], acc);
    acc = wave_sum(acc);
    if (lane == 0) {
      const float o = (float)acc;
      y[r] = o;
      if (SUMSQ) ss += (double)o * o;
    }
  }
  if (SUMSQ) {
    ss = block_sum<NT>(ss, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = ss;
  }
}

int sp_apply(trk_op* op, int tr, const float* x, int64_t ldx, float* y, int64_t ldy, int batch, double* sumsq,
             hipStream_t s) {
  auto* im = static_cast<SpImpl*>(op->impl);
  const Csr& M = tr ? im->at : im->a;
  int64_t want = M.long_rows ? (M.nrows + 3) / 4 : (M.nrows + NT - 1) / NT;
  if (want > kMaxPartialBlocks) want = kMaxPartialBlocks;
  const int grid = (int)(want < 1 ? 1 : want);
  double* part = nullptr;
  if (sumsq)
    if (int rc = scratch_doubles(s, (size_t)grid * batch, &part)) return rc;
  TimerScope tm(op->timer, op->timer_which, tr, s);
  for (int b = 0; b < batch; ++b) {
    const float* xb = x + (int64_t)b * ldx;
    float* yb = y + (int64_t)b * ldy;
    double* pb = part ? part + (size_t)b * grid : nullptr;
#define SP(K, SS) hipLaunchKernelGGL((K<SS>), dim3(grid), dim3(NT), 0, s, M.nrows, M.indptr, M.indices, M.vals, xb, yb, pb)
    if (M.long_rows) { if (sumsq) SP(k_csr_wave, true); else SP(k_csr_wave, false); }
    else             { if (sumsq) SP(k_csr_thread, true); else SP(k_csr_thread, false); }
#undef SP
  }
  tm.stop();
  TRK_LAUNCH_CHECK();
  if (sumsq) return finalize_sums(part, grid * batch, 1, 1, sumsq, s);
  return TRK_OK;
}

void csr_free(Csr& c) {
  if (c.indptr) (void)hipFree(c.indptr);
  if (c.indices) (void)hipFree(c.indices);
  if (c.vals) (void)hipFree(c.vals);
}

void sp_destroy(trk_op* op) {
  auto* im = static_cast<SpImpl*>(op->impl);
  csr_free(im->a);
  csr_free(im->at);
  delete im;
}

int csr_upload(Csr& c, int64_t nrows, int64_t ncols, int64_t nnz, const int64_t* indptr, const int* indices,
               const float* vals) {
  c = Csr{nrows, ncols, nnz, nullptr, nullptr, nullptr, nnz > 16 * nrows};
  TRK_HIP(hipMalloc(&c.indptr, sizeof(int64_t) * (size_t)(nrows + 1)));
  TRK_HIP(hipMalloc(&c.indices, sizeof(int) * (size_t)(nnz > 0 ? nnz : 1)));
  TRK_HIP(hipMalloc(&c.vals, sizeof(float) * (size_t)(nnz > 0 ? nnz : 1)));
  TRK_HIP(hipMemcpy(c.indptr, indptr, sizeof(int64_t) * (size_t)(nrows + 1), hipMemcpyHostToDevice));
  if (nnz > 0) {
    TRK_HIP(hipMemcpy(c.indices, indices, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
    TRK_HIP(hipMemcpy(c.vals, vals, sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice));
  }
  return TRK_OK;
}

}  // namespace

extern "C" int trk_csr_create(int64_t nrows, int64_t ncols, int64_t nnz, const int64_t* indptr_host, const int* indices_host,
                              const float* values_host, const int64_t* t_indptr_host, const int* t_indices_host,
                              const float* t_values_host, trk_op** out) {
  TRK_REQUIRE(out && indptr_host && t_indptr_host, "trk_csr_create: NULL argument");
  TRK_REQUIRE(nrows >= 1 && ncols >= 1 && nnz >= 0 && ncols < ((int64_t)1 << 31) && nrows < ((int64_t)1 << 31),
              "trk_csr_create: bad sizes");
  TRK_REQUIRE(nnz == 0 || (indices_host && values_host && t_indices_host && t_values_host), "trk_csr_create: NULL arrays");
  TRK_REQUIRE(indptr_host[nrows] == nnz && t_indptr_host[ncols] == nnz, "trk_csr_create: indptr does not end at nnz");
  auto* im = new SpImpl{};
  int rc = csr_upload(im->a, nrows, ncols, nnz, indptr_host, indices_host, values_host);
  if (!rc) rc = csr_upload(im->at, ncols, nrows, nnz, t_indptr_host, t_indices_host, t_values_host);
  if (rc) {
    trk_op tmp{6, 0, 0, im, nullptr, nullptr, nullptr, 0};
    sp_destroy(&tmp);
    return rc;
  }
  *out = new trk_op{6, nrows, ncols, im, sp_apply, sp_destroy, nullptr, 0};
  return TRK_OK;
}
