// cgls_loop.hip — whole stretches of CGLS iterations enqueued by ONE library call.
//
// With tol = 0 the recurrence of trips/solvers/CGLS.py:56-80 never needs a value on the host, so nothing but call
// overhead separates consecutive kernels.  Driving the 3-6 launches of an iteration from Python costs ~10 us of
// interpreter + ctypes time per launch — more than the kernels themselves for images up to ~1024^2 (a 512^2 blur runs in
// 3 us).  These two entry points run the same launch sequence as trips_py_amd.solvers.CGLS.CGLSRun.step /
// CGLSRunFused.step in a C loop (same kernels, same scalar layout, bit-identical results), leaving one call per solve.
#include "trk_internal.h"

using namespace trk;

extern "C" {

// Which pair of kernels carries the updates between the operator applies of the raw-partials iteration:
//   0: [x' = x + a p ; r -= a w] after A p, [p = t + b p] after A^T r           (36n bytes; trk_cgls_update_xr_src, trk_cgls_p_update)
//   1: [r -= a w] after A p, [x' = x + a p ; p = t + b p] after A^T r            (32n bytes: p is read once; trk_cgls_r_update,
//                                                                                 trk_cgls_xp_update)
// Measured (iterations/s, 0 vs 1): 512^2 39.1 k vs 38.3 k, 1536^2 23.7 k vs 23.4 k | 3072^2 10.6 k vs 11.1 k, 3584^2 8.30 k vs
// 8.72 k, 4096^2 6.85 k vs 7.26 k, 4608^2 5.48 k vs 5.70 k, 5120^2 4.45 k vs 4.36 k, 6144^2 2.82 k vs 2.74 k, 8192^2 1.53 k vs
// 1.64 k: the saved pass pays from ~8 M unknowns on (small images are launch-bound and the five-stream kernel is the
// slower one per byte).  TRK_CGLS_XP=0/1 overrides (tuning).
int trk_cgls_update_grouping(int64_t n) {
  static const int env = getenv("TRK_CGLS_XP") ? atoi(getenv("TRK_CGLS_XP")) : -1;
  if (env >= 0) return env != 0;
  return n >= ((int64_t)8 << 20) ? 1 : 0;
}

int trk_cgls_iterate(trk_op* A, int k_first, int n_iters, float* p, float* r, float* t, float* w, float* X, int64_t x_ld,
                     int keep_history, const float* x_prev, const float* x_true, double* S, double* NP,
                     int np_capacity_blocks, int* n_np_inout, double* PG, double* PD, int pcap, int grouping,
                     trk_stream stream) {
  TRK_REQUIRE(A && p && r && t && w && X && x_prev && S && NP && n_np_inout, "trk_cgls_iterate: NULL argument");
  TRK_REQUIRE(k_first >= 1 && n_iters >= 0, "trk_cgls_iterate: need k_first >= 1, n_iters >= 0");
  const int64_t m = A->rows, n = A->cols;
  int n_np = *n_np_inout;
  // raw-partials form: the operator leaves ||w||^2, ||t||^2 as block partials and the consumers add them up —
  // four launches per iteration instead of six (no reduction-finalize launches)
  const bool raw = PG && PD && pcap > 0 && A->apply_fused;
  const bool xp_form = (grouping < 0 ? trk_cgls_update_grouping(n) : grouping) == 1;
  for (int k = k_first; k < k_first + n_iters; ++k) {
    double* row = S + 5 * (int64_t)k;                       // [delta, gamma, ||x||^2, ||dx||^2, ||x-xt||^2]
    double *delta = row, *gamma = row + 1;
    const double* gamma_old = (k == 1) ? S : row - 4;
    float* x_new = X + (int64_t)(keep_history ? (k - 1) : ((k - 1) & 1)) * x_ld;
    int rc;
    if (raw && xp_form) {
      int n_d = 0, n_g = 0;
      rc = trk_op_apply_fused(A, 0, p, nullptr, 0.0, nullptr, 0, nullptr, 0, nullptr, w, PD, pcap, &n_d, stream);
      if (rc) return rc;
      rc = trk_cgls_r_update(m, gamma_old, PD, n_d, r, w, delta, stream);
      if (rc) return rc;
      rc = trk_op_apply_fused(A, 1, r, nullptr, 0.0, nullptr, 0, nullptr, 0, nullptr, t, PG, pcap, &n_g, stream);
      if (rc) return rc;
      rc = trk_cgls_xp_update(n, gamma_old, delta, PG, n_g, x_prev, p, t, x_new, x_true, gamma,
                              NP + 3 * (int64_t)n_np * (k - 1), np_capacity_blocks, &n_np, stream);
      if (rc) return rc;
      x_prev = x_new;
      continue;
    }
    if (raw) {
      int n_d = 0, n_g = 0;
      rc = trk_op_apply_fused(A, 0, p, nullptr, 0.0, nullptr, 0, nullptr, 0, nullptr, w, PD, pcap, &n_d, stream);
      if (rc) return rc;
      rc = trk_cgls_update_xr_src(n, m, gamma_old, 1, PD, n_d, x_prev, p, x_new, r, w, x_true, delta,
                                  NP + 3 * (int64_t)n_np * (k - 1), np_capacity_blocks, &n_np, stream);
      if (rc) return rc;
      rc = trk_op_apply_fused(A, 1, r, nullptr, 0.0, nullptr, 0, nullptr, 0, nullptr, t, PG, pcap, &n_g, stream);
      if (rc) return rc;
      rc = trk_cgls_p_update(n, t, p, PG, n_g, gamma_old, gamma, stream);
      if (rc) return rc;
      x_prev = x_new;
      continue;
    }
    rc = trk_op_apply(A, 0, p, 0, w, 0, 1, delta, stream);                                      // w = A p, ||w||^2   (:60-61)
    if (rc) return rc;
    rc = trk_cgls_update_xr_deferred(n, m, gamma_old, delta, x_prev, p, x_new, r, w, x_true,      // x, r updates   (:64-67)
                                     NP + 3 * (int64_t)n_np * (k - 1), np_capacity_blocks, &n_np, stream);
    if (rc) return rc;
    rc = trk_op_apply(A, 1, r, 0, t, 0, 1, gamma, stream);                                      // t = A^T r, ||t||^2 (:68-70)
    if (rc) return rc;
    rc = trk_axpby(n, 1.0, nullptr, nullptr, 0, t, 1.0, gamma, gamma_old, 0, p, p, nullptr, stream);   // p = t + (g/g_old) p (:72)
    if (rc) return rc;
    x_prev = x_new;
  }
  *n_np_inout = n_np;
  return TRK_OK;
}

int trk_cgls_iterate_fused(trk_op* A, int k_first, int n_iters, float* P, int64_t p_ld, float* R, int64_t r_ld, float* t,
                           float* w, float* X, int64_t x_ld, int keep_history, const float* x_prev, const float* x_true,
                           double* S, double* PG, double* PD, int pcap, double* NP, int np_capacity_blocks,
                           int* n_g_inout, int* n_np_inout, trk_stream stream) {
  TRK_REQUIRE(A && P && R && t && w && X && x_prev && S && PG && PD && NP && n_g_inout && n_np_inout,
              "trk_cgls_iterate_fused: NULL argument");
  TRK_REQUIRE(k_first >= 1 && n_iters >= 0, "trk_cgls_iterate_fused: need k_first >= 1, n_iters >= 0");
  const int64_t n = A->cols;
  int n_g = *n_g_inout, n_np = *n_np_inout, n_d = 0;
  for (int k = k_first; k < k_first + n_iters; ++k) {
    const int64_t b = 5 * (int64_t)k;
    float *p_old = P + (int64_t)((k - 1) & 1) * p_ld, *p_new = P + (int64_t)(k & 1) * p_ld;
    float *r_old = R + (int64_t)((k - 1) & 1) * r_ld, *r_new = R + (int64_t)(k & 1) * r_ld;
    const double* gprev = (k <= 2) ? S : S + 5 * (int64_t)(k - 2) + 1;      // gamma_{k-2}, published by K2 of iteration k-1
    // K1: p_k = t + (gamma_{k-1}/gamma_{k-2}) p_{k-1} ; w = A p_k
    int rc = trk_op_apply_fused(A, 0, t, p_old, k == 1 ? 0.0 : 1.0, PG, n_g, gprev, 1, p_new, w, PD, pcap, &n_d, stream);
    if (rc) return rc;
    // K2: x_k = x_{k-1} + (gamma_{k-1}/delta_k) p_k ; publishes delta_k -> S[5k], gamma_{k-1} -> S[5(k-1)+1] (S[0] for k = 1)
    float* x_new = X + (int64_t)(keep_history ? (k - 1) : ((k - 1) & 1)) * x_ld;
    double* gpub = (k == 1) ? S : S + b - 4;
    rc = trk_cgls_x_update(n, PG, n_g, PD, n_d, x_prev, p_new, x_new, x_true, S + b, gpub,
                           NP + 3 * (int64_t)n_np * (k - 1), np_capacity_blocks, &n_np, stream);
    if (rc) return rc;
    // K3: r_k = r_{k-1} - (gamma_{k-1}/delta_k) w ; t = A^T r_k ; ||t||^2 partials = gamma_k
    rc = trk_op_apply_fused(A, 1, r_old, w, -1.0, gpub, 1, S + b, 1, r_new, t, PG, pcap, &n_g, stream);
    if (rc) return rc;
    x_prev = x_new;
  }
  *n_g_inout = n_g;
  *n_np_inout = n_np;
  return TRK_OK;
}

}  // extern "C"
