// projected.hip — the projected (k-sized) Tikhonov problem of the Golub-Kahan hybrid solvers, solved ON the device so
// that an iteration with a fixed regularisation parameter never visits the host.
//
// Replaces   y = np.linalg.lstsq(vstack((B_k, sqrt(lam) I)), vstack((beta0 e1, 0)))   trips/solvers/Hybrid_LSQR.py:104
// (and GK_Tikhonov.py:60) for the lower-bidiagonal B_k of Golub-Kahan: the stacked matrix is reduced to an upper
// bidiagonal R by 2k Givens rotations (the damped-LSQR elimination of Paige & Saunders 1982, §2 of "LSQR: an algorithm
// for sparse linear equations and sparse least squares"), then R y = phi is back-substituted.  O(k), backward stable,
// float64 throughout; one lane does the (inherently sequential) recurrence: ~4 us at k = 100.
#include "trk_internal.h"

using namespace trk;

namespace {

constexpr int BIDIAG_MAX_K = 4096;   // rho, theta in LDS (64 KB)

__global__ __launch_bounds__(64) void k_bidiag_tikhonov(const double* __restrict__ alpha_sq, int64_t a_stride,
                                                        const double* __restrict__ beta_sq, int64_t b_stride, int k,
                                                        double mu, const double* __restrict__ beta0_sq,
                                                        double* __restrict__ y) {
  extern __shared__ double lds[];
  if (threadIdx.x != 0) return;
  double* rho = lds;
  double* theta = lds + k;
  double abar = sqrt(alpha_sq[0]);
  double phibar = sqrt(*beta0_sq);
  for (int j = 0; j < k; ++j) {
    const double bj = sqrt(beta_sq[(int64_t)j * b_stride]);   // B[j+1, j]
    // rotate the damping row (mu in column j) into abar
    const double rhat = hypot(abar, mu);
    const double phihat = (abar / rhat) * phibar;
    // rotate the sub-diagonal entry into rhat
    const double r = hypot(rhat, bj);
    const double c2 = rhat / r, s2 = bj / r;
    rho[j] = r;
    y[j] = c2 * phihat;                                        // phi_j, overwritten by the back substitution
    if (j + 1 < k) {
      const double an = sqrt(alpha_sq[(int64_t)(j + 1) * a_stride]);
      theta[j + 1] = s2 * an;
      abar = -c2 * an;
    }
    phibar = s2 * phihat;
  }
  double yn = y[k - 1] / rho[k - 1];
  y[k - 1] = yn;
  for (int j = k - 2; j >= 0; --j) {
    yn = (y[j] - theta[j + 1] * yn) / rho[j];
    y[j] = yn;
  }
}

}  // namespace

extern "C" int trk_bidiag_tikhonov(const double* alpha_sq, int64_t alpha_stride, const double* beta_sq,
                                   int64_t beta_stride, int k, double mu, const double* beta0_sq, double* y,
                                   trk_stream stream) {
  TRK_REQUIRE(alpha_sq && beta_sq && beta0_sq && y, "trk_bidiag_tikhonov: NULL argument");
  TRK_REQUIRE(k >= 1 && k <= BIDIAG_MAX_K, "trk_bidiag_tikhonov: k must be in [1, 4096]");
  TRK_REQUIRE(mu >= 0.0, "trk_bidiag_tikhonov: mu must be >= 0");
  hipLaunchKernelGGL(k_bidiag_tikhonov, dim3(1), dim3(64), 2 * sizeof(double) * (size_t)k, (hipStream_t)stream, alpha_sq,
                     alpha_stride, beta_sq, beta_stride, k, mu, beta0_sq, y);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}
