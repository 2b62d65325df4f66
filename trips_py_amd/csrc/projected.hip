// projected.hip — the projected (k-sized) Tikhonov problem of the Golub-Kahan hybrid solvers, solved ON the device so
// that an iteration with a fixed regularisation parameter never visits the host.
//
// Replaces   y = np.linalg.lstsq(vstack((B_k, sqrt(lam) I)), vstack((beta0 e1, 0)))   trips/solvers/Hybrid_LSQR.py:104
// (and GK_Tikhonov.py:60) for the lower-bidiagonal B_k of Golub-Kahan: the stacked matrix is reduced to an upper
// bidiagonal R by 2k Givens rotations (the damped-LSQR elimination of Paige & Saunders 1982, §2 of "LSQR: an algorithm
// for sparse linear equations and sparse least squares"), then R y = phi is back-substituted.  O(k), backward stable,
// float64 throughout; one lane does the (inherently sequential) recurrence: ~4 us at k = 100.
#include "trk_internal.h"

#include <cmath>
#include <vector>

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

// ---------------------------------------------------------------------------------------------------------------------
// HOST: generalised cross validation for a diagonalised projected problem, minimised by bounded Brent search.
//
// Replaces the per-iteration   fminbound(gcv_funct, 1e-9, 1e2, xtol=1e-12, maxfun=1000)   of
// trips/utilities/reg_param/gcv.py:94-95 when the projected pair has been brought to (diag(s), I) — the hybrid solvers'
// SVD of B_k / H_k (Hybrid_LSQR.py:81-84, Hybrid_GMRES.py:55-58) and, after the substitution z = R_L y, the GKS / MMGKS
// pair (R_A, R_L) (GKS.py:60-63, MMGKS.py:97-100).  Objective (gcv.py:25-78 on reduced inputs):
//     G(lam) = sum_i ((1 - f_i) rhs_i)^2 / (m_eff - sum_i f_i)^2 ,   f_i = s_i^2 / (s_i^2 + lam).
// The search restates SciPy's `_minimize_scalar_bounded` (Forsythe-Malcolm-Moler fmin: golden section + successive
// parabolic interpolation) step for step, and the sums use NumPy's pairwise summation order, so that the value agrees
// with the Python path it replaces to the last bit in almost all cases; ~60 objective evaluations of O(k) each cost
// ~20 us here against ~2 ms through scipy.optimize + numpy (measured at k = 50 on the MI355X host).
#pragma clang fp contract(off)
namespace {

double np_pairwise_sum(const double* a, int n) {
  if (n < 8) {
    double res = 0.;
    for (int i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

struct GcvDiag {
  const double *s, *rhs;
  int k;
  double m_eff;
  double *f, *t;   // work, k each
  double operator()(double lam) const {
    for (int i = 0; i < k; ++i) {
      const double s2 = s[i] * s[i];
      f[i] = s2 / (s2 + lam);
      const double d = (1.0 - f[i]) * rhs[i];
      t[i] = d * d;
    }
    const double num = 0.0 + np_pairwise_sum(t, k);
    const double den = m_eff - (0.0 + np_pairwise_sum(f, k));
    return num / pow(den, 2.0);
  }
};

inline double sign1(double v) { return v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : 1.0); }   // np.sign(v) + (v == 0)

}  // namespace

extern "C" int trk_host_gcv_fminbound(const double* s, const double* rhs, int k, double m_eff, double x1, double x2,
                                      double xatol, int maxfun, double* lam_out, double* fval_out, int* nfev_out) {
  TRK_REQUIRE(s && rhs && lam_out, "trk_host_gcv_fminbound: NULL argument");
  TRK_REQUIRE(k >= 1 && x1 <= x2 && maxfun >= 1, "trk_host_gcv_fminbound: bad argument");
  std::vector<double> work(2 * (size_t)k);
  const GcvDiag func{s, rhs, k, m_eff, work.data(), work.data() + k};
  const double sqrt_eps = std::sqrt(2.2e-16);
  const double golden_mean = 0.5 * (3.0 - std::sqrt(5.0));
  double a = x1, b = x2;
  double fulc = a + golden_mean * (b - a);
  double nfc = fulc, xf = fulc;
  double rat = 0.0, e = 0.0;
  double x = xf;
  double fx = func(x);
  int num = 1;
  double fu = INFINITY;
  double ffulc = fx, fnfc = fx;
  double xm = 0.5 * (a + b);
  double tol1 = sqrt_eps * std::fabs(xf) + xatol / 3.0;
  double tol2 = 2.0 * tol1;
  while (std::fabs(xf - xm) > (tol2 - 0.5 * (b - a))) {
    bool golden = true;
    if (std::fabs(e) > tol1) {   // parabolic fit
      golden = false;
      double r = (xf - nfc) * (fx - ffulc);
      double q = (xf - fulc) * (fx - fnfc);
      double p = (xf - fulc) * q - (xf - nfc) * r;
      q = 2.0 * (q - r);
      if (q > 0.0) p = -p;
      q = std::fabs(q);
      r = e;
      e = rat;
      if ((std::fabs(p) < std::fabs(0.5 * q * r)) && (p > q * (a - xf)) && (p < q * (b - xf))) {
        rat = (p + 0.0) / q;
        x = xf + rat;
        if (((x - a) < tol2) || ((b - x) < tol2)) rat = tol1 * sign1(xm - xf);
      } else {
        golden = true;
      }
    }
    if (golden) {
      e = (xf >= xm) ? a - xf : b - xf;
      rat = golden_mean * e;
    }
    x = xf + sign1(rat) * std::fmax(std::fabs(rat), tol1);
    fu = func(x);
    ++num;
    if (fu <= fx) {
      if (x >= xf) a = xf; else b = xf;
      fulc = nfc, ffulc = fnfc;
      nfc = xf, fnfc = fx;
      xf = x, fx = fu;
    } else {
      if (x < xf) a = x; else b = x;
      if ((fu <= fnfc) || (nfc == xf)) {
        fulc = nfc, ffulc = fnfc;
        nfc = x, fnfc = fu;
      } else if ((fu <= ffulc) || (fulc == xf) || (fulc == nfc)) {
        fulc = x, ffulc = fu;
      }
    }
    xm = 0.5 * (a + b);
    tol1 = sqrt_eps * std::fabs(xf) + xatol / 3.0;
    tol2 = 2.0 * tol1;
    if (num >= maxfun) break;
  }
  *lam_out = xf;
  if (fval_out) *fval_out = fx;
  if (nfev_out) *nfev_out = num;
  return TRK_OK;
}
