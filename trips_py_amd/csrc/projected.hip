// projected.hip — the projected (k-sized) Tikhonov problem of the Golub-Kahan hybrid solvers, solved ON the device so
// that an iteration with a fixed regularisation parameter never visits the host.
//
// Replaces   y = np.linalg.lstsq(vstack((B_k, sqrt(lam) I)), vstack((beta0 e1, 0)))   trips/solvers/Hybrid_LSQR.py:104
// (and GK_Tikhonov.py:60) for the lower-bidiagonal B_k of Golub-Kahan: the stacked matrix is reduced to an upper
// bidiagonal R by 2k Givens rotations (the damped-LSQR elimination of Paige & Saunders 1982, §2 of "LSQR: an algorithm
// for sparse linear equations and sparse least squares"), then R y = phi is back-substituted.  O(k), backward stable,
// float64 throughout; one lane does the (inherently sequential) recurrence.
#include "trk_internal.h"

#include <chrono>
#include <deque>

#include <map>
#include <mutex>

#include <cmath>
#include <vector>

using namespace trk;

namespace {

constexpr int BIDIAG_MAX_K = 2048;   // the back substitution stages 3 k doubles in LDS

// One workgroup of 64 lanes.  Square roots of the new columns' squared norms are taken in parallel; lane 0 then runs the
// rotation recurrence — per column two square roots and two reciprocals (r^2 = abar^2 + mu^2 + beta^2 needs no second
// hypot) — and the back substitution R y = phi (multiplications by the stored 1/rho) out of LDS, where the dependent chain
// waits ~60 cycles per step instead of an L2 round trip.  With a `work` array the recurrence state survives between
// calls: when mu is unchanged and one column was appended (the fixed-lambda hybrid iteration) only that column is rotated,
// O(1) instead of O(k) expensive operations; any other change restarts from column 0.
// work: [0] columns done, [1] mu, [2] abar, [3] phibar, then invrho[cap], theta[cap], phi[cap].
// y_over_alpha: y_j / alpha_j is written (coefficients with respect to the un-normalised vectors alpha_j v_j).
__global__ __launch_bounds__(64) void k_bidiag_tikhonov(const double* __restrict__ alpha_sq, int64_t a_stride,
                                                        const double* __restrict__ beta_sq, int64_t b_stride, int k,
                                                        double mu, const double* __restrict__ beta0_sq,
                                                        double* __restrict__ y, double* __restrict__ work, int cap,
                                                        double* __restrict__ scratch, int y_over_alpha) {
  extern __shared__ double sm[];
  __shared__ double sh_start;
  double *s_ir = sm, *s_th = sm + k, *s_ph = sm + 2 * k;
  double* st = work ? work : scratch;                          // scratch: same layout, always restarted
  double* invrho = st + 4;
  double* theta = invrho + cap;
  double* phi = theta + cap;
  if (threadIdx.x == 0) {
    const bool resume = work && st[1] == mu && st[0] >= 1.0 && (int)st[0] < k;
    sh_start = resume ? st[0] : 0.0;
  }
  __syncthreads();
  const int j0 = (int)sh_start;
  // al[j] of the columns to process, through y (as scratch) — y is written last
  double* al = y;                                               // al[j] for j in [j0, k)
  for (int j = j0 + (int)threadIdx.x; j < k; j += 64) al[j] = sqrt(alpha_sq[(int64_t)j * a_stride]);
  // columns rotated by earlier launches: their factors go to LDS for the back substitution
  for (int j = threadIdx.x; j < j0; j += 64) {
    s_ir[j] = invrho[j];
    s_th[j] = theta[j];
    s_ph[j] = phi[j];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double mu2 = mu * mu;
    double abar, phibar;
    if (j0 == 0) {
      abar = al[0];
      phibar = sqrt(*beta0_sq);
    } else {
      abar = st[2];
      phibar = st[3];
      // st[2] holds -c2 of the previous last column (its alpha_{j0} factor was not known then)
      abar *= al[j0];
      const double t = theta[j0] * al[j0];
      theta[j0] = t;
      s_th[j0] = t;
    }
    for (int j = j0; j < k; ++j) {
      const double bj2 = beta_sq[(int64_t)j * b_stride];         // B[j+1, j]^2
      const double rhat2 = abar * abar + mu2, r2 = rhat2 + bj2;
      const double rhat = sqrt(rhat2), r = sqrt(r2), bj = sqrt(bj2);
      const double ir = 1.0 / r;
      const double phihat = (abar / rhat) * phibar;
      const double c2 = rhat * ir, s2 = bj * ir;
      invrho[j] = ir;
      s_ir[j] = ir;
      const double ph = c2 * phihat;
      phi[j] = ph;
      s_ph[j] = ph;
      if (j + 1 < k) {
        const double th = s2 * al[j + 1];
        theta[j + 1] = th;
        s_th[j + 1] = th;
        abar = -c2 * al[j + 1];
      } else {
        theta[j + 1] = s2;                                        // completed with alpha_{j+1} on resume (cap >= k + 1)
        abar = -c2;
      }
      phibar = s2 * phihat;
    }
    st[0] = (double)k;
    st[1] = mu;
    st[2] = abar;
    st[3] = phibar;
    // back substitution, the solution left in s_ph
    double yn = s_ph[k - 1] * s_ir[k - 1];
    s_ph[k - 1] = yn;
    for (int j = k - 2; j >= 0; --j) {
      yn = (s_ph[j] - s_th[j + 1] * yn) * s_ir[j];
      s_ph[j] = yn;
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < k; j += 64)
    y[j] = y_over_alpha ? s_ph[j] / sqrt(alpha_sq[(int64_t)j * a_stride]) : s_ph[j];
}

}  // namespace

extern "C" int trk_bidiag_tikhonov(const double* alpha_sq, int64_t alpha_stride, const double* beta_sq,
                                   int64_t beta_stride, int k, double mu, const double* beta0_sq, double* y,
                                   int y_over_alpha, double* work, int work_doubles, trk_stream stream) {
  TRK_REQUIRE(alpha_sq && beta_sq && beta0_sq && y, "trk_bidiag_tikhonov: NULL argument");
  TRK_REQUIRE(k >= 1 && k <= BIDIAG_MAX_K, "trk_bidiag_tikhonov: k must be in [1, 2048]");
  TRK_REQUIRE(mu >= 0.0, "trk_bidiag_tikhonov: mu must be >= 0");
  hipStream_t s = (hipStream_t)stream;
  int cap = k + 1;
  double* scratch = nullptr;
  if (work) {
    cap = (work_doubles - 4) / 3;
    TRK_REQUIRE(cap >= k + 1, "trk_bidiag_tikhonov: work holds %d doubles, %d needed", work_doubles, 3 * (k + 1) + 4);
  } else {
    if (int rc = scratch_doubles(s, (size_t)3 * cap + 4, &scratch)) return rc;
  }
  hipLaunchKernelGGL(k_bidiag_tikhonov, dim3(1), dim3(64), 3 * sizeof(double) * (size_t)k, s, alpha_sq, alpha_stride, beta_sq,
                     beta_stride, k, mu, beta0_sq, y, work, cap, scratch, y_over_alpha);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

// HOST: the same projected solve for a caller that holds B_k on the host (the hybrid solvers once lambda_k has been chosen there,
// Hybrid_LSQR.py:104): the recurrence of k_bidiag_tikhonov in the same order.  ~10 ns per column on a CPU core against ~400 ns for
// the dependent fp64 square roots and divisions of one GPU lane; the solution reaches the device inside the launch that consumes
// it (trk_gemv_n_hosty).  alpha[k], beta_sub[k] (B[j+1, j]), beta0 = ||b||; y_over_alpha as in trk_bidiag_tikhonov.
extern "C" int trk_host_bidiag_tikhonov(const double* alpha, const double* beta_sub, int k, double beta0, double mu, int y_over_alpha,
                                        double* y) {
  TRK_REQUIRE(alpha && beta_sub && y && k >= 1, "trk_host_bidiag_tikhonov: bad argument");
  TRK_REQUIRE(mu >= 0.0, "trk_host_bidiag_tikhonov: mu must be >= 0");
  std::vector<double> buf(3 * (size_t)k + 1);
  double *ir = buf.data(), *th = ir + k, *ph = th + k + 1;
  const double mu2 = mu * mu;
  double abar = alpha[0], phibar = beta0;
  for (int j = 0; j < k; ++j) {
    const double bj = beta_sub[j];
    const double rhat2 = abar * abar + mu2, r2 = rhat2 + bj * bj;
    const double rhat = std::sqrt(rhat2), r = std::sqrt(r2);
    const double inv = 1.0 / r;
    const double phihat = (abar / rhat) * phibar;
    const double c2 = rhat * inv, s2 = bj * inv;
    ir[j] = inv;
    ph[j] = c2 * phihat;
    if (j + 1 < k) {
      th[j + 1] = s2 * alpha[j + 1];
      abar = -c2 * alpha[j + 1];
    }
    phibar = s2 * phihat;
  }
  double yn = ph[k - 1] * ir[k - 1];
  ph[k - 1] = yn;
  for (int j = k - 2; j >= 0; --j) {
    yn = (ph[j] - th[j + 1] * yn) * ir[j];
    ph[j] = yn;
  }
  for (int j = 0; j < k; ++j) y[j] = y_over_alpha ? ph[j] / alpha[j] : ph[j];
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

// scipy.optimize.fminbound (bounded Brent) restated step for step; `func` is the objective
template <class F>
static void fminbound_brent(const F& func, double x1, double x2, double xatol, int maxfun, double* lam_out, double* fval_out,
                            int* nfev_out) {
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
}

// G(lam) of the hybrid solvers' projected problem WITHOUT the SVD of B_k (Hybrid_LSQR.py:81-84): with B = Q [R; 0] (k Givens
// rotations, R upper bidiagonal) and q = the first k entries of Q^T e1,
//   sum_i ((1 - f_i) beta0 u0_i)^2 = beta0^2 lam^2 || (R R^T + lam I)^-1 q ||^2 ,   sum_i f_i = k - lam trace((R R^T + lam I)^-1) ,
// f_i = s_i^2 / (s_i^2 + lam) — the same function of lam that the diagonalised form evaluates (the left null vector of B drops
// out of both), by one LDL^T of a k x k tridiagonal matrix per evaluation: O(k) per lambda and O(k) setup instead of an
// O(k^2) bidiagonal SVD per iteration (dbdsqr with one row of U: 140 us at k ~ 50 on the MI355X host — most of a Hybrid-LSQR
// iteration with regparam = 'gcv').
struct GcvBidiag {
  int k;
  double beta0, m_eff;
  const double *md, *mo, *q;   // R R^T: diagonal (k), off-diagonal (k-1); q (k)
  double *d, *e, *y;           // work, k each
  double operator()(double lam) const {
    // forward / backward pivots of M = R R^T + lam I
    d[0] = md[0] + lam;
    for (int j = 1; j < k; ++j) d[j] = md[j] + lam - mo[j - 1] * mo[j - 1] / d[j - 1];
    e[k - 1] = md[k - 1] + lam;
    for (int j = k - 2; j >= 0; --j) e[j] = md[j] + lam - mo[j] * mo[j] / e[j + 1];
    double tr = 0.0;
    for (int j = 0; j < k; ++j) tr += 1.0 / (d[j] + e[j] - (md[j] + lam));       // (M^-1)_jj
    // M z = q
    y[0] = q[0];
    for (int j = 1; j < k; ++j) y[j] = q[j] - mo[j - 1] / d[j - 1] * y[j - 1];
    double z = y[k - 1] / d[k - 1], zz = z * z;
    for (int j = k - 2; j >= 0; --j) {
      z = (y[j] - mo[j] * z) / d[j];
      zz += z * z;
    }
    const double num = beta0 * beta0 * lam * lam * zz;
    const double den = m_eff - ((double)k - lam * tr);
    return num / (den * den);
  }
};


}  // namespace

// (M v)_i, i < k, for a SYMMETRIC k x k matrix in global memory (row stride ld) and v in LDS, by one workgroup: four lanes per
// row, each walking a quarter of COLUMN i (consecutive rows -> consecutive addresses), the four partial sums met by two lane
// exchanges.  These k x k kernels are latency-bound — one thread per row meant k dependent-in-order loads per thread (k = 53:
// 20-30 us per product); a quarter of them per lane, four times the lanes busy.  sink(i, value) is called once per row.
template <class F>
__device__ __forceinline__ void symv4(const double* M, int ld, int k, const double* v, F&& sink) {
  const int part = threadIdx.x & 3;
  for (int base = 0; base < k; base += (int)blockDim.x / 4) {
    const int i = base + ((int)threadIdx.x >> 2);
    double a0 = 0.0, a1 = 0.0;
    if (i < k) {
      int j = part;
      for (; j + 4 < k; j += 8) {
        a0 += M[(size_t)j * ld + i] * v[j];
        a1 += M[(size_t)(j + 4) * ld + i] * v[j + 4];
      }
      if (j < k) a0 += M[(size_t)j * ld + i] * v[j];
    }
    double a = a0 + a1;
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    if (i < k && part == 0) sink(i, a);
  }
}

// ------------------------------------------------------------------------------------------------ repeated Gram-Schmidt by Gram matrix
// `passes` sweeps of block classical Gram-Schmidt, r <- r - V (V^T r), amount to r - V c with
//     c_0 = 0,   c_{p+1} = c_p + (h - G c_p),   h = V^T r,  G = V^T V
// (sweep p+1 projects the result of sweep p: V^T (r - V c_p) = h - G c_p) — any V, orthonormal or not.  With G kept on the
// device (one new row per appended vector) the sweeps cost ONE pass over the basis for h and one for r - V c, instead of two
// per sweep (GKS.py:86-88: three sweeps; MMGKS.py:119-120, decompositions.py:216-218: two).  k x k work, one workgroup.
// rr != nullptr (*rr = r . r): also *rho2_out = || r - V c ||^2 = r.r - 2 c.h + c.(G c), so that the caller knows the norm of the
// orthogonalised vector BEFORE the pass that forms it (trk_gemv_orth_iterate; GKS / MMGKS: r is orthogonal to V but for rounding,
// c.h and c.G c are ~1e-14 of r.r).  Lifted to a tiny positive number should rounding ever drive it to <= 0.
__global__ __launch_bounds__(256) void k_cgs_coeffs(double* __restrict__ G, int ldg, const double* __restrict__ h,
                                                    const double* __restrict__ g_new, int k, int passes, double* __restrict__ c,
                                                    const double* __restrict__ rr = nullptr, double* __restrict__ rho2_out = nullptr) {
  extern __shared__ double sh[];           // c (k) | t (k)
  __shared__ double red[2][4];
  double* cs = sh;
  double* ts = sh + k;
  if (g_new) {                             // install the Gram row / column of the newest vector (index k - 1)
    for (int j = threadIdx.x; j < k; j += blockDim.x) {
      G[(size_t)(k - 1) * ldg + j] = g_new[j];
      G[(size_t)j * ldg + (k - 1)] = g_new[j];
    }
    __threadfence_block();
  }
  for (int j = threadIdx.x; j < k; j += blockDim.x) cs[j] = 0.0;
  __syncthreads();
  __syncthreads();                         // (the installed row is read below by other threads)
  for (int p = 0; p < passes; ++p) {
    if (p == 0) {                          // c_0 = 0: the first sweep's coefficients are h itself
      for (int j = threadIdx.x; j < k; j += blockDim.x) ts[j] = h[j];
    } else {
      symv4(G, ldg, k, cs, [&](int i, double a) { ts[i] = h[i] - a; });
    }
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += blockDim.x) cs[j] += ts[j];
    __syncthreads();
  }
  if (c)
    for (int j = threadIdx.x; j < k; j += blockDim.x) c[j] = cs[j];
  if (rr) {
    double p_ch = 0.0, p_cgc = 0.0;
    symv4(G, ldg, k, cs, [&](int i, double gc) {
      p_ch += cs[i] * h[i];
      p_cgc += cs[i] * gc;
    });
    p_ch = wave_sum(p_ch);
    p_cgc = wave_sum(p_cgc);
    if ((threadIdx.x & 63) == 0) {
      red[0][threadIdx.x >> 6] = p_ch;
      red[1][threadIdx.x >> 6] = p_cgc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const double ch = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
      const double cgc = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
      const double v = *rr - 2.0 * ch + cgc;
      *rho2_out = v > 0.0 ? v : 1e-300;
    }
  }
}

// k_finalize over the 2k sums of a trk_gemv_t2 sweep (h = V^T r | the Gram row of the newest vector) and k_cgs_coeffs in ONE launch:
// workgroup o adds up output o exactly as k_finalize does (finalize_block_256: the same bits) and stores it write-through; the
// workgroup that draws the last ticket then runs the k x k recurrence on the finished sums, which it loads past its L1 / the other
// XCDs' stale L2 lines (sc1) — the hand-off of k_radon_adj_tile's split tiles.  An Arnoldi / GKS step is a chain of dependent
// launches of ~4.5 us each whatever they compute: this takes one out (Hybrid-GMRES on the 512^2 blur: 9 launches per iteration).
__global__ __launch_bounds__(256) void k_finalize_cgs(const double* __restrict__ partials, int nblocks, double* W, unsigned* cnt,
                                                      double* __restrict__ G, int ldg, int k, int passes, double* __restrict__ c) {
  extern __shared__ double sh[];           // c (k) | t (k) | h (k) | g_new (k)
  __shared__ double lds[4];
  __shared__ unsigned ticket;
  const int nout = 2 * k;
  const int o = blockIdx.x;
  const double v = finalize_block_256(partials + o, nblocks, nout, lds);
  if (threadIdx.x == 0) {
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(W + o), "v"(v) : "memory");
    ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (ticket != (unsigned)(nout - 1)) return;
  if (threadIdx.x == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // for the next launch
  double* cs = sh;
  double* ts = sh + k;
  double* hs = sh + 2 * k;
  double* gs = sh + 3 * k;
  for (int j = threadIdx.x; j < nout; j += blockDim.x) {
    double t;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(t) : "v"(W + j) : "memory");
    hs[j] = t;                             // (hs and gs are adjacent: W = h | g_new)
  }
  __syncthreads();
  // from here on: k_cgs_coeffs with h and g_new in LDS
  for (int j = threadIdx.x; j < k; j += blockDim.x) {
    G[(size_t)(k - 1) * ldg + j] = gs[j];
    G[(size_t)j * ldg + (k - 1)] = gs[j];
  }
  __threadfence_block();
  for (int j = threadIdx.x; j < k; j += blockDim.x) cs[j] = 0.0;
  __syncthreads();
  __syncthreads();
  for (int p = 0; p < passes; ++p) {
    if (p == 0) {
      for (int j = threadIdx.x; j < k; j += blockDim.x) ts[j] = hs[j];
    } else {
      symv4(G, ldg, k, cs, [&](int i, double a) { ts[i] = hs[i] - a; });
    }
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += blockDim.x) cs[j] += ts[j];
    __syncthreads();
  }
  for (int j = threadIdx.x; j < k; j += blockDim.x) c[j] = cs[j];
}

int trk::finalize_cgs(const double* part, int nblk, int k, double* W, double* G, int ldg, int passes, double* c, hipStream_t s) {
  TRK_REQUIRE(part && W && G && c && k >= 1 && k <= 1024 && ldg >= k && passes >= 1 && nblk >= 1, "finalize_cgs: bad argument");
  unsigned* cnt = nullptr;
  if (int rc = stream_ticket(s, &cnt)) return rc;
  hipLaunchKernelGGL(k_finalize_cgs, dim3(2 * k), dim3(256), 4 * (size_t)k * sizeof(double), s, part, nblk, W, cnt, G, ldg, k, passes, c);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

extern "C" int trk_cgs_coeffs(double* G, int ldg, const double* h, const double* g_new, int k, int passes, double* c,
                              trk_stream st) {
  TRK_REQUIRE(G && k >= 1 && ldg >= k && passes >= 0 && (passes == 0 || (h && c)), "trk_cgs_coeffs: bad argument");
  TRK_REQUIRE(k <= 2048, "trk_cgs_coeffs: k <= 2048");
  hipLaunchKernelGGL(k_cgs_coeffs, dim3(1), dim3(256), 2 * (size_t)k * sizeof(double), (hipStream_t)st, G, ldg, h, g_new, k, passes, c, (const double*)nullptr, (double*)nullptr);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

extern "C" int trk_cgs_coeffs_rho(double* G, int ldg, const double* h, const double* g_new, int k, int passes, double* c,
                                  const double* rr, double* rho2_out, trk_stream st) {
  TRK_REQUIRE(G && h && c && rr && rho2_out && k >= 1 && ldg >= k && passes >= 1, "trk_cgs_coeffs_rho: bad argument");
  TRK_REQUIRE(k <= 2048, "trk_cgs_coeffs_rho: k <= 2048");
  hipLaunchKernelGGL(k_cgs_coeffs, dim3(1), dim3(256), 2 * (size_t)k * sizeof(double), (hipStream_t)st, G, ldg, h, g_new, k, passes, c, rr, rho2_out);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

// ------------------------------------------------------------------------------------------------ projected Tikhonov solve from Gram data
// y = argmin ||AV y - b||^2 + lam ||LV y||^2 = (G_A + lam G_L)^-1 c  with G_A = (AV)^T AV, G_L = (LV)^T LV, c = (AV)^T b —
// what `lstsq([R_A; sqrt(lam) R_L], [Q_A^T b; 0])` of GKS.py:74 / MMGKS.py:106 solves (R^T R = G, R_A^T Q_A^T b = c) — on
// the device, from the Gram data the device already holds: with a NUMERIC regparam no scalar visits the host inside the
// loop (GKS / MMGKS each paid a download, two k x k Cholesky factorisations, a stacked least-squares solve and an upload per
// iteration: ~0.15 ms of host time with the GPU idle).  One workgroup; Cholesky in LDS, float64.
// M (k x k, row stride ld, lower triangle read) z = rhs in LDS: Cholesky M = C C^T in place, then C z' = z, C^T y = z'.
// A pivot that rounding drove to <= 0 is lifted to a tiny positive number.  All threads of the workgroup call it.
__device__ __forceinline__ void chol_solve_lds(double* M, int ld, double* z, int k) {
  for (int j = 0; j < k; ++j) {
    if (threadIdx.x == 0) {
      const double d = M[j * ld + j];
      M[j * ld + j] = sqrt(d > 0.0 ? d : 1e-300);
    }
    __syncthreads();
    const double djj = M[j * ld + j];
    for (int i = j + 1 + threadIdx.x; i < k; i += blockDim.x) M[i * ld + j] /= djj;
    __syncthreads();
    const int rem = k - j - 1;
    for (int idx = threadIdx.x; idx < rem * rem; idx += blockDim.x) {
      const int a = j + 1 + idx / rem, b = j + 1 + idx % rem;
      if (b <= a) M[a * ld + b] -= M[a * ld + j] * M[b * ld + j];
    }
    __syncthreads();
  }
  for (int j = 0; j < k; ++j) {
    if (threadIdx.x == 0) z[j] /= M[j * ld + j];
    __syncthreads();
    const double zj = z[j];
    for (int i = j + 1 + threadIdx.x; i < k; i += blockDim.x) z[i] -= M[i * ld + j] * zj;
    __syncthreads();
  }
  for (int j = k - 1; j >= 0; --j) {
    if (threadIdx.x == 0) z[j] /= M[j * ld + j];
    __syncthreads();
    const double zj = z[j];
    for (int i = threadIdx.x; i < j; i += blockDim.x) z[i] -= M[j * ld + i] * zj;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_gram_tikhonov(const double* __restrict__ GA, int lda, const double* __restrict__ GL, int ldl,
                                                       const double* __restrict__ c, int k, double lam, double* __restrict__ y) {
  extern __shared__ double sm[];              // M (k x (k+1)) | z (k)
  const int ld = k + 1;
  double* M = sm;
  double* z = sm + (size_t)k * ld;
  for (int idx = threadIdx.x; idx < k * k; idx += blockDim.x) {
    const int i = idx / k, j = idx - i * k;
    M[i * ld + j] = GA[(size_t)i * lda + j] + lam * GL[(size_t)i * ldl + j];
  }
  for (int i = threadIdx.x; i < k; i += blockDim.x) z[i] = c[i];
  __syncthreads();
  chol_solve_lds(M, ld, z, k);
  for (int i = threadIdx.x; i < k; i += blockDim.x) y[i] = z[i];
}

// The same solve when M = G_A + lam G_L only GROWS between calls (GKS with a numeric regparam: one basis vector, hence one row
// and column of both Gram matrices, per iteration; same lam): Minv holds M^-1 of the leading k_from x k_from block from the
// previous call and is bordered by rows k_from .. k-1 — u = Minv g, s = M[j][j] - g.u, Minv <- [[Minv + u u^T / s, -u / s],
// [-u^T / s, 1 / s]] — then y = Minv c: O(k^2) per new row instead of the O(k^3) Cholesky from scratch (k = 53: 79 -> ~12 us).
// k_from = 0 builds the inverse from nothing (the first call, k = projection_dim).
__device__ __forceinline__ void tikhonov_border_body(const double* GA, int lda, const double* GL, int ldl, const double* c, int k,
                                                     int k_from, double lam, double* Minv, int ldm, double* __restrict__ y, double* sm,
                                                     double* red) {
  double* g = sm;                             // sm: g (k) | u (k) | c (k)
  double* u = sm + k;
  double* cl = sm + 2 * k;
  for (int i = threadIdx.x; i < k; i += blockDim.x) cl[i] = c[i];
  for (int j = k_from; j < k; ++j) {
    for (int i = threadIdx.x; i <= j; i += blockDim.x) g[i] = GA[(size_t)j * lda + i] + lam * GL[(size_t)j * ldl + i];
    __syncthreads();
    double part = 0.0;
    symv4(Minv, ldm, j, g, [&](int i, double a) {             // u = Minv g
      u[i] = a;
      part += a * g[i];
    });
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    const double sinv = 1.0 / (g[j] - (red[0] + red[1] + red[2] + red[3]));
    for (int idx = threadIdx.x; idx < j * j; idx += blockDim.x) {
      const int a = idx / j, b = idx - a * j;
      Minv[(size_t)a * ldm + b] += u[a] * u[b] * sinv;
    }
    for (int i = threadIdx.x; i < j; i += blockDim.x) {
      Minv[(size_t)i * ldm + j] = -u[i] * sinv;
      Minv[(size_t)j * ldm + i] = -u[i] * sinv;
    }
    if (threadIdx.x == 0) Minv[(size_t)j * ldm + j] = sinv;
    __syncthreads();                            // (one workgroup: its global writes are visible to it after the barrier)
  }
  __syncthreads();
  symv4(Minv, ldm, k, cl, [&](int i, double a) { y[i] = a; });
}
__global__ __launch_bounds__(256) void k_gram_tikhonov_border(const double* __restrict__ GA, int lda, const double* __restrict__ GL,
                                                              int ldl, const double* __restrict__ c, int k, int k_from, double lam,
                                                              double* Minv, int ldm, double* __restrict__ y) {
  extern __shared__ double sm[];              // g (k) | u (k) | c (k)
  __shared__ double red[4];
  tikhonov_border_body(GA, lda, GL, ldl, c, k, k_from, lam, Minv, ldm, y, sm, red);
}

// GKS: row / column k of a Gram matrix G = V^T M V (M = A^T A or L^T L) for the basis vector v_k = (r - V c) / rho that the sweep
// has just produced, from what the sweep's own pass over V left behind — a = V^T (M r), the coefficients c, s = r . M r and
// rho^2 = ||r - V c||^2 — with no further pass over the basis:
//   G[i][k] = (a_i - (G c)_i) / rho  (i < k),     G[k][k] = (s - 2 c.a + c.(G c)) / rho^2 ;
// optionally the same for a projected right-hand side  rhs_k = (t - c . rhs[0..k)) / rho  (t = r . (A^T b)).
// r is the residual of the projected normal equations, so V^T r = 0 but for rounding and c is of rounding size: nothing cancels.
__device__ __forceinline__ void gram_row_body(double* G, int ldg, int k, const double* __restrict__ a, const double* __restrict__ c,
                                              const double* __restrict__ s_rr, const double* __restrict__ rho2, double* rhs,
                                              const double* tb, double* cl, double (*red)[4]) {
  const double rho = sqrt(*rho2);
  for (int i = threadIdx.x; i < k; i += blockDim.x) cl[i] = c[i];
  __syncthreads();
  double p_ca = 0.0, p_cgc = 0.0, p_cr = 0.0;
  symv4(G, ldg, k, cl, [&](int i, double gc) {
    const double v = (a[i] - gc) / rho;
    G[(size_t)i * ldg + k] = v;              // (row / column k: outside what symv4 reads)
    G[(size_t)k * ldg + i] = v;
    p_ca += cl[i] * a[i];
    p_cgc += cl[i] * gc;
    if (rhs) p_cr += cl[i] * rhs[i];
  });
  p_ca = wave_sum(p_ca);
  p_cgc = wave_sum(p_cgc);
  p_cr = wave_sum(p_cr);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = p_ca;
    red[1][threadIdx.x >> 6] = p_cgc;
    red[2][threadIdx.x >> 6] = p_cr;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double ca = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const double cgc = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    G[(size_t)k * ldg + k] = (*s_rr - 2.0 * ca + cgc) / (rho * rho);
    if (rhs) rhs[k] = (*tb - ((red[2][0] + red[2][1]) + (red[2][2] + red[2][3]))) / rho;
  }
}
__global__ __launch_bounds__(256) void k_gram_row_from_sweep(double* G, int ldg, int k, const double* __restrict__ a,
                                                             const double* __restrict__ c, const double* __restrict__ s_rr,
                                                             const double* __restrict__ rho2, double* rhs, const double* tb) {
  extern __shared__ double cl[];              // c (k)
  __shared__ double red[3][4];
  gram_row_body(G, ldg, k, a, c, s_rr, rho2, rhs, tb, cl, red);
}

// GKS's one-pass form (trk_gemv_orth_iterate): everything between the h-sweep's products and the pass that needs y' is k x k work of
// one workgroup — rows k of G_A and G_L, then the bordered inverse over k + 1 vectors and y' — and was three launches of ~6 us each
// whatever they compute (an install or a row for G_A, a row for G_L, the solve).  One launch: the bodies above, one after the other.
//   A side: ga_new != NULL — the k + 1 entries of row k as a pass over the kept images left them (the Radon operator) — or a_A != NULL —
//           from the sweep's products, with the projected right-hand side's entry k (stencil operators);   L side: from the sweep.
__global__ __launch_bounds__(256) void k_gks_rows_solve(double* GA, double* GL, int ldg, int k, const double* __restrict__ ga_new,
                                                        const double* __restrict__ a_A, const double* __restrict__ s_A,
                                                        const double* __restrict__ tb, const double* __restrict__ a_L,
                                                        const double* __restrict__ s_L, const double* __restrict__ c_sweep,
                                                        const double* __restrict__ rho2, double* c_rhs, double lam, double* Minv, int ldm,
                                                        int k_from, double* __restrict__ y) {
  extern __shared__ double sm[];              // 3 (k + 1) doubles
  __shared__ double red[3][4];
  if (ga_new) {
    for (int j = threadIdx.x; j <= k; j += blockDim.x) {
      GA[(size_t)k * ldg + j] = ga_new[j];
      GA[(size_t)j * ldg + k] = ga_new[j];
    }
  } else {
    gram_row_body(GA, ldg, k, a_A, c_sweep, s_A, rho2, c_rhs, tb, sm, red);
  }
  __syncthreads();
  gram_row_body(GL, ldg, k, a_L, c_sweep, s_L, rho2, nullptr, nullptr, sm, red);
  __threadfence_block();
  __syncthreads();                            // (one workgroup: its own global writes are visible to it after the barrier)
  tikhonov_border_body(GA, ldg, GL, ldg, c_rhs, k + 1, k_from, lam, Minv, ldm, y, sm, &red[0][0]);
}

extern "C" int trk_gks_rows_solve(double* GA, double* GL, int ldg, int k, const double* ga_new, const double* a_A, const double* s_A,
                                  const double* tb, const double* a_L, const double* s_L, const double* c_sweep, const double* rho2,
                                  double* c_rhs, double lam, double* Minv, int ldm, int k_from, double* y, trk_stream st) {
  TRK_REQUIRE(GA && GL && a_L && s_L && c_sweep && rho2 && c_rhs && Minv && y && k >= 1 && ldg >= k + 1 && ldm >= k + 1,
              "trk_gks_rows_solve: bad argument");
  TRK_REQUIRE((ga_new != nullptr) != (a_A != nullptr), "trk_gks_rows_solve: the A-side row either as ga_new or from the sweep (a_A, s_A, tb)");
  TRK_REQUIRE(ga_new || (s_A && tb), "trk_gks_rows_solve: a_A needs s_A = r . A^T A r and tb = r . A^T b");
  TRK_REQUIRE(k_from >= 0 && k_from <= k + 1 && k + 1 <= 2048, "trk_gks_rows_solve: 0 <= k_from <= k + 1 <= 2048");
  hipLaunchKernelGGL(k_gks_rows_solve, dim3(1), dim3(256), 3 * (size_t)(k + 1) * sizeof(double), (hipStream_t)st, GA, GL, ldg, k, ga_new, a_A,
                     s_A, tb, a_L, s_L, c_sweep, rho2, c_rhs, lam, Minv, ldm, k_from, y);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

extern "C" int trk_gram_row_from_sweep(double* G, int ldg, int k, const double* a, const double* c, const double* s_rr,
                                       const double* rho2, double* rhs, const double* tb, trk_stream st) {
  TRK_REQUIRE(G && a && c && s_rr && rho2 && k >= 1 && ldg >= k + 1, "trk_gram_row_from_sweep: bad argument");
  TRK_REQUIRE(!rhs || tb, "trk_gram_row_from_sweep: rhs given without t = r . (A^T b)");
  hipLaunchKernelGGL(k_gram_row_from_sweep, dim3(1), dim3(256), (size_t)k * sizeof(double), (hipStream_t)st, G, ldg, k, a, c, s_rr, rho2, rhs, tb);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

// Hybrid-GMRES's projected problem on the device (Hybrid_GMRES.py:69-77 with a numeric regparam):
//   y = argmin || H_k y - beta0 e1 ||^2 + lam || y ||^2 ,   H_k the (k+1) x k Hessenberg matrix of Arnoldi.
// One workgroup appends column k-1 of H from the scalars the orthogonalisation sweep left on the device (its k combined
// Gram-Schmidt coefficients and ||w||^2), extends G = H^T H by its new row/column, and solves (G + lam I) y = beta0 H[0,:]^T.
// H: column-major, column stride ldh >= k+1; G, Minv: row stride ldg.
//   mode 0  Cholesky of G + lam I in LDS, from scratch: O(k^3), any k, any history of lam (what the first version always did:
//           ~50 us per call averaged over k = 1..100 — serial time the host path hides behind the next Arnoldi step);
//   mode 1  M = G + lam I grew by one row and column since the previous call with the same lam, and Minv holds its previous
//           inverse: bordering update  u = Minv g, s = gamma - g.u,  Minv <- [[Minv + u u^T / s, -u / s], [-u^T / s, 1 / s]],
//           then y = Minv c — O(k^2), four barriers, no sequential dependency (cond(M) <= ||G|| / lam: harmless in float64);
//   mode 2  k <= 2: Minv formed directly (the start of the chain; the reference solves its first problem with lam = 0).
__global__ __launch_bounds__(256) void k_hess_tikhonov(double* __restrict__ H, int ldh, double* __restrict__ G, double* Minv,
                                                       int ldg, const double* __restrict__ coef, const double* __restrict__ coef2,
                                                       const double* __restrict__ nrm2_sq, double beta0, int k, double lam,
                                                       int mode, double* __restrict__ y) {
  extern __shared__ double sm[];              // new column (k+1) | g (k) | u (k) | c (k) | [mode 0: M (k x (k+1)) | z (k)]
  double* hc = sm;
  double* g = hc + (k + 1);
  double* u = g + k;
  double* cv = u + k;
  __shared__ double red[4];
  for (int r = threadIdx.x; r <= k; r += blockDim.x) {
    const double v = r < k ? coef[r] + (coef2 ? coef2[r] : 0.0) : sqrt(*nrm2_sq);
    hc[r] = v;
    H[(size_t)(k - 1) * ldh + r] = v;
  }
  __syncthreads();
  // g[i] = G[i][k-1] = sum_r H[r][i] H[r][k-1]; column i of H is non-zero in rows 0 .. i+1
  for (int i = threadIdx.x; i < k; i += blockDim.x) {
    double a = 0.0;
    if (i == k - 1) {
      for (int r = 0; r <= k; ++r) a += hc[r] * hc[r];
    } else {
      const double* col = H + (size_t)i * ldh;
      for (int r = 0; r <= i + 1; ++r) a += col[r] * hc[r];
    }
    g[i] = a;
    G[(size_t)i * ldg + (k - 1)] = a;
    G[(size_t)(k - 1) * ldg + i] = a;
    cv[i] = beta0 * (i == k - 1 ? hc[0] : H[(size_t)i * ldh]);
  }
  __syncthreads();                            // (one workgroup: its own global writes are visible to it after the barrier)
  if (mode == 0) {
    const int ld = k + 1;
    double* M = cv + k;
    double* z = M + (size_t)k * ld;
    for (int idx = threadIdx.x; idx < k * k; idx += blockDim.x) {
      const int i = idx / k, j = idx - i * k;
      M[i * ld + j] = G[(size_t)i * ldg + j] + (i == j ? lam : 0.0);
    }
    for (int i = threadIdx.x; i < k; i += blockDim.x) z[i] = cv[i];
    __syncthreads();
    chol_solve_lds(M, ld, z, k);
    for (int i = threadIdx.x; i < k; i += blockDim.x) y[i] = z[i];
    return;
  }
  const double gamma = g[k - 1] + lam;
  if (mode == 2) {                            // k = 1 or 2: the inverse written out
    if (threadIdx.x == 0) {
      if (k == 1) {
        Minv[0] = 1.0 / gamma;
      } else {
        const double a = G[0] + lam, b = g[0], det = a * gamma - b * b;
        Minv[0] = gamma / det;
        Minv[1] = Minv[ldg] = -b / det;
        Minv[ldg + 1] = a / det;
      }
    }
    __syncthreads();
  } else {
    const int m = k - 1;
    // u = Minv g (Minv symmetric: thread i walks column i, consecutive threads read consecutive addresses)
    double part = 0.0;
    symv4(Minv, ldg, m, g, [&](int i, double a) {
      u[i] = a;
      part += a * g[i];
    });
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    const double sinv = 1.0 / (gamma - (red[0] + red[1] + red[2] + red[3]));     // Schur complement of the new diagonal entry
    for (int idx = threadIdx.x; idx < m * m; idx += blockDim.x) {
      const int i = idx / m, j = idx - i * m;
      Minv[(size_t)i * ldg + j] += u[i] * u[j] * sinv;
    }
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
      Minv[(size_t)i * ldg + m] = -u[i] * sinv;
      Minv[(size_t)m * ldg + i] = -u[i] * sinv;
    }
    if (threadIdx.x == 0) Minv[(size_t)m * ldg + m] = sinv;
    __syncthreads();
  }
  symv4(Minv, ldg, k, cv, [&](int i, double a) { y[i] = a; });                    // y = Minv c
}

// both kernels keep the k x (k+1) factor in LDS: up to 160 KB per workgroup on gfx950 (opt-in above 64 KB)
static int tikhonov_lds(const void* kernel, size_t bytes) {
  constexpr size_t kMaxDyn = 159 * 1024;        // 160 KB per workgroup minus the kernels' few static bytes
  if (bytes > kMaxDyn) return fail(TRK_EINVAL, "projected Tikhonov solve: k too large for 160 KB of LDS");
  if (bytes > 64 * 1024) {
    static std::mutex mu;
    static std::map<const void*, size_t> granted;
    std::lock_guard<std::mutex> lk(mu);
    if (granted[kernel] < bytes) {
      if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDyn) != hipSuccess)
        return fail(TRK_EHIP, "projected Tikhonov solve: cannot raise the dynamic LDS limit");
      granted[kernel] = kMaxDyn;
    }
  }
  return TRK_OK;
}

extern "C" int trk_gram_tikhonov(const double* GA, int lda, const double* GL, int ldl, const double* c, int k, double lam,
                                 double* Minv, int ldm, int k_from, double* y, trk_stream st) {
  TRK_REQUIRE(GA && GL && c && y && k >= 1 && lda >= k && ldl >= k, "trk_gram_tikhonov: bad argument");
  if (Minv) {
    TRK_REQUIRE(ldm >= k && k_from >= 0 && k_from <= k, "trk_gram_tikhonov: bordering form needs ldm >= k and 0 <= k_from <= k");
    hipLaunchKernelGGL(k_gram_tikhonov_border, dim3(1), dim3(256), 3 * (size_t)k * sizeof(double), (hipStream_t)st, GA, lda, GL, ldl,
                       c, k, k_from, lam, Minv, ldm, y);
    TRK_LAUNCH_CHECK();
    return TRK_OK;
  }
  TRK_REQUIRE(k <= 139, "trk_gram_tikhonov: k <= 139 (the factor lives in LDS)");
  const size_t bytes = ((size_t)k * (k + 1) + k) * sizeof(double);
  if (int rc = tikhonov_lds(reinterpret_cast<const void*>(k_gram_tikhonov), bytes)) return rc;
  hipLaunchKernelGGL(k_gram_tikhonov, dim3(1), dim3(256), bytes, (hipStream_t)st, GA, lda, GL, ldl, c, k, lam, y);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

extern "C" int trk_hess_tikhonov(double* H, int ldh, double* G, double* Minv, int ldg, const double* coef, const double* coef2,
                                 const double* nrm2_sq, double beta0, int k, double lam, int mode, double* y, trk_stream st) {
  TRK_REQUIRE(H && G && coef && nrm2_sq && y && k >= 1 && ldh >= k + 1 && ldg >= k, "trk_hess_tikhonov: bad argument");
  TRK_REQUIRE(mode >= 0 && mode <= 2 && (mode == 0 || Minv), "trk_hess_tikhonov: mode 0 (Cholesky), 1 (bordering update), 2 (k <= 2) ; modes 1, 2 need Minv");
  TRK_REQUIRE(mode != 2 || k <= 2, "trk_hess_tikhonov: mode 2 starts the chain at k <= 2");
  TRK_REQUIRE(mode != 1 || (k >= 2 && lam > 0.0), "trk_hess_tikhonov: the bordering update needs k >= 2 and lam > 0");
  TRK_REQUIRE(mode != 0 || k <= 139, "trk_hess_tikhonov: Cholesky mode needs k <= 139 (the factor lives in LDS)");
  TRK_REQUIRE(lam >= 0.0, "trk_hess_tikhonov: lam must be >= 0");
  size_t bytes = ((size_t)(k + 1) + 3 * (size_t)k) * sizeof(double);
  if (mode == 0) bytes += ((size_t)k * (k + 1) + k) * sizeof(double);
  if (int rc = tikhonov_lds(reinterpret_cast<const void*>(k_hess_tikhonov), bytes)) return rc;
  hipLaunchKernelGGL(k_hess_tikhonov, dim3(1), dim3(256), bytes, (hipStream_t)st, H, ldh, G, Minv, ldg, coef, coef2, nrm2_sq,
                     beta0, k, lam, mode, y);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

// The discrepancy principle's Newton iteration (trk_host_dp_newton) for the bidiagonal projected problem without the SVD of B_k:
// with B = Q [R; 0] and w = Q^T bproj, || bhat / (sv beta + 1) ||^2 = || (beta R R^T + I)^-1 w_{1..k} ||^2 + w_{k+1}^2 — the last entry
// is the component along the left null vector of B, which the Newton step does not move — so every step is one LDL^T of a
// k x k tridiagonal matrix and two solves with it.  Same start (beta = 1e-8), same stopping rule, same `testzero` branch
// (discrepancy_principle.py:68-99).
extern "C" int trk_host_dp_bidiag(const double* alpha, const double* beta_sub, int k, const double* bproj, double target,
                                  double extra, double* alpha_out, int* alpha_set, int* iters_out, double* testzero_out) {
  TRK_REQUIRE(alpha && beta_sub && bproj && alpha_out && alpha_set, "trk_host_dp_bidiag: NULL argument");
  TRK_REQUIRE(k >= 1, "trk_host_dp_bidiag: k must be >= 1");
  std::vector<double> wk(7 * (size_t)k + 1);
  double *md = wk.data(), *mo = md + k, *w = mo + k, *d = w + (k + 1), *z = d + k, *y = z + k, *tmp = y + k;
  for (int j = 0; j <= k; ++j) w[j] = bproj[j];
  double diag = alpha[0], t_prev = 0.0;
  for (int j = 0; j < k; ++j) {                                  // rotation j mixes rows j, j+1 of B and of w
    const double rr = std::hypot(diag, beta_sub[j]);
    const double c = rr > 0.0 ? diag / rr : 1.0, sn = rr > 0.0 ? beta_sub[j] / rr : 0.0;
    const double wj = c * w[j] + sn * w[j + 1], wn = -sn * w[j] + c * w[j + 1];
    w[j] = wj;
    w[j + 1] = wn;
    const double t = (j + 1 < k) ? sn * alpha[j + 1] : 0.0;
    md[j] = rr * rr + t * t;
    if (j > 0) mo[j - 1] = t_prev * rr;
    t_prev = t;
    diag = (j + 1 < k) ? c * alpha[j + 1] : 0.0;
  }
  const double null2 = w[k] * w[k];
  const double testzero = null2 - target + extra;                // (:71-76) the discrepancy cannot be reached yet
  if (testzero_out) *testzero_out = testzero;
  *alpha_out = 0.0;
  *alpha_set = 1;
  if (iters_out) *iters_out = 0;
  if (!(testzero < 0.0)) return TRK_OK;
  auto solve = [&](double bt, const double* rhs, double* out) {  // (bt M + I) out = rhs with the pivots in d (Thomas)
    tmp[0] = rhs[0];
    for (int j = 1; j < k; ++j) tmp[j] = rhs[j] - bt * mo[j - 1] / d[j - 1] * tmp[j - 1];
    out[k - 1] = tmp[k - 1] / d[k - 1];
    for (int j = k - 2; j >= 0; --j) out[j] = (tmp[j] - bt * mo[j] * out[j + 1]) / d[j];
  };
  double bt = 1e-8, al = 0.0;
  int it = 0, have = 0;
  while (it < 30 || (it <= 100 && std::fabs(al) < 1e-16)) {
    d[0] = bt * md[0] + 1.0;
    for (int j = 1; j < k; ++j) d[j] = bt * md[j] + 1.0 - (bt * mo[j - 1]) * (bt * mo[j - 1]) / d[j - 1];
    solve(bt, w, z);
    solve(bt, z, y);
    double zz = null2, zwz = 0.0;
    for (int j = 0; j < k; ++j) {
      zz += z[j] * z[j];
      zwz += z[j] * (y[j] - z[j]);
    }
    const double f = zz + extra - target;
    const double fp = 2.0 / bt * zwz;
    const double bt_new = bt - f / fp;
    if (std::fabs(bt_new - bt) < 1e-12 * bt) break;
    bt = bt_new;
    al = 1.0 / bt_new;
    have = 1;
    ++it;
  }
  *alpha_out = al;
  *alpha_set = have;
  if (iters_out) *iters_out = it;
  return TRK_OK;
}

extern "C" int trk_host_gcv_fminbound(const double* s, const double* rhs, int k, double m_eff, double x1, double x2,
                                      double xatol, int maxfun, double* lam_out, double* fval_out, int* nfev_out) {
  TRK_REQUIRE(s && rhs && lam_out, "trk_host_gcv_fminbound: NULL argument");
  TRK_REQUIRE(k >= 1 && x1 <= x2 && maxfun >= 1, "trk_host_gcv_fminbound: bad argument");
  std::vector<double> work(2 * (size_t)k);
  const GcvDiag func{s, rhs, k, m_eff, work.data(), work.data() + k};
  fminbound_brent(func, x1, x2, xatol, maxfun, lam_out, fval_out, nfev_out);
  return TRK_OK;
}

extern "C" int trk_host_gcv_bidiag(const double* alpha, const double* beta, int k, double beta0, double m_eff, double x1,
                                   double x2, double xatol, int maxfun, double* lam_out, double* fval_out, int* nfev_out) {
  TRK_REQUIRE(alpha && beta && lam_out, "trk_host_gcv_bidiag: NULL argument");
  TRK_REQUIRE(k >= 1 && x1 <= x2 && maxfun >= 1, "trk_host_gcv_bidiag: bad argument");
  std::vector<double> w(7 * (size_t)k);
  double *md = w.data(), *mo = md + k, *q = mo + k, *d = q + k, *e = d + k, *y = e + k, *r = y + k;
  // B = Q [R; 0]: rotation j mixes rows j, j+1 and removes beta[j]; t = R[j][j+1]; g = what is left of e1 for the rows below
  double diag = alpha[0], g = 1.0, t_prev = 0.0;
  for (int j = 0; j < k; ++j) {
    const double rr = std::hypot(diag, beta[j]);
    const double c = rr > 0.0 ? diag / rr : 1.0, sn = rr > 0.0 ? beta[j] / rr : 0.0;
    r[j] = rr;
    q[j] = c * g;
    g = -sn * g;
    const double t = (j + 1 < k) ? sn * alpha[j + 1] : 0.0;          // R[j][j+1]
    md[j] = rr * rr + t * t;
    if (j > 0) mo[j - 1] = t_prev * rr;                                // (R R^T)[j-1][j] = R[j-1][j] R[j][j]
    t_prev = t;
    diag = (j + 1 < k) ? c * alpha[j + 1] : 0.0;
  }
  const GcvBidiag func{k, beta0, m_eff, md, mo, q, d, e, y};
  fminbound_brent(func, x1, x2, xatol, maxfun, lam_out, fval_out, nfev_out);
  return TRK_OK;
}

// HOST: the Newton iteration of the discrepancy principle on beta = 1/alpha
// (trips/utilities/reg_param/discrepancy_principle.py:80-99, dptype 'tikhonov'):
//   f(beta) = || bhat / (sv*beta + 1) ||^2 + extra - target,  beta_0 = 1e-8, at least 30 steps unless the update falls
//   below 1e-12*beta.  *alpha_set = 0 when the loop ended before alpha was assigned (the reference then returns None).
extern "C" int trk_host_dp_newton(const double* sv, const double* bhat, int n, double target, double extra,
                                  double* alpha_out, int* alpha_set, int* iters_out) {
  TRK_REQUIRE(sv && bhat && alpha_out && alpha_set, "trk_host_dp_newton: NULL argument");
  TRK_REQUIRE(n >= 1, "trk_host_dp_newton: n must be >= 1");
  double beta = 1e-8, alpha = 0.0;
  int it = 0, have = 0;
  while (it < 30 || (it <= 100 && std::fabs(alpha) < 1e-16)) {
    double zz = 0.0, zwz = 0.0;
    for (int i = 0; i < n; ++i) {
      const double den = sv[i] * beta + 1.0;
      const double z = bhat[i] / den;
      const double w = z / den;
      zz += z * z;
      zwz += z * (w - z);
    }
    const double nz = std::sqrt(zz);
    const double f = nz * nz + extra - target;
    const double fp = 2.0 / beta * zwz;
    const double beta_new = beta - f / fp;
    if (std::fabs(beta_new - beta) < 1e-12 * beta) break;
    beta = beta_new;
    alpha = 1.0 / beta_new;
    have = 1;
    ++it;
  }
  *alpha_out = alpha;
  *alpha_set = have;
  if (iters_out) *iters_out = it;
  return TRK_OK;
}


// ------------------------------------------------------------------ lambda searches on a worker thread
// The hybrid solvers choose lambda_k on the host from B_k while the device runs the steps after k; at 512^2 the bounded Brent
// search for GCV (a dependent chain of divisions, O(k) per evaluation, ~40 evaluations) is 40 of the ~100 us the host spends per
// iteration — more than the device needs for it.  A worker thread of the library takes the search: posting and collecting are
// two cheap calls, the search overlaps the host's enqueueing of the next step.  One job at a time; inputs are copied at post.
#include <atomic>
#include <condition_variable>
#include <thread>

struct trk_host_worker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::atomic<int> state{0};      // 0 idle, 1 posted, 2 done, 3 stop
  int kind = 0;                   // 0 gcv_bidiag, 1 dp_bidiag
  std::vector<double> a, b, c;
  int k = 0;
  double p[6] = {0, 0, 0, 0, 0, 0};
  int maxfun = 0;
  double lam = 0.0;
  int have = 0, rc = 0;
  // kind 2 (Hybrid-GMRES): the whole projected problem of one iterate — bidiagonalisation of [beta0 e1 | H] by the caller's LAPACK
  // (dgebrd / dormbr: plain C pointers, every argument by reference), the GCV search, the Tikhonov solve, y = P' z, the residual
  void* gebrd = nullptr;
  void* ormbr = nullptr;
  std::vector<double> M, H, d, e, tq, tp, work, y;
  double resid = 0.0;
  int dp_solves_zero = 0;         // kind 3: "the discrepancy cannot be reached yet" (lambda = 0) is solved here too (trk_hgmres's workers)
  int y_valid = 0;                // the last Hessenberg job left y and resid
};

namespace {
typedef void (*gebrd_fn)(int*, int*, double*, int*, double*, double*, double*, double*, double*, int*, int*);
typedef void (*ormbr_fn)(char*, char*, char*, int*, int*, int*, double*, int*, double*, double*, int*, double*, int*, int*);

// Hybrid_GMRES.py:54-80 for one k, from H_k ((k+1) x k, column-major, ld = k+1) and beta0: M = [beta0 e1 | H] = Q B P^T (dgebrd; the
// first column is a multiple of e1, so Q^T (beta0 e1) = d[0] e1 and P = diag(1, P')): H = Q B[:, 1:] P'^T with B[:, 1:] LOWER bidiagonal,
// diagonal e[0..k), sub-diagonal d[1..k].  lambda by 'standard' GCV on that triple (fullsize k: the k x k diag(s) of :58), z the
// Tikhonov minimiser, y = P' z, and the reference's relResidual (:80: a (k+1,) minus a (k+1, 1) — the Frobenius norm of a matrix).
int hess_job(trk_host_worker* w, bool dp, bool fixed = false) {
  const int k = w->k, n = k + 1;
  w->y_valid = 0;
  w->M.assign((size_t)n * n, 0.0);
  w->M[0] = w->p[0];
  for (int j = 0; j < k; ++j)
    for (int i = 0; i < n; ++i) w->M[(size_t)(j + 1) * n + i] = w->H[(size_t)j * n + i];
  w->d.resize(n); w->e.resize(n); w->tq.resize(n); w->tp.resize(n);
  int lwork = 64 * n, info = 0, nn = n, one = 1;
  w->work.resize(lwork);
  ((gebrd_fn)w->gebrd)(&nn, &nn, w->M.data(), &nn, w->d.data(), w->e.data(), w->tq.data(), w->tp.data(), w->work.data(), &lwork, &info);
  if (info != 0) return ::trk::fail(TRK_EINVAL, "trk_host_worker (hess_gcv): dgebrd failed (info = %d)", info);
  const double* alpha = w->e.data();
  const double* beta = w->d.data() + 1;
  const double b0 = w->d[0];
  double lam = 0.0;
  char vect = 'P', side = 'L', trans = 'N';
  if (dp) {
    // the discrepancy principle (discrepancy_principle.py:68-99) wants V_{k+1}^T b in the left basis of the bidiagonal form: Q^T bproj
    vect = 'Q'; trans = 'T';
    ((ormbr_fn)w->ormbr)(&vect, &side, &trans, &nn, &one, &nn, w->M.data(), &nn, w->tq.data(), w->c.data(), &nn, w->work.data(), &lwork, &info);
    if (info != 0) return ::trk::fail(TRK_EINVAL, "trk_host_worker (hess_dp): dormbr failed (info = %d)", info);
    w->have = 0;
    if (int rc = trk_host_dp_bidiag(alpha, beta, k, w->c.data(), w->p[1], w->p[2], &lam, &w->have, nullptr, nullptr)) return rc;
    w->lam = lam;
    // the caller's in-line branches (unassigned / not reachable yet) — unless it asked for the unreachable case's lambda = 0 solve
    if (!w->have || !(lam > 0.0 || (lam == 0.0 && w->dp_solves_zero))) return TRK_OK;
    vect = 'P'; trans = 'N';
  } else if (fixed) {                                           // (kind 4: lambda is the caller's number — no search)
    lam = w->p[5];
    w->lam = lam;
    w->have = 1;
  } else {
    if (int rc = trk_host_gcv_bidiag(alpha, beta, k, b0, w->p[1], w->p[2], w->p[3], w->p[4], w->maxfun, &lam, nullptr, nullptr)) return rc;
    w->lam = lam;
    w->have = 1;
  }
  w->y.assign(n, 0.0);
  if (int rc = trk_host_bidiag_tikhonov(alpha, beta, k, b0, sqrt(lam), 0, w->y.data() + 1)) return rc;
  ((ormbr_fn)w->ormbr)(&vect, &side, &trans, &nn, &one, &nn, w->M.data(), &nn, w->tp.data(), w->y.data(), &nn, w->work.data(), &lwork, &info);
  if (info != 0) return ::trk::fail(TRK_EINVAL, "trk_host_worker (hess_gcv): dormbr failed (info = %d)", info);
  double r2 = 0.0;
  for (int i = 0; i < n; ++i) {
    double hy = 0.0;
    for (int j = 0; j < k; ++j) hy += w->H[(size_t)j * n + i] * w->y[1 + j];
    r2 += (w->p[0] - hy) * (w->p[0] - hy) + (double)k * hy * hy;
  }
  w->resid = sqrt(r2);
  w->y_valid = 1;
  return TRK_OK;
}

// poll the worker's state for up to ~0.4 ms before going to sleep on the condition variable: a wake-up through the kernel costs
// 50-100 us, a job 30-150 us — a caller that arrives a little early (the one-call-per-iteration loop does) must not pay for a sleep
template <class F>
void spin_until(trk_host_worker* w, F&& ready) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    for (int i = 0; i < 512; ++i) {
      if (ready(w->state.load(std::memory_order_acquire))) return;
      __builtin_ia32_pause();
    }
    if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400)) return;
  }
}

void host_worker_main(trk_host_worker* w) {
  for (;;) {
    // spin for a while (jobs arrive every ~60 us inside a solve), then sleep
    spin_until(w, [](int st) { return st == 1 || st == 3; });
    if (w->state.load(std::memory_order_acquire) != 1 && w->state.load(std::memory_order_acquire) != 3) {
      std::unique_lock<std::mutex> lk(w->m);
      w->cv.wait(lk, [&] { const int s = w->state.load(std::memory_order_acquire); return s == 1 || s == 3; });
    }
    if (w->state.load(std::memory_order_acquire) == 3) return;
    if (w->kind == 0) {
      w->have = 1;
      w->rc = trk_host_gcv_bidiag(w->a.data(), w->b.data(), w->k, w->p[0], w->p[1], w->p[2], w->p[3], w->p[4], w->maxfun, &w->lam,
                                  nullptr, nullptr);
    } else if (w->kind == 2 || w->kind == 3 || w->kind == 4) {
      w->rc = hess_job(w, w->kind == 3, w->kind == 4);
    } else {
      w->rc = trk_host_dp_bidiag(w->a.data(), w->b.data(), w->k, w->c.data(), w->p[0], w->p[1], &w->lam, &w->have, nullptr, nullptr);
    }
    {
      std::lock_guard<std::mutex> lk(w->m);
      w->state.store(2, std::memory_order_release);
    }
    w->cv.notify_all();
  }
}
}  // namespace

extern "C" int trk_host_worker_create(trk_host_worker** out) {
  TRK_REQUIRE(out, "trk_host_worker_create: NULL argument");
  auto* w = new trk_host_worker;
  w->th = std::thread(host_worker_main, w);
  *out = w;
  return TRK_OK;
}

extern "C" int trk_host_worker_destroy(trk_host_worker* w) {
  if (!w) return TRK_OK;
  {
    std::unique_lock<std::mutex> lk(w->m);
    w->cv.wait(lk, [&] { return w->state.load() != 1; });          // a running job finishes first
    w->state.store(3, std::memory_order_release);
  }
  w->cv.notify_all();
  w->th.join();
  delete w;
  return TRK_OK;
}

static int host_worker_post(trk_host_worker* w, int kind) {
  {
    std::lock_guard<std::mutex> lk(w->m);
    w->kind = kind;
    w->state.store(1, std::memory_order_release);
  }
  w->cv.notify_all();
  return TRK_OK;
}

extern "C" int trk_host_worker_post_gcv_bidiag(trk_host_worker* w, const double* alpha, const double* beta, int k, double beta0,
                                               double m_eff, double x1, double x2, double xatol, int maxfun) {
  TRK_REQUIRE(w && alpha && beta && k >= 1, "trk_host_worker_post_gcv_bidiag: bad argument");
  TRK_REQUIRE(w->state.load() != 1, "trk_host_worker_post_gcv_bidiag: a job is still running (collect it first)");
  w->a.assign(alpha, alpha + k);
  w->b.assign(beta, beta + k);
  w->k = k;
  w->p[0] = beta0; w->p[1] = m_eff; w->p[2] = x1; w->p[3] = x2; w->p[4] = xatol;
  w->maxfun = maxfun;
  return host_worker_post(w, 0);
}

extern "C" int trk_host_worker_post_dp_bidiag(trk_host_worker* w, const double* alpha, const double* beta_sub, int k,
                                              const double* bproj, double target, double extra) {
  TRK_REQUIRE(w && alpha && beta_sub && bproj && k >= 1, "trk_host_worker_post_dp_bidiag: bad argument");
  TRK_REQUIRE(w->state.load() != 1, "trk_host_worker_post_dp_bidiag: a job is still running (collect it first)");
  w->a.assign(alpha, alpha + k);
  w->b.assign(beta_sub, beta_sub + k);
  w->c.assign(bproj, bproj + k + 1);
  w->k = k;
  w->p[0] = target; w->p[1] = extra;
  return host_worker_post(w, 1);
}

extern "C" int trk_host_worker_set_lapack(trk_host_worker* w, void* dgebrd, void* dormbr) {
  TRK_REQUIRE(w && dgebrd && dormbr, "trk_host_worker_set_lapack: NULL argument");
  TRK_REQUIRE(w->state.load() != 1, "trk_host_worker_set_lapack: a job is running");
  w->gebrd = dgebrd;
  w->ormbr = dormbr;
  return TRK_OK;
}

extern "C" int trk_host_worker_post_hess_gcv(trk_host_worker* w, const double* H, int64_t h_row_stride, int64_t h_col_stride, int k,
                                             double beta0, double m_eff, double x1, double x2, double xatol, int maxfun) {
  TRK_REQUIRE(w && H && k >= 1, "trk_host_worker_post_hess_gcv: bad argument");
  TRK_REQUIRE(w->gebrd && w->ormbr, "trk_host_worker_post_hess_gcv: trk_host_worker_set_lapack first");
  TRK_REQUIRE(w->state.load() != 1, "trk_host_worker_post_hess_gcv: a job is still running (collect it first)");
  const int n = k + 1;
  w->H.resize((size_t)n * k);
  for (int j = 0; j < k; ++j)
    for (int i = 0; i < n; ++i) w->H[(size_t)j * n + i] = H[i * h_row_stride + j * h_col_stride];
  w->k = k;
  w->p[0] = beta0; w->p[1] = m_eff; w->p[2] = x1; w->p[3] = x2; w->p[4] = xatol;
  w->maxfun = maxfun;
  return host_worker_post(w, 2);
}

extern "C" int trk_host_worker_post_hess_fixed(trk_host_worker* w, const double* H, int64_t h_row_stride, int64_t h_col_stride, int k,
                                               double beta0, double lam) {
  TRK_REQUIRE(w && H && k >= 1 && lam >= 0.0, "trk_host_worker_post_hess_fixed: bad argument");
  TRK_REQUIRE(w->gebrd && w->ormbr, "trk_host_worker_post_hess_fixed: trk_host_worker_set_lapack first");
  TRK_REQUIRE(w->state.load() != 1, "trk_host_worker_post_hess_fixed: a job is still running (collect it first)");
  const int n = k + 1;
  w->H.resize((size_t)n * k);
  for (int j = 0; j < k; ++j)
    for (int i = 0; i < n; ++i) w->H[(size_t)j * n + i] = H[i * h_row_stride + j * h_col_stride];
  w->k = k;
  w->p[0] = beta0; w->p[5] = lam;
  return host_worker_post(w, 4);
}

extern "C" int trk_host_worker_post_hess_dp(trk_host_worker* w, const double* H, int64_t h_row_stride, int64_t h_col_stride, int k,
                                            double beta0, const double* bproj, double target, double extra) {
  TRK_REQUIRE(w && H && bproj && k >= 1, "trk_host_worker_post_hess_dp: bad argument");
  TRK_REQUIRE(w->gebrd && w->ormbr, "trk_host_worker_post_hess_dp: trk_host_worker_set_lapack first");
  TRK_REQUIRE(w->state.load() != 1, "trk_host_worker_post_hess_dp: a job is still running (collect it first)");
  const int n = k + 1;
  w->H.resize((size_t)n * k);
  for (int j = 0; j < k; ++j)
    for (int i = 0; i < n; ++i) w->H[(size_t)j * n + i] = H[i * h_row_stride + j * h_col_stride];
  w->c.assign(bproj, bproj + n);
  w->k = k;
  w->p[0] = beta0; w->p[1] = target; w->p[2] = extra;
  return host_worker_post(w, 3);
}

extern "C" int trk_host_worker_collect_vec(trk_host_worker* w, double* lam_out, int* have_out, double* y, int k, double* resid_out) {
  TRK_REQUIRE(w && lam_out && have_out && y && resid_out, "trk_host_worker_collect_vec: NULL argument");
  TRK_REQUIRE(w->state.load() != 0, "trk_host_worker_collect_vec: nothing was posted");
  TRK_REQUIRE((w->kind == 2 || w->kind == 3 || w->kind == 4) && k == w->k, "trk_host_worker_collect_vec: the posted job is not a Hessenberg job of this size");
  const int rc = trk_host_worker_collect(w, lam_out, have_out);
  if (rc == TRK_OK && *have_out && w->y_valid) {
    for (int j = 0; j < k; ++j) y[j] = w->y[1 + j];
    *resid_out = w->resid;
  }
  return rc;
}

extern "C" int trk_host_worker_collect(trk_host_worker* w, double* lam_out, int* have_out) {
  TRK_REQUIRE(w && lam_out && have_out, "trk_host_worker_collect: NULL argument");
  TRK_REQUIRE(w->state.load() != 0, "trk_host_worker_collect: nothing was posted");
  spin_until(w, [](int st) { return st == 2; });
  if (w->state.load(std::memory_order_acquire) != 2) {
    std::unique_lock<std::mutex> lk(w->m);
    w->cv.wait(lk, [&] { return w->state.load(std::memory_order_acquire) == 2; });
  }
  *lam_out = w->lam;
  *have_out = w->have;
  const int rc = w->rc;
  w->state.store(0, std::memory_order_release);
  return rc;
}

// ------------------------------------------------------------------ GKS / MMGKS with regparam = 'gcv': the host's projected problem in one call
// GKS.py:54-74 / MMGKS.py:94-106 as the engine runs them on the host (the reference's DEFAULT regparam): from the Gram data
// G_A = (AV)^T AV, G_L = (LV)^T LV, c = (AV)^T b — R_A, R_L by Cholesky (the economic QRs' R up to row signs), Q_A^T b = R_A^-T c,
// GCV on (R_A, R_L) brought to (diag(s), I) through M = R_A R_L^-1 = U diag(s) W^T, the Tikhonov minimiser by the stacked
// least-squares problem.  The interpreter's version of this sequence (SciPy wrappers around LAPACK) was 250-300 us per iteration WITH
// THE DEVICE IDLE — the next basis vector needs x = V y.  GCV sees M only through s and U^T rhs: M is bidiagonalised (dgebrd), Q^T is
// applied to rhs (dormbr) and the bidiagonal's singular values are found with the left rotations applied to that ONE vector (dbdsqr,
// ncc = 1) — no singular vectors are formed (the dense SVD with both vector sets, what sla.svd computes, is ~5 x the work).  The caller
// hands the LAPACK routines (SciPy's, as plain C pointers).  *ok_out = 0: a factor failed (semi-definite Gram matrix, singular R_L, no
// convergence) — the caller's own branches take over.
namespace {
typedef void (*potrf_fn)(char*, int*, double*, int*, int*);
typedef void (*trtrs_fn)(char*, char*, char*, int*, int*, double*, int*, double*, int*, int*);
typedef void (*bdsqr_fn)(char*, int*, int*, int*, int*, double*, double*, double*, int*, double*, int*, double*, int*, double*, int*);
typedef void (*gelsy_fn)(int*, int*, int*, double*, int*, double*, int*, int*, double*, int*, double*, int*, int*);
}  // namespace

// lapack: {dpotrf, dtrtrs, dgebrd, dormbr, dbdsqr, dgelsy}
extern "C" int trk_host_gram_gcv(void* const* lapack, const double* GA, const double* GL, int ldg, const double* c_select,
                                 const double* c_solve, int k, double m_eff, double* lam_out, double* y_out, int* ok_out) {
  TRK_REQUIRE(lapack && GA && GL && c_select && c_solve && lam_out && y_out && ok_out && k >= 1 && ldg >= k, "trk_host_gram_gcv: bad argument");
  for (int i = 0; i < 6; ++i) TRK_REQUIRE(lapack[i], "trk_host_gram_gcv: six LAPACK routines (dpotrf, dtrtrs, dgebrd, dormbr, dbdsqr, dgelsy)");
  const potrf_fn dpotrf = (potrf_fn)lapack[0];
  const trtrs_fn dtrtrs = (trtrs_fn)lapack[1];
  const gebrd_fn dgebrd = (gebrd_fn)lapack[2];
  const ormbr_fn dormbr = (ormbr_fn)lapack[3];
  const bdsqr_fn dbdsqr = (bdsqr_fn)lapack[4];
  const gelsy_fn dgelsy = (gelsy_fn)lapack[5];
  *ok_out = 0;
  static thread_local std::vector<double> buf, wk;
  static thread_local std::vector<int> ibuf;
  const size_t kk = (size_t)k * k;
  buf.resize(6 * kk + 16 * (size_t)k + 64);
  ibuf.resize((size_t)k + 8);
  double* RA = buf.data();
  double* RL = RA + kk;
  double* X = RL + kk;           // R_L^-T R_A^T
  double* M = X + kk;            // M = X^T = R_A R_L^-1, overwritten by its bidiagonal form
  double* ST = M + kk;           // stacked [R_A; sqrt(lam) R_L], 2k x k
  double* sv = ST + 2 * kk;      // d of the bidiagonal form, then the singular values
  double* e = sv + k;
  double* tq = e + k;
  double* tp = tq + k;
  double* rs = tp + k;           // R_A^-T c_select, then Q^T of it, then U^T of it
  double* rb = rs + k;           // R_A^-T c_solve
  double* b2 = rb + k;           // 2k
  int n = k, one = 1, zero = 0, info = 0;
  char U_ = 'U', T_ = 'T', N_ = 'N', Q_ = 'Q', L_ = 'L';
  // column-major copies of the symmetrised Gram matrices (symmetric: the layout does not matter), upper Cholesky factors
  for (int j = 0; j < k; ++j)
    for (int i = 0; i < k; ++i) {
      RA[i + (size_t)j * k] = 0.5 * (GA[(size_t)i * ldg + j] + GA[(size_t)j * ldg + i]);
      RL[i + (size_t)j * k] = 0.5 * (GL[(size_t)i * ldg + j] + GL[(size_t)j * ldg + i]);
    }
  dpotrf(&U_, &n, RA, &n, &info);
  if (info != 0) return TRK_OK;
  dpotrf(&U_, &n, RL, &n, &info);
  if (info != 0) return TRK_OK;
  for (int j = 0; j < k; ++j)
    for (int i = j + 1; i < k; ++i) RA[i + (size_t)j * k] = RL[i + (size_t)j * k] = 0.0;      // (dpotrf leaves the other triangle as it was)
  double dmin = fabs(RL[0]), dmax = dmin;
  for (int i = 1; i < k; ++i) {
    const double d = fabs(RL[i + (size_t)i * k]);
    dmin = d < dmin ? d : dmin;
    dmax = d > dmax ? d : dmax;
  }
  if (dmin <= 1e-12 * dmax) return TRK_OK;                                                       // (gcv._diagonalise's test)
  for (int i = 0; i < k; ++i) {
    rs[i] = c_select[i];
    rb[i] = c_solve[i];
  }
  dtrtrs(&U_, &T_, &N_, &n, &one, RA, &n, rs, &n, &info);                                        // Q_A^T b = R_A^-T c
  if (info != 0) return TRK_OK;
  dtrtrs(&U_, &T_, &N_, &n, &one, RA, &n, rb, &n, &info);
  if (info != 0) return TRK_OK;
  for (int j = 0; j < k; ++j)
    for (int i = 0; i < k; ++i) X[i + (size_t)j * k] = RA[j + (size_t)i * k];                    // R_A^T
  dtrtrs(&U_, &T_, &N_, &n, &n, RL, &n, X, &n, &info);                                           // R_L^T X = R_A^T
  if (info != 0) return TRK_OK;
  for (int j = 0; j < k; ++j)
    for (int i = 0; i < k; ++i) M[i + (size_t)j * k] = X[j + (size_t)i * k];                     // M = R_A R_L^-1
  // s and U^T rhs without singular vectors: M = Q B P^T (dgebrd), w = Q^T rhs (dormbr), B = U_B diag(s) V_B^T with w <- U_B^T w (dbdsqr)
  int lwork = 64 * k + 64;
  if ((int)wk.size() < lwork) wk.resize(lwork);
  dgebrd(&n, &n, M, &n, sv, e, tq, tp, wk.data(), &lwork, &info);
  if (info != 0) return TRK_OK;
  dormbr(&Q_, &L_, &T_, &n, &one, &n, M, &n, tq, rs, &n, wk.data(), &lwork, &info);
  if (info != 0) return TRK_OK;
  {
    double dummy = 0.0;
    if ((int)wk.size() < 4 * k + 8) wk.resize(4 * k + 8);
    dbdsqr(&U_, &n, &zero, &zero, &one, sv, e, &dummy, &one, &dummy, &one, rs, &n, wk.data(), &info);
    if (info != 0) return TRK_OK;
  }
  for (int i = 0; i < k; ++i)
    if (!std::isfinite(sv[i]) || !std::isfinite(rs[i])) return TRK_OK;
  double lam = 0.0;
  if (int rc = trk_host_gcv_fminbound(sv, rs, k, m_eff, 1e-9, 1e2, 1e-12, 1000, &lam, nullptr, nullptr)) return rc;
  // y = argmin || R_A y - Q_A^T b ||^2 + lam || R_L y ||^2 = (G_A + lam G_L)^-1 c: by a Cholesky factorisation of the k x k sum — what the
  // device solves with a numeric lambda (trk_gram_tikhonov); R_A and R_L are Cholesky factors of the Gram matrices themselves, so the
  // stacked least-squares problem on them (the reference's lstsq, SciPy's gelsy with rcond = eps: 2.7 k^3 flops of pivoted QR, a third of
  // this call at k = 50) sees the same conditioning.  TRK_GRAM_GCV_LSTSQ=1, or a sum that is not positive definite: the stacked problem.
  static const bool stacked = getenv("TRK_GRAM_GCV_LSTSQ") != nullptr;
  if (!stacked) {
    for (int j = 0; j < k; ++j)
      for (int i = 0; i < k; ++i)
        ST[i + (size_t)j * k] = 0.5 * (GA[(size_t)i * ldg + j] + GA[(size_t)j * ldg + i]) +
                                lam * (0.5 * (GL[(size_t)i * ldg + j] + GL[(size_t)j * ldg + i]));
    dpotrf(&U_, &n, ST, &n, &info);
    if (info == 0) {
      for (int i = 0; i < k; ++i) b2[i] = c_solve[i];
      dtrtrs(&U_, &T_, &N_, &n, &one, ST, &n, b2, &n, &info);                                    // U^T z = c
      if (info == 0) dtrtrs(&U_, &N_, &N_, &n, &one, ST, &n, b2, &n, &info);                     // U y = z
      bool fin = info == 0;
      for (int i = 0; fin && i < k; ++i) fin = std::isfinite(b2[i]);
      if (fin) {
        for (int i = 0; i < k; ++i) y_out[i] = b2[i];
        *lam_out = lam;
        *ok_out = 1;
        return TRK_OK;
      }
    }
  }
  const int m2 = 2 * k;
  const double sl = sqrt(lam);
  for (int j = 0; j < k; ++j)
    for (int i = 0; i < k; ++i) {
      ST[i + (size_t)j * m2] = RA[i + (size_t)j * k];
      ST[k + i + (size_t)j * m2] = sl * RL[i + (size_t)j * k];
    }
  for (int i = 0; i < k; ++i) {
    b2[i] = rb[i];
    b2[k + i] = 0.0;
  }
  int* jpvt = ibuf.data();
  for (int i = 0; i < k; ++i) jpvt[i] = 0;
  double rcond = 2.220446049250313e-16, wq = 0.0;
  int rank = 0, mm = m2;
  lwork = -1;
  dgelsy(&mm, &n, &one, ST, &mm, b2, &mm, jpvt, &rcond, &rank, &wq, &lwork, &info);
  if (info != 0) return TRK_OK;
  lwork = (int)wq + 1;
  if ((int)wk.size() < lwork) wk.resize(lwork);
  dgelsy(&mm, &n, &one, ST, &mm, b2, &mm, jpvt, &rcond, &rank, wk.data(), &lwork, &info);
  if (info != 0) return TRK_OK;
  for (int i = 0; i < k; ++i) y_out[i] = b2[i];
  *lam_out = lam;
  *ok_out = 1;
  return TRK_OK;
}

// ------------------------------------------------------------------ Hybrid-LSQR with automatic lambda: the host's turn of an iteration
// Hybrid_LSQR.py:80-110 as the engine runs it (the search for lambda_k on the worker thread, the iterate of the step before formed when its
// lambda is collected): collect the search posted by the call before, post the search for step k_post (mode 0: gcv, 1: the discrepancy
// principle; B_k's entries and U^T b are the caller's host arrays, read before this returns), and — when k_done > 0 and x_out != NULL —
// solve the projected Tikhonov problem of step k_done with the collected lambda (trk_host_bidiag_tikhonov, y over alpha) and launch
// x_out = V_{k_done} y with y in the kernel's arguments (trk_gemv_n_hosty; ref != NULL: with the error partials).  Four library calls and
// three NumPy temporaries of the interpreter's loop in one.  *have_out = 0: nothing was collected (k_done == 0) or the search set no lambda.
extern "C" int trk_hlsqr_select(trk_host_worker* w, int mode, const double* alphas, const double* betas, int k_post, double beta0,
                                double m_eff_or_target, const double* bproj, double extra, int k_done, const float* V, int64_t ld,
                                int64_t n, float* x_out, const float* ref, double* err_partials, int err_cap, int* n_blocks,
                                double* lam_out, int* have_out, trk_stream stream) {
  TRK_REQUIRE(w && alphas && betas && lam_out && have_out && n_blocks && k_post >= 0 && k_done >= 0, "trk_hlsqr_select: bad argument");
  TRK_REQUIRE(mode == 0 || (mode == 1 && (bproj || k_post == 0)), "trk_hlsqr_select: mode 0 (gcv) or 1 (dp, with U^T b)");
  *have_out = 0;
  *n_blocks = 0;
  double lam = 0.0;
  int have = 0;
  if (k_done > 0) {
    if (int rc = trk_host_worker_collect(w, &lam, &have)) return rc;
    *lam_out = lam;
    *have_out = have;
  }
  if (k_post > 0) {
    if (mode == 0) {
      if (int rc = trk_host_worker_post_gcv_bidiag(w, alphas, betas, k_post, beta0, m_eff_or_target, 1e-9, 1e2, 1e-12, 1000)) return rc;
    } else if (int rc = trk_host_worker_post_dp_bidiag(w, alphas, betas, k_post, bproj, m_eff_or_target, extra)) return rc;
  }
  if (k_done > 0 && have && x_out) {
    TRK_REQUIRE(V && n >= 0 && ld >= n, "trk_hlsqr_select: x_out given without the basis");
    static thread_local std::vector<double> y;
    y.resize((size_t)k_done);
    if (int rc = trk_host_bidiag_tikhonov(alphas, betas, k_done, beta0, sqrt(lam), 1, y.data())) return rc;
    if (int rc = trk_gemv_n_hosty(V, ld, k_done, n, y.data(), x_out, ref, err_partials, err_cap, n_blocks, stream)) return rc;
  }
  return TRK_OK;
}

// ------------------------------------------------------------------ Hybrid-GMRES: the host side of one iteration in one call
// Hybrid_GMRES.py:46-80 with regparam = 'gcv' as this library runs it: the Arnoldi steps run ahead on the stream (each posts its column
// of H from its last kernel), iterate k's projected problem — bidiagonalisation of [beta0 e1 | H_k], the GCV search, the Tikhonov solve —
// is one job of a worker thread, and x_k = V_k y_k is launched with y_k in the kernel's arguments when the job is collected.  Nothing
// in the Arnoldi process waits for a projected solution, so the jobs of consecutive iterates run on SEVERAL workers side by side (a job
// is O(k^3): ~150 us at k = 60 against ~55 us of kernels per step) and are collected in order, `workers` iterations late.  What the
// interpreter did per iteration (seven library calls, three NumPy temporaries, ~70 us) is one call here.
struct trk_hgmres {
  trk_op* op;
  float* V;
  int64_t ld;
  int cap;                 // Arnoldi steps at most (H is (cap + 1) x cap)
  float* w;
  double *G, *W, *S;
  int ldg;
  trk_mailbox* mb;         // borrowed: 2 slots, a region of 2 cap + 4 doubles each
  double* mb_host;
  std::vector<trk_host_worker*> ws;   // borrowed
  double beta0;
  std::vector<double> H;   // column-major, column stride ldh
  int ldh;
  int k_enq, k_abs;        // steps enqueued / columns of H installed
  std::deque<int> posted;  // iterates (0-based) whose projected problems the workers hold, oldest first
  unsigned long long post_seq, collect_seq;
  std::vector<double> y;
  hipStream_t stream;
  double t_wait_step = 0, t_enqueue = 0, t_collect = 0, t_post = 0, t_launch = 0;     // host seconds by phase (trk_hgmres_stats)
  double fixed_lam = -1.0;                                                              // >= 0: the jobs solve with this lambda (no search)
  // the discrepancy principle: V_{k+1}^T b grows by one entry per step (taken by the step's normalising pass, posted with its scalars)
  const float* bvec = nullptr;
  std::vector<double> bproj;
  double dp_target = 0.0, dp_extra = 0.0;
};
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

extern "C" int trk_hgmres_fixed_lambda(trk_hgmres* g, double lam) {
  TRK_REQUIRE(g, "trk_hgmres_fixed_lambda: NULL handle");
  g->fixed_lam = lam;                          // (< 0: back to gcv)
  return TRK_OK;
}

extern "C" int trk_hgmres_dp(trk_hgmres* g, const float* bvec, double bproj0, double target, double extra, double** bproj_out) {
  TRK_REQUIRE(g && bvec && g->k_enq == 0, "trk_hgmres_dp: before trk_hgmres_start, with the right-hand side on the device");
  g->bvec = bvec;
  g->bproj.assign((size_t)g->cap + 2, 0.0);
  g->bproj[0] = bproj0;
  g->dp_target = target;
  g->dp_extra = extra;
  for (trk_host_worker* w : g->ws) w->dp_solves_zero = 1;      // (reset by trk_hgmres_destroy: the workers are borrowed)
  if (bproj_out) *bproj_out = g->bproj.data();
  return TRK_OK;
}

extern "C" int trk_hgmres_stats(trk_hgmres* g, double* seconds5) {
  TRK_REQUIRE(g && seconds5, "trk_hgmres_stats: NULL argument");
  seconds5[0] = g->t_wait_step; seconds5[1] = g->t_enqueue; seconds5[2] = g->t_collect; seconds5[3] = g->t_post; seconds5[4] = g->t_launch;
  return TRK_OK;
}

extern "C" int trk_hgmres_create(trk_op* op, float* V, int64_t ld, int capacity, float* w, double* G, int ldg, double* W, double* S,
                                 trk_mailbox* mb, trk_host_worker* const* workers, int n_workers, double beta0, trk_stream stream,
                                 trk_hgmres** out) {
  TRK_REQUIRE(op && V && w && G && W && S && out && mb && workers && capacity >= 1 && ldg >= capacity, "trk_hgmres_create: bad argument");
  TRK_REQUIRE(op->rows == op->cols && ld >= op->rows, "trk_hgmres_create: square operator, ld >= n");
  TRK_REQUIRE(n_workers >= 1 && n_workers <= 16, "trk_hgmres_create: 1..16 workers");
  for (int i = 0; i < n_workers; ++i)
    TRK_REQUIRE(workers[i] && workers[i]->gebrd && workers[i]->ormbr, "trk_hgmres_create: every worker with trk_host_worker_set_lapack done");
  auto* g = new trk_hgmres{};
  g->op = op; g->V = V; g->ld = ld; g->cap = capacity; g->w = w; g->G = G; g->W = W; g->S = S; g->ldg = ldg;
  g->beta0 = beta0; g->ldh = capacity + 2; g->stream = (hipStream_t)stream;
  g->H.assign((size_t)g->ldh * (size_t)(capacity + 1), 0.0);
  g->y.assign((size_t)capacity + 1, 0.0);
  g->ws.assign(workers, workers + n_workers);
  g->mb = mb;
  int rc = trk_mailbox_host(mb, &g->mb_host);
  if (!rc && (trk_mailbox_doubles(mb) < 2 * (2 * capacity + 4) || trk_mailbox_slots(mb) < 2))
    rc = fail(TRK_EINVAL, "trk_hgmres_create: the mailbox needs 2 slots and 2 (2 capacity + 4) doubles (a region per slot: two steps are in flight)");
  if (rc) {
    delete g;
    return rc;
  }
  *out = g;
  return TRK_OK;
}

// (the mailbox and the workers are the caller's: pinned memory and threads are pooled above the library — creating and freeing them
// per solve costs more than the iterations of a short solve)
extern "C" int trk_hgmres_destroy(trk_hgmres* g) {
  if (!g) return TRK_OK;
  if (g->k_enq > g->k_abs) (void)trk_mailbox_wait(g->mb, g->k_enq & 1);              // a posted step still writes to the mailbox
  while (!g->posted.empty()) {                                                        // jobs nobody collected: the workers go back idle
    double lam, r;
    int have;
    (void)trk_host_worker_collect_vec(g->ws[g->collect_seq % g->ws.size()], &lam, &have, g->y.data(), g->posted.front() + 1, &r);
    g->posted.pop_front();
    ++g->collect_seq;
  }
  for (trk_host_worker* w : g->ws) w->dp_solves_zero = 0;
  delete g;
  return TRK_OK;
}

extern "C" int trk_hgmres_hessenberg(trk_hgmres* g, double** H, int* ldh, int* columns) {
  TRK_REQUIRE(g && H && ldh && columns, "trk_hgmres_hessenberg: NULL argument");
  *H = g->H.data();
  *ldh = g->ldh;
  *columns = g->k_abs;
  return TRK_OK;
}

// the next Arnoldi step, its scalars posted to slot (k & 1)
static int hgmres_enqueue(trk_hgmres* g) {
  const int k = g->k_enq + 1;
  TRK_REQUIRE(k <= g->cap, "trk_hgmres: more steps than the basis was planned for");
  if (int rc = trk_arnoldi_step_post_dot(g->op, g->V, g->ld, k, g->w, g->G, g->ldg, g->W, g->S, g->mb, k & 1, 0, 1 + 2 * k,
                                         (k & 1) * (2 * g->cap + 4), g->bvec, 2 * g->cap + 2, g->stream))
    return rc;
  g->k_enq = k;
  return TRK_OK;
}
// Two steps ahead of the columns installed: step k + 1 needs nothing from the host, and enqueued only once step k's scalars had
// arrived it left the device idle for a launch latency per step (S is rewritten by step k + 1 only after step k's last kernel has
// posted it: stream order)
static int hgmres_keep_ahead(trk_hgmres* g) {
  while (g->k_enq < g->cap && g->k_enq < g->k_abs + 2)
    if (int rc = hgmres_enqueue(g)) return rc;
  return TRK_OK;
}

extern "C" int trk_hgmres_start(trk_hgmres* g) {
  TRK_REQUIRE(g && g->k_enq == 0, "trk_hgmres_start: once, first");
  return hgmres_keep_ahead(g);
}

/* One pass of the loop.  absorb: wait for the oldest posted step and install its column of H (column k = the count so far + 1);
 * enqueue_next: keep two steps on the stream ahead of the columns installed (up to `capacity`); x_done != NULL: collect the OLDEST posted job — *done_ii names its iterate, with its
 * lambda and the reference's relResidual in *done_lam / *done_resid — and launch x_done = V y (ref != NULL: with the block partials of
 * ||x - ref||^2 in err_partials, *done_blocks of them); post_job: hand iterate k - 1's projected problem (H_k, gcv) to the worker that
 * is free (with all of them busy, the caller collects in the same call).  The collect comes before the post, the launch after it. */
extern "C" int trk_hgmres_iter(trk_hgmres* g, int absorb, int enqueue_next, int post_job, float* x_done, const float* ref,
                               double* err_partials, int err_cap, int* done_ii, double* done_lam, double* done_resid, int* done_blocks) {
  TRK_REQUIRE(g && done_ii && done_lam && done_resid && done_blocks, "trk_hgmres_iter: NULL argument");
  *done_ii = -1;
  *done_blocks = 0;
  const size_t nw = g->ws.size();
  if (absorb) {
    TRK_REQUIRE(g->k_enq > g->k_abs, "trk_hgmres_iter: no step is pending");
    const int k = g->k_abs + 1;
    const double t0 = now_s();
    if (int rc = trk_mailbox_wait(g->mb, k & 1)) return rc;
    g->t_wait_step += now_s() - t0;
    const double* h = g->mb_host + (size_t)(k & 1) * (2 * g->cap + 4);   // S[0] = h_{k+1,k}^2, S[1..1+k) + S[1+k..1+2k) = the column above it
    double* col = g->H.data() + (size_t)(k - 1) * g->ldh;
    for (int i = 0; i < k; ++i) col[i] = h[1 + i] + h[1 + k + i];
    col[k] = sqrt(h[0]);
    if (g->bvec) g->bproj[k] = h[1 + 2 * k];
    g->k_abs = k;
  }
  double t1 = now_s();
  if (enqueue_next)
    if (int rc = hgmres_keep_ahead(g)) return rc;
  g->t_enqueue += now_s() - t1;
  t1 = now_s();
  int done = -1;
  if (x_done) {
    TRK_REQUIRE(!g->posted.empty(), "trk_hgmres_iter: x_done given but no job is posted");
    int have = 0;
    done = g->posted.front();
    if (int rc = trk_host_worker_collect_vec(g->ws[g->collect_seq % nw], done_lam, &have, g->y.data(), done + 1, done_resid)) return rc;
    g->posted.pop_front();
    ++g->collect_seq;
    if (!have || (g->bvec && !g->ws[(g->collect_seq - 1) % nw]->y_valid)) {   // the discrepancy principle's "unassigned": the caller's branch
      TRK_REQUIRE(g->bvec, "trk_hgmres_iter: the worker returned no lambda");
      *done_ii = done;
      *done_blocks = -1;
      done = -1;
    }
  }
  g->t_collect += now_s() - t1;
  t1 = now_s();
  if (post_job) {
    TRK_REQUIRE(g->k_abs >= 1 && g->posted.size() < nw, "trk_hgmres_iter: post_job needs a column of H and a free worker (collect first)");
    const int k = g->k_abs;
    if (g->bvec) {
      if (int rc = trk_host_worker_post_hess_dp(g->ws[g->post_seq % nw], g->H.data(), 1, g->ldh, k, g->beta0, g->bproj.data(), g->dp_target, g->dp_extra))
        return rc;
    } else if (g->fixed_lam >= 0.0) {
      if (int rc = trk_host_worker_post_hess_fixed(g->ws[g->post_seq % nw], g->H.data(), 1, g->ldh, k, g->beta0, g->fixed_lam)) return rc;
    } else if (int rc = trk_host_worker_post_hess_gcv(g->ws[g->post_seq % nw], g->H.data(), 1, g->ldh, k, g->beta0, (double)k, 1e-9, 1e2, 1e-12, 1000))
      return rc;
    g->posted.push_back(k - 1);
    ++g->post_seq;
  }
  g->t_post += now_s() - t1;
  t1 = now_s();
  if (done >= 0) {
    if (int rc = trk_gemv_n_hosty(g->V, g->ld, done + 1, g->op->rows, g->y.data(), x_done, ref, err_partials, err_cap, done_blocks,
                                  g->stream))
      return rc;
    *done_ii = done;
  }
  g->t_launch += now_s() - t1;
  return TRK_OK;
}
