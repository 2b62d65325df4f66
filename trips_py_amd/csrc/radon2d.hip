// radon2d.hip — parallel-beam Radon transform (Joseph / linear-interpolation projector) and its matched adjoint.
//
// Replaces astra.OpTomo over create_proj_geom('parallel', 1, N, theta) + create_projector('linear', ...) and the
// 1/N scaling of trips/utilities/io.py:392-399.  PARITY UNPINNED: astra-toolbox is not installable in the build image
// and the reference holds no reproducible output at that boundary; the convention below is the one recorded in
// oracle/cpu_ref.py (Radon2D) and is pinned by the adjoint identity and analytic line integrals.
//
// Geometry: pixel (i,j) centre (x,y) = (j - h, h - i), h = (N-1)/2; detector bin d at s = d - (n_det-1)/2 along
// (cos t, sin t); rays run along (sin t, -cos t).  Per angle one of two marching modes:
//   mode 0 (|cos| >= |sin|): march image rows  tt = i, coordinate q = column:  q = s/cos + (h - h tan) + i tan
//   mode 1 (otherwise)     : march image cols  tt = j, coordinate q = row   :  q = -s/sin + (h - h cot) + j cot
// Both are   q(d, tt) = fmaf(tt, dq, fmaf(s_d, inv, k0))   with taps at floor(q), floor(q)+1, weights (1-f), f, times
// wgt = scale/|cos| (or /|sin|).  Forward and adjoint evaluate q with the SAME float expression, so the adjoint uses
// bit-identical matrix entries (exact transpose; only the summation order differs).
//
// Forward: mode-1 angles read a transposed copy of the image so that both modes read ROWS: the 64 adjacent detectors of
// a wave touch 64..90 contiguous floats per marching step (coalesced, L1/L2-resident bands).  A workgroup is 64
// detectors x 4 row-quarters of one angle; the quarters are combined through LDS.
// Adjoint: gather form, one thread per pixel, no atomics: for each angle the <= 3 detectors whose ray passes within one
// pixel are found from the inverse of q and re-evaluated exactly.
//
// Roofline note: algorithmic bytes are only 4(N^2 + n_ang n_det) against 2 N n_det n_ang taps, so this
// operator is bound by L1/L2 gather + fp32 ALU, not HBM; bench.py reports taps/s next to GB/s.
#include "trk_internal.h"

#include <cmath>
#include <vector>

using namespace trk;

namespace {

struct AngleParam {
  float inv, dq, k0, wgt;
  int mode;
};

struct RadonImpl {
  int N, nd, na;   // na = angles PER FRAME
  int nt;          // time frames sharing one launch (block-diagonal dynamic operator, io.py:391-420); 1 = static
  AngleParam* ang_dev;   // nt*na entries, frame-major
  float* xT;  // nt*N*N transposed images (forward, mode-1 angles); owned by the handle (non-reentrant across streams)
  int n_mode1;
};

// ---------------------------------------------------------------------------------------- transpose (LDS tile 32x33)
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ in, float* __restrict__ out, int N) {
  __shared__ float tile[32][33];
  in += (int64_t)blockIdx.z * N * N;
  out += (int64_t)blockIdx.z * N * N;
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int i = by + r, j = bx + tx;
    if (i < N && j < N) tile[r][tx] = in[(int64_t)i * N + j];
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = bx + r, j = by + tx;  // out[i][j] = in[j][i]
    if (i < N && j < N) out[(int64_t)i * N + j] = tile[tx][r];
  }
}

// ---------------------------------------------------------------------------------------- forward
// grid = (ceil(nd/64), n_ang_total, batch) ; block = 256 = 64 detectors x 4 marching quarters.
// Both taps of a step come from ONE 8-byte buffer load at (row, floor(q)); out-of-range taps get weight 0 (the buffer
// range check returns 0 for the two addresses that fall outside the image allocation).
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_radon_fwd(const float* __restrict__ img, const float* __restrict__ imgT,
                                                   int64_t ld_img, float* __restrict__ sino, int64_t ld_sino, int N,
                                                   int nd, const AngleParam* __restrict__ ang, int na_per_frame) {
  __shared__ float part[4][64];
  const int a = blockIdx.y;                       // global angle index (frame-major)
  const AngleParam p = ang[a];
  const int frame = a / na_per_frame;
  const float* __restrict__ I = (p.mode ? imgT : img) + (int64_t)blockIdx.z * ld_img + (int64_t)frame * N * N;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)I, 0, (unsigned)N * (unsigned)N * 4u, 0x00020000);
  const int lane = threadIdx.x & 63, q4 = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + lane;
  const float s = (float)d - 0.5f * (float)(nd - 1);
  const float base = fmaf(s, p.inv, p.k0);
  const int t0 = (int)(((int64_t)N * q4) >> 2), t1 = (int)(((int64_t)N * (q4 + 1)) >> 2);
  double total = 0.0;
  if (d < nd) {
    for (int tb = t0; tb < t1; tb += 64) {
      const int te = (tb + 64 < t1) ? tb + 64 : t1;
      float acc = 0.f;
#pragma unroll 8
      for (int tt = tb; tt < te; ++tt) {
        const float q = fmaf((float)tt, p.dq, base);
        const float qf = floorf(q);
        const float f = q - qf;
        const int c = (int)qf;
        // c == -1: only the right tap (column 0) is inside; start the 8-byte load at column 0 instead (an access that
        // STARTS below the buffer is dropped whole by the range check — measured on gfx950 — while one that runs off
        // the end returns its in-range dword)
        const bool neg1 = (c == -1);
        const int cl = neg1 ? 0 : c;
        const f2v v = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (tt * N + cl) * 4, 0, 0));
        const float w0 = neg1 ? f : (((unsigned)c < (unsigned)N) ? 1.0f - f : 0.f);
        const float w1 = neg1 ? 0.f : (((unsigned)(c + 1) < (unsigned)N) ? f : 0.f);
        acc = fmaf(w0, v[0], acc);
        acc = fmaf(w1, v[1], acc);
      }
      total += (double)acc;
    }
  }
  part[q4][lane] = (float)total;
  __syncthreads();
  if (q4 == 0 && d < nd) {
    const float v = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    sino[(int64_t)blockIdx.z * ld_sino + (int64_t)a * nd + d] = p.wgt * v;
  }
}

// ---------------------------------------------------------------------------------------- adjoint (gather)
// One thread per pixel; grid = (ceil(N*N/256), n_frames, batch).  The forward weights of ray d on its two taps are
// (1-f, f) with f = q - floor(q), i.e. hat(q - col) = max(0, 1 - |q - col|) on pixel `col`; the gather evaluates exactly
// that for the three detectors nearest to the pixel's inverse image d* (every ray with |q - col| < 1 is among them,
// because |dq/dd| = 1/|cos| >= 1), with q computed by the SAME float expression as the forward kernel.  The weights are
// bit-identical to the forward ones (the subtractions involved are exact), so this is the exact transpose.
__global__ __launch_bounds__(256) void k_radon_adj(const float* __restrict__ sino, int64_t ld_sino,
                                                   float* __restrict__ img, int64_t ld_img, int N, int nd, int na,
                                                   const AngleParam* __restrict__ ang) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)N * N) return;
  const int i = (int)(idx / N), j = (int)(idx - (int64_t)i * N);
  const int frame = blockIdx.y;                   // blockIdx.y = time frame, blockIdx.z = batch vector
  const float* __restrict__ S = sino + (int64_t)blockIdx.z * ld_sino + (int64_t)frame * na * nd;
  ang += (int64_t)frame * na;
  const float sdh = 0.5f * (float)(nd - 1);
  const float fi = (float)i, fj = (float)j;
  float acc = 0.f;
#pragma unroll 2
  for (int a = 0; a < na; ++a) {
    const AngleParam p = ang[a];
    const float ftt = p.mode ? fj : fi;       // marching index
    const float fcol = p.mode ? fi : fj;      // interpolated coordinate this pixel sits on
    const float off = fmaf(ftt, p.dq, p.k0);
    const float dstar = (fcol - off) / p.inv + sdh;
    const int d0 = (int)rintf(dstar);
    const float* __restrict__ Sa = S + (int64_t)a * nd;
    float sum = 0.f;
#pragma unroll
    for (int e = -1; e <= 1; ++e) {
      const int d = d0 + e;
      const float sd = (float)d - sdh;
      const float q = fmaf(ftt, p.dq, fmaf(sd, p.inv, p.k0));
      const float wgt = fmaxf(1.0f - fabsf(q - fcol), 0.f);
      const float sv = ((unsigned)d < (unsigned)nd) ? Sa[(unsigned)d < (unsigned)nd ? d : 0] : 0.f;
      sum = fmaf(wgt, sv, sum);
    }
    acc = fmaf(p.wgt, sum, acc);
  }
  img[(int64_t)blockIdx.z * ld_img + (int64_t)frame * N * N + idx] = acc;
}

int radon_apply(trk_op* op, int tr, const float* x, int64_t ldx, float* y, int64_t ldy, int batch, double* sumsq,
                hipStream_t s) {
  auto* im = static_cast<RadonImpl*>(op->impl);
  const int N = im->N, nd = im->nd, na = im->na, nt = im->nt;
  TimerScope tm(op->timer, op->timer_which, tr, s);
  if (!tr) {
    for (int b = 0; b < batch; ++b) {  // the transposed copy is per vector
      const float* xb = x + (int64_t)b * ldx;
      if (im->n_mode1 > 0) {
        dim3 g(ceil_div(N, 32), ceil_div(N, 32), nt);
        hipLaunchKernelGGL(k_transpose, g, dim3(256), 0, s, xb, im->xT, N);
      }
      dim3 grid(ceil_div(nd, 64), nt * na, 1);
      hipLaunchKernelGGL(k_radon_fwd, grid, dim3(256), 0, s, xb, im->xT, (int64_t)0, y + (int64_t)b * ldy, (int64_t)0, N, nd, im->ang_dev, na);
      TRK_LAUNCH_CHECK();
    }
  } else {
    dim3 grid(ceil_div((int64_t)N * N, 256), nt, batch);
    hipLaunchKernelGGL(k_radon_adj, grid, dim3(256), 0, s, x, ldx, y, ldy, N, nd, na, im->ang_dev);
    TRK_LAUNCH_CHECK();
  }
  tm.stop();
  if (sumsq) {
    // the output is small next to the tap work: one extra streaming pass for the fused norm
    const int64_t nout = tr ? (int64_t)nt * N * N : (int64_t)nt * na * nd;
    if (batch == 1 || ldy == nout) return trk_nrm2sq(y, nout * batch, sumsq, (trk_stream)s);
    return fail(TRK_EUNSUPPORTED, "radon: fused sum of squares needs contiguous batch outputs");
  }
  return TRK_OK;
}

void radon_destroy(trk_op* op) {
  auto* im = static_cast<RadonImpl*>(op->impl);
  if (im->ang_dev) (void)hipFree(im->ang_dev);
  if (im->xT) (void)hipFree(im->xT);
  delete im;
}

}  // namespace

static int radon_create_impl(int N, int n_det, const double* angles, int nt, int na, double scale, trk_op** out) {
  const int n_ang = nt * na;
  std::vector<AngleParam> h(n_ang);
  const double half = 0.5 * (N - 1);
  int n1 = 0;
  for (int a = 0; a < n_ang; ++a) {
    const double ct = std::cos(angles[a]), st = std::sin(angles[a]);
    AngleParam p;
    if (std::fabs(ct) >= std::fabs(st)) {
      p.mode = 0;
      p.inv = (float)(1.0 / ct);
      p.dq = (float)(st / ct);
      p.k0 = (float)(half - half * st / ct);
      p.wgt = (float)(scale / std::fabs(ct));
    } else {
      p.mode = 1;
      p.inv = (float)(-1.0 / st);
      p.dq = (float)(ct / st);
      p.k0 = (float)(half - half * ct / st);
      p.wgt = (float)(scale / std::fabs(st));
      ++n1;
    }
    h[a] = p;
  }
  auto* im = new RadonImpl{N, n_det, na, nt, nullptr, nullptr, n1};
  hipError_t e = hipMalloc(&im->ang_dev, sizeof(AngleParam) * n_ang);
  if (e == hipSuccess) e = hipMemcpy(im->ang_dev, h.data(), sizeof(AngleParam) * n_ang, hipMemcpyHostToDevice);
  if (e == hipSuccess && n1 > 0) e = hipMalloc(&im->xT, sizeof(float) * (size_t)nt * N * N);
  if (e != hipSuccess) {
    trk_op tmp{2, 0, 0, im, nullptr, nullptr, nullptr, 0};
    radon_destroy(&tmp);
    return fail(TRK_EHIP, "trk_radon2d_create: %s", hipGetErrorString(e));
  }
  *out = new trk_op{2, (int64_t)n_ang * n_det, (int64_t)nt * N * N, im, radon_apply, radon_destroy, nullptr, 0};
  return TRK_OK;
}

extern "C" int trk_radon2d_create(int N, int n_det, const double* angles, int n_ang, double scale, trk_op** out) {
  TRK_REQUIRE(out && angles, "trk_radon2d_create: NULL argument");
  TRK_REQUIRE(N >= 1 && n_det >= 1 && n_ang >= 1, "trk_radon2d_create: sizes must be >= 1");
  return radon_create_impl(N, n_det, angles, 1, n_ang, scale, out);
}

extern "C" int trk_radon2d_dynamic_create(int N, int n_det, const double* angles, int n_frames, int n_ang_per_frame,
                                          double scale, trk_op** out) {
  TRK_REQUIRE(out && angles, "trk_radon2d_dynamic_create: NULL argument");
  TRK_REQUIRE(N >= 1 && n_det >= 1 && n_frames >= 1 && n_ang_per_frame >= 1, "trk_radon2d_dynamic_create: sizes must be >= 1");
  return radon_create_impl(N, n_det, angles, n_frames, n_ang_per_frame, scale, out);
}
