// fanbeam2d.hip — fan-beam (flat detector) line projector and its matched adjoint.
//
// Replaces astra.OpTomo over create_proj_geom('fanflat', det_pitch, p, theta, SOD, ODD) + create_projector('line_fanflat')
// of trips/test_problems/Tomography.py:53-88 (p = int(sqrt(2) nx) detector pixels, SOD = 3 nx, ODD = nx, pitch = 4/3).
// 'line' = the ray from the source to a detector-pixel centre weighs every image pixel by the LENGTH of their
// intersection (Siddon).  PARITY UNPINNED like the parallel-beam operator (astra-toolbox absent): recorded convention —
// pixel (r,c) is the square [c - N/2, c+1 - N/2] x [N/2 - r - 1, N/2 - r]; at angle t the source sits at
// (SOD sin t, -SOD cos t), the detector centre at (-ODD sin t, ODD cos t), detector axis (cos t, sin t); detector pixel d
// is centred at (d - (p-1)/2) * pitch along it.  Sinogram (n_ang, n_det) row-major.
//
// Forward: one thread per ray, Amanatides-Woo grid traversal (<= 2N steps), lengths from successive crossings.
// Adjoint: gather, one thread per pixel: for every angle the pixel's four corners are projected onto the detector, every
// detector whose centre ray falls inside that interval is clipped against the pixel square (slab method) and contributes
// length * sinogram value.  Same matrix as the forward traversal up to fp32 rounding of the lengths (no atomics).
#include "trk_internal.h"

#include <cmath>
#include <vector>

using namespace trk;

namespace {

struct FanAngle {
  float sx, sy;      // source
  float d0x, d0y;    // centre of detector pixel 0
  float ux, uy;      // detector pixel pitch vector
  float nx, ny;      // unit normal source -> detector centre
};

struct FanImpl {
  int N, nd, na;
  float dsd;         // source-detector distance
  float pitch;
  FanAngle* ang_dev;
};

__global__ __launch_bounds__(256) void k_fan_fwd(const float* __restrict__ img, int64_t ld_img, float* __restrict__ sino,
                                                 int64_t ld_sino, int N, int nd, int na, const FanAngle* __restrict__ ang) {
  const int64_t ray = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (ray >= (int64_t)na * nd) return;
  const int a = (int)(ray / nd), d = (int)(ray - (int64_t)a * nd);
  const FanAngle g = ang[a];
  img += (int64_t)blockIdx.y * ld_img;
  const float half = 0.5f * (float)N;
  const float ex = g.d0x + (float)d * g.ux, ey = g.d0y + (float)d * g.uy;   // detector pixel centre
  const float dx = ex - g.sx, dy = ey - g.sy;
  const float L = sqrtf(dx * dx + dy * dy);
  // clip the segment source->detector (t in [0,1]) against the image square
  float t0 = 0.f, t1 = 1.f;
  if (fabsf(dx) > 1e-12f) {
    const float ta = (-half - g.sx) / dx, tb = (half - g.sx) / dx;
    t0 = fmaxf(t0, fminf(ta, tb));
    t1 = fminf(t1, fmaxf(ta, tb));
  } else if (g.sx <= -half || g.sx >= half) t1 = -1.f;
  if (fabsf(dy) > 1e-12f) {
    const float ta = (-half - g.sy) / dy, tb = (half - g.sy) / dy;
    t0 = fmaxf(t0, fminf(ta, tb));
    t1 = fminf(t1, fmaxf(ta, tb));
  } else if (g.sy <= -half || g.sy >= half) t1 = -1.f;
  float acc = 0.f;
  if (t1 > t0) {
    // first pixel: from the midpoint of a tiny first step
    const float tm = t0 + 1e-4f * (t1 - t0);
    int ix = (int)floorf(g.sx + tm * dx + half), iy = (int)floorf(g.sy + tm * dy + half);
    ix = ix < 0 ? 0 : (ix > N - 1 ? N - 1 : ix);
    iy = iy < 0 ? 0 : (iy > N - 1 ? N - 1 : iy);
    const int stepx = dx > 0.f ? 1 : -1, stepy = dy > 0.f ? 1 : -1;
    const float inf = 3.0e38f;
    const float dtx = fabsf(dx) > 1e-12f ? fabsf(1.0f / dx) : inf, dty = fabsf(dy) > 1e-12f ? fabsf(1.0f / dy) : inf;
    float tnx = fabsf(dx) > 1e-12f ? ((float)(ix + (stepx > 0 ? 1 : 0)) - half - g.sx) / dx : inf;
    float tny = fabsf(dy) > 1e-12f ? ((float)(iy + (stepy > 0 ? 1 : 0)) - half - g.sy) / dy : inf;
    float t = t0;
    for (int it = 0; it < 2 * N + 2; ++it) {
      const float tn = fminf(fminf(tnx, tny), t1);
      const float len = fmaxf(tn - t, 0.f) * L;
      acc = fmaf(len, img[(int64_t)(N - 1 - iy) * N + ix], acc);
      if (tn >= t1) break;
      if (tnx <= tny) {
        ix += stepx;
        tnx += dtx;
      } else {
        iy += stepy;
        tny += dty;
      }
      t = tn;
      if ((unsigned)ix >= (unsigned)N || (unsigned)iy >= (unsigned)N) break;
    }
  }
  sino[(int64_t)blockIdx.y * ld_sino + ray] = acc;
}

__global__ __launch_bounds__(256) void k_fan_adj(const float* __restrict__ sino, int64_t ld_sino, float* __restrict__ img,
                                                 int64_t ld_img, int N, int nd, int na, float dsd, float inv_pitch,
                                                 const FanAngle* __restrict__ ang) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)N * N) return;
  const int r = (int)(idx / N), c = (int)(idx - (int64_t)r * N);
  const float half = 0.5f * (float)N;
  const float x0 = (float)c - half, x1 = x0 + 1.0f;
  const float y1 = half - (float)r, y0 = y1 - 1.0f;
  const float* __restrict__ S = sino + (int64_t)blockIdx.y * ld_sino;
  float acc = 0.f;
  for (int a = 0; a < na; ++a) {
    const FanAngle g = ang[a];
    // detector index coordinate of the projection of a point P: u = ((S + s (P-S)) - D0) . u_hat / pitch
    const float uhx = g.ux * inv_pitch, uhy = g.uy * inv_pitch;
    float umin = 3.0e38f, umax = -3.0e38f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float px = (k & 1) ? x1 : x0, py = (k & 2) ? y1 : y0;
      const float vx = px - g.sx, vy = py - g.sy;
      const float s = dsd / (vx * g.nx + vy * g.ny);
      const float hx = g.sx + s * vx - g.d0x, hy = g.sy + s * vy - g.d0y;
      const float u = (hx * uhx + hy * uhy) * inv_pitch;
      umin = fminf(umin, u);
      umax = fmaxf(umax, u);
    }
    int dlo = (int)ceilf(umin - 1e-3f), dhi = (int)floorf(umax + 1e-3f);
    dlo = dlo < 0 ? 0 : dlo;
    dhi = dhi > nd - 1 ? nd - 1 : dhi;
    const float* __restrict__ Sa = S + (int64_t)a * nd;
    for (int d = dlo; d <= dhi; ++d) {
      const float ex = g.d0x + (float)d * g.ux, ey = g.d0y + (float)d * g.uy;
      const float dx = ex - g.sx, dy = ey - g.sy;
      const float L = sqrtf(dx * dx + dy * dy);
      float t0 = 0.f, t1 = 1.f;
      if (fabsf(dx) > 1e-12f) {
        const float ta = (x0 - g.sx) / dx, tb = (x1 - g.sx) / dx;
        t0 = fmaxf(t0, fminf(ta, tb));
        t1 = fminf(t1, fmaxf(ta, tb));
      } else if (g.sx < x0 || g.sx > x1) t1 = -1.f;
      if (fabsf(dy) > 1e-12f) {
        const float ta = (y0 - g.sy) / dy, tb = (y1 - g.sy) / dy;
        t0 = fmaxf(t0, fminf(ta, tb));
        t1 = fminf(t1, fmaxf(ta, tb));
      } else if (g.sy < y0 || g.sy > y1) t1 = -1.f;
      if (t1 > t0) acc = fmaf((t1 - t0) * L, Sa[d], acc);
    }
  }
  img[(int64_t)blockIdx.y * ld_img + idx] = acc;
}

int fan_apply(trk_op* op, int tr, const float* x, int64_t ldx, float* y, int64_t ldy, int batch, double* sumsq,
              hipStream_t s) {
  auto* im = static_cast<FanImpl*>(op->impl);
  TimerScope tm(op->timer, op->timer_which, tr, s);
  if (!tr) {
    dim3 grid(ceil_div((int64_t)im->na * im->nd, 256), batch);
    hipLaunchKernelGGL(k_fan_fwd, grid, dim3(256), 0, s, x, ldx, y, ldy, im->N, im->nd, im->na, im->ang_dev);
  } else {
    dim3 grid(ceil_div((int64_t)im->N * im->N, 256), batch);
    hipLaunchKernelGGL(k_fan_adj, grid, dim3(256), 0, s, x, ldx, y, ldy, im->N, im->nd, im->na, im->dsd, 1.0f / im->pitch, im->ang_dev);
  }
  tm.stop();
  TRK_LAUNCH_CHECK();
  if (sumsq) {
    const int64_t nout = tr ? (int64_t)im->N * im->N : (int64_t)im->na * im->nd;
    if (batch == 1 || ldy == nout) return trk_nrm2sq(y, nout * batch, sumsq, (trk_stream)s);
    return fail(TRK_EUNSUPPORTED, "fanbeam: fused sum of squares needs contiguous batch outputs");
  }
  return TRK_OK;
}

void fan_destroy(trk_op* op) {
  auto* im = static_cast<FanImpl*>(op->impl);
  if (im->ang_dev) (void)hipFree(im->ang_dev);
  delete im;
}

}  // namespace

extern "C" int trk_fanbeam2d_create(int N, int n_det, double det_pitch, double sod, double odd, const double* angles,
                                    int n_ang, trk_op** out) {
  TRK_REQUIRE(out && angles, "trk_fanbeam2d_create: NULL argument");
  TRK_REQUIRE(N >= 1 && n_det >= 1 && n_ang >= 1 && det_pitch > 0 && sod > 0 && odd >= 0, "trk_fanbeam2d_create: bad geometry");
  TRK_REQUIRE(sod > 0.7072 * N, "trk_fanbeam2d_create: the source must lie outside the image square");
  std::vector<FanAngle> h(n_ang);
  for (int a = 0; a < n_ang; ++a) {
    const double ct = std::cos(angles[a]), st = std::sin(angles[a]);
    FanAngle g;
    g.sx = (float)(sod * st);
    g.sy = (float)(-sod * ct);
    const double dcx = -odd * st, dcy = odd * ct;
    g.ux = (float)(det_pitch * ct);
    g.uy = (float)(det_pitch * st);
    g.d0x = (float)(dcx - 0.5 * (n_det - 1) * det_pitch * ct);
    g.d0y = (float)(dcy - 0.5 * (n_det - 1) * det_pitch * st);
    g.nx = (float)(-st);
    g.ny = (float)(ct);
    h[a] = g;
  }
  auto* im = new FanImpl{N, n_det, n_ang, (float)(sod + odd), (float)det_pitch, nullptr};
  hipError_t e = hipMalloc(&im->ang_dev, sizeof(FanAngle) * n_ang);
  if (e == hipSuccess) e = hipMemcpy(im->ang_dev, h.data(), sizeof(FanAngle) * n_ang, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    trk_op tmp{7, 0, 0, im, nullptr, nullptr, nullptr, 0};
    fan_destroy(&tmp);
    return fail(TRK_EHIP, "trk_fanbeam2d_create: %s", hipGetErrorString(e));
  }
  *out = new trk_op{7, (int64_t)n_ang * n_det, (int64_t)N * N, im, fan_apply, fan_destroy, nullptr, 0};
  return TRK_OK;
}
