// cgls_tiled.hip — CGLS on small blur problems in TWO launches per iteration.
//
// At 512^2 the four-launch iteration of cgls_loop.hip retires in 22 us for 12 MB of traffic: every kernel is a dependent
// dispatch of 4-7 us whatever it does (trips/solvers/CGLS.py:56-80 has two global reductions per iteration, and each forces
// a boundary; profiles: DESIGN.md §4.1b).  Here a workgroup owns a 32 x 32 pixel TILE of the image and redoes, in LDS, what
// it needs of its neighbours' work instead of waiting for them:
//
//   K_A(k)   p_k = t + (gamma_{k-1}/gamma_{k-2}) p_{k-1} on the tile AND its 4-pixel halo (reads t, p_{k-1} there; writes
//            p_k on the tile, into the other half of a ping-pong pair);  w = A p_k on the tile is NOT stored — only
//            ||w||^2 is needed before the next boundary (block partials; the consumer adds them up).
//   K_B(k)   alpha = gamma_{k-1}/||w||^2;  w = A p_k AGAIN, on tile + halo (p_k on tile + 8 is complete: K_A has ended);
//            r_k = r_{k-1} - alpha w on tile + halo (r ping-pong);  t = A^T r_k on the tile, ||t||^2 partials;
//            x_k = x_{k-1} + alpha p_k on the tile with the three norms of CGLS.py:76-80 as partials.
//
// The blur is cheap next to a dispatch (a separable 9 x 9 on 48 x 48 floats from LDS: ~1 us), so recomputing it on halos costs
// less than the two launches and ~26 MB of round trips it saves.  Reflective boundaries (scipy.ndimage 'reflect',
// Deblurring2D.py:70): the operand of a blur is loaded through reflected indices; a vector that is itself an OUTPUT (r on the
// halo outside the image) takes the value of its mirror pixel, which lies inside the same tile's halo region.
// Separable PSFs up to 9 x 9 (what Deblurring2D.Gauss makes), images of at least 16 x 16, at most 1024 tiles (the norm-partial
// rows of the callers); everything else keeps the streaming kernels.  Same scalar layout and ping-pong buffers as
// trk_cgls_iterate_fused, so one host driver serves both.
#include "trk_internal.h"

#include <cstdlib>
#include <cstring>

using namespace trk;

namespace {

constexpr int CT = 32, CH = 4;
constexpr int E1 = CT + 2 * CH;      // 40: tile + halo
constexpr int E2 = CT + 4 * CH;      // 48: tile + two halos
constexpr int NT = 256;

struct TiledGeom {
  int nx, ny, tiles_x;
  float rwf[9], cwf[9], rwt[9], cwt[9];     // centred 9-tap row / column weights: forward and "transpose" (flipped PSF)
};

__device__ __forceinline__ int refl(int i, int n) {
  if ((unsigned)i < (unsigned)n) return i;
  const int p = 2 * n;
  i %= p;
  if (i < 0) i += p;
  return (i >= n) ? (p - 1 - i) : i;
}

// out[R][C] (row stride so) = separable 9 x 9 correlation of in[R + 8][C + 8] (row stride si) with row weights rw, column
// weights cw; tmp holds (R + 8) x C floats (row stride C + 1).  All 256 threads; ends with the result visible (barrier).
template <int R, int C>
__device__ __forceinline__ void blur_lds(const float* __restrict__ in, int si, float* __restrict__ tmp, float* __restrict__ out,
                                         int so, const float* rw, const float* cw) {
  constexpr int ST = C + 1;
  for (int idx = threadIdx.x; idx < (R + 8) * C; idx += NT) {
    const int r = idx / C, c = idx - r * C;
    const float* p = in + r * si + c;
    float a = 0.f;
#pragma unroll
    for (int o = 0; o < 9; ++o) a = fmaf(rw[o], p[o], a);
    tmp[r * ST + c] = a;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < R * C; idx += NT) {
    const int r = idx / C, c = idx - r * C;
    const float* p = tmp + r * ST + c;
    float a = 0.f;
#pragma unroll
    for (int o = 0; o < 9; ++o) a = fmaf(cw[o], p[o * ST], a);
    out[r * so + c] = a;
  }
  __syncthreads();
}

__global__ __launch_bounds__(NT) void k_cgls_tile_a(TiledGeom g, const float* __restrict__ t, const float* __restrict__ p_old,
                                                    float* __restrict__ p_new, ScalarSrc gam, const double* __restrict__ gprev,
                                                    double* __restrict__ gpub, int first, double* __restrict__ PD) {
  __shared__ float P1[E1 * (E1 + 1)];
  __shared__ float tmp[E1 * (CT + 1)];
  __shared__ float W[CT * (CT + 1)];
  __shared__ double red[NT / 64];
  __shared__ float bc;
  const int ty = blockIdx.x / g.tiles_x, tx = blockIdx.x - ty * g.tiles_x;
  const int i0 = ty * CT, j0 = tx * CT;
  if (threadIdx.x < 64) {                                   // beta = gamma_{k-1} / gamma_{k-2}; block 0 publishes gamma_{k-1}
    const double gk = scalar_from_wave(gam, threadIdx.x);
    if (threadIdx.x == 0) {
      bc = first ? 0.f : (float)(gk / *gprev);
      if (blockIdx.x == 0) *gpub = gk;
    }
  }
  __syncthreads();
  const float beta = bc;
  for (int idx = threadIdx.x; idx < E1 * E1; idx += NT) {
    const int r = idx / E1, c = idx - r * E1;
    const int i = i0 - CH + r, j = j0 - CH + c;
    const int64_t gidx = (int64_t)refl(i, g.nx) * g.ny + refl(j, g.ny);
    const float pn = fmaf(beta, p_old[gidx], t[gidx]);       // p = t + beta p   (CGLS.py:72)
    P1[r * (E1 + 1) + c] = pn;
    if (r >= CH && r < CH + CT && c >= CH && c < CH + CT && i < g.nx && j < g.ny) p_new[gidx] = pn;
  }
  __syncthreads();
  blur_lds<CT, CT>(P1, E1 + 1, tmp, W, CT + 1, g.rwf, g.cwf);   // w = A p on the tile   (CGLS.py:60)
  double ss = 0.0;
  for (int idx = threadIdx.x; idx < CT * CT; idx += NT) {
    const int r = idx / CT, c = idx - r * CT;
    if (i0 + r < g.nx && j0 + c < g.ny) {
      const float w = W[r * (CT + 1) + c];
      ss += (double)w * w;
    }
  }
  ss = block_sum<NT>(ss, red);
  if (threadIdx.x == 0) PD[blockIdx.x] = ss;
}

template <bool HAS_XT>
__global__ __launch_bounds__(NT) void k_cgls_tile_b(TiledGeom g, const float* __restrict__ p, const float* __restrict__ r_old,
                                                    float* __restrict__ r_new, float* __restrict__ t,
                                                    const float* __restrict__ x_old, float* __restrict__ x_new,
                                                    const float* __restrict__ x_true, ScalarSrc del, const double* __restrict__ gamma,
                                                    double* __restrict__ dpub, double* __restrict__ PG, double* __restrict__ NP) {
  __shared__ float P2[E2 * (E2 + 1)];
  __shared__ float tmp[E2 * (E1 + 1)];
  __shared__ float W[E1 * (E1 + 1)];          // w on tile + halo, then r_k there
  __shared__ float T[CT * (CT + 1)];
  __shared__ double red[NT / 64];
  __shared__ float bc;
  const int ty = blockIdx.x / g.tiles_x, tx = blockIdx.x - ty * g.tiles_x;
  const int i0 = ty * CT, j0 = tx * CT;
  if (threadIdx.x < 64) {                                   // alpha = gamma_{k-1} / ||w||^2; block 0 publishes ||w||^2
    const double d = scalar_from_wave(del, threadIdx.x);
    if (threadIdx.x == 0) {
      bc = (float)(*gamma / d);
      if (blockIdx.x == 0) *dpub = d;
    }
  }
  for (int idx = threadIdx.x; idx < E2 * E2; idx += NT) {
    const int r = idx / E2, c = idx - r * E2;
    P2[r * (E2 + 1) + c] = p[(int64_t)refl(i0 - 2 * CH + r, g.nx) * g.ny + refl(j0 - 2 * CH + c, g.ny)];
  }
  __syncthreads();
  const float alpha = bc;
  blur_lds<E1, E1>(P2, E2 + 1, tmp, W, E1 + 1, g.rwf, g.cwf);   // w = A p on tile + halo (positions outside the image: unused)
  // r_k = r_{k-1} - alpha w   (CGLS.py:67) where the position is a pixel; its mirror pixel's value where it is not
  for (int idx = threadIdx.x; idx < E1 * E1; idx += NT) {
    const int r = idx / E1, c = idx - r * E1;
    const int i = i0 - CH + r, j = j0 - CH + c;
    if ((unsigned)i < (unsigned)g.nx && (unsigned)j < (unsigned)g.ny) {
      const int64_t gidx = (int64_t)i * g.ny + j;
      const float rn = fmaf(-alpha, W[r * (E1 + 1) + c], r_old[gidx]);
      W[r * (E1 + 1) + c] = rn;
      if (r >= CH && r < CH + CT && c >= CH && c < CH + CT) r_new[gidx] = rn;
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < E1 * E1; idx += NT) {
    const int r = idx / E1, c = idx - r * E1;
    const int i = i0 - CH + r, j = j0 - CH + c;
    if (!((unsigned)i < (unsigned)g.nx && (unsigned)j < (unsigned)g.ny)) {
      const int mr = refl(i, g.nx) - (i0 - CH), mc = refl(j, g.ny) - (j0 - CH);
      // the mirror of a halo position that an output of this tile reads lies inside the halo region; others are not read
      W[r * (E1 + 1) + c] = ((unsigned)mr < (unsigned)E1 && (unsigned)mc < (unsigned)E1) ? W[mr * (E1 + 1) + mc] : 0.f;
    }
  }
  __syncthreads();
  blur_lds<CT, CT>(W, E1 + 1, tmp, T, CT + 1, g.rwt, g.cwt);    // t = A^T r on the tile   (CGLS.py:68)
  double sg = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int idx = threadIdx.x; idx < CT * CT; idx += NT) {
    const int r = idx / CT, c = idx - r * CT;
    const int i = i0 + r, j = j0 + c;
    if (i < g.nx && j < g.ny) {
      const int64_t gidx = (int64_t)i * g.ny + j;
      const float tv = T[r * (CT + 1) + c];
      t[gidx] = tv;
      sg += (double)tv * tv;
      const float d = alpha * P2[(r + 2 * CH) * (E2 + 1) + c + 2 * CH];       // x += alpha p   (CGLS.py:65)
      const float xn = x_old[gidx] + d;
      x_new[gidx] = xn;
      s0 += (double)xn * xn;
      s1 += (double)d * d;
      if (HAS_XT) {
        const double e = (double)xn - (double)x_true[gidx];
        s2 += e * e;
      }
    }
  }
  sg = block_sum<NT>(sg, red);
  s0 = block_sum<NT>(s0, red);
  s1 = block_sum<NT>(s1, red);
  if (HAS_XT) s2 = block_sum<NT>(s2, red);
  if (threadIdx.x == 0) {
    PG[blockIdx.x] = sg;
    NP[(size_t)blockIdx.x * 3 + 0] = s0;
    NP[(size_t)blockIdx.x * 3 + 1] = s1;
    NP[(size_t)blockIdx.x * 3 + 2] = HAS_XT ? s2 : 0.0;
  }
}

int tiled_geom(trk_op* A, TiledGeom* g, int* ntiles) {
  int nx, ny, kh, kw;
  const float *sf, *st;
  if (!blur_separable_params(A, &nx, &ny, &kh, &kw, &sf, &st)) return 0;
  if (kh > 9 || kw > 9 || nx < 16 || ny < 16) return 0;
  float hf[18], ht[18];
  if (hipMemcpy(hf, sf, sizeof(float) * (kw + kh), hipMemcpyDeviceToHost) != hipSuccess) return 0;
  if (hipMemcpy(ht, st, sizeof(float) * (kw + kh), hipMemcpyDeviceToHost) != hipSuccess) return 0;
  for (int o = 0; o < 9; ++o) g->rwf[o] = g->cwf[o] = g->rwt[o] = g->cwt[o] = 0.f;
  const int T = kh - 1 - kh / 2, L = kw - 1 - kw / 2;        // y[i][j] = sum c[a][b] x~[i - T + a][j - L + b]
  for (int b = 0; b < kw; ++b) {
    g->rwf[b - L + 4] = hf[b];
    g->rwt[b - L + 4] = ht[b];
  }
  for (int a = 0; a < kh; ++a) {
    g->cwf[a - T + 4] = hf[kw + a];
    g->cwt[a - T + 4] = ht[kw + a];
  }
  g->nx = nx;
  g->ny = ny;
  g->tiles_x = ceil_div(ny, CT);
  *ntiles = g->tiles_x * ceil_div(nx, CT);
  return 1;
}

}  // namespace

extern "C" {

int trk_cgls_tiled_caps(trk_op* A, int np_capacity_blocks, int pcap, int* can) {
  TRK_REQUIRE(A && can, "trk_cgls_tiled_caps: NULL argument");
  TiledGeom g;
  int ntiles = 0;
  *can = (tiled_geom(A, &g, &ntiles) && ntiles <= np_capacity_blocks && ntiles <= pcap) ? 1 : 0;
  return TRK_OK;
}

int trk_cgls_iterate_tiled(trk_op* A, int k_first, int n_iters, float* P, int64_t p_ld, float* R, int64_t r_ld, float* t,
                           float* X, int64_t x_ld, int keep_history, const float* x_prev, const float* x_true, double* S,
                           double* PG, double* PD, int pcap, double* NP, int np_capacity_blocks, int* n_g_inout,
                           int* n_np_inout, trk_stream stream) {
  TRK_REQUIRE(A && P && R && t && X && x_prev && S && PG && PD && NP && n_g_inout && n_np_inout,
              "trk_cgls_iterate_tiled: NULL argument");
  TRK_REQUIRE(k_first >= 1 && n_iters >= 0, "trk_cgls_iterate_tiled: need k_first >= 1, n_iters >= 0");
  struct Cache { TiledGeom g; int ntiles; };                // the geometry (two small device-to-host copies) once per handle
  if (!A->aux) {
    Cache c;
    if (!tiled_geom(A, &c.g, &c.ntiles)) return fail(TRK_EUNSUPPORTED, "trk_cgls_iterate_tiled: needs a separable blur <= 9x9 on an image >= 16x16");
    A->aux = malloc(sizeof(Cache));
    if (!A->aux) return fail(TRK_ENOMEM, "trk_cgls_iterate_tiled: out of memory");
    memcpy(A->aux, &c, sizeof(Cache));
  }
  const TiledGeom g = static_cast<Cache*>(A->aux)->g;
  const int ntiles = static_cast<Cache*>(A->aux)->ntiles;
  TRK_REQUIRE(ntiles <= np_capacity_blocks && ntiles <= pcap, "trk_cgls_iterate_tiled: %d tiles exceed the partial buffers", ntiles);
  hipStream_t s = (hipStream_t)stream;
  int n_g = *n_g_inout;
  for (int k = k_first; k < k_first + n_iters; ++k) {
    const int64_t b = 5 * (int64_t)k;
    float *p_old = P + (int64_t)((k - 1) & 1) * p_ld, *p_new = P + (int64_t)(k & 1) * p_ld;
    float *r_old = R + (int64_t)((k - 1) & 1) * r_ld, *r_new = R + (int64_t)(k & 1) * r_ld;
    const double* gprev = (k <= 2) ? S : S + 5 * (int64_t)(k - 2) + 1;      // gamma_{k-2}
    double* gpub = (k == 1) ? S : S + b - 4;                                 // gamma_{k-1} goes here
    float* x_new = X + (int64_t)(keep_history ? (k - 1) : ((k - 1) & 1)) * x_ld;
    hipLaunchKernelGGL(k_cgls_tile_a, dim3(ntiles), dim3(NT), 0, s, g, t, p_old, p_new, ScalarSrc{PG, n_g}, gprev, gpub,
                       k == 1 ? 1 : 0, PD);
    double* np = NP + 3 * (int64_t)ntiles * (k - 1);
    if (x_true)
      hipLaunchKernelGGL(k_cgls_tile_b<true>, dim3(ntiles), dim3(NT), 0, s, g, p_new, r_old, r_new, t, x_prev, x_new, x_true,
                         ScalarSrc{PD, ntiles}, gpub, S + b, PG, np);
    else
      hipLaunchKernelGGL(k_cgls_tile_b<false>, dim3(ntiles), dim3(NT), 0, s, g, p_new, r_old, r_new, t, x_prev, x_new, x_true,
                         ScalarSrc{PD, ntiles}, gpub, S + b, PG, np);
    TRK_LAUNCH_CHECK();
    n_g = ntiles;
    x_prev = x_new;
  }
  *n_g_inout = n_g;
  *n_np_inout = ntiles;
  return TRK_OK;
}

}  // extern "C"
