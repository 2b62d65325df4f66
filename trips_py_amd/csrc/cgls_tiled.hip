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
// The blur is cheap next to a dispatch (a separable 9 x 9 on 48 x 48 floats from LDS), so recomputing it on halos costs
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
constexpr int NT = 512;          // 8 waves per workgroup: one workgroup per CU at 512^2, so latency is hidden inside it

struct TiledGeom {
  int nx, ny, tiles_x;
  float rwf[9], cwf[9], rwt[9], cwt[9];     // centred 9-tap row / column weights: forward and "transpose" (flipped PSF)
};

// half-sample symmetric reflection: one fold, no division — exact for an overshoot of at most n (any position a blur of an
// in-image pixel reads: overshoot <= 8 on axes >= 16).  A partial last tile also LOADS positions up to 39 past its origin, i.e.
// further out than n on short axes (n < (i0 + 40) / 2: 16..19, 33..35); those cells only ever feed out-of-image LDS cells, but
// the address must stay inside the buffer, so the folded index is clamped.
__device__ __forceinline__ int refl(int i, int n) {
  const int f = i < 0 ? -1 - i : (i >= n ? 2 * n - 1 - i : i);
  return min(max(f, 0), n - 1);
}

// out[R][C] (row stride so) = separable 9 x 9 correlation of in[R + 8][C + 8] (row stride si, a multiple of 4: rows are
// 16-byte aligned) with row weights rw, column weights cw; tmp holds (R + 8) rows of stride C + 4.  Register blocking: an item
// is FOUR outputs along the filter direction from 12 inputs (3 LDS reads per output instead of 9); with 512 threads every
// pass of every blur in this file is at most one item per thread.  Ends with the result visible (barrier).
typedef float f4t __attribute__((ext_vector_type(4)));
template <int R, int C, bool END_BARRIER = true>
__device__ __forceinline__ void blur_lds(const float* __restrict__ in, int si, float* __restrict__ tmp, float* __restrict__ out,
                                         int so, const float* rw, const float* cw) {
  static_assert(R % 4 == 0 && C % 4 == 0, "blocks of four outputs");
  constexpr int ST = C + 4;
  for (int idx = threadIdx.x; idx < (R + 8) * (C / 4); idx += NT) {
    const int r = idx / (C / 4), c = (idx - r * (C / 4)) * 4;
    const f4t* p = reinterpret_cast<const f4t*>(in + r * si + c);
    const f4t v0 = p[0], v1 = p[1], v2 = p[2];
    const float v[12] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3], v2[0], v2[1], v2[2], v2[3]};
    f4t a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 9; ++o) {
      a[0] = fmaf(rw[o], v[o], a[0]);
      a[1] = fmaf(rw[o], v[o + 1], a[1]);
      a[2] = fmaf(rw[o], v[o + 2], a[2]);
      a[3] = fmaf(rw[o], v[o + 3], a[3]);
    }
    *reinterpret_cast<f4t*>(tmp + r * ST + c) = a;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < (R / 4) * C; idx += NT) {
    const int r = (idx / C) * 4, c = idx - (idx / C) * C;
    const float* p = tmp + r * ST + c;
    float v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = p[k * ST];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int o = 0; o < 9; ++o) {
      a0 = fmaf(cw[o], v[o], a0);
      a1 = fmaf(cw[o], v[o + 1], a1);
      a2 = fmaf(cw[o], v[o + 2], a2);
      a3 = fmaf(cw[o], v[o + 3], a3);
    }
    out[r * so + c] = a0;
    out[(r + 1) * so + c] = a1;
    out[(r + 2) * so + c] = a2;
    out[(r + 3) * so + c] = a3;
  }
  if (END_BARRIER) __syncthreads();
}

// A scalar that is still the block partials of the previous kernel: the loads are ISSUED at the top of the kernel (lane l of
// wave 0 takes p[l], p[l + 64], ...) and SUMMED only where the value is needed, behind the LDS work that does not depend on it
// — a partial is a trip to the memory side (written through other XCDs' L2), and waiting for it up front cost every wave of
// the workgroup ~1.5 us at the first barrier.
constexpr int PMAX = 16;                                   // 64 * 16 = 1024 partials at most
__device__ __forceinline__ void partials_issue(const ScalarSrc s, double (&v)[PMAX]) {
#pragma unroll
  for (int i = 0; i < PMAX; ++i) {
    const int idx = (int)threadIdx.x + 64 * i;
    v[i] = (threadIdx.x < 64 && idx < s.n) ? s.p[idx] : 0.0;
  }
}
__device__ __forceinline__ double partials_sum(const double (&v)[PMAX]) {   // wave 0 only; value in every lane
  double a = 0.0;
#pragma unroll
  for (int i = 0; i < PMAX; ++i) a += v[i];
  return wave_sum_all(a);
}

// four block sums with one pair of barriers
__device__ __forceinline__ void block_sum4(double& a, double& b, double& c, double& d, double* lds /* 4 * NT/64 */) {
  a = wave_sum(a);
  b = wave_sum(b);
  c = wave_sum(c);
  d = wave_sum(d);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) {
    lds[wid * 4 + 0] = a;
    lds[wid * 4 + 1] = b;
    lds[wid * 4 + 2] = c;
    lds[wid * 4 + 3] = d;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = b = c = d = 0.0;
    for (int w = 0; w < NT / 64; ++w) {
      a += lds[w * 4 + 0];
      b += lds[w * 4 + 1];
      c += lds[w * 4 + 2];
      d += lds[w * 4 + 3];
    }
  }
}

__global__ __launch_bounds__(NT) void k_cgls_tile_a(TiledGeom g, const float* __restrict__ t, const float* __restrict__ p_old,
                                                    float* __restrict__ p_new, ScalarSrc gam, const double* __restrict__ gprev,
                                                    double* __restrict__ gpub, int first, double* __restrict__ PD) {
  constexpr int S1 = E1 + 4;                                // LDS row strides: multiples of 4 (16-byte aligned rows)
  __shared__ __attribute__((aligned(16))) float T1[E1 * S1];
  __shared__ __attribute__((aligned(16))) float P1[E1 * S1];
  __shared__ __attribute__((aligned(16))) float tmp[E1 * (CT + 4)];
  __shared__ __attribute__((aligned(16))) float WA[CT * (CT + 1)];
  __shared__ __attribute__((aligned(16))) float WB[CT * (CT + 1)];
  __shared__ double red[NT / 64];
  __shared__ float bc;
  const int ty = blockIdx.x / g.tiles_x, tx = blockIdx.x - ty * g.tiles_x;
  const int i0 = ty * CT, j0 = tx * CT;
  // every global load of the kernel is issued before anything waits: one memory latency
  constexpr int NL1 = (E1 * E1 + NT - 1) / NT;
  float tv[NL1], pv[NL1];
  int gi[NL1];
#pragma unroll
  for (int q = 0; q < NL1; ++q) {
    const int idx = threadIdx.x + q * NT;
    const int r = idx / E1, c = idx - r * E1;
    const int i = i0 - CH + r, j = j0 - CH + c;
    const bool in = idx < E1 * E1;
    gi[q] = in ? refl(i, g.nx) * g.ny + refl(j, g.ny) : 0;
    tv[q] = t[gi[q]];
    pv[q] = p_old[gi[q]];
  }
  double part[PMAX];
  partials_issue(gam, part);
  const double gp = first ? 1.0 : *gprev;
#pragma unroll
  for (int q = 0; q < NL1; ++q) {
    const int idx = threadIdx.x + q * NT;
    if (idx < E1 * E1) {
      const int r = idx / E1, c = idx - r * E1;
      T1[r * S1 + c] = tv[q];
      P1[r * S1 + c] = pv[q];
    }
  }
  __syncthreads();
  // the blur is linear: A (t + beta p) = A t + beta A p — both halves are formed before beta (the previous kernel's partials) is in
  blur_lds<CT, CT>(T1, S1, tmp, WA, CT + 1, g.rwf, g.cwf);
  blur_lds<CT, CT, false>(P1, S1, tmp, WB, CT + 1, g.rwf, g.cwf);
  if (threadIdx.x < 64) {                                   // beta = gamma_{k-1} / gamma_{k-2}; block 0 publishes gamma_{k-1}
    const double gk = partials_sum(part);
    if (threadIdx.x == 0) {
      bc = first ? 0.f : (float)(gk / gp);
      if (blockIdx.x == 0) *gpub = gk;
    }
  }
  __syncthreads();
  const float beta = bc;
#pragma unroll
  for (int q = 0; q < NL1; ++q) {                            // p = t + beta p on the tile   (CGLS.py:72)
    const int idx = threadIdx.x + q * NT;
    const int r = idx / E1, c = idx - r * E1;
    const int i = i0 - CH + r, j = j0 - CH + c;
    if (idx < E1 * E1 && r >= CH && r < CH + CT && c >= CH && c < CH + CT && i < g.nx && j < g.ny)
      p_new[gi[q]] = fmaf(beta, pv[q], tv[q]);
  }
  double ss = 0.0;
  for (int idx = threadIdx.x; idx < CT * CT; idx += NT) {    // ||A p||^2 on the tile   (CGLS.py:60-61)
    const int r = idx / CT, c = idx - r * CT;
    if (i0 + r < g.nx && j0 + c < g.ny) {
      const float w = fmaf(beta, WB[r * (CT + 1) + c], WA[r * (CT + 1) + c]);
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
  constexpr int S2 = E2 + 4, S1 = E1 + 4;                   // LDS row strides: multiples of 4 (16-byte aligned rows)
  __shared__ __attribute__((aligned(16))) float P2[E2 * S2];
  __shared__ __attribute__((aligned(16))) float tmp[E2 * (E1 + 4)];
  __shared__ __attribute__((aligned(16))) float W[E1 * S1];  // w on tile + halo, then r_k there
  __shared__ __attribute__((aligned(16))) float T[CT * (CT + 1)];
  __shared__ double red[4 * (NT / 64)];
  __shared__ float bc;
  const int ty = blockIdx.x / g.tiles_x, tx = blockIdx.x - ty * g.tiles_x;
  const int i0 = ty * CT, j0 = tx * CT;
  // every global load of the kernel is issued before anything waits: p on tile + 8, r on tile + 4, x (and x_true) on the tile
  constexpr int NL2 = (E2 * E2 + NT - 1) / NT, NL1 = (E1 * E1 + NT - 1) / NT, NL0 = (CT * CT + NT - 1) / NT;
  float pv[NL2], rv[NL1], xv[NL0], xtv[NL0];
#pragma unroll
  for (int q = 0; q < NL2; ++q) {
    const int idx = threadIdx.x + q * NT;
    const int r = idx / E2, c = idx - r * E2;
    pv[q] = idx < E2 * E2 ? p[refl(i0 - 2 * CH + r, g.nx) * g.ny + refl(j0 - 2 * CH + c, g.ny)] : 0.f;
  }
#pragma unroll
  for (int q = 0; q < NL1; ++q) {
    const int idx = threadIdx.x + q * NT;
    const int r = idx / E1, c = idx - r * E1;
    const int i = i0 - CH + r, j = j0 - CH + c;
    rv[q] = (idx < E1 * E1 && (unsigned)i < (unsigned)g.nx && (unsigned)j < (unsigned)g.ny) ? r_old[i * g.ny + j] : 0.f;
  }
#pragma unroll
  for (int q = 0; q < NL0; ++q) {
    const int idx = threadIdx.x + q * NT;
    const int r = idx / CT, c = idx - r * CT;
    const int i = i0 + r, j = j0 + c;
    const bool in = idx < CT * CT && i < g.nx && j < g.ny;
    xv[q] = in ? x_old[i * g.ny + j] : 0.f;
    xtv[q] = (HAS_XT && in) ? x_true[i * g.ny + j] : 0.f;
  }
  double part[PMAX];
  partials_issue(del, part);
  const double gm = *gamma;
#pragma unroll
  for (int q = 0; q < NL2; ++q) {
    const int idx = threadIdx.x + q * NT;
    if (idx < E2 * E2) {
      const int r = idx / E2, c = idx - r * E2;
      P2[r * S2 + c] = pv[q];
    }
  }
  __syncthreads();
  blur_lds<E1, E1, false>(P2, S2, tmp, W, S1, g.rwf, g.cwf);   // w = A p on tile + halo (positions outside the image: unused)
  if (threadIdx.x < 64) {                                   // alpha = gamma_{k-1} / ||w||^2, needed only now; block 0 publishes ||w||^2
    const double d = partials_sum(part);
    if (threadIdx.x == 0) {
      bc = (float)(gm / d);
      if (blockIdx.x == 0) *dpub = d;
    }
  }
  __syncthreads();
  const float alpha = bc;
  // r_k = r_{k-1} - alpha w   (CGLS.py:67) where the position is a pixel; its mirror pixel's value where it is not
#pragma unroll
  for (int q = 0; q < NL1; ++q) {
    const int idx = threadIdx.x + q * NT;
    const int r = idx / E1, c = idx - r * E1;
    const int i = i0 - CH + r, j = j0 - CH + c;
    if (idx < E1 * E1 && (unsigned)i < (unsigned)g.nx && (unsigned)j < (unsigned)g.ny) {
      const float rn = fmaf(-alpha, W[r * S1 + c], rv[q]);
      W[r * S1 + c] = rn;
      if (r >= CH && r < CH + CT && c >= CH && c < CH + CT) r_new[i * g.ny + j] = rn;
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < E1 * E1; idx += NT) {
    const int r = idx / E1, c = idx - r * E1;
    const int i = i0 - CH + r, j = j0 - CH + c;
    if (!((unsigned)i < (unsigned)g.nx && (unsigned)j < (unsigned)g.ny)) {
      const int mr = refl(i, g.nx) - (i0 - CH), mc = refl(j, g.ny) - (j0 - CH);
      // the mirror of a halo position that an output of this tile reads lies inside the halo region; others are not read
      W[r * S1 + c] = ((unsigned)mr < (unsigned)E1 && (unsigned)mc < (unsigned)E1) ? W[mr * S1 + mc] : 0.f;
    }
  }
  __syncthreads();
  blur_lds<CT, CT>(W, S1, tmp, T, CT + 1, g.rwt, g.cwt);        // t = A^T r on the tile   (CGLS.py:68)
  double sg = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int q = 0; q < NL0; ++q) {
    const int idx = threadIdx.x + q * NT;
    const int r = idx / CT, c = idx - r * CT;
    const int i = i0 + r, j = j0 + c;
    if (idx < CT * CT && i < g.nx && j < g.ny) {
      const int gidx = i * g.ny + j;
      const float tv = T[r * (CT + 1) + c];
      t[gidx] = tv;
      sg += (double)tv * tv;
      const float d = alpha * P2[(r + 2 * CH) * S2 + c + 2 * CH];             // x += alpha p   (CGLS.py:65)
      const float xn = xv[q] + d;
      x_new[gidx] = xn;
      s0 += (double)xn * xn;
      s1 += (double)d * d;
      if (HAS_XT) {
        const double e = (double)xn - (double)xtv[q];
        s2 += e * e;
      }
    }
  }
  block_sum4(sg, s0, s1, s2, red);
  if (threadIdx.x == 0) {
    PG[blockIdx.x] = sg;
    NP[(size_t)blockIdx.x * 3 + 0] = s0;
    NP[(size_t)blockIdx.x * 3 + 1] = s1;
    NP[(size_t)blockIdx.x * 3 + 2] = HAS_XT ? s2 : 0.0;
  }
}

// ------------------------------------------------------------------------------------------------ two launches, TWO blurs
// The form above blurs four times per iteration (K_A: A t and A p_old on the tile; K_B: A p on tile + halo, A^T r on the tile).
// Two of them are a linear combination of vectors the iteration already has: w_k = A p_k = A t_{k-1} + beta w_{k-1}.  With w kept as
// a full image:
//     K_A2(k)  beta from the gamma partials;  q = A t_{k-1} on the tile (t on tile + 4);  w_k = q + beta w_{k-1},  p_k = t_{k-1} + beta
//              p_{k-1} on the tile, both in place (only the own tile of either is read);  block partial of ||w_k||^2 (measured on
//              the vector itself: nothing accumulates from iteration to iteration)
//     K_B2(k)  alpha = gamma_{k-1} / ||w_k||^2;  r_k = r_{k-1} - alpha w_k on tile + 4 straight from global (w_k is complete: the
//              previous launch has ended);  t_k = A^T r_k on the tile (+ ||t_k||^2 partial);  x_k = x_{k-1} + alpha p_k on the tile
// One blur per kernel: K_A2 has 4 barrier-separated LDS phases instead of 7, K_B2 6 instead of 10; the same partial-sum interface
// (PG, PD) as the four-blur form.
__global__ __launch_bounds__(NT) void k_cgls_tile_a2(TiledGeom g, const float* __restrict__ t, float* __restrict__ p, float* __restrict__ w,
                                                     ScalarSrc gam, const double* __restrict__ gprev, double* __restrict__ gpub,
                                                     int first, double* __restrict__ PD) {
  constexpr int S1 = E1 + 4;
  __shared__ __attribute__((aligned(16))) float T1[E1 * S1];
  __shared__ __attribute__((aligned(16))) float tmp[E1 * (CT + 4)];
  __shared__ __attribute__((aligned(16))) float WA[CT * (CT + 1)];
  __shared__ double red[NT / 64];
  __shared__ float bc;
  const int ty = blockIdx.x / g.tiles_x, tx = blockIdx.x - ty * g.tiles_x;
  const int i0 = ty * CT, j0 = tx * CT;
  constexpr int NL1 = (E1 * E1 + NT - 1) / NT, NL0 = (CT * CT + NT - 1) / NT;
  float tv[NL1], wv[NL0], pv[NL0];
#pragma unroll
  for (int k = 0; k < NL1; ++k) {
    const int idx = threadIdx.x + k * NT;
    const int r = idx / E1, c = idx - r * E1;
    tv[k] = idx < E1 * E1 ? t[refl(i0 - CH + r, g.nx) * g.ny + refl(j0 - CH + c, g.ny)] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < NL0; ++k) {
    const int idx = threadIdx.x + k * NT;
    const int r = idx / CT, c = idx - r * CT;
    const int i = i0 + r, j = j0 + c;
    const bool in = idx < CT * CT && i < g.nx && j < g.ny;
    wv[k] = (in && !first) ? w[i * g.ny + j] : 0.f;
    pv[k] = (in && !first) ? p[i * g.ny + j] : 0.f;
  }
  double part[PMAX];
  partials_issue(gam, part);
  const double gp = first ? 1.0 : *gprev;
#pragma unroll
  for (int k = 0; k < NL1; ++k) {
    const int idx = threadIdx.x + k * NT;
    if (idx < E1 * E1) {
      const int r = idx / E1, c = idx - r * E1;
      T1[r * S1 + c] = tv[k];
    }
  }
  __syncthreads();
  blur_lds<CT, CT, false>(T1, S1, tmp, WA, CT + 1, g.rwf, g.cwf);             // q = A t on the tile
  if (threadIdx.x < 64) {                                   // beta = gamma_{k-1} / gamma_{k-2}; block 0 publishes gamma_{k-1}
    const double gk = partials_sum(part);
    if (threadIdx.x == 0) {
      bc = first ? 0.f : (float)(gk / gp);
      if (blockIdx.x == 0) *gpub = gk;
    }
  }
  __syncthreads();
  const float beta = bc;
  double ss = 0.0;
#pragma unroll
  for (int k = 0; k < NL0; ++k) {
    const int idx = threadIdx.x + k * NT;
    const int r = idx / CT, c = idx - r * CT;
    const int i = i0 + r, j = j0 + c;
    if (idx < CT * CT && i < g.nx && j < g.ny) {
      const float wn = fmaf(beta, wv[k], WA[r * (CT + 1) + c]);               // w_k = A t + beta w_{k-1}     (= A p_k, CGLS.py:60)
      w[i * g.ny + j] = wn;
      p[i * g.ny + j] = fmaf(beta, pv[k], T1[(r + CH) * S1 + c + CH]);        // p_k = t + beta p_{k-1}       (CGLS.py:72)
      ss += (double)wn * wn;
    }
  }
  ss = block_sum<NT>(ss, red);
  if (threadIdx.x == 0) PD[blockIdx.x] = ss;
}

template <bool HAS_XT>
__global__ __launch_bounds__(NT) void k_cgls_tile_b2(TiledGeom g, const float* __restrict__ w, const float* __restrict__ p,
                                                     const float* __restrict__ r_old, float* __restrict__ r_new, float* __restrict__ t,
                                                     const float* __restrict__ x_old, float* __restrict__ x_new,
                                                     const float* __restrict__ x_true, ScalarSrc del, const double* __restrict__ gamma,
                                                     double* __restrict__ dpub, double* __restrict__ PG, double* __restrict__ NP) {
  constexpr int S1 = E1 + 4;
  __shared__ __attribute__((aligned(16))) float W[E1 * S1];                  // r_k on tile + halo
  __shared__ __attribute__((aligned(16))) float tmp[E1 * (CT + 4)];
  __shared__ __attribute__((aligned(16))) float T[CT * (CT + 1)];
  __shared__ double red[4 * (NT / 64)];
  __shared__ float bc;
  const int ty = blockIdx.x / g.tiles_x, tx = blockIdx.x - ty * g.tiles_x;
  const int i0 = ty * CT, j0 = tx * CT;
  constexpr int NL1 = (E1 * E1 + NT - 1) / NT, NL0 = (CT * CT + NT - 1) / NT;
  float wv[NL1], rv[NL1], pv[NL0], xv[NL0], xtv[NL0];
#pragma unroll
  for (int k = 0; k < NL1; ++k) {
    const int idx = threadIdx.x + k * NT;
    const int r = idx / E1, c = idx - r * E1;
    const int i = i0 - CH + r, j = j0 - CH + c;
    const bool in = idx < E1 * E1 && (unsigned)i < (unsigned)g.nx && (unsigned)j < (unsigned)g.ny;
    const int gi = in ? i * g.ny + j : 0;
    wv[k] = in ? w[gi] : 0.f;
    rv[k] = in ? r_old[gi] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < NL0; ++k) {
    const int idx = threadIdx.x + k * NT;
    const int r = idx / CT, c = idx - r * CT;
    const int i = i0 + r, j = j0 + c;
    const bool in = idx < CT * CT && i < g.nx && j < g.ny;
    const int gi = in ? i * g.ny + j : 0;
    pv[k] = in ? p[gi] : 0.f;
    xv[k] = in ? x_old[gi] : 0.f;
    xtv[k] = (HAS_XT && in) ? x_true[gi] : 0.f;
  }
  double part[PMAX];
  partials_issue(del, part);
  const double gm = *gamma;
  if (threadIdx.x < 64) {                                   // alpha = gamma_{k-1} / ||w||^2; block 0 publishes ||w||^2
    const double d = partials_sum(part);
    if (threadIdx.x == 0) {
      bc = (float)(gm / d);
      if (blockIdx.x == 0) *dpub = d;
    }
  }
  __syncthreads();
  const float alpha = bc;
  // r_k = r_{k-1} - alpha w   (CGLS.py:67) where the position is a pixel; its mirror pixel's value where it is not (below)
#pragma unroll
  for (int k = 0; k < NL1; ++k) {
    const int idx = threadIdx.x + k * NT;
    const int r = idx / E1, c = idx - r * E1;
    const int i = i0 - CH + r, j = j0 - CH + c;
    if (idx < E1 * E1 && (unsigned)i < (unsigned)g.nx && (unsigned)j < (unsigned)g.ny) {
      const float rn = fmaf(-alpha, wv[k], rv[k]);
      W[r * S1 + c] = rn;
      if (r >= CH && r < CH + CT && c >= CH && c < CH + CT) r_new[i * g.ny + j] = rn;
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < E1 * E1; idx += NT) {
    const int r = idx / E1, c = idx - r * E1;
    const int i = i0 - CH + r, j = j0 - CH + c;
    if (!((unsigned)i < (unsigned)g.nx && (unsigned)j < (unsigned)g.ny)) {
      const int mr = refl(i, g.nx) - (i0 - CH), mc = refl(j, g.ny) - (j0 - CH);
      W[r * S1 + c] = ((unsigned)mr < (unsigned)E1 && (unsigned)mc < (unsigned)E1) ? W[mr * S1 + mc] : 0.f;
    }
  }
  __syncthreads();
  blur_lds<CT, CT>(W, S1, tmp, T, CT + 1, g.rwt, g.cwt);                      // t_k = A^T r_k on the tile   (CGLS.py:68)
  double sg = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int k = 0; k < NL0; ++k) {
    const int idx = threadIdx.x + k * NT;
    const int r = idx / CT, c = idx - r * CT;
    const int i = i0 + r, j = j0 + c;
    if (idx < CT * CT && i < g.nx && j < g.ny) {
      const int gidx = i * g.ny + j;
      const float tn = T[r * (CT + 1) + c];
      t[gidx] = tn;
      sg += (double)tn * tn;
      const float d = alpha * pv[k];                                           // x += alpha p   (CGLS.py:65)
      const float xn = xv[k] + d;
      x_new[gidx] = xn;
      s0 += (double)xn * xn;
      s1 += (double)d * d;
      if (HAS_XT) {
        const double e = (double)xn - (double)xtv[k];
        s2 += e * e;
      }
    }
  }
  block_sum4(sg, s0, s1, s2, red);
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
  if (hipMemcpy(hf, sf, sizeof(float) * (kw + kh), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(ht, st, sizeof(float) * (kw + kh), hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    return -1;                                      // a HIP error, not an answer about the operator: the caller must not cache it
  }
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

// the tile geometry of a handle (two small device-to-host copies: blocking) once per handle, kept in A->aux; ntiles = 0 records
// "not a separable blur <= 9 x 9 on an image >= 16 x 16" so that the question is not asked of the device again either.  Only
// that STRUCTURAL "no" is cached: a failed copy returns NULL with the error string set and is asked again next time.
struct TiledCache { TiledGeom g; int ntiles; };
static const TiledCache* tiled_cache(trk_op* A) {
  if (!A->aux) {
    TiledCache c{};
    const int have = tiled_geom(A, &c.g, &c.ntiles);
    if (have < 0) {
      (void)fail(TRK_EHIP, "tiled CGLS: could not read the blur's separable factors from the device");
      return nullptr;
    }
    if (!have) c.ntiles = 0;
    A->aux = malloc(sizeof(TiledCache));
    if (!A->aux) {
      (void)fail(TRK_ENOMEM, "tiled CGLS: out of memory");
      return nullptr;
    }
    memcpy(A->aux, &c, sizeof(TiledCache));
  }
  return static_cast<const TiledCache*>(A->aux);
}

int trk_cgls_tiled_caps(trk_op* A, int np_capacity_blocks, int pcap, int* can) {
  TRK_REQUIRE(A && can, "trk_cgls_tiled_caps: NULL argument");
  const TiledCache* c = tiled_cache(A);                    // (asked by every CGLS() call: must not touch the device after the first)
  *can = 0;
  if (!c) return TRK_EHIP;                                 // the error string is set; nothing was cached: the next call asks again
  *can = (c->ntiles > 0 && c->ntiles <= np_capacity_blocks && c->ntiles <= pcap) ? 1 : 0;
  return TRK_OK;
}

int trk_cgls_iterate_tiled(trk_op* A, int k_first, int n_iters, float* P, int64_t p_ld, float* R, int64_t r_ld, float* t,
                           float* X, int64_t x_ld, int keep_history, const float* x_prev, const float* x_true, double* S,
                           double* PG, double* PD, int pcap, double* NP, int np_capacity_blocks, int* n_g_inout,
                           int* n_np_inout, trk_stream stream) {
  TRK_REQUIRE(A && P && R && t && X && x_prev && S && PG && PD && NP && n_g_inout && n_np_inout,
              "trk_cgls_iterate_tiled: NULL argument");
  TRK_REQUIRE(k_first >= 1 && n_iters >= 0, "trk_cgls_iterate_tiled: need k_first >= 1, n_iters >= 0");
  const TiledCache* tc = tiled_cache(A);
  if (!tc) return TRK_EHIP;                                // (message set by tiled_cache)
  if (tc->ntiles <= 0) return fail(TRK_EUNSUPPORTED, "trk_cgls_iterate_tiled: needs a separable blur <= 9x9 on an image >= 16x16");
  const TiledGeom g = tc->g;
  const int ntiles = tc->ntiles;
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

int trk_cgls_iterate_tiled2(trk_op* A, int k_first, int n_iters, float* p, float* w, float* R, int64_t r_ld, float* t, float* X,
                            int64_t x_ld, int keep_history, const float* x_prev, const float* x_true, double* S, double* PG,
                            double* PD, int pcap, double* NP, int np_capacity_blocks, int* n_g_inout, int* n_np_inout,
                            trk_stream stream) {
  TRK_REQUIRE(A && p && w && R && t && X && x_prev && S && PG && PD && NP && n_g_inout && n_np_inout,
              "trk_cgls_iterate_tiled2: NULL argument");
  TRK_REQUIRE(k_first >= 1 && n_iters >= 0, "trk_cgls_iterate_tiled2: need k_first >= 1, n_iters >= 0");
  const TiledCache* tc = tiled_cache(A);
  if (!tc) return TRK_EHIP;
  if (tc->ntiles <= 0) return fail(TRK_EUNSUPPORTED, "trk_cgls_iterate_tiled2: needs a separable blur <= 9x9 on an image >= 16x16");
  const TiledGeom g = tc->g;
  const int ntiles = tc->ntiles;
  TRK_REQUIRE(ntiles <= np_capacity_blocks && ntiles <= pcap, "trk_cgls_iterate_tiled2: %d tiles exceed the partial buffers", ntiles);
  hipStream_t s = (hipStream_t)stream;
  int n_g = *n_g_inout;
  for (int k = k_first; k < k_first + n_iters; ++k) {
    const int64_t b = 5 * (int64_t)k;
    float *r_old = R + (int64_t)((k - 1) & 1) * r_ld, *r_new = R + (int64_t)(k & 1) * r_ld;
    const double* gprev = (k <= 2) ? S : S + 5 * (int64_t)(k - 2) + 1;      // gamma_{k-2}
    double* gpub = (k == 1) ? S : S + b - 4;                                 // gamma_{k-1} goes here
    float* x_new = X + (int64_t)(keep_history ? (k - 1) : ((k - 1) & 1)) * x_ld;
    hipLaunchKernelGGL(k_cgls_tile_a2, dim3(ntiles), dim3(NT), 0, s, g, t, p, w, ScalarSrc{PG, n_g}, gprev, gpub, k == 1 ? 1 : 0, PD);
    double* np = NP + 3 * (int64_t)ntiles * (k - 1);
    if (x_true)
      hipLaunchKernelGGL(k_cgls_tile_b2<true>, dim3(ntiles), dim3(NT), 0, s, g, w, p, r_old, r_new, t, x_prev, x_new, x_true,
                         ScalarSrc{PD, ntiles}, gpub, S + b, PG, np);
    else
      hipLaunchKernelGGL(k_cgls_tile_b2<false>, dim3(ntiles), dim3(NT), 0, s, g, w, p, r_old, r_new, t, x_prev, x_new, x_true,
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
