// fanbeam2d.hip — fan-beam (flat detector) line projector and its matched adjoint.
//
// Replaces astra.OpTomo over create_proj_geom('fanflat', det_pitch, p, theta, SOD, ODD) + create_projector('line_fanflat')
// of trips/test_problems/Tomography.py:53-88 (p = int(sqrt(2) nx) detector pixels, SOD = 3 nx, ODD = nx, pitch = 4/3).
// 'line' = the ray from the source to a detector-pixel centre weighs every image pixel by the LENGTH of their
// intersection (Siddon).  Pinned to the reference's two rendered ASTRA outputs of this geometry (tests/golden/
// fanbeam_demo_image.npz: correlation 0.9999 with the sinogram; astra-toolbox itself is absent): the convention —
// pixel (r,c) is the square [c - N/2, c+1 - N/2] x [N/2 - r - 1, N/2 - r]; at angle t the source sits at
// (SOD sin t, -SOD cos t), the detector centre at (-ODD sin t, ODD cos t), detector axis (cos t, sin t); detector pixel d
// is centred at (d - (p-1)/2) * pitch along it.  Sinogram (n_ang, n_det) row-major.
//
// Row-march form (source and detector both outside the image's circumscribed circle — the reference's geometry).  A ray is
// STEEP (|dy| >= |dx|: it crosses every image row once) or shallow (every column once).  In index coordinates a steep ray is
// X(Y) = X0 + Y M, |M| <= 1: inside row r it runs from Xa = X0 + r M to Xb = Xa + M, a segment of length len = sqrt(1 + M^2)
// that touches one column, or two neighbouring ones split at the integer between Xa and Xb:
//     cl = floor(min), ch = floor(max);  ch == cl: weight len on (r, cl);  else f = (ch - min) / |M|: len f on cl, len (1 - f) on ch.
// (Shallow rays: the same with rows and columns exchanged.)  {X0, M, len, class} per ray is tabulated once per operator in
// float64 and stored in FIXED POINT (X0 64 bits, M 32 bits, 30 fraction bits: 16 bytes per ray), so that a step's position
// X0 + r M is exact integer arithmetic.  Lengths come out of pixel-sized quantities, not out of differences of ray parameters
// along a ~4N-long segment as in a Siddon traversal, and carry no error that grows with N: agreement with the brute-force float64
// oracle at 1e-5 on white noise (Siddon pair: 1e-4; the same row-march in fp32 positions: 1.5e-5 at N = 132 and growing).
// Forward: one thread per ray marches the N rows (columns), two taps per step, shallow rays on a transposed copy of the image.
// Adjoint: gather, one thread per pixel: per angle the detectors whose rays can touch the pixel (its centre's projection +- the
// projected half diagonal) are looked up in the table and weighed BY THE FORWARD'S OWN EXPRESSIONS — the same floats, so the pair
// is matched to summation order — no atomics.
// General fallback (detector inside the circumscribed circle): the Siddon traversal / slab-clipping pair below, as before.
#include "trk_internal.h"

#include <cmath>
#include <cstdlib>
#include <vector>

using namespace trk;

namespace {

struct FanAngle {
  float sx, sy;      // source
  float d0x, d0y;    // centre of detector pixel 0
  float ux, uy;      // detector pixel pitch vector
  float nx, ny;      // unit normal source -> detector centre
  float ex, ey;      // source - centre of detector pixel 0 (the adjoint's locator)
  float uxs, uys;    // pitch vector / pitch^2: a point of the detector line -> detector coordinate
};

struct FanRay {      // row-march form of one ray (index coordinates; see the header), fixed point with FAN_Q fraction bits
  unsigned x0_lo;    // X0 * 2^30 as a 64-bit integer, low word; BIT 0 = class: 1 = shallow ray (marches over columns, rows and
  int x0_hi;         //   columns exchanged), high word
  int m;             // M * 2^30, |M| <= 1
  float len;         // sqrt(1 + M^2) (the adjoint's per-apply records: times the sinogram value)
};
constexpr int FAN_Q = 30;

// The adjoint's per-view constants (round 6: k_fan_adj_views), folded on the host in float64 so that the locator of a pixel (column c,
// row r) is  den = c nx + r nny + kd ,  lo = (c Ux + r Uy + Uklo) / den + kelo ,  hi = lo + dr2 / den  (+ 0.02 in the comparisons):
// [lo, hi] = the detector interval whose rays can touch the pixel (uc -+ w of k_fan_adj_march), candidates ceil(lo) .. floor(hi).
// Pairs {den's, lo's} side by side: one packed FMA forms both.
struct FanAdjView {
  float nx, Ux, nny, Uy;
  float kd, Uklo, kelo, dr2;
};
static_assert(sizeof(FanAdjView) == 32, "two 16-byte LDS reads per view");

struct FanImpl {
  int N, nd, na;
  float dsd;         // source-detector distance
  float pitch;
  FanAngle* ang_dev;
  FanRay* rays;      // [na * nd], NULL: general fallback kernels
  FanRay* recs;      // [na][nd + 4] FanRec per apply (adjoint); the forward's band partials in between
  float* xT;         // two padded copies of the image (forward: as it is / transposed), owned by the handle
  float reach;       // half width, in detector pixels per unit magnification, of the detector interval a pixel can touch
  int max_cand;      // most detectors that interval can hold anywhere in the image
  // band-resident forward (small images): the rays by class (steep first), and band partials of their own
  int* cls_list;     // [na * nd] ray indices: the n_steep steep rays in table order, then the shallow ones
  int n_steep;
  float* band_part;  // [N / 64][na * nd]
  FanAdjView* views; // [na] (adjoint, round 6)
  int* view_cls;     // [na] 0: every ray of the view is steep (marching index = the pixel's row), 1: every ray shallow, 2: both kinds
  int dense;         // 1: no pixel's interval is ever empty (2 w >= 1 everywhere: the reference's geometry)
};

// One marching step of a ray: column (row) cl and its right neighbour, with their weights as fractions of the segment length.
// The position is a 64-bit integer X0 + tt * M in units of 2^-30 — in fp32 it carried 6e-8 N of error, 1.5e-5 relative on white
// noise in the adjoint at N = 132 already and growing with N; like this every weight is within 2^-24 of its exact value whatever
// N, and the forward and the adjoint evaluate the very same integers.
// The fraction of a step's segment that lies in its first column: the interval [lo, lo + |M|) reaches the next integer, `dist` away,
// when |M| >= dist and is split there at dist / |M|; otherwise all of it lies in the first column.  Both cases are ONE multiply with
// the hardware clamp to [0, 1] (dist / |M| > 1 exactly when the integer is not reached; |M| = 0 comes with a reciprocal of 2^-2,
// dist >= 4): the compare / min / select form was five of the ~20 vector instructions of a marching step.
// `frac32` is the position's fraction in units of 2^-32 (the 30 fraction bits of the tables, shifted up by two: then the carry of a
// 32-bit add IS the step into the next column); dist = 2^32 - frac32 = ~frac32 + 1 — the NOT in the integer unit (2^32 itself does
// not fit a register), the + 1 inside the FMA: f = clamp((float)(~frac32) * inv32 + inv32), exact for small distances, where it
// matters.  `inv32` = 1 / (|M| in units of 2^-32); |M| = 0: 1.0 (any distance then clamps to 1).  The forward and the adjoint both
// call this: the same bits on the same integers.
__device__ __forceinline__ float fan_split(unsigned frac32, float inv32) {
  float f;
  const float fd = (float)(~frac32);
  asm("v_fma_f32 %0, %1, %2, %2 clamp" : "=v"(f) : "v"(fd), "v"(inv32));
  return f;
}

struct FanRayRegs {
  long long x0;      // low bit cleared
  int m, mneg;
  float inv32;       // 1 / (|M| in units of 2^-32); |M| = 0: 1.0 (fan_split)
  bool shallow;
};
__device__ __forceinline__ float fan_inv32(int m) {
  const int absm = m < 0 ? -m : m;
  return absm > 0 ? 0.25f * __builtin_amdgcn_rcpf((float)absm) : 1.0f;
}
__device__ __forceinline__ FanRayRegs fan_ray_regs(const FanRay& q) {
  FanRayRegs r;
  r.shallow = (q.x0_lo & 1u) != 0u;
  r.x0 = ((long long)q.x0_hi << 32) | (long long)(q.x0_lo & ~1u);
  r.m = q.m;
  r.mneg = q.m < 0 ? q.m : 0;
  r.inv32 = fan_inv32(q.m);
  return r;
}
// the step whose interval starts at `lo` (= X0 + tt M + min(M, 0), 64-bit fixed point)
__device__ __forceinline__ void fan_step_at(long long lo, const FanRayRegs& r, int& cl, float& w0, float& w1) {
  cl = (int)(lo >> FAN_Q);
  w0 = fan_split((unsigned)lo << (32 - FAN_Q), r.inv32);
  w1 = 1.f - w0;
}
__device__ __forceinline__ void fan_step(int tt, const FanRayRegs& r, int& cl, float& w0, float& w1) {
  fan_step_at(r.x0 + (long long)tt * (long long)r.m + (long long)r.mneg, r, cl, w0, w1);
}

// Two padded copies of the image per forward apply, rows of N + 4 floats with two zero columns on either side: P0 as it is
// (steep rays), P1 transposed (shallow rays).  A marching step then takes its two taps with ONE 8-byte load at column
// clamp(cl, -2, N) + 2 and needs no range test: whatever lies outside the image reads zeros.
constexpr int FAN_PAD = 2;
__global__ __launch_bounds__(256) void k_fan_pad_copies(const float* __restrict__ in, int64_t ld_in, float* __restrict__ P0,
                                                        float* __restrict__ P1, int N) {
  __shared__ float tile[32][33];
  const int W = N + 2 * FAN_PAD;
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const bool ok = by + k < N && bx + tx < N;
    const float v = ok ? in[(int64_t)(by + k) * N + bx + tx] : 0.f;
    tile[k][tx] = v;
    if (ok) P0[(int64_t)(by + k) * W + FAN_PAD + bx + tx] = v;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8)
    if (bx + k < N && by + tx < N) P1[(int64_t)(bx + k) * W + FAN_PAD + by + tx] = tile[tx][k];
}

typedef float fan_f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4f __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4r __attribute__((ext_vector_type(4)));

// gridDim.y > 1: the march is cut into that many bands of `band` steps, band b leaving its sum (without the ray's length) in
// part[b][ray] for k_fan_bands_sum — one thread per ray is 2 waves per SIMD at 512^2 x 180 x 724, too few to hide the gathers.
__global__ __launch_bounds__(256) void k_fan_fwd_march(const float* __restrict__ P0, int64_t padded,
                                                       float* __restrict__ sino, int N, int64_t nrays,
                                                       const FanRay* __restrict__ rays, int band, float* __restrict__ part) {
  const int64_t ray_raw = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live_ray = ray_raw < nrays;                 // (the lane stays: the wave takes a minimum / maximum over its rays below)
  const int64_t ray = live_ray ? ray_raw : nrays - 1;
  const FanRay gq = rays[ray];
  const FanRayRegs g = fan_ray_regs(gq);
  const int W = N + 2 * FAN_PAD;
  // both padded copies behind ONE buffer resource (they are one allocation: P1 = P0 + padded): the step's address is then
  //   [scalar]  row (t0 + u) W 4   +   [vector]  4 clamp(cl) + (the ray's copy, the pad)   — one v_lshl_add per step
  // where pointer arithmetic took a 64-bit add for the row and a sign extension + 64-bit add for the column
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)P0, 0, (unsigned)(2 * padded * 4), 0x00020000);
  const unsigned lane_base = (g.shallow ? (unsigned)(padded * 4) : 0u) + FAN_PAD * 4u;
  float acc0 = 0.f, acc1 = 0.f;
  int t0 = blockIdx.y * band;
  int t_end = (t0 + band < N) ? t0 + band : N;
  const int Nimg = N;
  // Rows in which NONE of the wave's 64 rays touches the image are skipped: outside columns [-2, N] both taps read the zero padding,
  // so those steps add exact zeros.  With 1.41 N detectors over the image's diagonal a quarter of the rays of an axis-aligned view
  // miss the square altogether, and oblique rays enter and leave through its sides.  Per ray the row interval comes from the fp32
  // line X(t) = X0 + t M with four columns of margin (the estimate is good to a small fraction of one); the wave marches the union.
  {
    const float x0f = (float)((double)(g.x0 + (long long)g.mneg) * (1.0 / 1073741824.0));
    const float mff = (float)g.m * (1.0f / 1073741824.0f);
    const float lo = -6.f - x0f, hi = (float)Nimg + 5.f - x0f;                     // lo <= t M <= hi
    float ta_f, tb_f;
    if (fabsf(mff) < 1e-6f) {
      const bool in = lo <= 0.f && hi >= 0.f;
      ta_f = in ? -1e9f : 1e9f;
      tb_f = in ? 1e9f : -1e9f;
    } else {
      const float r = __builtin_amdgcn_rcpf(mff), u1 = lo * r, u2 = hi * r;
      ta_f = fminf(u1, u2) - 2.f;
      tb_f = fmaxf(u1, u2) + 3.f;
    }
    int ta = (int)fminf(fmaxf(floorf(ta_f), -1e9f), 1e9f), tb = (int)fminf(fmaxf(ceilf(tb_f), -1e9f), 1e9f);   // rows [ta, tb)
    if (!live_ray) { ta = 0x7fffffff; tb = -0x7fffffff; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      ta = min(ta, __shfl_xor(ta, off, 64));
      tb = max(tb, __shfl_xor(tb, off, 64));
    }
    ta = __builtin_amdgcn_readfirstlane(ta);             // (every lane holds the wave's value: a scalar for the row offsets below)
    tb = __builtin_amdgcn_readfirstlane(tb);
    ta = ta > t0 ? ta : t0;
    tb = tb < t_end ? tb : t_end;
    if (ta >= tb) {
      t0 = t_end;                                        // nothing to march: the sum is zero
    } else {
      t0 += (ta - t0) & ~7;                              // whole blocks of eight steps from the band's first row
      const int te = t0 + ((tb - t0 + 7) & ~7);
      t_end = te < t_end ? te : t_end;
    }
  }
  N = t_end;                                             // (the march below runs to N)
  // the position as column + fraction in units of 2^-32: stepping by M is a 32-bit add whose carry moves the column (exact integers:
  // nothing accumulates); M = mi + mf with mi = floor(M) in {-1, 0, 1}, mf in [0, 1)
  const long long pos0 = g.x0 + (long long)g.mneg + (long long)t0 * (long long)g.m;
  int col = (int)(pos0 >> FAN_Q);
  unsigned frac = (unsigned)pos0 << (32 - FAN_Q);
  const int mi = g.m >> FAN_Q;
  const unsigned mf = (unsigned)g.m << (32 - FAN_Q);
  for (; t0 + 8 <= N; t0 += 8) {
    float w0[8], w1[8];
    fan_f2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      w0[u] = fan_split(frac, g.inv32);
      w1[u] = 1.f - w0[u];
      int cl;
      asm("v_med3_i32 %0, %1, %2, %3" : "=v"(cl) : "v"(col), "n"(-FAN_PAD), "s"(Nimg));
      int rowoff = (t0 + u) * W * 4;                     // wave-uniform: a scalar
      asm("" : "+s"(rowoff));
      unsigned off;
      asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(off) : "v"(cl), "v"(lane_base));
      v[u] = __builtin_bit_cast(fan_f2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off, rowoff, 0));
      const unsigned nf = frac + mf;
      col += mi + (nf < frac ? 1 : 0);
      frac = nf;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float a = v[u][0], b = v[u][1];
      acc0 = fmaf(w0[u], a, acc0);
      acc1 = fmaf(w1[u], b, acc1);
    }
  }
  for (; t0 < N; ++t0) {
    int cl;
    float w0, w1;
    fan_step(t0, g, cl, w0, w1);
    cl = cl < -FAN_PAD ? -FAN_PAD : (cl > Nimg ? Nimg : cl);
    const fan_f2 q = __builtin_bit_cast(fan_f2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(((unsigned)cl << 2) + lane_base), t0 * W * 4, 0));
    acc0 = fmaf(w0, q[0], acc0);
    acc1 = fmaf(w1, q[1], acc1);
  }
  if (!live_ray) return;
  if (gridDim.y == 1) sino[ray] = gq.len * (acc0 + acc1);
  else part[(int64_t)blockIdx.y * nrays + ray] = acc0 + acc1;
}

// Band-resident forward march (small images; round 5, as k_radon_fwd_band of radon2d.hip): a 64-row band of a 512-wide image is
// 130 KB and fits the LDS of one CU.  k_fan_fwd_march takes its two taps per step with one scattered 8-byte load per lane, and the
// texture addresser's rate for such loads (16.6 cycles per wave-load) IS its time (28 of 37 us at 512^2 x 180 x 724), behind a launch
// that makes two padded copies of the image.  Here a workgroup of 16 waves loads its band once — rows of the image for the steep rays,
// 64 COLUMNS transposed while loading for the shallow ones, zero columns either side: no padded copies — and every wave marches
// tasks of 64 rays of the band's class (neighbours in the table) through the band's rows with LDS taps: the same integers, the same
// weights (fan_split), the same skipping of rows no ray of the wave touches.  Band partials [band][ray] for k_fan_bands_sum.
constexpr int FB_ROWS = 64, FB_NT = 1024, FB_PAD = 4, FB_NMAX = 1024;
// rows per band: 64 where 64 x (N + 8) floats fit the LDS (N <= 512), else 32 (N <= 1024: 132 KB)
inline int fb_rows(int N) { return (size_t)FB_ROWS * (N + 2 * FB_PAD) * 4 <= 150 * 1024 ? FB_ROWS : FB_ROWS / 2; }
__global__ __launch_bounds__(FB_NT, 4) void k_fan_fwd_band(const float* __restrict__ img, float* __restrict__ part, int N, int64_t nrays,
                                                           const FanRay* __restrict__ rays, const int* __restrict__ cls_list, int n_steep,
                                                           int nslice, int rows) {
  extern __shared__ __attribute__((aligned(16))) float fband[];   // rows x (N + 2 FB_PAD) floats, then the task counter
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int nw = (int)(blockDim.x >> 6), nthr = (int)blockDim.x;
  const int nbands = N / rows;
  const int RS = N + 2 * FB_PAD;
  int& next_task = *reinterpret_cast<int*>(fband + rows * RS);
  int bid = blockIdx.x;
  const int slice = bid % nslice; bid /= nslice;
  const int b = bid % nbands;
  const int cls = bid / nbands;                                    // 0: steep rays (march over rows), 1: shallow (over columns)
  const int cnt = cls ? (int)nrays - n_steep : n_steep;
  const int* __restrict__ list = cls_list + (cls ? n_steep : 0);
  const int all_tasks = (cnt + 63) / 64;
  const int task0 = (int)((int64_t)all_tasks * slice / nslice), task1 = (int)((int64_t)all_tasks * (slice + 1) / nslice);
  if (task1 <= task0) return;
  typedef float f4b __attribute__((ext_vector_type(4)));
  if (cls) {
    // the band of the transposed image = 64 columns of the image: a wave-load takes 16 image rows x 16 columns (whole 64-byte
    // sectors), a lane's four values go to four rows of the band (consecutive lanes: consecutive addresses)
    const float* __restrict__ X = img + (int64_t)b * rows;
    const int r = lane & 15, jq = lane >> 4;
    const int cgs = rows / 16, pieces = (N / 16) * cgs;
    for (int p0 = wv; p0 < pieces; p0 += 4 * nw) {
      f4b v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pc = p0 + u * nw;
        const int rg = pc / cgs, cg = pc - rg * cgs;
        v[u] = pc < pieces ? *reinterpret_cast<const f4b*>(X + (int64_t)(16 * rg + r) * N + 16 * cg + 4 * jq) : (f4b){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pc = p0 + u * nw;
        if (pc < pieces) {
          const int rg = pc / cgs, cg = pc - rg * cgs;
#pragma unroll
          for (int e = 0; e < 4; ++e) fband[(16 * cg + 4 * jq + e) * RS + FB_PAD + 16 * rg + r] = v[u][e];
        }
      }
    }
  } else {
    const float* __restrict__ I = img + (int64_t)b * rows * N;
    const int q4 = N / 4, tot = rows * q4;
    for (int i0 = threadIdx.x; i0 < tot; i0 += 8 * nthr) {
      f4b v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = i0 + u * nthr;
        v[u] = idx < tot ? *reinterpret_cast<const f4b*>(I + 4 * (int64_t)idx) : (f4b){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = i0 + u * nthr;
        if (idx < tot) {
          const int row = idx / q4, c4 = idx - row * q4;
          *reinterpret_cast<f4b*>(&fband[row * RS + FB_PAD + 4 * c4]) = v[u];
        }
      }
    }
  }
  if (threadIdx.x < rows * 2) {
    const int row = threadIdx.x >> 1, side = threadIdx.x & 1;
    *reinterpret_cast<f4b*>(&fband[row * RS + (side ? FB_PAD + N : 0)]) = (f4b){0.f, 0.f, 0.f, 0.f};
  }
  if (threadIdx.x == 0) next_task = task0 + nw;
  __syncthreads();
  const int band0 = b * rows;
  for (int task = task0 + wv; task < task1;) {
    const int li = task * 64 + lane;
    const bool live_ray = li < cnt;
    const int ray = list[live_ray ? li : cnt - 1];
    const FanRay gq = rays[ray];
    const FanRayRegs g = fan_ray_regs(gq);
    float acc0 = 0.f, acc1 = 0.f;
    int t0 = band0, t_end = band0 + rows;
    {
      // the rows in which any ray of the wave can touch the image (k_fan_fwd_march: the same estimate, the same margins)
      const float x0f = (float)((double)(g.x0 + (long long)g.mneg) * (1.0 / 1073741824.0));
      const float mff = (float)g.m * (1.0f / 1073741824.0f);
      const float lo = -6.f - x0f, hi = (float)N + 5.f - x0f;
      float ta_f, tb_f;
      if (fabsf(mff) < 1e-6f) {
        const bool in = lo <= 0.f && hi >= 0.f;
        ta_f = in ? -1e9f : 1e9f;
        tb_f = in ? 1e9f : -1e9f;
      } else {
        const float r = __builtin_amdgcn_rcpf(mff), u1 = lo * r, u2 = hi * r;
        ta_f = fminf(u1, u2) - 2.f;
        tb_f = fmaxf(u1, u2) + 3.f;
      }
      int ta = (int)fminf(fmaxf(floorf(ta_f), -1e9f), 1e9f), tb = (int)fminf(fmaxf(ceilf(tb_f), -1e9f), 1e9f);
      if (!live_ray) { ta = 0x7fffffff; tb = -0x7fffffff; }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        ta = min(ta, __shfl_xor(ta, off, 64));
        tb = max(tb, __shfl_xor(tb, off, 64));
      }
      ta = __builtin_amdgcn_readfirstlane(ta);
      tb = __builtin_amdgcn_readfirstlane(tb);
      ta = ta > t0 ? ta : t0;
      tb = tb < t_end ? tb : t_end;
      if (ta >= tb) {
        t0 = t_end;
      } else {
        t0 += (ta - t0) & ~7;
        const int te = t0 + ((tb - t0 + 7) & ~7);
        t_end = te < t_end ? te : t_end;
      }
    }
    const long long pos0 = g.x0 + (long long)g.mneg + (long long)t0 * (long long)g.m;
    int col = (int)(pos0 >> FAN_Q);
    unsigned frac = (unsigned)pos0 << (32 - FAN_Q);
    const int mi = g.m >> FAN_Q;
    const unsigned mf = (unsigned)g.m << (32 - FAN_Q);
    const char* base = reinterpret_cast<const char*>(fband) + FB_PAD * 4;
    for (; t0 + 8 <= t_end; t0 += 8) {                             // (bands and the skipped prefix are multiples of eight rows)
      float w0[8], w1[8];
      fan_f2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        w0[u] = fan_split(frac, g.inv32);
        w1[u] = 1.f - w0[u];
        int cl;
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(cl) : "v"(col), "n"(-2), "s"(N));
        int rowoff = (t0 - band0 + u) * RS * 4;                  // wave-uniform: a scalar
        asm("" : "+s"(rowoff));
        const float* tp = reinterpret_cast<const float*>(base + rowoff + (cl << 2));
        v[u] = (fan_f2){tp[0], tp[1]};
        const unsigned nf = frac + mf;
        col += mi + (nf < frac ? 1 : 0);
        frac = nf;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc0 = fmaf(w0[u], v[u][0], acc0);
        acc1 = fmaf(w1[u], v[u][1], acc1);
      }
    }
    if (live_ray) part[(int64_t)b * nrays + ray] = acc0 + acc1;
    int nx = 0;
    if (lane == 0) nx = atomicAdd(&next_task, 1);
    task = __builtin_amdgcn_readfirstlane(nx);
  }
}

// sino[ray] = len * (the bands' sums, added in band order)
__global__ __launch_bounds__(256) void k_fan_bands_sum(const float* __restrict__ part, int nb, int64_t nrays,
                                                       const FanRay* __restrict__ rays, float* __restrict__ sino) {
  const int64_t ray = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (ray >= nrays) return;
  float t = 0.f;
  for (int b = 0; b < nb; ++b) t += part[(int64_t)b * nrays + ray];
  sino[ray] = rays[ray].len * t;
}

// Records of one apply: per view a row of nd + 2 FAN_RP records, FAN_RP zero records either side (a candidate index outside the
// detector then needs no test: it weighs nothing).  A record is the ray's table entry prepared for the pixel's side of the
// computation: a pixel weighs a candidate ray by the ray's position RELATIVE to the pixel, which lies within +-1.5 columns for every
// candidate, so the low 32 bits of the 2^-30 fixed-point position — a range of 4 columns — decide it exactly (the 64-bit
// multiply-add of the forward's absolute position and the reciprocal per candidate were a third of the adjoint's instructions):
//   x0m   = (X0 + min(M, 0)) mod 2^32   (the start of the step's interval at marching index 0; even: X0's class bit cleared)
//   inv32 = fan_inv32(M), NEGATIVE for a shallow ray (the class: one float compare; the weight takes |inv32|)
//   m, len_s = len * S[a][d]
constexpr int FAN_RP = 2;
struct FanRec {
  unsigned x0m;
  float inv32;
  int m;
  float len_s;
};
static_assert(sizeof(FanRec) == sizeof(FanRay), "the record array doubles as the forward's band partials: 16 bytes per ray");

// NEG (k_fan_adj_views): x0m and m stored as ~x0m and -m — the kernel then forms ~a of its position word a directly (see there)
template <bool NEG>
__global__ __launch_bounds__(256) void k_fan_adj_prep(const float* __restrict__ sino, int64_t ld_sino, const FanRay* __restrict__ rays,
                                                      FanRec* __restrict__ recs, int nd, int na) {
  const int ndp = nd + 2 * FAN_RP;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // over na x ndp
  if (i >= (int64_t)na * ndp) return;
  const int a = (int)(i / ndp), d = (int)(i - (int64_t)a * ndp) - FAN_RP;
  FanRec o{0u, 1.0f, 0, 0.f};
  if (d >= 0 && d < nd) {
    const int64_t k = (int64_t)a * nd + d;
    const FanRay q = rays[k];
    o.x0m = (q.x0_lo & ~1u) + (unsigned)(q.m < 0 ? q.m : 0);
    if (NEG) o.x0m = ~o.x0m;
    const float iv = fan_inv32(q.m);
    o.inv32 = (q.x0_lo & 1u) ? -iv : iv;
    o.m = NEG ? -q.m : q.m;
    o.len_s = q.len * sino[(int64_t)blockIdx.y * ld_sino + k];
  }
  recs[(int64_t)blockIdx.y * na * ndp + i] = o;
}

// One thread per pixel.  Per view: the detector coordinate uc of the pixel centre's projection and the half width w of the detector
// interval a ray must lie in to touch the pixel give the candidates dlo = ceil(uc - w) .. floor(uc + w): one or two nearly
// everywhere (NC = 2, decided at creation: three only next to the source) — the first two are always fetched (16-byte records
// through a buffer resource: view row in the scalar offset, 16 dlo in the vector offset, the second 16 bytes on) and weighed, each
// with weight 0 when it is not in the interval (it may lie more than two columns away, where the 32-bit relative position
// would alias), further ones in a rare loop; the view loop is unrolled by four so that the gathers of several views are in flight
// together (with a candidate loop every view waited for its own gather: 0.9 us per view and wave at four waves per SIMD).
// The kernel is bound by vector-instruction issue (PMC: 96 % busy): 76 instructions per pixel and view in round 3's form, ~58 here.
template <int NC>
__global__ __launch_bounds__(256) void k_fan_adj_march(float* __restrict__ img,
                                                       int64_t ld_img, int N, int nd, int na, float dsd, float reach,
                                                       const FanAngle* __restrict__ ang, const FanRec* __restrict__ recs) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)N * N) return;
  const int r = (int)(idx / N), c = (int)(idx - (int64_t)r * N);
  const float half = 0.5f * (float)N;
  const float px = (float)c + 0.5f - half, py = half - (float)r - 0.5f;     // pixel centre
  const int ndp = nd + 2 * FAN_RP;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(recs + (int64_t)blockIdx.y * na * ndp), 0, (unsigned)((int64_t)na * ndp * 16), 0x00020000);
  // marching index / picked index by class, the picked one already negated and shifted (rel = x0m + tt M - (want << 30))
  const unsigned nw_steep = 0u - ((unsigned)c << FAN_Q), nw_shallow = 0u - ((unsigned)r << FAN_Q);
  const float dsd_reach = dsd * reach;
  float acc = 0.f;
  struct Iv { float clo, hi; int dlo; };
  auto interval = [&](const FanAngle& g) -> Iv {
    // (only the candidate interval hangs on these numbers — every candidate is then weighed exactly — so the hardware reciprocal
    // does: 1 ulp against the 2 % + 0.01 of slack in `reach`)
    const float vx = px - g.sx, vy = py - g.sy;
    const float rden = __builtin_amdgcn_rcpf(fmaf(vx, g.nx, vy * g.ny));
    const float mag = dsd * rden;
    const float hx = fmaf(mag, vx, g.ex), hy = fmaf(mag, vy, g.ey);
    const float uc = fmaf(hx, g.uxs, hy * g.uys);
    const float w = fmaf(dsd_reach, rden, 0.01f);
    Iv v;
    v.clo = ceilf(uc - w);
    v.hi = uc + w;
    int dlo = (int)v.clo;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(v.dlo) : "v"(dlo), "n"(-FAN_RP), "s"(nd));     // dlo and dlo + 1 stay inside the padded row
    return v;
  };
  auto weigh = [&](const FanRec& q) -> float {
    const bool shallow = q.inv32 < 0.f;
    const unsigned tt = shallow ? (unsigned)c : (unsigned)r;                // marching index
    const unsigned nwant = shallow ? nw_shallow : nw_steep;
    // fan_step_at() on the position relative to the pixel, modulo 4 columns: rel = X0 + tt M + min(M, 0) - want   (2^-30 units)
    const unsigned rel = q.x0m + tt * (unsigned)q.m + nwant;
    float f;
    {
      const float fd = (float)(~(rel << (32 - FAN_Q)));                     // fan_split with |inv32| (the sign is the class)
      asm("v_fma_f32 %0, %1, |%2|, |%2| clamp" : "=v"(f) : "v"(fd), "v"(q.inv32));
    }
    // rel in [0, 2^30): the pixel is the ray's first column of the step (weight f); in [-2^30, 0): its second (1 - f); else none
    const float s = (int)rel >= 0 ? f : 1.f - f;
    const bool touches = (rel + (1u << FAN_Q)) < (2u << FAN_Q);
    return touches ? s * q.len_s : 0.f;
  };
  auto fetch = [&](int a, int d) -> FanRec {
    const u4f t = __builtin_bit_cast(u4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (d + FAN_RP) << 4, a * ndp * 16, 0));
    // (elements copied to scalars first: __builtin_bit_cast(float, t[k]) on an ext-vector ELEMENT reads element 0 whatever k is —
    //  hipcc / ROCm 7.2, met before in radon2d.hip's adj_gather)
    const unsigned t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
    return FanRec{t0, __builtin_bit_cast(float, t1), (int)t2, __builtin_bit_cast(float, t3)};
  };
  if (NC == 2) {
    constexpr int UA = 4;
    int a = 0;
    for (; a + UA <= na; a += UA) {
      FanRec q[UA][2];
      bool ok0[UA], ok1[UA];
#pragma unroll
      for (int u = 0; u < UA; ++u) {
        const Iv v = interval(ang[a + u]);
        // an EMPTY interval (2 w < 1: detector pitch coarser than a pixel's footprint) has no candidate at all: the ray at
        // ceil(uc - w) may then lie several columns away, where the 32-bit relative position of weigh() would alias
        ok0[u] = v.clo <= v.hi;
        ok1[u] = v.clo + 1.f <= v.hi;
        q[u][0] = fetch(a + u, v.dlo);
        q[u][1] = fetch(a + u, v.dlo + 1);
        if (v.clo + 2.f <= v.hi) {                                          // a third candidate and beyond: rare (pixels near the source)
          const int dhi = min((int)floorf(v.hi), nd - 1);
          for (int d = max((int)v.clo + 2, 0); d <= dhi; ++d) acc += weigh(fetch(a + u, d));
        }
      }
#pragma unroll
      for (int u = 0; u < UA; ++u) {
        acc += ok0[u] ? weigh(q[u][0]) : 0.f;
        acc += ok1[u] ? weigh(q[u][1]) : 0.f;
      }
    }
    for (; a < na; ++a) {
      const Iv v = interval(ang[a]);
      const int dhi = min((int)floorf(v.hi), nd - 1);
      for (int d = max((int)v.clo, 0); d <= dhi; ++d) acc += weigh(fetch(a, d));
    }
  } else {
    for (int a = 0; a < na; ++a) {
      const Iv v = interval(ang[a]);
      const int dhi = min((int)floorf(v.hi), nd - 1);
      for (int d = max((int)v.clo, 0); d <= dhi; ++d) acc += weigh(fetch(a, d));
    }
  }
  img[(int64_t)blockIdx.y * ld_img + idx] = acc;
}

// Round 6: the same gather with the per-view work taken off the pixel.  k_fan_adj_march spent ~58 vector instructions per pixel and view
// (96 % issue-bound): 17 on the locator, and per candidate 3 selects + 1 compare on the ray's class, an exec-mask branch around the
// weighing (two s_nop and the mask bookkeeping), 2 on the record's address and ~13 on the weight itself.  Here
//  (i)   the locator's constants are folded per view on the host (FanAdjView) and live in LDS (two broadcast ds_read_b128 per view): the
//        pixel's {den, lo-numerator} are TWO packed FMAs, then a reciprocal, two FMAs, ceil, a conversion, a difference: 10 instructions;
//  (ii)  a view's records are fetched through a buffer resource of THAT view's row: anything outside it reads zeros (a record of zeros
//        weighs nothing), so the detector index needs no clamp;
//  (iii) 70 % of the views (fan half angle 13 degrees: all but the views within it of a diagonal) hold rays of ONE class: marching index
//        and wanted index are chosen once per view by a scalar condition, not per candidate;
//  (iv)  the candidates are weighed without a branch — one outside the pixel's interval has its len_s replaced by 0 — and the weight
//        enters the sum by one FMA; DENSE geometries (no empty interval) skip the first candidate's test;
//  (v)   the records hold ~x0m and -m (k_fan_adj_prep<true>): with a = x0m + tt M - (want << 30) + 2^30 the position word of
//        k_fan_adj_march (first column of the step for a in [2^30, 2^31), second for a in [0, 2^30)), na = ~a comes out of the same
//        multiply-add, ~(a << 2) = (na << 2) | 3 is one v_lshl_or, "touches" is the sign of na and "second column" na >= 0xC0000000;
//  (vi)  tt M (32-bit, a quarter-rate v_mul_lo_u32) as two 24-bit multiplies on the halves of M (SDWA word selects) and a shift-add.
// The weights are k_fan_adj_march's to the bit (the same integers, the same clamped FMA); the sum differs by the FMA.
template <bool DENSE>
__global__ __launch_bounds__(256) void k_fan_adj_views(float* __restrict__ img, int64_t ld_img, int N, int nd, int na,
                                                       const FanAdjView* __restrict__ views, const int* __restrict__ view_cls,
                                                       const FanRec* __restrict__ recs) {
  extern __shared__ __attribute__((aligned(16))) float vlds[];              // na x 8 floats
  for (int i = threadIdx.x; i < 2 * na; i += 256)
    reinterpret_cast<f4r*>(vlds)[i] = reinterpret_cast<const f4r*>(views)[i];
  __syncthreads();
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)N * N) return;
  const int r = (int)(idx / N), c = (int)(idx - (int64_t)r * N);
  const f2v fc2 = {(float)c, (float)c}, fr2 = {(float)r, (float)r};
  const int ndp = nd + 2 * FAN_RP;
  const FanRec* __restrict__ rbase = recs + (int64_t)blockIdx.y * na * ndp;
  const unsigned nwn_steep = ((unsigned)c - 1u) << FAN_Q, nwn_shallow = ((unsigned)r - 1u) << FAN_Q;     // -((1 - want) << 30)
  float acc = 0.f;
  struct Loc { float t, clo, hi; int d; };
  auto locate = [&](int a) -> Loc {
    const f4r A = reinterpret_cast<const f4r*>(vlds)[2 * a], K = reinterpret_cast<const f4r*>(vlds)[2 * a + 1];
    const f2v P = __builtin_elementwise_fma(fc2, (f2v){A[0], A[1]}, __builtin_elementwise_fma(fr2, (f2v){A[2], A[3]}, (f2v){K[0], K[1]}));
    const float rden = __builtin_amdgcn_rcpf(P[0]);
    const float lo = fmaf(P[1], rden, K[2]);
    Loc L;
    L.clo = ceilf(lo);
    L.hi = fmaf(K[3], rden, lo);                            // (without the 0.02 of slack: it sits in the comparisons)
    L.t = L.clo - L.hi;                                     // <= 0.02: ceil(lo) is a candidate, <= -0.98: the next one too, <= -1.98: a third
    L.d = (int)L.clo;
    return L;
  };
  auto fetch = [&](int a, int d, int more) -> FanRec {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(rbase + (int64_t)a * ndp), 0, (unsigned)(ndp * 16), 0x00020000);
    const u4f t = __builtin_bit_cast(u4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((d + FAN_RP) << 4) + more * 16, 0, 0));
    const unsigned t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
    return FanRec{t0, __builtin_bit_cast(float, t1), (int)t2, __builtin_bit_cast(float, t3)};
  };
  // the weight of record q on this pixel times its len_s, added to acc; in: the candidate lies in the pixel's interval
  auto weigh = [&](const FanRec& q, unsigned tt, unsigned nwn, bool in) {
    unsigned p0, p1;
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(p0) : "v"(tt), "v"(q.m));
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(p1) : "v"(tt), "v"(q.m));
    unsigned na_ = q.x0m + nwn + p0;                                          // ~a, but for the high half of the product
    asm("v_lshl_add_u32 %0, %1, 16, %0" : "+v"(na_) : "v"(p1));
    float f;
    {
      const float fd = (float)((na_ << (32 - FAN_Q)) | 3u);                   // ~(a << 2): fan_split with |inv32| (the sign is the class)
      asm("v_fma_f32 %0, %1, |%2|, |%2| clamp" : "=v"(f) : "v"(fd), "v"(q.inv32));
    }
    const float s = na_ >= 0xC0000000u ? 1.f - f : f;
    const float w = (in && (int)na_ < 0) ? q.len_s : 0.f;
    acc = fmaf(s, w, acc);
  };
  auto weigh_any = [&](const FanRec& q, bool in) {                           // the ray's own class
    const bool shallow = q.inv32 < 0.f;
    weigh(q, shallow ? (unsigned)c : (unsigned)r, shallow ? nwn_shallow : nwn_steep, in);
  };
  constexpr int UA = 4;
  int a = 0;
  for (; a + UA <= na; a += UA) {
    FanRec q[UA][2];
    float t[UA];
    int cls[UA];
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const Loc L = locate(a + u);
      t[u] = L.t;
      cls[u] = view_cls[a + u];                                               // (scalar load)
      q[u][0] = fetch(a + u, L.d, 0);
      q[u][1] = fetch(a + u, L.d, 1);
      if (L.t <= -1.98f) {                                                    // a third candidate and beyond: rare (pixels near the source)
        const int dhi = min((int)floorf(L.hi + 0.02f), nd - 1);
        for (int d = max(L.d + 2, 0); d <= dhi; ++d) weigh_any(fetch(a + u, d, 0), true);
      }
    }
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const bool in0 = DENSE ? true : t[u] <= 0.02f, in1 = t[u] <= -0.98f;
      if (cls[u] != 2) {                                                      // one class for the whole view: a scalar choice
        const unsigned tt = cls[u] ? (unsigned)c : (unsigned)r, nwn = cls[u] ? nwn_shallow : nwn_steep;
        weigh(q[u][0], tt, nwn, in0);
        weigh(q[u][1], tt, nwn, in1);
      } else {
        weigh_any(q[u][0], in0);
        weigh_any(q[u][1], in1);
      }
    }
  }
  for (; a < na; ++a) {
    const Loc L = locate(a);
    const int dhi = min((int)floorf(L.hi + 0.02f), nd - 1);
    for (int d = max(L.d, 0); d <= dhi; ++d) weigh_any(fetch(a, d, 0), true);
  }
  img[(int64_t)blockIdx.y * ld_img + idx] = acc;
}

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
  static const bool siddon = getenv("TRK_FAN_SIDDON") != nullptr;       // the general pair, for comparison
  const bool march = im->rays && !siddon;
  if (!tr && march) {
    const int nb = ceil_div(im->N, 32);
    const int64_t padded = (int64_t)im->N * (im->N + 2 * FAN_PAD);
    const int64_t nrays = (int64_t)im->na * im->nd;
    // bands of the march: enough waves to hide the gathers (about 8 per SIMD), at most the 4 floats per ray that the adjoint's
    // record array (unused during a forward apply) has room for, bands of at least 64 steps and a multiple of 8
    static const int fb_env = getenv("TRK_FAN_FWD_BANDS") ? atoi(getenv("TRK_FAN_FWD_BANDS")) : 0;
    int nbands = fb_env > 0 ? fb_env : (int)((8 * 4 * (int64_t)cu_count() * 64 + nrays - 1) / nrays);
    if (nbands > 4) nbands = 4;
    if (nbands > im->N / 64) nbands = im->N / 64;
    if (nbands < 1) nbands = 1;
    const int band = ((im->N + nbands - 1) / nbands + 7) / 8 * 8;
    nbands = ceil_div(im->N, band);
    dim3 grid(ceil_div(nrays, 256), nbands);
    float* part = reinterpret_cast<float*>(im->recs);
    const bool no_bandres = getenv("TRK_FAN_NO_BANDRES") != nullptr;       // (read per call: tests switch it)
    if (im->band_part && !no_bandres && ldx >= (int64_t)im->N * im->N && (reinterpret_cast<uintptr_t>(x) & 15u) == 0 && (batch == 1 || ldx % 4 == 0)) {
      // small images: 64-row bands resident in LDS (k_fan_fwd_band), no padded copies
      const int rows = fb_rows(im->N), nbr = im->N / rows;
      const size_t lds_bytes = sizeof(float) * (size_t)rows * (im->N + 2 * FB_PAD) + 16;
      int nslice = (cu_count() + nbr) / (2 * nbr);
      if (nslice < 1) nslice = 1;
      static bool attr_set = false;
      if (!attr_set) {
        TRK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fan_fwd_band), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
      }
      for (int b = 0; b < batch; ++b) {
        hipLaunchKernelGGL(k_fan_fwd_band, dim3((unsigned)(2 * nbr * nslice)), dim3(FB_NT), lds_bytes, s, x + (int64_t)b * ldx, im->band_part,
                           im->N, nrays, im->rays, im->cls_list, im->n_steep, nslice, rows);
        hipLaunchKernelGGL(k_fan_bands_sum, dim3(ceil_div(nrays, 256)), dim3(256), 0, s, im->band_part, nbr, nrays, im->rays, y + (int64_t)b * ldy);
      }
    } else
    for (int b = 0; b < batch; ++b) {                                      // one pair of padded copies per handle: columns go one by one
      hipLaunchKernelGGL(k_fan_pad_copies, dim3(nb, nb, 1), dim3(256), 0, s, x + (int64_t)b * ldx, ldx, im->xT, im->xT + padded, im->N);
      hipLaunchKernelGGL(k_fan_fwd_march, grid, dim3(256), 0, s, im->xT, padded, y + (int64_t)b * ldy, im->N,
                         nrays, im->rays, band, part);
      if (nbands > 1)
        hipLaunchKernelGGL(k_fan_bands_sum, dim3(ceil_div(nrays, 256)), dim3(256), 0, s, part, nbands, nrays, im->rays, y + (int64_t)b * ldy);
    }
  } else if (!tr) {
    dim3 grid(ceil_div((int64_t)im->na * im->nd, 256), batch);
    hipLaunchKernelGGL(k_fan_fwd, grid, dim3(256), 0, s, x, ldx, y, ldy, im->N, im->nd, im->na, im->ang_dev);
  } else if (march) {
    const int64_t nrays = (int64_t)im->na * im->nd;
    dim3 grid(ceil_div((int64_t)im->N * im->N, 256), 1);
    for (int b = 0; b < batch; ++b) {                                      // one record array per handle: columns go one by one
      const bool old_form = getenv("TRK_FAN_ADJ_MARCH2") != nullptr;            // round 3-5's kernel, for comparison (read per call: tests switch it)
      const bool by_views = im->max_cand <= 3 && im->views && !old_form && (size_t)im->na * sizeof(FanAdjView) <= 48 * 1024;
      const dim3 pgrid(ceil_div((int64_t)im->na * (im->nd + 2 * FAN_RP), 256), 1);
      if (by_views)
        hipLaunchKernelGGL(k_fan_adj_prep<true>, pgrid, dim3(256), 0, s, x + (int64_t)b * ldx, ldx, im->rays, reinterpret_cast<FanRec*>(im->recs), im->nd, im->na);
      else
        hipLaunchKernelGGL(k_fan_adj_prep<false>, pgrid, dim3(256), 0, s, x + (int64_t)b * ldx, ldx, im->rays, reinterpret_cast<FanRec*>(im->recs), im->nd, im->na);
      if (by_views) {
        const size_t lds = (size_t)im->na * sizeof(FanAdjView);
        if (im->dense)
          hipLaunchKernelGGL(k_fan_adj_views<true>, grid, dim3(256), lds, s, y + (int64_t)b * ldy, ldy, im->N, im->nd, im->na, im->views,
                             im->view_cls, reinterpret_cast<const FanRec*>(im->recs));
        else
          hipLaunchKernelGGL(k_fan_adj_views<false>, grid, dim3(256), lds, s, y + (int64_t)b * ldy, ldy, im->N, im->nd, im->na, im->views,
                             im->view_cls, reinterpret_cast<const FanRec*>(im->recs));
      } else if (im->max_cand <= 3)
        hipLaunchKernelGGL(k_fan_adj_march<2>, grid, dim3(256), 0, s, y + (int64_t)b * ldy, ldy, im->N, im->nd, im->na, im->dsd,
                           im->reach, im->ang_dev, reinterpret_cast<const FanRec*>(im->recs));
      else
        hipLaunchKernelGGL(k_fan_adj_march<0>, grid, dim3(256), 0, s, y + (int64_t)b * ldy, ldy, im->N, im->nd, im->na, im->dsd,
                           im->reach, im->ang_dev, reinterpret_cast<const FanRec*>(im->recs));
    }
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
  if (im->rays) (void)hipFree(im->rays);
  if (im->recs) (void)hipFree(im->recs);
  if (im->xT) (void)hipFree(im->xT);
  if (im->cls_list) (void)hipFree(im->cls_list);
  if (im->band_part) (void)hipFree(im->band_part);
  if (im->views) (void)hipFree(im->views);
  if (im->view_cls) (void)hipFree(im->view_cls);
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
    g.ex = g.sx - g.d0x;                                 // (fp32 differences of the fp32 fields: what the kernel used to form itself)
    g.ey = g.sy - g.d0y;
    g.uxs = (float)(ct / det_pitch);
    g.uys = (float)(st / det_pitch);
    h[a] = g;
  }
  auto* im = new FanImpl{N, n_det, n_ang, (float)(sod + odd), (float)det_pitch, nullptr, nullptr, nullptr, nullptr, 0.f, 1 << 30, nullptr, 0, nullptr,
                         nullptr, nullptr, 0};
  hipError_t e = hipMalloc(&im->ang_dev, sizeof(FanAngle) * n_ang);
  if (e == hipSuccess) e = hipMemcpy(im->ang_dev, h.data(), sizeof(FanAngle) * n_ang, hipMemcpyHostToDevice);
  // row-march table: needs every ray to cross the whole image, i.e. source and detector outside its circumscribed circle
  if (e == hipSuccess && odd > 0.7072 * N) {
    const double half = 0.5 * N;
    std::vector<FanRay> rt((size_t)n_ang * n_det);
    for (int a = 0; a < n_ang; ++a) {
      const double ct = std::cos(angles[a]), st = std::sin(angles[a]);
      const double sx = sod * st, sy = -sod * ct;
      for (int d = 0; d < n_det; ++d) {
        const double off = (d - 0.5 * (n_det - 1)) * det_pitch;
        const double dx = -odd * st + off * ct - sx, dy = odd * ct + off * st - sy;
        FanRay q;
        double X0, M;
        int shallow;
        if (std::fabs(dy) >= std::fabs(dx)) {              // steep: X(Y) = X0 + Y M over rows Y = half - y
          const double k = dx / dy;
          X0 = sx + half + (half - sy) * k;
          M = -k;
          shallow = 0;
        } else {                                           // shallow: Y(X) = Y0 + X My over columns X = x + half
          const double k = dy / dx;
          X0 = half - sy + (half + sx) * k;
          M = -k;
          shallow = 1;
        }
        const long long x0 = (std::llround(std::ldexp(X0, FAN_Q)) & ~1LL) | (long long)shallow;
        q.x0_lo = (unsigned)(x0 & 0xFFFFFFFFLL);
        q.x0_hi = (int)(x0 >> 32);
        q.m = (int)std::llround(std::ldexp(M, FAN_Q));
        q.len = (float)std::sqrt(1.0 + M * M);
        rt[(size_t)a * n_det + d] = q;
      }
    }
    e = hipMalloc(&im->rays, sizeof(FanRay) * rt.size());
    if (e == hipSuccess) e = hipMemcpy(im->rays, rt.data(), sizeof(FanRay) * rt.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess && N % FB_ROWS == 0 && N >= 2 * FB_ROWS && N <= FB_NMAX) {
      // the band-resident forward: rays by class, in table order
      std::vector<int> lst;
      lst.reserve(rt.size());
      for (size_t i = 0; i < rt.size(); ++i) if (!(rt[i].x0_lo & 1u)) lst.push_back((int)i);
      im->n_steep = (int)lst.size();
      for (size_t i = 0; i < rt.size(); ++i) if (rt[i].x0_lo & 1u) lst.push_back((int)i);
      e = hipMalloc(&im->cls_list, sizeof(int) * lst.size());
      if (e == hipSuccess) e = hipMemcpy(im->cls_list, lst.data(), sizeof(int) * lst.size(), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMalloc(&im->band_part, sizeof(float) * (size_t)(N / fb_rows(N)) * rt.size());
    }
    if (e == hipSuccess) e = hipMalloc(&im->recs, sizeof(FanRay) * (size_t)n_ang * (n_det + 2 * 2));   // rows padded for the adjoint (FAN_RP)
    const size_t padded = (size_t)N * (N + 2 * FAN_PAD);                 // two padded copies; the pad columns stay zero for good
    if (e == hipSuccess) e = hipMalloc(&im->xT, sizeof(float) * 2 * padded);
    if (e == hipSuccess) e = hipMemset(im->xT, 0, sizeof(float) * 2 * padded);
    // a pixel's half diagonal seen from the source, on a flat detector: (sqrt(2)/2) mag / pitch / cos^2(fan half angle), plus slack
    const double tan_max = 0.5 * n_det * det_pitch / (sod + odd);
    im->reach = (float)(0.7072 / det_pitch * (1.0 + tan_max * tan_max) * 1.02);
    const double mag_max = (sod + odd) / (sod - 0.7072 * N);              // the pixel nearest to the source
    im->max_cand = (int)std::floor(2.0 * ((double)im->reach * mag_max * 1.001 + 0.01)) + 1;
    // the adjoint's per-view constants (FanAdjView) and whether any pixel's interval can be empty
    if (e == hipSuccess) {
      const double dsd = sod + odd, reach = (double)im->reach;
      std::vector<FanAdjView> vw(n_ang);
      std::vector<int> vcls(n_ang);
      for (int a = 0; a < n_ang; ++a) {
        const double ct = std::cos(angles[a]), st = std::sin(angles[a]);
        const double sx = sod * st, sy = -sod * ct, nx = -st, ny = ct;
        const double d0x = -odd * st - 0.5 * (n_det - 1) * det_pitch * ct, d0y = odd * ct - 0.5 * (n_det - 1) * det_pitch * st;
        const double uxs = ct / det_pitch, uys = st / det_pitch;
        const double cx = 0.5 - half - sx, cy = half - 0.5 - sy;         // pixel centre minus source = (c + cx, -r + cy)
        FanAdjView v{};
        v.nx = (float)nx;
        v.nny = (float)(-ny);
        v.kd = (float)(cx * nx + cy * ny);
        v.Ux = (float)(dsd * uxs);
        v.Uy = (float)(-dsd * uys);
        v.Uklo = (float)(dsd * (cx * uxs + cy * uys) - dsd * reach);
        v.kelo = (float)((sx - d0x) * uxs + (sy - d0y) * uys - 0.01);
        v.dr2 = (float)(2.0 * dsd * reach);
        int n_sh = 0;
        for (int d = 0; d < n_det; ++d) n_sh += (int)(rt[(size_t)a * n_det + d].x0_lo & 1u);
        vcls[a] = n_sh == 0 ? 0 : (n_sh == n_det ? 1 : 2);
        vw[a] = v;
      }
      const double den_max = sod + 0.7072 * N;                           // the pixel farthest from the source along the central ray
      im->dense = 2.0 * (dsd * reach / den_max) >= 1.0 ? 1 : 0;          // (+ 0.02 of slack in the kernel's interval on top)
      e = hipMalloc(&im->views, sizeof(FanAdjView) * n_ang);
      if (e == hipSuccess) e = hipMemcpy(im->views, vw.data(), sizeof(FanAdjView) * n_ang, hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMalloc(&im->view_cls, sizeof(int) * n_ang);
      if (e == hipSuccess) e = hipMemcpy(im->view_cls, vcls.data(), sizeof(int) * n_ang, hipMemcpyHostToDevice);
    }
  }
  if (e != hipSuccess) {
    trk_op tmp{7, 0, 0, im, nullptr, nullptr, nullptr, 0};
    fan_destroy(&tmp);
    return fail(TRK_EHIP, "trk_fanbeam2d_create: %s", hipGetErrorString(e));
  }
  *out = new trk_op{7, (int64_t)n_ang * n_det, (int64_t)N * N, im, fan_apply, fan_destroy, nullptr, 0};
  return TRK_OK;
}
