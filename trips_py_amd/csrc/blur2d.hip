// blur2d.hip — 2-D blur with reflective ('reflect' = half-sample symmetric) boundary for gfx950.
//
// Replaces scipy.ndimage.convolve(X.reshape(nx,ny), PSF, mode='reflect') and its flipped-PSF "transpose"
// (trips/test_problems/Deblurring2D.py:66-73):
//     y[i,j] = sum_{a,b} PSF[a,b] * xr[i + kh/2 - a, j + kw/2 - b]
// written here as a correlation  y[i,j] = sum_{a',b'} c[a',b'] * xr[i - T + a', j - L + b'],
//     c[a',b'] = PSF[kh-1-a', kw-1-b'],  T = kh-1-kh/2,  L = kw-1-kw/2.
//
// Kernel plan (HBM-bound: 8 bytes of traffic per pixel):
//   * k_blur_strip<KH,KW,SEP>: a workgroup (256 threads = 4 waves) owns a 128-column strip of a band of rows and SLIDES
//     DOWN it in steps of 32 rows.  The KH-1 halo rows shared by consecutive steps stay in an LDS ring of 32+KH-1 rows,
//     so every input row is fetched once per strip (no vertical re-reads); only the KW-1 halo columns between
//     neighbouring strips are read twice (6 % at 9x9, served by L2).  While a step is being computed from LDS, the 32
//     new rows of the next step are already in flight from HBM into registers (16-byte coalesced loads) and are written
//     to the ring after the step's barrier: loads overlap compute, 2 barriers per step, 22 KB of LDS per workgroup.
//     Each thread owns a 4 x 4 register block of outputs; per staged row it reads 12 floats (3 x ds_read_b128), forms the
//     row-filtered values in registers and scatters them into its 4 output rows (separable PSFs: 9+... FMAs; general
//     PSFs: the direct KH*KW form from the same 12 floats).  Stores are 16-byte coalesced.  Optionally sum(y^2) is
//     accumulated per thread in fp64 over all steps and reduced once per workgroup (wave64 shuffles).
//   * grid = strips x row bands, sized to ~4 workgroups per CU so that a 4096^2 image is split evenly (no tail).
//   * k_blur_generic: any PSF size (even, rectangular, longer than the image: repeated reflection), no tiling.
#include "trk_internal.h"
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

using namespace trk;

namespace {

constexpr int NT = 256;
constexpr int TW = 128;  // output tile width  (32 lanes x float4)
constexpr int TH = 32;   // output tile height (8 thread rows x 4)

typedef float f4 __attribute__((ext_vector_type(4)));  // one 16-byte register quad: keeps loads/stores as b128

__host__ __device__ constexpr int rup4(int v) { return (v + 3) & ~3; }

__device__ __forceinline__ int reflect(int i, int n) {
  // half-sample symmetric extension, any distance; in-range indices (the common case) skip the division
  if ((unsigned)i < (unsigned)n) return i;
  const int p = 2 * n;
  i %= p;
  if (i < 0) i += p;
  return (i >= n) ? (p - 1 - i) : i;
}

struct BlurImpl {
  int nx, ny, kh, kw;
  bool separable;
  bool tiled;          // a k_blur_strip instantiation exists for (kh,kw)
  float* w_dev[2];     // [kh*kw] correlation weights: 0 forward, 1 "transpose" (flipped PSF)
  float* sep_dev[2];   // [kw row weights | kh column weights]
};

// ------------------------------------------------------------------------------------------------ strip kernel
// min waves per SIMD asked of the register allocator: 4 (= 4 workgroups per CU) for the separable kernels up to 9x9,
// 2 for the register-hungry direct / large-PSF forms
template <int KH, int KW, bool SEP, bool SUMSQ>
__global__ __launch_bounds__(NT, (SEP && KH <= 9) ? 4 : 2) void k_blur_strip(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                   int64_t ldy, int nx, int ny, const float* __restrict__ wts,
                                                   double* __restrict__ partials, int strips_x, int rows_per_band) {
  constexpr int T = KH - 1 - KH / 2;   // halo above
  constexpr int L = KW - 1 - KW / 2;   // halo left
  constexpr int R = KW / 2;            // halo right
  constexpr int LP = rup4(L), RP = rup4(R);
  constexpr int SW = TW + LP + RP;     // staged row length (floats, multiple of 4)
  constexpr int RING = TH + KH - 1;    // staged rows per step
  constexpr int OFF = LP - L;          // first needed column inside the 16-byte aligned read
  constexpr int NV = (OFF + KW + 3 + 3) / 4;  // float4s covering 4 outputs' taps
  constexpr int SW4 = SW / 4;
  constexpr int NPF = (TH * SW4 + NT - 1) / NT;  // float4 prefetched per thread per step

  __shared__ f4 S4[RING * SW4];
  float* const S = reinterpret_cast<float*>(S4);
  __shared__ double red[NT / 64];

  const int sj = blockIdx.x % strips_x, band = blockIdx.x / strips_x;
  const int j0 = sj * TW;
  const int i_begin = band * rows_per_band;
  const int i_end = (i_begin + rows_per_band < nx) ? i_begin + rows_per_band : nx;
  x += (int64_t)blockIdx.y * ldx;
  y += (int64_t)blockIdx.y * ldy;

  // staged row `rel` is image row (i_begin - T + rel), reflected; it lives in ring slot rel % RING.
  // With ny % 4 == 0 and an aligned base every 4-column group of a staged row is either wholly inside the image
  // (one 16-byte load) or wholly outside (halo of the first / last strip: four reflected scalar loads).
  const bool fast = ((ny & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0);

  auto load_group = [&](int rel, int c4) -> f4 {   // 4 staged columns of staged row `rel`
    const int gi = reflect(i_begin - T + rel, nx);
    const int gc = j0 - LP + 4 * c4;
    const float* row = x + (int64_t)gi * ny;
    if (fast && gc >= 0 && gc + 3 < ny) return *reinterpret_cast<const f4*>(row + gc);
    return (f4){row[reflect(gc, ny)], row[reflect(gc + 1, ny)], row[reflect(gc + 2, ny)], row[reflect(gc + 3, ny)]};
  };

  auto load_direct = [&](int rel0, int count) {   // global -> LDS, rows rel0 .. rel0+count-1
    for (int idx = threadIdx.x; idx < count * SW4; idx += NT) {
      const int r = idx / SW4, c4 = idx - r * SW4;
      S4[((rel0 + r) % RING) * SW4 + c4] = load_group(rel0 + r, c4);
    }
  };

  float wr[SEP ? KW : 1], wc[SEP ? KH : 1];
  if (SEP) {
#pragma unroll
    for (int b = 0; b < KW; ++b) wr[b] = wts[b];
#pragma unroll
    for (int a = 0; a < KH; ++a) wc[a] = wts[KW + a];
  }

  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8 threads; thread block of 4 rows x 4 cols
  const int gj = j0 + 4 * tx;
  // block-uniform: every thread's 4 columns are inside the image and 16-byte aligned
  const bool vec_ok = ((ny & 3) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15u) == 0) && (j0 + TW <= ny);
  double ss = 0.0;

  load_direct(0, RING);
  int rb = 0;  // ring slot of the step's first staged row
  for (int i0 = i_begin; i0 < i_end; i0 += TH) {
    const int rel_base = i0 - i_begin;
    const bool has_next = (i0 + TH < i_end);
    __syncthreads();  // the step's RING rows are in LDS

    // ---- next step's 32 new rows: HBM -> registers, in flight during the compute below
    f4 pf[NPF];
#pragma unroll
    for (int q = 0; q < NPF; ++q) pf[q] = (f4){0.f, 0.f, 0.f, 0.f};
    if (has_next) {
#pragma unroll
      for (int q = 0; q < NPF; ++q) {
        const int idx = threadIdx.x + q * NT;
        if (idx < TH * SW4) {
          const int r = idx / SW4, c4 = idx - r * SW4;
          pf[q] = load_group(rel_base + RING + r, c4);
        }
      }
    }

    // ---- compute the 4 x 4 output block from LDS
    float acc[4][4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[rr][c] = 0.f;
#pragma unroll
    for (int a = 0; a < KH + 3; ++a) {
      int slot = rb + 4 * ty + a;
      if (slot >= RING) slot -= RING;
      float v[NV * 4];
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const f4 t = S4[slot * SW4 + tx + q];
        v[4 * q] = t[0];
        v[4 * q + 1] = t[1];
        v[4 * q + 2] = t[2];
        v[4 * q + 3] = t[3];
      }
      if (SEP) {
        float h[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < KW; ++b)
#pragma unroll
          for (int c = 0; c < 4; ++c) h[c] = fmaf(wr[b], v[OFF + c + b], h[c]);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          if (a - rr >= 0 && a - rr < KH) {
            const float w = wc[a - rr];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[rr][c] = fmaf(w, h[c], acc[rr][c]);
          }
        }
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          if (a - rr >= 0 && a - rr < KH) {
#pragma unroll
            for (int b = 0; b < KW; ++b) {
              const float w = wts[(a - rr) * KW + b];
#pragma unroll
              for (int c = 0; c < 4; ++c) acc[rr][c] = fmaf(w, v[OFF + c + b], acc[rr][c]);
            }
          }
        }
      }
    }

    // ---- store (16-byte coalesced when the strip is aligned and fully inside: a block-uniform branch)
    if (vec_ok) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int gi = i0 + 4 * ty + rr;
        if (gi < i_end) {
          *reinterpret_cast<f4*>(y + (int64_t)gi * ny + gj) = (f4){acc[rr][0], acc[rr][1], acc[rr][2], acc[rr][3]};
          if (SUMSQ)
            ss += (double)acc[rr][0] * acc[rr][0] + (double)acc[rr][1] * acc[rr][1] +
                  (double)acc[rr][2] * acc[rr][2] + (double)acc[rr][3] * acc[rr][3];
        }
      }
    } else {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int gi = i0 + 4 * ty + rr;
        if (gi < i_end) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (gj + c < ny) {
              y[(int64_t)gi * ny + gj + c] = acc[rr][c];
              if (SUMSQ) ss += (double)acc[rr][c] * acc[rr][c];
            }
        }
      }
    }

    if (has_next) {
      __syncthreads();  // every wave is done reading the rows about to be overwritten
#pragma unroll
      for (int q = 0; q < NPF; ++q) {
        const int idx = threadIdx.x + q * NT;
        if (idx < TH * SW4) {
          const int r = idx / SW4, c4 = idx - r * SW4;
          S4[((rel_base + RING + r) % RING) * SW4 + c4] = pf[q];
        }
      }
      rb += TH;
      if (rb >= RING) rb -= RING;
    }
  }
  if (SUMSQ) {
    ss = block_sum<NT>(ss, red);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = ss;
  }
}

// ------------------------------------------------------------------------------------------------ sliding kernel
// Separable PSFs with halos <= 4 columns (KW <= 9): no LDS, no barriers.  One wave owns a 256-column span (4 columns
// per lane) of a band of rows and marches down it.  Per staged row a lane issues three 16-byte buffer loads (its own
// 4 columns and the 4 to the left / right: L1/L2 hits of the neighbour lanes' lines), D rows ahead of use; the row is
// filtered horizontally in registers (KW taps from the 12 values), and the result is scattered with packed FMAs into
// KH rolling output-row accumulators (output o += wc[a] * h[o+a]); the oldest accumulator is complete and is stored
// with one 16-byte coalesced store.  Row addresses are wave-uniform (SGPR soffset of the buffer instruction), column
// offsets are per-lane constants: the inner loop has no address arithmetic in VALU.  The loop is unrolled by
// U = lcm(KH, D) so that every ring index is a compile-time constant (registers, not scratch).
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__host__ __device__ constexpr int gcd_c(int a, int b) { return b == 0 ? a : gcd_c(b, a % b); }
__host__ __device__ constexpr int lcm_c(int a, int b) { return a / gcd_c(a, b) * b; }
__host__ __device__ constexpr int pmod(int a, int m) { return ((a % m) + m) % m; }

constexpr int SPAN = 256;  // columns per wave

// FUSE: the operand is formed on load as  comb = x + cb * x2,  cb = sign * S(num) / S(den)  (device scalars, possibly
// still block partials), and the rows a band OWNS are also written to `comb` (never aliasing x / x2: neighbouring bands
// read each other's halo rows).  This is how CGLS's  p = t + (gamma/gamma_old) p  and  r = r - beta w  ride along with
// the blur that consumes them (CGLS.py:67-68,72 + :60) instead of being separate passes over memory.
struct SlideFuse {
  const float* x2;
  float* comb;
  double sign;
  ScalarSrc num, den;
};

// EPI: the output is combined on its way out,  y = a * (A x) + b * z  (trk_op_apply_axpby: MMGKS's A x - b, a Golub-Kahan half
// step) — trk_axpby's arithmetic on the finished row, fmaf(a, o, b * z), so the result equals apply followed by trk_axpby to the
// bit; z's rows travel D outputs ahead of their use like the operand's.  SUMSQ then sums the combined output.
struct SlideEpi {
  const float* z;      // NULL: y = a * (A x)
  Coef a, b;
};

template <int KH, int KW, int D, bool SUMSQ, bool FUSE, bool EPI = false>
__global__ __launch_bounds__(64) void k_blur_slide(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                   int64_t ldy, int nx, int ny, const float* __restrict__ wts,
                                                   double* __restrict__ partials, int spans_x, int nbands,
                                                   int rows_per_band, SlideFuse fz, int nt_store, SlideEpi ep = SlideEpi{}) {
  constexpr int T = KH - 1 - KH / 2;
  constexpr int Lh = KW - 1 - KW / 2;
  constexpr int OFFC = 4 - Lh;          // v[] index of tap 0 of output column 0
  constexpr int U = lcm_c(KH, D);
  static_assert(Lh <= 4 && KW / 2 <= 4, "sliding kernel handles halos of at most one 4-column group");

  // XCD-aware placement: ids are dealt round-robin over the 8 XCDs; give each XCD a contiguous block of bands so that
  // horizontally and vertically adjacent waves (which share halo lines) share an L2.
  int band, span;
  {
    // XCD x receives the ids congruent to x mod 8 — (total - x + 7) / 8 of them; it takes that many consecutive tiles of
    // the row-major (band, span) order, starting where the lower XCDs' shares end (any band count, no remainder rule)
    const int id = blockIdx.x, total = nbands * spans_x;
    const int xcd = id & 7, local = id >> 3;
    const int t = xcd * (total >> 3) + min(xcd, total & 7) + local;
    band = t / spans_x;
    span = t - band * spans_x;
  }
  x += (int64_t)blockIdx.y * ldx;
  y += (int64_t)blockIdx.y * ldy;
  const int i_begin = band * rows_per_band;
  const int i_end = (i_begin + rows_per_band < nx) ? i_begin + rows_per_band : nx;
  const int lane = threadIdx.x;
  const int c0 = span * SPAN + 4 * lane;
  const bool active = c0 < ny;
  const int cc = active ? c0 : ny - 4;                 // clamped: every load address is valid
  const int cl = (cc >= 4) ? cc - 4 : 0;
  const int cr = (cc + 8 <= ny) ? cc + 4 : ny - 4;
  const bool ledge = (c0 == 0), redge = (c0 + 4 == ny);
  const bool edge_span = (span == 0) || ((span + 1) * SPAN >= ny);   // wave-uniform

  const unsigned img_bytes = (unsigned)nx * (unsigned)ny * 4u;
  const auto rin = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, img_bytes, 0x00020000);
  const auto rout = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, img_bytes, 0x00020000);
  const auto rin2 = __builtin_amdgcn_make_buffer_rsrc((void*)(FUSE ? fz.x2 : x), 0, img_bytes, 0x00020000);
  const auto rcomb = __builtin_amdgcn_make_buffer_rsrc((void*)(FUSE ? fz.comb : y), 0, img_bytes, 0x00020000);
  float cb = 0.f;
  if (FUSE) cb = (fz.sign == 0.0) ? 0.f : (float)(fz.sign * scalar_from_wave(fz.num, threadIdx.x) / scalar_from_wave(fz.den, threadIdx.x));
  const int vl = cl * 4, vr = cr * 4;
  const int vc = cc * 4;
  const int vst = c0 * 4;

  // Odd bands march UPWARD (from their bottom halo to their top), even bands downward: the halo rows two neighbouring
  // bands share are then read by both at (nearly) the same time, at the start or at the end of their marches, and
  // the second reader hits in the XCD's L2 instead of fetching the rows again (measured fabric reads 1.13x -> ~1.0x
  // of the image at 64-row bands).  Marching up = the same recurrence on the reversed row sequence with the column
  // weights reversed.
  const bool up = (band & 1) != 0;                     // wave-uniform
  float wr[KW], wc[KH];
#pragma unroll
  for (int b = 0; b < KW; ++b) wr[b] = wts[b];
#pragma unroll
  for (int a = 0; a < KH; ++a) wc[a] = up ? wts[KW + KH - 1 - a] : wts[KW + a];

  const int band_rows = i_end - i_begin;
  const int total = ((band_rows + KH - 1 + U - 1) / U) * U;   // staged rows processed (multiple of U)
  const int rowbytes = ny * 4;
  // a band whose staged rows all lie inside the image needs no row reflection: the row offset is linear in t
  // staged row t is image row  first + dir*t  (down: from i_begin - T ; up: from i_end - 1 + KH/2)
  const int dir = up ? -1 : 1;
  const int first = up ? (i_end - 1 + KH / 2) : (i_begin - T);
  const int last = first + dir * (total - 1);
  const bool interior = (first >= 0) && (first < nx) && (last >= 0) && (last < nx);   // wave-uniform
  // finished output o (counted in marching order) is image row  ofirst + dir*o
  const int ofirst = up ? (i_end - 1) : i_begin;

  f4 pL[D], pC[D], pR[D];
  f4 qL[FUSE ? D : 1], qC[FUSE ? D : 1], qR[FUSE ? D : 1];
  auto issue = [&](int t, int slot) {
    int gi = first + dir * t;
    if (!interior) gi = reflect(gi, nx);
    const int so = gi * rowbytes;
    pL[slot] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rin, vl, so, 0));
    pC[slot] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rin, vc, so, 0));
    pR[slot] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rin, vr, so, 0));
    if (FUSE) {
      qL[slot] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rin2, vl, so, 0));
      qC[slot] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rin2, vc, so, 0));
      qR[slot] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rin2, vr, so, 0));
    }
  };
#pragma unroll
  for (int d = 0; d < D; ++d) issue(d, d);
  // EPI: coefficients (device scalars, wave-uniform) and the ring of z rows, output o in slot o % D
  float ea = 1.f, eb = 0.f;
  f4 zq[EPI ? D : 1];
  const auto rz = __builtin_amdgcn_make_buffer_rsrc((void*)((EPI && ep.z) ? ep.z : x), 0, img_bytes, 0x00020000);
  const bool has_z = EPI && ep.z != nullptr;
  auto issue_z = [&](int o, int slot) {
    const int oc = o < band_rows - 1 ? o : band_rows - 1;                 // (beyond the band: any valid row, never used)
    zq[slot] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rz, vc, (ofirst + dir * oc) * rowbytes, 0));
  };
  if (EPI) {
    ea = (float)coef_eval(ep.a);
    eb = has_z ? (float)coef_eval(ep.b) : 0.f;
    if (has_z) {
#pragma unroll
      for (int d = 0; d < D; ++d) issue_z(d, d);
    }
  }

  f2 acc[KH][2];
  double ss = 0.0;
  float qacc = 0.f;

  // One block of U staged rows.  GUARD = false is the steady state: every row refills its prefetch slot and stores one
  // finished output row, with no bounds tests (see the loop split below).
  auto block = [&](int t0, auto guard_tag) {
    constexpr bool GUARD = decltype(guard_tag)::value;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u;
      const int slot = u % D;
      f4 Lv = pL[slot], Cv = pC[slot], Rv = pR[slot];
      if (FUSE) {
        Lv = Lv + cb * qL[slot];
        Cv = Cv + cb * qC[slot];
        Rv = Rv + cb * qR[slot];
        // the combined operand of the rows this band owns goes back to memory (16-byte coalesced)
        const int graw = first + dir * t;
        if (graw >= i_begin && graw < i_end && active)                                   // first test is wave-uniform
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, Cv), rcomb, vst + graw * rowbytes, 0, 0);
      }
      if (!GUARD || t + D < total) issue(t + D, slot);  // refill the slot D rows ahead
      if (edge_span) {                                  // reflect across the image's left / right border
        const f4 rev = (f4){Cv[3], Cv[2], Cv[1], Cv[0]};
        if (ledge) Lv = rev;
        if (redge) Rv = rev;
      }
      const float v[12] = {Lv[0], Lv[1], Lv[2], Lv[3], Cv[0], Cv[1], Cv[2], Cv[3], Rv[0], Rv[1], Rv[2], Rv[3]};
      float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f;
#pragma unroll
      for (int b = 0; b < KW; ++b) {
        h0 = fmaf(wr[b], v[OFFC + b], h0);
        h1 = fmaf(wr[b], v[OFFC + b + 1], h1);
        h2 = fmaf(wr[b], v[OFFC + b + 2], h2);
        h3 = fmaf(wr[b], v[OFFC + b + 3], h3);
      }
      const f2 hlo = {h0, h1}, hhi = {h2, h3};
      // scatter into the rolling accumulators: output o = t - a gets wc[a] * h_t
#pragma unroll
      for (int a = 0; a < KH; ++a) {
        const int k = pmod(u - a, KH);
        if (a == 0) {
          acc[k][0] = wc[0] * hlo;
          acc[k][1] = wc[0] * hhi;
        } else {
          acc[k][0] = wc[a] * hlo + acc[k][0];
          acc[k][1] = wc[a] * hhi + acc[k][1];
        }
      }
      // output o = t - (KH-1) is complete
      const int o = t - (KH - 1);
      if (!GUARD || (o >= 0 && o < band_rows)) {        // uniform
        const int kd = pmod(u - (KH - 1), KH);
        f4 out = (f4){acc[kd][0][0], acc[kd][0][1], acc[kd][1][0], acc[kd][1][1]};
        if (EPI) {
          const int zs = pmod(u - (KH - 1), D);                             // = o % D: U is a multiple of D
          if (has_z) {                                                       // grid-uniform
            const f4 zv = zq[zs];
#pragma unroll
            for (int e = 0; e < 4; ++e) out[e] = fmaf(ea, out[e], eb * zv[e]);
            issue_z(o + D, zs);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) out[e] = ea * out[e];
          }
        }
        // NOTE the row offset goes into the VGPR offset, not the SGPR soffset: with an SGPR soffset hipcc (ROCm 7.2)
        // emits no wait state between a >64-bit buffer store and a VALU overwrite of its data registers, and on
        // gfx950 the last dword of the store was then observed corrupted (lanes 12-15 of each row of 16).
        if (active) {
          // aux = 2: non-temporal store, for images too large for the next kernel to find the output cached
          // (stream_nontemporal(); 4096^2: 23.9 -> 23.2 us in the CGLS loop); nt_store is grid-uniform
          if (nt_store & 1)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, out), rout, vst + (ofirst + dir * o) * rowbytes, 0, 2);
          else
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, out), rout, vst + (ofirst + dir * o) * rowbytes, 0, 0);
          if (SUMSQ) qacc = fmaf(out[0], out[0], fmaf(out[1], out[1], fmaf(out[2], out[2], fmaf(out[3], out[3], qacc))));
        }
      }
      if (SUMSQ && (u & 3) == 3) {                      // fp32 partial of <= 16 squares, then into the fp64 sum
        ss += (double)qacc;
        qacc = 0.f;
      }
    }
  };
  // first block (outputs start after KH-1 rows) and last block (prefetch stops) are guarded; the middle is not:
  // for t < total - U:  t + D < total  and  o = t - (KH-1) < band_rows  hold by construction, o >= 0 after block 0.
  block(0, std::true_type{});
  int t0 = U;
  for (; t0 + U < total; t0 += U) block(t0, std::false_type{});
  if (t0 < total) block(t0, std::true_type{});
  if (SUMSQ) ss += (double)qacc;
  if (SUMSQ) {
    ss = wave_sum(ss);
    if (lane == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = ss;
  }
}

// ------------------------------------------------------------------------------------------------ generic kernel
template <bool SUMSQ>
__global__ __launch_bounds__(NT) void k_blur_generic(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                     int64_t ldy, int nx, int ny, int kh, int kw,
                                                     const float* __restrict__ w, double* __restrict__ partials) {
  __shared__ double red[NT / 64];
  const int T = kh - 1 - kh / 2, L = kw - 1 - kw / 2;
  x += (int64_t)blockIdx.y * ldx;
  y += (int64_t)blockIdx.y * ldy;
  const int64_t npix = (int64_t)nx * ny;
  double ss = 0.0;
  for (int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x; idx < npix; idx += (int64_t)gridDim.x * NT) {
    const int i = (int)(idx / ny), j = (int)(idx - (int64_t)i * ny);
    float acc = 0.f;
    for (int a = 0; a < kh; ++a) {
      const int gi = reflect(i - T + a, nx);
      const float* row = x + (int64_t)gi * ny;
      for (int b = 0; b < kw; ++b) acc = fmaf(w[a * kw + b], row[reflect(j - L + b, ny)], acc);
    }
    y[idx] = acc;
    if (SUMSQ) ss += (double)acc * acc;
  }
  if (SUMSQ) {
    ss = block_sum<NT>(ss, red);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = ss;
  }
}

// ------------------------------------------------------------------------------------------------ dispatch
// strips x row bands: about 4 workgroups per CU, bands a whole number of 32-row steps
inline void strip_grid(int nx, int ny, int batch, int* strips_x, int* rows_per_band, int* nband) {
  const int sx = ceil_div(ny, TW), steps = ceil_div(nx, TH);
  int want = (4 * cu_count()) / (sx * (batch > 0 ? batch : 1));
  if (want < 1) want = 1;
  if (want > steps) want = steps;
  const int steps_per_band = ceil_div(steps, want);
  *strips_x = sx;
  *rows_per_band = steps_per_band * TH;
  *nband = ceil_div(nx, *rows_per_band);
}

template <int K>
int launch_strip(const BlurImpl* im, int tr, const float* x, int64_t ldx, float* y, int64_t ldy, int batch,
                 double* part, int strips_x, int rows_per_band, int nband, hipStream_t s) {
  dim3 grid(strips_x * nband, batch), block(NT);
  const float* w = im->separable ? im->sep_dev[tr] : im->w_dev[tr];
#define BL(SEP, SS) \
  hipLaunchKernelGGL((k_blur_strip<K, K, SEP, SS>), grid, block, 0, s, x, ldx, y, ldy, im->nx, im->ny, w, part, strips_x, rows_per_band)
  if (im->separable) { if (part) BL(true, true); else BL(true, false); }
  else               { if (part) BL(false, true); else BL(false, false); }
#undef BL
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

template <int K, int D>
int launch_slide(const BlurImpl* im, int tr, const float* x, int64_t ldx, float* y, int64_t ldy, int batch,
                 double* part, int spans_x, int nbands, int rows_per_band, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1,
                 const SlideFuse* fuse = nullptr, const SlideEpi* epi = nullptr) {
  dim3 grid(spans_x * nbands, batch), block(64);
  const float* w = im->sep_dev[tr];
  const int nts = stream_nontemporal((int64_t)im->nx * im->ny);
  if (fuse) {   // fused-operand form: always with the sum of squares (raw partials)
    hipExtLaunchKernelGGL((k_blur_slide<K, K, D, true, true>), grid, block, 0, s, ev0, ev1, 0, x, ldx, y, ldy, im->nx, im->ny, w, part, spans_x, nbands, rows_per_band, *fuse, nts, SlideEpi{});
    TRK_LAUNCH_CHECK();
    return TRK_OK;
  }
  const SlideFuse nofuse{nullptr, nullptr, 0.0, {nullptr, 0}, {nullptr, 0}};
  if (epi) {
    if (part)
      hipExtLaunchKernelGGL((k_blur_slide<K, K, D, true, false, true>), grid, block, 0, s, ev0, ev1, 0, x, ldx, y, ldy, im->nx, im->ny, w, part, spans_x, nbands, rows_per_band, nofuse, nts, *epi);
    else
      hipExtLaunchKernelGGL((k_blur_slide<K, K, D, false, false, true>), grid, block, 0, s, ev0, ev1, 0, x, ldx, y, ldy, im->nx, im->ny, w, part, spans_x, nbands, rows_per_band, nofuse, nts, *epi);
    TRK_LAUNCH_CHECK();
    return TRK_OK;
  }
  // hipExtLaunchKernelGGL attaches the (optional) events to the dispatch itself: their timestamps are the kernel's own
  // begin / end, the same quantity rocprofv3's kernel trace reports.
  if (part)
    hipExtLaunchKernelGGL((k_blur_slide<K, K, D, true, false>), grid, block, 0, s, ev0, ev1, 0, x, ldx, y, ldy, im->nx, im->ny, w, part, spans_x, nbands, rows_per_band, nofuse, nts, SlideEpi{});
  else
    hipExtLaunchKernelGGL((k_blur_slide<K, K, D, false, false>), grid, block, 0, s, ev0, ev1, 0, x, ldx, y, ldy, im->nx, im->ny, w, part, spans_x, nbands, rows_per_band, nofuse, nts, SlideEpi{});
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

// Bands for the sliding kernel.  A wave processes roundup(rows_per_band + KH-1, U) staged rows, so band heights of the
// form j*U - (KH-1) waste nothing.  Which j: band-height sweeps INSIDE the CGLS loop at 3072^2 ... 5120^2
// (tools/bench_rpb_sweep.sh) show the kernel time following  staged rows per wave x G(x),
// x = waves / (4 x CUs) the waves per SIMD:
//     G = 1                                        x <= 1   (fewer waves than SIMDs do not help: the rows per wave rule)
//     G = 0.6 + 0.45 x + 0.5 (ceil(x) - 1)         x >  1   (a second wave on a SIMD costs ~1.6x at once: both stream
//                                                            from HBM, neither hides the other's latency)
// i.e. the best grid is the shortest band that still gives every SIMD at most one wave, and the worst sit just above a
// multiple (4096^2, 9x9, forward matvec inside CGLS: 64-row bands, x = 1: 24.9 us; 55-row bands, x = 1.17: 34.0 us;
// 37-row bands, x = 1.73: 28.3 us).  Back-to-back launches on the same two buffers, which the 256 MB memory-side cache
// then holds, behave differently (37-row bands 22.5 us, 64-row bands 23.5 us): the rule follows the solver loop.
// Bands shorter than KH + 1 rows are never used (on a small image they only multiply the waves and the block
// partials every consumer sums).
inline void slide_grid(int nx, int ny, int batch, int kh, int U, int* spans_x, int* nbands, int* rows_per_band) {
  const int sx = ceil_div(ny, SPAN);
  const double slots = 4.0 * cu_count();
  const int nb = batch > 0 ? batch : 1;
  int rows = U;
  while (rows - (kh - 1) < kh + 1) rows += U;
  int rpb = rows - (kh - 1);
  double best = 1e300;
  for (;; rows += U) {
    const int r = rows - (kh - 1);
    const double x = (double)sx * ceil_div(nx, r) * nb / slots;
    const double g = x <= 1.0 ? 1.0 : 0.6 + 0.45 * x + 0.5 * (ceil(x) - 1.0);
    const bool last = x <= 1.0 || r >= nx;   // taller bands only add rows per wave from here on
    // more than 3840 waves (3.75 per SIMD on 256 CUs) are not considered: one block partial per wave, and the callers'
    // partial buffers (CGLS: 4096 doubles) are sized for that
    if ((x * slots <= 3840.0 || last) && rows * g < best) {
      best = rows * g;
      rpb = r;
    }
    if (last) break;
  }
  static const int rpb_env = getenv("TRK_BLUR_RPB") ? atoi(getenv("TRK_BLUR_RPB")) : 0;   // tuning knob (rows per band)
  if (rpb_env > 0) rpb = rpb_env;
  *spans_x = sx;
  *nbands = ceil_div(nx, rpb);
  *rows_per_band = rpb;
}

inline bool slide_shape_ok(const BlurImpl* im) {
  return im->separable && im->kh == im->kw && (im->kh & 1) && im->kh >= 3 && im->kh <= 9 && (im->ny & 3) == 0 &&
         im->ny >= 8 && (int64_t)im->nx * im->ny < ((int64_t)1 << 29);   // byte offsets are 32-bit signed ints in the kernel
}

// y = A (x1 + cb * x2), comb written out, sum(y^2) as raw partials (CGLS fast path; 9x9-class separable PSFs only)
int blur_apply_fused(trk_op* op, int tr, const float* x1, const float* x2, double sign, ScalarSrc num, ScalarSrc den,
                     float* comb, float* y, double* partials, int cap, int* n_partials, hipStream_t s) {
  auto* im = static_cast<BlurImpl*>(op->impl);
  if (!slide_shape_ok(im) || !aligned16(x1) || !aligned16(y) || (x2 && (!aligned16(x2) || !aligned16(comb))))
    return fail(TRK_EUNSUPPORTED, "blur2d fused apply: needs a separable odd PSF <= 9x9, ny %% 4 == 0, 16-byte aligned buffers");
  if (x2 && (comb == x1 || comb == x2))
    return fail(TRK_EINVAL, "blur2d fused apply: comb must not alias an input (halo rows are shared)");
  int spans_x, nbands, rpb;
  const int Usel = (im->kh == 9) ? 9 : (im->kh == 7) ? 7 : (im->kh == 5) ? 5 : 6;      // lcm(KH, D) of the instantiations below
  slide_grid(im->nx, im->ny, 1, im->kh, Usel, &spans_x, &nbands, &rpb);
  const int nblk = spans_x * nbands;
  if (nblk > cap) return fail(TRK_EINVAL, "blur2d fused apply: partial buffer holds %d doubles, %d needed", cap, nblk);
  *n_partials = nblk;
  const SlideFuse fz{x2, comb, sign, num, den};
  const SlideFuse* fzp = x2 ? &fz : nullptr;      // x2 == NULL: the plain one-operand kernel, partials left raw
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (trk_timer* t = op->timer)
    if ((op->timer_which == 2 || op->timer_which == tr) && t->used < t->cap) {
      ev0 = t->ev[2 * t->used];
      ev1 = t->ev[2 * t->used + 1];
      ++t->used;
    }
  switch (im->kh) {
    case 3: return launch_slide<3, 6>(im, tr, x1, 0, y, 0, 1, partials, spans_x, nbands, rpb, s, ev0, ev1, fzp);
    case 5: return launch_slide<5, 5>(im, tr, x1, 0, y, 0, 1, partials, spans_x, nbands, rpb, s, ev0, ev1, fzp);
    case 7: return launch_slide<7, 7>(im, tr, x1, 0, y, 0, 1, partials, spans_x, nbands, rpb, s, ev0, ev1, fzp);
    default:
      return launch_slide<9, 9>(im, tr, x1, 0, y, 0, 1, partials, spans_x, nbands, rpb, s, ev0, ev1, fzp);
  }
}

int blur_apply(trk_op* op, int tr, const float* x, int64_t ldx, float* y, int64_t ldy, int batch, double* sumsq,
               hipStream_t s) {
  auto* im = static_cast<BlurImpl*>(op->impl);
  double* part = nullptr;
  int nblk;
  const bool slide_ok = slide_shape_ok(im) && aligned16(x) && aligned16(y) &&
                        (batch == 1 || ((ldx & 3) == 0 && (ldy & 3) == 0));
  if (slide_ok) {
    int spans_x, nbands, rpb;
    const int Usel = (im->kh == 9) ? 9 : (im->kh == 7) ? 7 : (im->kh == 5) ? 5 : 6;   // lcm(KH, D) of the instantiations below
    slide_grid(im->nx, im->ny, batch, im->kh, Usel, &spans_x, &nbands, &rpb);
    nblk = spans_x * nbands;
    if (sumsq)
      if (int rc = scratch_doubles(s, (size_t)nblk * batch, &part)) return rc;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (trk_timer* t = op->timer)
      if ((op->timer_which == 2 || op->timer_which == tr) && t->used < t->cap) {
        ev0 = t->ev[2 * t->used];
        ev1 = t->ev[2 * t->used + 1];
        ++t->used;
      }
    int rc;
    switch (im->kh) {
      case 3: rc = launch_slide<3, 6>(im, tr, x, ldx, y, ldy, batch, part, spans_x, nbands, rpb, s, ev0, ev1); break;
      case 5: rc = launch_slide<5, 5>(im, tr, x, ldx, y, ldy, batch, part, spans_x, nbands, rpb, s, ev0, ev1); break;
      case 7: rc = launch_slide<7, 7>(im, tr, x, ldx, y, ldy, batch, part, spans_x, nbands, rpb, s, ev0, ev1); break;
      default:
        rc = launch_slide<9, 9>(im, tr, x, ldx, y, ldy, batch, part, spans_x, nbands, rpb, s, ev0, ev1);
        break;
    }
    if (rc) return rc;
    if (sumsq) return finalize_sums(part, nblk * batch, 1, 1, sumsq, s);
    return TRK_OK;
  }
  if (im->tiled) {
    int strips_x, rows_per_band, nband;
    strip_grid(im->nx, im->ny, batch, &strips_x, &rows_per_band, &nband);
    nblk = strips_x * nband;
    if (sumsq)
      if (int rc = scratch_doubles(s, (size_t)nblk * batch, &part)) return rc;
    TimerScope tm(op->timer, op->timer_which, tr, s);
    int rc;
    switch (im->kh) {
      case 3: rc = launch_strip<3>(im, tr, x, ldx, y, ldy, batch, part, strips_x, rows_per_band, nband, s); break;
      case 5: rc = launch_strip<5>(im, tr, x, ldx, y, ldy, batch, part, strips_x, rows_per_band, nband, s); break;
      case 7: rc = launch_strip<7>(im, tr, x, ldx, y, ldy, batch, part, strips_x, rows_per_band, nband, s); break;
      case 9: rc = launch_strip<9>(im, tr, x, ldx, y, ldy, batch, part, strips_x, rows_per_band, nband, s); break;
      case 11: rc = launch_strip<11>(im, tr, x, ldx, y, ldy, batch, part, strips_x, rows_per_band, nband, s); break;
      case 13: rc = launch_strip<13>(im, tr, x, ldx, y, ldy, batch, part, strips_x, rows_per_band, nband, s); break;
      case 15: rc = launch_strip<15>(im, tr, x, ldx, y, ldy, batch, part, strips_x, rows_per_band, nband, s); break;
      default: return fail(TRK_EUNSUPPORTED, "blur2d: no strip kernel for %dx%d", im->kh, im->kw);
    }
    tm.stop();
    if (rc) return rc;
  } else {
    const int64_t npix = (int64_t)im->nx * im->ny;
    int64_t want = (npix + NT - 1) / NT;
    if (want > 4096) want = 4096;
    nblk = (int)(want < 1 ? 1 : want);
    if (sumsq)
      if (int rc = scratch_doubles(s, (size_t)nblk * batch, &part)) return rc;
    dim3 grid(nblk, batch);
    TimerScope tm(op->timer, op->timer_which, tr, s);
    if (part)
      hipLaunchKernelGGL((k_blur_generic<true>), grid, dim3(NT), 0, s, x, ldx, y, ldy, im->nx, im->ny, im->kh, im->kw, im->w_dev[tr], part);
    else
      hipLaunchKernelGGL((k_blur_generic<false>), grid, dim3(NT), 0, s, x, ldx, y, ldy, im->nx, im->ny, im->kh, im->kw, im->w_dev[tr], part);
    tm.stop();
    TRK_LAUNCH_CHECK();
  }
  if (sumsq) return finalize_sums(part, nblk * batch, 1, 1, sumsq, s);
  return TRK_OK;
}

// out = a * Op(x) + b * z (+ ||out||^2) in the sliding kernel's store (trk_op_apply_axpby on a blur handle; caps stay 0)
int blur_apply_axpby_plain(trk_op* op, int tr, const float* x, Coef a, Coef b, const float* z, float* out, double* sumsq,
                           hipStream_t s) {
  auto* im = static_cast<BlurImpl*>(op->impl);
  if (!slide_shape_ok(im) || !aligned16(x) || !aligned16(out) || (z && !aligned16(z))) return TRK_EUNSUPPORTED;
  int spans_x, nbands, rpb;
  const int Usel = (im->kh == 9) ? 9 : (im->kh == 7) ? 7 : (im->kh == 5) ? 5 : 6;
  slide_grid(im->nx, im->ny, 1, im->kh, Usel, &spans_x, &nbands, &rpb);
  const int nblk = spans_x * nbands;
  double* part = nullptr;
  if (sumsq)
    if (int rc = scratch_doubles(s, (size_t)nblk, &part)) return rc;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (trk_timer* t = op->timer)
    if ((op->timer_which == 2 || op->timer_which == tr) && t->used < t->cap) {
      ev0 = t->ev[2 * t->used];
      ev1 = t->ev[2 * t->used + 1];
      ++t->used;
    }
  const SlideEpi ep{z, a, b};
  int rc;
  switch (im->kh) {
    case 3: rc = launch_slide<3, 6>(im, tr, x, 0, out, 0, 1, part, spans_x, nbands, rpb, s, ev0, ev1, nullptr, &ep); break;
    case 5: rc = launch_slide<5, 5>(im, tr, x, 0, out, 0, 1, part, spans_x, nbands, rpb, s, ev0, ev1, nullptr, &ep); break;
    case 7: rc = launch_slide<7, 7>(im, tr, x, 0, out, 0, 1, part, spans_x, nbands, rpb, s, ev0, ev1, nullptr, &ep); break;
    default: rc = launch_slide<9, 9>(im, tr, x, 0, out, 0, 1, part, spans_x, nbands, rpb, s, ev0, ev1, nullptr, &ep); break;
  }
  if (rc) return rc;
  if (sumsq) return finalize_sums(part, nblk, 1, 1, sumsq, s);
  return TRK_OK;
}

void blur_destroy(trk_op* op) {
  auto* im = static_cast<BlurImpl*>(op->impl);
  for (int t = 0; t < 2; ++t) {
    if (im->w_dev[t]) (void)hipFree(im->w_dev[t]);
    if (im->sep_dev[t]) (void)hipFree(im->sep_dev[t]);
  }
  delete im;
}

}  // namespace

// what the tiled small-image CGLS (cgls_tiled.hip) needs of a blur handle: sizes and the separable weights on the device
namespace trk {
bool blur_separable_params(trk_op* op, int* nx, int* ny, int* kh, int* kw, const float** sep_fwd, const float** sep_adj) {
  if (!op || op->kind != 1) return false;
  auto* im = static_cast<BlurImpl*>(op->impl);
  if (!im->separable || !im->sep_dev[0] || !im->sep_dev[1]) return false;
  *nx = im->nx;
  *ny = im->ny;
  *kh = im->kh;
  *kw = im->kw;
  *sep_fwd = im->sep_dev[0];
  *sep_adj = im->sep_dev[1];
  return true;
}
}  // namespace trk

extern "C" int trk_blur2d_create(const double* psf, int kh, int kw, int nx, int ny, trk_op** out) {
  TRK_REQUIRE(psf && out, "trk_blur2d_create: NULL argument");
  TRK_REQUIRE(kh >= 1 && kw >= 1 && nx >= 1 && ny >= 1, "trk_blur2d_create: sizes must be >= 1");
  TRK_REQUIRE((int64_t)kh * kw <= (1 << 24), "trk_blur2d_create: PSF too large");
  auto* im = new BlurImpl{nx, ny, kh, kw, false, false, {nullptr, nullptr}, {nullptr, nullptr}};

  // correlation weights: forward c[a'][b'] = psf[kh-1-a'][kw-1-b'];  "transpose" convolves with flip(psf): c = psf
  std::vector<float> wf((size_t)kh * kw), wt((size_t)kh * kw);
  for (int a = 0; a < kh; ++a)
    for (int b = 0; b < kw; ++b) {
      wf[(size_t)a * kw + b] = (float)psf[(size_t)(kh - 1 - a) * kw + (kw - 1 - b)];
      wt[(size_t)a * kw + b] = (float)psf[(size_t)a * kw + b];
    }

  // rank-1 test in double: psf[a][b] == col[a] * row[b]  (pivot = largest |entry|)
  int pa = 0, pb = 0;
  double pmax = 0.0;
  for (int a = 0; a < kh; ++a)
    for (int b = 0; b < kw; ++b)
      if (std::fabs(psf[(size_t)a * kw + b]) > pmax) {
        pmax = std::fabs(psf[(size_t)a * kw + b]);
        pa = a;
        pb = b;
      }
  std::vector<double> col(kh), row(kw);
  bool sep = pmax > 0.0;
  if (sep) {
    const double piv = psf[(size_t)pa * kw + pb];
    for (int a = 0; a < kh; ++a) col[a] = psf[(size_t)a * kw + pb];
    for (int b = 0; b < kw; ++b) row[b] = psf[(size_t)pa * kw + b] / piv;
    for (int a = 0; a < kh && sep; ++a)
      for (int b = 0; b < kw; ++b)
        if (std::fabs(psf[(size_t)a * kw + b] - col[a] * row[b]) > 1e-12 * pmax) {
          sep = false;
          break;
        }
  }
  im->separable = sep;
  im->tiled = (kh == kw) && (kh & 1) && kh >= 3 && kh <= 15;

  auto upload = [](const std::vector<float>& h, float** d) -> int {
    TRK_HIP(hipMalloc(d, h.size() * sizeof(float)));
    TRK_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return TRK_OK;
  };
  int rc = upload(wf, &im->w_dev[0]);
  if (!rc) rc = upload(wt, &im->w_dev[1]);
  if (!rc && sep) {
    std::vector<float> sf(kw + kh), st(kw + kh);
    for (int b = 0; b < kw; ++b) {
      sf[b] = (float)row[kw - 1 - b];
      st[b] = (float)row[b];
    }
    for (int a = 0; a < kh; ++a) {
      sf[kw + a] = (float)col[kh - 1 - a];
      st[kw + a] = (float)col[a];
    }
    rc = upload(sf, &im->sep_dev[0]);
    if (!rc) rc = upload(st, &im->sep_dev[1]);
  }
  if (rc) {
    trk_op tmp{1, 0, 0, im, nullptr, nullptr, nullptr, 0};
    blur_destroy(&tmp);
    return rc;
  }
  const int64_t n = (int64_t)nx * ny;
  auto* op = new trk_op{1, n, n, im, blur_apply, blur_destroy, nullptr, 0};
  if (slide_shape_ok(im)) op->apply_fused = blur_apply_fused;
  if (slide_shape_ok(im)) op->apply_axpby_plain = blur_apply_axpby_plain;
  *out = op;
  return TRK_OK;
}
