// radon2d.hip — parallel-beam Radon transform (Joseph / linear-interpolation projector) and its matched adjoint.
//
// Replaces astra.OpTomo over create_proj_geom('parallel', 1, N, theta) + create_projector('linear', ...) and the
// 1/N scaling of trips/utilities/io.py:392-399.  astra-toolbox is not installable in the build image; the convention below
// (oracle/cpu_ref.py, Radon2D) is pinned to the ASTRA outputs the reference holds as images — through the fan-beam operator whose
// far-source limit this one is (tests/test_oracle_golden.py, tests/test_gpu_operators.py) — and by the adjoint identity, analytic
// line integrals and the axis-aligned views; the interpolation weights are Joseph's published kernel.
//
// Geometry: pixel (i,j) centre (x,y) = (j - h, h - i), h = (N-1)/2; detector bin d at s = d - (n_det-1)/2 along
// (cos t, sin t); rays run along (sin t, -cos t).  Per angle one of two marching modes:
//   mode 0 (|cos| >= |sin|): march image rows  tt = i, coordinate q = column:  q = s/cos + (h - h tan) + i tan
//   mode 1 (otherwise)     : march image cols  tt = j, coordinate q = row   :  q = -s/sin + (h - h cot) + j cot
// i.e.  q(d, tt) = A(d) + B(tt),  A(d) = s_d inv + k0,  B(tt) = tt dq;  taps at floor(q), floor(q)+1, weights (1-f), f,
// times wgt = scale/|cos| (or /|sin|).
//
// FIXED-POINT RAY COORDINATE (round 2).  Evaluated in fp32, q — a number up to N — carries an error of ~6e-8 N, i.e. an
// interpolation weight off by 2e-4 at N = 4096 (measured 1.9e-4 relative on white noise).  Here the two terms come from
// TABLES made once per operator in float64 and rounded to 24 fractional bits, A32[a][d] = round(A(d) 2^24) mod 2^32 and
// B32[a][tt] = round(B(tt) 2^24) mod 2^32, and the kernels add them as INTEGERS:
//     Q = A32 + B32 (mod 2^32):   column mod 256 = Q >> 24,   f = (Q & 0xFFFFFF) 2^-24   (exact in fp32, as is 1 - f).
// Every weight is within 2^-24 of its float64 value whatever N; forward and adjoint read the same tables, so the nearest
// ray's weight in the adjoint is bit-identical to the forward's; the absolute column comes from where the tap is looked
// for (the LDS window start, the gathering pixel) — 8 integer bits are plenty for that.  The march costs what it did
// (7 vector instructions per step); measured against the float64 oracle: DESIGN.md §4.4.
//
// Forward: mode-1 angles read a transposed copy of the image so that both modes read ROWS: the 64 adjacent detectors of
// a wave touch 64..90 contiguous floats per marching step.  The grid runs over bands of 128 marching rows (see
// k_radon_fwd); a workgroup is 64 detectors x 4 neighbouring angles.
// Adjoint: gather form, no atomics (k_radon_adj_tile): per pixel and angle the nearest ray d0 and its two neighbours —
// every ray within one pixel is among them, because |dq/dd| = 1/|cos| >= 1 — from 16-byte records staged in LDS.
//
// Roofline note: algorithmic bytes are only 4(N^2 + n_ang n_det) against 2 N n_det n_ang taps, so this
// operator is bound by LDS reads / vector issue, not HBM; bench.py reports taps/s next to GB/s.
#include "trk_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <vector>

using namespace trk;

namespace {

constexpr int QF = 24;                          // fractional bits of the fixed-point ray coordinate
constexpr float QONE = 16777216.0f;             // 2^24: the adjoint's weights are in units of 2^-24
constexpr float QTWO32 = 4294967296.0f;         // 2^32: the forward's weights are in units of 2^-32 ((float)(Q << 8))
constexpr int A32_PAD = 2;                      // A32 rows hold d = -2 .. nd+1

struct AngleParam {
  float inv, dq, k0, wgt;   // fp32 copies: only for ESTIMATES (window placement, candidate location); wgt includes the forward's 2^-32
  int mode;
  float rinv;               // ~1/inv
};

struct alignas(8) AdjAngle { // the adjoint's per-angle constants, sorted by marching mode per frame
  float c1m, c1p;           // 1 - |inv| (<= 0) ON THE 2^-24 GRID, for the neighbour on the smaller-q / larger-q side: a neighbouring ray at
                            // distance |inv| weighs clamp(c1m + t0) / clamp(c1p - t0).  Read as ONE scalar pair: the packed FMA's
                            // addend (adj_gather).  Which grid neighbours of the real 1 - |inv| the two hold is chosen when the handle is
                            // made (radon_create_impl: error diffusion over the angles)
  float rinv, dq, k0;
  int orig;                 // index of the angle within its frame
  int flip;                 // inv < 0: the ray on the larger-q side is d0 - 1 (the records store neighbours by side)
  int pad_;
};

struct QuadParam;
struct QuadPlan;
struct alignas(8) AdjQuad {  // base geometry of a quad for the adjoint: c1m / c1p = 1 - inv (<= 0) on the 2^-24 grid (as AdjAngle's),
  float c1m, c1p;           // rinv = cos(beta), dq = tan(beta), k0 = h (1 - dq)
  float rinv, dq, k0;
  int pad_;
};

struct RadonImpl {
  int N, nd, na;   // na = angles PER FRAME
  int nt;          // time frames sharing one launch (block-diagonal dynamic operator, io.py:391-420); 1 = static
  AngleParam* ang_dev;   // nt*na entries, frame-major
  float* xT;  // nt*N*N transposed images (forward, mode-1 angles); owned by the handle (non-reentrant across streams)
  int n_mode1;
  float* part;  // [n_bands][nt*na][nd] forward band partial sums (n_bands > 1 only); owned by the handle
  float* fidx;  // fidx[i] = (float) i, i < N + 16
  unsigned* A32;  // [nt*na][nd + 4]
  unsigned* B32;  // [nt*na][npad]
  uint2* CB;      // [nt*na][npad]: {C[a][tt] as float bits, B32[a][tt]}, C = the adjoint's locator offset
  int npad;
  // adjoint: angles sorted by mode per frame; per apply a record array {w S[d-1], w S[d], w S[d+1], A32[d]}
  AdjAngle* adj_ang;
  float* adj_wgt;
  int* adj_n0;
  uint4* rec;
  int4* adj_pos;     // [nt*na]: angle row (caller's order) -> {its sorted row (the inverse of adj_ang[].orig), that row's
                     // adjoint weight (bits), its flip flag, 0}: one load where the record writer chased three
  int n_bands, band;
  int band_res;     // forward by k_radon_fwd_band: 64-row bands resident in LDS (small images)
  // adjoint with the angles of a tile split over `nsplit` workgroups (small images): partial tiles and one counter per tile
  float* adj_part;
  unsigned* adj_cnt;
  int64_t adj_part_cap, adj_cnt_cap;
  // quads: groups of up to four symmetric angles served by one wave of k_radon_fwd_quad (nq per frame, padded with empty ones)
  QuadParam* quad_dev;
  unsigned* A32q;   // [nt*nq][nd + 4]  base tables
  unsigned* B32q;   // [nt*nq][npad]
  int nq;
  // round 6: the quad kernel's per-workgroup bookkeeping made once per operator (k_radon_quad_plan) and the compact list of the
  // workgroups the lean kernel (k_radon_fwd_quadf) does not serve
  // round 6, adjoint by mirrored tile pairs (k_radon_adj_quad): per quad {c1, rinv, dq, k0} of the BASE geometry, {C, B32} of the base per
  // marching index, the members' weights; recq = per-apply records indexed by (quad, slot, BASE detector)
  struct AdjQuad* adjq;
  uint2* CBq;        // [nt*nq][npad]
  float* wq;         // [nt*nq][4]
  uint4* recq;       // [nt*nq][4][nd + 4]
  int adjq_ok;
  struct QuadPlan* qplan;
  int* qslow;       // [0] = count, then the workgroup ids (band * grid_x + block) k_radon_fwd_quad still runs
  int qslow_n;      // host copy of the count
  int qplan_gx, qplan_nb;   // the grid the plan was made for
  // what the side buffers currently hold, when a fused apply left them behind for the next apply of the other direction
  // (trk_op_apply_axpby hints): rec = the records of the sinogram at rec_src, xT = the transpose of the image at xT_src
  const float* rec_src;
  const float* xT_src;
  // a fused norm left as block partials (TRK_HINT_SUMSQ_DEFERRED): pend_n partials in pend_buf[pend_which ^ 1] belong to
  // *pend_target; the next chained apply's epilogue kernel finishes it, anything else calls radon_flush first
  double* pend_buf[2];
  int64_t pend_cap;
  int pend_which;
  double* pend_target;
  const double* pend_part;
  int pend_n;
  // the float64 instrument (ref64.hip): the angles in float64, and the arithmetic this handle's applies run in —
  // 0 the product's kernels; 1 float64 geometry and sums (fp32 vectors); 2 the fixed-point tables' weights, float64 sums
  RadonRefAngle* ref_ang;
  int ref_mode;
  int ref_chunk_fwd, ref_chunk_adj;   // emulated fp32 partial sums of the instrument (0: float64 sums)
  float* ref_tmp;     // max(rows, cols) floats: Op(x) before the half step's combination (ref_mode != 0 only)
};

// Optional epilogue of the kernel that writes an apply's output (trk_op_apply_axpby): out = a * Op(x) + b * z.
struct Epi {
  int on;            // 0: out = Op(x)
  Coef a, b;
  const float* z;    // NULL: out = a * Op(x)
  // a norm the previous fused apply of this operator left as block partials (TRK_HINT_SUMSQ_DEFERRED): coefficients that
  // point at pend_target take the sum of the partials instead, and workgroup 0 stores the finished value there
  double* pend_target;
  const double* pend_part;
  int pend_n;
  // forward only (trk_gk_step_proj): block partials of <out, dotv> next to the fused norm's
  const float* dotv;
  double* dot_part;
  // adjoint only (trk_gk_step_lsqr): the damped-LSQR update whose vk is this epilogue's z
  LsqrReq lq;
  // adjoint only (trk_gk_step_post): a mailbox post carried by workgroup 0
  PostReq pq;
};

// the sum of the pending partials — the same bits in every workgroup (one wave, fixed order) — in all threads
__device__ __forceinline__ double pend_total(const Epi& e, double* lds1) {
  if (threadIdx.x < 64) {
    // eight loads in flight per trip, added in the order a one-by-one loop would (the convention of trk_internal.h: every
    // consumer of the same partials gets the same bits); one by one, the 1024 partials of a 512^2 adjoint were 16 dependent
    // L2 round trips at the head of every workgroup of the kernel that follows
    double v = 0.0;
    for (int i = threadIdx.x; i < e.pend_n; i += 512) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = (i + 64 * u < e.pend_n) ? e.pend_part[i + 64 * u] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
    v = wave_sum_all(v);
    if (threadIdx.x == 0) *lds1 = v;
  }
  __syncthreads();
  return *lds1;
}
// (num_v / den_v: the scalars *k.num / *k.den as fetched BEFORE the pending sum was formed — behind its barrier they were one more
//  exposed round trip at the end of every workgroup of the band reduction and of the adjoint's finishers; a value that is the
//  pending target itself is stale there and not used)
__device__ __forceinline__ double coef_eval_pend(const Coef& k, const double* target, double total, double num_v, double den_v) {
  double v = k.c;
  if (k.num) {
    const double t = (k.num == target) ? total : num_v;
    v *= (k.flags & TRK_SQRT_NUM) ? sqrt(t) : t;
  }
  if (k.den) {
    const double t = (k.den == target) ? total : den_v;
    v /= (k.flags & TRK_SQRT_DEN) ? sqrt(t) : t;
  }
  return v;
}
// both coefficients of the epilogue (uniform over the grid; ends with every thread past a barrier when a norm is pending)
__device__ __forceinline__ void epi_coefs(const Epi& e, bool first_block, double* lds1, float& ca, float& cb, double* total_out = nullptr,
                                          double* cad = nullptr, double* cbd = nullptr) {
  ca = 1.f;
  cb = 0.f;
  double da = 1.0, db = 0.0;
  if (e.on) {
    if (e.pend_target) {
      const double an = e.a.num ? *e.a.num : 0.0, ad = e.a.den ? *e.a.den : 0.0;
      const double bn = (e.z && e.b.num) ? *e.b.num : 0.0, bd = (e.z && e.b.den) ? *e.b.den : 0.0;
      const double total = pend_total(e, lds1);
      if (total_out) *total_out = total;
      da = coef_eval_pend(e.a, e.pend_target, total, an, ad);
      if (e.z) db = coef_eval_pend(e.b, e.pend_target, total, bn, bd);
      if (first_block && threadIdx.x == 0) *e.pend_target = total;
    } else {
      da = coef_eval(e.a);
      if (e.z) db = coef_eval(e.b);
    }
    ca = (float)da;
    cb = (float)db;
  }
  if (cad) *cad = da;
  if (cbd) *cbd = db;
}
// the epilogue's arithmetic: e.on == 2 (default) float64 coefficients and products, one rounding of the result; e.on == 1
// (TRK_RADON_EPI_F32=1) that of trk_axpby — fp32 coefficients, one FMA: the fused half step equals apply + trk_axpby to the bit
__device__ __forceinline__ float epi_combine(int on, float ca, float cb, double cad, double cbd, float o, float z, bool has_z) {
  if (on == 2) return (float)(has_z ? fma(cad, (double)o, cbd * (double)z) : cad * (double)o);
  return has_z ? fmaf(ca, o, cb * z) : ca * o;
}

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
// Both taps of a step come from ONE 8-byte load at (row, floor(q)); out-of-range taps get weight 0.
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
typedef float f4r __attribute__((ext_vector_type(4)));

// Absolute left-tap column of a step: the fixed-point sum Q knows it mod 256, the fp32 estimate of q (off by far less than a
// column) says which multiple of 256.
__device__ __forceinline__ int radon_abs_col(unsigned Q, float qest) {
  const int ce = (int)floorf(qest);
  const int cm = (int)(Q >> QF);
  return ce + (((cm - ce + 128) & 255) - 128);
}

// One marching step of one ray with full edge handling (direct-gather paths): offset of the 8-byte load and the two tap
// weights in units of 2^-32 (taps outside the image, steps outside [.., te) and rays outside the detector weigh 0).
__device__ __forceinline__ int radon_edge_tap(int tt, int te, bool live, int N, float dq, float base, unsigned A,
                                              const unsigned* __restrict__ Brow, f2v& w) {
  const bool valid = live && tt < te;
  const int tr = tt < te ? tt : te - 1;
  const unsigned Q = A + Brow[tr];
  const int c = radon_abs_col(Q, fmaf((float)tr, dq, base));
  const float f1 = (float)(Q << 8), f0 = QTWO32 - f1;              // units of 2^-32, like the staged march
  // c == -1: only the right tap (column 0) is inside; start the 8-byte load at column 0 instead (an access that
  // STARTS below the buffer is dropped whole by the range check — measured on gfx950 — while one that runs off
  // the end returns its in-range dword)
  const bool neg1 = (c == -1);
  const int cl = neg1 ? 0 : c;
  const float w0 = neg1 ? f1 : (((unsigned)c < (unsigned)N) ? f0 : 0.f);
  const float w1 = neg1 ? 0.f : (((unsigned)(c + 1) < (unsigned)N) ? f1 : 0.f);
  w[0] = valid ? w0 : 0.f;
  w[1] = valid ? w1 : 0.f;
  const bool anyin = (unsigned)cl < (unsigned)N;
  return anyin ? (tr * N + cl) * 4 : 0x7FFFFFF0;                   // far outside: returns 0, fetches nothing
}

// Forward kernel (direct gathers; any N).  grid = (ceil(nd/64) * n_angle_groups, n_bands); block = 256 = 4 waves = 4
// CONSECUTIVE ANGLES of one frame x 64 detectors, marching the RADON_BAND image rows (mode 1: columns, through the transposed
// copy) of band blockIdx.y.  Why this shape (measured at 4096^2 x 180, MI355X): a wave marching the whole image touches 3-4
// new cache lines per step and never returns to them, and every angle sweeps the whole 67 MB image, so the first version
// (one wave = a quarter of the image) moved ~12 GB through the fabric per apply and was bound by L2 misses (2.15 ms;
// halving its VALU work changed nothing).  With row bands the grid runs band by band (blockIdx.x is the fast index),
// the 2 MB band stays in every XCD's 4 MB L2 while all angles and detectors pass over it, and the four waves of a
// workgroup - neighbouring angles, same detectors, same rows at the same time - share most of their L1 lines.
// Band partial sums go to a scratch array [band][angle][detector] that k_radon_bands_sum adds up in a fixed order.
#define RADON_CHUNK 32
#define RADON_BAND 128

template <bool FINAL>
__global__ __launch_bounds__(256) void k_radon_fwd(const float* __restrict__ img, const float* __restrict__ imgT,
                                                   float* __restrict__ out, int N, int nd,
                                                   const AngleParam* __restrict__ ang, int na_per_frame, int ngrp_per_frame,
                                                   int ndblk, int64_t band_stride, int bh,
                                                   const unsigned* __restrict__ A32, const unsigned* __restrict__ B32, int npad) {
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int grp = blockIdx.x / ndblk, dblk = blockIdx.x - grp * ndblk;
  const int frame = grp / ngrp_per_frame;
  const int af = (grp - frame * ngrp_per_frame) * 4 + wv;        // angle within the frame
  if (af >= na_per_frame) return;
  const int a = frame * na_per_frame + af;                         // global angle index (frame-major)
  const AngleParam p = ang[a];
  const float* __restrict__ I = (p.mode ? imgT : img) + (int64_t)frame * N * N;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)I, 0, (unsigned)N * (unsigned)N * 4u, 0x00020000);
  const int d = dblk * 64 + lane;
  const bool live = d < nd;
  const unsigned A = A32[(int64_t)a * (nd + 2 * A32_PAD) + (live ? d : nd - 1) + A32_PAD];
  const unsigned* __restrict__ Brow = B32 + (int64_t)a * npad;
  const float s = (float)d - 0.5f * (float)(nd - 1);
  const float base = fmaf(s, p.inv, p.k0);
  const int t0 = blockIdx.y * bh, t1 = (t0 + bh < N) ? t0 + bh : N;
  const float qmax = (float)(N - 2);
  double total = 0.0;
  for (int tb = t0; tb < t1; tb += RADON_CHUNK) {
    const int te = (tb + RADON_CHUNK < t1) ? tb + RADON_CHUNK : t1;
    const float qa = fmaf((float)tb, p.dq, base), qb = fmaf((float)(te - 1), p.dq, base);
    // a chunk whose taps are inside the image for EVERY ray of the wave (q is monotone in tt; one column of margin for
    // the estimate) runs without any edge logic
    const bool inside = !live || (fminf(qa, qb) >= 1.f && fmaxf(qa, qb) < qmax);
    if (te - tb == RADON_CHUNK && __builtin_amdgcn_ballot_w64(inside) == ~0ull) {
      f2v acc2 = {0.f, 0.f};
      // two batches of 8 steps in flight: the loads of batch k+1 are issued before batch k is accumulated
      f2v w[2][8], v[2][8];
      auto issue = [&](int k) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int tt = tb + 8 * k + u;
          const unsigned Q = A + Brow[tt];
          const int c = radon_abs_col(Q, fmaf((float)tt, p.dq, base));
          w[k & 1][u][1] = (float)(Q << 8);
          w[k & 1][u][0] = QTWO32 - w[k & 1][u][1];
          v[k & 1][u] = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(rsrc, c << 2, (unsigned)tt * (unsigned)N * 4u, 0));
        }
      };
      issue(0);
#pragma unroll
      for (int k = 0; k < RADON_CHUNK / 8; ++k) {
        if (k + 1 < RADON_CHUNK / 8) issue(k + 1);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc2 = __builtin_elementwise_fma(w[k & 1][u], v[k & 1][u], acc2);
      }
      total += (double)(acc2[0] + acc2[1]);
    } else {
      // edge chunk (or the short last one): same batching, weights carry the edge logic
      f2v acc2 = {0.f, 0.f};
#pragma unroll 1
      for (int k = 0; k < RADON_CHUNK / 8 && tb + 8 * k < te; ++k) {
        f2v w[8], v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int off = radon_edge_tap(tb + 8 * k + u, te, live, N, p.dq, base, A, Brow, w[u]);
          v[u] = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc2 = __builtin_elementwise_fma(w[u], v[u], acc2);
      }
      total += (double)(acc2[0] + acc2[1]);
    }
  }
  if (live) {
    if (FINAL) out[(int64_t)a * nd + d] = p.wgt * (float)total;
    else out[(int64_t)blockIdx.y * band_stride + (int64_t)a * nd + d] = (float)total;
  }
}

// LDS-staged forward kernel (N % 4 == 0, 16-byte aligned images).  Same grid, same band partial sums, same arithmetic per
// tap as k_radon_fwd; what changes is how the taps reach the lanes.  Measured on k_radon_fwd: once the bands made the
// image L2-resident, the 8-byte per-lane gathers ran at ~3.6 lanes/clk/CU — the texture addresser's rate for scattered
// 64-bit loads — whatever the band height.  Here each wave stages the window of the image its 64 rays cross during a
// chunk of LDS_R = 16 rows — at most 64 sqrt(2) + 15 + 2 columns, rounded to 16-byte groups: LDS_W = 112 floats — with
// 7 coalesced 16-byte loads per lane (28 consecutive lanes read 448 contiguous bytes), and takes the two taps of a step
// with one ds_read2_b32 (the 32 lanes of a half-wave hit distinct banks: the rays' columns are strictly increasing and
// span < 64 floats).  Columns outside the image are staged as zeros (out-of-range buffer offsets), so the marching loop
// carries no edge logic at all.  The tile is private to its wave: no workgroup barrier, LDS operations of one wave
// execute in order.  DMA = true (default): the 7 loads go straight into LDS (buffer_load_dwordx4 ... lds — lane l of load i
// lands at float4 64 i + l of the tile, which is exactly the staging order; out-of-range lanes store zeros), which takes
// the texture-data -> register -> LDS detour out of the path.
// The march per step (7 vector instructions): Q = A' + B32[tt] (A' = the ray's table entry minus the window start, per chunk;
// B32[tt] through the scalar cache), column within the window = Q >> 24, weights (float)(Q & 0xFFFFFF) and 2^24 minus that,
// LDS address, one ds_read2_b32, one packed FMA.
#define LDS_R 16
#define LDS_W 112
#define LDS_WT 116    // row stride of a tile staged through the transposing path (direct1)

template <bool FINAL, bool DMA = false>
__global__ __launch_bounds__(256) void k_radon_fwd_lds(const float* __restrict__ img, const float* __restrict__ imgT,
                                                       float* __restrict__ out, int N, int nd,
                                                       const AngleParam* __restrict__ ang, int na_per_frame,
                                                       int ngrp_per_frame, int ndblk, int64_t band_stride, int bh,
                                                       const unsigned* __restrict__ A32, const unsigned* __restrict__ B32, int npad,
                                                       int direct1) {
  __shared__ __attribute__((aligned(16))) float tile[4][LDS_R * LDS_WT];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int grp = blockIdx.x / ndblk, dblk = blockIdx.x - grp * ndblk;
  const int frame = grp / ngrp_per_frame;
  const int af = (grp - frame * ngrp_per_frame) * 4 + wv;        // angle within the frame
  if (af >= na_per_frame) return;
  const int a = frame * na_per_frame + af;                         // global angle index (frame-major)
  const AngleParam p = ang[a];
  // direct1: angles marched along COLUMNS read the image itself and transpose while staging (no transposed copy, no launch
  // for it): the window is then 112 image rows x 16 columns, a lane's float4 is four marching steps of one row
  const bool tdir = direct1 && p.mode;                             // wave-uniform
  const float* __restrict__ I = ((p.mode && !tdir) ? imgT : img) + (int64_t)frame * N * N;
  const int rs4 = __builtin_amdgcn_readfirstlane((tdir ? LDS_WT : LDS_W) * 4);   // byte stride of a tile row
  const unsigned img_bytes = (unsigned)N * (unsigned)N * 4u;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)I, 0, img_bytes, 0x00020000);
  float* __restrict__ T = tile[wv];
  const int d = dblk * 64 + lane;
  const float sdh = 0.5f * (float)(nd - 1);
  const float base = fmaf((float)d - sdh, p.inv, p.k0);
  const int nlive = (nd - dblk * 64 < 64) ? nd - dblk * 64 : 64;   // live lanes are 0 .. nlive-1 (wave-uniform)
  const bool live = lane < nlive;
  const unsigned A = A32[(int64_t)a * (nd + 2 * A32_PAD) + (live ? d : nd - 1) + A32_PAD];
  const unsigned* __restrict__ Ball = B32 + (int64_t)a * npad;
  float two32 = 4294967296.0f;                     // kept in an SGPR (opaque to the optimiser): no 32-bit literal per step
  asm("" : "+s"(two32));
  // base is monotone in the lane: the window's column range comes from the first and the last live ray
  const float b0 = fmaf((float)(dblk * 64) - sdh, p.inv, p.k0), b1 = fmaf((float)(dblk * 64 + nlive - 1) - sdh, p.inv, p.k0);
  const float blo = fminf(b0, b1), bhi = fmaxf(b0, b1);
  // staging slots of this lane: float4 number lane + 64 i of the 16 x 28 tile (row, 4-column group) — chunk-invariant
  int sc4[7], srowN4[7], slds[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int idx = lane + 64 * i;
    const int row = idx / (LDS_W / 4);
    sc4[i] = (idx - row * (LDS_W / 4)) * 4;
    srowN4[i] = row * N * 4;
    slds[i] = row * LDS_W + sc4[i];
  }
  const int t0 = blockIdx.y * bh, t1 = (t0 + bh < N) ? t0 + bh : N;
  double total = 0.0;
  for (int tb = t0; tb < t1; tb += LDS_R) {
    const int te = (tb + LDS_R < t1) ? tb + LDS_R : t1;
    // column range of all taps of the chunk (q is monotone in tt as well): wave-uniform
    const float ta = (float)tb * p.dq, tz = (float)(te - 1) * p.dq;
    const float qlo = blo + fminf(ta, tz), qhi = bhi + fmaxf(ta, tz);
    // a chunk in which every tap of the wave lies outside the image (oblique views: the corners the detector overhangs) adds exact
    // zeros: skipped (two columns of margin for the fp32 estimate)
    if (__builtin_amdgcn_readfirstlane((qhi < -2.f || qlo > (float)N + 1.f) ? 1 : 0)) continue;
    // (readfirstlane: the values are wave-uniform but were computed in vector registers)
    const int cs = __builtin_amdgcn_readfirstlane(((int)floorf(qlo) - 1) & ~3);   // one column of slack for the fp32 estimate
    const bool fits = (__builtin_amdgcn_readfirstlane((int)floorf(qhi)) + 2 - cs) < LDS_W;
    const bool full = (te - tb == LDS_R);
    // 64-byte aligned (rows are padded to multiples of 32 entries, tb is a multiple of 16): one s_load_dwordx16 per chunk
    const unsigned* __restrict__ Brow = static_cast<const unsigned*>(__builtin_assume_aligned(Ball + tb, 64));
    f2v acc2 = {0.f, 0.f};
    if (fits) {
      // stage: rows tb .. tb+15 (beyond te: not fetched), columns cs .. cs+111 (outside the image: zeros); the row part of the
      // address is the wave-uniform soffset, the lane part is chunk-invariant but for the window start cs
      f4r v[7];
      const unsigned rowbase = (unsigned)tb * (unsigned)N * 4u;
      if (tdir) {
        // slot idx = lane + 64 i: window coordinate cw = idx / 4 (an image ROW cs + cw), marching steps 4 q .. 4 q + 3, q = idx % 4
        // (image COLUMNS tb + 4 q ..: inside the row because N % 4 == 0); element e goes to tile row 4 q + e, column cw.  The
        // transposed tile has row stride LDS_WT = 116: the four q of a row then fall into different banks
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const int idx = lane + 64 * i;
          const int cw = idx >> 2, q = idx & 3;
          const int row = cs + cw;
          const bool ok = (unsigned)row < (unsigned)N && tb + 4 * q < N;
          const int voff = ok ? (row * N + tb + 4 * q) * 4 : (int)img_bytes;
          v[i] = __builtin_bit_cast(f4r, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const int idx = lane + 64 * i;
          const int cw = idx >> 2, q = idx & 3;
#pragma unroll
          for (int e = 0; e < 4; ++e) T[(4 * q + e) * LDS_WT + cw] = v[i][e];
        }
      } else {
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int col = cs + sc4[i];
        bool ok = (unsigned)col < (unsigned)N;
        if (!full) ok = ok && (tb + (lane + 64 * i) / (LDS_W / 4) < te);
        const int voff = ok ? (col << 2) + srowN4[i] : (int)img_bytes;          // out of range: returns 0, fetches nothing
        if (DMA)   // straight into LDS (buffer_load_dwordx4 ... lds: lane l of load i lands at float4 64 i + l of the tile)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(T + 256 * i), 16, voff, rowbase, 0, 0);
        else
          v[i] = __builtin_bit_cast(f4r, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, rowbase, 0));
      }
      if (DMA) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): the tile is in LDS
      } else {
#pragma unroll
        for (int i = 0; i < 7; ++i) *reinterpret_cast<f4r*>(&T[slds[i]]) = v[i];
      }
      }
      __builtin_amdgcn_wave_barrier();
      const unsigned Ac = A - ((unsigned)cs << QF);               // column relative to the window (mod 256)
      auto march = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;         // FULL: 16 rows, all 64 rays live — no guards at all
        f2v w[LDS_R], t2[LDS_R];
#pragma unroll
        for (int u = 0; u < LDS_R; ++u) {
          const unsigned Q = Ac + Brow[u];                       // Brow is padded: rows beyond te read valid table entries
          const float f1 = (float)(Q << 8);                      // the 24 fraction bits, in units of 2^-32 (exact: 24 significant bits)
          w[u][1] = (FULL || tb + u < te) ? f1 : 0.f;
          w[u][0] = (FULL || tb + u < te) ? two32 - f1 : 0.f;
          unsigned c = Q >> QF;
          if (!FULL) c = c > (unsigned)(LDS_W - 2) ? (unsigned)(LDS_W - 2) : c;   // dead lanes / rows beyond te may point anywhere
          // byte address = row base [scalar, opaque to the optimiser so that it stays a scalar add and is not turned into a
          // per-lane one] + 4 c [one v_lshl_add]; both taps with one ds_read2_b32
          int rowoff4 = u * rs4;
          asm("" : "+s"(rowoff4));
          const float* tp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(T) + rowoff4 + (c << 2));
          t2[u] = (f2v){tp[0], tp[1]};
        }
#pragma unroll
        for (int u = 0; u < LDS_R; ++u) acc2 = __builtin_elementwise_fma(w[u], t2[u], acc2);
      };
      if (full && nlive == 64) march(std::true_type{});
      else march(std::false_type{});
      __builtin_amdgcn_wave_barrier();
    } else {
      // cannot happen for 64 rays and 16 rows unless the float sums above round unfavourably: direct gathers
#pragma unroll 1
      for (int k = 0; k < LDS_R / 8 && tb + 8 * k < te; ++k) {
        f2v w[8], v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int off = radon_edge_tap(tb + 8 * k + u, te, live, N, p.dq, base, A, Ball, w[u]);
          v[u] = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc2 = __builtin_elementwise_fma(w[u], v[u], acc2);
      }
    }
    total += (double)(acc2[0] + acc2[1]);
  }
  if (live) {
    if (FINAL) out[(int64_t)a * nd + d] = p.wgt * (float)total;
    else out[(int64_t)blockIdx.y * band_stride + (int64_t)a * nd + d] = (float)total;
  }
}

// Window-sharing forward kernel (N % 4 == 0, more than one band).  In k_radon_fwd_lds every wave fetches its own window: the
// texture path moves 7 KB per wave and 16 steps and is saturated (PMC: TD 98 % busy, VALU 54 %).  Rays of NEIGHBOURING ANGLES
// that cross the same columns need the same window, so here a workgroup is 4 consecutive angles x ONE image window:
//   * within a band, a ray belongs to the window its column at the band's TOP row falls into:
//         jj = floor((q(d, t0) + OFFS) / WO),   WO = floor(61 min_w |inv_w|) columns  (<= 61 rays of any of the 4 angles);
//     a wave takes the 64 detectors around the window's pre-image and keeps those whose fp32 estimate of q(d, t0) — one
//     expression, evaluated identically by every workgroup — lies in it, so every ray has exactly one owner per band;
//   * per chunk of 16 rows the union of the 4 waves' column ranges (a few columns wider than one wave's: angles 1 degree apart
//     drift < 12 columns over a 128-row band) is staged ONCE, 2 direct-to-LDS 16-byte loads per thread into a double-buffered
//     16 x 128 tile, one workgroup barrier per chunk; the march is that of k_radon_fwd_lds.
// If the four angles are not neighbours (arbitrary angle order), or the group mixes row- and column-driven angles, the
// union does not fit / the images differ: those chunks (groups) fall back to direct gathers — slower, same result.
// Rays that cannot touch the image inside a band are owned by no window: the band partials are zeroed before the launch.
#define WIN_R 16
#define WIN_W 128
#define WIN_MAXCH 32

// (7 waves per SIMD asked of the register allocator: 72 VGPRs, no spills; at the 78 it takes unasked the kernel ran 16 % longer)
template <int DUMMY = 0>
__global__ __launch_bounds__(256, 7) void k_radon_fwd_win(const float* __restrict__ img, const float* __restrict__ imgT,
                                                       float* __restrict__ out, int N, int nd,
                                                       const AngleParam* __restrict__ ang, int na_per_frame,
                                                       int ngrp_per_frame, int nwin, int64_t band_stride, int bh,
                                                       const float* __restrict__ fidx,
                                                       const unsigned* __restrict__ A32, const unsigned* __restrict__ B32, int npad) {
  __shared__ __attribute__((aligned(16))) float tile[2][WIN_R * WIN_W];
  __shared__ float ext[4][WIN_MAXCH][2];
  __shared__ int chinfo[WIN_MAXCH][2];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int grp = blockIdx.x / nwin, jj = blockIdx.x - grp * nwin;
  const int frame = grp / ngrp_per_frame;
  const int af0 = (grp - frame * ngrp_per_frame) * 4;            // first angle of the group within the frame
  const int nval = (na_per_frame - af0 < 4) ? na_per_frame - af0 : 4;   // valid waves: 0 .. nval-1
  const AngleParam* __restrict__ ag = ang + (int64_t)frame * na_per_frame + af0;
  // group-wide quantities (every wave computes the same scalars)
  float invmin = fabsf(ag[0].inv);
  bool mixed = false;
  for (int w = 1; w < nval; ++w) {
    invmin = fminf(invmin, fabsf(ag[w].inv));
    mixed = mixed || (ag[w].mode != ag[0].mode);
  }
  const int WO = (int)floorf(61.0f * invmin);
  const int OFFS = bh + 4;
  const int t0 = blockIdx.y * bh, t1 = (t0 + bh < N) ? t0 + bh : N;
  if ((int64_t)jj * WO - OFFS > (int64_t)N + bh + 4) return;     // window beyond every ray that can touch the band (uniform)
  const bool valid = wv < nval;
  const AngleParam p = ag[valid ? wv : 0];
  const int a = frame * na_per_frame + af0 + (valid ? wv : 0);
  const float* __restrict__ I = (p.mode ? imgT : img) + (int64_t)frame * N * N;
  const unsigned img_bytes = (unsigned)N * (unsigned)N * 4u;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)I, 0, img_bytes, 0x00020000);
  const float sdh = 0.5f * (float)(nd - 1);
  // candidate detectors: the 64 around the pre-image of the window [qa, qb) at row t0
  const float qa = (float)(jj * WO - OFFS), qb = (float)((jj + 1) * WO - OFFS);
  const float t0f = fidx[t0];
  const float offs = fmaf(t0f, p.dq, p.k0);
  const float dA = (qa - offs) * p.rinv + sdh, dB = (qb - offs) * p.rinv + sdh;
  const int dstart = (int)floorf(fminf(dA, dB)) - 1;
  const int d = dstart + lane;
  const float base = fmaf((float)d - sdh, p.inv, p.k0);
  const float qtop = fmaf(t0f, p.dq, base);
  const bool owned = valid && (unsigned)d < (unsigned)nd && qtop >= qa && qtop < qb;
  const unsigned long long omask = __builtin_amdgcn_ballot_w64(owned);
  const bool any = omask != 0ull;
  const int l_first = any ? __builtin_ctzll(omask) : 0, l_last = any ? 63 - __builtin_clzll(omask) : 0;
  const float b_first = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, base), l_first));
  const float b_last = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, base), l_last));
  const float blo = fminf(b_first, b_last), bhi = fmaxf(b_first, b_last);
  const int dcl = d < 0 ? 0 : (d >= nd ? nd - 1 : d);
  const unsigned A = A32[(int64_t)a * (nd + 2 * A32_PAD) + dcl + A32_PAD];
  const unsigned A_first = (unsigned)__builtin_amdgcn_readlane((int)A, l_first);
  const unsigned A_m = owned ? A : A_first;                       // lanes without a ray follow an owned one: always inside the tile
  const unsigned* __restrict__ Ball = B32 + (int64_t)a * npad;
  float two32 = 4294967296.0f;                     // kept in an SGPR (opaque to the optimiser): no 32-bit literal per step
  asm("" : "+s"(two32));
  double total = 0.0;

  if (mixed) {
    // the group straddles the 45-degree switch of the marching axis: no common image, every wave gathers for itself
    if (any) {
      for (int tb = t0; tb < t1; tb += 8) {
        f2v w[8], v[8], acc2 = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int off = radon_edge_tap(tb + u, t1, owned, N, p.dq, base, A, Ball, w[u]);
          v[u] = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc2 = __builtin_elementwise_fma(w[u], v[u], acc2);
        total += (double)(acc2[0] + acc2[1]);
      }
    }
  } else {
    // column range of every chunk, per wave -> LDS -> every wave knows the union (one barrier for the whole band)
    const int nch = (t1 - t0 + WIN_R - 1) / WIN_R;
    if (lane < nch) {
      const int tb = t0 + lane * WIN_R, te = (tb + WIN_R < t1) ? tb + WIN_R : t1;
      const float ta = (float)tb * p.dq, tz = (float)(te - 1) * p.dq;
      ext[wv][lane][0] = any ? blo + fminf(ta, tz) : 3.0e38f;
      ext[wv][lane][1] = any ? bhi + fmaxf(ta, tz) : -3.0e38f;
    }
    __syncthreads();
    // the union window of every chunk, once per band instead of once per wave and chunk (8 LDS reads, 6 min/max, two floors:
    // a sixth of the kernel's vector instructions were this bookkeeping): lane ch of wave 0 does chunk ch
    if (wv == 0 && lane < nch) {
      const float ulo = fminf(fminf(ext[0][lane][0], ext[1][lane][0]), fminf(ext[2][lane][0], ext[3][lane][0]));
      const float uhi = fmaxf(fmaxf(ext[0][lane][1], ext[1][lane][1]), fmaxf(ext[2][lane][1], ext[3][lane][1]));
      const bool nobody = ulo > uhi;                             // no wave owns a ray in this window
      const int cs = nobody ? 0 : (((int)floorf(ulo) - 1) & ~3);
      const bool fits = !nobody && ((int)floorf(nobody ? 0.f : uhi) + 2 - cs) < WIN_W;
      chinfo[lane][0] = cs;
      chinfo[lane][1] = fits ? 1 : 0;
    }
    __syncthreads();
    // staging slots of this thread: float4 numbers t and t + 256 of the 16 x 32 tile
    int sc4[2], srowN4[1], srow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = (int)threadIdx.x + 256 * i;
      srow[i] = idx / (WIN_W / 4);
      sc4[i] = (idx - srow[i] * (WIN_W / 4)) * 4;
    }
    static_assert(256 / (WIN_W / 4) == 8, "the two staging slots of a thread are 8 tile rows apart");
    srowN4[0] = srow[0] * N * 4;
    asm volatile("" : "+v"(srowN4[0]));              // one register, kept: not re-multiplied in every chunk
    for (int ch = 0; ch < nch; ++ch) {
      const int tb = t0 + ch * WIN_R, te = (tb + WIN_R < t1) ? tb + WIN_R : t1;
      const int buf = ch & 1;
      float* __restrict__ T = tile[buf];
      const int cs = __builtin_amdgcn_readfirstlane(chinfo[ch][0]);
      const bool fits = __builtin_amdgcn_readfirstlane(chinfo[ch][1]) != 0;
      const bool full = (te - tb == WIN_R);
      const unsigned* __restrict__ Brow = static_cast<const unsigned*>(__builtin_assume_aligned(Ball + tb, 64));
      f2v acc2 = {0.f, 0.f};
      if (fits) {
        const unsigned rowbase = (unsigned)tb * (unsigned)N * 4u;
        if (full) {
          // every row of the chunk is an image row: the thread's two slots are the same columns 8 rows apart, so ONE vector
          // offset serves both loads and the second row offset rides in the scalar offset (which the range check ignores: the
          // out-of-image columns are still caught through the vector offset).  4 vector instructions per chunk instead of 14,
          // two of them quarter-rate 32-bit multiplies the register allocator kept re-deriving.
          const int col = cs + sc4[0];
          const int voff = ((unsigned)col < (unsigned)N) ? (col << 2) + srowN4[0] : (int)img_bytes;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(T + (wv * 64) * 4), 16, voff, rowbase, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(T + (wv * 64 + 256) * 4), 16, voff,
                                                   rowbase + 8u * (unsigned)N * 4u, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int col = cs + sc4[i];
            const bool ok = ((unsigned)col < (unsigned)N) && (tb + srow[i] < te);
            const int voff = ok ? (col << 2) + srow[i] * N * 4 : (int)img_bytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(T + (wv * 64 + 256 * i) * 4), 16,
                                                     voff, rowbase, 0, 0);
          }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): this wave's part of the tile is in LDS
      }
      __syncthreads();                                           // tile complete; also: everyone is done with the other buffer
      if (fits && any) {
        const unsigned Ac = A_m - ((unsigned)cs << QF);
        auto march = [&](auto full_tag) {
          constexpr bool FULL = decltype(full_tag)::value;
          f2v w[WIN_R], t2[WIN_R];
#pragma unroll
          for (int u = 0; u < WIN_R; ++u) {
            const unsigned Q = Ac + Brow[u];
            const float f1 = (float)(Q << 8);                    // the 24 fraction bits, in units of 2^-32 (exact)
            w[u][1] = (FULL || tb + u < te) ? f1 : 0.f;
            w[u][0] = (FULL || tb + u < te) ? two32 - f1 : 0.f;
            unsigned c = Q >> QF;
            if (!FULL) c = c > (unsigned)(WIN_W - 2) ? (unsigned)(WIN_W - 2) : c;
            int rowoff4 = u * WIN_W * 4;
            asm("" : "+s"(rowoff4));
            const float* tp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(T) + rowoff4 + (c << 2));
            t2[u] = (f2v){tp[0], tp[1]};
          }
#pragma unroll
          for (int u = 0; u < WIN_R; ++u) acc2 = __builtin_elementwise_fma(w[u], t2[u], acc2);
        };
        if (full) march(std::true_type{});
        else march(std::false_type{});
      } else if (!fits && any) {
        // the four angles are too far apart for one window: direct gathers for this chunk
#pragma unroll 1
        for (int k = 0; k < WIN_R / 8 && tb + 8 * k < te; ++k) {
          f2v w[8], v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int off = radon_edge_tap(tb + 8 * k + u, te, owned, N, p.dq, base, A, Ball, w[u]);
            v[u] = __builtin_bit_cast(f2v, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0));
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc2 = __builtin_elementwise_fma(w[u], v[u], acc2);
        }
      }
      total += (double)(acc2[0] + acc2[1]);
    }
  }
  if (owned) out[(int64_t)blockIdx.y * band_stride + (int64_t)a * nd + d] = (float)total;
}

// Quad forward kernel (round 4): FOUR symmetric angles from one set of taps, conflict-free LDS gathers.
//
// What bounds k_radon_fwd_win (tools/microbench/issue_rate.hip, profiles/r04/issue_rate.txt): its march is seven vector
// instructions per ray and row of which five issue at 4 cycles, not 2 (an SGPR operand, v_cvt_f32_u32, three-source integer
// forms, v_pk_fma_f32): 25-30 cycles per wave and step on the SIMD; and its ds_read2_b32 — 64 rays 1 ... 1.41 columns apart span
// up to 91 columns, so a 32-lane group meets two addresses per bank — takes 8 LDS cycles instead of 4.  Both pipes level at
// ~9 cycles per wave-step and CU.  Two changes take each off the critical path:
//   * SYMMETRY.  With beta in [0, 45 deg], ct = cos(beta), t = tan(beta), q_b(s, i) = s/ct + h(1 - t) + i t  (h = (N-1)/2):
//       angle beta        rows of x,   q = q_b(s, i)                      slot 0
//       angle 180 - beta  rows of x,   q = (N-1) - q_b(s, i)              slot 1  (mirrored columns: taps swap their weights)
//       angle 90 - beta   rows of xT,  q = q_b(-s, j)                     slot 2  (detector index flipped)
//       angle 90 + beta   rows of xT,  q = (N-1) - q_b(s, j)              slot 3
//     (all four identities exact; checked against the oracle to 1e-15; the general sign cases are in radon_create_impl).  A wave
//     computes Q, both weights and the LDS address ONCE per step and uses them for the four members: the window [cs, cs + W) of
//     x and of xT (region F) and the mirrored window [N - cs - W, N - cs) of both (region M, stored in descending order so that its
//     address is one constant minus the forward address).  7 shared instructions + 1 (mirror address) + 4 packed FMAs per
//     256 ray-steps instead of 28: the vector unit drops to ~40 % and the march is bound by its four ds_read2_b32.
//   * HALF-WAVE WINDOWS.  A 32-lane group (what one LDS cycle serves for ds_read_b32) owns the rays whose column at the band's top
//     row lies in a 31-column interval: at any row its <= 32 columns are distinct mod 32 — no bank conflicts by construction
//     (lanes without a ray repeat an owned address: broadcast), at 31 / (32 inv) of the lanes busy.  Measured in isolation:
//     17.7 cycles per 256 ray-steps and CU against 36.7 for the march above.
// The member tables A32 / B32 / CB of every angle are DERIVED from its quad's base tables at create time, so the adjoint (which
// reads the members' tables) still sees bit-identical weights.  Angles without partners run as quads with fewer members.
// Workgroup = 4 waves = 4 quads of neighbouring beta sharing the staged tiles; chunks of QD_R = 8 rows, double-buffered with a true
// prefetch (the taps are inline-assembly LDS reads, so the compiler does not drain the direct-to-LDS loads in front of them).
struct QuadParam {
  float inv, dq, k0, rinv;   // base geometry: inv = 1/cos(beta) in [1, sqrt 2], dq = tan(beta) in [0, 1], k0 = h (1 - dq), rinv = cos(beta)
  int am[4];                 // member angle of each slot (index within the frame), -1: none
  int flip;                  // bit m: member m writes detector nd - 1 - d
  int mask;                  // bit m: slot m has a member
  int pad0, pad1;
};
#define QD_R 8
#define QD_W 120                          // window width (floats): 62 owned columns + 8 rows of slope + the drift of 4 neighbouring quads + alignment
#define QD_MAXCH 32
#define QD_HALF 31
#define QD_WO (2 * QD_HALF)
// LDS layout of a region (F: windows as they are; M: mirrored windows): [row pair p][source: x, xT][row in pair][QD_W] floats.
//   * one wave-load (60 lanes x 16 bytes = 240 floats) fills the two rows of ONE source: full-width loads with one buffer
//     resource (half-masked loads per source cost the texture path twice as much per byte: measured, TD 87 % busy);
//   * the xT window sits QD_SRC = 240 floats behind the x window: inside the 8-bit offsets of ds_read2_b32, so ONE address register
//     serves both sources;
//   * region M stores pairs, rows and columns in DESCENDING order: address_M(u, 118 - k) = constant - address_F(u, k).
#define QD_SRC (2 * QD_W)                 // 240
#define QD_PAIR (2 * QD_SRC)              // 480
#define QD_REGION ((QD_R / 2) * QD_PAIR)  // 1920 floats

__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }

template <int NBUF>
__global__ __launch_bounds__(256, 4) void k_radon_fwd_quad(const float* __restrict__ img, const float* __restrict__ imgT,
                                                        float* __restrict__ out, int N, int nd,
                                                        const QuadParam* __restrict__ quads, int nq_per_frame, int ngrp_per_frame,
                                                        int na_per_frame, int nwin, int64_t band_stride, int bh,
                                                        const float* __restrict__ fidx, const unsigned* __restrict__ A32q,
                                                        const unsigned* __restrict__ B32q, int npad,
                                                        const AngleParam* __restrict__ ang, const unsigned* __restrict__ A32,
                                                        const unsigned* __restrict__ B32, const int* __restrict__ wg_list, int grid_x) {
  __shared__ __attribute__((aligned(16))) float tile[NBUF][2 * QD_REGION];
  __shared__ float ext[4][QD_MAXCH][2];
  __shared__ int chinfo[QD_MAXCH][2];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // wg_list (round 6): this launch runs only the workgroups of the full grid that k_radon_fwd_quadf leaves to it
  const int wg_id = wg_list ? wg_list[1 + blockIdx.x] : 0;
  const int blk_y = wg_list ? wg_id / grid_x : (int)blockIdx.y, blk_x = wg_list ? wg_id - blk_y * grid_x : (int)blockIdx.x;
  // workgroups b and b + 8 run on the same XCD (round-robin placement: speed only, never correctness): give every XCD one
  // contiguous eighth of the windows, for all quad groups — its L2 then holds an eighth of the band (and the mirrored eighth)
  // instead of every fourth window of all of it
  const int nw8 = (nwin + 7) >> 3;
  const int xcd = blk_x & 7, bidx = blk_x >> 3;
  const int grp = bidx / nw8, jj = xcd * nw8 + (bidx - grp * nw8);
  if (jj >= nwin) return;
  const int frame = grp / ngrp_per_frame;
  const int q0 = (grp - frame * ngrp_per_frame) * 4;
  const int nval = (nq_per_frame - q0 < 4) ? nq_per_frame - q0 : 4;
  const QuadParam* __restrict__ qg = quads + (int64_t)frame * nq_per_frame + q0;
  int smask = 0;                                                    // slots any of the workgroup's quads uses: what gets staged
  for (int w = 0; w < nval; ++w) smask |= qg[w].mask;
  const int t0 = blk_y * bh, t1 = (t0 + bh < N) ? t0 + bh : N;
  const int OFFS = bh + 4;
  const bool valid = wv < nval;
  const QuadParam p = qg[valid ? wv : 0];
  const int qrow = frame * nq_per_frame + q0 + (valid ? wv : 0);
  const int ndp = nd + 2 * A32_PAD;
  const unsigned img_bytes = (unsigned)N * (unsigned)N * 4u;
  const auto rsrc0 = __builtin_amdgcn_make_buffer_rsrc((void*)(img + (int64_t)frame * N * N), 0, img_bytes, 0x00020000);
  const auto rsrc1 = __builtin_amdgcn_make_buffer_rsrc((void*)((imgT ? imgT : img) + (int64_t)frame * N * N), 0, img_bytes, 0x00020000);
  const float sdh = 0.5f * (float)(nd - 1);
  // a half-wave owns the rays whose column at the band's top row lies in [qa, qb), 31 columns: its candidates are the 32
  // detectors from the first one inside (found exactly: the estimate of the interval's pre-image is good to a small fraction of a
  // detector, one test decides between its two possible values)
  const int hw = lane >> 5, li = lane & 31;
  const float qa = (float)(jj * QD_WO - OFFS + QD_HALF * hw), qb = qa + (float)QD_HALF;
  const float t0f = fidx[t0];
  const float offs = fmaf(t0f, p.dq, p.k0);
  const float dA = (qa - offs) * p.rinv + sdh;
  const int d0 = (int)ceilf(dA - 0.05f);
  const float qt0 = fmaf(t0f, p.dq, fmaf((float)d0 - sdh, p.inv, p.k0));
  const int d = d0 + (qt0 < qa ? 1 : 0) + li;
  const float base = fmaf((float)d - sdh, p.inv, p.k0);
  const float qtop = fmaf(t0f, p.dq, base);
  const bool owned = valid && p.mask != 0 && (unsigned)d < (unsigned)nd && qtop >= qa && qtop < qb;
  const unsigned long long omask = __builtin_amdgcn_ballot_w64(owned);
  const bool any = omask != 0ull;
  const unsigned om_lo = (unsigned)omask, om_hi = (unsigned)(omask >> 32);
  const int l_first = any ? __builtin_ctzll(omask) : 0, l_last = any ? 63 - __builtin_clzll(omask) : 0;
  const float blo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, base), l_first));   // inv > 0: increasing
  const float bhi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, base), l_last));
  const int dcl = d < 0 ? 0 : (d >= nd ? nd - 1 : d);
  const unsigned A = A32q[(int64_t)qrow * ndp + dcl + A32_PAD];
  // lanes without a ray follow the first owned ray of their half (of the other half if theirs owns none): a broadcast, inside the tile
  const int lf0 = om_lo ? __builtin_ctz(om_lo) : (om_hi ? 32 + __builtin_ctz(om_hi) : 0);
  const int lf1 = om_hi ? 32 + __builtin_ctz(om_hi) : lf0;
  const unsigned A_f0 = (unsigned)__builtin_amdgcn_readlane((int)A, lf0), A_f1 = (unsigned)__builtin_amdgcn_readlane((int)A, lf1);
  const unsigned A_m = owned ? A : (hw ? A_f1 : A_f0);
  const unsigned* __restrict__ Ball = B32q + (int64_t)qrow * npad;
  float two32v = 4294967296.0f;                    // in a VECTOR register: an SGPR operand would make the subtraction a 4-cycle issue
  asm volatile("" : "+v"(two32v));
  double total[4] = {0.0, 0.0, 0.0, 0.0};

  // column range of every chunk, per wave -> LDS -> the union over the four waves, once per band (dq >= 0: q grows with the row)
  const int nch = (t1 - t0 + QD_R - 1) / QD_R;
  if (lane < nch) {
    const int tb = t0 + lane * QD_R, te = (tb + QD_R < t1) ? tb + QD_R : t1;
    ext[wv][lane][0] = any ? blo + (float)tb * p.dq : 3.0e38f;
    ext[wv][lane][1] = any ? bhi + (float)(te - 1) * p.dq : -3.0e38f;
  }
  __syncthreads();
  if (wv == 0 && lane < nch) {
    const float ulo = fminf(fminf(ext[0][lane][0], ext[1][lane][0]), fminf(ext[2][lane][0], ext[3][lane][0]));
    const float uhi = fmaxf(fmaxf(ext[0][lane][1], ext[1][lane][1]), fmaxf(ext[2][lane][1], ext[3][lane][1]));
    const bool nobody = ulo > uhi;
    const int cs = nobody ? 0 : (((int)floorf(ulo) - 1) & ~3);
    const bool fits = !nobody && ((int)floorf(nobody ? 0.f : uhi) + 2 - cs) < QD_W;
    chinfo[lane][0] = cs;
    chinfo[lane][1] = nobody ? 2 : (fits ? 1 : 0);                // 2: no wave owns a ray here — nothing to stage, nothing to march
  }
  __syncthreads();

  // staging: wave wv fills row pair wv of region F and of region M, one load per source (lanes 0-59: row in pair = lane / 30,
  // four columns from 4 (lane % 30)).  Rows beyond te and columns outside the image arrive as zeros (offset out of range).
  const int sr = lane >= 30 ? 1 : 0, sk = (lane - 30 * sr) << 2;                   // chunk-invariant
  const int rowF = (2 * wv + sr) * N * 4, rowM = (2 * (QD_R / 2 - 1 - wv) + 1 - sr) * N * 4;
  const int uF = 2 * wv + sr, uM = 2 * (QD_R / 2 - 1 - wv) + 1 - sr;
  auto stage = [&](int ch, int cs) {
    const int tb = t0 + ch * QD_R, te = (tb + QD_R < t1) ? tb + QD_R : t1;
    float* __restrict__ T = tile[ch % NBUF];
    const unsigned rowbase = (unsigned)tb * (unsigned)N * 4u;
    const int colF = cs + sk, colM = (N - cs - QD_W) + sk;
    const bool okF = ((unsigned)colF < (unsigned)N) && (tb + uF < te), okM = ((unsigned)colM < (unsigned)N) && (tb + uM < te);
    const int voffF = okF ? (colF << 2) + rowF : (int)img_bytes, voffM = okM ? (colM << 2) + rowM : (int)img_bytes;
    auto* dF = (__attribute__((address_space(3))) void*)(T + wv * QD_PAIR);
    auto* dFt = (__attribute__((address_space(3))) void*)(T + wv * QD_PAIR + QD_SRC);
    auto* dM = (__attribute__((address_space(3))) void*)(T + QD_REGION + wv * QD_PAIR);
    auto* dMt = (__attribute__((address_space(3))) void*)(T + QD_REGION + wv * QD_PAIR + QD_SRC);
    if (lane < 60) {
      if (smask & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc0, dF, 16, voffF, rowbase, 0, 0);
      if (smask & 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, dFt, 16, voffF, rowbase, 0, 0);
      if (smask & 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc0, dM, 16, voffM, rowbase, 0, 0);
      if (smask & 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, dMt, 16, voffM, rowbase, 0, 0);
    }
  };

  f2v acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  // Every chunk's window in a register (lane ch holds chunk ch): inside the loop nothing the COMPILER sees touches LDS, so it has no
  // reason to drain the direct-to-LDS loads (it orders every LDS access it knows of behind all of them), and chunks can be
  // staged NBUF - 1 ahead.  A workgroup's chunks form a dependent chain — barrier, wait for a tile, march 8 rows — and at small
  // images (few workgroups per CU) the chain is bound by the latency of ONE staging round trip per chunk (512^2: 16 chunks x
  // 1.7 us); with three chunks in flight the round trips overlap.
  const int cs_all = lane < nch ? chinfo[lane][0] : 0, st_all = lane < nch ? chinfo[lane][1] : 2;
  const int per_stage = __builtin_popcount(smask & 15);          // wave-level load instructions one staged chunk issues
  auto st_of = [&](int c) { return c < nch ? __builtin_amdgcn_readlane(st_all, c) : 2; };
  auto cs_of = [&](int c) { return c < nch ? __builtin_amdgcn_readlane(cs_all, c) : 0; };
  auto wait_loads = [&](int later) {                             // until at most `later` of this wave's loads are outstanding
    switch (later) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
#pragma unroll
  for (int c = 0; c < NBUF - 1; ++c)
    if (st_of(c) == 1) stage(c, cs_of(c));
  for (int ch = 0; ch < nch; ++ch) {
    const int tb = t0 + ch * QD_R, te = (tb + QD_R < t1) ? tb + QD_R : t1;
    const float* __restrict__ T = tile[ch % NBUF];
    const int cs = cs_of(ch), st = st_of(ch);
    // The chunk's eight B32 entries in ONE scalar load, requested BEFORE the barrier (round 6).  Fetched one by one inside the march
    // (as rounds 4-5 had it: `Ac + Brow[u]` next to the reads) every entry came with `s_waitcnt lgkmcnt(0)` — scalar loads return
    // out of order, so a wait for one is a wait for everything counted in lgkmcnt, the LDS reads included: every step drained the
    // three steps of reads "in flight", and the waves issued 38 % of their resident time (profiles/r05/radon_4096_pmc.txt).
    typedef unsigned u8v __attribute__((ext_vector_type(8)));
    const u8v Bv = *reinterpret_cast<const u8v*>(__builtin_assume_aligned(Ball + tb, 32));
    // this wave's loads of chunk ch have landed (those of the chunks staged after it may still fly), then the workgroup meets:
    // chunk ch is complete in LDS and everybody has left the buffer chunk ch + NBUF - 1 goes into
    int later = 0;
#pragma unroll
    for (int c = 1; c < NBUF - 1; ++c) later += (st_of(ch + c) == 1) ? per_stage : 0;
    wait_loads(later);
    asm volatile("s_barrier" ::: "memory");
    if (st_of(ch + NBUF - 1) == 1) stage(ch + NBUF - 1, cs_of(ch + NBUF - 1));
    const bool full = (te - tb == QD_R);
    if (st == 1 && any) {
      const unsigned Ac = A_m - ((unsigned)cs << QF);
      const unsigned Toff = lds_off(T);
      unsigned Cm = 2u * Toff + 4u * (unsigned)(QD_REGION + (QD_R / 2 - 1) * QD_PAIR + QD_W + (QD_W - 2));
      asm volatile("" : "+v"(Cm));
      auto march = [&](auto full_tag, auto all_tag) {
        constexpr bool FULL = decltype(full_tag)::value, ALL = decltype(all_tag)::value;
        f2v w[QD_R];
        unsigned a0[QD_R], a1[QD_R];
        // weights and addresses of a step are made three steps ahead of their use, just before its reads are issued: the vector
        // work of step u + 3 runs while the reads of steps u .. u + 2 are in flight, and few of these registers are live at once
        auto prep = [&](int u) {
          const unsigned Q = Ac + Bv[u];
          float f1 = (float)(Q << 8);                             // the 24 fraction bits, in units of 2^-32 (exact)
          float f0 = two32v - f1;
          if (!FULL) {
            f1 = (tb + u < te) ? f1 : 0.f;
            f0 = (tb + u < te) ? f0 : 0.f;
          }
          w[u] = (f2v){f0, f1};
          unsigned c = Q >> QF;
          if (!FULL) c = c > (unsigned)(QD_W - 2) ? (unsigned)(QD_W - 2) : c;
          int rowoff = (int)Toff + ((u >> 1) * QD_PAIR + (u & 1) * QD_W) * 4;   // a scalar add (opaque to the optimiser, or it becomes a second vector add)
          asm("" : "+s"(rowoff));
          a0[u] = (c << 2) + (unsigned)rowoff;
          a1[u] = Cm - a0[u];
        };
        // three steps (12 reads) in flight; LDS returns in order, so "at most 8 outstanding" means step u has arrived
        f2v tA[QD_R], tB[QD_R], tC[QD_R], tD[QD_R];
        auto issue = [&](int u) {
          asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(tA[u]) : "v"(a0[u]));
          asm volatile("ds_read2_b32 %0, %1 offset0:240 offset1:241" : "=v"(tB[u]) : "v"(a0[u]));
          asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(tC[u]) : "v"(a1[u]));
          asm volatile("ds_read2_b32 %0, %1 offset0:240 offset1:241" : "=v"(tD[u]) : "v"(a1[u]));
          static_assert(QD_SRC == 240, "the offsets above are QD_SRC and QD_SRC + 1");
        };
        prep(0);
        issue(0);
        prep(1);
        issue(1);
        prep(2);
        issue(2);
#pragma unroll
        for (int u = 0; u < QD_R; ++u) {
          if (u + 3 < QD_R) prep(u + 3);
          if (u <= QD_R - 3) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
          else if (u == QD_R - 2) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          asm volatile("" : "+v"(tA[u]), "+v"(tB[u]), "+v"(tC[u]), "+v"(tD[u]));
          if (u + 3 < QD_R) issue(u + 3);
          if (ALL || (p.mask & 1)) acc[0] = __builtin_elementwise_fma(w[u], tA[u], acc[0]);
          if (ALL || (p.mask & 4)) acc[2] = __builtin_elementwise_fma(w[u], tB[u], acc[2]);
          // mirrored windows: the pair read at the mirrored address is (tap c+1, tap c): the weights swap halves
          if (ALL || (p.mask & 2)) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[1]) : "v"(w[u]), "v"(tC[u]));
          if (ALL || (p.mask & 8)) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[3]) : "v"(w[u]), "v"(tD[u]));
        }
      };
      if (p.mask == 15) {
        if (full) march(std::true_type{}, std::true_type{});
        else march(std::false_type{}, std::true_type{});
      } else {
        if (full) march(std::true_type{}, std::false_type{});
        else march(std::false_type{}, std::false_type{});
      }
    } else if (st == 0 && any) {
      // the four quads are too far apart for one window: every member gathers for itself from its own image and tables
#pragma unroll 1
      for (int m = 0; m < 4; ++m) {
        if (!((p.mask >> m) & 1)) continue;
        const int am_m = m == 0 ? p.am[0] : (m == 1 ? p.am[1] : (m == 2 ? p.am[2] : p.am[3]));   // (no dynamic index into p: it would move to scratch)
        const int a = frame * na_per_frame + am_m;
        const AngleParam pm = ang[a];
        const int dm = ((p.flip >> m) & 1) ? nd - 1 - d : d;
        const int dmc = dm < 0 ? 0 : (dm >= nd ? nd - 1 : dm);
        const float base_m = fmaf((float)dm - sdh, pm.inv, pm.k0);
        const unsigned Amm = A32[(int64_t)a * ndp + dmc + A32_PAD];
        const unsigned* __restrict__ Bm = B32 + (int64_t)a * npad;
        f2v w[8], v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int off = radon_edge_tap(tb + u, te, owned, N, pm.dq, base_m, Amm, Bm, w[u]);
          v[u] = __builtin_bit_cast(f2v, pm.mode ? __builtin_amdgcn_raw_buffer_load_b64(rsrc1, off, 0, 0)
                                                 : __builtin_amdgcn_raw_buffer_load_b64(rsrc0, off, 0, 0));
        }
        f2v am2 = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u) am2 = __builtin_elementwise_fma(w[u], v[u], am2);
        const double tm = (double)(am2[0] + am2[1]);
        total[0] += m == 0 ? tm : 0.0;
        total[1] += m == 1 ? tm : 0.0;
        total[2] += m == 2 ? tm : 0.0;
        total[3] += m == 3 ? tm : 0.0;
      }
    }
    if ((ch & 1) || ch == nch - 1) {                               // fp32 partial sums over 16 rows, then fp64 (as the other forward kernels)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        total[m] += (double)(acc[m][0] + acc[m][1]);
        acc[m] = (f2v){0.f, 0.f};
      }
    }
  }
  if (owned) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (!((p.mask >> m) & 1)) continue;
      const int dm = ((p.flip >> m) & 1) ? nd - 1 - d : d;
      out[(int64_t)blk_y * band_stride + ((int64_t)frame * na_per_frame + p.am[m]) * nd + dm] = (float)total[m];
    }
  }
}

// ---------------------------------------------------------------------------------------- forward, quads: plan + lean kernel (round 6)
// Counters of k_radon_fwd_quad at 4096^2 x 180 (profiles/r05/radon_4096_pmc.txt, r06): 241 M vector instructions of which the march
// itself is 134 M; 150 M scalar; the vector unit 68 % busy and the waves stalled at issue — the kernel is bound by its instruction
// count, and 44 % of it is bookkeeping that depends on the GEOMETRY only: which detector a lane owns, the column range of every chunk
// (two barriers and an LDS round per workgroup), whether a window fits, the clamps of partial chunks, four march variants, and
// scalar registers spilled to vector lanes by all of it.  k_radon_quad_plan works that out ONCE per operator, per workgroup of the grid
// (the same arithmetic, statement for statement, as k_radon_fwd_quad's prologue: the two kernels own the same rays), and
// k_radon_fwd_quadf is the march alone: whole chunks of eight rows, all four members, one window start per chunk from the plan.
// Workgroups it cannot serve (a window that does not fit, a ragged band, groups of mostly single angles) are LISTED by the plan and
// run by k_radon_fwd_quad as before; both write the same band partials, the same bits.
constexpr int QD_NONE = INT32_MIN;
struct QuadPlan {
  int cs[QD_MAXCH];              // window start of every chunk; QD_NONE: no wave owns a ray there
  int dfirst[4][2];              // [wave][half]: the detector of lane li = 0 of that half (lane li: + li)
  unsigned omask_lo[4], omask_hi[4];   // the lanes that own a ray
  int fast;                      // 1: k_radon_fwd_quadf; 0: k_radon_fwd_quad (listed); 2: nothing to do
  int pad[64 - QD_MAXCH - 8 - 8 - 1];
};
static_assert(sizeof(QuadPlan) == 256, "one plan entry = 256 bytes");

__global__ __launch_bounds__(256) void k_radon_quad_plan(int N, int nd, const QuadParam* __restrict__ quads, int nq_per_frame,
                                                         int ngrp_per_frame, int nwin, int bh, const float* __restrict__ fidx,
                                                         int have_xT, QuadPlan* __restrict__ plan, int* __restrict__ slow) {
  __shared__ float ext[4][QD_MAXCH][2];
  __shared__ int nfit;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  QuadPlan& P = plan[(int64_t)blockIdx.y * gridDim.x + blockIdx.x];
  const int nw8 = (nwin + 7) >> 3;
  const int xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
  const int grp = bidx / nw8, jj = xcd * nw8 + (bidx - grp * nw8);
  if (jj >= nwin) {
    if (threadIdx.x == 0) P.fast = 2;
    return;
  }
  const int frame = grp / ngrp_per_frame;
  const int q0 = (grp - frame * ngrp_per_frame) * 4;
  const int nval = (nq_per_frame - q0 < 4) ? nq_per_frame - q0 : 4;
  const QuadParam* __restrict__ qg = quads + (int64_t)frame * nq_per_frame + q0;
  int members = 0;
  for (int w = 0; w < nval; ++w) members += __builtin_popcount(qg[w].mask & 15);
  const int t0 = blockIdx.y * bh, t1 = (t0 + bh < N) ? t0 + bh : N;
  const int OFFS = bh + 4;
  const bool valid = wv < nval;
  const QuadParam p = qg[valid ? wv : 0];
  const float sdh = 0.5f * (float)(nd - 1);
  // ---- k_radon_fwd_quad's ownership, statement for statement
  const int hw = lane >> 5, li = lane & 31;
  const float qa = (float)(jj * QD_WO - OFFS + QD_HALF * hw), qb = qa + (float)QD_HALF;
  const float t0f = fidx[t0];
  const float offs = fmaf(t0f, p.dq, p.k0);
  const float dA = (qa - offs) * p.rinv + sdh;
  const int d0 = (int)ceilf(dA - 0.05f);
  const float qt0 = fmaf(t0f, p.dq, fmaf((float)d0 - sdh, p.inv, p.k0));
  const int dl0 = d0 + (qt0 < qa ? 1 : 0);
  const int d = dl0 + li;
  const float base = fmaf((float)d - sdh, p.inv, p.k0);
  const float qtop = fmaf(t0f, p.dq, base);
  const bool owned = valid && p.mask != 0 && (unsigned)d < (unsigned)nd && qtop >= qa && qtop < qb;
  const unsigned long long omask = __builtin_amdgcn_ballot_w64(owned);
  const bool any = omask != 0ull;
  const int l_first = any ? __builtin_ctzll(omask) : 0, l_last = any ? 63 - __builtin_clzll(omask) : 0;
  const float blo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, base), l_first));
  const float bhi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, base), l_last));
  const int nch = (t1 - t0 + QD_R - 1) / QD_R;
  if (lane < nch) {
    const int tb = t0 + lane * QD_R, te = (tb + QD_R < t1) ? tb + QD_R : t1;
    ext[wv][lane][0] = any ? blo + (float)tb * p.dq : 3.0e38f;
    ext[wv][lane][1] = any ? bhi + (float)(te - 1) * p.dq : -3.0e38f;
  }
  if (threadIdx.x == 0) nfit = 0;
  __syncthreads();
  if (li == 0) P.dfirst[wv][hw] = dl0;
  if (lane == 0) {
    P.omask_lo[wv] = (unsigned)omask;
    P.omask_hi[wv] = (unsigned)(omask >> 32);
  }
  if (wv == 0 && lane < QD_MAXCH) {
    int csv = QD_NONE;
    if (lane < nch) {
      const float ulo = fminf(fminf(ext[0][lane][0], ext[1][lane][0]), fminf(ext[2][lane][0], ext[3][lane][0]));
      const float uhi = fmaxf(fmaxf(ext[0][lane][1], ext[1][lane][1]), fmaxf(ext[2][lane][1], ext[3][lane][1]));
      const bool nobody = ulo > uhi;
      const int cs = nobody ? 0 : (((int)floorf(ulo) - 1) & ~3);
      const bool fits = !nobody && ((int)floorf(nobody ? 0.f : uhi) + 2 - cs) < QD_W;
      if (!nobody) {
        csv = cs;
        if (!fits) atomicAdd(&nfit, 1);
      }
    }
    P.cs[lane] = csv;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool whole = (t1 - t0) % QD_R == 0;
    const bool fast = have_xT && whole && nfit == 0 && 4 * members >= 3 * 4 * nval;
    P.fast = fast ? 1 : 0;
    if (!fast) {
      const int k = atomicAdd(&slow[0], 1);
      slow[1 + k] = blockIdx.y * gridDim.x + blockIdx.x;
    }
  }
}

__global__ __launch_bounds__(256, 4) void k_radon_fwd_quadf(const float* __restrict__ img, const float* __restrict__ imgT,
                                                            float* __restrict__ out, int N, int nd,
                                                            const QuadParam* __restrict__ quads, int nq_per_frame, int ngrp_per_frame,
                                                            int na_per_frame, int nwin, int64_t band_stride, int bh,
                                                            const unsigned* __restrict__ A32q, const unsigned* __restrict__ B32q, int npad,
                                                            const QuadPlan* __restrict__ plan) {
  __shared__ __attribute__((aligned(16))) float tile[2][2 * QD_REGION];
  const QuadPlan& P = plan[(int64_t)blockIdx.y * gridDim.x + blockIdx.x];
  if (P.fast != 1) return;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int nw8 = (nwin + 7) >> 3;
  const int xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
  const int grp = bidx / nw8;
  const int frame = grp / ngrp_per_frame;
  const int q0 = (grp - frame * ngrp_per_frame) * 4;
  const int nval = (nq_per_frame - q0 < 4) ? nq_per_frame - q0 : 4;
  const bool valid = wv < nval;
  const int qrow = frame * nq_per_frame + q0 + (valid ? wv : 0);
  const int t0 = blockIdx.y * bh;
  const int nch = ((t0 + bh < N ? bh : N - t0)) / QD_R;              // whole chunks only (the plan's condition)
  const int ndp = nd + 2 * A32_PAD;
  const unsigned img_bytes = (unsigned)N * (unsigned)N * 4u;
  const auto rsrc0 = __builtin_amdgcn_make_buffer_rsrc((void*)(img + (int64_t)frame * N * N), 0, img_bytes, 0x00020000);
  const auto rsrc1 = __builtin_amdgcn_make_buffer_rsrc((void*)(imgT + (int64_t)frame * N * N), 0, img_bytes, 0x00020000);
  const int hw = lane >> 5, li = lane & 31;
  const int d = P.dfirst[wv][hw] + li;
  const unsigned om_lo = P.omask_lo[wv], om_hi = P.omask_hi[wv];
  const bool any = (om_lo | om_hi) != 0u;
  const bool owned = (((hw ? om_hi : om_lo) >> li) & 1u) != 0u;
  const int dcl = d < 0 ? 0 : (d >= nd ? nd - 1 : d);
  const unsigned A = A32q[(int64_t)qrow * ndp + dcl + A32_PAD];
  // lanes without a ray follow the first owned ray of their half (of the other half if theirs owns none): a broadcast, inside the tile
  const int lf0 = om_lo ? __builtin_ctz(om_lo) : (om_hi ? 32 + __builtin_ctz(om_hi) : 0);
  const int lf1 = om_hi ? 32 + __builtin_ctz(om_hi) : lf0;
  const unsigned A_f0 = (unsigned)__builtin_amdgcn_readlane((int)A, lf0), A_f1 = (unsigned)__builtin_amdgcn_readlane((int)A, lf1);
  const unsigned A_m = owned ? A : (hw ? A_f1 : A_f0);
  const unsigned* __restrict__ Ball = B32q + (int64_t)qrow * npad + t0;
  float two32v = 4294967296.0f;                    // in a VECTOR register: an SGPR operand would make the subtraction a 4-cycle issue
  asm volatile("" : "+v"(two32v));
  const int cs_all = lane < QD_MAXCH ? P.cs[lane] : QD_NONE;
  // staging: wave wv fills row pair wv of region F and of region M, one load per source (lanes 0-59: row in pair = lane / 30, four
  // columns from 4 (lane % 30)); columns outside the image arrive as zeros (offset out of range).  Per chunk: the window start times
  // four plus a per-lane constant, and a range test — nothing else
  const int sr = lane >= 30 ? 1 : 0, sk = (lane - 30 * sr) << 2;
  const int cF = sk, cM = N - QD_W + sk;                                                  // column = cs + cF / cM - cs
  const int oF = ((2 * wv + sr) * N + sk) * 4, oM = ((2 * (QD_R / 2 - 1 - wv) + 1 - sr) * N + (N - QD_W + sk)) * 4;
  const unsigned row8 = (unsigned)QD_R * (unsigned)N * 4u;
  unsigned rowbase = (unsigned)t0 * (unsigned)N * 4u + row8;                              // of the chunk being staged (chunk 1 first)
  auto stage = [&](int buf, int cs, unsigned rb) {
    float* __restrict__ T = tile[buf];
    const int cs4 = cs << 2;
    const int voffF = ((unsigned)(cs + cF) < (unsigned)N) ? cs4 + oF : (int)img_bytes;
    const int voffM = ((unsigned)(cM - cs) < (unsigned)N) ? oM - cs4 : (int)img_bytes;
    auto* dF = (__attribute__((address_space(3))) void*)(T + wv * QD_PAIR);
    auto* dFt = (__attribute__((address_space(3))) void*)(T + wv * QD_PAIR + QD_SRC);
    auto* dM = (__attribute__((address_space(3))) void*)(T + QD_REGION + wv * QD_PAIR);
    auto* dMt = (__attribute__((address_space(3))) void*)(T + QD_REGION + wv * QD_PAIR + QD_SRC);
    if (lane < 60) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc0, dF, 16, voffF, rb, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, dFt, 16, voffF, rb, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc0, dM, 16, voffM, rb, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, dMt, 16, voffM, rb, 0, 0);
    }
  };
  f2v acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  double total[4] = {0.0, 0.0, 0.0, 0.0};
  int cs = __builtin_amdgcn_readlane(cs_all, 0);
  if (cs != QD_NONE) stage(0, cs, rowbase - row8);
  typedef unsigned u8v __attribute__((ext_vector_type(8)));
  for (int ch = 0; ch < nch; ++ch) {
    const int cs_nx = ch + 1 < nch ? __builtin_amdgcn_readlane(cs_all, ch + 1) : QD_NONE;
    // the chunk's eight B32 entries: one scalar load, requested before the barrier
    const u8v Bv = *reinterpret_cast<const u8v*>(__builtin_assume_aligned(Ball + ch * QD_R, 32));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's share of chunk ch has landed ...
    asm volatile("s_barrier" ::: "memory");                          // ... everybody's; and everybody has left the other buffer
    if (cs_nx != QD_NONE) stage((ch + 1) & 1, cs_nx, rowbase);
    rowbase += row8;
    if (cs != QD_NONE && any) {
      const unsigned Ac = A_m - ((unsigned)cs << QF);
      const unsigned Toff = lds_off(tile[ch & 1]);
      unsigned Cm = 2u * Toff + 4u * (unsigned)(QD_REGION + (QD_R / 2 - 1) * QD_PAIR + QD_W + (QD_W - 2));
      asm volatile("" : "+v"(Cm));
      f2v w[QD_R];
      unsigned a0[QD_R], a1[QD_R];
      auto prep = [&](int u) {
        const unsigned Q = Ac + Bv[u];
        const float f1 = (float)(Q << 8);                           // the 24 fraction bits, in units of 2^-32 (exact)
        const float f0 = two32v - f1;
        w[u] = (f2v){f0, f1};
        int rowoff = (int)Toff + ((u >> 1) * QD_PAIR + (u & 1) * QD_W) * 4;
        asm("" : "+s"(rowoff));
        a0[u] = ((Q >> QF) << 2) + (unsigned)rowoff;
        a1[u] = Cm - a0[u];
      };
      f2v tA[QD_R], tB[QD_R], tC[QD_R], tD[QD_R];
      auto issue = [&](int u) {
        asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(tA[u]) : "v"(a0[u]));
        asm volatile("ds_read2_b32 %0, %1 offset0:240 offset1:241" : "=v"(tB[u]) : "v"(a0[u]));
        asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(tC[u]) : "v"(a1[u]));
        asm volatile("ds_read2_b32 %0, %1 offset0:240 offset1:241" : "=v"(tD[u]) : "v"(a1[u]));
      };
      prep(0);
      issue(0);
      prep(1);
      issue(1);
      prep(2);
      issue(2);
#pragma unroll
      for (int u = 0; u < QD_R; ++u) {
        if (u + 3 < QD_R) prep(u + 3);
        if (u <= QD_R - 3) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        else if (u == QD_R - 2) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(tA[u]), "+v"(tB[u]), "+v"(tC[u]), "+v"(tD[u]));
        if (u + 3 < QD_R) issue(u + 3);
        acc[0] = __builtin_elementwise_fma(w[u], tA[u], acc[0]);
        acc[2] = __builtin_elementwise_fma(w[u], tB[u], acc[2]);
        // mirrored windows: the pair read at the mirrored address is (tap c+1, tap c): the weights swap halves
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[1]) : "v"(w[u]), "v"(tC[u]));
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[3]) : "v"(w[u]), "v"(tD[u]));
      }
    }
    if ((ch & 1) || ch == nch - 1) {                               // fp32 partial sums over 16 rows, then fp64 (as the other forward kernels)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        total[m] += (double)(acc[m][0] + acc[m][1]);
        acc[m] = (f2v){0.f, 0.f};
      }
    }
    cs = cs_nx;
  }
  if (owned) {
    const QuadParam p = quads[qrow];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (!((p.mask >> m) & 1)) continue;
      const int dm = ((p.flip >> m) & 1) ? nd - 1 - d : d;
      out[(int64_t)blockIdx.y * band_stride + ((int64_t)frame * na_per_frame + p.am[m]) * nd + dm] = (float)total[m];
    }
  }
}

// sino[a][d] = wgt_a * sum over bands (fixed order, fp64) of the band partial sums, then the optional epilogue
// a * sino + b * z.  One thread per (angle row, e = d + A32_PAD in [0, nd + 4)): the padding positions exist for REC, which
// also writes the adjoint's record {w S[d -], w S[d +], w S[d], A32[d]} of every position (what k_radon_adj_prep would
// make from the finished sinogram; the neighbours come through LDS, the two at the block's edges are recomputed).
// ssq_part != NULL: sum(out^2) of this block's outputs in ssq_part[blockIdx.x].
// ---------------------------------------------------------------------------------------- forward, band-resident (small images; round 5)
// At 512^2 (C3) k_radon_fwd_lds spends 45 % of its vector instructions on staging a window per wave and 16 rows (addresses of 7
// loads, window bookkeeping, the wait for the loads), and its 2 160 workgroups of four waves run in 1.7 rounds.  A band of 64 rows of a
// 512-wide image is 130 KB: it FITS the LDS of one CU.  Here a workgroup of 16 waves loads its band once — rows of the image for the
// row-driven angles, rows of the transposed image for the column-driven ones, zero columns either side — and every wave then marches
// (angle, 64 detectors) tasks through all 64 rows with no staging, no barrier and no load left in the loop: per 16 rows one window
// start (the tables know the column mod 256), then the seven instructions of a step.  Chunks of 16 rows in fp32, flushed to float64,
// as k_radon_fwd_lds sums them; the band partials go to the same array (64-row bands).  Grid: frames x {row bands, column bands} x
// slices of that mode's angle list (adj_ang: the angles sorted by mode), about one workgroup per CU.
constexpr int BR_ROWS = 64, BR_NW = 16, BR_NT = 64 * BR_NW, BR_PAD = 4, BR_NMAX = 1024;
#ifdef TRK_FWD_EXPERIMENT_FLUSH16                      // (A/B build switch: rounds 2-5's fp32 sums of 16 rows)
constexpr int BR_FLUSH_SHIFT = 4;
#else
constexpr int BR_FLUSH_SHIFT = 2;                      // fp32 sums of 4 rows, then float64
#endif
// rows per band: 64 where 64 x (N + 8) floats fit (N <= 512), else 32 (N <= 1024: 132 KB)
inline int br_rows(int N) { return (size_t)BR_ROWS * (N + 2 * BR_PAD) * 4 <= 150 * 1024 ? BR_ROWS : BR_ROWS / 2; }
__global__ __launch_bounds__(BR_NT, 4) void k_radon_fwd_band(const float* __restrict__ img, const float* __restrict__ imgT,
                                                             float* __restrict__ part, int N, int nd,
                                                             const AngleParam* __restrict__ ang, int na,
                                                             const AdjAngle* __restrict__ sorted, const int* __restrict__ n_mode0,
                                                             int nslice, int64_t band_stride,
                                                             const unsigned* __restrict__ A32, const unsigned* __restrict__ B32, int npad,
                                                             int have_xT, int rows) {
  extern __shared__ __attribute__((aligned(16))) float band[];   // rows x (N + 2 BR_PAD) floats, then the task counter
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int nbands = N / rows;
  const int nw = (int)(blockDim.x >> 6), nthr = (int)blockDim.x;   // 16 waves (one workgroup per CU) or 8 (narrow images: two per CU)
  int& next_task = *reinterpret_cast<int*>(band + rows * (N + 2 * BR_PAD));
  // blockIdx.x = ((frame * 2 + mode) * nbands + b) * nslice + slice
  int bid = blockIdx.x;
  const int slice = bid % nslice; bid /= nslice;
  const int b = bid % nbands; bid /= nbands;
  const int mode = bid & 1, frame = bid >> 1;
  const int n0 = n_mode0[frame];
  const int cnt = mode ? na - n0 : n0;                           // angles of this mode in the frame
  const int ndblk = (nd + 63) / 64;
  // the mode's (angle, 64 detectors) tasks in list order, dealt to the slices in equal contiguous shares
#ifdef TRK_BAND_EXPERIMENT_PAIR
  // TIMING EXPERIMENT ONLY (wrong results): every second angle of the mode's list, each task also summing a MIRRORED ray with the same
  // weights (what a pair of symmetric angles sharing one march would cost)
  const int all_tasks = (cnt / 2) * ndblk;
#else
  const int all_tasks = cnt * ndblk;
#endif
  const int task0 = (int)((int64_t)all_tasks * slice / nslice), task1 = (int)((int64_t)all_tasks * (slice + 1) / nslice);
  if (task1 <= task0) return;
  const int RS = N + 2 * BR_PAD;                                 // row stride in floats (a multiple of 4)
#ifdef TRK_RADON_BAND_EXPERIMENT
  // timing experiments only (tools/r05_band_exp.sh builds a separate library; results are wrong with either bit): have_xT & 2 = the
  // band load alone, & 4 = the march alone (over whatever the LDS holds)
  const bool x_loadonly = (have_xT & 2) != 0, x_noload = (have_xT & 4) != 0;
  have_xT &= 1;
  if (!x_noload)
#endif
  if (mode && !have_xT) {
    // no transposed copy at hand: the band of the transposed image is 64 COLUMNS of the image — a wave-load takes 16 image rows x 16
    // columns (whole 64-byte sectors), a lane's four values go to four rows of the band (consecutive lanes: consecutive addresses)
    const float* __restrict__ X = img + (int64_t)frame * N * N + (int64_t)b * rows;
    const int r = lane & 15, jq = lane >> 4;
    const int cgs = rows / 16, pieces = (N / 16) * cgs;            // (16-row group, 16-column group)
    for (int p0 = wv; p0 < pieces; p0 += 4 * nw) {
      f4r v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pc = p0 + u * nw;
        const int rg = pc / cgs, cg = pc - rg * cgs;
        v[u] = pc < pieces ? *reinterpret_cast<const f4r*>(X + (int64_t)(16 * rg + r) * N + 16 * cg + 4 * jq) : (f4r){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pc = p0 + u * nw;
        if (pc < pieces) {
          const int rg = pc / cgs, cg = pc - rg * cgs;
#pragma unroll
          for (int e = 0; e < 4; ++e) band[(16 * cg + 4 * jq + e) * RS + BR_PAD + 16 * rg + r] = v[u][e];
        }
      }
    }
  } else {
    // the band: 64 rows x N floats as float4, 8 (N = 512) per thread in flight
    const float* __restrict__ I = (mode ? imgT : img) + (int64_t)frame * N * N + (int64_t)b * rows * N;
    const int q4 = N / 4, tot = rows * q4;
    for (int i0 = threadIdx.x; i0 < tot; i0 += 8 * nthr) {
      f4r v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = i0 + u * nthr;
        v[u] = idx < tot ? *reinterpret_cast<const f4r*>(I + 4 * (int64_t)idx) : (f4r){0.f, 0.f, 0.f, 0.f};   // (row * N + 4 c4 = 4 idx)
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = i0 + u * nthr;
        if (idx < tot) {
          const int row = idx / q4, c4 = idx - row * q4;
          *reinterpret_cast<f4r*>(&band[row * RS + BR_PAD + 4 * c4]) = v[u];
        }
      }
    }
  }
  // the pads are zeros
  if (threadIdx.x < rows * 2) {
    const int row = threadIdx.x >> 1, side = threadIdx.x & 1;
    *reinterpret_cast<f4r*>(&band[row * RS + (side ? BR_PAD + N : 0)]) = (f4r){0.f, 0.f, 0.f, 0.f};
  }
  if (threadIdx.x == 0) next_task = task0 + nw;
  __syncthreads();
#ifdef TRK_RADON_BAND_EXPERIMENT
  if (x_loadonly) {
    if (threadIdx.x == 0) part[blockIdx.x] = band[blockIdx.x & 1023];
    return;
  }
#endif
  float two32 = 4294967296.0f;
  asm("" : "+s"(two32));
  const int t0 = b * rows;
  const float sdh = 0.5f * (float)(nd - 1);
  const int ndp = nd + 2 * A32_PAD;
  // a wave takes the next task when it has finished one (tasks at the image's edge and beside it cost differently).  (Fetching the
  // NEXT task's angle constants and table entries while the current one is marched, and the four chunks' windows at once, one per lane:
  // measured 21.8 us against 20.4 — not kept.  Nor the band in two halves, rows 32-63 still in flight while every wave marches the first
  // two chunks of its first task: 21.9 us.  Nor the row stride as a compile-time constant with the row offsets as immediates of two hand-issued
  // ds_read_b32 per step (no scalar instruction per step: 80 -> 45 per chunk, twice the LDS instructions): 27.3 us per plain apply against 26.2.)
  for (int task = task0 + wv; task < task1;) {
#ifdef TRK_BAND_EXPERIMENT_PAIR
    const int ai = 2 * (task / ndblk), dblk = task - (ai / 2) * ndblk;
    double total_m = 0.0;
#else
    const int ai = task / ndblk, dblk = task - ai * ndblk;
#endif
    const int a = frame * na + sorted[frame * na + (mode ? n0 : 0) + ai].orig;                    // (scalar loads)
    const AngleParam p = ang[a];
    const int nlive = (nd - dblk * 64 < 64) ? nd - dblk * 64 : 64;
    const bool live = lane < nlive;
    const int d = dblk * 64 + lane;
    const unsigned A = A32[(int64_t)a * ndp + (live ? d : nd - 1) + A32_PAD];      // dead lanes repeat the last ray; never stored
    const unsigned* __restrict__ Ball = B32 + (int64_t)a * npad;
    const float b0 = fmaf((float)(dblk * 64) - sdh, p.inv, p.k0), b1 = fmaf((float)(dblk * 64 + nlive - 1) - sdh, p.inv, p.k0);
    const float blo = fminf(b0, b1), bhi = fmaxf(b0, b1);
    double total = 0.0;
#pragma unroll 1
    for (int c = 0; c < rows / 16; ++c) {
      const int tb = t0 + 16 * c;
      const float ta = (float)tb * p.dq, tz = (float)(tb + 15) * p.dq;
      const float qlo = blo + fminf(ta, tz), qhi = bhi + fmaxf(ta, tz);
      if (__builtin_amdgcn_readfirstlane((qhi < -2.f || qlo > (float)N + 1.f) ? 1 : 0)) continue;      // nothing of the wave touches the image here
      const int cs = __builtin_amdgcn_readfirstlane((int)floorf(qlo) - 1);
      const int ce = __builtin_amdgcn_readfirstlane((int)floorf(qhi) + 2);
      const unsigned* __restrict__ Brow = static_cast<const unsigned*>(__builtin_assume_aligned(Ball + tb, 64));
      const unsigned Ac = A - ((unsigned)cs << QF);              // column relative to cs (mod 256: the window is < 256 wide)
      const char* rowp = reinterpret_cast<const char*>(band) + (16 * c) * RS * 4;
      // fp32 sums of FOUR rows, then float64 (round 6; rounds 2-5: of sixteen).  The float64 instrument (profiles/r05/c3_instrument.txt)
      // showed what the longer fp32 chains cost where the solver amplifies roundings — iterates 5-7 of C3's transient sat 44-90 x
      // above the fp32-storage floor with 16-row sums and on it with 4-row sums (R.set_ref_sums(4, 32))
      f2v acc2[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#ifdef TRK_BAND_EXPERIMENT_PAIR
      f2v acc2m[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#endif
      if (cs >= 0 && ce <= N - 1) {
        // every tap of the wave inside the image: address = row (scalar) + 4 (cs + pad) (scalar) + 4 * relative column
        f2v w[16], t2[16];
#ifdef TRK_BAND_EXPERIMENT_PAIR
        f2v t2m[16];
#endif
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const unsigned Q = Ac + Brow[u];
          const float f1 = (float)(Q << 8);
          w[u][1] = f1;
          w[u][0] = two32 - f1;
          int off = u * RS * 4 + (cs + BR_PAD) * 4;
          asm("" : "+s"(off));
          const float* tp = reinterpret_cast<const float*>(rowp + off + ((Q >> QF) << 2));
          t2[u] = (f2v){tp[0], tp[1]};
#ifdef TRK_BAND_EXPERIMENT_PAIR
          int offm = u * RS * 4 + (BR_PAD + N - 2 - cs) * 4;
          asm("" : "+s"(offm));
          const float* tpm = reinterpret_cast<const float*>(rowp + offm - ((Q >> QF) << 2));
          t2m[u] = (f2v){tpm[0], tpm[1]};
#endif
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc2[u >> BR_FLUSH_SHIFT] = __builtin_elementwise_fma(w[u], t2[u], acc2[u >> BR_FLUSH_SHIFT]);
#ifdef TRK_BAND_EXPERIMENT_PAIR
#pragma unroll
        for (int u = 0; u < 16; ++u) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(acc2m[u >> BR_FLUSH_SHIFT]) : "v"(w[u]), "v"(t2m[u]));
#endif
      } else {
        // the window overhangs the image: columns clamped into the zero pads ([-2, N]: both taps of a clamped step read zeros)
        f2v w[16], t2[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const unsigned Q = Ac + Brow[u];
          const float f1 = (float)(Q << 8);
          w[u][1] = f1;
          w[u][0] = two32 - f1;
          int col = cs + (int)(Q >> QF);
          asm("v_med3_i32 %0, %1, -2, %2" : "=v"(col) : "v"(col), "s"(N));
          int off = u * RS * 4 + BR_PAD * 4;
          asm("" : "+s"(off));
          const float* tp = reinterpret_cast<const float*>(rowp + off + (col << 2));
          t2[u] = (f2v){tp[0], tp[1]};
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc2[u >> BR_FLUSH_SHIFT] = __builtin_elementwise_fma(w[u], t2[u], acc2[u >> BR_FLUSH_SHIFT]);
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) total += (double)(acc2[g4][0] + acc2[g4][1]);
#ifdef TRK_BAND_EXPERIMENT_PAIR
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) total_m += (double)(acc2m[g4][0] + acc2m[g4][1]);
#endif
    }
    if (live) part[(int64_t)b * band_stride + (int64_t)a * nd + d] = (float)total;
#ifdef TRK_BAND_EXPERIMENT_PAIR
    if (live && ai + 1 < cnt) part[(int64_t)b * band_stride + (int64_t)(frame * na + sorted[frame * na + (mode ? n0 : 0) + ai + 1].orig) * nd + d] = (float)total_m;
#endif
    int nx = 0;
    if (lane == 0) nx = atomicAdd(&next_task, 1);
    task = __builtin_amdgcn_readfirstlane(nx);
  }
}

template <bool REC>
__global__ __launch_bounds__(256) void k_radon_bands_post(const float* __restrict__ part, int nb, int64_t band_stride,
                                                          float* __restrict__ sino, int nd, const AngleParam* __restrict__ ang,
                                                          Epi epi, double* __restrict__ ssq_part,
                                                          uint4* __restrict__ rec, const int4* __restrict__ adj_pos,
                                                          const AdjAngle* __restrict__ adj_ang, const float* __restrict__ adj_wgt,
                                                          const unsigned* __restrict__ A32) {
  __shared__ double lds[4];
  __shared__ float sv[258];
  const int ndp = nd + 2 * A32_PAD;
  const int64_t rows = band_stride / nd;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = idx / ndp;
  const int e = (int)(idx - row * ndp), d = e - A32_PAD;
  const bool valid = row < rows;
  // the loads first, the coefficients (which may wait for the pending partials) after: two latency chains side by side
  struct Raw { float o, z; };
  auto raw = [&](int64_t r, int dd) -> Raw {
    if (dd < 0 || dd >= nd) return Raw{0.f, 0.f};
    const int64_t k = r * nd + dd;
    double t = 0.0;
    for (int b0 = 0; b0 < nb; b0 += 8) {           // eight band partials in flight (all of a 512-row image's 64-row bands), added in band order
      float pv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) pv[u] = (b0 + u < nb) ? part[(int64_t)(b0 + u) * band_stride + k] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) t += (double)pv[u];
    }
    return Raw{ang[r].wgt * (float)t, (epi.on && epi.z) ? epi.z[k] : 0.f};
  };
  const Raw r0 = valid ? raw(row, d) : Raw{0.f, 0.f};
  const float dv = (epi.dot_part && valid && d >= 0 && d < nd) ? epi.dotv[row * nd + d] : 0.f;
  Raw rm{0.f, 0.f}, rp{0.f, 0.f};
  if (REC) {
    if (threadIdx.x == 0 && valid && e > 0) rm = raw(row, d - 1);
    if (threadIdx.x == 255 && valid && e < ndp - 1) rp = raw(row, d + 1);
  }
  int64_t rs = 0;
  float w = 0.f;
  bool flip = false;
  unsigned a32 = 0u;
  if (REC && valid) {
    const int4 rr = adj_pos[row];
    rs = rr.x;
    w = __builtin_bit_cast(float, rr.y);
    flip = rr.z != 0;
    a32 = A32[row * ndp + e];
  }
  float ca, cb;
  double cad, cbd;
  epi_coefs(epi, blockIdx.x == 0, &lds[0], ca, cb, nullptr, &cad, &cbd);
  auto fin = [&](const Raw& v) -> float { return epi.on ? epi_combine(epi.on, ca, cb, cad, cbd, v.o, v.z, epi.z != nullptr) : v.o; };
  const bool inr = valid && d >= 0 && d < nd;
  const float v0 = inr ? fin(r0) : 0.f;
  if (inr) sino[row * nd + d] = v0;
  if (REC) {
    sv[threadIdx.x + 1] = v0;
    if (threadIdx.x == 0) sv[0] = (valid && e > 0 && d - 1 >= 0 && d - 1 < nd) ? fin(rm) : 0.f;
    if (threadIdx.x == 255) sv[257] = (valid && e < ndp - 1 && d + 1 >= 0 && d + 1 < nd) ? fin(rp) : 0.f;
    __syncthreads();
    if (valid) {
      const float vm = e > 0 ? sv[threadIdx.x] : 0.f, vp = e < ndp - 1 ? sv[threadIdx.x + 2] : 0.f;
      const float sm = w * vm, sp = w * vp;          // (0 outside the detector)
      uint4 o;
      o.x = __builtin_bit_cast(unsigned, flip ? sp : sm);
      o.y = __builtin_bit_cast(unsigned, flip ? sm : sp);
      o.z = __builtin_bit_cast(unsigned, w * v0);
      o.w = a32;
      rec[rs * ndp + e] = o;
    }
  }
  if (ssq_part) {                                                 // uniform over the grid
    const double q = block_sum<256>((double)v0 * v0, lds);
    if (threadIdx.x == 0) ssq_part[blockIdx.x] = q;
  }
  if (epi.dot_part) {                                             // uniform over the grid
    const double q = block_sum<256>((double)v0 * dv, lds);
    if (threadIdx.x == 0) epi.dot_part[blockIdx.x] = q;
  }
}

// ---------------------------------------------------------------------------------------- adjoint (gather)
// The forward weights of ray d on its two taps are (1-f, f) with f = q - floor(q), i.e. hat(q - col) = max(0, 1 - |q - col|) on
// pixel `col`.  Per pixel and angle the gather takes the ray d0 nearest to the pixel's inverse image d* (fp32 estimate) and
// its two neighbours — every ray with |q - col| < 1 is among them because |dq/dd| = 1/|cos| >= 1:
//   * t0 = q(d0, tt) - col comes from the SAME tables as the forward, as an integer: t_int = A32[d0] + B32[tt] - (col << 24)
//     (mod 2^32, |t0| <= 0.71 + the estimate's error), so hat(t0) = 1 - |t_int| 2^-24 is bit-identical to the forward's weight;
//   * the neighbours sit at t0 +- |inv| >= 1 away on either side, so their weights are clamp(c1 + t0) and clamp(c1 - t0),
//     c1 = 1 - |inv| <= 0: ONE packed FMA with the hardware clamp to [0, 1] (one fp32 rounding, 6e-8).
// What is read per pixel and angle is ONE 16-byte record {w S[d0 -], w S[d0 +], w S[d0], A32[d0]} (w = the angle's weight;
// -/+: the neighbour on the smaller-q / larger-q side, which is d0 -+ 1 or d0 +- 1 by the sign of inv), written per apply by
// k_radon_adj_prep for the angles sorted by marching mode.
//
// k_radon_adj_tile: a workgroup owns a T x T pixel tile and walks the angles in batches of AB (8 or 16).  Per batch it stages,
// with direct-to-LDS loads, (i) for every angle the 64 records around the tile's inverse image, as a RING indexed by d0 & 63
// (the tile's footprint is < 48 detectors, so no index arithmetic beyond a mask is needed to read a record), and (ii) the
// pairs {C[a][tt], B32[a][tt]} of the tile's marching indices (C: the locator offset, d* = col rinv + C).  A thread holds
// PX pixels that share the marching index — a run along the row for mode-0 angles, along the column for mode-1 angles — so
// that pair is read once per angle and thread; the two partial images meet through LDS at the end.  Per pixel and angle:
// 9.5 vector instructions and one ds_read_b128 — d* (packed FMA for two pixels), its rounding (a packed add of 1.5 x 2^23:
// the integer sits in the low mantissa bits), ring address (and, shift-add), t_int (one three-operand add), its conversion,
// the centre weight (one FMA), both neighbour weights (one packed FMA with clamp), two accumulating FMAs (one packed).  The
// kernel is bound by vector-instruction issue (4 cycles per wave instruction): the first gather form of round 1 needed 34
// instructions and three dword loads on the texture path, the second 26 and one 12-byte load, this kernel's first version 15.
constexpr int ADJ_T = 32;      // tile edge (pixels) of the large-image instantiation; small images: 16 x 16, one pixel per thread
constexpr float RND_MAGIC = 12582912.0f;   // 1.5 * 2^23: x + RND_MAGIC has rint(x) in its low mantissa bits (|x| < 2^22)

// records of one vector: rec[(frame*na + sorted angle)][e], e = d + 2 in [0, nd + 3]
__global__ __launch_bounds__(256) void k_radon_adj_prep(const float* __restrict__ sino, uint4* __restrict__ rec, int nd, int na,
                                                        const AdjAngle* __restrict__ ang, const float* __restrict__ wgt,
                                                        const unsigned* __restrict__ A32) {
  const int ndp = nd + 2 * A32_PAD;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // over (sorted angle of the frame) x ndp; blockIdx.y = frame
  const int64_t r = idx / ndp;
  if (r >= (int64_t)na) return;
  const int e = (int)(idx - r * ndp);
  const int64_t rs = (int64_t)blockIdx.y * na + r;                   // sorted row (frame-major)
  const int64_t ro = (int64_t)blockIdx.y * na + ang[rs].orig;        // the same angle in the caller's order
  const int d = e - A32_PAD;
  const float w = wgt[rs];
  const float* __restrict__ S = sino + ro * nd;
  const float sm = (d - 1 >= 0 && d - 1 < nd) ? w * S[d - 1] : 0.f, sp = (d + 1 >= 0 && d + 1 < nd) ? w * S[d + 1] : 0.f;
  const bool flip = ang[rs].flip != 0;
  uint4 o;
  o.x = __builtin_bit_cast(unsigned, flip ? sp : sm);             // the neighbour at t0 - |inv|
  o.y = __builtin_bit_cast(unsigned, flip ? sm : sp);             // the neighbour at t0 + |inv|
  o.z = __builtin_bit_cast(unsigned, (d >= 0 && d < nd) ? w * S[d] : 0.f);
  o.w = A32[ro * ndp + e];
  rec[rs * ndp + e] = o;
}

// LDS by byte offset: the ring slot of detector d0 is (d0 & 63) * 16 behind the ring's base, which is one v_and_b32 and one
// v_lshl_add_u32 with the (wave-uniform) base in an SGPR — spelled out, or the optimiser turns it into shift + mask + add
__device__ __forceinline__ unsigned lds_offset(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
typedef unsigned u4r __attribute__((ext_vector_type(4)));     // a record {w S[d0 -], w S[d0 +], w S[d0], A32[d0]}
// The read is issued as inline assembly: written as a C++ load, the compiler orders it after EVERY outstanding direct-to-LDS
// load (it cannot see that the prefetch of batch b + 1 lands in the other ring buffer) and put s_waitcnt vmcnt(0) in front of each
// record read — the prefetch issued a few instructions earlier was waited for before the gather of batch b began, twelve exposed
// L2 round trips per tile at 512^2 x 180.  The counterpart of hiding the read: the CALLER waits (ring_wait) before using r.
__device__ __forceinline__ u4r ring_read(unsigned ring_base, unsigned bits) {
  unsigned addr;
  const unsigned slot = bits & 63u;
  asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(addr) : "v"(slot), "s"(ring_base));
  u4r r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
  return r;
}
// a {C, B32} pair by inline assembly, for the same reason (a C++ LDS load in the angle loop was given `s_waitcnt vmcnt(0)`: the wait
// for the NEXT batch's direct-to-LDS loads).  The caller waits (ring_wait) and ties (pair_tie) before using OR COPYING it: the data
// lands after the instruction has issued, so nothing may touch the destination registers in between — no conditional assignment
// (a join would copy them), no element-wise repacking
typedef unsigned u2r __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u2r pair_read(const void* p) {
  u2r r;
  asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(lds_offset(p)) : "memory");
  return r;
}
__device__ __forceinline__ void pair_tie(u2r& r) { asm volatile("" : "+v"(r)); }
// all of this wave's LDS reads have returned; ring_tie makes a record's uses depend on the wait (volatile asm keeps its order)
__device__ __forceinline__ void ring_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void ring_tie(u4r& r) { asm volatile("" : "+v"(r)); }

// one pixel, one angle: the record, t0 from the tables, three hat weights.  accn: the two neighbour terms (packed), acc0: the centre
__device__ __forceinline__ void adj_gather(const u4r r, unsigned B, unsigned negcol24, f2v sc2, float nsc, f2v cr, f2v& accn,
                                           float& acc0) {
  // NOTE the elements are copied to scalars first: __builtin_bit_cast(float, r[k]) on an ext-vector ELEMENT reads element 0
  // whatever k is (hipcc / ROCm 7.2; found the hard way — the adjoint summed (w0 + wp + wm) S[d0-1])
  const unsigned slo = r[0], shi = r[1], s0 = r[2], a32 = r[3];
  unsigned ti;                                                   // t_int = A32 + B32 - (col << 24), wrap-around mod 2^32 is the point
  asm("v_add3_u32 %0, %1, %2, %3" : "=v"(ti) : "v"(a32), "v"(B), "v"(negcol24));
  const float tf = (float)(int)ti;                               // t0 in units of 2^-24, exact
  // {clamp(c1m + t0), clamp(c1p - t0)} in one packed FMA: both lanes read the LOW half of t2 and of the scale (op_sel_hi 0), each its own half of the addend; the high
  // lane negates the scale 2^-24.  64-bit operands must sit in even-aligned register pairs, hence the two-element carriers
  // whose high halves are never read.  The one scalar operand an instruction may have is the angle's {c1, rinv} pair as it came
  // from the scalar load (the scale, loop-invariant, lives in a vector pair): with c1 as the vector operand every angle paid a
  // v_mov to get it there, one of its eleven vector instructions.
  f2v t2;
  t2[0] = tf;
  f2v wn;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,0,1] neg_hi:[0,1,0] clamp" : "=v"(wn) : "v"(t2), "v"(sc2), "s"(cr));
  float w0;                                                      // 1 - |t0|: exact (t0 is a multiple of 2^-24)
  asm("v_fma_f32 %0, |%1|, %2, 1.0" : "=v"(w0) : "v"(tf), "s"(nsc));
  const f2v sn = {__builtin_bit_cast(float, slo), __builtin_bit_cast(float, shi)};
  accn = __builtin_elementwise_fma(wn, sn, accn);
  acc0 = fmaf(w0, __builtin_bit_cast(float, s0), acc0);
}

// G > 1 (round 5; a 512^2 image: 256 tiles = one per CU): the parts of a tile are G groups of four waves of ONE workgroup of 1024 threads —
// the same angle ranges, the same batches, their own rings — whose partial tiles meet in LDS in part order: the same bits as the
// split over workgroups, without write-through partial tiles, tickets and a finisher that starts when everybody else is done.
template <int T, int PX, int AB, bool PREP, int G = 1>
__global__ __launch_bounds__(256 * G) void k_radon_adj_tile(const float* __restrict__ sino, const uint4* __restrict__ rec,
                                                        float* __restrict__ img, int N, int nd, int na,
                                                        const AdjAngle* __restrict__ ang, const float* __restrict__ wgt,
                                                        const unsigned* __restrict__ A32, const int* __restrict__ n_mode0,
                                                        const uint2* __restrict__ CB, int npad, int tiles_x,
                                                        double* __restrict__ ssq_part, Epi epi, float* __restrict__ xT_out,
                                                        int nsplit, float* __restrict__ part_img, unsigned* __restrict__ tile_cnt) {
  __shared__ __attribute__((aligned(16))) uint4 ring_all[G][2][AB][64];
  __shared__ __attribute__((aligned(16))) uint2 cbs_all[G][2][AB][T];
  __shared__ double xch_all[G][PX > 1 ? T : 1][T + 1];     // (float64 since round 6: the two modes' totals meet unrounded)
  static_assert(T * T == 256 * PX && (T == 16 || T == 32), "256 threads x PX pixels cover the T x T tile");
  static_assert(G == 1 || (PX == 4 && T == 32), "groups: the 32 x 32 form only");
  __shared__ double lds[4];
  const int tid = G > 1 ? (int)(threadIdx.x & 255) : (int)threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = G > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
  auto& ring = ring_all[grp];
  auto& cbs = cbs_all[grp];
  auto& xch = xch_all[grp];
  const int frame = blockIdx.y;
  // nsplit > 1 (small images: too few tiles to fill the chip with 32 x 32 tiles): workgroup (part, tile) gathers the sorted
  // angles [a_lo, a_hi) of the frame for its tile; the partial tiles meet in the workgroup that finishes LAST (below).  Parts of
  // one tile are ntiles workgroups apart: the same XCD when ntiles % 8 == 0 (speed only)
  const int ntiles = tiles_x * tiles_x;
  if (G > 1) nsplit = G;
  const int part = G > 1 ? grp : (nsplit > 1 ? blockIdx.x / ntiles : 0);
  const int tile_id = G > 1 ? (int)blockIdx.x : (int)blockIdx.x - part * ntiles;
  const int ty = tile_id / tiles_x, tx = tile_id - ty * tiles_x;
  const int i0 = ty * T, j0 = tx * T;
  const int ndp = nd + 2 * A32_PAD;
  const int a_lo = nsplit > 1 ? (int)(((int64_t)part * na) / nsplit) : 0;
  const int a_hi = nsplit > 1 ? (int)(((int64_t)(part + 1) * na) / nsplit) : na;
  sino += (int64_t)frame * na * nd;
  A32 += (int64_t)frame * na * ndp;
  CB += (int64_t)frame * na * npad;
  int n0 = n_mode0[frame] - a_lo;
  ang += (int64_t)frame * na + a_lo;                   // from here on `na` is the part's angle count and angle 0 its first
  wgt += (int64_t)frame * na + a_lo;
  rec += ((int64_t)frame * na + a_lo) * ndp;
  const int na_frame = na;
  na = a_hi - a_lo;
  n0 = n0 < 0 ? 0 : (n0 > na ? na : n0);
  const auto rrec = __builtin_amdgcn_make_buffer_rsrc((void*)rec, 0, (unsigned)((int64_t)na * ndp * 16), 0x00020000);
  const float sdh = 0.5f * (float)(nd - 1);
  const auto rcb = __builtin_amdgcn_make_buffer_rsrc((void*)CB, 0, (unsigned)((int64_t)na_frame * npad * 8), 0x00020000);

  // thread -> pixels.  mode 0 (marching index = row): row r0, columns c0 + k T/PX;  mode 1 (= column): column c1, rows
  // r1 + k T/PX, k < PX (PX = 1: the same pixel in both).  Neighbouring lanes hold NEIGHBOURING pixels, so the 16 lanes of a
  // ds_read_b128 group read records at most ~10 detectors apart: distinct LDS banks (a record is 4 banks wide, 16 records fill
  // the 64) or the same record (a broadcast).  With 4 consecutive pixels per lane instead, neighbouring lanes were up to 4
  // detectors apart and 35 % of the LDS cycles were bank conflicts (PMC).
  constexpr int TS = T / PX;                           // threads along the run direction = pixel stride of one thread
  const int r0 = tid / TS, c0 = tid % TS;
  const int r1 = PX > 1 ? tid % TS : tid / T, c1 = PX > 1 ? tid / TS : tid % T;   // (mode 1: neighbouring lanes = neighbouring ROWS)
  float fcolA[PX], fcolB[PX];
  unsigned colA[PX], colB[PX];
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    fcolA[k] = (float)(j0 + c0 + k * TS);            // mode 0: interpolated coordinate = column
    colA[k] = 0u - ((unsigned)(j0 + c0 + k * TS) << QF);   // negated: t_int = A32 + B32 - (col << 24)
    fcolB[k] = (float)(i0 + r1 + k * TS);            // mode 1: interpolated coordinate = row
    colB[k] = 0u - ((unsigned)(i0 + r1 + k * TS) << QF);
  }
  f2v anA[PX], anB[PX];
  float accA[PX], accB[PX];
  // fp32 sums over at most ADJ_FLUSH angles, then float64 (round 6; rounds 1-5: fp32 over all angles of the part).  The float64
  // instrument (profiles/r05/c3_instrument.txt) put the 180-angle fp32 sum one amplification step (x 6.5 per iteration of C3's
  // transient) above the fp32-storage floor, a 32-angle cadence on it (R.set_ref_sums(4, 32))
  constexpr int ADJ_FLUSH = 32;
  double totA[PX], totB[PX];
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    accA[k] = accB[k] = 0.f;
    anA[k] = anB[k] = (f2v){0.f, 0.f};
    totA[k] = totB[k] = 0.0;
  }
  f2v sc2 = {5.9604644775390625e-8f, 5.9604644775390625e-8f};          // 2^-24, kept in an (aligned) VGPR pair
  float nsc = -5.9604644775390625e-8f;
  asm("" : "+v"(sc2));
  asm("" : "+s"(nsc));

  const int nbatch = (na + AB - 1) / AB;
  // groups: every group meets every barrier — the batches of the LARGEST part (a part may hold one angle more than another)
  int nbatch_all = nbatch;
  if (G > 1) {
    nbatch_all = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int sz = (int)(((int64_t)(g + 1) * na_frame) / G) - (int)(((int64_t)g * na_frame) / G);
      nbatch_all = (sz + AB - 1) / AB > nbatch_all ? (sz + AB - 1) / AB : nbatch_all;
    }
  }
  // Staging of batch b into buffer b & 1: wave w takes the rings of angles w, w + 4, ... of the batch, lane l the detector whose
  // ring slot is l.  PREP: the records {w S[d -], w S[d +], w S[d], A32[d]} were written by k_radon_adj_prep and go straight
  // to LDS (one 16-byte direct-to-LDS load per lane and angle).  !PREP: they are made here from the sinogram itself — four
  // dwords per lane into registers while the previous batch is gathered, written to LDS afterwards — which saves the pre-pass
  // launch and pays when a frame has few angles (dynamic problems, 15 per frame: 32 frames 21.0 -> 20.0 us, 4 frames 11.0 ->
  // 8.4 us) but costs more instructions per record (180 angles: 512^2 46 -> 57 us, 4096^2 1.04 -> 1.32 ms).
  // The {C, B32} pairs of the tile's marching indices always go straight to LDS (16 bytes = two indices per thread).
  uint4 sreg[AB / 4];
  int cb_orig = 0;
  if (tid < AB * T / 2) {
    const int a = tid / (T / 2);
    cb_orig = ang[a < na ? a : na - 1].orig;
  }
  // lane l holds {rinv, dq, k0} of angle l (mod AB) of the batch stage_load is called for next, fetched a batch ahead
  float nx_rinv, nx_dq, nx_k0;
  auto fetch_angles = [&](int b) {
    int a = b * AB + (lane & (AB - 1));
    a = a < na ? a : na - 1;
    nx_rinv = ang[a].rinv;
    nx_dq = ang[a].dq;
    nx_k0 = ang[a].k0;
  };
  fetch_angles(0);
  int nam1;                                          // na - 1 as a value the vector unit has no copy of, so that the row
  asm("s_add_i32 %0, %1, -1" : "=s"(nam1) : "s"(na) : "scc");   // offsets below stay scalar arithmetic
  auto stage_load = [&](int b) {
    // inverse image of the tile: d* = col rinv + (sdh - (k0 + tt dq) rinv) is linear, so the tile's d* are centred on the
    // image of its centre and span at most (T - 1) sqrt(2) detectors (22 / 44 for T = 16 / 32): a 64-slot ring around the centre
    // holds them and their +-1 neighbours.  The ring bases of the batch's AB angles are computed by AB LANES, one angle each, and
    // handed out by v_readlane: the arithmetic is wave-uniform per angle but gfx950 has no scalar float unit — done per ring
    // (first from the four corners, 35 vector instructions per ring, then from the centre, 20) staging was 43 % / 30 % of the
    // 16 x 16 kernel's vector instructions at 512^2 x 180 (PMC).
    int dbase_l;
    {
      int a = b * AB + (lane & (AB - 1));
      a = a < na ? a : na - 1;
      const bool m1 = a >= n0;
      const float tt_c = (float)(m1 ? j0 : i0) + 0.5f * (float)(T - 1), co_c = (float)(m1 ? i0 : j0) + 0.5f * (float)(T - 1);
      dbase_l = (int)floorf(fmaf(co_c - fmaf(tt_c, nx_dq, nx_k0), nx_rinv, sdh)) - 32;
    }
    if (b + 1 < nbatch) fetch_angles(b + 1);
#pragma unroll
    for (int h = 0; h < AB / 4; ++h) {
      const int al = wv + 4 * h;
      int a;
      asm("s_min_i32 %0, %1, %2" : "=s"(a) : "s"(b * AB + al), "s"(nam1) : "scc");
      const int row = a * ndp;
      const int dbase = __builtin_amdgcn_readlane(dbase_l, al);      // ring covers dbase .. dbase + 63
      const int d = dbase + ((lane - dbase) & 63);                   // the detector whose ring slot is this lane
      int e;                                                         // beyond the detector: weightless (S = 0) anyway
      asm("v_med3_i32 %0, %1, 0, %2" : "=v"(e) : "v"(d + A32_PAD), "s"(ndp - 1));
      if (PREP) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rrec, (__attribute__((address_space(3))) void*)&ring[b & 1][al][0], 16,
                                                 (row + e) * 16, 0, 0, 0);
        continue;
      }
      const AdjAngle p = ang[a];
      const float* __restrict__ S = sino + (int64_t)p.orig * nd;
      const float w = wgt[a];
      const int dm = d - 1, dp = d + 1;
      const float sm = ((unsigned)dm < (unsigned)nd) ? w * S[dm] : 0.f;
      const float s0 = ((unsigned)d < (unsigned)nd) ? w * S[d] : 0.f;
      const float sp = ((unsigned)dp < (unsigned)nd) ? w * S[dp] : 0.f;
      sreg[h].x = __builtin_bit_cast(unsigned, p.flip ? sp : sm);    // the neighbour at t0 - |inv|
      sreg[h].y = __builtin_bit_cast(unsigned, p.flip ? sm : sp);    // the neighbour at t0 + |inv|
      sreg[h].z = __builtin_bit_cast(unsigned, s0);
      sreg[h].w = A32[(int64_t)p.orig * ndp + e];
    }
    if (tid < AB * T / 2) {
      const int buf = b & 1;
      const int al = tid / (T / 2), pr = tid - al * (T / 2);
      int a = b * AB + al;
      a = a < na ? a : na - 1;
      const int tt0 = (a >= n0 ? j0 : i0) + 2 * pr;                  // even: 16-byte aligned pairs (npad is even)
      // (the LDS address of a direct-to-LDS load is wave-uniform base + 16 * lane: wave 1 lands 1 KB behind wave 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rcb, (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(&cbs[buf][0][0]) + wv * 1024),
                                               16, (cb_orig * npad + tt0) * 8, 0, 0, 0);
      // the table row of this thread's pair in the batch after: fetched a batch ahead — read here, the load and the vmcnt(0) its
      // use needs sat between the ring loads and the gather, one exposed L2 round trip per batch for waves 0 and 1
      int an = (b + 1) * AB + al;
      an = an < na ? an : na - 1;
      cb_orig = ang[an].orig;
    }
  };
  auto stage_store = [&](int b) {
    if (PREP) return;
#pragma unroll
    for (int h = 0; h < AB / 4; ++h) ring[b & 1][wv + 4 * h][lane] = sreg[h];
  };

  stage_load(0);
  stage_store(0);
  for (int b = 0; b < nbatch_all; ++b) {
    const int buf = b & 1;
    __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): this wave's share of batch b has landed
    __syncthreads();                                 // batch b complete; everyone is done with the other buffer
    if (G > 1 && b >= nbatch) continue;              // (a smaller part of the workgroup: only the barrier)
    if (b + 1 < nbatch) stage_load(b + 1);           // in flight while batch b is gathered
    const int nal = (na - b * AB < AB) ? na - b * AB : AB;
    // the batch's mode-0 angles come first (the angles are sorted by mode): two loops without a mode test inside, unrolled so
    // that the LDS reads of several angles are in flight together (small images run few waves per SIMD: latency, not issue)
    const int a0 = b * AB;
    const int nm0 = (n0 - a0 < 0) ? 0 : (n0 - a0 < nal ? n0 - a0 : nal);
    const unsigned rbase0 = __builtin_amdgcn_readfirstlane(lds_offset(&ring[buf][0][0]));
    // One pixel per thread (16 x 16 tiles): FOUR angles per trip, written so that their four {C, B32} reads and then their four
    // record reads are in flight together.  Angle by angle the compiler waited for each LDS read before the next (the requested
    // unrolling was not done): two exposed LDS round trips per angle and wave, which four waves per SIMD cannot cover — at
    // 512^2 x 180 the kernel was latency-bound at 36 us with 12 vector instructions per angle, 16 M in all (PMC).
    auto angles = [&](int al_lo, int al_hi, const float* fcol, const unsigned* ncol, int cbrow, f2v* an, float* ac) {
      int al = al_lo;
      if (PX == 1) {
        for (; al + 4 <= al_hi; al += 4) {
          AdjAngle p[4];
          uint2 cb[4];
          u4r rr[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) p[u] = ang[a0 + al + u];
#pragma unroll
          for (int u = 0; u < 4; ++u) cb[u] = cbs[buf][al + u][cbrow];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned bits = __builtin_bit_cast(unsigned, fmaf(fcol[0], p[u].rinv, __builtin_bit_cast(float, cb[u].x)) + RND_MAGIC);
            rr[u] = ring_read(rbase0 + (al + u) * 1024, bits);
          }
          ring_wait();
#pragma unroll
          for (int u = 0; u < 4; ++u) ring_tie(rr[u]);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            f2v cr = {p[u].c1m, p[u].c1p};           // the packed clamp-FMA takes its addend from this scalar pair
            asm("" : "+s"(cr));
            adj_gather(rr[u], cb[u].y, ncol[0], sc2, nsc, cr, an[0], ac[0]);
          }
        }
      }
      // (round 6) the NEXT angle's constants and {C, B32} pair are requested right behind this angle's record reads, so that one wait
      // covers both: angle by angle the loop had two exposed round trips (scalar + LDS for the pair, then LDS for the records)
#ifdef TRK_ADJ_EXPERIMENT_NO_PIPELINE                 // (A/B build switch: rounds 2-5's loop, two waits per angle)
#pragma unroll 2
      for (; al < al_hi; ++al) {
        const AdjAngle p = ang[a0 + al];
        f2v cr = {p.c1m, p.c1p};
        asm("" : "+s"(cr));
        const uint2 cb = cbs[buf][al][cbrow];
        const float C = __builtin_bit_cast(float, cb.x);
        u4r rr[PX];
#pragma unroll
        for (int k = 0; k < PX; ++k) {
          const unsigned bits = __builtin_bit_cast(unsigned, fmaf(fcol[k], p.rinv, C) + RND_MAGIC);
          rr[k] = ring_read(rbase0 + al * 1024, bits);
        }
        ring_wait();
#pragma unroll
        for (int k = 0; k < PX; ++k) {
          ring_tie(rr[k]);
          adj_gather(rr[k], cb.y, ncol[k], sc2, nsc, cr, an[k], ac[k]);
        }
      }
#else
      if (al < al_hi) {
        float pn_c1 = ang[a0 + al].c1m, pn_c1p = ang[a0 + al].c1p, pn_rinv = ang[a0 + al].rinv;     // wave-uniform: scalar loads
        u2r cbn = pair_read(&cbs[buf][al][cbrow]);
        ring_wait();
        pair_tie(cbn);
#pragma unroll 2
        for (; al < al_hi; ++al) {
          const float p_rinv = pn_rinv;
          f2v cr = {pn_c1, pn_c1p};
          asm("" : "+s"(cr));
          const u2r cbv = cbn;                       // (a copy made AFTER the tie)
          const float C = __builtin_bit_cast(float, (unsigned)cbv[0]);
          const unsigned cb_y = cbv[1];
          u4r rr[PX];
#pragma unroll
          for (int k = 0; k < PX; ++k) {
            const unsigned bits = __builtin_bit_cast(unsigned, fmaf(fcol[k], p_rinv, C) + RND_MAGIC);
            rr[k] = ring_read(rbase0 + al * 1024, bits);
          }
          const int aln = al + 1 < al_hi ? al + 1 : al;      // (the last angle re-reads itself: unconditional, no join)
          pn_c1 = ang[a0 + aln].c1m;
          pn_c1p = ang[a0 + aln].c1p;
          pn_rinv = ang[a0 + aln].rinv;
          cbn = pair_read(&cbs[buf][aln][cbrow]);
          ring_wait();
          pair_tie(cbn);
#pragma unroll
          for (int k = 0; k < PX; ++k) {
            ring_tie(rr[k]);
            adj_gather(rr[k], cb_y, ncol[k], sc2, nsc, cr, an[k], ac[k]);
          }
        }
      }
#endif
    };
    angles(0, nm0, fcolA, colA, r0, anA, accA);
    angles(nm0, nal, fcolB, colB, c1, anB, accB);
#ifndef TRK_ADJ_EXPERIMENT_NO_F64_TOTALS             // (A/B build switch: tools/r06_ab_libs.sh)
    if (((b + 1) * AB) % ADJ_FLUSH == 0) {           // wave-uniform; only the batch where the modes change flushes both
      if (nm0 > 0) {
#pragma unroll
        for (int k = 0; k < PX; ++k) {
          totA[k] += (double)(accA[k] + (anA[k][0] + anA[k][1]));
          accA[k] = 0.f;
          anA[k] = (f2v){0.f, 0.f};
        }
      }
      if (nm0 < nal) {
#pragma unroll
        for (int k = 0; k < PX; ++k) {
          totB[k] += (double)(accB[k] + (anB[k][0] + anB[k][1]));
          accB[k] = 0.f;
          anB[k] = (f2v){0.f, 0.f};
        }
      }
    }
#endif
    if (b + 1 < nbatch) stage_store(b + 1);          // the other buffer: nobody reads it before the next barrier
  }
  // the two partial images meet: mode-0 sums go through LDS to the thread that holds the pixel in the mode-1 layout
  if (PX > 1) {
#pragma unroll
    for (int k = 0; k < PX; ++k) xch[r0][c0 + k * TS] = totA[k] + (double)(accA[k] + (anA[k][0] + anA[k][1]));
    __syncthreads();
  }
  float oraw[PX];
#pragma unroll
  for (int k = 0; k < PX; ++k)
    oraw[k] = (float)((totB[k] + (double)(accB[k] + (anB[k][0] + anB[k][1]))) +
                      (PX > 1 ? xch[r1 + k * TS][c1] : totA[k] + (double)(accA[k] + (anA[k][0] + anA[k][1]))));
  if (G > 1) {
    // the groups' partial tiles meet in LDS, added in part order by the first group, which carries the epilogue alone
    __shared__ float red[G > 1 ? G - 1 : 1][256][PX];
    if (grp > 0) {
#pragma unroll
      for (int k = 0; k < PX; ++k) red[grp - 1][tid][k] = oraw[k];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int k = 0; k < PX; ++k) {
      float t = oraw[k];
#pragma unroll
      for (int s = 1; s < G; ++s) t += red[s - 1][tid][k];
      oraw[k] = t;
    }
  } else if (PX == 4 && nsplit > 1) {                              // (the host splits only the 32 x 32 form)
    // The parts of a tile meet: every workgroup leaves its partial tile (thread-major, 16 bytes per lane), then takes a ticket; the
    // one that draws the LAST ticket adds the partial tiles in part order (the same bits whoever comes last) and carries the
    // epilogue.  The parts may have run on different XCDs, whose L2s are not coherent: the bytes are stored WRITE-THROUGH (sc1) and
    // loaded past the L1 (sc1), every storing wave drains its stores before the workgroup's one ticket (an agent-scope atomic add),
    // and the loads are issued only after the add has returned and the workgroup has met (MI355X_MICROARCH.md, inter-workgroup
    // visibility: the "workgroup whose add came last" hand-off).  A release / acquire fence pair per workgroup instead
    // (__threadfence) writes back and invalidates whole caches: 512^2 x 180 ran 130 us instead of 33.
    constexpr int MAXSPLIT = 8;
    const int64_t pstride = (int64_t)gridDim.y * ntiles * (T * T);
    float* P = part_img + ((int64_t)frame * ntiles + tile_id) * (T * T) + tid * PX;     // (not __restrict__: the other parts write it too)
    {
      f4r v = {oraw[0], oraw[PX > 1 ? 1 : 0], oraw[PX > 2 ? 2 : 0], oraw[PX > 3 ? 3 : 0]};
      asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(P + part * pstride), "v"(v) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned ticket;
    unsigned* cnt = tile_cnt + (int64_t)frame * ntiles + tile_id;
    if (tid == 0) ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != (unsigned)(nsplit - 1)) return;
    if (tid == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // for the next launch
    f4r pv[MAXSPLIT];
#pragma unroll
    for (int s = 0; s < MAXSPLIT; ++s) {
      // straight-line code (parts beyond nsplit re-read part 0 and are not added): a branch around an inline-assembly load would
      // let the compiler copy its destination at the join, before the data has arrived
      const float* src = P + (s < nsplit ? s : 0) * pstride;
      asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(pv[s]) : "v"(src) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < MAXSPLIT; ++s) asm volatile("" : "+v"(pv[s]));
#pragma unroll
    for (int k = 0; k < PX; ++k) {
      float t = pv[0][k];
#pragma unroll
      for (int s = 1; s < MAXSPLIT; ++s) t = s < nsplit ? t + pv[s][k] : t;
      oraw[k] = t;
    }
  }
  const bool lead = tile_id == 0 && blockIdx.y == 0;               // the workgroup that carries the once-per-launch duties
  double q = 0.0;
  img += (int64_t)frame * N * N;
  // epilogue (trk_op_apply_axpby): out = a * (A^T s) + b * z; xT_out: also the transposed image the next forward apply wants
  // (measured at 512^2: fetching z and the coefficients before the angle loop instead costs 1.6 us — registers)
  float zv[PX];
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    const int i = i0 + r1 + k * TS, j = j0 + c1;
    zv[k] = (epi.on && epi.z && i < N && j < N) ? epi.z[((int64_t)frame * N + i) * N + j] : 0.f;
  }
  // (the operands of the rider below: requested here, so that they travel while the coefficients are worked out)
  float lw[PX], lx[PX], lr[PX];
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    const int i = i0 + r1 + k * TS, j = j0 + c1;
    const bool in = epi.lq.on && i < N && j < N;
    const int64_t g = ((int64_t)frame * N + i) * N + j;
    lw[k] = (in && !epi.lq.first) ? epi.lq.w[g] : 0.f;
    lx[k] = (in && epi.lq.x_in) ? epi.lq.x_in[g] : 0.f;
    lr[k] = (in && epi.lq.ref) ? epi.lq.ref[g] : 0.f;
  }
  float ca, cb;
  double pend_sum = 0.0, cad, cbd;
  epi_coefs(epi, lead, &lds[0], ca, cb, &pend_sum, &cad, &cbd);
  if (xT_out) xT_out += (int64_t)frame * N * N;
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    const int i = i0 + r1 + k * TS, j = j0 + c1;
    float o = oraw[k];
    if (i < N && j < N) {
      if (epi.on) o = epi_combine(epi.on, ca, cb, cad, cbd, o, zv[k], epi.z != nullptr);
      img[(int64_t)i * N + j] = o;
      if (xT_out) xT_out[(int64_t)j * N + i] = o;
      q += (double)o * o;
    }
  }
  if (ssq_part) {                                                 // uniform over the grid
    q = block_sum<256>(q, lds);
    if (tid == 0) ssq_part[(size_t)blockIdx.y * ntiles + tile_id] = q;
  }
  if (epi.pq.on && lead && tid < 64) {
    // the mailbox post of the step before (k_mailbox_post / k_mailbox_post_sum, core.hip): its scalars are final here — the
    // deferred one is `pend_sum`, which this workgroup has just stored — and the host polls the sequence word
    const PostReq& Q = epi.pq;
    double sum = 0.0;
    if (Q.part) sum = scalar_from_wave(ScalarSrc{Q.part, Q.n_part}, tid);
    if (tid == 0) {
      for (int c = 0; c < Q.count; ++c) {
        const double* sp = Q.src + c;
        Q.dst[c] = (epi.pend_target && sp == epi.pend_target) ? pend_sum : *sp;
      }
      if (Q.part) {
        *Q.sum_dev = sum;
        *Q.sum_host = sum;
      }
      __threadfence_system();
      __hip_atomic_store(Q.seq, Q.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (epi.lq.on) {                                                // uniform over the grid
    // the damped-LSQR step of the iterate that z = V[k-1] belongs to, on this workgroup's pixels: k_lsqr_damped_update's
    // arithmetic, expression for expression (vecops.hip) — the same floats whichever kernel forms them
    const LsqrReq& L = epi.lq;
    __shared__ double lcf[3];
    __syncthreads();
    if (tid == 0) {
      const double b2v = (epi.pend_target && epi.pend_target == L.b2) ? pend_sum : *L.b2;
      const double alpha = sqrt(*L.a2), beta = sqrt(b2v);
      double rhobar, phibar, tw = 0.0;
      if (L.first) {
        rhobar = alpha;
        phibar = sqrt(*L.beta0_sq);
      } else {
        rhobar = -L.st_in[0] * alpha;
        tw = L.st_in[1] * alpha / L.st_in[2];
        phibar = L.st_in[3];
      }
      const double rhobar1 = sqrt(rhobar * rhobar + L.damp * L.damp);
      phibar *= rhobar / rhobar1;
      const double rho = sqrt(rhobar1 * rhobar1 + beta * beta);
      const double cs = rhobar1 / rho, sn = beta / rho;
      lcf[0] = 1.0 / alpha;
      lcf[1] = tw;
      lcf[2] = cs * phibar / rho;
      if (lead) {
        L.st_out[0] = cs;
        L.st_out[1] = sn;
        L.st_out[2] = rho;
        L.st_out[3] = sn * phibar;
      }
    }
    __syncthreads();
    const double ia = lcf[0], tw = lcf[1], px = lcf[2];
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < PX; ++k) {
      const int i = i0 + r1 + k * TS, j = j0 + c1;
      if (i < N && j < N) {
        const int64_t g = ((int64_t)frame * N + i) * N + j;
        const float wo = lw[k], xo = lx[k];
        const float wn = (float)(ia * (double)zv[k] - (L.first ? 0.0 : tw * (double)wo));
        const float xn = (float)((L.x_in ? (double)xo : 0.0) + px * (double)wn);
        L.w[g] = wn;
        L.x_out[g] = xn;
        if (L.ref) {
          const double e = (double)xn - lr[k];
          acc += e * e;
        }
      }
    }
    if (L.ref) {
      acc = block_sum<256>(acc, lds);
      if (tid == 0) L.err_part[(size_t)blockIdx.y * ntiles + tile_id] = acc;
    }
  }
}

// ---------------------------------------------------------------------------------------- adjoint by mirrored tile pairs (round 6)
// The symmetry that carries the forward's quads (k_radon_fwd_quad) applied to the adjoint.  With the BASE geometry of a quad — beta in
// [0, 45 deg], q_b(d, tt) = (A_b[d] + B_b[tt]) 2^-24 — one evaluation of {nearest base detector d0, t0 = q_b(d0, tt) - col, the three hat
// weights} at (tt, col) serves
//     slot 0 (rows of x):            pixel (tt, col)            slot 1 (rows of x, mirrored):  pixel (tt, N-1-col)
//     slot 2 (rows of x^T):          pixel (col, tt)            slot 3 (rows of x^T, mirrored): pixel (N-1-col, tt)
// each with ITS member's sinogram values at the base detectors d0 - 1, d0, d0 + 1 (a mirrored member sees t = -t0: the hat is even; a
// flipped member's detector index runs the other way: the record array recq is written per (quad, slot, BASE detector) by
// k_radon_adj_prepq, so that one ring slot index serves all slots).  A pixel's four members need four different geometries, so the
// sharing is between MIRRORED PIXELS: (tt, col) and (tt, N-1-col) exchange slots 0 / 1, (col, tt) and (N-1-col, tt) slots 2 / 3.  A
// workgroup therefore owns the orbit of a 32 x 32 tile under the two mirrors — tiles (a, b), (a, b~), (a~, b), (a~, b~) — and runs four
// sub-phases over all quads: rows of a / rows of a~ (slots 0 and 1, the column-mirrored pair of tiles each), columns of b / columns
// of b~ (slots 2 and 3, the row-mirrored pair each).  Per geometry: 8 shared vector instructions + 2 per member (k_radon_adj_tile: 10.25
// per pixel and angle; here 6), one ds_read_b128 per pixel and angle as before.  The four sums of a pixel (two sub-phases, two members
// each) meet in LDS; the tiles leave through one coalesced pass that carries the epilogue (a * A^T s + b * z, the norm partials,
// the transposed copy for the next forward apply).  Same taps and weights as k_radon_adj_tile, another summation order: tested
// against the float64 oracle and against that kernel.
__global__ __launch_bounds__(256) void k_radon_adj_prepq(const float* __restrict__ sino, uint4* __restrict__ recq, int nd, int na, int nq,
                                                         const QuadParam* __restrict__ quads, const float* __restrict__ wq,
                                                         const unsigned* __restrict__ A32q) {
  const int ndp = nd + 2 * A32_PAD;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // over (quad of the frame, slot) x ndp; blockIdx.y = frame
  const int64_t r = idx / ndp;
  if (r >= (int64_t)nq * 4) return;
  const int e = (int)(idx - r * ndp), q = (int)(r >> 2), m = (int)(r & 3);
  const int64_t qr = (int64_t)blockIdx.y * nq + q;
  const QuadParam p = quads[qr];
  const int am = m == 0 ? p.am[0] : (m == 1 ? p.am[1] : (m == 2 ? p.am[2] : p.am[3]));
  const bool flip = ((p.flip >> m) & 1) != 0;
  const float w = wq[qr * 4 + m];
  const int d = e - A32_PAD;
  const float* __restrict__ S = sino + ((int64_t)blockIdx.y * na + (am < 0 ? 0 : am)) * nd;
  auto val = [&](int db) -> float {                                 // the member's sample at BASE detector db
    const int dm = flip ? nd - 1 - db : db;
    return (am >= 0 && db >= 0 && db < nd) ? w * S[dm] : 0.f;
  };
  uint4 o;
  o.x = __builtin_bit_cast(unsigned, val(d - 1));                   // the base ray at t0 - inv (inv > 0)
  o.y = __builtin_bit_cast(unsigned, val(d + 1));                   // the base ray at t0 + inv
  o.z = __builtin_bit_cast(unsigned, val(d));
  o.w = A32q[qr * ndp + e];
  recq[(qr * 4 + m) * ndp + e] = o;
}

template <int QB>
__global__ __launch_bounds__(256, 3) void k_radon_adj_quad(const uint4* __restrict__ recq, float* __restrict__ img, int N, int nd, int nq,
                                                        const AdjQuad* __restrict__ aq, const uint2* __restrict__ CBq, int npad,
                                                        int tiles_h, double* __restrict__ ssq_part, Epi epi,
                                                        float* __restrict__ xT_out) {
  constexpr int T = 32, PX = 4, TS = 8;
  __shared__ __attribute__((aligned(16))) uint4 ring[2][QB][4][64];   // [buffer][quad of the batch][base tile 0: slot A, slot B; base tile 1: A, B]
  __shared__ __attribute__((aligned(16))) uint2 cbs[2][QB][T];        // {C, B32} of the base at the sub-phase's 32 marching indices
  __shared__ float sum[4][T][T + 1];                                   // the orbit's four tiles: [2 (row >= N/2) + (col >= N/2)]
  __shared__ double lds[4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frame = blockIdx.y;
  const int ta = blockIdx.x / tiles_h, tb = blockIdx.x - ta * tiles_h;
  const int i0 = ta * T, j0 = tb * T;                                  // tile (a, b); its mirrors start at N - T - i0 / N - T - j0
  const int ndp = nd + 2 * A32_PAD;
  aq += (int64_t)frame * nq;
  CBq += (int64_t)frame * nq * npad;
  recq += (int64_t)frame * nq * 4 * ndp;
#ifdef TRK_ADJQ_EXPERIMENT_SPLIT
  {  // TIMING EXPERIMENT ONLY (wrong results: the splits overwrite each other): blockIdx.z takes a share of the quads
    const int per = (nq + (int)gridDim.z - 1) / (int)gridDim.z, qs = blockIdx.z * per;
    aq += qs; CBq += (int64_t)qs * npad; recq += (int64_t)qs * 4 * ndp;
    nq = nq - qs < per ? (nq - qs < 0 ? 0 : nq - qs) : per;
  }
#endif
  const auto rrec = __builtin_amdgcn_make_buffer_rsrc((void*)recq, 0, (unsigned)((int64_t)nq * 4 * ndp * 16), 0x00020000);
  const auto rcb = __builtin_amdgcn_make_buffer_rsrc((void*)CBq, 0, (unsigned)((int64_t)nq * npad * 8), 0x00020000);
  const float sdh = 0.5f * (float)(nd - 1);
  f2v sc2 = {5.9604644775390625e-8f, 5.9604644775390625e-8f};          // 2^-24, kept in an (aligned) VGPR pair
  float nsc = -5.9604644775390625e-8f;
  asm("" : "+v"(sc2));
  asm("" : "+s"(nsc));
  const int nbatch = (nq + QB - 1) / QB;
  const int r0 = tid / TS, c0 = tid % TS;                              // sub-phases 0, 1: row r0, columns c0 + 8 k
  int nqm1;
  asm("s_add_i32 %0, %1, -1" : "=s"(nqm1) : "s"(nq) : "scc");

#pragma unroll 1
  for (int sp = 0; sp < 4; ++sp) {
    // the sub-phase's geometry: marching index tt (fixed per thread), interpolated coordinates col_k and their mirrors N-1-col_k
    const bool colmode = sp >= 2;
    const int u0 = colmode ? j0 : i0, v0 = colmode ? i0 : j0;          // marching tile start / interpolated tile start (unmirrored)
    const int tl = colmode ? tid / TS : r0;                            // marching index within the tile
    const int cl = colmode ? tid % TS : c0;                            // first interpolated index within the tile
    const bool mir_t = (sp & 1) != 0;                                  // sub-phases 1, 3: the mirrored marching tile
    const int tt0 = mir_t ? N - T - u0 : u0;                           // its first marching index (ascending table order)
    const int tt = mir_t ? N - 1 - (u0 + tl) : u0 + tl;
    const int ttl = tt - tt0;
    const int slotA = colmode ? 2 : 0;
    float fcol[2][PX];
    unsigned ncol[2][PX];
#pragma unroll
    for (int k = 0; k < PX; ++k) {
      const int c = v0 + cl + k * TS;
      fcol[0][k] = (float)c;
      ncol[0][k] = 0u - ((unsigned)c << QF);
      fcol[1][k] = (float)(N - 1 - c);
      ncol[1][k] = 0u - ((unsigned)(N - 1 - c) << QF);
    }
    // centres of the two base tiles (the interpolated tile and its mirror) for the ring bases
    const float tt_c = (float)tt0 + 0.5f * (float)(T - 1);
    const float co_c0 = (float)v0 + 0.5f * (float)(T - 1), co_c1 = (float)(N - T - v0) + 0.5f * (float)(T - 1);
    f2v an[2][PX];
    float ac[2][PX];
#pragma unroll
    for (int k = 0; k < PX; ++k) {
      an[0][k] = an[1][k] = (f2v){0.f, 0.f};
      ac[0][k] = ac[1][k] = 0.f;
    }
    // staging of batch b into buffer b & 1: wave w takes rings w, w + 4, ... of the batch's 4 QB (ring = quad * 4 + 2 * base tile + slot
    // B), lane l the base detector whose ring slot is l; threads 0 .. 16 QB - 1 the {C, B32} pairs (16 bytes = two indices each)
    float nx_rinv, nx_dq, nx_k0;
    auto fetch_quads = [&](int b) {
      int q = b * QB + (lane & (QB - 1));
      q = q < nq ? q : nq - 1;
      nx_rinv = aq[q].rinv;
      nx_dq = aq[q].dq;
      nx_k0 = aq[q].k0;
    };
    fetch_quads(0);
    auto stage_load = [&](int b) {
      // lane (q, base tile) = (lane & (QB-1), (lane / QB) & 1) works out one ring base; handed out by v_readlane (no scalar float unit)
      int dbase_l;
      {
        const float co_c = ((lane / QB) & 1) ? co_c1 : co_c0;
        dbase_l = (int)floorf(fmaf(co_c - fmaf(tt_c, nx_dq, nx_k0), nx_rinv, sdh)) - 32;
      }
      if (b + 1 < nbatch) fetch_quads(b + 1);
#pragma unroll
      for (int h = 0; h < QB; ++h) {
        const int rr = wv + 4 * h;                                     // ring of the batch: quad rr / 4, base tile (rr / 2) & 1, slot A + (rr & 1)
        const int ql = rr >> 2, bt = (rr >> 1) & 1;
        int q;
        asm("s_min_i32 %0, %1, %2" : "=s"(q) : "s"(b * QB + ql), "s"(nqm1) : "scc");
        const int row = (q * 4 + slotA + (rr & 1)) * ndp;
        const int dbase = __builtin_amdgcn_readlane(dbase_l, ql + QB * bt);
        const int d = dbase + ((lane - dbase) & 63);
        int e;
        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(e) : "v"(d + A32_PAD), "s"(ndp - 1));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rrec, (__attribute__((address_space(3))) void*)&ring[b & 1][ql][rr & 3][0], 16, (row + e) * 16, 0, 0, 0);
      }
      if (tid < QB * T / 2) {
        const int ql = tid / (T / 2), pr = tid - ql * (T / 2);
        int q = b * QB + ql;
        q = q < nq ? q : nq - 1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rcb, (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(&cbs[b & 1][0][0]) + wv * 1024), 16,
                                                 (q * npad + tt0 + 2 * pr) * 8, 0, 0, 0);
      }
    };
    stage_load(0);
    for (int b = 0; b < nbatch; ++b) {
      const int buf = b & 1;
      __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): this wave's share of batch b has landed
      __syncthreads();                               // batch b complete; everyone is done with the other buffer
      if (b + 1 < nbatch) stage_load(b + 1);
      const int nql = (nq - b * QB < QB) ? nq - b * QB : QB;
      const unsigned rbase = __builtin_amdgcn_readfirstlane(lds_offset(&ring[buf][0][0][0]));
      // Software pipeline over the batch's 2 QB half-quads (round 6, second pass): the eight record reads of half-quad i + 1 are in
      // flight while half-quad i is weighed — LDS returns in order, so "at most nine younger reads outstanding" means half-quad i has
      // landed.  No scalar load may be outstanding inside (they return out of order: any wait would have to be lgkmcnt(0)), so the
      // batch's constants are fetched up front; the {C, B32} pair of the next quad rides between the two halves' reads.
      float q_c1m[QB], q_c1p[QB], q_rinv[QB];
#pragma unroll
      for (int ql = 0; ql < QB; ++ql) {
        const int qi = b * QB + (ql < nql ? ql : 0);          // wave-uniform: scalar loads
        q_c1m[ql] = aq[qi].c1m;
        q_c1p[ql] = aq[qi].c1p;
        q_rinv[ql] = aq[qi].rinv;
      }
      u2r cbq = pair_read(&cbs[buf][0][ttl]);
      ring_wait();
      pair_tie(cbq);
      u4r ra[2][PX], rb[2][PX];
      auto issue = [&](int ql, int h, float rinv_q, float C) {
        const unsigned rb_h = rbase + (unsigned)(ql * 4 + 2 * h) * 1024u;
#pragma unroll
        for (int k = 0; k < PX; ++k) {
          const unsigned bits = __builtin_bit_cast(unsigned, fmaf(fcol[h][k], rinv_q, C) + RND_MAGIC);
          unsigned addr;
          const unsigned slot = bits & 63u;
          asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(addr) : "v"(slot), "s"(rb_h));
          asm volatile("ds_read_b128 %0, %1" : "=v"(ra[h][k]) : "v"(addr) : "memory");
          asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(rb[h][k]) : "v"(addr) : "memory");
        }
      };
      auto weigh = [&](int h, unsigned cb_y, f2v cr) {
#pragma unroll
        for (int k = 0; k < PX; ++k) {
          ring_tie(ra[h][k]);
          ring_tie(rb[h][k]);
          // slot A's member sees the pixel of this geometry, slot B's its mirror: tile set h / 1 - h
          const unsigned slo = ra[h][k][0], shi = ra[h][k][1], s0 = ra[h][k][2], a32 = ra[h][k][3];
          const unsigned mlo = rb[h][k][0], mhi = rb[h][k][1], m0 = rb[h][k][2];
          unsigned ti;
          asm("v_add3_u32 %0, %1, %2, %3" : "=v"(ti) : "v"(a32), "v"(cb_y), "v"(ncol[h][k]));
          const float tf = (float)(int)ti;
          f2v t2;
          t2[0] = tf;
          f2v wn;
          asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,0,1] neg_hi:[0,1,0] clamp" : "=v"(wn) : "v"(t2), "v"(sc2), "s"(cr));
          float w0;
          asm("v_fma_f32 %0, |%1|, %2, 1.0" : "=v"(w0) : "v"(tf), "s"(nsc));
          const f2v sn = {__builtin_bit_cast(float, slo), __builtin_bit_cast(float, shi)};
          const f2v mn = {__builtin_bit_cast(float, mlo), __builtin_bit_cast(float, mhi)};
          an[h][k] = __builtin_elementwise_fma(wn, sn, an[h][k]);
          ac[h][k] = fmaf(w0, __builtin_bit_cast(float, s0), ac[h][k]);
          an[1 - h][k] = __builtin_elementwise_fma(wn, mn, an[1 - h][k]);
          ac[1 - h][k] = fmaf(w0, __builtin_bit_cast(float, m0), ac[1 - h][k]);
        }
      };
      issue(0, 0, q_rinv[0], __builtin_bit_cast(float, (unsigned)cbq[0]));
#pragma unroll
      for (int ql = 0; ql < QB; ++ql) {
        if (ql < nql) {                                      // wave-uniform
          const u2r cbv = cbq;                               // (a copy of a pair that has landed and been tied)
          const float C = __builtin_bit_cast(float, (unsigned)cbv[0]);
          const unsigned cb_y = cbv[1];
          f2v cr = {q_c1m[ql], q_c1p[ql]};
          asm("" : "+s"(cr));
          issue(ql, 1, q_rinv[ql], C);                       // 8 more reads ...
          const int qln = ql + 1 < QB ? ql + 1 : ql;         // (the last quad of a batch re-reads its own pair: unconditional, no join)
          cbq = pair_read(&cbs[buf][qln][ttl]);              // ... and the next quad's pair behind them
          asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");  // half 0 of this quad has landed
          weigh(0, cb_y, cr);
          ring_wait();                                       // half 1 and the pair have landed (they had half 0's arithmetic to do so)
          pair_tie(cbq);
          if (ql + 1 < QB) {                                 // (compile-time; the reads themselves are unconditional — a quad beyond the batch's
            const bool more = ql + 1 < nql;                  //  last re-reads this one's rings: no join behind an asynchronous read)
            issue(more ? ql + 1 : ql, 0, more ? q_rinv[ql + 1 < QB ? ql + 1 : ql] : q_rinv[ql], __builtin_bit_cast(float, (unsigned)cbq[0]));
          }
          weigh(1, cb_y, cr);
        }
      }
    }
    // the sub-phase's sums meet the other one's in LDS: pixel (row, col) of tile set h, interpolated index col_k (h = 0) or its mirror
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int k = 0; k < PX; ++k) {
        const int c = h ? N - 1 - (v0 + cl + k * TS) : v0 + cl + k * TS;
        const int row = colmode ? c : tt, col = colmode ? tt : c;
        float* dst = &sum[2 * (row >= N / 2 ? 1 : 0) + (col >= N / 2 ? 1 : 0)][row & (T - 1)][col & (T - 1)];
        const float v = ac[h][k] + (an[h][k][0] + an[h][k][1]);
        *dst = colmode ? *dst + v : v;
      }
    __syncthreads();
  }

  // ---- the four tiles leave: rows of 32 contiguous pixels per quarter-wave, the epilogue of trk_op_apply_axpby on the way
  const bool lead = blockIdx.x == 0 && blockIdx.y == 0;
  float ca, cb;
  double pend_sum = 0.0, cad, cbd;
  img += (int64_t)frame * N * N;
  const int pc = tid & 31, pr = tid >> 5;
  float zv[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = ((t >> 1) ? N - T - i0 : i0) + pr + 8 * k, j = ((t & 1) ? N - T - j0 : j0) + pc;
      zv[t][k] = (epi.on && epi.z) ? epi.z[((int64_t)frame * N + i) * N + j] : 0.f;
    }
  epi_coefs(epi, lead, &lds[0], ca, cb, &pend_sum, &cad, &cbd);
  double q = 0.0;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = ((t >> 1) ? N - T - i0 : i0) + pr + 8 * k, j = ((t & 1) ? N - T - j0 : j0) + pc;
      float o = sum[t][pr + 8 * k][pc];
      if (epi.on) o = epi_combine(epi.on, ca, cb, cad, cbd, o, zv[t][k], epi.z != nullptr);
      img[(int64_t)i * N + j] = o;
      if (xT_out) sum[t][pr + 8 * k][pc] = o;
      q += (double)o * o;
    }
  if (xT_out) {
    xT_out += (int64_t)frame * N * N;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = ((t >> 1) ? N - T - i0 : i0) + pc, j = ((t & 1) ? N - T - j0 : j0) + pr + 8 * k;
        xT_out[(int64_t)j * N + i] = sum[t][pc][pr + 8 * k];
      }
  }
  if (ssq_part) {
    q = block_sum<256>(q, lds);
    if (tid == 0) ssq_part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = q;
  }
}

// The same arithmetic without LDS (one thread per pixel, records and table pairs read from memory): the reference form the
// tiled kernel is tested against (TRK_RADON_ADJ_SIMPLE=1 selects it) and the path for frames too small to tile.
__global__ __launch_bounds__(256) void k_radon_adj_simple(const uint4* __restrict__ rec, float* __restrict__ img, int N, int nd, int na,
                                                          const AdjAngle* __restrict__ ang, const int* __restrict__ n_mode0,
                                                          const uint2* __restrict__ CB, int npad, double* __restrict__ ssq_part) {
  __shared__ double lds[4];
  const int64_t idx_raw = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool inside = idx_raw < (int64_t)N * N;
  const int64_t idx = inside ? idx_raw : (int64_t)N * N - 1;
  const int i = (int)(idx / N), j = (int)(idx - (int64_t)i * N);
  const int frame = blockIdx.y;
  const int ndp = nd + 2 * A32_PAD;
  ang += (int64_t)frame * na;
  rec += (int64_t)frame * na * ndp;
  CB += (int64_t)frame * na * npad;
  const int n0 = n_mode0[frame];
  float acc0 = 0.f;
  f2v accn = {0.f, 0.f};
  f2v sc2 = {5.9604644775390625e-8f, 5.9604644775390625e-8f};
  float nsc = -5.9604644775390625e-8f;
  asm("" : "+v"(sc2));
  asm("" : "+s"(nsc));
  for (int a = 0; a < na; ++a) {
    const AdjAngle p = ang[a];
    const int tt = a < n0 ? i : j, col = a < n0 ? j : i;
    const uint2 cb = CB[(int64_t)p.orig * npad + tt];
    int d0 = (int)rintf(fmaf((float)col, p.rinv, __builtin_bit_cast(float, cb.x)));
    int e = d0 + A32_PAD;
    e = e < 0 ? 0 : (e > ndp - 1 ? ndp - 1 : e);
    const uint4 rr = rec[(int64_t)a * ndp + e];
    f2v cr = {p.c1m, p.c1p};
    asm("" : "+s"(cr));
    adj_gather((u4r){rr.x, rr.y, rr.z, rr.w}, cb.y, 0u - ((unsigned)col << QF), sc2, nsc, cr, accn, acc0);
  }
  const float acc = acc0 + (accn[0] + accn[1]);
  if (inside) img[(int64_t)frame * N * N + idx] = acc;
  if (ssq_part) {
    const double q = block_sum<256>(inside ? (double)acc * acc : 0.0, lds);
    if (threadIdx.x == 0) ssq_part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = q;
  }
}

// Which forward kernel an input at `xb` gets: per-wave LDS windows (N % 4 == 0, 16-byte aligned input; DMA = staged by direct-to-LDS
// loads), the window-sharing kernel from 1024^2 on, else direct gathers.  direct1: the per-wave-window kernel reads the angles
// marched along columns from the image itself, transposing while it stages — no transposed copy, no launch for it (512^2 x 180:
// the copy was 5 of the apply's 31 us; 32 frames of 256^2: 5.8 of 23).  The window-sharing kernel keeps the copy (2.6 % at 4096^2).
struct FwdPath {
  bool lds, dma, win, direct1, quad;
};
FwdPath fwd_path(const RadonImpl* im, const float* xb) {
  static const bool no_lds = getenv("TRK_RADON_NO_LDS") != nullptr;
  // global -> LDS directly (buffer_load_dwordx4 ... lds, new on gfx950) instead of through registers: 1.30 -> 1.11 ms at 4096^2
  static const bool dma = getenv("TRK_RADON_NO_DMA") == nullptr;
  static const bool no_win = getenv("TRK_RADON_NO_WIN") != nullptr;
  static const bool no_direct1 = getenv("TRK_RADON_NO_DIRECT1") != nullptr;
  FwdPath f;
  f.lds = !no_lds && (im->N % 4 == 0) && ((reinterpret_cast<uintptr_t>(xb) & 15u) == 0);
  f.dma = dma;
  // measured: 512^2 35 us (shared) vs 32 us (per-wave windows); 2048^2 0.256 vs 0.277 ms; 4096^2 0.96 vs 1.11 ms
  static const int win_min = getenv("TRK_RADON_WIN_MIN") ? atoi(getenv("TRK_RADON_WIN_MIN")) : 1024;     // tuning knob
  f.win = im->n_bands > 1 && im->N >= win_min && f.lds && dma && !no_win && im->band <= WIN_R * WIN_MAXCH && !im->band_res;
  f.direct1 = f.lds && !f.win && !no_direct1;
  // four symmetric angles per wave, conflict-free half-wave windows (k_radon_fwd_quad): 4096^2 x 180 0.94 -> see DESIGN.md 4.4
  static const bool no_quad = getenv("TRK_RADON_NO_QUAD") != nullptr;
  f.quad = f.win && !no_quad && im->nq > 0 && im->band <= QD_R * QD_MAXCH;
  return f;
}

constexpr int HINT_OUT_FEEDS_OPPOSITE = 1, HINT_INPUT_FROM_OPPOSITE = 2, HINT_SUMSQ_DEFERRED = 4;   // = TRK_HINT_* (trk.h)

int radon_flush(trk_op* op, hipStream_t s) {
  auto* im = static_cast<RadonImpl*>(op->impl);
  if (!im->pend_target) return TRK_OK;
  double* target = im->pend_target;
  im->pend_target = nullptr;
  return finalize_sums(im->pend_part, im->pend_n, 1, 1, target, s);
}

// Plain apply (epi.on = 0, hints = 0) and the fused form of trk_op_apply_axpby (batch 1) share this.
// ext_part / ext_cap / ext_n (trk_op_apply_fused with x2 = NULL): the fused norm is LEFT as block partials in the caller's buffer
// (*ext_n of them; one finished value if they do not fit) for a consumer kernel to add up — no finalize launch.
int radon_run(trk_op* op, int tr, const float* x, int64_t ldx, float* y, int64_t ldy, int batch, double* sumsq, Epi epi,
              int hints, hipStream_t s, double* ext_part = nullptr, int ext_cap = 0, int* ext_n = nullptr) {
  auto* im = static_cast<RadonImpl*>(op->impl);
  const int N = im->N, nd = im->nd, na = im->na, nt = im->nt;
  TimerScope tm(op->timer, op->timer_which, tr, s);
  // a norm the previous fused apply left unfinished: this apply's epilogue kernel finishes it if the caller chained the two
  // (it reads the partials for its coefficients), anything else gets it finished first
  epi.pend_target = nullptr;
  epi.pend_part = nullptr;
  epi.pend_n = 0;
  epi.dotv = nullptr;
  epi.dot_part = nullptr;
  if (im->pend_target) {
    if (epi.on && (hints & HINT_INPUT_FROM_OPPOSITE)) {
      epi.pend_target = im->pend_target;
      epi.pend_part = im->pend_part;
      epi.pend_n = im->pend_n;
      im->pend_target = nullptr;
    } else if (int rc = radon_flush(op, s)) {
      return rc;
    }
  }
  // fused ||y||^2 (batch 1): block partials from the kernel that writes y (band reduction / gather), then one finalize —
  // or none, when the caller lets the next chained apply finish it (TRK_HINT_SUMSQ_DEFERRED)
  double* ssq_part = nullptr;
  const bool adj_simple = getenv("TRK_RADON_ADJ_SIMPLE") != nullptr;   // read per call: tests switch it
  const bool tile = !adj_simple && N >= 16;
  // 32 x 32 tiles with 4 pixels per thread (fewest instructions per pixel) need enough tiles to fill the chip (frames of a
  // dynamic problem count); below that 16 x 16 tiles with one pixel per thread (4 x the waves)
  static const int tile_env = getenv("TRK_RADON_ADJ_TILE") ? atoi(getenv("TRK_RADON_ADJ_TILE")) : 0;
  // measured: 512^2 x 180 (one frame: 256 / 1024 tiles) 51 vs 42 us; 32 frames x 256^2 x 15 (2048 / 8192 tiles) 21 vs 34 us
  const int64_t tiles32 = (int64_t)ceil_div(N, 32) * ceil_div(N, 32) * nt;
  // Too few 32 x 32 tiles to fill the chip, many angles: the angles of a tile are SPLIT over nsplit workgroups whose partial tiles
  // meet in the one that finishes last (k_radon_adj_tile) — the instructions per pixel and angle of the 32 x 32 form (10.25 against
  // 14.5 for 16 x 16 tiles with one pixel per thread, where the staging of a batch is shared by a quarter of the pixels) at the
  // same number of waves.  TRK_RADON_ADJ_SPLIT: 1 = off, 2 / 4 / 8 forces.
  static const int split_env = getenv("TRK_RADON_ADJ_SPLIT") ? atoi(getenv("TRK_RADON_ADJ_SPLIT")) : 0;
  int nsplit = 1;
  // measured (us per apply, 180 angles; 16 x 16 tiles -> split 2 / 4 / 8): 256^2 24.1 -> 32.6 / 21.4 / 16.3, 512^2 32.9 -> 36.5 / 29.8 /
  // 29.8, 768^2 63.2 -> 54.8 / 50.4 / 50.9; 1024^2 (1024 tiles: 32 x 32 unsplit already) 83 with or without
  if (tile && !tile_env && batch == 1 && tiles32 < (split_env > 1 ? 4096 : 1024) && na > 32 && N >= 64) {
    nsplit = split_env ? split_env : (tiles32 <= 128 ? 8 : 4);
    nsplit = nsplit >= 8 ? 8 : (nsplit >= 4 ? 4 : (nsplit >= 2 ? 2 : 1));
  }
  const int tile_T = tile_env ? tile_env : ((tiles32 >= 1024 || nsplit > 1) ? 32 : 16);
  static const int ab_env = getenv("TRK_RADON_ADJ_AB") ? atoi(getenv("TRK_RADON_ADJ_AB")) : 0;
  const int tiles_x = ceil_div(N, tile_T);
  const int64_t adj_blocks = tile ? (int64_t)tiles_x * tiles_x : (int64_t)ceil_div((int64_t)N * N, 256);
  const int ndp = nd + 2 * A32_PAD;
  const bool adj_prep = !tile || na > 32;            // few angles per frame: the tile kernel makes its records itself
  // round 6: the adjoint by mirrored tile pairs (k_radon_adj_quad) where the handle allows it (whole 64 x 64 super-tiles, mostly
  // complete quads) and no rider travels on the epilogue (the damped-LSQR update and the mailbox post stay with k_radon_adj_tile);
  // TRK_RADON_NO_ADJQ=1: k_radon_adj_tile everywhere (read per call: the tests switch it)
  const bool adjq_riders = epi.on && (op->post.on || op->lsqr.on);
#ifdef TRK_ADJQ_EXPERIMENT_SPLIT
  if (tr && tile && im->adjq_ok && getenv("TRK_ADJQ_SPLIT")) nsplit = 1;      // (timing experiment: the quad kernel splits by itself)
#endif
  const bool adjq = tr && tile && im->adjq_ok && tile_T == 32 && nsplit == 1 && !adjq_riders && getenv("TRK_RADON_NO_ADJQ") == nullptr;
  const int adjq_th = N / 64;
  const int64_t adjq_blocks = (int64_t)adjq_th * adjq_th;
  if (epi.on && (batch != 1 || (tr && !tile))) return fail(TRK_EUNSUPPORTED, "radon: fused epilogue needs batch 1 and the tiled adjoint");
  // the forward's band reduction carries the epilogue / the fused norm / the adjoint's records
  const bool post = epi.on || im->n_bands > 1 || ext_part;
  if (ext_part) sumsq = ext_part;          // where a finished value goes when the partials do not fit / no kernel makes any
  const bool fuse_ssq = sumsq && batch == 1 && (tr ? tile : post);
  const int64_t post_blocks = ceil_div((int64_t)nt * na * ndp, 256);
  const int64_t n_part = tr ? (adjq ? adjq_blocks : adj_blocks) * nt : post_blocks;
  epi.pq = PostReq{};
  if (tr && tile && batch == 1 && epi.on && op->post.on) {
    epi.pq = op->post;                         // trk_gk_step_post: the mailbox post on the adjoint kernel's first workgroup
    op->post_taken = 1;
  }
  epi.lq = LsqrReq{};
  if (tr && tile && batch == 1 && epi.on && epi.z && op->lsqr.on && (!op->lsqr.ref || n_part <= op->lsqr.err_cap)) {
    epi.lq = op->lsqr;                         // trk_gk_step_lsqr: the iterate's update on the adjoint's pixel pass (z = its vk)
    op->lsqr_blocks = (int)n_part;
  }
  if (!tr && post && batch == 1 && op->probe_vec && post_blocks <= op->probe_cap) {      // trk_gk_step_proj: <out, probe_vec> partials
    epi.dotv = op->probe_vec;
    epi.dot_part = op->probe_part;
    op->probe_n = (int)post_blocks;
  }
  const bool defer = fuse_ssq && epi.on && (hints & HINT_SUMSQ_DEFERRED) && n_part <= im->pend_cap;
  const bool raw_out = ext_part && fuse_ssq && n_part <= ext_cap;
  if (ext_n) *ext_n = raw_out ? (int)n_part : 1;
  if (raw_out) {
    ssq_part = ext_part;
  } else if (defer) {
    ssq_part = im->pend_buf[im->pend_which];        // not the buffer a pending norm of the previous apply is read from
    im->pend_which ^= 1;
  } else if (fuse_ssq) {
    if (int rc = scratch_doubles(s, (size_t)n_part, &ssq_part)) return rc;
  }
  auto finish_norm = [&]() -> int {
    if (raw_out) return TRK_OK;
    if (defer) {
      im->pend_target = sumsq;
      im->pend_part = ssq_part;
      im->pend_n = (int)n_part;
      return TRK_OK;
    }
    return finalize_sums(ssq_part, (int)n_part, 1, 1, sumsq, s);
  };
  if (!tr) {
    for (int b = 0; b < batch; ++b) {  // the transposed copy is per vector
      const float* xb = x + (int64_t)b * ldx;
      const FwdPath fp = fwd_path(im, xb);
      const bool lds = fp.lds, dma = fp.dma;
      // the adjoint that produced xb may have left its transpose in xT already (hinted chain): then the copy costs nothing and the
      // kernel without the transposing staging is the faster one (512^2 x 180 inside Golub-Kahan: 24.5 vs 27.6 us)
      const bool have_xT = im->n_mode1 > 0 && (hints & HINT_INPUT_FROM_OPPOSITE) && im->xT_src == xb;
      const bool band_res = im->band_res && !fp.win && lds;
      const int direct1 = (fp.direct1 && !have_xT && !band_res) ? 1 : 0;
      if (im->n_mode1 > 0 && !direct1 && !band_res) {
        if (have_xT) {
        } else {
          dim3 g(ceil_div(N, 32), ceil_div(N, 32), nt);
          hipLaunchKernelGGL(k_transpose, g, dim3(256), 0, s, xb, im->xT, N);
        }
      }
      im->xT_src = nullptr;              // holds for this apply only: the caller's promise covers the very next one
      const int ndblk = ceil_div(nd, 64), ngrp = ceil_div(na, 4), nb = im->n_bands;
      const int64_t bs = (int64_t)nt * na * nd;
      dim3 grid(ndblk * ngrp * nt, nb, 1);
      float* yb = y + (int64_t)b * ldy;
      static const bool no_rec_out = getenv("TRK_RADON_NO_REC_OUT") != nullptr;     // tuning knobs: the producer side of the hints off
      // (the mirrored-pair adjoint reads records of its own kind, made by its own pre-pass: nothing to leave behind for it)
      const bool want_rec = (hints & HINT_OUT_FEEDS_OPPOSITE) && tile && adj_prep && batch == 1 && !no_rec_out &&
                            !(im->adjq_ok && tiles32 >= 1024 && getenv("TRK_RADON_NO_ADJQ") == nullptr);
      // measured: 512^2 35 us (shared) vs 32 us (per-wave windows); 2048^2 0.256 vs 0.277 ms; 4096^2 0.96 vs 1.11 ms
      if (band_res) {
        const int rows = im->band, nbr = N / rows;
        static const int slice_env = getenv("TRK_RADON_BANDRES_SLICES") ? atoi(getenv("TRK_RADON_BANDRES_SLICES")) : 0;
        // one workgroup of 16 waves per CU whatever the width (measured at 32 frames of 256^2: two workgroups of 8 waves per CU, which the
        // narrower band's LDS would allow, 18.4 us against 16.5)
        const size_t lds_bytes = sizeof(float) * (size_t)rows * (N + 2 * BR_PAD) + 16;
        static const int per_cu_env = getenv("TRK_RADON_BANDRES_PER_CU") ? atoi(getenv("TRK_RADON_BANDRES_PER_CU")) : 1;
        const int per_cu = (per_cu_env >= 2 && 2 * (lds_bytes + 512) <= 160 * 1024) ? 2 : 1;
        int nslice = slice_env > 0 ? slice_env : (cu_count() * per_cu + nt * 2 * nbr / 2) / (nt * 2 * nbr);
        if (nslice < 1) nslice = 1;
        static bool attr_set = false;
        if (!attr_set) {
          TRK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_radon_fwd_band), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
          attr_set = true;
        }
        hipLaunchKernelGGL(k_radon_fwd_band, dim3((unsigned)(nt * 2 * nbr * nslice)), dim3(per_cu >= 2 ? BR_NT / 2 : BR_NT), lds_bytes, s, xb, im->xT, im->part, N, nd, im->ang_dev,
                           na, im->adj_ang, im->adj_n0, nslice, bs, im->A32, im->B32, im->npad,
#ifdef TRK_RADON_BAND_EXPERIMENT
                           (have_xT ? 1 : 0) | (getenv("TRK_RADON_BAND_X") ? atoi(getenv("TRK_RADON_BAND_X")) & 6 : 0), rows);
#else
                           have_xT ? 1 : 0, rows);
#endif
      } else if (fp.win) {
        // window-sharing kernel: band partials of rays no window owns must read as zero
        if (hipMemsetAsync(im->part, 0, sizeof(float) * (size_t)nb * bs, s) != hipSuccess) return fail(TRK_EHIP, "radon: hipMemsetAsync failed");
        if (fp.quad) {
          const int nwq = ceil_div(N + im->band + 4, QD_WO), ngq = ceil_div(im->nq, 4);
          dim3 gq(8 * ceil_div(nwq, 8) * ngq * nt, nb, 1);           // windows dealt to the XCDs in contiguous eighths (see the kernel)
          // tiles in flight per workgroup: 2 where the chip is full of workgroups anyway (four per CU hide each other's staging round
          // trips), 4 where it is not (the chain of a workgroup's chunks is then bound by ONE round trip per chunk: see the kernel)
          static const int qbuf_env = getenv("TRK_RADON_QBUF") ? atoi(getenv("TRK_RADON_QBUF")) : 0;
          const int64_t wgs = (int64_t)nwq * ngq * nt * nb;
          // (measured: deeper staging does not pay at any size — the extra LDS costs resident workgroups, 512^2: 40 -> 60 us with four
          //  tiles in flight, 4096^2: 0.76 -> 1.34 ms; the knob stays for experiments)
          (void)wgs;
          const int qbuf = qbuf_env ? qbuf_env : 2;
          // round 6: the lean kernel for the workgroups its plan allows, k_radon_fwd_quad for the listed rest (TRK_RADON_NO_QUADF=1: all
          // of them, as rounds 4-5).  The plan depends on the geometry and the grid only: made at the first apply, kept with the handle
          const bool no_quadf = getenv("TRK_RADON_NO_QUADF") != nullptr;      // read per call: the tests switch it
          dim3 gslow = gq;
          const int* wg_list = nullptr;
          if (!no_quadf && qbuf == 2) {
            if (!im->qplan || im->qplan_gx != (int)gq.x || im->qplan_nb != nb) {
              if (im->qplan) (void)hipFree(im->qplan);
              if (im->qslow) (void)hipFree(im->qslow);
              im->qplan = nullptr;
              im->qslow = nullptr;
              const size_t nwg = (size_t)gq.x * nb;
              TRK_HIP(hipMalloc((void**)&im->qplan, sizeof(QuadPlan) * nwg));
              TRK_HIP(hipMalloc((void**)&im->qslow, sizeof(int) * (nwg + 1)));
              TRK_HIP(hipMemsetAsync(im->qslow, 0, sizeof(int), s));
              hipLaunchKernelGGL(k_radon_quad_plan, gq, dim3(256), 0, s, N, nd, im->quad_dev, im->nq, ngq, nwq, im->band, im->fidx,
                                 im->xT ? 1 : 0, im->qplan, im->qslow);
              TRK_HIP(hipMemcpyAsync(&im->qslow_n, im->qslow, sizeof(int), hipMemcpyDeviceToHost, s));
              TRK_HIP(hipStreamSynchronize(s));
              im->qplan_gx = (int)gq.x;
              im->qplan_nb = nb;
            }
            hipLaunchKernelGGL(k_radon_fwd_quadf, gq, dim3(256), 0, s, xb, im->xT, im->part, N, nd, im->quad_dev, im->nq, ngq, na, nwq, bs, im->band,
                               im->A32q, im->B32q, im->npad, im->qplan);
            gslow = dim3((unsigned)im->qslow_n, 1, 1);
            wg_list = im->qslow;
          }
#define QUAD(NB) hipLaunchKernelGGL(k_radon_fwd_quad<NB>, gslow, dim3(256), 0, s, xb, im->xT, im->part, N, nd, im->quad_dev, im->nq, ngq, na, nwq, bs, im->band, \
                                    im->fidx, im->A32q, im->B32q, im->npad, im->ang_dev, im->A32, im->B32, wg_list, (int)gq.x)
          if (gslow.x > 0) {
            if (qbuf >= 4) QUAD(4); else if (qbuf == 3) QUAD(3); else QUAD(2);
          }
#undef QUAD
        } else {
          const int nwin = ceil_div(N + 2 * im->band + 16, 61);
          dim3 gw(nwin * ngrp * nt, nb, 1);
          hipLaunchKernelGGL(k_radon_fwd_win<0>, gw, dim3(256), 0, s, xb, im->xT, im->part, N, nd, im->ang_dev, na, ngrp, nwin, bs, im->band, im->fidx, im->A32, im->B32, im->npad);
        }
      } else if (!post) {
        if (lds && dma) hipLaunchKernelGGL((k_radon_fwd_lds<true, true>), grid, dim3(256), 0, s, xb, im->xT, yb, N, nd, im->ang_dev, na, ngrp, ndblk, bs, im->band, im->A32, im->B32, im->npad, direct1);
        else if (lds) hipLaunchKernelGGL(k_radon_fwd_lds<true>, grid, dim3(256), 0, s, xb, im->xT, yb, N, nd, im->ang_dev, na, ngrp, ndblk, bs, im->band, im->A32, im->B32, im->npad, direct1);
        else hipLaunchKernelGGL(k_radon_fwd<true>, grid, dim3(256), 0, s, xb, im->xT, yb, N, nd, im->ang_dev, na, ngrp, ndblk, bs, im->band, im->A32, im->B32, im->npad);
      } else {
        if (lds && dma) hipLaunchKernelGGL((k_radon_fwd_lds<false, true>), grid, dim3(256), 0, s, xb, im->xT, im->part, N, nd, im->ang_dev, na, ngrp, ndblk, bs, im->band, im->A32, im->B32, im->npad, direct1);
        else if (lds) hipLaunchKernelGGL(k_radon_fwd_lds<false>, grid, dim3(256), 0, s, xb, im->xT, im->part, N, nd, im->ang_dev, na, ngrp, ndblk, bs, im->band, im->A32, im->B32, im->npad, direct1);
        else hipLaunchKernelGGL(k_radon_fwd<false>, grid, dim3(256), 0, s, xb, im->xT, im->part, N, nd, im->ang_dev, na, ngrp, ndblk, bs, im->band, im->A32, im->B32, im->npad);
      }
      if (post) {
        if (want_rec)
          hipLaunchKernelGGL(k_radon_bands_post<true>, dim3((unsigned)post_blocks), dim3(256), 0, s, im->part, nb, bs, yb, nd, im->ang_dev, epi,
                             ssq_part, im->rec, im->adj_pos, im->adj_ang, im->adj_wgt, im->A32);
        else
          hipLaunchKernelGGL(k_radon_bands_post<false>, dim3((unsigned)post_blocks), dim3(256), 0, s, im->part, nb, bs, yb, nd, im->ang_dev, epi,
                             ssq_part, im->rec, im->adj_pos, im->adj_ang, im->adj_wgt, im->A32);
        if (want_rec) im->rec_src = yb;
      }
      TRK_LAUNCH_CHECK();
    }
    if (ssq_part) {
      tm.stop();
      return finish_norm();
    }
  } else {
    for (int b = 0; b < batch; ++b) {            // the record array is per vector
      const float* xb = x + (int64_t)b * ldx;
      if (adjq) {
        static const bool no_xt_out_q = getenv("TRK_RADON_NO_XT_OUT") != nullptr;
        float* xT_out = ((hints & HINT_OUT_FEEDS_OPPOSITE) && im->n_mode1 > 0 && batch == 1 && !no_xt_out_q) ? im->xT : nullptr;
        hipLaunchKernelGGL(k_radon_adj_prepq, dim3(ceil_div((int64_t)im->nq * 4 * ndp, 256), nt), dim3(256), 0, s, xb, im->recq, nd, na, im->nq,
                           im->quad_dev, im->wq, im->A32q);
        im->rec_src = nullptr;
        static const int qb_env = getenv("TRK_RADON_ADJQ_QB") ? atoi(getenv("TRK_RADON_ADJQ_QB")) : 0;
        if (qb_env == 2)
          hipLaunchKernelGGL(k_radon_adj_quad<2>, dim3((unsigned)adjq_blocks, nt), dim3(256), 0, s, im->recq, y + (int64_t)b * ldy, N, nd, im->nq,
                             im->adjq, im->CBq, im->npad, adjq_th, ssq_part, epi, xT_out);
        else
#ifdef TRK_ADJQ_EXPERIMENT_SPLIT
          hipLaunchKernelGGL(k_radon_adj_quad<4>, dim3((unsigned)adjq_blocks, nt, getenv("TRK_ADJQ_SPLIT") ? atoi(getenv("TRK_ADJQ_SPLIT")) : 1), dim3(256), 0, s, im->recq, y + (int64_t)b * ldy, N, nd, im->nq,
                             im->adjq, im->CBq, im->npad, adjq_th, ssq_part, epi, xT_out);
#else
          hipLaunchKernelGGL(k_radon_adj_quad<4>, dim3((unsigned)adjq_blocks, nt), dim3(256), 0, s, im->recq, y + (int64_t)b * ldy, N, nd, im->nq,
                             im->adjq, im->CBq, im->npad, adjq_th, ssq_part, epi, xT_out);
#endif
        if (xT_out) im->xT_src = y + (int64_t)b * ldy;
        TRK_LAUNCH_CHECK();
        continue;
      }
      if (adj_prep) {
        if (!((hints & HINT_INPUT_FROM_OPPOSITE) && im->rec_src == xb))   // else: the forward that produced xb left its records
          hipLaunchKernelGGL(k_radon_adj_prep, dim3(ceil_div((int64_t)na * ndp, 256), nt), dim3(256), 0, s, xb, im->rec, nd, na,
                             im->adj_ang, im->adj_wgt, im->A32);
        im->rec_src = nullptr;
      }
      static const bool no_xt_out = getenv("TRK_RADON_NO_XT_OUT") != nullptr;
      float* xT_out = ((hints & HINT_OUT_FEEDS_OPPOSITE) && tile && im->n_mode1 > 0 && batch == 1 && !no_xt_out) ? im->xT : nullptr;
      if (nsplit > 1) {
        const int64_t need = (int64_t)nsplit * nt * adj_blocks * 1024, need_c = (int64_t)nt * adj_blocks;
        if (im->adj_part_cap < need) {
          if (im->adj_part) hipFree(im->adj_part);
          im->adj_part = nullptr;
          im->adj_part_cap = 0;
          if (hipMalloc((void**)&im->adj_part, sizeof(float) * (size_t)need) != hipSuccess) return fail(TRK_EHIP, "radon: hipMalloc (adjoint partial tiles) failed");
          im->adj_part_cap = need;
        }
        if (im->adj_cnt_cap < need_c) {
          if (im->adj_cnt) hipFree(im->adj_cnt);
          im->adj_cnt = nullptr;
          im->adj_cnt_cap = 0;
          if (hipMalloc((void**)&im->adj_cnt, sizeof(unsigned) * (size_t)need_c) != hipSuccess) return fail(TRK_EHIP, "radon: hipMalloc (adjoint tile counters) failed");
          if (hipMemsetAsync(im->adj_cnt, 0, sizeof(unsigned) * (size_t)need_c, s) != hipSuccess) return fail(TRK_EHIP, "radon: hipMemsetAsync failed");
          im->adj_cnt_cap = need_c;
        }
      }
#define ADJ_TILE(TT, PP, BB, PR)                                                                                              \
  hipLaunchKernelGGL((k_radon_adj_tile<TT, PP, BB, PR>), dim3((unsigned)(adj_blocks * nsplit), nt), dim3(256), 0, s, xb, im->rec, \
                     y + (int64_t)b * ldy, N, nd, na, im->adj_ang, im->adj_wgt, im->A32, im->adj_n0, im->CB, im->npad, tiles_x, \
                     ssq_part, epi, xT_out, nsplit, im->adj_part, im->adj_cnt)
      // the parts of a tile as groups of ONE workgroup where that gives about one workgroup per CU (512^2: 256 tiles)
      static const int grp_env = getenv("TRK_RADON_ADJ_GROUPS") ? atoi(getenv("TRK_RADON_ADJ_GROUPS")) : 1;
      const bool groups = grp_env != 0 && tile && tile_T == 32 && nsplit == 4 && adj_blocks * nt <= cu_count() && adj_blocks * nt * 4 >= 3 * cu_count();
      if (groups) {
        if (adj_prep) hipLaunchKernelGGL((k_radon_adj_tile<32, 4, 8, true, 4>), dim3((unsigned)adj_blocks, nt), dim3(1024), 0, s, xb, im->rec,
                                         y + (int64_t)b * ldy, N, nd, na, im->adj_ang, im->adj_wgt, im->A32, im->adj_n0, im->CB, im->npad, tiles_x,
                                         ssq_part, epi, xT_out, 1, im->adj_part, im->adj_cnt);
        else hipLaunchKernelGGL((k_radon_adj_tile<32, 4, 8, false, 4>), dim3((unsigned)adj_blocks, nt), dim3(1024), 0, s, xb, im->rec,
                                y + (int64_t)b * ldy, N, nd, na, im->adj_ang, im->adj_wgt, im->A32, im->adj_n0, im->CB, im->npad, tiles_x,
                                ssq_part, epi, xT_out, 1, im->adj_part, im->adj_cnt);
      } else if (tile && tile_T == 32) {
        // (16 angles per batch — half the barriers, twice the rings — measured at 512^2 x 180: 34.7 us against 29.9; not instantiated)
        // (TRK_RADON_ADJ_AB=16 — one batch for a 15-angle frame, half the barriers at 180 angles — measured again in round 6 on C5's
        //  shape, 32 frames of 256^2 x 15 angles: see profiles/r06/adj_ab16.txt)
        if (ab_env == 16) { if (adj_prep) ADJ_TILE(32, 4, 16, true); else ADJ_TILE(32, 4, 16, false); }
        else if (adj_prep) ADJ_TILE(32, 4, 8, true); else ADJ_TILE(32, 4, 8, false);
      } else if (tile && (ab_env ? ab_env == 16 : (na > 32 || adj_blocks * nt <= 4 * (int64_t)cu_count()))) {
        // (16 angles per batch also for the few frames of a dynamic problem one rank of eight holds — 4 frames of 256^2 x 15 angles, 1 024
        //  workgroups: 10.2 -> 9.1 us per apply, profiles/r06/adj_ab16.txt; with more workgroups than that the shorter batches win)
        if (adj_prep) ADJ_TILE(16, 1, 16, true); else ADJ_TILE(16, 1, 16, false);
      } else if (tile) {   // few angles per frame (dynamic problems: 15): short batches, so that staging and gathering still overlap
        if (adj_prep) ADJ_TILE(16, 1, 4, true); else ADJ_TILE(16, 1, 4, false);
      }
#undef ADJ_TILE
      else
        hipLaunchKernelGGL(k_radon_adj_simple, dim3((unsigned)adj_blocks, nt), dim3(256), 0, s, im->rec, y + (int64_t)b * ldy, N, nd, na,
                           im->adj_ang, im->adj_n0, im->CB, im->npad, (double*)nullptr);
      if (xT_out) im->xT_src = y + (int64_t)b * ldy;
      TRK_LAUNCH_CHECK();
    }
    if (ssq_part) {
      tm.stop();
      return finish_norm();
    }
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

// ref_mode != 0 (trk_radon2d_set_arithmetic): every apply of the handle through ref64.hip's float64-arithmetic kernels
int radon_ref_geom_of(RadonImpl* im, RadonRefGeom* g) {
  *g = RadonRefGeom{im->N, im->nd, im->na, im->nt, im->npad, im->ref_ang, im->A32, im->B32, im->ref_chunk_fwd, im->ref_chunk_adj};
  return TRK_OK;
}
int radon_apply_refmode(trk_op* op, int tr, const float* x, int64_t ldx, float* y, int64_t ldy, int batch, double* sumsq, hipStream_t s) {
  auto* im = static_cast<RadonImpl*>(op->impl);
  if (int rc = radon_flush(op, s)) return rc;
  im->rec_src = nullptr;
  im->xT_src = nullptr;
  RadonRefGeom g;
  radon_ref_geom_of(im, &g);
  for (int bi = 0; bi < batch; ++bi)
    if (int rc = radon_ref_apply_f32(g, tr, im->ref_mode - 1, x + (int64_t)bi * ldx, y + (int64_t)bi * ldy, s)) return rc;
  if (sumsq) {
    const int64_t nout = tr ? op->cols : op->rows;
    if (batch == 1 || ldy == nout) return trk_nrm2sq(y, nout * batch, sumsq, (trk_stream)s);
    return fail(TRK_EUNSUPPORTED, "radon: fused sum of squares needs contiguous batch outputs");
  }
  return TRK_OK;
}

int radon_apply(trk_op* op, int tr, const float* x, int64_t ldx, float* y, int64_t ldy, int batch, double* sumsq,
                hipStream_t s) {
  if (static_cast<RadonImpl*>(op->impl)->ref_mode) return radon_apply_refmode(op, tr, x, ldx, y, ldy, batch, sumsq, s);
  return radon_run(op, tr, x, ldx, y, ldy, batch, sumsq, Epi{}, 0, s);
}

// trk_op_apply_fused for the projector: only its one-operand form (x2 = NULL) — y = Op(x1) with sum(y^2) left as the block
// partials of the kernel that writes y (band reduction / tile gather), which the CGLS update kernels add up themselves
// (trk_cgls_update_xr_src, trk_cgls_p_update): the two reduction launches of a CGLS iteration disappear.
int radon_apply_fused(trk_op* op, int tr, const float* x1, const float* x2, double, ScalarSrc, ScalarSrc, float*, float* y,
                      double* partials, int cap, int* n_partials, hipStream_t s) {
  if (x2) return fail(TRK_EUNSUPPORTED, "radon: the two-operand fused apply is not available (trk_op_fused_caps reports 2)");
  if (cap < 1) return fail(TRK_EINVAL, "radon: fused apply needs room for at least one partial");
  if (static_cast<RadonImpl*>(op->impl)->ref_mode) {          // the instrument: one finished partial
    if (n_partials) *n_partials = 1;
    return radon_apply_refmode(op, tr, x1, tr ? op->rows : op->cols, y, tr ? op->cols : op->rows, 1, partials, s);
  }
  return radon_run(op, tr, x1, tr ? op->rows : op->cols, y, tr ? op->cols : op->rows, 1, nullptr, Epi{}, 0, s, partials, cap,
                   n_partials);
}

// out = a * Op(x) + b * z (+ ||out||^2) inside the kernel that writes the output: the band reduction (forward) or the tile
// gather (adjoint).  Small frames the tiled adjoint does not take fall back to apply + trk_axpby.
int radon_apply_axpby(trk_op* op, int tr, const float* x, Coef a, Coef b, const float* z, float* out, double* sumsq, int hints,
                      hipStream_t s) {
  auto* im = static_cast<RadonImpl*>(op->impl);
  if (im->ref_mode) {
    // the instrument: Op(x) by the float64-arithmetic kernels, then the half step's combination as the product's epilogue forms
    // it (float64 coefficients and products, one rounding) with the norm finished at once; the riders of trk_gk_step_* are not
    // taken (their callers run them in launches of their own)
    const int64_t nout = tr ? op->cols : op->rows;
    if (!im->ref_tmp) {
      const int64_t cap = op->rows > op->cols ? op->rows : op->cols;
      TRK_HIP(hipMalloc((void**)&im->ref_tmp, sizeof(float) * (size_t)cap));
    }
    if (int rc = radon_apply_refmode(op, tr, x, tr ? op->rows : op->cols, im->ref_tmp, nout, 1, nullptr, s)) return rc;
    return ref_axpby_f32(nout, a, im->ref_tmp, b, z, out, sumsq, s);
  }
  const bool tile = getenv("TRK_RADON_ADJ_SIMPLE") == nullptr && im->N >= 16;
  if (tr && !tile) {
    if (int rc = radon_run(op, tr, x, op->rows, out, op->cols, 1, nullptr, Epi{}, 0, s)) return rc;
    return trk_axpby(op->cols, a.c, a.num, a.den, a.flags, out, b.c, b.num, b.den, b.flags, z, out, sumsq, (trk_stream)s);
  }
  // The half step's combination a Op(x) + b z in float64 (coefficients as they are, one rounding of the result) — not in trk_axpby's
  // fp32 arithmetic, whose rounded coefficients perturb EVERY entry of a Golub-Kahan vector the same way: un-reorthogonalised
  // Lanczos amplified that 4.5 x at C3's semi-convergence transient (iterate 7 of Hybrid-LSQR at 512^2 x 180 against the float64
  // oracle: 1.57e-3 -> 3.5e-4; profiles/r04/c3_parity_epilogue.txt).  The kernels that write the output are far from the vector
  // unit's float64 rate.  TRK_RADON_EPI_F32=1 restores apply + trk_axpby to the bit.
  static const int epi_mode = getenv("TRK_RADON_EPI_F32") ? 1 : 2;
  return radon_run(op, tr, x, tr ? op->rows : op->cols, out, tr ? op->cols : op->rows, 1, sumsq, Epi{epi_mode, a, b, z, nullptr, nullptr, 0}, hints, s);
}

void radon_destroy(trk_op* op) {
  auto* im = static_cast<RadonImpl*>(op->impl);
  void* ptrs[] = {im->ref_ang, im->ref_tmp, im->quad_dev, im->A32q, im->B32q, im->ang_dev, im->xT, im->part, im->fidx, im->A32, im->B32, im->CB, im->adj_ang, im->adj_wgt, im->adj_n0, im->rec, im->adj_pos, im->pend_buf[0], im->pend_buf[1], im->adj_part, im->adj_cnt, im->qplan, im->qslow, im->adjq, im->CBq, im->wq, im->recq};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  delete im;
}

}  // namespace

static int radon_create_impl(int N, int n_det, const double* angles, int nt, int na, double scale, trk_op** out) {
  // the kernels address a frame through one buffer descriptor with 32-bit byte offsets
  if ((int64_t)N * N >= ((int64_t)1 << 29) || (int64_t)n_det >= ((int64_t)1 << 22))
    return fail(TRK_EUNSUPPORTED, "radon2d: frames of %d x %d pixels / %d detectors exceed the addressing of the kernels", N, N, n_det);
  if ((int64_t)nt * na > (int64_t)INT32_MAX / 4) return fail(TRK_EUNSUPPORTED, "radon2d: too many angles");
  const int n_ang = nt * na;
  const int ndp = n_det + 2 * A32_PAD;
  const int npad = ((N + ADJ_T - 1) / ADJ_T) * ADJ_T + 32;         // whole adjoint tiles + the forward's 16-entry scalar reads
  if ((int64_t)na * ndp * 16 >= ((int64_t)1 << 31) || (int64_t)na * npad * 8 >= ((int64_t)1 << 31))
    return fail(TRK_EUNSUPPORTED, "radon2d: %d angles x %d detectors per frame exceed the 2 GiB record addressing of the adjoint", na, n_det);
  std::vector<AngleParam> h(n_ang);
  std::vector<RadonRefAngle> href(n_ang);
  std::vector<float> wadj(n_ang);
  std::vector<unsigned> a32((size_t)n_ang * ndp), b32((size_t)n_ang * npad);
  std::vector<uint2> cb((size_t)n_ang * npad);
  const double half = 0.5 * (N - 1), sdh = 0.5 * (n_det - 1), one = (double)(1 << QF);
  auto fx = [&](double v) -> unsigned {                                   // round(v 2^24) mod 2^32
    return (unsigned)((uint64_t)(int64_t)std::llrint(v * one) & 0xFFFFFFFFull);
  };
  int n1 = 0;
  for (int a = 0; a < n_ang; ++a) {
    const double ct = std::cos(angles[a]), st = std::sin(angles[a]);
    double inv, dq, k0, w, rinv;
    AngleParam p;
    if (std::fabs(ct) >= std::fabs(st)) {
      p.mode = 0;
      inv = 1.0 / ct; dq = st / ct; k0 = half - half * st / ct; w = scale / std::fabs(ct); rinv = ct;
    } else {
      p.mode = 1;
      inv = -1.0 / st; dq = ct / st; k0 = half - half * ct / st; w = scale / std::fabs(st); rinv = -st;
      ++n1;
    }
    href[a] = RadonRefAngle{inv, dq, k0, w, p.mode, 0};
    p.inv = (float)inv; p.dq = (float)dq; p.k0 = (float)k0; p.rinv = (float)rinv;
    p.wgt = (float)(w / 4294967296.0);        // the forward kernels' weights are in units of 2^-32
    h[a] = p;
    wadj[a] = (float)w;                       // the adjoint's are plain
    for (int e = 0; e < ndp; ++e) a32[(size_t)a * ndp + e] = fx(((double)(e - A32_PAD) - sdh) * inv + k0);
    for (int t = 0; t < npad; ++t) {
      const unsigned B = fx((double)t * dq);
      b32[(size_t)a * npad + t] = B;
      const float C = (float)(sdh - (k0 + (double)t * dq) * rinv);      // d* = col rinv + C
      cb[(size_t)a * npad + t] = uint2{__builtin_bit_cast(unsigned, C), B};
    }
  }
  // ---- quads (k_radon_fwd_quad): per frame, angles that are images of one base angle beta in [0, 45 deg] under the symmetries of the
  // square grid share one wave.  With ctb = max(|cos|, |sin|), t = min / max (= tan beta), q_b(s, i) = s / ctb + h (1 - t) + i t:
  //   marching rows (|cos| >= |sin|):  cos > 0, sin >= 0: q = q_b(s, i)            cos < 0, sin >= 0: q = (N-1) - q_b(s, i)
  //                                    cos > 0, sin <  0: q = (N-1) - q_b(-s, i)   cos < 0, sin <  0: q = q_b(-s, i)
  //   marching columns (rows of xT):   sin > 0, cos >= 0: q = q_b(-s, j)           sin > 0, cos <  0: q = (N-1) - q_b(s, j)
  //                                    sin < 0, cos >= 0: q = (N-1) - q_b(-s, j)   sin < 0, cos <  0: q = q_b(s, j)
  // slot = 2 [xT] + 1 [mirrored]; -s = the detector index flipped.  The members' tables are then DERIVED from the base's, so that
  // the forward (base tables) and the adjoint (member tables) weigh every tap with the same bits:
  //   plain: A_m[e] = A_b[e'], B_m = B_b;   mirrored: A_m[e] = (N-1) 2^24 - A_b[e'], B_m = -B_b   (e' = e, or ndp-1-e when flipped).
  std::vector<QuadParam> quads;
  std::vector<unsigned> a32q, b32q;
  std::vector<AdjQuad> adjq_h;
  std::vector<uint2> cbq_h;
  std::vector<float> wq_h;
  bool adjq_mostly_full = false;
  int nq = 0;
  {
    struct Cand { double beta, ctb, t; int a, slot, flip; };
    std::vector<std::vector<QuadParam>> per_frame(nt);
    std::vector<std::vector<std::pair<double, double>>> geo(nt);          // (ctb, t) of every quad
    for (int f = 0; f < nt; ++f) {
      std::vector<Cand> c(na);
      for (int a = 0; a < na; ++a) {
        const double ct = std::cos(angles[(size_t)f * na + a]), st = std::sin(angles[(size_t)f * na + a]);
        Cand k;
        k.a = a;
        if (std::fabs(ct) >= std::fabs(st)) {
          k.ctb = std::fabs(ct); k.t = std::fabs(st) / std::fabs(ct);
          const bool cp = ct > 0, sp = st >= 0;
          k.slot = (cp == sp) ? 0 : 1;
          k.flip = sp ? 0 : 1;
        } else {
          k.ctb = std::fabs(st); k.t = std::fabs(ct) / std::fabs(st);
          const bool sp = st > 0, cp = ct >= 0;
          k.slot = 2 + ((sp == cp) ? 0 : 1);
          k.flip = (sp == cp) ? (sp ? 1 : 0) : (sp ? 0 : 1);
        }
        k.beta = std::atan2(k.t, 1.0);
        c[a] = k;
      }
      std::stable_sort(c.begin(), c.end(), [](const Cand& x, const Cand& y) { return x.beta < y.beta; });
      size_t i = 0;
      while (i < c.size()) {
        size_t j = i;
        while (j < c.size() && c[j].beta - c[i].beta <= 1e-12) ++j;
        // members i .. j-1 share the base; a slot met twice (the same angle given twice) opens another quad of the same base
        std::vector<char> used(j - i, 0);
        size_t left = j - i;
        while (left > 0) {
          QuadParam q{};
          q.inv = (float)(1.0 / c[i].ctb); q.dq = (float)c[i].t; q.k0 = (float)(half - half * c[i].t); q.rinv = (float)c[i].ctb;
          for (int m = 0; m < 4; ++m) q.am[m] = -1;
          for (size_t k = i; k < j; ++k) {
            if (used[k - i] || q.am[c[k].slot] >= 0) continue;
            q.am[c[k].slot] = c[k].a;
            q.mask |= 1 << c[k].slot;
            q.flip |= c[k].flip << c[k].slot;
            used[k - i] = 1;
            --left;
          }
          per_frame[f].push_back(q);
          geo[f].push_back({c[i].ctb, c[i].t});
        }
        i = j;
      }
      nq = std::max(nq, (int)per_frame[f].size());
    }
    quads.assign((size_t)nt * nq, QuadParam{});
    a32q.assign((size_t)nt * nq * ndp, 0u);
    b32q.assign((size_t)nt * nq * npad, 0u);
    const unsigned KN = (unsigned)(((uint64_t)(N - 1) << QF) & 0xFFFFFFFFull);
    for (int f = 0; f < nt; ++f) {
      for (size_t k = 0; k < per_frame[f].size(); ++k) {
        const size_t qr = (size_t)f * nq + k;
        quads[qr] = per_frame[f][k];
        const double ctb = geo[f][k].first, t = geo[f][k].second;
        const double inv = 1.0 / ctb, k0 = half - half * t;
        unsigned* Ab = &a32q[qr * ndp];
        unsigned* Bb = &b32q[qr * npad];
        for (int e = 0; e < ndp; ++e) Ab[e] = fx(((double)(e - A32_PAD) - sdh) * inv + k0);
        for (int tt = 0; tt < npad; ++tt) Bb[tt] = fx((double)tt * t);
        for (int m = 0; m < 4; ++m) {
          const int am = quads[qr].am[m];
          if (am < 0) continue;
          const size_t ar = (size_t)f * na + am;
          const bool mir = (m & 1) != 0, flp = ((quads[qr].flip >> m) & 1) != 0;
          for (int e = 0; e < ndp; ++e) {
            const unsigned v = Ab[flp ? ndp - 1 - e : e];
            a32[ar * ndp + e] = mir ? KN - v : v;
          }
          for (int tt = 0; tt < npad; ++tt) {
            const unsigned v = mir ? 0u - Bb[tt] : Bb[tt];
            b32[ar * npad + tt] = v;
            cb[ar * npad + tt].y = v;
          }
        }
      }
      for (size_t k = per_frame[f].size(); k < (size_t)nq; ++k)
        for (int m = 0; m < 4; ++m) quads[(size_t)f * nq + k].am[m] = -1;
    }
    // the adjoint by mirrored tile pairs (k_radon_adj_quad): base geometry per quad, its locator offsets and the members' weights
    adjq_h.assign((size_t)nt * nq, AdjQuad{0.f, 0.f, 1.f, 0.f, 0.f, 0});
    cbq_h.assign((size_t)nt * nq * npad, uint2{0u, 0u});
    wq_h.assign((size_t)nt * nq * 4, 0.f);
    int64_t members = 0;
    for (int f = 0; f < nt; ++f) {
      double carry = 0.0;                             // c1 on the 2^-24 grid, its rounding diffused over the quads in order of beta (see adj_ang below)
      for (size_t k = 0; k < per_frame[f].size(); ++k) {
        const size_t qr = (size_t)f * nq + k;
        const double ctb = geo[f][k].first, t = geo[f][k].second, k0 = half - half * t;
        const double c1x = (1.0 - 1.0 / ctb) * one, c1lo = std::floor(c1x), c1hi = std::ceil(c1x);
        const double cm = std::fabs(carry + (c1lo - c1x)) <= std::fabs(carry + (c1hi - c1x)) ? c1lo : c1hi, cp = cm;
        carry += cm - c1x;
        adjq_h[qr] = AdjQuad{(float)(cm / one), (float)(cp / one), (float)ctb, (float)t, (float)k0, 0};
        for (int tt = 0; tt < npad; ++tt) {
          const float C = (float)(sdh - (k0 + (double)tt * t) * ctb);
          cbq_h[qr * npad + tt] = uint2{__builtin_bit_cast(unsigned, C), b32q[qr * npad + tt]};
        }
        for (int m = 0; m < 4; ++m)
          if (quads[qr].am[m] >= 0) {
            wq_h[qr * 4 + m] = wadj[(size_t)f * na + quads[qr].am[m]];
            ++members;
          }
      }
    }
    adjq_mostly_full = 4 * members >= 3 * 4 * (int64_t)nt * nq;
  }
  int band = RADON_BAND;
  {
    // small frames (dynamic problems: 256^2 x 15 angles): 64-row bands double the workgroups of a grid that cannot fill the chip —
    // measured per apply: 4 frames (one rank's share of 32 on 8 GPUs) 13.2 -> 10.7 us, 8 frames 14.5 -> 11.6, 16 frames 17.7 -> 17.4,
    // 32 frames 24.6 -> 25.4.  Decided per FRAME, so that a dynamic handle and its frames' own handles sum in the same order.
    const int64_t wgs128 = (int64_t)((n_det + 63) / 64) * ((na + 3) / 4) * ((N + RADON_BAND - 1) / RADON_BAND);
    if (wgs128 < 64 && N > 64) band = 64;
  }
  // large images (the quad kernels): 256-row bands — half the band partials to clear, write and add up (4096^2 x 180: 94 -> 47 MB each
  // way; measured per apply 0.760 -> 0.725 ms with k_radon_fwd_quad, round 6), the four quads of a workgroup still inside one window
  if (N >= 2048 && N % QD_R == 0) band = 2 * RADON_BAND;
  // small images: a 64-row band of the image fits the LDS of a CU (k_radon_fwd_band).  TRK_RADON_NO_BANDRES=1: the per-wave windows
  const bool band_res = N % BR_ROWS == 0 && N >= 2 * BR_ROWS && N <= BR_NMAX && getenv("TRK_RADON_NO_BANDRES") == nullptr;
  if (band_res) band = br_rows(N);
  if (const char* e = getenv("TRK_RADON_BAND")) {               // tuning knob; kept a positive multiple of RADON_CHUNK
    band = atoi(e);
    band = band < RADON_CHUNK ? RADON_CHUNK : (band / RADON_CHUNK) * RADON_CHUNK;
  }
  const int nb = (N + band - 1) / band;
  auto* im = new RadonImpl{};
  im->N = N; im->nd = n_det; im->na = na; im->nt = nt; im->n_mode1 = n1; im->npad = npad; im->n_bands = nb; im->band = band;
  im->band_res = (band_res && band == br_rows(N)) ? 1 : 0;
  hipError_t e = hipSuccess;
  auto up = [&](void** dst, const void* src, size_t bytes) {
    if (e == hipSuccess) e = hipMalloc(dst, bytes);
    if (e == hipSuccess && src) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
  };
  up((void**)&im->ang_dev, h.data(), sizeof(AngleParam) * n_ang);
  up((void**)&im->ref_ang, href.data(), sizeof(RadonRefAngle) * n_ang);
  if (n1 > 0) up((void**)&im->xT, nullptr, sizeof(float) * (size_t)nt * N * N);
  up((void**)&im->part, nullptr, sizeof(float) * (size_t)nb * n_ang * n_det);   // nb = 1: used by the fused epilogue form
  {
    std::vector<float> fi((size_t)N + 16);
    for (size_t i = 0; i < fi.size(); ++i) fi[i] = (float)i;
    up((void**)&im->fidx, fi.data(), sizeof(float) * fi.size());
  }
  up((void**)&im->A32, a32.data(), sizeof(unsigned) * a32.size());
  up((void**)&im->B32, b32.data(), sizeof(unsigned) * b32.size());
  up((void**)&im->CB, cb.data(), sizeof(uint2) * cb.size());
  im->nq = nq;
  up((void**)&im->quad_dev, quads.data(), sizeof(QuadParam) * quads.size());
  up((void**)&im->A32q, a32q.data(), sizeof(unsigned) * a32q.size());
  up((void**)&im->B32q, b32q.data(), sizeof(unsigned) * b32q.size());
  // the mirrored-pair adjoint: images of whole 64 x 64 super-tiles, mostly complete quads (single angles would pay for four), from
  // 2048^2 on (measured per apply, 180 angles, k_radon_adj_tile -> k_radon_adj_quad: 4096^2 1.009 -> 0.794 ms, 2048^2 0.263 -> 0.222,
  // 1024^2 0.084 -> 0.101: 256 workgroups of four sub-phases leave the chip half empty there).  TRK_RADON_ADJQ_MIN: the tests' knob
  const int adjq_min = getenv("TRK_RADON_ADJQ_MIN") ? atoi(getenv("TRK_RADON_ADJQ_MIN")) : 2048;
  im->adjq_ok = (nq > 0 && N % 64 == 0 && N >= adjq_min && adjq_mostly_full && (int64_t)nq * 4 * ndp * 16 < ((int64_t)1 << 31)) ? 1 : 0;
  if (im->adjq_ok) {
    up((void**)&im->adjq, adjq_h.data(), sizeof(AdjQuad) * adjq_h.size());
    up((void**)&im->CBq, cbq_h.data(), sizeof(uint2) * cbq_h.size());
    up((void**)&im->wq, wq_h.data(), sizeof(float) * wq_h.size());
    up((void**)&im->recq, nullptr, sizeof(uint4) * (size_t)nt * nq * 4 * ndp);
  }
  {
    // adjoint tables: per frame, the angles with marching mode 0 first
    std::vector<AdjAngle> aa(n_ang);
    std::vector<int> n0(nt);
    std::vector<float> wg(n_ang);
    std::vector<int4> pos_of(n_ang);
    // What is left of that mismatch is rounding again, and again one value per angle.  The neighbour weights are formed as
    // clamp(c1 -+ t0 2^-24) by ONE fp32 FMA: t0 2^-24 lies on the 2^-24 grid, so whatever c1 holds below that grid is rounded away in
    // the result (exactly so for the weights in [0.5, 1), the ones that matter) — always the same way for an angle.  The float64
    // instrument separates it (profiles/r06/c3_instrument.txt: the product's neighbour RULE with unrounded weights sits on the fp32-
    // storage floor, the product 6-10 x above it).  So c1 is put ON the grid — the FMA is then exact, no rounding at all — and WHICH of
    // its two grid neighbours an angle gets is chosen by error diffusion over the angles in order of their direction: neighbouring
    // views back-project nearly the same field, so their biases (< 6e-8, alternating) cancel instead of adding up over 180 views.
    // (The kernels take the addend per SIDE — c1m for the neighbour on the smaller-q side, c1p for the other — so that a split
    //  {floor, ceil} with half the mean bias could be dealt as a third level.  Measured on C3, distance of iterates 5 / 6 / 7 / 8 from
    //  the float64 oracle: fp32 c1 from the fp32 inv 6.4e-6 / 4.5e-5 / 3.6e-4 / 1.06e-3 (round 5), fp32 c1 from the float64 inv
    //  7.9e-7 / 5.5e-6 / 3.3e-5 / 2.3e-4, on the grid with diffusion 3.2e-7 / 2.0e-6 / 1.2e-5 / 6.9e-5, with the split levels 3.8e-7 /
    //  2.5e-6 / 1.5e-5 / 8.9e-5; the fp32-storage floor 1.2e-7 / 6.5e-7 / 3.9e-6 / 2.0e-5 — so both sides get the same value.)
    std::vector<float> c1m_d(n_ang), c1p_d(n_ang);
    for (int f = 0; f < nt; ++f) {
      std::vector<int> order(na);
      for (int a = 0; a < na; ++a) order[a] = a;
      auto dir = [&](int a) { double t = std::fmod(angles[(size_t)f * na + a], M_PI); return t < 0 ? t + M_PI : t; };
      std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return dir(x) < dir(y); });
      double carry = 0.0;
      for (int a : order) {
        const double exact = (1.0 - std::fabs(href[(size_t)f * na + a].inv)) * one;       // in units of 2^-24 (<= 0)
        const double lo = std::floor(exact), hi = std::ceil(exact);
        const double pick = std::fabs(carry + (lo - exact)) <= std::fabs(carry + (hi - exact)) ? lo : hi;
        carry += pick - exact;
        c1m_d[(size_t)f * na + a] = c1p_d[(size_t)f * na + a] = (float)(pick / one);        // |pick| < 2^23: exact
      }
    }
    for (int f = 0; f < nt; ++f) {
      int pos = 0;
      for (int pass = 0; pass < 2; ++pass) {
        for (int a = 0; a < na; ++a) {
          const AngleParam& q = h[(size_t)f * na + a];
          if (q.mode != pass) continue;
          // c1 = 1 - |inv| from the float64 inv (round 6).  Formed from the fp32-rounded inv it was off by up to 7e-8 — the same
          // amount for EVERY neighbour weight of the angle, while the forward's weights are exact: a systematic mismatch between A
          // and A^T of the size the float64 instrument showed to cost C3's transient iterates 50 x (profiles/r06/c3_instrument.txt)
          aa[(size_t)f * na + pos] = AdjAngle{c1m_d[(size_t)f * na + a], c1p_d[(size_t)f * na + a], q.rinv, q.dq, q.k0, a, q.inv < 0.f ? 1 : 0, 0};
          wg[(size_t)f * na + pos] = wadj[(size_t)f * na + a];
          const float wa = wadj[(size_t)f * na + a];
          const int wbits = __builtin_bit_cast(int, wa);
          pos_of[(size_t)f * na + a] = int4{f * na + pos, wbits, q.inv < 0.f ? 1 : 0, 0};
          ++pos;
        }
        if (pass == 0) n0[f] = pos;
      }
    }
    up((void**)&im->adj_ang, aa.data(), sizeof(AdjAngle) * n_ang);
    up((void**)&im->adj_wgt, wg.data(), sizeof(float) * n_ang);
    up((void**)&im->adj_n0, n0.data(), sizeof(int) * nt);
    up((void**)&im->rec, nullptr, sizeof(uint4) * (size_t)n_ang * ndp);
    up((void**)&im->adj_pos, pos_of.data(), sizeof(int4) * n_ang);
    const int64_t t16 = (int64_t)ceil_div(N, 16) * ceil_div(N, 16) * nt, pb = ceil_div((int64_t)n_ang * ndp, 256);
    im->pend_cap = t16 > pb ? t16 : pb;
    up((void**)&im->pend_buf[0], nullptr, sizeof(double) * (size_t)im->pend_cap);
    up((void**)&im->pend_buf[1], nullptr, sizeof(double) * (size_t)im->pend_cap);
  }
  if (e != hipSuccess) {
    trk_op tmp{2, 0, 0, im, nullptr, nullptr, nullptr, 0};
    radon_destroy(&tmp);
    return fail(TRK_EHIP, "trk_radon2d_create: %s", hipGetErrorString(e));
  }
  *out = new trk_op{2, (int64_t)n_ang * n_det, (int64_t)nt * N * N, im, radon_apply, radon_destroy, nullptr, 0};
  (*out)->apply_axpby = radon_apply_axpby;
  (*out)->flush = radon_flush;
  (*out)->apply_fused = radon_apply_fused;
  (*out)->fused_caps = 2;                  // raw block partials only, no two-operand form
  return TRK_OK;
}

namespace trk {
bool radon_ref_geometry(trk_op* op, RadonRefGeom* g) {
  if (!op || op->kind != 2 || op->apply != radon_apply) return false;
  radon_ref_geom_of(static_cast<RadonImpl*>(op->impl), g);
  return true;
}
}  // namespace trk

extern "C" int trk_radon2d_set_arithmetic(trk_op* op, int mode) {
  TRK_REQUIRE(op && op->kind == 2 && op->apply == radon_apply, "trk_radon2d_set_arithmetic: not a parallel-beam handle");
  TRK_REQUIRE(mode >= 0 && mode <= 2, "trk_radon2d_set_arithmetic: mode 0 (product), 1 (float64 geometry and sums) or 2 (table weights, float64 sums)");
  auto* im = static_cast<RadonImpl*>(op->impl);
  if (im->pend_target) return fail(TRK_EINVAL, "trk_radon2d_set_arithmetic: a deferred norm is pending (trk_op_flush first)");
  im->ref_mode = mode;
  im->rec_src = nullptr;
  im->xT_src = nullptr;
  return TRK_OK;
}

extern "C" int trk_radon2d_set_ref_sums(trk_op* op, int chunk_fwd, int chunk_adj) {
  TRK_REQUIRE(op && op->kind == 2 && op->apply == radon_apply, "trk_radon2d_set_ref_sums: not a parallel-beam handle");
  TRK_REQUIRE(chunk_fwd >= 0 && chunk_adj >= -2, "trk_radon2d_set_ref_sums: chunk_fwd >= 0, chunk_adj >= 0 (0 = float64 sums) or -1 / -2 (the product's neighbour rule)");
  auto* im = static_cast<RadonImpl*>(op->impl);
  im->ref_chunk_fwd = chunk_fwd;
  im->ref_chunk_adj = chunk_adj;
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
