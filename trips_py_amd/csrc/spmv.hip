// spmv.hip — general sparse operator (CSR SpMV), for operators given as matrices: the reference builds its
// regularisers as scipy.sparse matrices (trips/utilities/operators.py:24-45 derivative operators, :50-113 framelet
// analysis operators) and the real-data dynamic problems ship a precomputed sparse forward matrix, used whole (F) and sliced
// into per-frame blocks (trips/utilities/io.py:132-135,197-229; demos/2_demo_dynamic_CrossPhantom.ipynb).  y = A x uses the CSR
// of A; y = A^T x uses the CSR of A^T (built once by the host) — both directions are gathers (no atomics).
//
// One kernel, k_csr_group<G>: a GROUP of G = 2 .. 64 neighbouring lanes owns a row (G = the largest power of two not above a quarter of the
// mean row length, 2 .. 16: 2 for first-difference and framelet rows, 16 for tomography rays), the wave's 64 / G groups own neighbouring
// rows — so the wave's loads of `vals` and `indices` are contiguous runs whatever the row length (round 4's one-thread-per-row
// kernel read rows of 2 .. 9 non-zeros with a stride of the row length across lanes) — every lane keeps four fp32 FMA chains,
// and float64 appears only in the reduction across the group (round 4: a float64 FMA per non-zero).  Row pointers are 32-bit on
// the device (nnz < 2^31): 4 bytes per row next to the 8 per non-zero.
// HBM-bound: 8 bytes per non-zero (value + column index) + 4 (m + n) for the vectors + 4 m row pointers; the gathers of x hit L2
// (a 256^2 frame is 256 KB).  Measured: profiles/r05/spmv.txt, DESIGN.md section 4.
#include "trk_internal.h"

#include <cstdlib>
#include <vector>

using namespace trk;

namespace {

constexpr int NT = 256;

struct Csr {
  int64_t nrows, ncols, nnz;
  unsigned* indptr;      // nrows + 1 row pointers
  int* indices;
  float* vals;
  int group;             // lanes per row
};

struct SpImpl {
  Csr a, at;
};

// (non-temporal loads of the matrix streams — nothing of them is re-read within an apply — LOSE: 38 -> 59 us on the Joseph matrix;
//  TRK_CSR_NT=1 keeps the variant for measurements)
template <bool NTL> __device__ __forceinline__ float ldm(const float* p) { return NTL ? __builtin_nontemporal_load(p) : *p; }
template <bool NTL> __device__ __forceinline__ int ldm(const int* p) { return NTL ? __builtin_nontemporal_load(p) : *p; }
// four consecutive entries of a stream in one 16-byte load at 4-byte alignment (a row starts wherever indptr says)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef int i4u __attribute__((ext_vector_type(4), aligned(4)));
template <bool NTL> __device__ __forceinline__ f4u ldm4(const float* p) {
  return NTL ? __builtin_nontemporal_load(reinterpret_cast<const f4u*>(p)) : *reinterpret_cast<const f4u*>(p);
}
template <bool NTL> __device__ __forceinline__ i4u ldm4(const int* p) {
  return NTL ? __builtin_nontemporal_load(reinterpret_cast<const i4u*>(p)) : *reinterpret_cast<const i4u*>(p);
}

// Accumulator type of the per-lane chains: fp32 (four chains per lane, float64 only across the group).  ADVICE r05 suggested float64 chains
// for the 2-lane rows (first differences, framelets: at most two products per chain) as "almost free" — measured in round 6
// (profiles/r06/spmv_ab.txt): first differences 2048^2 forward 48 -> 57 us, transpose 36 -> 45, space-time differences 20 -> 24, framelets
// 129 -> 138, and the one deviation it was meant for (MMGKS with the framelet regulariser, `Residual` against the reference) 3.2e-4 -> 2.2e-4:
// the deviation is not the SpMV's.  Not adopted; -DTRK_CSR_EXPERIMENT_F64_CHAINS builds it.
template <int G> struct ChainT { typedef float type; };
#ifdef TRK_CSR_EXPERIMENT_F64_CHAINS                // (A/B build switch; see the note above: measured, not adopted)
template <> struct ChainT<2> { typedef double type; };
#endif
__device__ __forceinline__ float chain_fma(float v, float x, float a) { return fmaf(v, x, a); }
__device__ __forceinline__ double chain_fma(float v, float x, double a) { return fma((double)v, (double)x, a); }

// KB right-hand sides per pass (KB = 1: y = A x; KB = 2 / 4 / 8: Y[:, b] = A X[:, b], columns ldx / ldy apart — `A @ V` on an (n, k)
// block, GKS.py:37 / MMGKS.py:44): the matrix streams (8 bytes per non-zero) are read ONCE for the KB columns; each column keeps
// the single-vector kernel's chains and summation order, so its result is bit-identical to a y = A x of that column alone.
// V4 (late round 6; rows of eight lanes and more): the row's non-zeros are dealt to the lanes in QUADS of four consecutive ones — lane g
// takes quads g, g + G, ... — so that a lane fetches its four values and its four column indices with ONE 16-byte load each, two quads in
// flight: half the load instructions of the entry-per-lane form (one load per value, one per index, one gather).  Measured on the 16-frame
// Joseph matrix, same box: forward (431 per row, 16 lanes) cold 44.8 -> 43.2 us, warm 34.1 -> 32.8; the transpose (17 per row, 4 lanes)
// LOSES with it, 43.8 -> 49.1 cold — hence eight lanes and more.  So the matrix streams' instruction count is not the bound either: what
// is left is the gathers of x, 17.7 M of them fetching a 64-byte sector each.  A lane's four chains take the four entries of its quads;
// what is left of a row after its last whole quad (<= 3 entries) goes to lanes 0 .. 2 on their first chain.  Every column of a batch keeps
// this order: bit-identical to its single apply, as before.
template <int G, int KB, bool SUMSQ, bool NTL, bool V4>
__global__ __launch_bounds__(NT) void k_csr_group(int64_t nrows, const unsigned* __restrict__ indptr,
                                                  const int* __restrict__ indices, const float* __restrict__ vals,
                                                  const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
                                                  double* __restrict__ partials, int pstride) {
  typedef typename ChainT<G>::type acc_t;
  __shared__ double red[NT / 64];
  const int g = threadIdx.x & (G - 1);
  // Rows in CONTIGUOUS spans per workgroup, the spans dealt to the XCDs in contiguous eighths (round 6; workgroups b and b + 8 share an
  // XCD under round-robin placement: speed only).  A block-diagonal matrix of frames (io.py:223-225) then has every XCD's L2 gather
  // from its own frames' slice of x instead of all of x: the grid-stride form fetched 192.8 MB for 145.7 MB of algorithmic bytes on
  // the 16-frame Joseph matrix (profiles/r05/traffic_spmv.txt), the gathers of x leaving L2 once per XCD and stride.
  const int nb = gridDim.x;
  // ... for rows of more than two lanes.  Two-lane rows (2 .. 9 non-zeros: a workgroup's trip is 128 rows = 2 KB of each stream) keep the
  // grid-stride map — trip t of the whole grid covers one contiguous run of rows: framelets 512^2 forward 110 us against 129 with spans,
  // first differences the same either way (profiles/r06/spmv_ab.txt).
  constexpr bool SPANS = G > 2;
  const int bid = (nb & 7) == 0 ? (int)(blockIdx.x & 7) * (nb >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  constexpr int RPT = NT / G;                                        // rows per trip of a workgroup
  const int64_t span = ((nrows + nb - 1) / nb + RPT - 1) / RPT * RPT;
  const int64_t r_end = (bid + 1) * span < nrows ? (bid + 1) * span : nrows;
  double ss[KB];
#pragma unroll
  for (int b = 0; b < KB; ++b) ss[b] = 0.0;
  const int64_t r_first = SPANS ? bid * span + threadIdx.x / G : ((int64_t)blockIdx.x * NT + threadIdx.x) / G;
  const int64_t r_stop = SPANS ? r_end : nrows, r_step = SPANS ? (int64_t)RPT : (int64_t)nb * RPT;
  for (int64_t r = r_first; r < r_stop; r += r_step) {
    const unsigned p0 = indptr[r], p1 = indptr[r + 1];
    acc_t a0[KB], a1[KB], a2[KB], a3[KB];
#pragma unroll
    for (int b = 0; b < KB; ++b) a0[b] = a1[b] = a2[b] = a3[b] = (acc_t)0;
    unsigned p = p0 + g;
    if (V4) {
      const unsigned nq = (p1 - p0) >> 2;
      unsigned q = g;
      for (; q + G < nq; q += 2 * G) {                                 // two quads of each stream in flight per lane
        const unsigned ba = p0 + 4 * q, bb = p0 + 4 * (q + G);
        const f4u va = ldm4<NTL>(vals + ba), vb = ldm4<NTL>(vals + bb);
        const i4u ca = ldm4<NTL>(indices + ba), cb = ldm4<NTL>(indices + bb);
#pragma unroll
        for (int b = 0; b < KB; ++b) {
          const float* __restrict__ xb = x + (int64_t)b * ldx;
          const float x0 = xb[ca[0]], x1 = xb[ca[1]], x2 = xb[ca[2]], x3 = xb[ca[3]];
          const float x4 = xb[cb[0]], x5 = xb[cb[1]], x6 = xb[cb[2]], x7 = xb[cb[3]];
          a0[b] = chain_fma(va[0], x0, a0[b]);
          a1[b] = chain_fma(va[1], x1, a1[b]);
          a2[b] = chain_fma(va[2], x2, a2[b]);
          a3[b] = chain_fma(va[3], x3, a3[b]);
          a0[b] = chain_fma(vb[0], x4, a0[b]);
          a1[b] = chain_fma(vb[1], x5, a1[b]);
          a2[b] = chain_fma(vb[2], x6, a2[b]);
          a3[b] = chain_fma(vb[3], x7, a3[b]);
        }
      }
      if (q < nq) {
        const unsigned ba = p0 + 4 * q;
        const f4u va = ldm4<NTL>(vals + ba);
        const i4u ca = ldm4<NTL>(indices + ba);
#pragma unroll
        for (int b = 0; b < KB; ++b) {
          const float* __restrict__ xb = x + (int64_t)b * ldx;
          a0[b] = chain_fma(va[0], xb[ca[0]], a0[b]);
          a1[b] = chain_fma(va[1], xb[ca[1]], a1[b]);
          a2[b] = chain_fma(va[2], xb[ca[2]], a2[b]);
          a3[b] = chain_fma(va[3], xb[ca[3]], a3[b]);
        }
      }
      {                                                                 // the row's last <= 3 entries: lanes 0 .. 2, first chain
        const unsigned pt = p0 + 4 * nq + g;
        const bool ht = g < 3 && pt < p1;
        const float vt = ht ? ldm<NTL>(vals + pt) : 0.f;
        const int ct = ht ? ldm<NTL>(indices + pt) : 0;
#pragma unroll
        for (int b = 0; b < KB; ++b) {
          const float* __restrict__ xb = x + (int64_t)b * ldx;
          a0[b] = chain_fma(vt, ht ? xb[ct] : 0.f, a0[b]);
        }
      }
      p = p1;                                                           // (the entry-per-lane loops below find nothing left)
    }
    // one right-hand side, long rows: EIGHT loads of each stream in flight per lane (round 6).  From HBM — the matrix of a real dynamic
    // problem does not fit the memory-side cache — a trip is one memory round trip plus one gather round trip, and a row of 431 non-zeros
    // over 16 lanes was seven dependent trips of four: 46 us cold against 33 us warm on the 16-frame Joseph matrix.  Same four chains,
    // the second four products behind the first four.
    if (KB == 1 && !V4) {
      // software-pipelined: the NEXT trip's values and indices are requested behind this trip's gathers, so that a trip costs the longer
      // of the two round trips (matrix from HBM, x from L2) instead of their sum
      if (p + 7 * G < p1) {
        float v[8];
        int c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ldm<NTL>(vals + p + u * G);
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = ldm<NTL>(indices + p + u * G);
        for (;;) {
          float xv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) xv[u] = x[c[u]];
          const unsigned pn = p + 8 * G;
          const bool more = pn + 7 * G < p1;                         // (per group; the loads below are predicated, not branched around)
          float vn[8];
          int cn[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) vn[u] = more ? ldm<NTL>(vals + pn + u * G) : 0.f;
#pragma unroll
          for (int u = 0; u < 8; ++u) cn[u] = more ? ldm<NTL>(indices + pn + u * G) : 0;
          a0[0] = chain_fma(v[0], xv[0], a0[0]);
          a1[0] = chain_fma(v[1], xv[1], a1[0]);
          a2[0] = chain_fma(v[2], xv[2], a2[0]);
          a3[0] = chain_fma(v[3], xv[3], a3[0]);
          a0[0] = chain_fma(v[4], xv[4], a0[0]);
          a1[0] = chain_fma(v[5], xv[5], a1[0]);
          a2[0] = chain_fma(v[6], xv[6], a2[0]);
          a3[0] = chain_fma(v[7], xv[7], a3[0]);
          p = pn;
          if (!more) break;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            v[u] = vn[u];
            c[u] = cn[u];
          }
        }
      }
    }
    // four loads of each stream in flight per lane, four independent chains per column (long rows: a wave streams 2 KB per trip)
    for (; p + 3 * G < p1; p += 4 * G) {
      const float v0 = ldm<NTL>(vals + p), v1 = ldm<NTL>(vals + p + G), v2 = ldm<NTL>(vals + p + 2 * G), v3 = ldm<NTL>(vals + p + 3 * G);
      const int c0 = ldm<NTL>(indices + p), c1 = ldm<NTL>(indices + p + G), c2 = ldm<NTL>(indices + p + 2 * G), c3 = ldm<NTL>(indices + p + 3 * G);
#pragma unroll
      for (int b = 0; b < KB; ++b) {
        const float* __restrict__ xb = x + (int64_t)b * ldx;
        a0[b] = chain_fma(v0, xb[c0], a0[b]);
        a1[b] = chain_fma(v1, xb[c1], a1[b]);
        a2[b] = chain_fma(v2, xb[c2], a2[b]);
        a3[b] = chain_fma(v3, xb[c3], a3[b]);
      }
    }
    // up to three more, issued together (predicated: a row's tail, or the whole of a short row)
    if (!V4) {
      const bool h0 = p < p1, h1 = p + G < p1, h2 = p + 2 * G < p1;
      const float v0 = h0 ? ldm<NTL>(vals + p) : 0.f, v1 = h1 ? ldm<NTL>(vals + p + G) : 0.f, v2 = h2 ? ldm<NTL>(vals + p + 2 * G) : 0.f;
      const int c0 = h0 ? ldm<NTL>(indices + p) : 0, c1 = h1 ? ldm<NTL>(indices + p + G) : 0, c2 = h2 ? ldm<NTL>(indices + p + 2 * G) : 0;
#pragma unroll
      for (int b = 0; b < KB; ++b) {
        const float* __restrict__ xb = x + (int64_t)b * ldx;
        a0[b] = chain_fma(v0, h0 ? xb[c0] : 0.f, a0[b]);
        a1[b] = chain_fma(v1, h1 ? xb[c1] : 0.f, a1[b]);
        a2[b] = chain_fma(v2, h2 ? xb[c2] : 0.f, a2[b]);
      }
    }
#pragma unroll
    for (int b = 0; b < KB; ++b) {
      double acc = ((double)a0[b] + (double)a1[b]) + ((double)a2[b] + (double)a3[b]);
#pragma unroll
      for (int off = G / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
      if (g == 0) {
        const float o = (float)acc;
        y[(int64_t)b * ldy + r] = o;
        if (SUMSQ) ss[b] += (double)o * o;
      }
    }
  }
  if (SUMSQ) {
#pragma unroll
    for (int b = 0; b < KB; ++b) {
      const double t = block_sum<NT>(ss[b], red);
      if (threadIdx.x == 0) partials[(size_t)b * pstride + blockIdx.x] = t;
      if (KB > 1) __syncthreads();
    }
  }
}

template <int G, int KB>
void launch_group(const Csr& M, int grid, const float* xb, int64_t ldx, float* yb, int64_t ldy, double* pb, hipStream_t s) {
  static const bool ntl = getenv("TRK_CSR_NT") && atoi(getenv("TRK_CSR_NT")) != 0;
  static const bool no_v4 = getenv("TRK_CSR_NO_VEC4") && atoi(getenv("TRK_CSR_NO_VEC4")) != 0;     // (the entry-per-lane form everywhere)
  constexpr bool V = G >= 8;
#define CG(SS, NN, VV) hipLaunchKernelGGL((k_csr_group<G, KB, SS, NN, VV>), dim3(grid), dim3(NT), 0, s, M.nrows, M.indptr, M.indices, M.vals, xb, ldx, yb, ldy, pb, grid)
  if (V && !no_v4) {
    if (pb) { if (ntl) CG(true, true, V); else CG(true, false, V); }
    else    { if (ntl) CG(false, true, V); else CG(false, false, V); }
  } else {
    if (pb) { if (ntl) CG(true, true, false); else CG(true, false, false); }
    else    { if (ntl) CG(false, true, false); else CG(false, false, false); }
  }
#undef CG
}

template <int KB>
void launch_cols(const Csr& M, int grid, const float* xb, int64_t ldx, float* yb, int64_t ldy, double* pb, hipStream_t s) {
  switch (M.group) {
    case 2: launch_group<2, KB>(M, grid, xb, ldx, yb, ldy, pb, s); break;
    case 4: launch_group<4, KB>(M, grid, xb, ldx, yb, ldy, pb, s); break;
    case 8: launch_group<8, KB>(M, grid, xb, ldx, yb, ldy, pb, s); break;
    case 16: launch_group<16, KB>(M, grid, xb, ldx, yb, ldy, pb, s); break;
    case 32: launch_group<32, KB>(M, grid, xb, ldx, yb, ldy, pb, s); break;
    default: launch_group<64, KB>(M, grid, xb, ldx, yb, ldy, pb, s); break;
  }
}

int sp_apply(trk_op* op, int tr, const float* x, int64_t ldx, float* y, int64_t ldy, int batch, double* sumsq,
             hipStream_t s) {
  auto* im = static_cast<SpImpl*>(op->impl);
  const Csr& M = tr ? im->at : im->a;
  // groups cover the rows once up to 16 workgroups per CU, grid-stride beyond (first differences of a 2048^2 image, 8.4 M rows of 2:
  // 55 us with one group per row, 46 with the cap; longer rows: no difference); a reduction leaves <= kMaxPartialBlocks partials
  int64_t want = (M.nrows * M.group + NT - 1) / NT;
  static const int64_t cap_env = getenv("TRK_CSR_GRID_PER_CU") ? atoll(getenv("TRK_CSR_GRID_PER_CU")) * cu_count() : 0;
  int64_t cap = cap_env > 0 ? cap_env : (int64_t)cu_count() * 16;
  if (sumsq && cap > kMaxPartialBlocks) cap = kMaxPartialBlocks;
  if (want > cap) want = cap;
  const int grid = (int)(want < 1 ? 1 : want);
  double* part = nullptr;
  if (sumsq)
    if (int rc = scratch_doubles(s, (size_t)grid * batch, &part)) return rc;
  // columns per pass (round 6): a batch — `A @ V`, GKS.py:37 / MMGKS.py:44 — goes through in passes of 8 / 4 / 2 columns, each pass
  // reading the matrix once; TRK_CSR_COLS=1 restores one launch per column (A/B runs)
  static const int max_cols = getenv("TRK_CSR_COLS") ? atoi(getenv("TRK_CSR_COLS")) : 8;
  TimerScope tm(op->timer, op->timer_which, tr, s);
  for (int b = 0; b < batch;) {
    const float* xb = x + (int64_t)b * ldx;
    float* yb = y + (int64_t)b * ldy;
    double* pb = part ? part + (size_t)b * grid : nullptr;
    const int left = batch - b;
    if (left >= 8 && max_cols >= 8) { launch_cols<8>(M, grid, xb, ldx, yb, ldy, pb, s); b += 8; }
    else if (left >= 4 && max_cols >= 4) { launch_cols<4>(M, grid, xb, ldx, yb, ldy, pb, s); b += 4; }
    else if (left >= 2 && max_cols >= 2) { launch_cols<2>(M, grid, xb, ldx, yb, ldy, pb, s); b += 2; }
    else { launch_cols<1>(M, grid, xb, ldx, yb, ldy, pb, s); b += 1; }
  }
  tm.stop();
  TRK_LAUNCH_CHECK();
  if (sumsq) return finalize_sums(part, grid * batch, 1, 1, sumsq, s);
  return TRK_OK;
}

void csr_free(Csr& c) {
  if (c.indptr) (void)hipFree(c.indptr);
  if (c.indices) (void)hipFree(c.indices);
  if (c.vals) (void)hipFree(c.vals);
}

void sp_destroy(trk_op* op) {
  auto* im = static_cast<SpImpl*>(op->impl);
  csr_free(im->a);
  csr_free(im->at);
  delete im;
}

int csr_upload(Csr& c, int64_t nrows, int64_t ncols, int64_t nnz, const int64_t* indptr, const int* indices,
               const float* vals) {
  // lanes per row: the largest power of two not above a QUARTER of the mean row length, 2 .. 16.  Measured (profiles/r05/spmv.txt,
  // us per apply): rows of 17 (Joseph transpose, 17.7 M non-zeros) 16 / 8 / 4 lanes -> 68.8 / 42.8 / 30.6; rows of 431 (Joseph)
  // 64 / 32 / 16 -> 38.4 / 36.7 / 36.2; rows of 169 (framelet transpose) 64 / 32 / 16 -> 162 / 134 / 112; rows of 7 (framelets)
  // 4 / 2 -> 139 / 108: many short chains per wave keep more loads in flight than one long coalesced one.
  // TRK_CSR_GROUP=<2..64> forces it, -1 / -2 halve / quarter the rule's choice (tuning).
  int group = 2;
  while (group < 16 && (int64_t)8 * group * nrows <= nnz) group *= 2;
  if (const char* e = getenv("TRK_CSR_GROUP")) {
    const int v = atoi(e);
    if (v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) group = v;
    if (v < 0 && group > 2 && -v <= 3) group = group >> (-v) < 2 ? 2 : group >> (-v);
  }
  c = Csr{nrows, ncols, nnz, nullptr, nullptr, nullptr, group};
  std::vector<unsigned> ip32((size_t)nrows + 1);
  for (int64_t r = 0; r <= nrows; ++r) {
    if (indptr[r] < 0 || indptr[r] > nnz || (r > 0 && indptr[r] < indptr[r - 1]))
      return fail(TRK_EINVAL, "trk_csr_create: row pointers must be non-decreasing inside [0, nnz]");
    ip32[(size_t)r] = (unsigned)indptr[r];
  }
  for (int64_t p = 0; p < nnz; ++p)
    if (indices[p] < 0 || indices[p] >= ncols) return fail(TRK_EINVAL, "trk_csr_create: column index %d outside [0, %lld)", indices[p], (long long)ncols);
  TRK_HIP(hipMalloc(&c.indptr, sizeof(unsigned) * (size_t)(nrows + 1)));
  TRK_HIP(hipMalloc(&c.indices, sizeof(int) * (size_t)(nnz > 0 ? nnz : 1)));
  TRK_HIP(hipMalloc(&c.vals, sizeof(float) * (size_t)(nnz > 0 ? nnz : 1)));
  TRK_HIP(hipMemcpy(c.indptr, ip32.data(), sizeof(unsigned) * (size_t)(nrows + 1), hipMemcpyHostToDevice));
  if (nnz > 0) {
    TRK_HIP(hipMemcpy(c.indices, indices, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
    TRK_HIP(hipMemcpy(c.vals, vals, sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice));
  }
  return TRK_OK;
}

}  // namespace

extern "C" int trk_csr_create(int64_t nrows, int64_t ncols, int64_t nnz, const int64_t* indptr_host, const int* indices_host,
                              const float* values_host, const int64_t* t_indptr_host, const int* t_indices_host,
                              const float* t_values_host, trk_op** out) {
  TRK_REQUIRE(out && indptr_host && t_indptr_host, "trk_csr_create: NULL argument");
  TRK_REQUIRE(nrows >= 1 && ncols >= 1 && nnz >= 0 && ncols < ((int64_t)1 << 31) && nrows < ((int64_t)1 << 31) && nnz < ((int64_t)1 << 31),
              "trk_csr_create: bad sizes (rows, columns and non-zeros must each be below 2^31)");
  TRK_REQUIRE(nnz == 0 || (indices_host && values_host && t_indices_host && t_values_host), "trk_csr_create: NULL arrays");
  TRK_REQUIRE(indptr_host[nrows] == nnz && t_indptr_host[ncols] == nnz, "trk_csr_create: indptr does not end at nnz");
  auto* im = new SpImpl{};
  int rc = csr_upload(im->a, nrows, ncols, nnz, indptr_host, indices_host, values_host);
  if (!rc) rc = csr_upload(im->at, ncols, nrows, nnz, t_indptr_host, t_indices_host, t_values_host);
  if (rc) {
    trk_op tmp{6, 0, 0, im, nullptr, nullptr, nullptr, 0};
    sp_destroy(&tmp);
    return rc;
  }
  *out = new trk_op{6, nrows, ncols, im, sp_apply, sp_destroy, nullptr, 0};
  return TRK_OK;
}
