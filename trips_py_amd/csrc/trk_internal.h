// trk_internal.h — shared host/device helpers of libtrk.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "trk.h"

namespace trk {

// ------------------------------------------------------------------ errors (thread-local string)
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define TRK_HIP(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return ::trk::fail(TRK_EHIP, "%s -> %s (%s:%d)", #expr,             \
                                             hipGetErrorString(e_), __FILE__, __LINE__);      \
  } while (0)

#define TRK_REQUIRE(cond, ...)                                  \
  do {                                                          \
    if (!(cond)) return ::trk::fail(TRK_EINVAL, __VA_ARGS__);   \
  } while (0)

#define TRK_LAUNCH_CHECK() TRK_HIP(hipGetLastError())

// ------------------------------------------------------------------ per-stream scratch (block partial sums)
// Partials of a reduction are written by the producing kernel and summed, in a fixed order, by a
// one-block finalize kernel on the same stream: deterministic, no atomics, no fences.
constexpr int kMaxPartialBlocks = 1024;
int scratch_doubles(hipStream_t s, size_t count, double** out);  // grows, never shrinks

// out[o] = sum_{b < nblocks} partials[b*stride + o]   for o < nout   (fixed summation order)
int finalize_sums(const double* partials, int nblocks, int stride, int nout, double* out_dev, hipStream_t s);
int finalize_sums_split(const double* partials, int nblocks, int stride, int nout, double* out_dev, int nsplit, double* out2_dev,
                        hipStream_t s);

// a zeroed word per (device, stream) for kernels whose LAST workgroup carries a once-per-launch duty (it takes tickets with
// agent-scope atomic adds and puts the word back to zero): launches on one stream run in order, so one word per stream is enough
int stream_ticket(hipStream_t s, unsigned** out);

// device facts (cached)
int cu_count();

// ------------------------------------------------------------------ device-evaluated coefficient
struct Coef {
  double c;
  const double* num;
  const double* den;
  int flags;
};

__device__ __forceinline__ double coef_eval(const Coef& k) {
  double v = k.c;
  if (k.num) {
    double t = *k.num;
    v *= (k.flags & TRK_SQRT_NUM) ? sqrt(t) : t;
  }
  if (k.den) {
    double t = *k.den;
    v /= (k.flags & TRK_SQRT_DEN) ? sqrt(t) : t;
  }
  return v;
}

// ------------------------------------------------------------------ wave64 / block reductions (fp64)
__device__ __forceinline__ double wave_sum_trees(double v) {      // (the reference form: six ds_bpermute pairs through the LDS crossbar)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // lane 0 holds the wave total
}

// ---- cross-lane moves on the vector unit (gfx950: v_permlane32_swap / v_permlane16_swap; DPP within a row of 16 lanes)
__device__ __forceinline__ void swap_halves32(double& a, double& b) {      // a's lanes 32-63 <-> b's lanes 0-31
  unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
  unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
  auto r = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
  alo = r[0]; blo = r[1];
  r = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
  ahi = r[0]; bhi = r[1];
  a = __hiloint2double((int)ahi, (int)alo);
  b = __hiloint2double((int)bhi, (int)blo);
}
__device__ __forceinline__ void swap_rows16(double& a, double& b) {        // a's odd rows of 16 lanes <-> b's even rows
  unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
  unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
  auto r = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
  alo = r[0]; blo = r[1];
  r = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
  ahi = r[0]; bhi = r[1];
  a = __hiloint2double((int)ahi, (int)alo);
  b = __hiloint2double((int)bhi, (int)blo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// the value of lane ^ OFF (OFF = 8, 4, 2, 1): row_ror:8 | row_half_mirror then the quads reversed | quad_perm [2,3,0,1] | [1,0,3,2]
template <int OFF>
__device__ __forceinline__ double lane_xor_f64(double v) {
  static_assert(OFF == 8 || OFF == 4 || OFF == 2 || OFF == 1, "within a row of 16 lanes");
  if (OFF == 8) return dpp_f64<0x128>(v);
  if (OFF == 4) return dpp_f64<0x1B>(dpp_f64<0x141>(v));
  if (OFF == 2) return dpp_f64<0x4E>(v);
  return dpp_f64<0xB1>(v);
}
// wave_sum_trees' tree (lane l with l + 32, the results with l + 16, ...) on the vector unit: two half / row swaps, then DPP; the same
// pairs in the same order, the same bits in lane 0 — the only lane either form promises.
__device__ __forceinline__ double wave_sum(double v) {
  double t = v, u = v;
  swap_halves32(t, u);            // u's lanes 0-31 <- v's lanes 32-63
  v = v + u;
  t = v; u = v;
  swap_rows16(t, u);              // u's even rows <- v's odd rows
  v = v + u;
  v = v + lane_xor_f64<8>(v);
  v = v + lane_xor_f64<4>(v);
  v = v + lane_xor_f64<2>(v);
  v = v + lane_xor_f64<1>(v);
  return v;  // lane 0 holds the wave total
}


// MM weight w = (v^2 + eps^2)^(e), e = p/2 - 1 (trips/solvers/MMGKS.py:57,93).  e == -0.5 (q = 1, the TV case) is an
// rsqrt; e == 0 is 1.
__device__ __forceinline__ float mm_w(float v, float eps2, float e, int special) {
  const float t = fmaf(v, v, eps2);
  if (special == 1) return 1.0f;
  // e = -1/2 (qnorm = 1, the TV weights): the hardware's reciprocal square root (1 ulp) for t in the normal range — t >= eps2, so
  // the test is uniform; 1/sqrtf costs ~25 instructions per weight with its IEEE square root and division, which made the weights
  // pass of MMGKS VALU-bound (51 us for 12n bytes at 4096^2 against 36 us for the same pass without them)
  if (special == 2) return (eps2 >= 1e-30f) ? __builtin_amdgcn_rsqf(t) : 1.0f / sqrtf(t);
  return powf(t, e);
}

// Sum over a block of NT threads (NT multiple of 64).  `lds` holds NT/64 doubles.  Result valid in thread 0.
template <int NT>
__device__ __forceinline__ double block_sum(double v, double* lds) {
  constexpr int NW = NT / 64;
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();  // protect lds reuse between consecutive calls
  if (lane == 0) lds[wid] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < NW; ++w) t += lds[w];
  }
  return t;
}

// k_finalize's sum of output `o` (core.hip) by a workgroup of 256 threads: thread t adds the partials of blocks t, t + 256, ... on four
// accumulators, then block_sum.  One definition for the finalize kernel and for every consumer kernel that adds the partials up itself
// instead of waiting for a finalize launch (k_finalize_cgs, k_scale_fin): the same association, the same bits.  Valid in thread 0.
__device__ __forceinline__ double finalize_block_256(const double* __restrict__ p, int nblocks, int stride, double* lds) {
  double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
  int b = threadIdx.x;
  for (; b + 768 < nblocks; b += 1024) {
    v0 += p[(size_t)b * stride];
    v1 += p[(size_t)(b + 256) * stride];
    v2 += p[(size_t)(b + 512) * stride];
    v3 += p[(size_t)(b + 768) * stride];
  }
  for (; b < nblocks; b += 256) v0 += p[(size_t)b * stride];
  return block_sum<256>((v0 + v1) + (v2 + v3), lds);
}

// NV block sums with two barriers instead of 2 NV: every value is reduced within its wave, lane 0 of each wave parks its NV
// sums in LDS, and thread i < NV adds the NT/64 wave sums of value i — in wave order, the order of block_sum, so the bits are
// the same.  `lds` holds (NT/64) * NV doubles.  The result of value i is valid in thread i.
template <int NT, int NV>
__device__ __forceinline__ double block_sum_many_trees(const double (&v)[NV], double* lds) {
  constexpr int NW = NT / 64;
  static_assert(NV <= NT, "one thread per value");
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  double w[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) w[i] = wave_sum_trees(v[i]);
  __syncthreads();  // protect lds reuse between consecutive calls
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) lds[wid * NV + i] = w[i];
  }
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x < NV) {
#pragma unroll
    for (int q = 0; q < NW; ++q) t += lds[q * NV + threadIdx.x];
  }
  return t;
}

// block_sum_many without the LDS crossbar: the NV wave sums by a TRANSPOSING butterfly.  wave_sum's tree pairs lane l with l + 32, the
// results with l + 16, ... ; here the lanes of a pair split the values between them at every level — the lower one keeps the first half
// and receives the partner's first half, the upper one the second — so a level moves half as many values as the one before: NV - 1
// + log2(64 / NV) exchanged doubles per wave instead of 6 NV, through v_permlane32_swap / v_permlane16_swap (gfx950) and DPP row /
// quad permutations, no ds_bpermute (k_gemv_t2 on a 512^2 basis is ONE float4 per thread and row, then 16 sums per wave: 192
// ds_bpermute per wave, a CU's 16 waves queueing on one LDS crossbar).  Same pairs in the same order as wave_sum, and a + b = b + a
// exactly: the same bits as block_sum_many_trees (tools/microbench/wave_sum_many.hip: bit-identical, 9 x faster at 8, 16 and 32
// values with a CU full of reducing workgroups).  Any NV <= 32 (padded with zeros to 8, 16 or 32 — values of their own, added to
// nothing); the result of value i is valid in thread i.
template <int NV, int N, int OFF>
__device__ __forceinline__ void butterfly_level(double (&v)[NV], int lane) {
  if constexpr (N >= 2) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      if constexpr (OFF == 32) {
        swap_halves32(v[i], v[i + N / 2]);
        v[i] = v[i] + v[i + N / 2];
      } else if constexpr (OFF == 16) {
        swap_rows16(v[i], v[i + N / 2]);
        v[i] = v[i] + v[i + N / 2];
      } else {
        const bool up = (lane & OFF) != 0;
        const double keep = up ? v[i + N / 2] : v[i], give = up ? v[i] : v[i + N / 2];
        v[i] = keep + lane_xor_f64<OFF>(give);
      }
    }
  } else {
    v[0] = v[0] + lane_xor_f64<OFF>(v[0]);
  }
  if constexpr (OFF > 1) butterfly_level<NV, (N >= 2 ? N / 2 : 1), OFF / 2>(v, lane);
}
template <int NT, int NV>
__device__ __forceinline__ double block_sum_many(const double (&vin)[NV], double* lds) {
  static_assert(NV >= 1 && NV <= 32 && NV <= NT, "at most 32 values, one thread per value");
  constexpr int P = NV <= 8 ? 8 : (NV <= 16 ? 16 : 32);
  constexpr int NW = NT / 64, LPV = 64 / P;                      // lanes that end up with the same value
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  double v[P];
#pragma unroll
  for (int i = 0; i < P; ++i) v[i] = i < NV ? vin[i < NV ? i : 0] : 0.0;
  butterfly_level<P, P, 32>(v, lane);                            // v[0]: the wave sum of value lane / LPV
  __syncthreads();  // protect lds reuse between consecutive calls
  if ((lane & (LPV - 1)) == 0 && lane / LPV < NV) lds[wid * NV + lane / LPV] = v[0];
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x < NV) {
#pragma unroll
    for (int q = 0; q < NW; ++q) t += lds[q * NV + threadIdx.x];
  }
  return t;
}

// ------------------------------------------------------------------ a scalar that may still be block partials
// n == 0: the constant 1 ; n == 1: *p ; n > 1: the sum of n block partials.  The sum is always formed the same way
// (lane l of one wave adds p[l], p[l+64], ... then a wave64 tree), so every kernel that consumes the same partials
// gets the bitwise same value, and no separate finalize launch is needed between producer and consumer.
constexpr int64_t kNontemporalMinFloats = (int64_t)11 << 20;        // see stream_nontemporal()
constexpr int64_t kNontemporalLoadsMinFloats = (int64_t)20 << 20;

struct ScalarSrc {
  const double* p;
  int n;
};

__device__ __forceinline__ double wave_sum_all(double v) {   // total in every lane
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// called by ONE full wave (64 lanes); returns the value in every lane
__device__ __forceinline__ double scalar_from_wave(const ScalarSrc s, int lane) {
  if (s.n == 0) return 1.0;
  if (s.n == 1) return *s.p;
  // eight independent accumulators: the loads of a round are in flight together (the partials were written by the
  // previous kernel, possibly through another XCD's L2 — each one is a trip to the memory side)
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0;
  int i = lane;
  for (; i + 448 < s.n; i += 512) {
    a0 += s.p[i];
    a1 += s.p[i + 64];
    a2 += s.p[i + 128];
    a3 += s.p[i + 192];
    a4 += s.p[i + 256];
    a5 += s.p[i + 320];
    a6 += s.p[i + 384];
    a7 += s.p[i + 448];
  }
  if (i < s.n) a0 += s.p[i];
  if (i + 64 < s.n) a1 += s.p[i + 64];
  if (i + 128 < s.n) a2 += s.p[i + 128];
  if (i + 192 < s.n) a3 += s.p[i + 192];
  if (i + 256 < s.n) a4 += s.p[i + 256];
  if (i + 320 < s.n) a5 += s.p[i + 320];
  if (i + 384 < s.n) a6 += s.p[i + 384];
  double v = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  // fixed tree (wave_sum: lane l with l + 32, the results with l + 16, ...: the association does not depend on the caller), lane 0's
  // total to every lane through the scalar unit
  v = wave_sum(v);
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Cache hints for the streamed vectors of the large-image CGLS loop, as a mask: stores — bit 0 the blur kernel's output,
// bit 1 the new iterate x' (bits 2, 3: the residual and the direction: measured, no gain); loads in the [x, p] update —
// bit 4 the old iterate x, bit 5 t = A^T r (both read exactly once).  What a kernel finds in the 256 MB memory-side
// cache decides its speed, and everything that is not re-read soon only evicts what is:
//   below ~11.5 M floats the next kernel finds plain-stored data cached and hints lose (2048^2: 18.7 k vs 16.6 k
//   iterations/s with stores hinted; crossover between 3072^2, -1 %, and 3584^2, +2 %);
//   4096^2: 6.68 k (none) / 6.73 k (blur) / 6.81 k (x') / 6.87 k (both) / 6.84 k (all four stores); load hints +-0;
//   5120^2 (regrouped updates): 4.36 k (mask 3) -> 4.76 k (+x load) -> 4.77 k (+t load), the forward blur launch in the
//   loop 51 -> 36 us; 6144^2: 2.78 k -> 3.01 k (79 -> 50 us); 8192^2: 1.59 k -> 1.70 k; 4608^2: 5.60 k -> 5.71 k.
// Bits 6, 7: the basis rows read by k_gemv_t / k_gemv_n (k x 4n bytes streamed once per kernel; hinted, they stop evicting
// the vector every row tile re-reads): 4096^2 GKS 343 -> 372, Hybrid-GMRES 907 -> 1001, MMGKS 331 -> 342 iterations/s.
// TRK_NT=<mask> overrides (tuning).
// Bit 8 (TRK_REV=1, an experiment — DESIGN.md 4.1b): the two CGLS update kernels sweep their vectors from the END: each then meets
// the rows its producer (a blur launch, sweeping forward) wrote last first, while they can still be in the 256 MB memory-side cache,
// and leaves its own output so that the next blur launch meets ITS first rows last-written.
inline int stream_nontemporal(int64_t n) {
  static const int env = getenv("TRK_NT") ? atoi(getenv("TRK_NT")) : -1;
  static const int rev = (getenv("TRK_REV") && atoi(getenv("TRK_REV"))) ? 256 : 0;
  if (env >= 0) return env | rev;
  if (n >= kNontemporalLoadsMinFloats) return 51 | 192 | rev;
  return (n >= kNontemporalMinFloats ? (3 | 192) : 0) | rev;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace trk

// ------------------------------------------------------------------ kernel timer (hipEvent pairs)
struct trk_timer {
  hipEvent_t* ev;  // 2*cap events
  int cap, used;
};

namespace trk {
// record the start / stop event of the next pair if a timer is attached for this direction
struct TimerScope {
  trk_timer* t;
  hipStream_t s;
  TimerScope(trk_timer* timer, int which, int transpose, hipStream_t st) : t(nullptr), s(st) {
    if (timer && (which == 2 || which == transpose) && timer->used < timer->cap) {
      t = timer;
      (void)hipEventRecord(t->ev[2 * t->used], s);
    }
  }
  void stop() {
    if (t) {
      (void)hipEventRecord(t->ev[2 * t->used + 1], s);
      ++t->used;
      t = nullptr;
    }
  }
};
}  // namespace trk

struct trk_op;
struct PostReq;
namespace trk {
// ---- the float64 instrument (ref64.hip) ----
// radon2d.hip: one angle of a parallel-beam handle in float64 (q(d, tt) = (d - (nd-1)/2) inv + k0 + tt dq, weight w = scale / |cos|
// or / |sin|, mode 1 = marching columns) and the handle's sizes + fixed-point tables
struct RadonRefAngle {
  double inv, dq, k0, w;
  int mode, pad_;
};
struct RadonRefGeom {
  int N, nd, na, nt, npad;          // na = angles per frame, nt frames
  const RadonRefAngle* ang;         // nt * na, frame-major (device)
  const unsigned* A32;              // [nt*na][nd + 4]
  const unsigned* B32;              // [nt*na][npad]
  // emulation of fp32 partial sums (0: float64 sums): the forward adds its products in fp32 and moves the sum to a float64 total
  // every chunk_fwd marching steps, the adjoint every chunk_adj angles; the angle's weight is then applied in fp32 as well
  int chunk_fwd, chunk_adj;
};
bool radon_ref_geometry(trk_op* op, RadonRefGeom* g);
int radon_ref_apply_f32(const RadonRefGeom& g, int transpose, int weights, const float* x, float* y, hipStream_t s);
int ref_axpby_f32(int64_t n, Coef a, const float* x, Coef b, const float* z, float* out, double* sumsq, hipStream_t s);
// vecops.hip: k_lsqr_damped_update<T, ..> for T = float (elem_bytes 4) / double (8), no error partials
int lsqr_damped_update_any(size_t elem_bytes, const void* vk, void* w, const void* x_in, void* x_out, int64_t n, const double* alpha_sq,
                           const double* beta_next_sq, const double* beta0_sq, double damp, const double* state_in, double* state_out,
                           int first, hipStream_t s);
// vecops.hip: trk_gemv_t2 / trk_gemv_n without their finalize launch (the block partials stay in the stream's scratch: *part, *nblk),
// and the normalisation x / sqrt(sum of partials) that adds them up itself (k_finalize's order) and leaves the sum in *sum_out;
// post.on: workgroup 0 also carries a mailbox post (as the Golub-Kahan adjoint half step does)
int gemv_t2_partials(const float* V, int64_t ld, int k, int64_t n, const float* r, const float* r2, double** part, int* nblk, hipStream_t s);
int gemv_n_partials(const float* V, int64_t ld, int k, int64_t n, const double* y, double a, const float* base, double sc, float* out,
                    double** part, int* nblk, hipStream_t s);
int scale_by_partials(int64_t n, const double* part, int nblk, const float* x, float* out, double* sum_out, const PostReq& post,
                      hipStream_t s, const float* dotv = nullptr, double* dot_out = nullptr);   // dotv: also *dot_out = <out, dotv>, posted with the rest
// projected.hip: finalize of the 2k sums of gemv_t2_partials and trk_cgs_coeffs(G, ldg, W, W + k, k, passes, c) in one launch
int finalize_cgs(const double* part, int nblk, int k, double* W, double* G, int ldg, int passes, double* c, hipStream_t s);
// blur2d.hip: sizes and device pointers to the separable weights [kw row weights | kh column weights] of a blur handle
bool blur_separable_params(trk_op* op, int* nx, int* ny, int* kh, int* kw, const float** sep_fwd, const float** sep_adj);
}  // namespace trk

// One step of damped LSQR's short recurrence (trk_lsqr_damped_update) as a rider on another kernel's pixel pass: the vector vk of
// that update is the `z` operand of a Golub-Kahan adjoint half step (V[k] = a A^T u + b V[k-1]: z = V[k-1]), so the half step's
// epilogue can carry the update of the iterate that V[k-1] belongs to (trk_gk_step_lsqr).
struct LsqrReq {
  int on = 0;
  int first = 0;
  float* w = nullptr;
  const float* x_in = nullptr;
  float* x_out = nullptr;
  const float* ref = nullptr;
  double* err_part = nullptr;
  int err_cap = 0;
  const double* a2 = nullptr;        // alpha^2 of vk (final)
  const double* b2 = nullptr;        // beta_next^2 (may still be the operator's pending block partials)
  const double* beta0_sq = nullptr;
  double damp = 0.0;
  const double* st_in = nullptr;
  double* st_out = nullptr;
};

// A mailbox post (trk_mailbox_post / _post_sum) as a rider on another kernel: workgroup 0 of a Golub-Kahan adjoint half step copies
// the scalars the host is waiting for and publishes the slot's sequence number itself (trk_gk_step_post) — the norms of the step
// before are final exactly there (the adjoint's epilogue finishes the deferred one), so the post needs no launch of its own.
struct PostReq {
  int on = 0;
  const double* src = nullptr;        // device doubles to copy ...
  double* dst = nullptr;              // ... into pinned host memory (device-visible)
  int count = 0;
  const double* part = nullptr;       // optional: n_part block partials, their sum to *sum_dev and *sum_host
  int n_part = 0;
  double* sum_dev = nullptr;
  double* sum_host = nullptr;
  unsigned long long* seq = nullptr;  // the slot's sequence word (pinned host memory) and the number to publish
  unsigned long long value = 0;
};

// ------------------------------------------------------------------ operator handle
struct trk_op {
  int kind;  // 1 blur2d, 2 radon2d, 3 deriv2d, 4 spacetime, 5 blockdiag
  int64_t rows, cols;
  void* impl;
  int (*apply)(trk_op*, int transpose, const float* x, int64_t ldx, float* y, int64_t ldy, int batch,
               double* sumsq_dev, hipStream_t s);
  void (*destroy)(trk_op*);
  trk_timer* timer = nullptr;
  int timer_which = 0;
  // optional: y = Op(x1 + cb*x2) with the combined operand written to `comb`, cb = sign*S(num)/S(den); the sum of y^2
  // is left as raw block partials (count returned on the host) for the consumer kernel to add up
  int (*apply_fused)(trk_op*, int transpose, const float* x1, const float* x2, double sign, trk::ScalarSrc num,
                     trk::ScalarSrc den, float* comb, float* y, double* partials, int cap, int* n_partials,
                     hipStream_t s) = nullptr;
  int fused_caps = 1;    // what apply_fused can do (trk_op_fused_caps): 1 = everything, 2 = only x2 = NULL (raw block partials)
  // optional: out = a * Op(x) + b * z (+ ||out||^2) inside the operator's own output pass (trk_op_apply_axpby); hints: TRK_HINT_*
  int (*apply_axpby)(trk_op*, int transpose, const float* x, trk::Coef a, trk::Coef b, const float* z, float* out,
                     double* sumsq, int hints, hipStream_t s) = nullptr;
  // optional: the same combination for operators WITHOUT the hinted chains of apply_axpby (trk_op_axpby_caps stays 0: the solvers'
  // Golub-Kahan chains do not change): trk_op_apply_axpby takes it instead of apply + trk_axpby.  May return TRK_EUNSUPPORTED for
  // shapes / alignments its kernel does not take — the caller then falls back.
  int (*apply_axpby_plain)(trk_op*, int transpose, const float* x, trk::Coef a, trk::Coef b, const float* z, float* out,
                           double* sumsq, hipStream_t s) = nullptr;
  // optional: finish what a TRK_HINT_SUMSQ_DEFERRED apply left unfinished (trk_op_flush)
  int (*flush)(trk_op*, hipStream_t s) = nullptr;
  // set for the duration of one trk_gk_step_proj call: the forward half step's output pass also leaves the block partials of
  // <out, probe_vec> in probe_part and their count in probe_n (0: this operator's pass does not — the caller takes a separate dot)
  const float* probe_vec = nullptr;
  double* probe_part = nullptr;
  int probe_cap = 0, probe_n = 0;
  // set for the duration of one trk_gk_step_lsqr call: the adjoint half step's output pass carries this update if it can and sets
  // lsqr_blocks to the number of error partials it wrote (>= 1; 0: not taken — the caller runs trk_lsqr_damped_update itself)
  LsqrReq lsqr;
  int lsqr_blocks = 0;
  // likewise for one trk_gk_step_post call: post_taken = 1 when the adjoint half step's kernel carries the post
  PostReq post;
  int post_taken = 0;
  void* aux = nullptr;   // malloc'ed per-handle cache of a consumer (cgls_tiled.hip: tile geometry + weights); freed with the handle
};
