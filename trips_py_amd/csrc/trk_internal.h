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
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
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

// NV block sums with two barriers instead of 2 NV: every value is reduced within its wave, lane 0 of each wave parks its NV
// sums in LDS, and thread i < NV adds the NT/64 wave sums of value i — in wave order, the order of block_sum, so the bits are
// the same.  `lds` holds (NT/64) * NV doubles.  The result of value i is valid in thread i.
template <int NT, int NV>
__device__ __forceinline__ double block_sum_many(const double (&v)[NV], double* lds) {
  constexpr int NW = NT / 64;
  static_assert(NV <= NT, "one thread per value");
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  double w[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) w[i] = wave_sum(v[i]);
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
  // fixed tree: shfl_down order so that the association does not depend on the caller
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return __shfl(v, 0, 64);
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
