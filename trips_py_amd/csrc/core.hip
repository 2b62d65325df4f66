// core.hip — library plumbing: error strings, per-stream scratch, the fixed-order finalize kernel,
// operator-handle dispatch and the block-diagonal (frame-major) composite.
#include "trk_internal.h"

#include <algorithm>
#include <cstring>

#include <cstdlib>

#include <cstdarg>
#include <cstdio>
#include <map>
#include <mutex>
#include <vector>

namespace trk {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

// ---------------------------------------------------------------- scratch
struct Scratch {
  double* ptr = nullptr;
  size_t cap = 0;
};
static std::mutex g_mu;
static std::map<std::pair<int, hipStream_t>, Scratch> g_scratch;
static std::vector<double*> g_retired;  // outgrown buffers stay alive: work enqueued earlier may still use them

int scratch_doubles(hipStream_t s, size_t count, double** out) {
  int dev = 0;
  TRK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  Scratch& sc = g_scratch[{dev, s}];
  if (sc.cap < count) {
    size_t cap = sc.cap ? sc.cap : (size_t)1 << 16;
    while (cap < count) cap *= 2;
    double* p = nullptr;
    hipError_t e = hipMalloc(&p, cap * sizeof(double));
    if (e != hipSuccess) return fail(TRK_ENOMEM, "hipMalloc(%zu B) for reduction scratch: %s", cap * sizeof(double), hipGetErrorString(e));
    if (sc.ptr) g_retired.push_back(sc.ptr);
    sc.ptr = p;
    sc.cap = cap;
  }
  *out = sc.ptr;
  return TRK_OK;
}

static std::map<std::pair<int, hipStream_t>, unsigned*> g_tickets;
int stream_ticket(hipStream_t s, unsigned** out) {
  int dev = 0;
  TRK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  unsigned*& p = g_tickets[{dev, s}];
  if (!p) {
    TRK_HIP(hipMalloc(&p, 256));
    TRK_HIP(hipMemset(p, 0, 256));
    TRK_HIP(hipStreamSynchronize(nullptr));  // zero before anything that takes tickets can be enqueued (once per stream)
  }
  *out = p;
  return TRK_OK;
}

// ---------------------------------------------------------------- finalize: fixed-order sum of block partials
// One workgroup per output.  Four independent accumulators keep four loads in flight per thread; the association of
// the sum is a fixed function of (nblocks, thread id), hence bitwise reproducible run to run.
__global__ __launch_bounds__(256) void k_finalize(const double* __restrict__ partials, int nblocks, int stride,
                                                  double* __restrict__ out) {
  __shared__ double lds[4];
  const int o = blockIdx.x;
  const double v = finalize_block_256(partials + o, nblocks, stride, lds);
  if (threadIdx.x == 0) out[o] = v;
}

// the same sums with the outputs from index nsplit on stored at out2[o - nsplit] (a value that belongs somewhere else than behind
// its neighbours: the extra row of trk_gemv_t_x)
__global__ __launch_bounds__(256) void k_finalize_split(const double* __restrict__ partials, int nblocks, int stride,
                                                        double* __restrict__ out, int nsplit, double* __restrict__ out2) {
  __shared__ double lds[4];
  const int o = blockIdx.x;
  const double* __restrict__ p = partials + o;
  double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
  int b = threadIdx.x;
  for (; b + 768 < nblocks; b += 1024) {
    v0 += p[(size_t)b * stride];
    v1 += p[(size_t)(b + 256) * stride];
    v2 += p[(size_t)(b + 512) * stride];
    v3 += p[(size_t)(b + 768) * stride];
  }
  for (; b < nblocks; b += 256) v0 += p[(size_t)b * stride];
  double v = block_sum<256>((v0 + v1) + (v2 + v3), lds);
  if (threadIdx.x == 0) {
    if (o < nsplit) out[o] = v;
    else out2[o - nsplit] = v;
  }
}

int finalize_sums_split(const double* partials, int nblocks, int stride, int nout, double* out_dev, int nsplit, double* out2_dev,
                        hipStream_t s) {
  if (nout <= 0) return TRK_OK;
  hipLaunchKernelGGL(k_finalize_split, dim3(nout), dim3(256), 0, s, partials, nblocks, stride, out_dev, nsplit, out2_dev);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int finalize_sums(const double* partials, int nblocks, int stride, int nout, double* out_dev, hipStream_t s) {
  if (nout <= 0) return TRK_OK;
  hipLaunchKernelGGL(k_finalize, dim3(nout), dim3(256), 0, s, partials, nblocks, stride, out_dev);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

// out[batch*out_stride + o] = sum_b partials[batch*batch_stride + b*stride + o]
__global__ __launch_bounds__(256) void k_finalize_batched(const double* __restrict__ partials, int nblocks, int stride,
                                                          size_t batch_stride, int nvals, double* __restrict__ out,
                                                          int out_stride) {
  __shared__ double lds[4];
  const int o = blockIdx.x % nvals, bt = blockIdx.x / nvals;
  const double* __restrict__ p = partials + (size_t)bt * batch_stride + o;
  double v = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) v += p[(size_t)b * stride];
  v = block_sum<256>(v, lds);
  if (threadIdx.x == 0) out[(size_t)bt * out_stride + o] = v;
}

int cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// ---------------------------------------------------------------- block-diagonal composite (frames)
struct BlockDiagImpl {
  std::vector<trk_op*> ops;
  std::vector<int64_t> roff, coff;
  double* frame_sums = nullptr;  // one double per frame for the fused sum(y*y); owned by the handle
};

static int blockdiag_apply(trk_op* op, int transpose, const float* x, int64_t ldx, float* y, int64_t ldy, int batch,
                           double* sumsq_dev, hipStream_t s) {
  auto* im = static_cast<BlockDiagImpl*>(op->impl);
  const size_t nf = im->ops.size();
  // sum(y*y) over all frames: every frame writes its own double, one finalize adds them in frame order
  // (the handle-owned buffer makes the fused-norm form non-reentrant across streams for composites)
  double* parts = sumsq_dev ? im->frame_sums : nullptr;
  for (size_t f = 0; f < nf; ++f) {
    const int64_t xo = transpose ? im->roff[f] : im->coff[f];
    const int64_t yo = transpose ? im->coff[f] : im->roff[f];
    int rc = im->ops[f]->apply(im->ops[f], transpose, x + xo, ldx, y + yo, ldy, batch, sumsq_dev ? parts + f : nullptr, s);
    if (rc) return rc;
  }
  if (sumsq_dev) return finalize_sums(parts, (int)nf, 1, 1, sumsq_dev, s);
  return TRK_OK;
}

static void blockdiag_destroy(trk_op* op) {
  auto* im = static_cast<BlockDiagImpl*>(op->impl);
  if (im->frame_sums) (void)hipFree(im->frame_sums);
  delete im;
}

}  // namespace trk

using namespace trk;

extern "C" {

int trk_version(void) { return 100; /* 0.1.0 */ }

const char* trk_last_error(void) { return g_err; }

int trk_device_info(int* cu, int* wavefront, int64_t* lds_per_cu, int64_t* hbm_bytes) {
  int dev = 0;
  TRK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  TRK_HIP(hipGetDeviceProperties(&p, dev));
  if (cu) *cu = p.multiProcessorCount;
  if (wavefront) *wavefront = p.warpSize;
  if (lds_per_cu) *lds_per_cu = (int64_t)p.maxSharedMemoryPerMultiProcessor;
  if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
  return TRK_OK;
}

int trk_blockdiag_create(trk_op* const* ops, int count, trk_op** out) {
  TRK_REQUIRE(ops && out && count > 0, "trk_blockdiag_create: ops/out NULL or count <= 0");
  auto* im = new BlockDiagImpl();
  int64_t r = 0, c = 0;
  for (int i = 0; i < count; ++i) {
    if (!ops[i]) {
      delete im;
      return fail(TRK_EINVAL, "trk_blockdiag_create: ops[%d] is NULL", i);
    }
    im->ops.push_back(ops[i]);
    im->roff.push_back(r);
    im->coff.push_back(c);
    r += ops[i]->rows;
    c += ops[i]->cols;
  }
  if (hipMalloc(&im->frame_sums, sizeof(double) * (size_t)count) != hipSuccess) {
    delete im;
    return fail(TRK_ENOMEM, "trk_blockdiag_create: hipMalloc of %d frame sums failed", count);
  }
  auto* op = new trk_op{5, r, c, im, blockdiag_apply, blockdiag_destroy, nullptr, 0};
  *out = op;
  return TRK_OK;
}

int trk_op_shape(const trk_op* op, int64_t* rows, int64_t* cols) {
  TRK_REQUIRE(op, "trk_op_shape: op is NULL");
  if (rows) *rows = op->rows;
  if (cols) *cols = op->cols;
  return TRK_OK;
}

int trk_op_apply(trk_op* op, int transpose, const float* x, int64_t ldx, float* y, int64_t ldy, int batch,
                 double* sumsq_dev, trk_stream stream) {
  TRK_REQUIRE(op && x && y, "trk_op_apply: NULL argument");
  TRK_REQUIRE(batch >= 1, "trk_op_apply: batch must be >= 1 (got %d)", batch);
  const int64_t nin = transpose ? op->rows : op->cols, nout = transpose ? op->cols : op->rows;
  TRK_REQUIRE(batch == 1 || (ldx >= nin && ldy >= nout), "trk_op_apply: ldx/ldy smaller than the vector length");
  TRK_REQUIRE(x != y, "trk_op_apply: in-place apply is not supported");
  return op->apply(op, transpose ? 1 : 0, x, ldx, y, ldy, batch, sumsq_dev, (hipStream_t)stream);
}

int trk_op_axpby_caps(const trk_op* op, int* native) {
  TRK_REQUIRE(op && native, "trk_op_axpby_caps: NULL argument");
  *native = op->apply_axpby ? 1 : 0;
  return TRK_OK;
}

int trk_op_apply_axpby(trk_op* op, int transpose, const float* x, double ca, const double* a_num, const double* a_den,
                       int a_flags, double cb, const double* b_num, const double* b_den, int b_flags, const float* z,
                       float* out, double* sumsq, int hints, trk_stream stream) {
  TRK_REQUIRE(op && x && out, "trk_op_apply_axpby: NULL argument");
  TRK_REQUIRE(out != x && out != z, "trk_op_apply_axpby: out must not alias x or z");
  const int tr = transpose ? 1 : 0;
  if (op->apply_axpby)
    return op->apply_axpby(op, tr, x, Coef{ca, a_num, a_den, a_flags}, Coef{cb, b_num, b_den, b_flags}, z, out, sumsq, hints,
                           (hipStream_t)stream);
  if (op->apply_axpby_plain) {
    const int rc = op->apply_axpby_plain(op, tr, x, Coef{ca, a_num, a_den, a_flags}, Coef{cb, b_num, b_den, b_flags}, z, out, sumsq,
                                         (hipStream_t)stream);
    if (rc != TRK_EUNSUPPORTED) return rc;
  }
  // any operator: the plain apply into `out`, then the vector kernel in place
  const int64_t nin = tr ? op->rows : op->cols, nout = tr ? op->cols : op->rows;
  if (int rc = op->apply(op, tr, x, nin, out, nout, 1, nullptr, (hipStream_t)stream)) return rc;
  return trk_axpby(nout, ca, a_num, a_den, a_flags, out, cb, b_num, b_den, b_flags, z, out, sumsq, stream);
}

int trk_op_flush(trk_op* op, trk_stream stream) {
  TRK_REQUIRE(op, "trk_op_flush: NULL operator");
  return op->flush ? op->flush(op, (hipStream_t)stream) : TRK_OK;
}

int trk_op_fused_caps(const trk_op* op, int* can_fuse) {
  TRK_REQUIRE(op && can_fuse, "trk_op_fused_caps: NULL argument");
  *can_fuse = op->apply_fused ? op->fused_caps : 0;
  return TRK_OK;
}

int trk_op_apply_fused(trk_op* op, int transpose, const float* x1, const float* x2, double sign, const double* num,
                       int num_n, const double* den, int den_n, float* comb_out, float* y, double* ysq_partials,
                       int capacity, int* n_partials, trk_stream stream) {
  TRK_REQUIRE(op && x1 && y && ysq_partials && n_partials && (!x2 || comb_out), "trk_op_apply_fused: NULL argument");
  TRK_REQUIRE(num_n >= 0 && den_n >= 0 && (num_n == 0 || num) && (den_n == 0 || den), "trk_op_apply_fused: bad scalar source");
  if (!op->apply_fused) return fail(TRK_EUNSUPPORTED, "trk_op_apply_fused: this operator has no fused form");
  return op->apply_fused(op, transpose ? 1 : 0, x1, x2, sign, ScalarSrc{num, num_n}, ScalarSrc{den, den_n}, comb_out, y,
                         ysq_partials, capacity, n_partials, (hipStream_t)stream);
}

int trk_finalize_batched(const double* partials, int nblocks, int nvals, int batches, double* out, int out_stride,
                         trk_stream stream) {
  TRK_REQUIRE(partials && out && nblocks >= 1 && nvals >= 1 && batches >= 0, "trk_finalize_batched: bad argument");
  if (batches == 0) return TRK_OK;
  hipLaunchKernelGGL(k_finalize_batched, dim3(nvals * batches), dim3(256), 0, (hipStream_t)stream, partials, nblocks, nvals,
                     (size_t)nblocks * nvals, nvals, out, out_stride);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_timer_create(int capacity, trk_timer** out) {
  TRK_REQUIRE(out && capacity > 0 && capacity <= (1 << 20), "trk_timer_create: bad capacity");
  auto* t = new trk_timer{new hipEvent_t[2 * (size_t)capacity], capacity, 0};
  for (int i = 0; i < 2 * capacity; ++i) {
    hipError_t e = hipEventCreate(&t->ev[i]);
    if (e != hipSuccess) {
      for (int j = 0; j < i; ++j) (void)hipEventDestroy(t->ev[j]);
      delete[] t->ev;
      delete t;
      return fail(TRK_EHIP, "hipEventCreate: %s", hipGetErrorString(e));
    }
  }
  *out = t;
  return TRK_OK;
}

int trk_timer_reset(trk_timer* t) {
  TRK_REQUIRE(t, "trk_timer_reset: NULL");
  t->used = 0;
  return TRK_OK;
}

int trk_timer_read(trk_timer* t, float* ms_out, int max_out, int* count_out) {
  TRK_REQUIRE(t && ms_out && count_out, "trk_timer_read: NULL argument");
  const int n = t->used < max_out ? t->used : max_out;
  for (int i = 0; i < n; ++i) {
    TRK_HIP(hipEventSynchronize(t->ev[2 * i + 1]));
    TRK_HIP(hipEventElapsedTime(&ms_out[i], t->ev[2 * i], t->ev[2 * i + 1]));
  }
  *count_out = n;
  return TRK_OK;
}

int trk_timer_destroy(trk_timer* t) {
  if (!t) return TRK_OK;
  for (int i = 0; i < 2 * t->cap; ++i) (void)hipEventDestroy(t->ev[i]);
  delete[] t->ev;
  delete t;
  return TRK_OK;
}

int trk_op_set_timer(trk_op* op, trk_timer* t, int which) {
  TRK_REQUIRE(op, "trk_op_set_timer: op is NULL");
  TRK_REQUIRE(which >= 0 && which <= 2, "trk_op_set_timer: which must be 0, 1 or 2");
  op->timer = t;
  op->timer_which = which;
  return TRK_OK;
}

// ------------------------------------------------------------------ asynchronous scalar downloads
// A one-wave kernel copies the scalars into pinned, host-coherent memory and then publishes a sequence number there
// (system-scope fence in between); the host spins on that number.  Against hipMemcpyAsync + hipEventRecord on the compute
// stream this keeps the copy engine and its ~8 us of stream latency out of a 70 us iteration, and the wait is a load loop.
struct trk_mailbox {
  double* host;                        // pinned
  int n;
  unsigned long long* seq;             // pinned: the last sequence number published per slot
  unsigned long long* expect;          // host: the sequence number the newest post of a slot will publish
  hipStream_t* stream;                 // the stream that post went to (for the error path of wait)
  unsigned long long counter;
  int slots;
};

namespace {
__global__ void k_mailbox_post(const double* __restrict__ src, double* dst, int count, unsigned long long* seq,
                               unsigned long long value) {
  for (int i = threadIdx.x; i < count; i += blockDim.x) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(seq, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// the same with one more value in front of the publication: the sum of n block partials (one wave, the fixed order of
// trk_internal.h's ScalarSrc), stored on the device (*sum_dev) and in the mailbox (*sum_host)
__global__ void k_mailbox_post_sum(const double* __restrict__ src, double* dst, int count, const double* __restrict__ part, int n,
                                   double* sum_dev, double* sum_host, unsigned long long* seq, unsigned long long value) {
  for (int i = threadIdx.x; i < count; i += blockDim.x) dst[i] = src[i];
  const double t = scalar_from_wave(ScalarSrc{part, n}, threadIdx.x);      // launched with one wave
  if (threadIdx.x == 0) {
    *sum_dev = t;
    *sum_host = t;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(seq, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
}  // namespace

// ------------------------------------------------------------------ host doubles -> device, inside a launch
// The counterpart of the mailbox: up to 128 doubles per launch travel in the kernel's own arguments (read through the
// kernel-argument segment: the struct is the first parameter) and one wave stores them.  Stream-ordered, the host array is free
// when the call returns, no staging copy and no synchronisation — what hipMemcpy from pageable memory costs a solver loop that
// uploads a k-vector per iteration (GKS / MMGKS with automatic lambda: the projected solution).
namespace {
struct PutArg { double v[128]; };
__global__ void k_scalars_put(const PutArg a, double* __restrict__ dst, int count) {
  const auto* src = (const __attribute__((address_space(4))) double*)__builtin_amdgcn_kernarg_segment_ptr();
  for (int i = threadIdx.x; i < count; i += blockDim.x) dst[i] = src[i];
}
}  // namespace

int trk_scalars_put(double* dst_dev, const double* src_host, int count, trk_stream stream) {
  TRK_REQUIRE(count >= 0 && (count == 0 || (dst_dev && src_host)), "trk_scalars_put: bad argument");
  for (int i0 = 0; i0 < count; i0 += 128) {
    const int c = std::min(128, count - i0);
    PutArg a;
    memcpy(a.v, src_host + i0, sizeof(double) * (size_t)c);
    hipLaunchKernelGGL(k_scalars_put, dim3(1), dim3(64), 0, (hipStream_t)stream, a, dst_dev + i0, c);
  }
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_mailbox_create(int n_doubles, int slots, trk_mailbox** out) {
  TRK_REQUIRE(out && n_doubles > 0 && slots > 0 && slots <= 4096, "trk_mailbox_create: bad argument");
  auto* mb = new trk_mailbox{nullptr, n_doubles, nullptr, new unsigned long long[slots](), new hipStream_t[slots](), 0, slots};
  hipError_t e = hipHostMalloc((void**)&mb->host, sizeof(double) * (size_t)n_doubles, hipHostMallocCoherent | hipHostMallocMapped);
  if (e == hipSuccess) e = hipHostMalloc((void**)&mb->seq, sizeof(unsigned long long) * (size_t)slots, hipHostMallocCoherent | hipHostMallocMapped);
  if (e != hipSuccess) {
    trk_mailbox_destroy(mb);
    return fail(TRK_EHIP, "trk_mailbox_create: %s", hipGetErrorString(e));
  }
  memset(mb->host, 0, sizeof(double) * (size_t)n_doubles);
  memset(mb->seq, 0, sizeof(unsigned long long) * (size_t)slots);
  *out = mb;
  return TRK_OK;
}

int trk_mailbox_destroy(trk_mailbox* mb) {
  if (!mb) return TRK_OK;
  if (mb->host) (void)hipHostFree(mb->host);
  if (mb->seq) (void)hipHostFree(mb->seq);
  delete[] mb->expect;
  delete[] mb->stream;
  delete mb;
  return TRK_OK;
}

int trk_mailbox_host(trk_mailbox* mb, double** host_out) {
  TRK_REQUIRE(mb && host_out, "trk_mailbox_host: NULL argument");
  *host_out = mb->host;
  return TRK_OK;
}

int trk_mailbox_doubles(trk_mailbox* mb) { return mb ? mb->n : 0; }
int trk_mailbox_slots(trk_mailbox* mb) { return mb ? mb->slots : 0; }

int trk_mailbox_post(trk_mailbox* mb, int slot, const double* src_dev, int offset, int count, trk_stream stream) {
  TRK_REQUIRE(mb && src_dev && slot >= 0 && slot < mb->slots, "trk_mailbox_post: bad mailbox / slot");
  TRK_REQUIRE(offset >= 0 && count > 0 && offset + count <= mb->n, "trk_mailbox_post: range outside the mailbox");
  mb->expect[slot] = ++mb->counter;
  mb->stream[slot] = (hipStream_t)stream;
  hipLaunchKernelGGL(k_mailbox_post, dim3(1), dim3(64), 0, (hipStream_t)stream, src_dev, mb->host + offset, count, mb->seq + slot,
                     mb->expect[slot]);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_mailbox_post_sum(trk_mailbox* mb, int slot, const double* src_dev, int offset, int count, const double* partials,
                         int n_partials, double* sum_dev, int sum_offset, trk_stream stream) {
  TRK_REQUIRE(mb && src_dev && partials && sum_dev && slot >= 0 && slot < mb->slots, "trk_mailbox_post_sum: bad mailbox / slot / NULL argument");
  TRK_REQUIRE(offset >= 0 && count > 0 && offset + count <= mb->n, "trk_mailbox_post_sum: range outside the mailbox");
  TRK_REQUIRE(n_partials >= 1 && sum_offset >= 0 && sum_offset < mb->n && (sum_offset < offset || sum_offset >= offset + count),
              "trk_mailbox_post_sum: need n_partials >= 1 and a sum position inside the mailbox, outside the copied range");
  mb->expect[slot] = ++mb->counter;
  mb->stream[slot] = (hipStream_t)stream;
  hipLaunchKernelGGL(k_mailbox_post_sum, dim3(1), dim3(64), 0, (hipStream_t)stream, src_dev, mb->host + offset, count, partials,
                     n_partials, sum_dev, mb->host + sum_offset, mb->seq + slot, mb->expect[slot]);
  TRK_LAUNCH_CHECK();
  return TRK_OK;
}

int trk_mailbox_wait(trk_mailbox* mb, int slot) {
  TRK_REQUIRE(mb && slot >= 0 && slot < mb->slots, "trk_mailbox_wait: bad mailbox / slot");
  const unsigned long long want = mb->expect[slot];
  volatile unsigned long long* p = mb->seq + slot;
  for (unsigned long long spins = 0;; ++spins) {
    if (__atomic_load_n(p, __ATOMIC_ACQUIRE) >= want) return TRK_OK;
    if ((spins & 0xFFFFF) == 0xFFFFF) {                        // every ~1 M polls: is the stream still alive?
      const hipError_t e = hipStreamQuery(mb->stream[slot]);
      if (e != hipSuccess && e != hipErrorNotReady) return fail(TRK_EHIP, "trk_mailbox_wait: %s", hipGetErrorString(e));
      if (e == hipSuccess && __atomic_load_n(p, __ATOMIC_ACQUIRE) < want)
        return fail(TRK_EHIP, "trk_mailbox_wait: the stream drained without the post arriving");
    }
  }
}

// ------------------------------------------------------------------ one Golub-Kahan step on unnormalised vectors
int trk_gk_step(trk_op* op, int k, const float* u_k, const float* v_prev, float* v_k, float* u_next, double* AB, int chained,
                int defer_alpha, int defer_beta, trk_stream stream) {
  TRK_REQUIRE(op && u_k && v_k && u_next && AB && k >= 0 && (k == 0 || v_prev), "trk_gk_step: bad argument");
  double* bk2 = AB + 2 * k;            // ||U[k]||^2
  double* a2 = AB + 2 * k + 1;         // ||V[k]||^2, written by the first half step
  double* b2 = AB + 2 * k + 2;         // ||U[k+1]||^2, written by the second
  const int feeds = TRK_HINT_OUT_FEEDS_OPPOSITE, takes = TRK_HINT_INPUT_FROM_OPPOSITE, later = TRK_HINT_SUMSQ_DEFERRED;
  // V[k] = (1/beta_k) A^T U[k] - (beta_k/alpha_{k-1}) V[k-1]
  if (int rc = trk_op_apply_axpby(op, 1, u_k, 1.0, nullptr, bk2, TRK_SQRT_DEN, k == 0 ? 0.0 : -1.0, k == 0 ? nullptr : bk2,
                                  k == 0 ? nullptr : AB + 2 * k - 1, k == 0 ? 0 : (TRK_SQRT_NUM | TRK_SQRT_DEN),
                                  k == 0 ? nullptr : v_prev, v_k, a2, feeds | (chained ? takes : 0) | (defer_alpha ? later : 0),
                                  stream))
    return rc;
  // U[k+1] = (1/alpha_k) A V[k] - (alpha_k/beta_k) U[k]
  return trk_op_apply_axpby(op, 0, v_k, 1.0, nullptr, a2, TRK_SQRT_DEN, -1.0, a2, bk2, TRK_SQRT_NUM | TRK_SQRT_DEN, u_k, u_next, b2,
                            feeds | takes | (defer_beta ? later : 0), stream);
}

// ------------------------------------------------------------------ one Arnoldi step (decompositions.py:207-228), one rank
// w = A V[k-1]; the two Gram-Schmidt sweeps against V[0..k) by Gram matrix (one pair of passes over the basis: trk_gemv_t2 with the
// newest vector's Gram row riding along, trk_cgs_coeffs, trk_gemv_n with the fused ||.||^2); V[k] = the result, normalised.
// The five library calls the Python step made, enqueued by one: on the 512^2 blur the host was the bound (65 % device-busy).
// Five kernels (round 6; seven before): the finalize launches of the two reductions are gone — the 2k sums of the sweep are added up
// by the workgroups of the launch whose last arriver runs the k x k recurrence (k_finalize_cgs), the norm by every workgroup of the
// normalising pass (k_scale_fin); both in k_finalize's own association: the bits of the seven-kernel form (TRK_ARNOLDI_7=1 keeps it).
static int arnoldi_step_impl(trk_op* op, float* V, int64_t ld, int k, float* w, double* G, int ldg, double* W, double* S,
                             const PostReq& post, hipStream_t s, const float* dotv = nullptr, double* dot_out = nullptr) {
  const int64_t n = op->rows;
  const float* vk1 = V + (int64_t)(k - 1) * ld;
  float* vk = V + (int64_t)k * ld;
  static const bool seven = getenv("TRK_ARNOLDI_7") && atoi(getenv("TRK_ARNOLDI_7"));
  if (int rc = trk_op_apply(op, 0, vk1, n, w, n, 1, nullptr, s)) return rc;
  if (seven) {
    if (int rc = trk_gemv_t2(V, ld, k, n, w, vk1, W, s)) return rc;                   // h = V^T w | Gram row of V[k-1]
    if (int rc = trk_cgs_coeffs(G, ldg, W, W + k, k, 2, S + 1, s)) return rc;         // column k of H (without its last entry) at S[1..1+k)
    if (int rc = trk_gemv_n(V, ld, k, n, S + 1, 1.0, w, -1.0, vk, S, s)) return rc;   // V[k] = w - V c, S[0] = ||.||^2
    if (int rc = trk_axpby(n, 1.0, nullptr, S, TRK_SQRT_DEN, vk, 0.0, nullptr, nullptr, 0, nullptr, vk, nullptr, s)) return rc;
    if (dotv) {
      if (int rc = trk_dot(vk, dotv, n, dot_out, s)) return rc;
      if (post.on && post.sum_host) {                    // (the seven-kernel form posts the dot by a launch of its own, ahead of the step's)
        hipLaunchKernelGGL(k_mailbox_post, dim3(1), dim3(64), 0, s, (const double*)dot_out, post.sum_host, 1, post.seq, post.value - 1);
        TRK_LAUNCH_CHECK();
      }
    }
    if (post.on) {
      hipLaunchKernelGGL(k_mailbox_post, dim3(1), dim3(64), 0, s, post.src, post.dst, post.count, post.seq, post.value);
      TRK_LAUNCH_CHECK();
    }
    return TRK_OK;
  }
  double* part = nullptr;
  int nblk = 0;
  if (int rc = gemv_t2_partials(V, ld, k, n, w, vk1, &part, &nblk, s)) return rc;
  if (int rc = finalize_cgs(part, nblk, k, W, G, ldg, 2, S + 1, s)) return rc;
  if (int rc = gemv_n_partials(V, ld, k, n, S + 1, 1.0, w, -1.0, vk, &part, &nblk, s)) return rc;
  return scale_by_partials(n, part, nblk, vk, vk, S, post, s, dotv, dot_out);
}

int trk_arnoldi_step(trk_op* op, float* V, int64_t ld, int k, float* w, double* G, int ldg, double* W, double* S, trk_stream stream) {
  TRK_REQUIRE(op && V && w && G && W && S && k >= 1 && ldg >= k, "trk_arnoldi_step: bad argument");
  TRK_REQUIRE(op->rows == op->cols && ld >= op->rows, "trk_arnoldi_step: square operator, ld >= n");
  return arnoldi_step_impl(op, V, ld, k, w, G, ldg, W, S, PostReq{}, (hipStream_t)stream);
}

// the same step with the mailbox post of its scalars S[offset .. offset + count) riding on its last kernel (trk_mailbox_post's
// contract for `slot`: trk_mailbox_wait(mb, slot) returns when they have arrived)
int trk_arnoldi_step_post(trk_op* op, float* V, int64_t ld, int k, float* w, double* G, int ldg, double* W, double* S,
                          trk_mailbox* mb, int slot, int offset, int count, trk_stream stream) {
  return trk_arnoldi_step_post_at(op, V, ld, k, w, G, ldg, W, S, mb, slot, offset, count, offset, stream);
}

// ... to host[host_offset ..] (a caller that keeps two steps in flight gives each slot a region of its own)
int trk_arnoldi_step_post_at(trk_op* op, float* V, int64_t ld, int k, float* w, double* G, int ldg, double* W, double* S,
                             trk_mailbox* mb, int slot, int offset, int count, int host_offset, trk_stream stream) {
  return trk_arnoldi_step_post_dot(op, V, ld, k, w, G, ldg, W, S, mb, slot, offset, count, host_offset, nullptr, 0, stream);
}

// ... with S[dot_index] = <V[k], dotv> taken by the normalising pass and posted with the rest (dotv = NULL: no dot)
int trk_arnoldi_step_post_dot(trk_op* op, float* V, int64_t ld, int k, float* w, double* G, int ldg, double* W, double* S,
                              trk_mailbox* mb, int slot, int offset, int count, int host_offset, const float* dotv, int dot_index,
                              trk_stream stream) {
  TRK_REQUIRE(op && V && w && G && W && S && k >= 1 && ldg >= k, "trk_arnoldi_step_post: bad argument");
  TRK_REQUIRE(op->rows == op->cols && ld >= op->rows, "trk_arnoldi_step_post: square operator, ld >= n");
  TRK_REQUIRE(mb && slot >= 0 && slot < mb->slots, "trk_arnoldi_step_post: bad mailbox / slot");
  TRK_REQUIRE(offset >= 0 && count > 0 && host_offset >= 0 && host_offset + count <= mb->n, "trk_arnoldi_step_post: range outside the mailbox");
  mb->expect[slot] = ++mb->counter;
  mb->stream[slot] = (hipStream_t)stream;
  PostReq q;
  q.on = 1;
  q.src = S + offset;
  q.dst = mb->host + host_offset;
  q.count = count;
  q.seq = mb->seq + slot;
  q.value = mb->expect[slot];
  if (dotv) {
    // the dot lives OUTSIDE the step's scalar ranges on the device (S[1+k .. 1+2k) must stay zero for the steps to come) and lands
    // right behind the posted range on the host
    TRK_REQUIRE(dot_index >= 0 && host_offset + count + 1 <= mb->n, "trk_arnoldi_step_post_dot: no room behind the posted range for the dot");
    q.sum_dev = S + dot_index;
    q.sum_host = mb->host + host_offset + count;
  }
  return arnoldi_step_impl(op, V, ld, k, w, G, ldg, W, S, q, (hipStream_t)stream, dotv, dotv ? S + dot_index : nullptr);
}

int trk_gk_step_proj(trk_op* op, int k, const float* u_k, const float* v_prev, float* v_k, float* u_next, double* AB, int chained,
                     int defer_alpha, int defer_beta, const float* proj, double* partials, int cap, int* n_partials,
                     trk_stream stream) {
  TRK_REQUIRE(op && proj && partials && n_partials && cap >= 1, "trk_gk_step_proj: bad argument");
  op->probe_vec = proj;
  op->probe_part = partials;
  op->probe_cap = cap;
  op->probe_n = 0;
  const int rc = trk_gk_step(op, k, u_k, v_prev, v_k, u_next, AB, chained, defer_alpha, defer_beta, stream);
  const int n = op->probe_n;
  op->probe_vec = nullptr;
  op->probe_part = nullptr;
  op->probe_cap = op->probe_n = 0;
  if (rc) return rc;
  *n_partials = n;
  if (n == 0) {                       // this operator's output pass carries no dot: one finished value, taken separately
    *n_partials = 1;
    return trk_dot(u_next, proj, op->rows, partials, stream);
  }
  return TRK_OK;
}

int trk_gk_step_post(trk_op* op, int k, const float* u_k, const float* v_prev, float* v_k, float* u_next, double* AB, int chained,
                     int defer_alpha, int defer_beta, const float* proj, double* proj_partials, int proj_cap, int* n_proj,
                     trk_mailbox* mb, int slot, const double* src_dev, int offset, int count, const double* sum_partials,
                     int n_sum, double* sum_dev, int sum_offset, trk_stream stream) {
  TRK_REQUIRE(op && mb && src_dev && slot >= 0 && slot < mb->slots, "trk_gk_step_post: bad mailbox / slot / NULL argument");
  TRK_REQUIRE(offset >= 0 && count > 0 && count <= 8 && offset + count <= mb->n, "trk_gk_step_post: 1..8 doubles inside the mailbox");
  TRK_REQUIRE(!sum_partials || (n_sum >= 1 && sum_dev && sum_offset >= 0 && sum_offset < mb->n &&
                                (sum_offset < offset || sum_offset >= offset + count)),
              "trk_gk_step_post: the sum needs partials, a device place and a mailbox place outside the copied range");
  TRK_REQUIRE(!proj || (proj_partials && n_proj && proj_cap >= 1), "trk_gk_step_post: proj given but no room for its partials");
  mb->expect[slot] = ++mb->counter;
  mb->stream[slot] = (hipStream_t)stream;
  PostReq q;
  q.on = 1;
  q.src = src_dev;
  q.dst = mb->host + offset;
  q.count = count;
  q.part = sum_partials;
  q.n_part = sum_partials ? n_sum : 0;
  q.sum_dev = sum_dev;
  q.sum_host = sum_partials ? mb->host + sum_offset : nullptr;
  q.seq = mb->seq + slot;
  q.value = mb->expect[slot];
  op->post = q;
  op->post_taken = 0;
  if (proj) {
    op->probe_vec = proj;
    op->probe_part = proj_partials;
    op->probe_cap = proj_cap;
    op->probe_n = 0;
  }
  const int rc = trk_gk_step(op, k, u_k, v_prev, v_k, u_next, AB, chained, defer_alpha, defer_beta, stream);
  const int taken = op->post_taken, np = op->probe_n;
  op->post = PostReq{};
  op->post_taken = 0;
  op->probe_vec = nullptr;
  op->probe_part = nullptr;
  op->probe_cap = op->probe_n = 0;
  if (rc) return rc;
  if (proj) {
    *n_proj = np;
    if (np == 0) {
      *n_proj = 1;
      if (int rc2 = trk_dot(u_next, proj, op->rows, proj_partials, stream)) return rc2;
    }
  }
  if (!taken) {                        // no kernel of this operator carries posts: the post in its own launch, behind the step
    if (sum_partials)
      hipLaunchKernelGGL(k_mailbox_post_sum, dim3(1), dim3(64), 0, (hipStream_t)stream, src_dev, q.dst, count, sum_partials, n_sum,
                         sum_dev, q.sum_host, q.seq, q.value);
    else
      hipLaunchKernelGGL(k_mailbox_post, dim3(1), dim3(64), 0, (hipStream_t)stream, src_dev, q.dst, count, q.seq, q.value);
    TRK_LAUNCH_CHECK();
  }
  return TRK_OK;
}

int trk_gk_step_lsqr(trk_op* op, int k, const float* u_k, const float* v_prev, float* v_k, float* u_next, double* AB, int chained,
                     int defer_alpha, int defer_beta, float* w, const float* x_in, float* x_out, const float* ref,
                     double* err_partials, int capacity_blocks, int* n_blocks, double damp, const double* state_in,
                     double* state_out, trk_stream stream) {
  TRK_REQUIRE(op && k >= 1 && v_prev && w && x_out && state_out && AB, "trk_gk_step_lsqr: bad argument (k >= 1: the update is that of V[k-1])");
  TRK_REQUIRE(k == 1 || (state_in && x_in), "trk_gk_step_lsqr: steps after the first need state_in and x_in");
  TRK_REQUIRE(!ref || (err_partials && n_blocks), "trk_gk_step_lsqr: ref given but no room for the partials");
  LsqrReq q;
  q.on = 1;
  q.first = (k == 1);
  q.w = w;
  q.x_in = x_in;
  q.x_out = x_out;
  q.ref = ref;
  q.err_part = err_partials;
  q.err_cap = capacity_blocks;
  q.a2 = AB + 2 * k - 1;               // alpha_{k-1}^2 = ||V[k-1]||^2 (finished by the forward half of step k-1)
  q.b2 = AB + 2 * k;                   // beta_k^2 = ||U[k]||^2 (this adjoint's epilogue finishes it if it is still pending)
  q.beta0_sq = AB;
  q.damp = damp;
  q.st_in = state_in;
  q.st_out = state_out;
  op->lsqr = q;
  op->lsqr_blocks = 0;
  const int rc = trk_gk_step(op, k, u_k, v_prev, v_k, u_next, AB, chained, defer_alpha, defer_beta, stream);
  const int taken = op->lsqr_blocks;
  op->lsqr = LsqrReq{};
  op->lsqr_blocks = 0;
  if (rc) return rc;
  if (taken > 0) {
    if (n_blocks) *n_blocks = taken;
    return TRK_OK;
  }
  // the operator's adjoint pass did not take it (no native epilogue, too many tiles for the partial buffer): the same update in
  // its own launch, behind the step — the order the caller had before
  int nb = 0;
  const int rc2 = trk_lsqr_damped_update(v_prev, w, x_in, x_out, op->cols, ref, err_partials, capacity_blocks, &nb, q.a2, q.b2, AB, damp,
                                         state_in, state_out, k == 1 ? 1 : 0, stream);
  if (n_blocks) *n_blocks = nb;
  return rc2;
}

int trk_op_destroy(trk_op* op) {
  if (!op) return TRK_OK;
  if (op->destroy) op->destroy(op);
  if (op->aux) free(op->aux);
  delete op;
  return TRK_OK;
}

}  // extern "C"
