/*
 * trk.h — C ABI of libtrk.so, the MI355X (gfx950) engine behind TRIPs-Py's Krylov hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, `extern "C"`, no torch / numpy types.
 * TRIPs-Py has no FFI today (pure NumPy/SciPy, duck-typed operators); every entry point below
 * names the reference code (path:line under /root/reference) whose arithmetic it replaces, and
 * INTEGRATION.md shows the ctypes binding a TRIPs-Py maintainer would add.
 *
 * Conventions
 *   - every function returns int: 0 = TRK_OK, negative = error; `trk_last_error()` returns a
 *     thread-local description of the last failure on the calling thread.
 *   - all vectors are CALLER-OWNED DEVICE buffers of float (fp32 storage); every reduction is
 *     accumulated in double and delivered to a CALLER-OWNED DEVICE double.  The library never
 *     frees or retains user buffers past the call.  Its only allocations are the opaque operator
 *     handles (including, for the projectors, handle-owned work buffers: a transposed image copy,
 *     band partial sums, a padded sinogram copy) and a small per-stream scratch area for block
 *     partial sums.  Consequence: blur / derivative / CSR handles may be applied from several
 *     streams at once; ONE projector handle must be applied from one stream at a time.
 *   - all work is enqueued on the caller's `stream` (a hipStream_t passed as void*; 0 = default
 *     stream).  No entry point synchronises the device or the stream.
 *   - "basis" arguments are ROW-PER-VECTOR: vector j of a basis V is the contiguous run
 *     V[j*ld .. j*ld+n).  (The reference stores n x k column-major-by-NumPy-view matrices and
 *     re-copies them with hstack/column_stack every iteration: decompositions.py:243-254,
 *     GKS.py:91-96.)
 *   - a COEFFICIENT is given by four arguments (double c, const double* num, const double* den,
 *     int flags) and means  c * f(*num) / g(*den)  evaluated ON THE DEVICE when the kernel runs:
 *     num/den may be NULL (=1); TRK_SQRT_NUM / TRK_SQRT_DEN apply sqrt() to the loaded value.
 *     This is how alpha = ||v||, beta = gamma/delta ... stay on the GPU between kernels
 *     (CGLS.py:61-72, decompositions.py:235-242) without a host round trip.
 */
#ifndef TRK_H
#define TRK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRK_OK 0
#define TRK_EINVAL (-1)       /* bad argument (NULL, size, shape)            */
#define TRK_EHIP (-2)         /* a HIP runtime call failed                    */
#define TRK_ENOMEM (-3)       /* device / host allocation failed              */
#define TRK_EUNSUPPORTED (-4) /* valid request this build cannot serve        */
#define TRK_ENCCL (-5)        /* RCCL missing or an RCCL call failed          */

#define TRK_SQRT_NUM 1
#define TRK_SQRT_DEN 2

typedef struct trk_op trk_op; /* opaque linear operator (geometry immutable after create) */
typedef void* trk_stream;     /* hipStream_t */
typedef struct trk_comm trk_comm; /* opaque communicator of the sharded (frames over GPUs) path: one per process */

/* ---------------------------------------------------------------- library ------------- */
int trk_version(void);                 /* 10000*major + 100*minor + patch */
const char* trk_last_error(void);      /* thread-local, never NULL */
/* device facts the host sizes launches / rooflines with (CU count, LDS per CU, ...). */
int trk_device_info(int* cu_count, int* wavefront, int64_t* lds_per_cu, int64_t* hbm_bytes);

/* ---------------------------------------------------------------- operators ----------- */
/* 2-D blur, reflective boundary.  Replaces scipy.ndimage.convolve(X.reshape(nx,ny), PSF,
 * mode='reflect') (forward) and the same with flipud(fliplr(PSF)) (the reference's "transpose"),
 * trips/test_problems/Deblurring2D.py:66-73.  psf_host: kh x kw row-major doubles on the HOST
 * (copied).  Rank-1 PSFs (every Deblurring2D.Gauss PSF, :48-64) take a separable LDS-tiled path.
 * A kh x 1 PSF on an nx x 1 image is the 1-D blur of Deblurring1D.py:56-62. */
int trk_blur2d_create(const double* psf_host, int kh, int kw, int nx, int ny, trk_op** out);

/* Parallel-beam Radon transform, Joseph / linear-interpolation projector, matched adjoint.
 * Replaces astra.OpTomo over create_proj_geom('parallel', 1, N, theta) + 'linear' projector and
 * the /N scaling of trips/utilities/io.py:392-399.  Image N x N row-major; sinogram
 * (n_ang, n_det) row-major.  Parity: astra-toolbox is absent from the build image; the CONVENTION (rotation sense, detector
 * order, layouts) is pinned to the ASTRA outputs the reference holds as images, through the fan-beam operator's far-source
 * limit; the interpolation weights follow Joseph's published kernel (oracle/cpu_ref.py, tests/test_oracle_golden.py). */
int trk_radon2d_create(int N, int n_det, const double* angles_host, int n_ang, double scale, trk_op** out);
/* The same projector for a DYNAMIC problem: n_frames time frames of N x N, frame t seen under its own n_ang_per_frame
 * angles (angles_host frame-major).  Equals pylops.BlockDiag of the per-frame operators (io.py:391-420) but runs every
 * frame in one launch.  x: frame-major images; y: (frame, angle, detector). */
int trk_radon2d_dynamic_create(int N, int n_det, const double* angles_host, int n_frames, int n_ang_per_frame,
                               double scale, trk_op** out);

/* Fan-beam (flat detector) line projector and matched adjoint: astra 'fanflat' geometry + 'line_fanflat' projector of
 * trips/test_problems/Tomography.py:53-88 (p = int(sqrt(2) nx) detector pixels of pitch (SOD+ODD)/SOD, SOD = 3 nx,
 * ODD = nx).  Weights = ray / pixel intersection lengths.  Sinogram (n_ang, n_det) row-major.  Pinned to the two ASTRA outputs of
 * this geometry the reference holds as rendered images (demos/demo_Tomo_small_scale.ipynb:145,179; tests/golden/
 * fanbeam_demo_image.npz): correlation 0.9999 with the sinogram, 0.985 with the matrix image, mirrored conventions <= 0.95. */
int trk_fanbeam2d_create(int N, int n_det, double det_pitch, double source_origin, double origin_detector,
                         const double* angles_host, int n_ang, trk_op** out);

/* First-difference regularisers, matrix-free.  Replace the scipy.sparse matrices of
 * trips/utilities/operators.py:24-36 (2-D: rows x[i,j]-x[i,j+1] then x[i,j]-x[i+1,j]) and :39-45
 * (space-time: nt copies of the 2-D operator, then temporal rows x_t - x_{t+1}).
 * For a time-sharded problem a rank creates the operator over its LOCAL frames and passes the
 * first frame of the next rank via trk_spacetime_set_halo before each forward apply (and adds
 * the returned halo contribution after each transpose apply). */
int trk_deriv2d_create(int N, trk_op** out);
int trk_spacetime_create(int N, int nt_local, int has_next, int has_prev, trk_op** out);
/* forward: x_next = first frame of the next rank (N*N floats) ; transpose: y_prev = last
 * temporal block of the previous rank's output rows (N*N floats).  Either may be NULL. */
int trk_spacetime_set_halo(trk_op* op, const float* x_next_dev, const float* y_prev_dev);

/* General sparse operator from host CSR arrays of A (nrows x ncols) and of A^T (built by the caller, e.g. scipy's
 * `.T.tocsr()`); all six arrays are copied to the device.  Serves operators the reference holds as scipy.sparse matrices:
 * the derivative regularisers (operators.py:24-45), the framelet analysis operators (:50-113) and the precomputed
 * forward matrices of the real dynamic data sets sliced per frame (io.py:132-135,197-229).  trk_op_apply with batch = k > 1 (`A @ V`
 * on an (n, k) block: GKS.py:37, MMGKS.py:44) reads the matrix once per pass of 8 / 4 / 2 columns; every column equals its own
 * single-vector apply to the bit. */
int trk_csr_create(int64_t nrows, int64_t ncols, int64_t nnz, const int64_t* indptr_host, const int* indices_host,
                   const float* values_host, const int64_t* t_indptr_host, const int* t_indices_host,
                   const float* t_values_host, trk_op** out);

/* Block-diagonal operator over frames (pylops.BlockDiag at io.py:420; sparse slicing :223-225).
 * The handle borrows `ops` (they must outlive it). x and y are frame-major. */
int trk_blockdiag_create(trk_op* const* ops, int count, trk_op** out);

int trk_op_shape(const trk_op* op, int64_t* rows, int64_t* cols);
/* y = A x (transpose=0) or y = A^T y (transpose=1) for `batch` vectors, vector b at x + b*ldx,
 * y + b*ldy.  If sumsq_dev != NULL it additionally receives sum(y*y) over all batch outputs,
 * fp64-accumulated (the ||w||^2, ||t||^2 of CGLS.py:61,69-70 fused into the producing kernel). */
int trk_op_apply(trk_op* op, int transpose, const float* x_dev, int64_t ldx, float* y_dev, int64_t ldy,
                 int batch, double* sumsq_dev, trk_stream stream);
int trk_op_destroy(trk_op* op);

/* ---------------------------------------------------------------- kernel timing -------- */
/* Measurement aid (bench.py's roofline leg): a timer holds `capacity` hipEvent pairs.  While a timer is attached to
 * an operator, every apply in the selected direction (0 forward, 1 transpose, 2 both) records one pair tightly around
 * its MAIN kernel, on the stream of the apply (not around the reduction finalize).  trk_timer_read synchronises on
 * the recorded events and returns the elapsed milliseconds of each pair.  Detach with timer = NULL. */
typedef struct trk_timer trk_timer;
int trk_timer_create(int capacity, trk_timer** out);
int trk_timer_reset(trk_timer* t);
int trk_timer_read(trk_timer* t, float* ms_out, int max_out, int* count_out);
int trk_timer_destroy(trk_timer* t);
int trk_op_set_timer(trk_op* op, trk_timer* t, int which);

/* ---------------------------------------------------------------- reductions ---------- */
/* out = sum x*y ; out = sum x*x ; out = sum (x-y)^2      (np.dot / np.linalg.norm call sites:
 * CGLS.py:49-50,61,70,73,76,79; decompositions.py:92,97,178,182,217,225,238,241; GKS.py:89-90) */
int trk_dot(const float* x, const float* y, int64_t n, double* out_dev, trk_stream stream);
int trk_nrm2sq(const float* x, int64_t n, double* out_dev, trk_stream stream);
int trk_diff_nrm2sq(const float* x, const float* y, int64_t n, double* out_dev, trk_stream stream);

/* ---------------------------------------------------------------- axpy family --------- */
/* out = A*x + B*y, A and B device-evaluated coefficients; y may be NULL (then out = A*x);
 * out may alias x or y.  If sumsq_dev != NULL it receives sum(out*out).
 * (CGLS.py:65,67,72; decompositions.py:94,177,181,218,237-242) */
int trk_axpby(int64_t n, double ca, const double* a_num, const double* a_den, int a_flags, const float* x,
              double cb, const double* b_num, const double* b_den, int b_flags, const float* y, float* out,
              double* sumsq_dev, trk_stream stream);
/* out = A*x and *dot_dev = <out, z> in the same pass (out may alias x): the new basis vector v = r / ||r|| (MMGKS.py:121-123,
 * GKS.py:92-96) together with c_j = v_j . (A^T b), the entry of the projected right-hand side it adds — optional fused path,
 * the same results as trk_axpby followed by trk_dot. */
int trk_scale_dot(int64_t n, double ca, const double* a_num, const double* a_den, int a_flags, const float* x, float* out,
                  const float* z, double* dot_dev, trk_stream stream);
/* out = x * y (element-wise; MMGKS.py:114,116 `wf * (...)`, `wr * (...)`). */
int trk_mul(int64_t n, const float* x, const float* y, float* out, trk_stream stream);
/* out = w * (x - y) in one pass (the weighted residual `wf * (AV@y - b)` of MMGKS.py:114); out may alias any input. */
int trk_mul_diff(int64_t n, const float* w, const float* x, const float* y, float* out, trk_stream stream);
/* out = (v*v + eps*eps)^(p/2 - 1) with v = x - y (y may be NULL): the smoothed-Holder MM weights of
 * trips/utilities/weights.py:66-68 applied to the residual A x - b (MMGKS.py:56-57) or to L x (:60,93). */
int trk_mm_weights(int64_t n, const float* x, const float* y, double eps, double p, float* out, trk_stream stream);
/* Group-sparsity weights of MMGKS (MMGKS.py:86-90): out[c*groups + i] = (sum_{t < group_len} d[i*group_len + t]^2 + add)^expo,
 * c < copies — one weight per group of consecutive entries, tiled `copies` times. */
int trk_group_weights(const float* d, int64_t groups, int group_len, double add, double expo, int copies, float* out,
                      trk_stream stream);
/* Isotropic-TV weights of MMGKS (MMGKS.py:61-77 = trips/utilities/weights.py:29-40).  x: the flat iterate, viewed as
 * X[N][N][nt] (t fastest: `x.reshape(nx**2, nt)`, :71); g1, g2: the centered first derivatives of operators_old.py:22-45
 * (pylops.FirstDerivative, zero first/last row) along j and i.  out[idx] = out[N*N*nt + idx] =
 * (g1^2 + g2^2 + eps^2)^((q-2)/4) (:75-76), then out[2*N*N*nt + k] = (u_tail[k]^2 + eps^2)^((q-2)/4) for the n_tail temporal
 * rows of L x (:77).  out: 2*N*N*nt + n_tail floats. */
int trk_isotv_weights(const float* x, int N, int nt, const float* u_tail, int64_t n_tail, double eps, double q, float* out,
                      trk_stream stream);

/* Fused forms of the first-difference regularisers — L from trk_deriv2d_create, or from trk_spacetime_create on a rank that owns
 * the whole time axis (no halos; weights laid out as the rows of L: per frame the 2N(N-1) spatial rows, then the temporal rows) —
 * for the re-weighted solvers; nothing of the length of L x is written and read back:
 *   trk_tv_weights:  w = ((L x)^2 + eps^2)^(q/2-1)                 replaces  L @ x, then the weights of MMGKS.py:60,93
 *   trk_tv_grad:     out = r_in + lam * L^T (w .* (L x))           replaces  MMGKS.py:116-118 (w .* (L x), L^T, r + lam*rb)
 * w == NULL: unit weights (lam * L^T L x, the regularisation term of the GKS residual, GKS.py:81-84); r_in == NULL: 0.
 * out must not alias x or r_in.  Products and sums are rounded as in the separate kernels. */
int trk_tv_weights(trk_op* L, const float* x, double eps, double q, float* w, trk_stream stream);
/* Time-sharded space-time operator (trk_spacetime_create with has_next / has_prev): the temporal rows x_t - x_{t+1}
 * (trips/utilities/operators.py:39-45) couple a rank's first / last frame with ONE frame of each neighbour rank.  Give the NEXT
 * fused call on this handle (trk_tv_weights / trk_tv_grad / trk_tv_grad_dot) its operand's neighbour frames — the previous rank's
 * LAST frame and the next rank's FIRST frame of the same vector, N*N floats each, NULL where there is no neighbour — and the rank
 * forms its own pixels of L^T (w .* L x) completely with the same kernel one rank runs: one two-sided exchange of x
 * (trk_halo_exchange2) per operand instead of one exchange of rows of L x per direction of L.  The frames must stay valid until
 * that call has completed on its stream; the call consumes them on every exit path (a call that fails its argument checks
 * disarms them too).  The handle holds ONE armed operand: not for concurrent fused calls from two threads / streams.  Weight layout
 * of a sharded handle (trk_tv_weights out,
 * trk_tv_grad in): nt_local * 2N(N-1) spatial | rows 0 .. nt_local-2 | row nt_local-1 (has_next) | the previous rank's boundary
 * row (has_prev: recomputed from the halo, the same bits as on the rank that owns it), N*N floats each. */
int trk_tv_halo(trk_op* L, const float* x_prev_last_dev, const float* x_next_first_dev);
int trk_tv_grad(trk_op* L, const float* x, const float* w, const float* r_in, double lam, float* out, trk_stream stream);
/* trk_tv_grad that also leaves *dot_out = <out, dotv> (dotv: a vector of the image's length): GKS needs r . L^T L r next to
 * z = L^T L r for the Gram row of the next basis vector (GKS.py:92-96 through the Gram form) — no pass over r and z of its own. */
int trk_tv_grad_dot(trk_op* L, const float* x, const float* w, const float* r_in, double lam, float* out, const float* dotv,
                    double* dot_out, trk_stream stream);
/* The same with *xsq_out = <x, x> over the rank's own pixels from the same pass (GKS's one-pass form, trk_gemv_orth_iterate: x = dotv = r,
 * and r . r is what trk_cgs_coeffs_rho turns into the norm of the next basis vector — no pass over r of its own). */
int trk_tv_grad_dot_xsq(trk_op* L, const float* x, const float* w, const float* r_in, double lam, float* out, const float* dotv,
                        double* dot_out, double* xsq_out, trk_stream stream);

/* One Golub-Kahan half step in the operator's own output pass (decompositions.py:240-252: v = A^T u - beta v_old, u = A v -
 * alpha u_old, their norms):   out = a * Op(x) + b * z ,  *sumsq = ||out||^2 (if sumsq != NULL)
 * with the coefficients of trk_axpby (a = ca * [sqrt] *a_num / [sqrt] *a_den, NULL pointers = 1; z = NULL: out = a * Op(x)).
 * Operators with a native form (trk_op_axpby_caps: the Radon projector — the band reduction of the forward and the tile
 * gather of the adjoint carry the epilogue and leave the norm as block partials) need no vector kernel; for every other
 * operator this is trk_op_apply into `out` followed by trk_axpby in place: same result, any operator.
 * out must not alias x or z.  hints (0 is always right):
 *   TRK_HINT_OUT_FEEDS_OPPOSITE  the caller promises that the NEXT apply of this operator is the opposite direction, takes
 *                                `out` as its input, and that `out` is not modified in between: the operator may leave behind
 *                                what that apply derives from its input first (the adjoint's detector records, the forward's
 *                                transposed image copy);
 *   TRK_HINT_INPUT_FROM_OPPOSITE x is such an `out`, unmodified: what was left behind may be used (checked against x);
 *   TRK_HINT_SUMSQ_DEFERRED      *sumsq may stay unfinished (block partials inside the operator) until the next
 *                                trk_op_apply_axpby of this operator that carries TRK_HINT_INPUT_FROM_OPPOSITE — whose kernel
 *                                adds the partials for its own coefficients and stores the finished value — or until
 *                                trk_op_flush; every other call on the operator finishes it first.  The caller promises
 *                                not to read *sumsq (host or device) before one of these. */
#define TRK_HINT_OUT_FEEDS_OPPOSITE 1
#define TRK_HINT_INPUT_FROM_OPPOSITE 2
#define TRK_HINT_SUMSQ_DEFERRED 4
int trk_op_flush(trk_op* op, trk_stream stream);
int trk_op_axpby_caps(const trk_op* op, int* native);
int trk_op_apply_axpby(trk_op* op, int transpose, const float* x, double ca, const double* a_num, const double* a_den,
                       int a_flags, double cb, const double* b_num, const double* b_den, int b_flags, const float* z,
                       float* out, double* sumsq, int hints, trk_stream stream);

/* Asynchronous downloads of device scalars into one pinned host block ("mailbox"): trk_mailbox_post enqueues the copy of
 * `count` doubles to host[offset ..] on `stream` and marks `slot` behind it; trk_mailbox_wait blocks until the work posted
 * under that slot has completed (re-using a slot before waiting on it only makes the wait later, never earlier).  What the
 * hybrid solvers use to fetch alpha_k, beta_{k+1} (the entries of B_k, Hybrid_LSQR.py:69-75) while later steps are already
 * running.  The copy is a one-wave kernel writing host-coherent memory and publishing a sequence number the host polls: no
 * copy-engine operation and no event on the compute stream. */
/* The other direction: `count` host doubles to device memory, stream-ordered, carried by the launch's own arguments (128 per
 * launch) — src_host is free when the call returns; no staging copy, no synchronisation. */
int trk_scalars_put(double* dst_dev, const double* src_host, int count, trk_stream stream);
typedef struct trk_mailbox trk_mailbox;
int trk_mailbox_create(int n_doubles, int slots, trk_mailbox** out);
int trk_mailbox_destroy(trk_mailbox* mb);
int trk_mailbox_host(trk_mailbox* mb, double** host_out);
int trk_mailbox_doubles(trk_mailbox* mb);   /* the sizes it was created with */
int trk_mailbox_slots(trk_mailbox* mb);
int trk_mailbox_post(trk_mailbox* mb, int slot, const double* src_dev, int offset, int count, trk_stream stream);
int trk_mailbox_wait(trk_mailbox* mb, int slot);
/* trk_mailbox_post plus one more value published with it: the sum of `n_partials` device doubles (block partials of an inner
 * product some kernel left behind: trk_gk_step_proj), stored at *sum_dev and at host[sum_offset] (outside [offset, offset+count)).
 * One launch where a reduction launch and a second post stood. */
int trk_mailbox_post_sum(trk_mailbox* mb, int slot, const double* src_dev, int offset, int count, const double* partials,
                         int n_partials, double* sum_dev, int sum_offset, trk_stream stream);

/* One whole Golub-Kahan step (decompositions.py:230-255) on UNNORMALISED vectors, two trk_op_apply_axpby calls:
 *   v_k    = (1/beta_k)  A^T u_k - (beta_k/alpha_{k-1}) v_prev ,  AB[2k+1] = ||v_k||^2    = alpha_k^2
 *   u_next = (1/alpha_k) A v_k   - (alpha_k/beta_k)     u_k    ,  AB[2k+2] = ||u_next||^2 = beta_{k+1}^2
 * with u_k = beta_k u_k(normalised), v_prev = alpha_{k-1} v_{k-1} (NULL for k = 0), AB[0] = ||b||^2, AB[2j+1] = alpha_j^2,
 * AB[2j+2] = beta_{j+1}^2 (device doubles).  chained: u_k came out of the previous trk_gk_step on this operator, untouched
 * (TRK_HINT_INPUT_FROM_OPPOSITE for the first half step); defer_alpha / defer_beta: TRK_HINT_SUMSQ_DEFERRED for the two norms
 * (single rank: the next chained apply finishes them). */
int trk_gk_step(trk_op* op, int k, const float* u_k, const float* v_prev, float* v_k, float* u_next, double* AB, int chained,
                int defer_alpha, int defer_beta, trk_stream stream);
/* One Arnoldi step on one rank (decompositions.py:207-228; Hybrid_GMRES.py:61-63, GMRES.py): w = A V[k-1], orthogonalised against
 * V[0..k) by two Gram-Schmidt sweeps in their Gram-matrix form (trk_gemv_t2 / trk_cgs_coeffs / trk_gemv_n), V[k] = the normalised
 * result.  V: rows of ld floats (row k is written), w: n floats of scratch, G: the basis' Gram matrix so far (ldg x ldg doubles, rows
 * 0..k-2 installed; row k-1 is installed here), W: 2k doubles of scratch, S: S[0] = h_{k+1,k}^2, S[1..1+k) = column k of H above it.
 * Optional: the same five calls in one — same results bit for bit, in five kernels where the five calls launch seven (the two
 * finalize launches are folded into their consumers; TRK_ARNOLDI_7=1 in the environment keeps the seven). */
int trk_arnoldi_step(trk_op* op, float* V, int64_t ld, int k, float* w, double* G, int ldg, double* W, double* S, trk_stream stream);
/* The same step whose last kernel also posts S[offset .. offset + count) to host[offset ..] of `mb` (trk_mailbox_post's contract:
 * trk_mailbox_wait(mb, slot) returns once they have arrived) — no launch for the post. */
int trk_arnoldi_step_post(trk_op* op, float* V, int64_t ld, int k, float* w, double* G, int ldg, double* W, double* S,
                          trk_mailbox* mb, int slot, int offset, int count, trk_stream stream);
/* ... to host[host_offset ..] instead (two steps in flight: a region of the mailbox per slot). */
int trk_arnoldi_step_post_at(trk_op* op, float* V, int64_t ld, int k, float* w, double* G, int ldg, double* W, double* S,
                             trk_mailbox* mb, int slot, int offset, int count, int host_offset, trk_stream stream);
/* ... and S[dot_index] = <V[k], dotv> formed by the step's normalising pass — a place OUTSIDE [0, 1 + 2 kmax): S[1+k .. 1+2k) must stay zero
 * for the steps to come — which reaches the host right behind the posted range, host[host_offset + count]: Hybrid-GMRES with the
 * discrepancy principle wants V_{k+1}^T b, one new entry per step (Hybrid_GMRES.py:64-67 through discrepancy_principle.py:58). */
int trk_arnoldi_step_post_dot(trk_op* op, float* V, int64_t ld, int k, float* w, double* G, int ldg, double* W, double* S,
                              trk_mailbox* mb, int slot, int offset, int count, int host_offset, const float* dotv, int dot_index,
                              trk_stream stream);
/* Hybrid-GMRES with regparam = 'gcv' (Hybrid_GMRES.py:46-80): the host side of one iteration in one call.  The handle keeps H (host,
 * column-major), runs the Arnoldi steps ahead on `stream` (trk_arnoldi_step_post into a mailbox of its own), hands iterate k's projected
 * problem to a worker thread (trk_host_worker_post_hess_gcv; `workers`: n_workers of them, set_lapack done, and `mb`, 2 slots and
 * 2 (2 capacity + 4) doubles, all the caller's for the life of the handle: nothing in the Arnoldi process waits for a projected
 * solution, so consecutive iterates' O(k^3) jobs run side by side) and launches x_k = V_k y_k (trk_gemv_n_hosty) when the job is
 * collected, n_workers calls later.  create: the arguments of
 * trk_arnoldi_step with V[0] = b / ||b|| in place, beta0 = ||b||, at most `capacity` steps; start: enqueues step 1.  iter (projected.hip):
 * absorb (wait for the oldest posted step, install its column of H), enqueue_next, x_done != NULL (collect the OLDEST posted job:
 * *done_ii, its lambda, the reference's relResidual (:80); x_done launched, + ||x - ref||^2 block partials, *done_blocks of them, when
 * ref != NULL), post_job (iterate = columns - 1; needs a free worker).  H: a view of the columns installed so far, H[i + j * ldh],
 * for iterates the caller solves itself (the first ones; the SVD route). */
typedef struct trk_hgmres trk_hgmres;
typedef struct trk_host_worker trk_host_worker;
int trk_hgmres_create(trk_op* op, float* V, int64_t ld, int capacity, float* w, double* G, int ldg, double* W, double* S,
                      trk_mailbox* mb, trk_host_worker* const* workers, int n_workers, double beta0, trk_stream stream, trk_hgmres** out);
int trk_hgmres_destroy(trk_hgmres* g);
int trk_hgmres_start(trk_hgmres* g);
int trk_hgmres_iter(trk_hgmres* g, int absorb, int enqueue_next, int post_job, float* x_done, const float* ref, double* err_partials,
                    int err_cap, int* done_ii, double* done_lam, double* done_resid, int* done_blocks);
int trk_hgmres_hessenberg(trk_hgmres* g, double** H, int* ldh, int* columns);
/* A numeric regparam: the jobs posted from now on solve with this lambda instead of searching (lam < 0: gcv again). */
int trk_hgmres_fixed_lambda(trk_hgmres* g, double lam);
/* regparam = 'dp': before trk_hgmres_start.  bvec: the right-hand side on the device (every step's normalising pass then takes
 * <V[k], b>: trk_arnoldi_step_post_dot), bproj0 = <V[0], b>; the jobs are trk_host_worker_post_hess_dp(…, target, extra).  A collected
 * job that set no positive lambda (the reference's unassigned / not-reachable-yet branches) is handed back: *done_blocks = -1, nothing
 * launched — the caller solves that iterate itself from H and *bproj_out (V_{k+1}^T b so far, the handle's array). */
int trk_hgmres_dp(trk_hgmres* g, const float* bvec, double bproj0, double target, double extra, double** bproj_out);
/* host seconds spent so far: waiting for steps | enqueueing steps | waiting for workers | posting jobs | launching x = V y */
int trk_hgmres_stats(trk_hgmres* g, double* seconds5);
/* trk_gk_step (optionally with the projection of trk_gk_step_proj: proj != NULL) that also carries a mailbox post — the copy of
 * `count` (<= 8) device doubles src_dev[0..count) to host[offset ..] of `mb`, optionally the sum of n_sum block partials to *sum_dev and
 * host[sum_offset], and the publication of `slot` (trk_mailbox_post / trk_mailbox_post_sum): on the projector the first workgroup of
 * the adjoint half step does it, where the norms of the PREVIOUS step are final (the deferred one is finished by that very
 * kernel) — the hybrid solvers' per-step download of B_k's new entries (Hybrid_LSQR.py:69-75) without a launch of its own.  Other
 * operators: the step, then the post in its own launch.  trk_mailbox_wait(mb, slot) as usual. */
int trk_gk_step_post(trk_op* op, int k, const float* u_k, const float* v_prev, float* v_k, float* u_next, double* AB, int chained,
                     int defer_alpha, int defer_beta, const float* proj, double* proj_partials, int proj_cap, int* n_proj,
                     trk_mailbox* mb, int slot, const double* src_dev, int offset, int count, const double* sum_partials,
                     int n_sum, double* sum_dev, int sum_offset, trk_stream stream);
/* trk_gk_step for k >= 1 that also advances damped LSQR's iterate by the step belonging to v_prev = V[k-1] (the arguments of
 * trk_lsqr_damped_update with vk = v_prev, alpha_sq = AB[2k-1], beta_next_sq = AB[2k], beta0_sq = AB[0], first = (k == 1)): on the
 * projector the update rides the adjoint half step's own pixel pass — v_prev is that pass's second operand — so a fixed-lambda
 * Hybrid-LSQR iteration (Hybrid_LSQR.py:69-110) is three launches; any other operator gets the step followed by
 * trk_lsqr_damped_update.  *n_blocks: error partials written (ref != NULL).  One trk_gk_step* call at a time per handle. */
int trk_gk_step_lsqr(trk_op* op, int k, const float* u_k, const float* v_prev, float* v_k, float* u_next, double* AB, int chained,
                     int defer_alpha, int defer_beta, float* w, const float* x_in, float* x_out, const float* ref,
                     double* err_partials, int capacity_blocks, int* n_blocks, double damp, const double* state_in,
                     double* state_out, trk_stream stream);
/* trk_gk_step that also leaves <u_next, proj> (proj: op rows floats — the discrepancy principle's U^T b, one new row per step,
 * discrepancy_principle.py:58) as *n_partials (<= cap) block partials in `partials`: the projector's band reduction forms them
 * next to the norm it already carries (no pass over u_next, no reduction launch; trk_mailbox_post_sum adds them up on their
 * way to the host); operators without such a pass get one finished value from trk_dot (*n_partials = 1).  Like the deferred norms of
 * trk_op_apply_axpby, the request travels in the handle for the duration of the call: one trk_gk_step* call at a time per handle. */
int trk_gk_step_proj(trk_op* op, int k, const float* u_k, const float* v_prev, float* v_k, float* u_next, double* AB, int chained,
                     int defer_alpha, int defer_beta, const float* proj, double* partials, int cap, int* n_partials,
                     trk_stream stream);

/* One fused CGLS vector update (CGLS.py:64-67):  step = *gamma / *delta ;
 *   x_new = x + step*p ; r = r - step*w ;  sums_dev[0] = ||x_new||^2, sums_dev[1] = ||step*p||^2
 *   (= ||x_new - x_old||^2, :76), sums_dev[2] = ||x_new - x_true||^2 if x_true != NULL (:79).
 * x_new may alias x (in place) or be the next slot of an on-device history (xHistory, :66). */
int trk_cgls_update_xr(int64_t n, int64_t m, const double* gamma_dev, const double* delta_dev, const float* x,
                       const float* p, float* x_new, float* r, const float* w, const float* x_true,
                       double* sums_dev, trk_stream stream);

/* The same update with the three norms left as *n_blocks x 3 raw block partials (summed once after the solve with
 * trk_finalize_batched) — saves one reduction-finalize launch per iteration when nothing needs the norms on the fly
 * (tol = 0, single rank). */
int trk_cgls_update_xr_deferred(int64_t n, int64_t m, const double* gamma_dev, const double* delta_dev, const float* x,
                                const float* p, float* x_new, float* r, const float* w, const float* x_true,
                                double* norm_partials, int capacity_blocks, int* n_blocks, trk_stream stream);
/* The same with gamma / delta given as scalar sources (a finished scalar, n = 1, or n block partials of the kernel that
 * produced them — see the fused fast path below); block 0 stores the finished delta to *publish_delta (may be NULL). */
int trk_cgls_update_xr_src(int64_t n, int64_t m, const double* gamma, int gamma_n, const double* delta, int delta_n,
                           const float* x, const float* p, float* x_new, float* r, const float* w, const float* x_true,
                           double* publish_delta, double* norm_partials, int capacity_blocks, int* n_blocks,
                           trk_stream stream);
/* p = t + (S(gamma_new) / *gamma_old) p  (CGLS.py:72; same arithmetic as trk_axpby) with gamma_new a scalar source; block 0
 * stores the finished gamma_new to *publish_gamma (may be NULL). */
int trk_cgls_p_update(int64_t n, const float* t, float* p, const double* gamma_new, int gamma_new_n, const double* gamma_old,
                      double* publish_gamma, trk_stream stream);
/* The two halves of the update regrouped so that p is read once: r -= (*gamma_old / S(delta)) w (block 0 publishes delta),
 * and, after t = A^T r: x_new = x + (*gamma_old / *delta) p together with p = t + (S(gamma_new) / *gamma_old) p (block 0
 * publishes gamma_new; norms as trk_cgls_update_xr_deferred). */
int trk_cgls_update_grouping(int64_t n);   /* 1: trk_cgls_iterate uses the r / xp grouping for vectors of n floats (measured rule) */
int trk_cgls_r_update(int64_t m, const double* gamma_old, const double* delta, int delta_n, float* r, const float* w,
                      double* publish_delta, trk_stream stream);
int trk_cgls_xp_update(int64_t n, const double* gamma_old, const double* delta, const double* gamma_new, int gamma_new_n,
                       const float* x, float* p, const float* t, float* x_new, const float* x_true, double* publish_gamma,
                       double* norm_partials, int capacity_blocks, int* n_blocks, trk_stream stream);

/* ---------------------------------------------------------------- fused CGLS fast path --- */
/* For operators whose kernel can combine two inputs on load (the blur): one CGLS iteration becomes three launches with no
 * reduction-finalize launches in between.  A "scalar source" (const double* p, int n) is the constant 1 (n = 0), a
 * finished device scalar (n = 1) or n block partials still to be added (summed by the CONSUMER, always in the same order).
 *   y = Op(x1 + cb*x2), cb = sign * S(num)/S(den); the combined operand is also written to comb_out (must not alias
 *   x1/x2); sum(y*y) is left as *n_partials raw block partials in ysq_partials (capacity given).
 *     forward : x1 = t, x2 = p_old, cb = +gamma_k/gamma_{k-1}  -> comb = p_new (CGLS.py:72), y = w = A p_new (:60), ||w||^2 (:61)
 *     adjoint : x1 = r_old, x2 = w, cb = -gamma/delta          -> comb = r_new (:67),       y = t = A^T r_new (:68), ||t||^2 (:70)
 *   x2 = NULL: y = Op(x1) by the plain one-operand kernel (sign, num, den, comb_out ignored), sum(y*y) still left raw.
 * trk_op_fused_caps: *can_fuse = 0 (no fused apply), 1 (both forms: the separable blur), 2 (only x2 = NULL: the Radon
 * projector, whose band reduction / tile gather leave the partials; enough for the four-launch CGLS iteration). */
int trk_op_fused_caps(const trk_op* op, int* can_fuse);
int trk_op_apply_fused(trk_op* op, int transpose, const float* x1, const float* x2, double sign, const double* num,
                       int num_n, const double* den, int den_n, float* comb_out, float* y, double* ysq_partials,
                       int capacity, int* n_partials, trk_stream stream);
/* x_new = x + (S(gamma)/S(delta)) p (CGLS.py:64-65); block 0 stores the two finished scalars to publish_* (may be NULL);
 * [||x_new||^2, ||step*p||^2, ||x_new - x_true||^2] are left as *n_blocks x 3 raw partials (:76-80). */
int trk_cgls_x_update(int64_t n, const double* gamma, int gamma_n, const double* delta, int delta_n, const float* x,
                      const float* p, float* x_new, const float* x_true, double* publish_delta, double* publish_gamma,
                      double* norm_partials, int capacity_blocks, int* n_blocks, trk_stream stream);
/* The projected Tikhonov problem of the Golub-Kahan hybrid solvers on the device (Hybrid_LSQR.py:104, GK_Tikhonov.py:60):
 *   y = argmin || B_k y - beta0 e1 ||^2 + mu^2 || y ||^2 ,  B_k lower bidiagonal (k+1) x k with diagonal alpha_j and
 *   sub-diagonal beta_{j+1}, given as the SQUARED norms the Golub-Kahan kernels leave in device doubles:
 *   alpha_j^2 = alpha_sq[j*alpha_stride], beta_{j+1}^2 = beta_sq[j*beta_stride] (j < k), beta0^2 = *beta0_sq.
 *   mu = sqrt(lam) of the reference's stacked system.  Writes y[0..k).  1 <= k <= 2048.
 *   y_over_alpha != 0: y_j / alpha_j is written instead — the coefficients of x = V y with respect to basis vectors
 *   stored un-normalised (alpha_j v_j), as the Golub-Kahan recurrence produces them before its division.
 *   work (may be NULL): >= 3 (k_max + 1) + 4 device doubles, zero-initialised by the caller, kept between calls: when mu is
 *   unchanged and columns were only appended since the last call (the fixed-lambda hybrid iteration) only the new
 *   columns are rotated; anything else restarts from the first column. */
int trk_bidiag_tikhonov(const double* alpha_sq, int64_t alpha_stride, const double* beta_sq, int64_t beta_stride, int k,
                        double mu, const double* beta0_sq, double* y, int y_over_alpha, double* work, int work_doubles,
                        trk_stream stream);
/* The same solve on the HOST (float64, same recurrence) for callers that hold B_k there: alpha[k], beta_sub[k] (= B[j+1, j]),
 * beta0 = ||b||.  Pairs with trk_gemv_n_hosty, which takes the k coefficients from host memory. */
int trk_host_bidiag_tikhonov(const double* alpha, const double* beta_sub, int k, double beta0, double mu, int y_over_alpha, double* y);

/* HOST function (no device work, no stream): lambda = argmin over [x1, x2] of the GCV function of a diagonalised
 * projected problem,  G(lam) = sum_i ((1 - f_i) rhs_i)^2 / (m_eff - sum_i f_i)^2,  f_i = s_i^2 / (s_i^2 + lam),
 * by the bounded Brent search of scipy.optimize.fminbound restated step for step — what
 * trips/utilities/reg_param/gcv.py:94-95 runs every iteration (objective :25-78; x1 = 1e-9, x2 = 1e2, xtol = 1e-12,
 * maxfun = 1000 there).  s, rhs: k host doubles.  fval_out / nfev_out may be NULL. */
int trk_host_gcv_fminbound(const double* s, const double* rhs, int k, double m_eff, double x1, double x2, double xatol,
                           int maxfun, double* lam_out, double* fval_out, int* nfev_out);
/* HOST function: the same search for the hybrid solvers' bidiagonal projected problem (Hybrid_LSQR.py:81-84: svd(B_k), then GCV
 * on (S, U^T bhat)) WITHOUT the SVD: B_k = lower bidiagonal (k+1) x k with diagonal alpha[0..k) and sub-diagonal beta[0..k),
 * bhat = beta0 e1.  G(lam) is evaluated through one LDL^T of the k x k tridiagonal R R^T + lam I per lambda (B = Q [R; 0]):
 * the same function of lam, O(k) per evaluation, no O(k^2) SVD per iteration. */
int trk_host_gcv_bidiag(const double* alpha, const double* beta, int k, double beta0, double m_eff, double x1, double x2,
                        double xatol, int maxfun, double* lam_out, double* fval_out, int* nfev_out);
/* HOST function: the Newton iteration of the discrepancy principle (discrepancy_principle.py:80-99, 'tikhonov'):
 * solves || bhat / (sv*beta + 1) ||^2 + extra = target for beta = 1/alpha from beta = 1e-8 with the reference's
 * stopping rule.  sv: squared singular values padded with zeros to n, bhat: U^T b (n host doubles).  *alpha_set = 0
 * when the reference would return its unassigned value (converged at the very first step). */
int trk_host_dp_newton(const double* sv, const double* bhat, int n, double target, double extra, double* alpha_out,
                       int* alpha_set, int* iters_out);

/* HOST function: the same Newton iteration for the hybrid solvers' bidiagonal projected problem (B_k as in trk_host_gcv_bidiag,
 * bproj = U^T b, k+1 host doubles) without the SVD of B_k: one LDL^T of a k x k tridiagonal matrix and two solves per step.
 * *alpha_out = 0 with *alpha_set = 1 when the discrepancy cannot be reached yet (testzero >= 0, discrepancy_principle.py:71-76;
 * *testzero_out, if not NULL, receives that quantity); *alpha_set = 0: the reference's unassigned value. */
int trk_host_dp_bidiag(const double* alpha, const double* beta_sub, int k, const double* bproj, double target, double extra,
                       double* alpha_out, int* alpha_set, int* iters_out, double* testzero_out);

/* HOST: a worker thread of the library for the two searches above, so that choosing lambda_k overlaps the host's own
 * enqueueing (Hybrid_LSQR.py:80-100: at 512^2 the GCV search is 40 % of the host's time per iteration).  One job at a time:
 * post copies its inputs and returns at once; collect blocks until the posted job has finished and returns its lambda
 * (*have = 0: the discrepancy principle's "unassigned" case, as trk_host_dp_bidiag's alpha_set) and its return code. */
typedef struct trk_host_worker trk_host_worker;
int trk_host_worker_create(trk_host_worker** out);
int trk_host_worker_destroy(trk_host_worker* w);
int trk_host_worker_post_gcv_bidiag(trk_host_worker* w, const double* alpha, const double* beta, int k, double beta0,
                                    double m_eff, double x1, double x2, double xatol, int maxfun);
int trk_host_worker_post_dp_bidiag(trk_host_worker* w, const double* alpha, const double* beta_sub, int k, const double* bproj,
                                   double target, double extra);
int trk_host_worker_collect(trk_host_worker* w, double* lam_out, int* have_out);
/* HOST function: the projected problem of GKS / MMGKS with regparam = 'gcv' — the reference's default — from the Gram data of the
 * projected operators, in one call (GKS.py:54-74, MMGKS.py:94-106): R_A, R_L = the Cholesky factors of G_A = (AV)^T AV, G_L = (LV)^T LV
 * (k x k, row stride ldg; what the economic QRs of AV, LV give up to row signs), Q_A^T b = R_A^-T c, lambda by GCV on (R_A, R_L) reduced
 * to (diag(s), I) through M = R_A R_L^-1 (gcv.py:25-95: 'standard' form, m_eff = k) — s and U^T rhs of M's SVD by bidiagonalisation
 * and a bidiagonal SVD that rotates the one vector, no singular vectors formed — y = (G_A + lam G_L)^-1 c by a Cholesky factorisation of
 * the sum (the normal equations of the stacked problem [R_A; sqrt(lam) R_L] y = [Q_A^T b; 0] the reference hands to lstsq; R_A, R_L being
 * Cholesky factors of the Gram matrices, both see the same conditioning; the stacked problem by pivoted QR where the sum does not
 * factor, or with TRK_GRAM_GCV_LSTSQ set).  c_select / c_solve: the right-hand side the selector sees and the one the solve uses (MMGKS
 * hands the weighted and the unweighted one, MMGKS.py:97-106; GKS the same array twice).  lapack: {dpotrf, dtrtrs, dgebrd, dormbr,
 * dbdsqr, dgelsy}, the caller's LAPACK as plain C pointers (Fortran calling convention).  *ok_out = 0: a factorisation failed
 * (semi-definite Gram matrix, singular R_L, no convergence) and nothing was written: the caller's own branches take over. */
int trk_host_gram_gcv(void* const* lapack, const double* GA, const double* GL, int ldg, const double* c_select, const double* c_solve,
                      int k, double m_eff, double* lam_out, double* y_out, int* ok_out);
/* Hybrid-LSQR with automatic lambda (Hybrid_LSQR.py:80-110), the host's turn of an iteration in one call: collect the search posted by
 * the call before (k_done > 0: the step it belongs to), post the search for step k_post (mode 0: trk_host_worker_post_gcv_bidiag with
 * m_eff, the reference's bounds and tolerance; mode 1: trk_host_worker_post_dp_bidiag with target, extra and bproj), and with x_out != NULL
 * solve step k_done's projected problem with the collected lambda (trk_host_bidiag_tikhonov, y over alpha) and launch x_out = V y
 * (trk_gemv_n_hosty; ref, err_partials, err_cap, n_blocks as there).  alphas / betas / bproj: host arrays of B_k's entries, read before the
 * call returns.  *have_out = 0 when nothing was collected or the search set no lambda (the caller's in-line branches). */
int trk_hlsqr_select(trk_host_worker* w, int mode, const double* alphas, const double* betas, int k_post, double beta0,
                     double m_eff_or_target, const double* bproj, double extra, int k_done, const float* V, int64_t ld, int64_t n,
                     float* x_out, const float* ref, double* err_partials, int err_cap, int* n_blocks, double* lam_out, int* have_out,
                     trk_stream stream);
/* Hybrid-GMRES (Hybrid_GMRES.py:54-80): the WHOLE projected problem of iterate k as one job.  H: the (k+1) x k Arnoldi Hessenberg
 * matrix on the host, element (i, j) at H[i * h_row_stride + j * h_col_stride] (copied at post); beta0 = ||b||.  The job
 * bidiagonalises [beta0 e1 | H] with the CALLER's LAPACK (trk_host_worker_set_lapack: pointers to dgebrd and dormbr with the C
 * signatures of scipy.linalg.cython_lapack — every argument by reference, no hidden string lengths; libtrk links no LAPACK),
 * chooses lambda by 'standard' GCV on the bidiagonal form (trk_host_gcv_bidiag with m_eff; gcv.py:94-95), solves the Tikhonov
 * problem there (trk_host_bidiag_tikhonov) and maps back: collect_vec returns lambda, y (k doubles) and the reference's
 * relResidual of that iterate (:80, the Frobenius norm of its broadcast). */
int trk_host_worker_set_lapack(trk_host_worker* w, void* dgebrd, void* dormbr);
int trk_host_worker_post_hess_gcv(trk_host_worker* w, const double* H, int64_t h_row_stride, int64_t h_col_stride, int k,
                                  double beta0, double m_eff, double x1, double x2, double xatol, int maxfun);
/* ... and with the discrepancy principle instead of GCV (discrepancy_principle.py:68-99 on the same bidiagonal form; bproj = V_{k+1}^T b,
 * k + 1 host doubles, copied at post; target = (eta delta)^2, extra as trk_host_dp_bidiag): y and the residual are formed only when
 * the Newton iteration returns a positive lambda (*have_out = 1 and *lam_out > 0); otherwise the caller takes the reference's other
 * branches itself. */
/* ... with a lambda the caller names (no search): y, the residual and *lam_out = lam through trk_host_worker_collect_vec. */
int trk_host_worker_post_hess_fixed(trk_host_worker* w, const double* H, int64_t h_row_stride, int64_t h_col_stride, int k,
                                    double beta0, double lam);
int trk_host_worker_post_hess_dp(trk_host_worker* w, const double* H, int64_t h_row_stride, int64_t h_col_stride, int k,
                                 double beta0, const double* bproj, double target, double extra);
int trk_host_worker_collect_vec(trk_host_worker* w, double* lam_out, int* have_out, double* y, int k, double* resid_out);

/* n_iters consecutive CGLS iterations (numbers k_first .. k_first + n_iters - 1, 1-based) enqueued by one call: the loop
 * body of trips/solvers/CGLS.py:56-80 with tol = 0, i.e. nothing is read back between iterations.  Same kernels, scalar
 * layout and results as calling trk_op_apply / trk_cgls_update_xr_deferred / trk_op_apply / trk_axpby per iteration:
 *   S[0] = gamma_0 = ||t_0||^2 (set up by the caller, with r, t, p = t);  S[5k .. 5k+4] = [delta_k, gamma_k, (norms)]
 *   X: iterate slots, row stride x_ld; iteration k writes slot k-1 (keep_history) or (k-1) & 1;  x_prev = x_{k_first-1}
 *   NP: >= 3 * np_capacity_blocks * (iterations so far) doubles of norm partials (trk_finalize_batched sums them);
 *   *n_np_inout: partial blocks per iteration (0 before the first iteration; constant afterwards).
 *   PG, PD (may be NULL): `pcap` doubles each.  Given them and an operator with a fused apply (trk_op_fused_caps), the
 *   operator leaves ||t||^2 / ||w||^2 as raw block partials there (trk_op_apply_fused with x2 = NULL) and the consumers
 *   (trk_cgls_update_xr_src, trk_cgls_p_update) add them up: four launches per iteration instead of six.
 *   grouping: 0 = [x, r] / [p] updates, 1 = [r] / [x, p] (trk_cgls_r_update, trk_cgls_xp_update), -1 = by size
 *   (trk_cgls_update_grouping); same results either way. */
int trk_cgls_iterate(trk_op* A, int k_first, int n_iters, float* p, float* r, float* t, float* w, float* X, int64_t x_ld,
                     int keep_history, const float* x_prev, const float* x_true, double* S, double* NP,
                     int np_capacity_blocks, int* n_np_inout, double* PG, double* PD, int pcap, int grouping,
                     trk_stream stream);
/* The same for operators with a fused apply (trk_op_fused_caps): three launches per iteration.  P, R: ping-pong pairs
 * [2][p_ld], [2][r_ld] (iteration k reads index (k-1) & 1, writes k & 1); PG / PD: gamma / delta block partials with
 * `pcap` doubles each; *n_g_inout: number of valid gamma partials in PG (set by the caller's r0/t0 setup). */
int trk_cgls_iterate_fused(trk_op* A, int k_first, int n_iters, float* P, int64_t p_ld, float* R, int64_t r_ld, float* t,
                           float* w, float* X, int64_t x_ld, int keep_history, const float* x_prev, const float* x_true,
                           double* S, double* PG, double* PD, int pcap, double* NP, int np_capacity_blocks,
                           int* n_g_inout, int* n_np_inout, trk_stream stream);

/* out[b*out_stride + v] = sum_j partials[(b*nblocks + j)*nvals + v]  for b < batches, v < nvals (fixed order). */
int trk_finalize_batched(const double* partials, int nblocks, int nvals, int batches, double* out, int out_stride,
                         trk_stream stream);

/* ---------------------------------------------------------------- tall-skinny basis ops */
/* h[j] = sum_i w2[i] * V[j][i] * r[i], j < k  (w2 may be NULL = 1).  One pass over V.
 * (V^T r of the (re)orthogonalisation: decompositions.py:90-94,216-218; GKS.py:86-88; MMGKS.py:119-120;
 *  one row of a weighted Gram matrix.) */
int trk_gemv_t(const float* V, int64_t ld, int k, int64_t n, const float* r, const float* w2, double* h_dev,
               trk_stream stream);
/* One fused step of repeated classical Gram-Schmidt (GKS.py:86-88, MMGKS.py:119-120; Arnoldi with reorthogonalisation):
 *   w_out = w_in - sum_j h[j] V[j]   and   g[j] = sum_i V[j][i] w_out[i]   (j < k <= 16)   with ONE pass over V.
 * w_out may alias w_in; h and g are k device doubles (g must not alias h). */
int trk_gemv_nt(const float* V, int64_t ld, int k, int64_t n, const double* h_dev, const float* w_in, float* w_out,
                double* g_dev, trk_stream stream);
/* trk_gemv_t with one more row that is not part of the basis: h[j] = V[j] . r (j < k) and *h_x = xrow . r from the same pass
 * (GKS.py:55: the projected right-hand side c_j = (A v_j) . b next to the Gram row (A V)^T (A v_j) — no dot launch of its own). */
int trk_gemv_t_x(const float* V, int64_t ld, int k, int64_t n, const float* r, const float* xrow, double* h, double* h_x,
                 trk_stream stream);
/* Two inner-product sets in one sweep over the basis: h2k[j] = V[j] . r, h2k[k + j] = V[j] . r2, j < k (local sums).
 * Serves the Gram-matrix form of the repeated Gram-Schmidt sweeps (trk_cgs_coeffs): r = the direction to orthogonalise,
 * r2 = the vector appended last (its Gram row). */
int trk_gemv_t2(const float* V, int64_t ld, int k, int64_t n, const float* r, const float* r2, double* h2k_dev,
                trk_stream stream);
/* Coefficients of `passes` classical Gram-Schmidt sweeps at once (GKS.py:86-88, MMGKS.py:119-120, decompositions.py:216-218):
 * r - V (V^T r) repeated `passes` times equals r - V c with c_0 = 0, c_{p+1} = c_p + (h - G c_p), h = V^T r, G = V^T V
 * (device doubles, row stride ldg).  g_new != NULL: first install it as row and column k-1 of G (the newest vector).
 * passes = 0 only installs.  One workgroup; k x k work. */
int trk_cgs_coeffs(double* G_dev, int ldg, const double* h_dev, const double* g_new_dev, int k, int passes, double* c_dev,
                   trk_stream stream);
/* The same (passes >= 1) and, from *rr_dev = r . r, the squared norm of the orthogonalised vector before any pass forms it:
 * *rho2_dev = || r - V c ||^2 = r.r - 2 c.h + c.(G c).  In GKS / MMGKS r is the residual of the projected normal equations —
 * orthogonal to V but for rounding, c.h and c.G c are ~1e-14 of r.r — so the value is the computed norm (GKS.py:89) to float64
 * rounding.  Serves trk_gemv_orth_iterate. */
int trk_cgs_coeffs_rho(double* G_dev, int ldg, const double* h_dev, const double* g_new_dev, int k, int passes, double* c_dev,
                       const double* rr_dev, double* rho2_dev, trk_stream stream);
/* out = a*base + s * sum_j y[j]*V[j]   (y: k device doubles; base may be NULL; out may alias base).
 * (x = V@y: Hybrid_LSQR.py:105, Hybrid_GMRES.py:77, GKS.py:76; r -= V h: GKS.py:86-88) */
int trk_gemv_n(const float* V, int64_t ld, int k, int64_t n, const double* y_dev, double a, const float* base,
               double s, float* out, double* sumsq_dev, trk_stream stream);
/* out = V^T-combination sum_j y[j] V[j] as trk_gemv_n (a = 0, scale 1) with sum (out - ref)^2 left as *n_blocks raw block
 * partials (the relError numerator ||x_k - x_true||^2 of the solvers' loops: summed once per solve with
 * trk_finalize_batched instead of one reduction-finalize launch per iterate). */
int trk_gemv_n_err(const float* V, int64_t ld, int k, int64_t n, const double* y, float* out, const float* ref,
                   double* err_partials, int capacity_blocks, int* n_blocks, trk_stream stream);
/* GKS's one-pass form, the k x k work between the h-sweep and trk_gemv_orth_iterate in ONE launch (one workgroup): row / column k of G_A —
 * ga_new != NULL: its k + 1 entries as a pass over the kept images left them (trk_gemv_t_x), else from the sweep's products as
 * trk_gram_row_from_sweep(G_A, ..., a_A, c_sweep, s_A, rho2, c_rhs, tb) — row / column k of G_L from the sweep's products
 * (a_L, s_L), then y = (G_A + lam G_L)^-1 c_rhs over k + 1 vectors by bordering Minv from k_from (trk_gram_tikhonov's bordering form).
 * Replaces three launches (GKS.py:92-96 and :74 of the next iteration). */
int trk_gks_rows_solve(double* GA_dev, double* GL_dev, int ldg, int k, const double* ga_new, const double* a_A, const double* s_A,
                       const double* tb, const double* a_L, const double* s_L, const double* c_sweep, const double* rho2,
                       double* c_rhs, double lam, double* Minv, int ldm, int k_from, double* y, trk_stream stream);
/* The new basis vector AND the next iterate in one pass over the basis (GKS.py:76 + :86-91, MMGKS.py:108 + :119-122):
 *   vn = (w - sum_{j<k} c[j] V[j]) / sqrt(*rho2)          (the orthogonalised, normalised direction: V[k] once the caller commits it)
 *   x_next = sum_{j<k} y_next[j] V[j] + y_next[k] vn       (the iterate of the NEXT iteration, k + 1 coefficients; NULL with y_next: vn only)
 * c, rho2, y_next: device doubles (rho2 from trk_cgs_coeffs_rho; y_next from the projected problem over k + 1 vectors, whose Gram rows
 * trk_gram_row_from_sweep derives without v_k).  ref != NULL: block partials of ||x_next - ref||^2 as in trk_gemv_n_err.
 * chk_sumsq != NULL: *chk_sumsq = ||w - V c||^2 as this pass computes it (what *rho2 stands for; one more finalize launch).
 * The reference forms x = V y at the top of every iteration and r - V (V^T r) at its bottom: two passes over the basis; here one. */
int trk_gemv_orth_iterate(const float* V, int64_t ld, int k, int64_t n, const float* w, const double* c_dev, const double* rho2_dev,
                          const double* y_next_dev, float* vn, float* x_next, const float* ref, double* err_partials,
                          int capacity_blocks, int* n_blocks, double* chk_sumsq_dev, trk_stream stream);
/* out = sum_j y_host[j] V[j] with the k coefficients read from HOST memory at the call: they travel in the launch's own arguments
 * (128 per launch; more rows = more launches adding to `out`), so a projected solution computed on the host (x = V y,
 * Hybrid_LSQR.py:105) needs neither an upload nor a kernel of its own.  ref != NULL: block partials of ||out - ref||^2 as in
 * trk_gemv_n_err; ref == NULL: err_partials / n_blocks unused. */
int trk_gemv_n_hosty(const float* V, int64_t ld, int k, int64_t n, const double* y_host, float* out, const float* ref,
                     double* err_partials, int capacity_blocks, int* n_blocks, trk_stream stream);
/* One step of damped LSQR's short recurrence (Paige & Saunders): the iterate x_k = V_k y_k,
 * y_k = argmin || [B_k; damp I] y - beta_1 e_1 || — what Hybrid_LSQR.py:104-105 computes with lstsq + V @ y when lambda is a
 * number (damp = sqrt(lambda)) — from the previous one in a single pass:  w <- vk / alpha_k - (theta_k / rho_{k-1}) w  (in
 * place; first: w <- vk / alpha_1),  x_out = x_in + (phi_k / rho_k) w  (first: x_in may be NULL = 0).  vk = alpha_k v_k,
 * alpha_sq / beta_next_sq = alpha_k^2 / beta_{k+1}^2 (device doubles), beta0_sq = ||b||^2 (first step only).  state_in /
 * state_out: 4 device doubles each {cs, sn, rho, phibar}, different slots.  ref != NULL: ||x_out - ref||^2 is left as
 * *n_blocks raw block partials (trk_finalize_batched sums them), as trk_gemv_n_err does. */
int trk_lsqr_damped_update(const float* vk, float* w, const float* x_in, float* x_out, int64_t n, const float* ref,
                           double* err_partials, int capacity_blocks, int* n_blocks, const double* alpha_sq,
                           const double* beta_next_sq, const double* beta0_sq, double damp, const double* state_in,
                           double* state_out, int first, trk_stream stream);
/* ---- The float64 instrument (csrc/ref64.hip): the Golub-Kahan / damped-LSQR chain instantiated on the element type ----
 * Diagnostics (hard part 2): not a fast path.  A parallel-beam handle's operator evaluated with float64 arithmetic
 * on vectors of `elem_bytes` = 4 (float) or 8 (double); `weights` 0: interpolation weights from the geometry in float64 (the
 * oracle's numbers, trips/utilities/io.py:392-399 as oracle/cpu_ref.py Radon2D restates it), 1: the product kernels' fixed-point
 * tables (24 fractional bits) — the product operator's own weights, summed in float64. */
int trk_radon2d_apply_ref(trk_op* op, int transpose, int elem_bytes, int weights, const void* x, void* y, trk_stream stream);
/* The arithmetic every apply of a parallel-beam handle runs in from now on: 0 the product's kernels (default); 1 float64 geometry
 * and sums, 2 table weights with float64 sums — both on the usual fp32 vectors, through trk_op_apply / trk_op_apply_axpby /
 * trk_gk_step* unchanged (the fused riders are then run in launches of their own).  For experiments that separate what fp32
 * STORAGE costs a solver from what the projector's own arithmetic adds. */
int trk_radon2d_set_arithmetic(trk_op* op, int mode);
/* The instrument's kernels with EMULATED fp32 partial sums (what separates the product kernels from exact arithmetic): the forward
 * adds its products in fp32 and moves the sum into a float64 total every chunk_fwd marching steps, the adjoint every chunk_adj
 * angles, and the angle's weight is applied in fp32; 0 (default) = float64 sums.  Applies to trk_radon2d_apply_ref,
 * trk_gk_lsqr_chain and the arithmetic modes 1 / 2 above.  chunk_adj = -1 / -2 (round 6, table weights only): float64 sums, but the
 * two NEIGHBOUR rays of a pixel weighed the way the product's gather adjoint weighs them — from the nearest ray's exact t0 and the
 * angle's detector spacing, clamp(1 - |inv| -+ t0), instead of from their own table entries — with 1 - |inv| in float64 (-1) or
 * rounded to fp32 as the product holds it (-2): what the adjoint's weight rule alone costs a solver, sums and storage apart. */
int trk_radon2d_set_ref_sums(trk_op* op, int chunk_fwd, int chunk_adj);
/* out = a x + b z on float / double vectors with the coefficients of trk_axpby: float64 coefficients and products, ONE rounding
 * to the element type (the arithmetic of the projector's fused half step), *sumsq = sum out^2 (may be NULL).  x may alias out. */
int trk_ref_axpby(int elem_bytes, int64_t n, double ca, const double* a_num, const double* a_den, int a_flags, const void* x,
                  double cb, const double* b_num, const double* b_den, int b_flags, const void* z, void* out, double* sumsq,
                  trk_stream stream);
/* Hybrid-LSQR at a fixed lambda in the engine's own arrangement (Golub-Kahan on unnormalised vectors, decompositions.py:230-255;
 * the iterate by damped LSQR's short recurrence, Hybrid_LSQR.py:104-105) on vectors of elem_bytes, n_iter steps enqueued by one
 * call.  b: rows elements.  x_hist: n_iter rows of cols elements, row k = the iterate after k + 1 steps (the reference reports rows
 * 1 .. n_iter - 1).  work: 2 rows + 3 cols elements.  AB: 2 n_iter + 1 device doubles (beta0^2, alpha_1^2, beta_2^2, ...).
 * state8: 8 device doubles. */
int trk_gk_lsqr_chain(trk_op* op, int elem_bytes, int weights, const void* b, int n_iter, double lambda, void* x_hist, void* work,
                      double* AB, double* state8, trk_stream stream);

/* The weighted Gram of L V for the 2-D first-difference operator L = [D_h; D_v] of an N x N image, formed from V itself:
 *   G[a][b] = sum_e w_e^2 (L v_a)_e (L v_b)_e ,   w = [w_h: N rows of N-1 | w_v: N-1 rows of N]  (what trk_tv_weights writes).
 * Same result as trk_wgram over the stored images L v_j (MMGKS.py:94-95 through its Gram matrix) for half the bytes — n floats per
 * basis vector instead of 2n — and L V is never stored.  1 <= k <= 48, N a multiple of 32, rows of V 16-byte aligned. */
int trk_wgram_tv(const float* V, int64_t ld, int k, int N, const float* w, double* G, trk_stream stream);
/* ACCURACY CONTRACT of trk_wgram_tv / trk_wgram_tv_z, and the switch.  The 16 x 16 tile products go through the matrix cores; `mode`:
 *   1 (default)  AUTO: two bf16 pieces unless the data says otherwise.  Every call first measures, on a sample (runs of 1024 pixels in 256 image
 *                rows; four of the k basis vectors spread over the basis and, from k = 8, the NEWEST four with all their pairs — a solver's basis
 *                changes character at its end first; vectors in the middle that neither set holds are the sample's blind spot), what the
 *                two-piece split would lose — max |S' - S| / sqrt(S_aa S_bb) of the sampled Gram with
 *                and without the split, in float64 — and the verdict stays on the DEVICE: ONE Gram launch holds both arithmetic forms, every
 *                workgroup works the verdict out from the probe's sums in its prologue and takes the sweep of the form it names — two pieces
 *                below 3e-7, the fp32 pipe above (nothing visits the host; the probe reads ~24 MB whatever the image size).  TRK_WGRAM_TV_AUTO_PAIR=1
 *                selects the older A/B arrangement instead (a gated PAIR of launches, one of which returns at once).
 *                <= 1e-6 per entry relative to sqrt(G_aa G_bb) on data the sample represents; the piecewise-
 *                constant / repeated-value images of tests/test_gpu_kernels.py trip it, noisy images and Krylov vectors do not.
 *   2            each weighted difference split into TWO bf16 pieces, all four partial products: what is lost is each operand's third
 *                piece, <= 2^-16 of it.  On data whose roundings are uncorrelated the Gram is within 5e-9 of the fp32-pipe one; on
 *                images that repeat a few values millions of times an entry may be off by up to 1.2e-5 (measured 5.8e-6);
 *   3            THREE bf16 pieces (the fp32 value exactly), six partial products: <= 1e-6 on those images (measured 4.7e-7), at
 *                1.1-1.5 x the time for 17 <= k <= 32;
 *   0            the fp32 matrix pipe (v_mfma_f32_16x16x4_f32): fp32 products, <= 1e-6 likewise, 0.51-0.57 ms at 4096^2 whatever k.
 * Returns the mode in force before the call; mode -1 only queries.  PROCESS-WIDE and NOT THREAD-SAFE: one plain global read by every
 * trk_wgram_tv* call on any stream — a caller that switches it around a solve (MMGKS(gram_precision=)) must not run another solve on
 * another thread meanwhile (the engine's model is one process per GPU, one solve at a time).  Environment TRK_WGRAM_TV_F32=1 /
 * TRK_WGRAM_TV_PIECES=2|3 set the default. */
int trk_wgram_tv_precision(int mode);
/* Diagnostics: {verdict (0 two pieces ran, 1 the fp32 pipe ran), the sampled deviation} of the LAST call in mode 1, copied to two host
 * doubles after a device synchronisation; {-1, -1} before the first such call.  (The record is written by workgroup 0 of the Gram launch
 * alone — every workgroup derives the same verdict from the same sums, one of them reports it.) */
int trk_wgram_tv_last_probe(double* verdict_and_deviation_host);
/* The same pass also taking h[j] = V[j] . z for one more image z (n floats, 16-byte aligned): MMGKS forms the new Gram row
 * V^T (A^T A v_new) of the fidelity term (MMGKS.py:58 through its Gram matrix) and the re-weighted Gram of the regulariser for the
 * next iteration in ONE sweep over the basis. */
int trk_wgram_tv_z(const float* V, int64_t ld, int k, int N, const float* w, double* G, const float* z, double* h, trk_stream stream);

/* G[a][b] = sum_i w[i]^2 * W[a][i] * W[b][i]  (k x k, fp64, full symmetric; w may be NULL), and, if
 * b1 != NULL, c1[a] = sum_i w[i]*W[a][i]*b1[i], c2[a] = sum_i w[i]^2*W[a][i]*b1[i].
 * Replaces the from-scratch economic QR of AV*wf / LV*wr (MMGKS.py:58-59,94-95; GKS.py:54-56): the host
 * takes R = chol(G) and Q^T b = R^-T c. */
int trk_wgram(const float* W, int64_t ld, int k, int64_t m, const float* w, const float* b1, double* G_dev,
              double* c1_dev, double* c2_dev, trk_stream stream);

/* GKS without a pass over the basis for the Gram rows of a new vector (GKS.py:86-96).  The sweep that orthogonalises the residual r
 * against V also takes V^T (A^T A r) and V^T (L^T L r) (trk_gemv_tn: 3 or 4 right-hand sides in one pass, out[q*k + j] = V[j] . rhs[q]);
 * the new vector is v_k = (r - V c) / rho, so row k of G = V^T M V follows from a = V^T (M r), c, s = r . M r and rho^2 = ||r - V c||^2:
 *   G[i][k] = G[k][i] = (a_i - (G c)_i) / rho ,  G[k][k] = (s - 2 c.a + c.(G c)) / rho^2 ,  rhs_k = (t - c . rhs) / rho  (t = r . A^T b)
 * (trk_gram_row_from_sweep; ldg >= k+1; rhs / t may be NULL).  c is of rounding size (r is the residual of the projected
 * normal equations, orthogonal to V), so nothing cancels. */
int trk_gemv_tn(const float* V, int64_t ld, int k, int64_t n, const float* const* rhs, int n_rhs, double* out_dev,
                trk_stream stream);
int trk_gram_row_from_sweep(double* G_dev, int ldg, int k, const double* a_dev, const double* c_dev, const double* s_rr_dev,
                            const double* rho2_dev, double* rhs_dev, const double* t_dev, trk_stream stream);

/* y = (G_A + lam G_L)^-1 c on the device (float64, one workgroup): the projected Tikhonov problem of GKS.py:74 / MMGKS.py:106,
 * `lstsq([R_A; sqrt(lam) R_L], [Q_A^T b; 0])`, from the Gram data G_A = (AV)^T AV, G_L = (LV)^T LV (row strides lda, ldl) and
 * c = (AV)^T b that trk_gemv_t / trk_gemv_t2 / trk_wgram leave on the device — a numeric regparam then needs no host round
 * trip inside the loop.
 *   Minv_dev == NULL: Cholesky from scratch in LDS, O(k^3), k <= 139 (MMGKS: both Gram matrices change every iteration);
 *   Minv_dev != NULL (row stride ldm >= k): the inverse of the leading k_from x k_from block of G_A + lam G_L, left there by the
 *     previous call with the same lam, is bordered by rows k_from .. k-1 — O(k^2) per new row, any k (GKS: the Gram matrices
 *     only grow).  k_from = 0 builds it from nothing. */
int trk_gram_tikhonov(const double* GA_dev, int lda, const double* GL_dev, int ldl, const double* c_dev, int k, double lam,
                      double* Minv_dev, int ldm, int k_from, double* y_dev, trk_stream stream);
/* Hybrid-GMRES's projected problem on the device (Hybrid_GMRES.py:69-77 with a numeric regparam):
 *   y = argmin || H_k y - beta0 e1 ||^2 + lam || y ||^2 ,  H_k the (k+1) x k Hessenberg matrix of Arnoldi (decompositions.py:207-228).
 * One call per Arnoldi step k = 1, 2, ...: column k-1 of H is appended from what the orthogonalisation left on the device —
 * coef[0..k) (+ coef2[0..k) if not NULL: a second sweep's coefficients) above sqrt(*nrm2_sq) — into H_dev (column-major,
 * column stride ldh >= k+1; the caller keeps it between calls), G_dev = H^T H (row stride ldg >= k; kept between calls) gets its
 * new row and column, and (G + lam I) y = beta0 H[0,:]^T is solved (float64, one workgroup):
 *   mode 0  by Cholesky in LDS from scratch (k <= 139; any sequence of lam);
 *   mode 1  by the bordering update of the inverse kept in Minv_dev (row stride ldg): valid when the previous call (k-1) used
 *           the same lam > 0 in mode 1 or 2 — O(k^2) instead of O(k^3);
 *   mode 2  k <= 2: Minv_dev written directly (start of a mode-1 chain; lam may differ from the previous call's). */
int trk_hess_tikhonov(double* H_dev, int ldh, double* G_dev, double* Minv_dev, int ldg, const double* coef, const double* coef2,
                      const double* nrm2_sq, double beta0, int k, double lam, int mode, double* y_dev, trk_stream stream);

/* CGLS on SMALL blur problems in two launches per iteration (CGLS.py:56-80): a workgroup owns a 32 x 32 tile and recomputes
 * in LDS what it needs of its neighbours' halo (p = t + beta p and w = A p on tile + halo) instead of waiting for them at a
 * kernel boundary; same buffers, scalar layout and results (to fp32 rounding of the partial sums' order) as
 * trk_cgls_iterate_fused (w is never stored).  trk_cgls_tiled_caps: *can = 1 for separable PSFs up to 9 x 9 on images of at
 * least 16 x 16 whose tile count fits the norm-partial rows (np_capacity_blocks) and the PG / PD buffers (pcap). */
int trk_cgls_tiled_caps(trk_op* A, int np_capacity_blocks, int pcap, int* can);
int trk_cgls_iterate_tiled(trk_op* A, int k_first, int n_iters, float* P, int64_t p_ld, float* R, int64_t r_ld, float* t,
                           float* X, int64_t x_ld, int keep_history, const float* x_prev, const float* x_true, double* S,
                           double* PG, double* PD, int pcap, double* NP, int np_capacity_blocks, int* n_g_inout,
                           int* n_np_inout, trk_stream stream);

/* The same iteration with TWO blurs instead of four: w_k = A p_k is never blurred — it is A t_{k-1} + beta w_{k-1} (csrc/cgls_tiled.hip,
 * k_cgls_tile_a2 / _b2); ||w_k||^2 is measured on the vector itself.  p, w: n floats each, updated in place (w: any finite content
 * before the first iteration); R: [2][r_ld] ping-pong pair (iteration k reads index (k-1) & 1, writes k & 1); PG / PD: gamma / delta
 * block partials, `pcap` doubles each; *n_g_inout: valid gamma partials in PG (set by the caller's t_0 = A^T r_0).  Scalar layout
 * and everything else as trk_cgls_iterate_tiled. */
int trk_cgls_iterate_tiled2(trk_op* A, int k_first, int n_iters, float* p, float* w, float* R, int64_t r_ld, float* t, float* X,
                            int64_t x_ld, int keep_history, const float* x_prev, const float* x_true, double* S, double* PG,
                            double* PD, int pcap, double* NP, int np_capacity_blocks, int* n_g_inout, int* n_np_inout,
                            trk_stream stream);

/* ---------------------------------------------------------------- collectives ----
 * The sharded path — frames of a dynamic problem over the GPUs of a node, one process per GPU (io.py:420: F = blkdiag(A_t)) —
 * has two exchanges: the sum over ranks of a few device doubles (what np.dot / np.linalg.norm of the reference's solver loops
 * become: CGLS.py:61,70; decompositions.py:236-241; GKS.py:86-88) and a one-frame shift between time-neighbours for the
 * temporal rows of the space-time regulariser (operators.py:39-45).  RCCL over xGMI; resolved at run time, so the library loads
 * without it.  All calls enqueue on the caller's stream. */
/* 128 bytes of ncclUniqueId from rank 0, to be handed to every rank by the host's own means (file, MPI, torch store). */
int trk_comm_unique_id(void* id128_out);
/* Collective over all `world` ranks: ncclCommInitRank on the calling process's current device. */
int trk_comm_init(const void* id128, int rank, int world, trk_comm** out);
/* Wrap a communicator the host already has (an ncclComm_t, e.g. of its framework); not destroyed with the handle. */
int trk_comm_attach(void* nccl_comm, int rank, int world, trk_comm** out);
int trk_comm_info(const trk_comm* comm, int* rank, int* world);
int trk_comm_destroy(trk_comm* comm);
/* dev[0..count) <- sum over ranks, in place (ncclAllReduce, double); a no-op for world = 1 (unless TRK_COMM_FORCE is set in the
 * environment: diagnostics — a one-rank communicator then goes through RCCL all the same). */
int trk_allreduce_f64(trk_comm* comm, double* dev, int count, trk_stream stream);
/* Send `count` floats to rank send_to and receive `count` from rank recv_from in one group (either side is skipped when its
 * pointer is NULL or its rank outside [0, world): the first / last frame block has one neighbour only). */
int trk_halo_exchange(trk_comm* comm, const float* send, int send_to, float* recv, int recv_from, int64_t count,
                      trk_stream stream);
/* Both neighbours in one RCCL group: send_prev -> rank-1 and recv_prev <- rank-1 (rank > 0), send_next -> rank+1 and
 * recv_next <- rank+1 (rank < world-1), `count` floats each; buffers of a neighbour that does not exist are ignored.  The
 * exchange behind trk_tv_halo (the boundary frames of a time-sharded vector, operators.py:39-45). */
int trk_halo_exchange2(trk_comm* comm, const float* send_prev, float* recv_prev, const float* send_next, float* recv_next,
                       int64_t count, trk_stream stream);

/* ---- CGLS with ONE all-reduce per iteration, for unknowns spread over ranks (csrc/cgls_sharded.hip) ----
 * The recurrence of trips/solvers/CGLS.py:56-80 needs two global sums per iteration, ||A p||^2 (:61) and ||A^T r||^2 (:70), the
 * second depending on the first.  With q = A t_{k-1} formed explicitly, w_k = A p_k = q + beta w_{k-1} and
 *   delta_k = ||w_k||^2 = ||q||^2 + 2 beta <q, w_{k-1}> + beta^2 ||w_{k-1}||^2,   beta = gamma_{k-1} / gamma_{k-2},
 * so G4 = {gamma_{k-1}, ||q||^2, <q, w_{k-1}>, ||w_{k-1}||^2} — all from vectors the previous iteration left — is the one
 * exchange, and the rest of the iteration is local.  Same operator applies, same iterates in exact arithmetic.
 *
 * trk_dot_pair: out3[0] = sum q*q, out3[1] = sum q*w, out3[2] = sum w*w (w may be NULL: 0, 0) — local sums, one pass, fp64.
 * trk_cgls_sharded_update: from the all-reduced G4 and *gamma_prev (= gamma_{k-2}; unused when `first`):
 *   p = t + beta p; w = q + beta w (first: p = t, w = q); x_new = x + alpha p; r -= alpha w, alpha = gamma_{k-1} / delta_k;
 *   *publish_delta = delta_k, *publish_gamma = gamma_{k-1}; the rank's share of ||x_new||^2, ||alpha p||^2, ||x_new - x_true||^2
 *   (CGLS.py:76-80) as *n_blocks x 3 block partials (trk_finalize_batched, then one sum over the ranks after the solve).
 * trk_cgls_sharded_scalars: G4[1..3] as trk_dot_pair, and G4[0] = the sum of the n_gamma raw block partials of ||t||^2 the adjoint
 *   apply left (trk_op_apply_fused with x2 = NULL; n_gamma = 0: G4[0] is finished already) — by ONE workgroup for a rank's share of
 *   up to 2^19 samples (three launches less per iteration), else by trk_dot_pair + a finalize.
 * trk_cgls_iterate_sharded: n_iters iterations enqueued by one call — q = A t; trk_cgls_sharded_scalars; trk_allreduce_f64(comm, G4, 4)
 *   (skipped for comm = NULL); trk_cgls_sharded_update; t = A^T r.  Scalar layout as trk_cgls_iterate: S[0] = gamma_0, S[5k] =
 *   delta_k, S[5k+1] = gamma_k (gamma_k is published by iteration k+1).  PG / pcap / *n_g_inout (may be NULL / 0 / NULL): with them
 *   and an operator that has a fused apply, the rank's ||t||^2 stays *n_g_inout raw block partials in PG between the adjoint apply
 *   and the next iteration's scalars kernel (and after the last iteration); without, it is the finished local sum in G4[0].
 *   Before the first iteration: r = b - A x0, t = A^T r, and the rank's ||t||^2 in the same form. */
int trk_dot_pair(const float* q, const float* w, int64_t n, double* out3, trk_stream stream);
int trk_cgls_sharded_scalars(const float* q, const float* w, int64_t m, const double* gamma_partials, int n_gamma, double* G4,
                             trk_stream stream);
int trk_cgls_sharded_update(int64_t n, int64_t m, const double* G4, const double* gamma_prev, int first, const float* x,
                            float* p, const float* t, float* x_new, float* r, const float* q, float* w, const float* x_true,
                            double* publish_delta, double* publish_gamma, double* norm_partials, int capacity_blocks,
                            int* n_blocks, trk_stream stream);
int trk_cgls_iterate_sharded(trk_op* A, trk_comm* comm, int k_first, int n_iters, float* p, float* r, float* t, float* q,
                             float* w, float* X, int64_t x_ld, int keep_history, const float* x_prev, const float* x_true,
                             double* S, double* G4, double* NP, int np_capacity_blocks, int* n_np_inout, double* PG, int pcap,
                             int* n_g_inout, trk_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* TRK_H */
