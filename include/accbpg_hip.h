/*
 * accbpg_hip.h -- C-ABI of libaccbpg_hip.so: the MI355X (gfx950) implementation of the
 * per-iteration hot path of the accbpg package for D-optimal experiment design.
 *
 * The reference (DredderGun/accbpg_and_fw) is pure Python/NumPy and has no FFI of its own.
 * The seam this library sits behind is the duck-typed f/h object protocol its solver loops
 * call; each entry point below names the reference method whose
 * arithmetic it replaces.  All citations are relative to the reference tree.
 *
 * Conventions
 *   - every pointer named *_dev is a raw device pointer (e.g. torch.Tensor.data_ptr()),
 *     fp64, contiguous unless a leading dimension is given;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     0 / NULL means the default stream;
 *   - *_host outputs are plain host pointers; a call that has host outputs synchronises the
 *     stream before it returns, so the values are valid on return;
 *   - no entry point allocates memory the caller must free, none throws across the ABI;
 *   - return value: ACCBPG_OK, or an ACCBPG_ERR_* code that the Python shim maps to the
 *     exception type the reference raises in the same situation.
 *   - one handle per host thread; handles are bound to the device current at creation.
 */
#ifndef ACCBPG_HIP_H
#define ACCBPG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACCBPG_OK            0
#define ACCBPG_ERR_ASSERT    1   /* precondition violated -> AssertionError (functions.py:44-45,251-252,269-270,340) */
#define ACCBPG_ERR_NOT_PD    2   /* Gram matrix not positive definite -> ValueError (functions.py:48-50) */
#define ACCBPG_ERR_HIP       3   /* HIP runtime failure -> RuntimeError; see accbpg_last_error() */
#define ACCBPG_ERR_ARG       4   /* malformed call (null pointer, bad size) -> ValueError */

typedef struct accbpg_dopt accbpg_dopt;   /* opaque D-optimal objective handle */

int         accbpg_abi_version(void);
const char* accbpg_last_error(void);

/* ---- D-optimal objective: replaces DOptimalObj (accbpg/functions.py:27-59) ------------ */

/* Borrow V (m x n, row-major, leading dimension ldv >= n; the reference's `self.H`,
 * functions.py:32).  V must outlive the handle and must not change while it lives (for rows a megabyte or more
 * apart the handle keeps a copy of V stored in column blocks, which its Gram launches read).  The handle owns all workspace
 * (Gram matrix, Cholesky factor, inverse factor, stream-K slabs, pinned scalars).
 * is_shard != 0: V is a column block of a larger instance (design-point sharding), so the
 * reference's m < n precondition (functions.py:35) applies to the whole instance, not to it. */
int accbpg_dopt_create(const double* V_dev, int64_t m, int64_t n, int64_t ldv,
                       void* stream, accbpg_dopt** out, int is_shard);
int accbpg_dopt_destroy(accbpg_dopt* h);
int accbpg_dopt_set_stream(accbpg_dopt* h, void* stream);

/* func_grad(x, flag) of functions.py:43-59.  flag 0: value only (f_host), 1: gradient only
 * (g_dev), 2: both.  f = -log det(V diag(x) V^T), g_i = -v_i^T (V diag(x) V^T)^-1 v_i.
 * Returns ACCBPG_ERR_ASSERT if min(x) < 0 (functions.py:45), ACCBPG_ERR_NOT_PD if the
 * Cholesky factorisation meets a non-positive pivot (the reference tests slogdet sign <= 0,
 * functions.py:48-50). */
int accbpg_dopt_func_grad(accbpg_dopt* h, const double* x_dev, int flag,
                          double* f_host, double* g_dev);

/* func_grad split into enqueue / wait.  `begin` queues the whole evaluation on the handle's stream
 * and returns; `end` waits for it and reports f and the status.  Two handles on the same V with
 * different streams let INDEPENDENT evaluations overlap -- e.g. F[k] = f(x) and func_grad(y) of the
 * accelerated methods (accbpg/algorithms.py:135 and :148, :347 and :371; y does not depend on f(x)). */
int accbpg_dopt_func_grad_begin(accbpg_dopt* h, const double* x_dev, int flag, double* g_dev);
int accbpg_dopt_func_grad_end(accbpg_dopt* h, double* f_host);

/* *ms_host <- milliseconds from the completion (results on the host) of `first`'s last begin/end evaluation to
 * that of `second`'s; negative when `second` completed earlier.  Both evaluations must have been waited for.
 * Lets the solver stamp T[k] (the reference stamps it right after F[k] = f(x), accbpg/algorithms.py:135-137,
 * 347-349) at the moment f(x) was known although it ran beside the gradient evaluation. */
int accbpg_dopt_eval_gap_ms(accbpg_dopt* first, accbpg_dopt* second, double* ms_host);

/* The same computation in three stages, for design-point sharding across GPUs
 * (SURVEY.md 8(e).2): each rank forms its local Gram contribution, the caller all-reduces
 * gram_dev (m*m doubles, lower triangle significant) over RCCL, every rank factors it and
 * evaluates the gradient slice of its own columns.
 *   gram  : gram_dev <- lower(V diag(x) V^T) of this handle's columns      (functions.py:46)
 *   factor: in-place Cholesky of gram_dev into the handle; f_host <- -logdet (functions.py:48-51)
 *   grad  : g_dev[i] <- -|| L^-1 v_i ||^2 for this handle's columns          (functions.py:57-58) */
int accbpg_dopt_gram(accbpg_dopt* h, const double* x_dev, double* gram_dev);
int accbpg_dopt_factor(accbpg_dopt* h, const double* gram_dev, double* f_host);
int accbpg_dopt_grad(accbpg_dopt* h, double* g_dev);

/* Message forms for the sharded evaluation (SURVEY.md 8(e).2, section 5 "packed triangle"): only the lower
 * triangle of the local Gram matrix is significant, so the all-reduce carries m(m+1)/2 doubles,
 * packed[r(r+1)/2 + c] = G[r*m + c] for c <= r; accbpg_vec_count_bad leaves the number of entries of x that
 * violate the reference's `x.min() >= 0` (accbpg/functions.py:45; NaN counts) in count_dev[0] as a double, so
 * that it can travel as one more element of the same message and every rank sees every rank's violations. */
int accbpg_tri_pack(const double* G_dev, int64_t m, double* packed_dev, void* stream);
int accbpg_tri_unpack(const double* packed_dev, int64_t m, double* G_dev, void* stream);
int accbpg_vec_count_bad(const double* x_dev, int64_t n, double* count_dev, void* stream);

/* The sharded evaluation as one call, with the collectives inside the library (SURVEY.md 8(b) "dopt_shard_*",
 * 8(e).2): RCCL (librccl.so.1) is opened at run time.  `local` is a handle over this rank's columns V[:, lo:hi],
 * lo and hi from accbpg_dopt_shard_bounds (contiguous slices whose lengths differ by at most one).  Either `comm`
 * is an RCCL communicator of the caller's (an ncclComm_t; used, not owned), or it is null and the communicator
 * is made from the ACCBPG_SHARD_ID_BYTES-byte token that accbpg_shard_unique_id gave rank 0 and rank 0 handed round.
 * accbpg_dopt_shard_func_grad evaluates f(x) = -log det(V diag(x) V^T) (accbpg/functions.py:40-60) at the whole
 * length-n device vector x, which every rank holds: one all-reduce of the packed Gram triangle (+ the x >= 0
 * violation count) and, for flags 1 and 2, one all-gather of the gradient slices; g_dev receives the whole
 * gradient on every rank.  Return codes as accbpg_dopt_func_grad, the same on every rank -- EXCEPT ACCBPG_ERR_HIP: a HIP
 * or RCCL failure on one rank returns on that rank before the collective the other ranks are entering, and they stay in
 * it (RCCL has no timeout of its own); a caller that survives ACCBPG_ERR_HIP must tear the job down.  At world > 1 these
 * entry points have not run on hardware yet (one GPU per box in the builds so far): the message layout and the assembly
 * of unequal slices are covered with two gloo ranks on the CPU and with one rank on a GPU. */
#define ACCBPG_SHARD_ID_BYTES 128
typedef struct accbpg_dopt_shard accbpg_dopt_shard;
int accbpg_dopt_shard_bounds(int64_t n, int world, int rank, int64_t* lo, int64_t* hi);
int accbpg_shard_unique_id(void* id_out);
int accbpg_dopt_shard_create(accbpg_dopt* local, int64_t n, int world, int rank, const void* unique_id,
                             void* comm, accbpg_dopt_shard** out);
int accbpg_dopt_shard_func_grad(accbpg_dopt_shard* s, const double* x_dev, int flag, double* f_host, double* g_dev);
int accbpg_dopt_shard_destroy(accbpg_dopt_shard* s);
/* test hook: gather the gradient through the padded staging buffer (the path of unequal slices), `extra` spare
 * entries per rank */
int accbpg_debug_shard_pad(accbpg_dopt_shard* s, int64_t extra);

/* Linearity of the Gram matrix in x (an extension with no reference counterpart; opt-in from the
 * Python side): out <- a*G1 + b*G2 is the Gram matrix at a*x1 + b*x2 when G1, G2 are those at x1, x2;
 * accbpg_dopt_eval_gram finishes func_grad from a Gram matrix (Cholesky, log det, gradient). */
int accbpg_dopt_gram_lincomb(accbpg_dopt* h, double a, const double* G1_dev, double b,
                             const double* G2_dev, double* out_dev);
int accbpg_dopt_eval_gram(accbpg_dopt* h, const double* gram_dev, int flag, double* f_host, double* g_dev);

/* ---- Batches of same-shaped instances on one GPU (BASELINE config 4; SURVEY.md 8(b) "*_batched", 8(e).1) ----------
 * K independent D-optimal problems of one shape advance in lock-step: every entry point below covers the ACTIVE
 * instances (active_host[i] != 0; NULL = all) with ONE launch per kernel family and ONE readback, and reports a status
 * per instance.  Vectors are K x n arrays, row i = instance i, leading dimension ld.  The arithmetic of an instance is
 * that of accbpg_dopt_func_grad / accbpg_burg_simplex_div_prox / accbpg_ls_terms / accbpg_vec_axpby on the handle
 * accbpg_dopt_batch_instance(b, i) -- bit for bit -- so a batch and a loop over its instances give identical results.
 * The per-instance decisions (stopping, line search) stay with the caller, who re-issues the instances that need
 * another pass.  Shapes outside the fused path (m not a multiple of 256, n not a multiple of 128, unaligned rows) are
 * evaluated instance by instance behind the same interface.  A batch holds at most ACCBPG_BATCH_MAX instances;
 * accbpg_dopt_batch_create returns ACCBPG_ERR_ARG for more (split them over several batches). */
#define ACCBPG_BATCH_MAX 64
typedef struct accbpg_dopt_batch accbpg_dopt_batch;

/* V_dev_host: HOST array of K device pointers (m x n row-major matrices, leading dimension ldv, borrowed). */
int accbpg_dopt_batch_create(const double* const* V_dev_host, int K, int64_t m, int64_t n, int64_t ldv, void* stream,
                             accbpg_dopt_batch** out);
int accbpg_dopt_batch_destroy(accbpg_dopt_batch* b);
int accbpg_dopt_batch_set_stream(accbpg_dopt_batch* b, void* stream);
int accbpg_dopt_batch_size(accbpg_dopt_batch* b);
int accbpg_dopt_batch_is_fused(accbpg_dopt_batch* b);          /* 1: one launch per kernel family covers the batch */
int accbpg_dopt_batch_chunk(accbpg_dopt_batch* b);             /* instances one launch covers: all of them, unless their
                                                                  one-launch factorisations do not fit the chip together */
accbpg_dopt* accbpg_dopt_batch_instance(accbpg_dopt_batch* b, int i);   /* owned by the batch */

/* DOptimalObj.func_grad (accbpg/functions.py:43-59) for the active instances.  f_host[i], status_host[i]
 * (ACCBPG_OK / ACCBPG_ERR_ASSERT: min(x_i) < 0 / ACCBPG_ERR_NOT_PD) are written for those only. */
int accbpg_dopt_batch_func_grad(accbpg_dopt_batch* b, const double* x_dev, int64_t ldx, const int* active_host, int flag,
                                double* f_host, double* g_dev, int64_t ldg, int* status_host);
/* The same split into enqueue / wait (as accbpg_dopt_func_grad_begin / _end): two batches over the same matrices on
 * two streams let the value evaluations F[k] = f(x_i) run beside the gradient evaluations at y_i. */
int accbpg_dopt_batch_func_grad_begin(accbpg_dopt_batch* b, const double* x_dev, int64_t ldx, const int* active_host,
                                      int flag, double* g_dev, int64_t ldg);
int accbpg_dopt_batch_func_grad_end(accbpg_dopt_batch* b, double* f_host, int* status_host);

/* BurgEntropySimplex.div_prox_map(y_i, g_i, L_host[i]) (accbpg/functions.py:264-271, 336-356; y_dev NULL: prox_map).
 * info_host (optional): {bisection steps, Newton steps} per instance. */
int accbpg_dopt_batch_burg_simplex_div_prox(accbpg_dopt_batch* b, const double* y_dev, const double* g_dev, int64_t ld,
                                            const double* L_host, double eps, double* x_out_dev, const int* active_host,
                                            int* status_host, int* info_host);

/* out_host[3i..3i+2] = { <g_i, x_i - y_i>, D_h(x_i, y_i), D_h(z_i, z1_i) } (accbpg/algorithms.py:153-154, 377-378, 387). */
int accbpg_dopt_batch_ls_terms(accbpg_dopt_batch* b, const double* g_dev, const double* x_dev, const double* y_dev,
                               const double* z_dev, const double* z1_dev, int64_t ld, const int* active_host,
                               double* out_host, int* status_host);

/* out_i = a_host[i] * x_i + b_host[i] * z_i (accbpg/algorithms.py:147,150). */
int accbpg_dopt_batch_axpby(accbpg_dopt_batch* b, const double* a_host, const double* x_dev, const double* b_host,
                            const double* z_dev, int64_t ld, const int* active_host, double* out_dev);

/* ---- Burg entropy on the simplex: replaces BurgEntropy / BurgEntropySimplex ------------
 * (accbpg/functions.py:238-271, 326-356).  `ws_dev` is caller-provided scratch of at least
 * accbpg_vec_workspace_doubles(n) doubles. */
int64_t accbpg_vec_workspace_doubles(int64_t n);

/* x_out <- argmin_{x in simplex} <g,x> + L*D_h(x,y)  = prox_map(g + L/y, L)
 * (functions.py:264-271 then 336-356): same start c = -min(gg)+1, same bisection, same
 * Newton recurrence, stall test and |phi| <= eps stopping rule; result not renormalised.
 * y_dev == NULL selects plain prox_map(g, L) (functions.py:336).  info_host (optional,
 * 2 ints) receives {bisection steps, Newton steps}.  ACCBPG_ERR_ASSERT if L <= 0 or
 * min(y) <= 0. */
int accbpg_burg_simplex_div_prox(const double* y_dev, const double* g_dev, double L, double eps,
                                 int64_t n, double* x_out_dev, double* ws_dev,
                                 int* info_host, void* stream);

/* out_host[0] <- D_h(x,y) = sum_i (x_i/y_i - log(x_i/y_i) - 1)   (functions.py:250-253).
 * ACCBPG_ERR_ASSERT if min(x) <= 0 or min(y) <= 0. */
int accbpg_burg_divergence(const double* x_dev, const double* y_dev, int64_t n,
                           double* out_host, double* ws_dev, void* stream);

/* Line-search terms in one readback: out_host = { <g, x - y>, D_h(x,y), D_h(z,z1) }
 * (algorithms.py:53, 153-154, 377-378, 387).  Any of (g), (z,z1) may be NULL to skip. */
int accbpg_ls_terms(const double* g_dev, const double* x_dev, const double* y_dev,
                    const double* z_dev, const double* z1_dev, int64_t n,
                    double* out_host, double* ws_dev, void* stream);

/* out <- a*x + b*z elementwise, rounded as NumPy rounds (1-theta)*x + theta*z
 * (algorithms.py:147,150,369,374): two products, one sum, no fused multiply-add. */
int accbpg_vec_axpby(double a, const double* x_dev, double b, const double* z_dev, int64_t n,
                     double* out_dev, void* stream);

/* out_host[0] <- <g, x - y>   (algorithms.py:53,168,387,406) */
int accbpg_vec_dot_diff(const double* g_dev, const double* x_dev, const double* y_dev, int64_t n,
                        double* out_host, double* ws_dev, void* stream);

/* out_host = { min(x), sum(x) } -- precondition checks and diagnostics. */
int accbpg_vec_min_sum(const double* x_dev, int64_t n, double* out_host, double* ws_dev, void* stream);

/* ---- callers either side of the path ------------------------------------- */

/* out <- x / d, NumPy true division (gavg/csum of ABDA, accbpg/algorithms.py:485) */
int accbpg_vec_div_scalar(const double* x_dev, double d, int64_t n, double* out_dev, void* stream);

/* out[i] <- fill, out[idx] <- value: the simplex vertex returned by lmo_simplex
 * (accbpg/functions_lmo.py:152-157: 1e-15 everywhere, radius at the first argmin of g). */
int accbpg_vec_vertex(int64_t idx, double value, double fill, int64_t n, double* out_dev, void* stream);

/* idx_host = {first argmin, first argmax} of x, val_host (optional) = {min, max}
 * (np.argmin / np.argmax, accbpg/applications.py:80-81; np.where(g == g.min())[0][0],
 * functions_lmo.py:155-156). */
int accbpg_vec_argminmax(const double* x_dev, int64_t n, int64_t* idx_host, double* val_host,
                         double* ws_dev, void* stream);

/* u <- V^T q (length n) for a length-m q: np.dot(q, V) of D_opt_KYinit (accbpg/applications.py:79). */
int accbpg_dopt_vt_times(accbpg_dopt* h, const double* q_dev, double* u_dev);

/* out <- V[:, j] (length m): the column reads of D_opt_KYinit (accbpg/applications.py:84). */
int accbpg_dopt_get_column(accbpg_dopt* h, int64_t j, double* out_dev);

/* ---- Frank-Wolfe / Wolfe-Atwood state: replaces the bodies of D_opt_FW and --------------
 * D_opt_FW_away (accbpg/D_opt_alg.py:9-88, 91-185).  State (x, inverse H = (V X V^T)^-1,
 * w_i = v_i^T H v_i) lives in the handle. */

typedef struct accbpg_fw_probe {
    int64_t i;         /* argmax_i w_i                         (D_opt_alg.py:59,145)            */
    int64_t j;         /* away index                           (D_opt_alg.py:60-61 / 146-147)   */
    double  w_i;       /* w[i]                                                                */
    double  w_j;       /* w[j] (FW: min of w over x > 0)                                       */
    double  x_j;       /* x[j]                                                                */
    double  logdet_H;  /* log det of the maintained inverse H  (D_opt_alg.py:136)              */
    double  q_prev;    /* v_p^T H v_p of the LAST accbpg_fw_update's pivot column p in the inverse as it stood before
                          that update (0 before the first): det(H+) = det(H) (1 + hcoef q) / hdiv^m             */
} accbpg_fw_probe;

/* x <- x0; H <- (V diag(x0) V^T)^-1; w <- diag(V^T H V)   (D_opt_alg.py:39-45 / 123-129).
 * logdet_gram_host <- log det(V diag(x0) V^T). */
int accbpg_fw_init(accbpg_dopt* h, const double* x0_dev, double* logdet_gram_host);

/* Reduction pass over w, x: fills *probe_host.  away = 0: j = argmin of w over {x > 0}
 * (D_opt_alg.py:60-61); away = 1: j = argmin of (w - w_i) * [x > 1e-8], first index on ties
 * (D_opt_alg.py:146-147).  refresh_logdet != 0 also refactors H for logdet_H (D_opt_alg.py:136). */
int accbpg_fw_probe_step(accbpg_dopt* h, int away, int refresh_logdet, accbpg_fw_probe* probe_host);

/* refresh_logdet = 2 is the pipelined form of 1: log det(H) is a logged value that no decision of the iteration reads
 * (D_opt_alg.py:136 -> F[k] only), so a copy of the current H is factored on a side stream (by an auxiliary handle with
 * buffers of its own) while the caller goes on to probe, decide and update.  `depth` such factorisations may be in
 * flight (accbpg_fw_logdet_ring; default 1): probe_host->logdet_H holds the value that belongs to the call made this way
 * `depth` calls earlier (NaN while there is none), accbpg_fw_logdet_flush collects the oldest one still in flight (NaN
 * when none is), accbpg_fw_logdet_pending counts them.  Same kernels on the same matrices as form 1.
 * small_launches: how the side factorisations run, as accbpg_dopt_factor_in_small_launches (0 one launch, 1 a launch
 * per block column, 2 by size).  A caller that refreshes only every R-th call and advances log det(H) in between by
 * the matrix determinant lemma uses q_prev (D_opt_FW_away(..., logdet_refresh=R) of the Python package). */
int accbpg_fw_logdet_ring(accbpg_dopt* h, int depth, int small_launches);
int accbpg_fw_logdet_pending(accbpg_dopt* h);
int accbpg_fw_logdet_flush(accbpg_dopt* h, double* logdet_host);

/* One rank-one update with pivot column p (i for a Frank-Wolfe step, j for an away step):
 *   x <- x*xscale; x[p] += xadd; Hv = H V[:,p];
 *   H <- (H + hcoef * Hv Hv^T) * hscale_inv ...  written exactly as
 *   H <- (H + hcoef*outer(Hv,Hv)) / hdiv and w <- (w + hcoef*(Hv^T V)^2) / hdiv
 * with the scalars computed by the caller as the reference writes them
 * (D_opt_alg.py:75-82, 163-170, 172-179; hcoef carries its sign). */
int accbpg_fw_update(accbpg_dopt* h, int64_t p, double xscale, double xadd, double hcoef, double hdiv);

/* Copy state out (device to device): x (n), w (n), H (m*m, row-major).  NULL skips. */
int accbpg_fw_get_state(accbpg_dopt* h, double* x_dev, double* w_dev, double* H_dev);

/* ---- Poisson linear inverse problem with Burg L1 / L2 kernels -------- */

typedef struct accbpg_poisson accbpg_poisson;

/* f(x) = D_KL(b, Ax): replaces PoissonRegression.__init__ (accbpg/functions.py:89-94).  A is a row-major
 * m x n device matrix with leading dimension lda, b a length-m device vector; both stay owned by the caller
 * and must outlive the handle. */
int accbpg_poisson_create(const double* A_dev, int64_t m, int64_t n, int64_t lda, const double* b_dev,
                          void* stream, accbpg_poisson** out);
int accbpg_poisson_destroy(accbpg_poisson* h);
int accbpg_poisson_set_stream(accbpg_poisson* h, void* stream);

/* PoissonRegression.func_grad (accbpg/functions.py:102-120): flag 0 -> *f_host = sum(b*log(b/Ax) + Ax - b);
 * flag 1 -> g_dev = A^T (1 - b/Ax); flag 2 -> both.  Synchronises the stream when a value is returned. */
int accbpg_poisson_func_grad(accbpg_poisson* h, const double* x_dev, int flag, double* f_host, double* g_dev);

/* out_dev <- Ax of the last func_grad (length m). */
int accbpg_poisson_get_ax(accbpg_poisson* h, double* out_dev);

/* Closed-form Burg-entropy prox maps on x > 0.  kind 0: BurgEntropy.prox_map L/g (accbpg/functions.py:255-262);
 * kind 1: BurgEntropyL1.prox_map L/(lamda+g) (:290-298); kind 2: BurgEntropyL2.prox_map (:316-323).  With
 * y_dev != NULL the argument is g - L*(-1/y) first, i.e. BurgEntropy.div_prox_map (:264-271).
 * ACCBPG_ERR_ASSERT where the reference asserts: L <= 0, y.min() <= 0, g.min() <= 0 (kind 0),
 * g.min() <= -lamda (kind 1). */
int accbpg_burg_reg_div_prox(int kind, const double* y_dev, const double* g_dev, double L, double lamda,
                             int64_t n, double* x_out_dev, void* stream);

/* *out_host = <x, y>  (np.dot(x, x) of BurgEntropyL2.extra_Psi, accbpg/functions.py:314). */
int accbpg_vec_dot(const double* x_dev, const double* y_dev, int64_t n, double* out_host, double* ws_dev,
                   void* stream);

/* ---- diagnostics ---------------------------------------------------------------------- */

/* Kernel time accounting for the roofline line of bench.py: accumulates HIP-event durations
 * of the named kernel family on the handle's stream between reset and read.
 * which: 0 = Gram stream-K kernel (weighted SYRK), 1 = Cholesky (all its launches), 2 = triangular
 * inverse (all its launches), 3 = gradient kernel (triangular product + column norms), 4 = Gram
 * fix-up kernel, 5 = the pass over V of a Frank-Wolfe update (u = Hv^T V).  enable != 0 turns event recording on.
 * For a batch (accbpg_dopt_batch_*) the batched launches are accounted on the handle accbpg_dopt_batch_instance(b, 0). */
int accbpg_dopt_profile_enable(accbpg_dopt* h, int enable);
int accbpg_dopt_profile_read(accbpg_dopt* h, int which, double* total_ms_host, int64_t* launches_host);
int accbpg_dopt_profile_reset(accbpg_dopt* h);

/* fp64 MFMA peak microbenchmark: runs `iters` dependent-free v_mfma_f64_16x16x4_f64 per wave on
 * every CU and reports achieved TFLOP/s (used to confirm the roofline constant on the box). */
int accbpg_mfma_f64_peak(int iters, double* tflops_host, void* stream);

/* Development probe: milliseconds for `iters` loop trips of 8 fp64 MFMAs (mode bit 0) and / or 128 fp64 vector FMAs
 * (mode bit 1) per wave, one wave per SIMD on every CU -- do the matrix and the vector pipe run at the same time? */
int accbpg_debug_pipe_probe(int iters, int mode, double* ms_host, void* stream);

/* Unit-test hook: C (MxN) <- alpha * A (MxK, row-major) * op(B) + beta*C on the MFMA engine.
 * b_kmajor != 0: B is K x N row-major; else B is N x K row-major (A * B^T). */
int accbpg_test_gemm(const double* A_dev, int64_t lda, const double* B_dev, int64_t ldb,
                     double* C_dev, int64_t ldc, int64_t M, int64_t N, int64_t K,
                     int b_kmajor, double alpha, double beta, int config, void* stream);

/* Timing ablations of the Gram kernel (development aid; variant 0 is the product kernel, the
 * others drop global loads / LDS staging / the barrier / fragment reads and produce wrong data
 * into scratch).  Average milliseconds per launch over `iters` launches. */
int accbpg_debug_gram_variant(accbpg_dopt* h, const double* x_dev, int variant, int iters, double* ms_host);
/* For a handle whose evaluations run BESIDE another stream's MFMA-bound launches (the value evaluation the solvers
 * start next to a gradient evaluation, accbpg/algorithms.py:135 beside :148): factor with one launch per 64-wide
 * block column -- a few workgroups at a time -- instead of the one-launch kernel, whose waiting workgroups would
 * hold the compute units the other stream needs.  Same arithmetic in the same order: bit-identical results.
 * on: 0 = one launch (default), 1 = small launches, 2 = small launches where the one launch would put at least a
 * workgroup on every compute unit (m > 1408 on 256 CUs), one launch below that. */
int accbpg_dopt_factor_in_small_launches(accbpg_dopt* h, int on);

/* Development switch for handles created AFTER the call: bit 0 = plain stream-K ranges of the Gram kernel also where the
 * tile list is longer than the grid (m >= 4096), instead of whole tiles per workgroup (A/B measurements). */
int accbpg_debug_plan_flags(int flags);

/* Timing ablation bits for the Cholesky step kernel (development aid; 0 = product behaviour):
 * 1 skip the diagonal-block factorisation, 2 skip the panel solve, 4 skip the MFMA products; 8, 16, 32 switch
 * off the row solves / MFMA updates / 16x16 factor inside the 64x64 factorisation.  256 / 512 select the
 * register-staged / direct-to-LDS Gram and gradient kernels; bits 30..31 select the schedule of the direct-to-LDS
 * kernels (0 product, 1 / 2 development variants; bit-identical results).  2048 sets the scheme of the factorisation
 * (results agree to rounding): bits 12..23 = number of 64-wide block columns from which the two-level scheme
 * runs (default 64), bits 24..29 = block columns per outer panel (default 8; 0 leaves it). */
int accbpg_debug_chol_variant(accbpg_dopt* h, int bits);
/* Development aid for the one-launch Cholesky (m <= 2048): factor gram_dev once while the workgroups on its critical
 * chain stamp the 100 MHz wall clock at their stage boundaries.  stamps_host: 32 x ceil(m/64) int64 (per block column:
 * workgroup start, left updates in, first piece of the previous factor here, last piece here, panel solve done,
 * diagonal tile staged, factored, published; then per 16-column piece of the streamed panel solve: asked for,
 * in LDS, solved, folded).  with_inverse != 0 also forms the inverses of the diagonal blocks (as a gradient evaluation). */
int accbpg_debug_chol_trace(accbpg_dopt* h, const double* gram_dev, int with_inverse, int64_t* stamps_host);

#ifdef __cplusplus
}
#endif
#endif /* ACCBPG_HIP_H */
