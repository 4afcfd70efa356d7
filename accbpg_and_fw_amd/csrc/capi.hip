// C-ABI entry points of libaccbpg_hip.so (see include/accbpg_hip.h for the contract).
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>

#include "internal.h"

namespace accbpg {

static thread_local char g_err[512] = "";

void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// the one-launch Cholesky gave up a wait and the handle stays on the launch-per-column kernels from here on: results are
// the same, every factorisation is slower -- worth one line on stderr (once per process)
void note_tiles_fallback(const char* where) {
    static bool said = false;
    if (said) return;
    said = true;
    fprintf(stderr, "libaccbpg_hip: %s: the one-launch Cholesky abandoned a wait (another process on this GPU, or fewer "
                    "workgroups resident than the occupancy query promised); this handle factors with one launch per block "
                    "column from now on\n", where);
}

static int read_status(accbpg_dopt* h) {
    ACC_HIP(hipMemcpyAsync(h->hpin, h->dscal, sizeof(double) * STATUS_DOUBLES, hipMemcpyDeviceToHost, h->stream));   // scalars + flags
    ACC_HIP(hipStreamSynchronize(h->stream));
    return ACCBPG_OK;
}

}  // namespace accbpg

using namespace accbpg;

// 2: batches, one-launch Cholesky, pipelined FW log det
// 3: accbpg_fw_probe grew q_prev; ring of side factorisations (accbpg_fw_logdet_ring / _pending); accbpg_dopt_batch_chunk
extern "C" int accbpg_abi_version(void) { return 3; }
extern "C" const char* accbpg_last_error(void) { return g_err; }

namespace accbpg {
int dopt_init(accbpg_dopt* h) {
    const int64_t m = h->m, n = h->n;
    ACC_HIP(hipGetDevice(&h->device));
    hipDeviceProp_t prop;
    ACC_HIP(hipGetDeviceProperties(&prop, h->device));
    h->num_cu = prop.multiProcessorCount;
    h->vec_ok = ((reinterpret_cast<uintptr_t>(h->V) & 15) == 0) && ((h->ldv & 1) == 0);
    h->big = (m >= 768) || (h->force_big && h->vec_ok && m >= 256 && (m % 256 == 0) && (n % 128 == 0));
    const size_t mm = sizeof(double) * (size_t)m * (size_t)m;
    ACC_HIP(hipMalloc(&h->Lbuf, mm));
    ACC_HIP(hipMalloc(&h->Wbuf, mm));
    ACC_HIP(hipMalloc(&h->Tbuf, mm));
    ACC_HIP(hipMemset(h->Lbuf, 0, mm));
    ACC_HIP(hipMemset(h->Wbuf, 0, mm));     // the upper triangle of W must read as zero
    ACC_HIP(hipMemset(h->Tbuf, 0, mm));
    if (h->dscal_ext) {
        h->dscal = h->dscal_ext;                                // a slice of the batch's array (zeroed by the batch)
    } else {
        ACC_HIP(hipMalloc(&h->dscal, sizeof(double) * 24));     // 16 scalars, then the status flags: one readback
        ACC_HIP(hipMemset(h->dscal, 0, sizeof(double) * 24));
    }
    h->dflag = reinterpret_cast<int*>(h->dscal + 16);
    ACC_HIP(hipHostMalloc(&h->hpin, sizeof(double) * 32, hipHostMallocDefault));
    const int64_t vws = std::max<int64_t>(vec_ws_doubles(n), 64 * n);
    ACC_HIP(hipMalloc(&h->vws, sizeof(double) * (size_t)vws));
    ACC_HIP(hipEventCreate(&h->ev_done));
    return build_plans(h);
}
}  // namespace accbpg

extern "C" int accbpg_dopt_create(const double* V_dev, int64_t m, int64_t n, int64_t ldv, void* stream,
                                  accbpg_dopt** out, int is_shard) {
    if (!V_dev || !out || m <= 0 || n <= 0 || ldv < n) {
        set_last_error("accbpg_dopt_create: bad arguments (m=%lld n=%lld ldv=%lld)", (long long)m, (long long)n,
                       (long long)ldv);
        return ACCBPG_ERR_ARG;
    }
    if (!is_shard && !(m < n)) {            // DOptimalObj: need m < n   (functions.py:35)
        set_last_error("DOptimalObj: need m < n");
        return ACCBPG_ERR_ASSERT;
    }
    accbpg_dopt* h = new accbpg_dopt();
    h->V = V_dev; h->m = m; h->n = n; h->ldv = ldv; h->stream = (hipStream_t)stream;
    const int rc = dopt_init(h);
    if (rc != ACCBPG_OK) {                  // nothing of a half-built handle stays behind
        accbpg_dopt_destroy(h);
        return rc;
    }
    *out = h;
    return ACCBPG_OK;
}

extern "C" int accbpg_dopt_destroy(accbpg_dopt* h) {
    if (!h) return ACCBPG_OK;
    hipFree(h->Vblk);
    hipFree(h->Lbuf); hipFree(h->Wbuf); hipFree(h->Tbuf); hipFree(h->slabs); hipFree(h->tiles); hipFree(h->wg_ranges); hipFree(h->gram_cstart); hipFree(h->gram_contrib);
    if (!h->dscal_ext) hipFree(h->dscal);
    hipFree(h->vws); hipFree(h->xbuf); hipFree(h->ops); hipFree(h->chol_op); hipFree(h->red); hipFree(h->Pbuf);
    hipFree(h->fw_x); hipFree(h->fw_w); hipFree(h->fw_H); hipFree(h->fw_hv);
    hipFree(h->chol_jobs); hipFree(h->chol_ready); hipFree(h->chol_aux); hipFree(h->chol_hand); hipFree(h->Gbuf);
    if (h->hpin) hipHostFree(h->hpin);
    if (h->ev_done) hipEventDestroy(h->ev_done);
    for (auto& sl : h->fw_ring) {
        if (sl.stream) hipStreamSynchronize(sl.stream);
        if (sl.aux) accbpg_dopt_destroy(sl.aux);
        if (sl.stream) hipStreamDestroy(sl.stream);
        if (sl.ev_snap) hipEventDestroy(sl.ev_snap);
    }
    for (auto& p : h->prof)
        for (auto e : p.ev) hipEventDestroy(e);
    delete h;
    return ACCBPG_OK;
}

extern "C" int accbpg_dopt_set_stream(accbpg_dopt* h, void* stream) {
    if (!h) return ACCBPG_ERR_ARG;
    h->stream = (hipStream_t)stream;
    return ACCBPG_OK;
}

extern "C" int accbpg_dopt_gram(accbpg_dopt* h, const double* x_dev, double* gram_dev) {
    if (!h || !x_dev || !gram_dev) return ACCBPG_ERR_ARG;
    return launch_gram(h, x_dev, gram_dev);
}

extern "C" int accbpg_dopt_factor(accbpg_dopt* h, const double* gram_dev, double* f_host) {
    if (!h || !gram_dev) return ACCBPG_ERR_ARG;
    ACC_TRY(launch_cholesky(h, h->Lbuf, nullptr, nullptr, gram_dev));
    ACC_TRY(read_status(h));
    const int* fl = reinterpret_cast<const int*>(h->hpin + 16);
    if (fl[FLAG_ABORT]) {                    // the one-launch factorisation gave up a wait: launch per block column
        if (gram_dev == h->Lbuf) {
            set_last_error("accbpg_dopt_factor: in-place factorisation was abandoned; pass the Gram matrix in a buffer of its own");
            return ACCBPG_ERR_HIP;
        }
        h->chol_tiles_off = true;
        note_tiles_fallback("accbpg_dopt_factor");
        return accbpg_dopt_factor(h, gram_dev, f_host);
    }
    if (fl[FLAG_NOT_PD]) {
        set_last_error("HXHT is singular or not positive definite");
        return ACCBPG_ERR_NOT_PD;
    }
    if (f_host) *f_host = -h->hpin[0];       // f = -logdet   (functions.py:51)
    return ACCBPG_OK;
}

extern "C" int accbpg_dopt_grad(accbpg_dopt* h, double* g_dev) {
    if (!h || !g_dev) return ACCBPG_ERR_ARG;
    ACC_TRY(launch_trtri(h));
    ACC_TRY(launch_colnorm(h, h->Wbuf, g_dev, -1.0));
    return ACCBPG_OK;
}

/* Enqueue a whole func_grad on the handle's stream without waiting: Gram, Cholesky, (gradient),
 * and the copy of the scalars / flags to the pinned mirror.  accbpg_dopt_func_grad_end waits for it.
 * Two handles on the same V with different streams let independent evaluations overlap (the
 * latency-bound factorisation of one under the MFMA-bound products of the other). */
extern "C" int accbpg_dopt_func_grad_begin(accbpg_dopt* h, const double* x_dev, int flag, double* g_dev) {
    if (!h || !x_dev || flag < 0 || flag > 2) return ACCBPG_ERR_ARG;
    if (flag != 0 && !g_dev) return ACCBPG_ERR_ARG;
    h->last_x = x_dev; h->last_flag = flag; h->last_g = g_dev;
    // (with the one-launch Cholesky the Gram matrix gets a buffer of its own and stays intact for a redo)
    double* gram = chol_tiles_usable(h) ? h->Gbuf : h->Lbuf;
    ACC_TRY(launch_gram(h, x_dev, gram));
    // resets the scalars and flags first, and checks x >= 0 in the same launch (functions.py:45)
    ACC_TRY(launch_cholesky(h, h->Lbuf, flag != 0 ? h->Wbuf : nullptr, x_dev, gram));
    if (flag != 0) {
        ACC_TRY(launch_trtri(h));
        ACC_TRY(launch_colnorm(h, h->Wbuf, g_dev, -1.0));
    }
    ACC_HIP(hipMemcpyAsync(h->hpin, h->dscal, sizeof(double) * STATUS_DOUBLES, hipMemcpyDeviceToHost, h->stream));   // scalars + flags
    ACC_HIP(hipEventRecord(h->ev_done, h->stream));            // when this evaluation's results are on the host
    return ACCBPG_OK;
}

/* Milliseconds from the completion of `first`'s last begin/end evaluation to the completion of `second`'s
 * (negative when `second` finished earlier).  Both must have been waited for with accbpg_dopt_func_grad_end.
 * The solvers use it to stamp T[k] at the moment F[k] was known when f(x) ran beside the gradient evaluation. */
extern "C" int accbpg_dopt_eval_gap_ms(accbpg_dopt* first, accbpg_dopt* second, double* ms_host) {
    if (!first || !second || !ms_host) return ACCBPG_ERR_ARG;
    float ms = 0.f;
    const hipError_t e = hipEventElapsedTime(&ms, first->ev_done, second->ev_done);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        float back = 0.f;                                      // some runtimes refuse a negative span: ask the other way
        if (hipEventElapsedTime(&back, second->ev_done, first->ev_done) != hipSuccess) {
            (void)hipGetLastError();
            set_last_error("accbpg_dopt_eval_gap_ms: no completed evaluation on one of the handles");
            return ACCBPG_ERR_ARG;
        }
        ms = -back;
    }
    *ms_host = (double)ms;
    return ACCBPG_OK;
}

extern "C" int accbpg_dopt_func_grad_end(accbpg_dopt* h, double* f_host) {
    if (!h) return ACCBPG_ERR_ARG;
    ACC_HIP(hipStreamSynchronize(h->stream));
    const int* fl = reinterpret_cast<const int*>(h->hpin + 16);
    if (fl[FLAG_ABORT] && !h->chol_tiles_off) {
        // the one-launch factorisation gave up a wait (its workgroups were not all resident in time, e.g. another
        // process shares the GPU): redo this evaluation with one launch per block column, and stay with that
        h->chol_tiles_off = true;
        note_tiles_fallback("accbpg_dopt_func_grad");
        ACC_TRY(accbpg_dopt_func_grad_begin(h, h->last_x, h->last_flag, h->last_g));
        return accbpg_dopt_func_grad_end(h, f_host);
    }
    if (fl[FLAG_NEG_X]) {
        set_last_error("DOptimalObj: x needs to be nonnegative");
        return ACCBPG_ERR_ASSERT;
    }
    if (fl[FLAG_NOT_PD]) {
        set_last_error("HXHT is singular or not positive definite");
        return ACCBPG_ERR_NOT_PD;
    }
    if (f_host) *f_host = -h->hpin[0];
    return ACCBPG_OK;
}

extern "C" int accbpg_dopt_func_grad(accbpg_dopt* h, const double* x_dev, int flag, double* f_host, double* g_dev) {
    ACC_TRY(accbpg_dopt_func_grad_begin(h, x_dev, flag, g_dev));
    return accbpg_dopt_func_grad_end(h, f_host);
}

namespace accbpg {
__global__ void lincomb_kernel(double a, const double* __restrict__ G1, double b, const double* __restrict__ G2,
                               int64_t total, double* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 2;
    for (int64_t e = 2 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x); e < total; e += stride) {
        if (e + 1 < total) {
            const double2 u = *reinterpret_cast<const double2*>(G1 + e);
            const double2 v = *reinterpret_cast<const double2*>(G2 + e);
            double2 r;
            r.x = a * u.x + b * v.x;
            r.y = a * u.y + b * v.y;
            *reinterpret_cast<double2*>(out + e) = r;
        } else {
            out[e] = a * G1[e] + b * G2[e];
        }
    }
}
}  // namespace accbpg

namespace accbpg {
// packed[r(r+1)/2 + c] <-> G[r*m + c] for c <= r: the significant half of a Gram matrix, as one contiguous message
__global__ __launch_bounds__(256) void tri_pack_kernel(const double* __restrict__ G, int64_t m, double* __restrict__ packed,
                                                      int unpack, double* __restrict__ Gout) {
    const int64_t r = blockIdx.x;
    const int64_t base = r * (r + 1) / 2;
    for (int64_t c = threadIdx.x; c <= r; c += 256) {
        if (unpack) Gout[r * m + c] = packed[base + c];
        else packed[base + c] = G[r * m + c];
    }
}
// count[0] <- number of entries of x that are not >= 0 (NaN counts), as a double so that it can ride at the end of
// a floating-point all-reduce buffer
__global__ __launch_bounds__(1024) void count_bad_kernel(const double* __restrict__ x, int64_t n, double* __restrict__ count) {
    __shared__ int sh[16];
    int c = 0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) c += !(x[i] >= 0.0);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int i = 0; i < 16; ++i) t += sh[i];
        count[0] = (double)t;
    }
}
}  // namespace accbpg

extern "C" int accbpg_tri_pack(const double* G_dev, int64_t m, double* packed_dev, void* stream) {
    if (!G_dev || !packed_dev || m <= 0) return ACCBPG_ERR_ARG;
    tri_pack_kernel<<<(unsigned)m, 256, 0, (hipStream_t)stream>>>(G_dev, m, packed_dev, 0, nullptr);
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

extern "C" int accbpg_tri_unpack(const double* packed_dev, int64_t m, double* G_dev, void* stream) {
    if (!G_dev || !packed_dev || m <= 0) return ACCBPG_ERR_ARG;
    tri_pack_kernel<<<(unsigned)m, 256, 0, (hipStream_t)stream>>>(nullptr, m, const_cast<double*>(packed_dev), 1, G_dev);
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

extern "C" int accbpg_vec_count_bad(const double* x_dev, int64_t n, double* count_dev, void* stream) {
    if (!x_dev || !count_dev || n <= 0) return ACCBPG_ERR_ARG;
    count_bad_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(x_dev, n, count_dev);
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

/* out <- a*G1 + b*G2 over m*m doubles: the Gram matrix is linear in x, so the Gram matrix at
 * a*x1 + b*x2 is this combination of the Gram matrices at x1 and x2 (out may alias G1 or G2). */
extern "C" int accbpg_dopt_gram_lincomb(accbpg_dopt* h, double a, const double* G1_dev, double b,
                                        const double* G2_dev, double* out_dev) {
    if (!h || !G1_dev || !G2_dev || !out_dev) return ACCBPG_ERR_ARG;
    const int64_t total = h->m * h->m;
    lincomb_kernel<<<2048, 256, 0, h->stream>>>(a, G1_dev, b, G2_dev, total, out_dev);
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

/* func_grad from a Gram matrix already formed (by accbpg_dopt_gram or accbpg_dopt_gram_lincomb):
 * Cholesky + log det, and for flag 1/2 the gradient (functions.py:48-58 without :46). */
extern "C" int accbpg_dopt_eval_gram(accbpg_dopt* h, const double* gram_dev, int flag, double* f_host, double* g_dev) {
    if (!h || !gram_dev || flag < 0 || flag > 2) return ACCBPG_ERR_ARG;
    if (flag != 0 && !g_dev) return ACCBPG_ERR_ARG;
    ACC_TRY(launch_cholesky(h, h->Lbuf, flag != 0 ? h->Wbuf : nullptr, nullptr, gram_dev));
    if (flag != 0) {
        ACC_TRY(launch_trtri(h));
        ACC_TRY(launch_colnorm(h, h->Wbuf, g_dev, -1.0));
    }
    ACC_TRY(read_status(h));
    const int* fl = reinterpret_cast<const int*>(h->hpin + 16);
    if (fl[FLAG_ABORT]) {
        if (gram_dev == h->Lbuf) {
            set_last_error("accbpg_dopt_eval_gram: in-place factorisation was abandoned; pass the Gram matrix in a buffer of its own");
            return ACCBPG_ERR_HIP;
        }
        h->chol_tiles_off = true;
        note_tiles_fallback("accbpg_dopt_eval_gram");
        return accbpg_dopt_eval_gram(h, gram_dev, flag, f_host, g_dev);
    }
    if (fl[FLAG_NOT_PD]) {
        set_last_error("HXHT is singular or not positive definite");
        return ACCBPG_ERR_NOT_PD;
    }
    if (f_host) *f_host = -h->hpin[0];
    return ACCBPG_OK;
}

extern "C" int accbpg_dopt_profile_enable(accbpg_dopt* h, int enable) {
    if (!h) return ACCBPG_ERR_ARG;
    h->prof_on = enable != 0;
    return ACCBPG_OK;
}

extern "C" int accbpg_dopt_profile_reset(accbpg_dopt* h) {
    if (!h) return ACCBPG_ERR_ARG;
    ACC_HIP(hipStreamSynchronize(h->stream));
    for (auto& p : h->prof) { p.used = 0; p.total_ms = 0.0; p.launches = 0; }
    return ACCBPG_OK;
}

extern "C" int accbpg_dopt_profile_read(accbpg_dopt* h, int which, double* total_ms_host, int64_t* launches_host) {
    if (!h || which < 0 || which >= PROF_COUNT) return ACCBPG_ERR_ARG;
    ACC_HIP(hipStreamSynchronize(h->stream));
    ProfSlot& p = h->prof[which];
    double tot = 0.0;
    for (size_t i = 0; i + 1 < p.used; i += 2) {
        float ms = 0.f;
        ACC_HIP(hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]));
        tot += ms;
    }
    p.total_ms = tot;
    if (total_ms_host) *total_ms_host = tot;
    if (launches_host) *launches_host = p.launches;
    return ACCBPG_OK;
}

extern "C" int accbpg_mfma_f64_peak(int iters, double* tflops_host, void* stream) {
    if (!tflops_host || iters <= 0) return ACCBPG_ERR_ARG;
    return mfma_peak(iters, tflops_host, (hipStream_t)stream);
}

/* Development probe (see pipe_probe_kernel): milliseconds for `iters` loop trips of 8 fp64 MFMAs (mode bit 0) and / or
 * 128 fp64 vector FMAs (mode bit 1) per wave, one wave per SIMD on every CU. */
extern "C" int accbpg_debug_pipe_probe(int iters, int mode, double* ms_host, void* stream) {
    if (!ms_host || iters <= 0) return ACCBPG_ERR_ARG;
    return pipe_probe(iters, mode, ms_host, (hipStream_t)stream);
}

extern "C" int accbpg_test_gemm(const double* A_dev, int64_t lda, const double* B_dev, int64_t ldb, double* C_dev,
                                int64_t ldc, int64_t M, int64_t N, int64_t K, int b_kmajor, double alpha,
                                double beta, int config, void* stream) {
    if (!A_dev || !B_dev || !C_dev || M <= 0 || N <= 0 || K <= 0) return ACCBPG_ERR_ARG;
    return launch_test_gemm(A_dev, lda, B_dev, ldb, C_dev, ldc, M, N, K, b_kmajor, alpha, beta, config,
                            (hipStream_t)stream);
}

/* A handle whose evaluations run beside another stream's MFMA-bound launches (the value evaluation that the solvers
 * start next to a gradient evaluation) should not fill the chip with the one-launch factorisation's waiting
 * workgroups: with `on` it factors with one launch per block column -- a few workgroups at a time, the same
 * arithmetic, bit-identical results.  Measured at config 2: ABPG_gain 50.6 -> 52.0 it/s in the steady state,
 * 88.8 -> 92.2 in the transient window. */
extern "C" int accbpg_dopt_factor_in_small_launches(accbpg_dopt* h, int on) {
    if (!h || on < 0 || on > 2) return ACCBPG_ERR_ARG;
    // 2 = where it pays: only a factorisation with at least a workgroup per compute unit crowds the other stream out
    // (m > 1408 on 256 CUs); below that the single launch is the faster neighbour too ((512,8192): 1507 against 1464 it/s)
    const int64_t T = (h->m + NB - 1) / NB;
    h->chol_tiles_off = on == 1 || (on == 2 && T * (T + 1) / 2 >= h->num_cu);
    return ACCBPG_OK;
}

/* Development switch for handles created AFTER the call: bit 0 = plain stream-K ranges of the Gram kernel also where
 * there are more tiles than workgroups (m >= 4096), instead of whole tiles per workgroup (A/B measurements). */
extern "C" int accbpg_debug_plan_flags(int flags) {
    set_plan_flags(flags);
    return ACCBPG_OK;
}

extern "C" int accbpg_debug_chol_variant(accbpg_dopt* h, int bits) {
    if (!h) return ACCBPG_ERR_ARG;
    h->chol_dbg = bits & 63;
    h->chol_tiles_off = (bits & 64) != 0;       // bit 6: launch-per-block-column Cholesky instead of the one-launch kernel
    if (bits & 2048) h->chol_tiles_off = true;   // a forced scheme of the launch-per-column kernels
    h->chol_stall_test = (bits & 128) ? 1 : 0;  // bit 7: make the one-launch kernel time out (exercises the redo); short limit
    h->chol_spin_limit = (bits & 128) ? 200000 : 20000000;
    h->kern_variant = (bits >> 30) & 3;       // bits 30..31: schedule of the direct-to-LDS Gram / gradient kernels
    if (bits & 256) h->use_glds = false;      // bit 8: register-staged Gram / gradient kernels
    if (bits & 512) h->use_glds = true;
    if (bits & 2048) {                                                    // bit 11: two-level threshold (block columns)
        h->chol_two_level_T = (bits >> 12) & 0xfff;                       // ... and, if given, the outer panel width
        if ((bits >> 24) & 0x3f) h->chol_nk = (bits >> 24) & 0x3f;
    }
    return ACCBPG_OK;
}

/* Development aid: factor gram_dev once with the one-launch Cholesky while its chain workgroups stamp the 100 MHz
 * wall clock at their stage boundaries; stamps_host receives 32 stamps per 64-wide block column (workgroup start,
 * left updates in, previous factor seen, staged, panel solve done, diagonal tile up to date, factored, published). */
extern "C" int accbpg_debug_chol_trace(accbpg_dopt* h, const double* gram_dev, int with_inverse, int64_t* stamps_host) {
    if (!h || !gram_dev || !stamps_host) return ACCBPG_ERR_ARG;
    if (!chol_tiles_usable(h)) {
        set_last_error("accbpg_debug_chol_trace: the one-launch Cholesky is not in use for this handle");
        return ACCBPG_ERR_ARG;
    }
    const int T = (int)((h->m + NB - 1) / NB);
    const size_t bytes = sizeof(long long) * (size_t)T * 32;
    if (!h->chol_trace) ACC_HIP(hipMalloc(&h->chol_trace, bytes));
    ACC_HIP(hipMemsetAsync(h->chol_trace, 0, bytes, h->stream));
    int rc = launch_cholesky(h, h->Lbuf, with_inverse ? h->Wbuf : nullptr, nullptr, gram_dev);
    if (rc == ACCBPG_OK) {
        hipError_t e = hipMemcpyAsync(stamps_host, h->chol_trace, bytes, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = ACCBPG_ERR_HIP;
    }
    hipFree(h->chol_trace);
    h->chol_trace = nullptr;
    return rc;
}

extern "C" int accbpg_debug_gram_variant(accbpg_dopt* h, const double* x_dev, int variant, int iters, double* ms_host) {
    if (!h || !x_dev || !ms_host || iters <= 0) return ACCBPG_ERR_ARG;
    return debug_gram_variant(h, x_dev, variant, iters, ms_host);
}
