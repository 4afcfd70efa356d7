// Poisson linear inverse problem f(x) = D_KL(b, Ax) on gfx950 (accbpg/functions.py:85-120).
//
// Two passes over the row-major m x n matrix A per func_grad, both HBM-bound:
//   pass 1  Ax = A x, one wavefront (or one workgroup for few, long rows) per row, 16-byte loads,
//           with the per-row epilogue  r_i = 1 - b_i/Ax_i,  t_i = b_i log(b_i/Ax_i) + Ax_i - b_i  fused in;
//   pass 2  g = A^T r on the split-row kernel shared with the Frank-Wolfe update (fw_kernels.hip).
// The value is the fixed-tree sum of t.  Algorithmic traffic: 2 * 8 * m * n bytes per func_grad, 8*m*n for a
// value-only call.  Compiled with -ffp-contract=off so that the epilogue rounds like the NumPy ufunc chain.
#include "internal.h"

struct accbpg_poisson {
    const double* A = nullptr;
    const double* b = nullptr;
    int64_t m = 0, n = 0, lda = 0;
    hipStream_t stream = nullptr;
    int device = 0, num_cu = 256;
    bool vec_ok = false;
    double* Ax = nullptr;      // m
    double* r = nullptr;       // m
    double* t = nullptr;       // m
    double* upart = nullptr;   // VT_MAXSPLIT * n
    double* dout = nullptr;    // device scalar
    double* hpin = nullptr;    // pinned host scalar
    unsigned hold_lds = 0;     // dynamic-LDS request that holds the A x kernel to two workgroups per CU
};

namespace accbpg {

constexpr int QB = 256;

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

__device__ __forceinline__ void row_epilogue(double ax, double bi, int64_t row, double* __restrict__ Ax,
                                             double* __restrict__ r, double* __restrict__ t) {
    const double q = bi / ax;                 // b / Ax            functions.py:107,111
    Ax[row] = ax;
    r[row] = 1.0 - q;
    const double lg = log(q);
    const double p = bi * lg;
    t[row] = (p + ax) - bi;                   // b*log(b/Ax) + Ax - b
}

// TPR threads cooperate on one row (64: a wavefront per row, QB: a workgroup per row)
template <int TPR>
__global__ __launch_bounds__(QB) void poisson_ax_kernel(const double* __restrict__ A, int64_t lda, int64_t m,
                                                       int64_t n, const double* __restrict__ x,
                                                       const double* __restrict__ b, double* __restrict__ Ax,
                                                       double* __restrict__ r, double* __restrict__ t, bool vec_ok) {
    __shared__ double sh[QB / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int sub = (TPR == 64) ? lane : (int)threadIdx.x;
    const int64_t row = (TPR == 64) ? (int64_t)blockIdx.x * (QB / 64) + w : (int64_t)blockIdx.x;
    const bool live = row < m;
    double s0 = 0.0, s1 = 0.0;
    if (live) {
        const double* ar = A + row * lda;
        if (vec_ok) {
            const int64_t n2 = n >> 1;
            const double2* a2 = reinterpret_cast<const double2*>(ar);
            const double2* x2 = reinterpret_cast<const double2*>(x);
#pragma unroll 4
            for (int64_t c = sub; c < n2; c += TPR) {
                // A is streamed once per pass: non-temporal loads keep it out of the caches (x stays cached)
                const double* ap = reinterpret_cast<const double*>(a2 + c);
                const double2 av = double2{__builtin_nontemporal_load(ap), __builtin_nontemporal_load(ap + 1)};
                const double2 xv = x2[c];
                s0 = fma(av.x, xv.x, s0);
                s1 = fma(av.y, xv.y, s1);
            }
            if ((n & 1) && sub == 0) s0 = fma(ar[n - 1], x[n - 1], s0);
        } else {
            for (int64_t c = sub; c < n; c += TPR) s0 = fma(ar[c], x[c], s0);
        }
    }
    double s = wsum(s0 + s1);
    if (TPR == 64) {
        if (live && lane == 0) row_epilogue(s, b[row], row, Ax, r, t);
    } else {
        if (lane == 0) sh[w] = s;
        __syncthreads();
        if (threadIdx.x == 0 && live) {
            double a = 0.0;
            for (int i = 0; i < QB / 64; ++i) a += sh[i];
            row_epilogue(a, b[row], row, Ax, r, t);
        }
    }
}

// fx = sum_i t_i on one workgroup with a fixed tree (reproducible run to run)
__global__ __launch_bounds__(1024) void poisson_fsum_kernel(const double* __restrict__ t, int64_t m,
                                                           double* __restrict__ out) {
    __shared__ double sh[16];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < m; i += 1024) s += t[i];
    s = wsum(s);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int i = 0; i < 16; ++i) a += sh[i];
        out[0] = a;
    }
}

}  // namespace accbpg

using namespace accbpg;

static int poisson_init(accbpg_poisson* h) {
    ACC_HIP(hipGetDevice(&h->device));
    hipDeviceProp_t prop;
    ACC_HIP(hipGetDeviceProperties(&prop, h->device));
    h->num_cu = prop.multiProcessorCount;
    // LDS request that holds the wave-per-row A x kernel to two workgroups per CU: a little over a third of what
    // a CU has (60 KiB of gfx950's 160 KiB), never more than one workgroup may ask for
    const size_t per_cu = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : 65536;
    size_t hold = per_cu * 3 / 8;
    if (hold > prop.sharedMemPerBlock) hold = prop.sharedMemPerBlock;
    h->hold_lds = (unsigned)(hold & ~(size_t)255);
    h->vec_ok = ((reinterpret_cast<uintptr_t>(h->A) & 15) == 0) && ((h->lda & 1) == 0);
    ACC_HIP(hipMalloc(&h->Ax, sizeof(double) * (size_t)h->m));
    ACC_HIP(hipMalloc(&h->r, sizeof(double) * (size_t)h->m));
    ACC_HIP(hipMalloc(&h->t, sizeof(double) * (size_t)h->m));
    ACC_HIP(hipMalloc(&h->upart, sizeof(double) * (size_t)VT_MAXSPLIT * (size_t)h->n));
    ACC_HIP(hipMalloc(&h->dout, sizeof(double) * 4));
    ACC_HIP(hipHostMalloc(&h->hpin, sizeof(double) * 4, hipHostMallocDefault));
    return ACCBPG_OK;
}

extern "C" int accbpg_poisson_destroy(accbpg_poisson* h);

extern "C" int accbpg_poisson_create(const double* A_dev, int64_t m, int64_t n, int64_t lda, const double* b_dev,
                                     void* stream, accbpg_poisson** out) {
    if (!A_dev || !b_dev || !out || m <= 0 || n <= 0 || lda < n) return ACCBPG_ERR_ARG;
    accbpg_poisson* h = new accbpg_poisson();
    h->A = A_dev; h->b = b_dev; h->m = m; h->n = n; h->lda = lda;
    h->stream = (hipStream_t)stream;
    const int rc = poisson_init(h);
    if (rc != ACCBPG_OK) {                  // nothing of a half-built handle stays behind
        accbpg_poisson_destroy(h);
        return rc;
    }
    *out = h;
    return ACCBPG_OK;
}

extern "C" int accbpg_poisson_destroy(accbpg_poisson* h) {
    if (!h) return ACCBPG_OK;
    hipFree(h->Ax); hipFree(h->r); hipFree(h->t); hipFree(h->upart); hipFree(h->dout);
    if (h->hpin) hipHostFree(h->hpin);
    delete h;
    return ACCBPG_OK;
}

extern "C" int accbpg_poisson_set_stream(accbpg_poisson* h, void* stream) {
    if (!h) return ACCBPG_ERR_ARG;
    h->stream = (hipStream_t)stream;
    return ACCBPG_OK;
}

extern "C" int accbpg_poisson_func_grad(accbpg_poisson* h, const double* x_dev, int flag, double* f_host,
                                        double* g_dev) {
    if (!h || !x_dev || flag < 0 || flag > 2) return ACCBPG_ERR_ARG;
    if (flag != 1 && !f_host) return ACCBPG_ERR_ARG;
    if (flag != 0 && !g_dev) return ACCBPG_ERR_ARG;
    hipStream_t s = h->stream;
    const bool xvec = h->vec_ok && ((reinterpret_cast<uintptr_t>(x_dev) & 15) == 0);
    // few long rows: a workgroup per row keeps more of the chip busy than a wavefront per row
    if (h->m < 8 * (int64_t)h->num_cu && h->n >= 4096)
        poisson_ax_kernel<QB><<<(unsigned)h->m, QB, 0, s>>>(h->A, h->lda, h->m, h->n, x_dev, h->b, h->Ax, h->r, h->t,
                                                           xvec);
    else {
        // Long rows: hold the kernel to two workgroups (eight row streams) per CU by asking for 60 KiB of LDS
        // it does not use, and let the dispatcher hand out the remaining rows as workgroups retire.  Measured
        // at (8192,65536): 0.641 ms per pass against 0.683 with every CU filled to its wave limit (and 0.689
        // for a grid-stride loop at the same two workgroups per CU: the dynamic hand-out is what balances
        // the streams; four rows per wavefront sharing each piece of x: 0.667).
        const unsigned hold = (h->n >= 32768) ? h->hold_lds : 0;
        poisson_ax_kernel<64><<<(unsigned)((h->m + QB / 64 - 1) / (QB / 64)), QB, hold, s>>>(
            h->A, h->lda, h->m, h->n, x_dev, h->b, h->Ax, h->r, h->t, xvec);
    }
    if (flag != 1) poisson_fsum_kernel<<<1, 1024, 0, s>>>(h->t, h->m, h->dout);
    ACC_HIP(hipGetLastError());
    if (flag != 0)
        ACC_TRY(launch_vt_times(h->A, h->lda, h->m, h->n, h->r, h->upart, vt_nsplit(h->m, h->n, h->num_cu), g_dev,
                                h->vec_ok, s));
    if (flag != 1) {
        ACC_HIP(hipMemcpyAsync(h->hpin, h->dout, sizeof(double), hipMemcpyDeviceToHost, s));
        ACC_HIP(hipStreamSynchronize(s));
        f_host[0] = h->hpin[0];
    }
    return ACCBPG_OK;
}

/* Ax of the last func_grad (length m), for callers that want the fitted intensities */
extern "C" int accbpg_poisson_get_ax(accbpg_poisson* h, double* out_dev) {
    if (!h || !out_dev) return ACCBPG_ERR_ARG;
    ACC_TRY(device_copy(out_dev, h->Ax, (size_t)h->m, h->stream));
    return ACCBPG_OK;
}
