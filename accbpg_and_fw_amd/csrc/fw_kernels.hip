// Frank-Wolfe / Wolfe-Atwood step kernels for D-optimal design on gfx950.
// Replaces the loop bodies of D_opt_FW and D_opt_FW_away (accbpg/D_opt_alg.py:51-82, 135-179).
// Every kernel here is HBM- or latency-bound: the step reads V once (8*m*n bytes) for
// u = Hv^T V, reads and rewrites the m x m inverse H for the rank-one update, and makes a few
// passes over the length-n vectors.  Compiled with -ffp-contract=off (NumPy ufunc rounding).
#include "internal.h"

namespace accbpg {

constexpr int FB = 256;
constexpr int VG_COLS = 2 * FB;      // columns of V per workgroup in the V^T Hv pass
constexpr int VG_MAXSPLIT = 64;

struct ValIdx {
    double v;
    int64_t i;
};

// first-index-on-ties argmax / argmin, the NumPy convention (D_opt_alg.py:59,61,145,147); a NaN is the
// extremum for both (np.argmax / np.argmin return the first NaN)
__device__ __forceinline__ ValIdx better_max(ValIdx a, ValIdx b) {
    const bool an = a.v != a.v, bn = b.v != b.v;
    const bool take = (bn && !an) || (!an && b.v > a.v) || ((b.v == a.v || (an && bn)) && b.i < a.i);
    return take ? b : a;
}
__device__ __forceinline__ ValIdx better_min(ValIdx a, ValIdx b) {
    const bool an = a.v != a.v, bn = b.v != b.v;
    const bool take = (bn && !an) || (!an && b.v < a.v) || ((b.v == a.v || (an && bn)) && b.i < a.i);
    return take ? b : a;
}
__device__ __forceinline__ ValIdx shfl_down_vi(ValIdx a, int off) {
    ValIdx r;
    r.v = __shfl_down(a.v, off);
    r.i = __shfl_down((long long)a.i, off);
    return r;
}

template <bool IS_MAX, int NT>
__device__ __forceinline__ ValIdx block_reduce_vi_n(ValIdx a, ValIdx* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ValIdx o = shfl_down_vi(a, off);
        a = IS_MAX ? better_max(a, o) : better_min(a, o);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = a;
    __syncthreads();
    ValIdx r = sh[0];
    for (int i = 1; i < NT / 64; ++i) r = IS_MAX ? better_max(r, sh[i]) : better_min(r, sh[i]);
    return r;
}

// Probe, stage 1 (one pass over w and x, many workgroups): per-block first-index argmax of w and
// first-index argmin of w over the support mask (away: x > 1e-8, D_opt_alg.py:147; FW: x > 0, :60).
//
// For the Frank-Wolfe variant the masked minimum of w is all that is needed (only w_j enters eps_neg).  The
// away index j = argmin((w - w_i) * [x > 1e-8]) (:146-147) is taken on the ROUNDED differences, and two
// masked entries with different w can round to the same difference from a much larger w_i, in which case
// NumPy returns the first of them: stage 2 therefore evaluates that expression itself once w_i is known.
__global__ __launch_bounds__(FB) void fw_probe_partial_kernel(const double* __restrict__ w,
                                                             const double* __restrict__ x, int64_t n, int away,
                                                             ValIdx* __restrict__ part) {
    __shared__ ValIdx sh[FB / 64];
    const double inf = __builtin_inf();
    const double thr = away ? 1.0e-8 : 0.0;
    ValIdx best{-inf, INT64_MAX}, lo{inf, INT64_MAX};
    const int64_t stride = (int64_t)gridDim.x * FB;
    for (int64_t k = (int64_t)blockIdx.x * FB + threadIdx.x; k < n; k += stride) {
        const double wk = w[k];
        best = better_max(best, ValIdx{wk, k});
        if (x[k] > thr) lo = better_min(lo, ValIdx{wk, k});
    }
    const ValIdx mx = block_reduce_vi_n<true, FB>(best, sh);
    const ValIdx mn = block_reduce_vi_n<false, FB>(lo, sh);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = mx;
        part[2 * blockIdx.x + 1] = mn;
    }
}

// stage 2: combine the partials (block order), resolve the all-zero case, fetch w_j and x_j
__global__ __launch_bounds__(FB) void fw_probe_final_kernel(const ValIdx* __restrict__ part, int nblk,
                                                           const double* __restrict__ w,
                                                           const double* __restrict__ x, int64_t n, int away,
                                                           double* __restrict__ dout, int64_t* __restrict__ iout,
                                                           const double* __restrict__ qsrc) {
    // (dout / iout point into the handle's PINNED HOST record: the results -- and q of the last update, qsrc -- are
    //  written where the host reads them, no copy operation behind the kernel)
    __shared__ ValIdx sh[FB / 64];
    const double inf = __builtin_inf();
    ValIdx best{-inf, INT64_MAX}, lo{inf, INT64_MAX};
    for (int b = threadIdx.x; b < nblk; b += FB) {
        best = better_max(best, part[2 * b]);
        lo = better_min(lo, part[2 * b + 1]);
    }
    const ValIdx mx = block_reduce_vi_n<true, FB>(best, sh);
    ValIdx mn = block_reduce_vi_n<false, FB>(lo, sh);
    if (away) {
        // d_k = (w_k - w_i) * [x_k > 1e-8] exactly as the reference forms it (one subtraction, one product with
        // 1.0 or 0.0), first index of its minimum.  One workgroup over 2n doubles: a few microseconds at the
        // sizes the away variant runs at, next to a factorisation of H per iteration.
        ValIdx dm{inf, INT64_MAX};
        for (int64_t k = threadIdx.x; k < n; k += FB) {
            const double diff = w[k] - mx.v;
            const double dk = diff * ((x[k] > 1.0e-8) ? 1.0 : 0.0);
            dm = better_min(dm, ValIdx{dk, k});
        }
        mn = block_reduce_vi_n<false, FB>(dm, sh);
    }
    if (threadIdx.x == 0) {
        const int64_t j = mn.i;
        iout[0] = mx.i;
        iout[1] = j;
        dout[0] = mx.v;
        const bool ok = j >= 0 && j < n;
        dout[1] = ok ? w[j] : inf;
        dout[2] = ok ? x[j] : 0.0;
        dout[6] = *qsrc;
    }
}

// Away variant, stage 2 on many workgroups (one workgroup scanning 2n doubles took 40 us of a 170 us step at n = 32768):
// every workgroup combines the stage-1 maxima itself (at most 512 records: the same w_i everywhere), then takes the
// first-index minimum of d_k = (w_k - w_i) * [x_k > 1e-8] -- formed exactly as the reference forms it, D_opt_alg.py:146-147
// -- over its slice; workgroup 0 leaves (w_i, i) for the final stage.
constexpr int AWAY_NB = 128;        // at most this many slices
__global__ __launch_bounds__(FB) void fw_probe_away_partial_kernel(const ValIdx* __restrict__ part, int nblk,
                                                                  const double* __restrict__ w,
                                                                  const double* __restrict__ x, int64_t n,
                                                                  ValIdx* __restrict__ part2, ValIdx* __restrict__ mxslot) {
    __shared__ ValIdx sh[FB / 64];
    const double inf = __builtin_inf();
    ValIdx best{-inf, INT64_MAX};
    for (int b = threadIdx.x; b < nblk; b += FB) best = better_max(best, part[2 * b]);
    const ValIdx mx = block_reduce_vi_n<true, FB>(best, sh);
    ValIdx dm{inf, INT64_MAX};
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t k0 = (int64_t)blockIdx.x * per, k1 = min(n, k0 + per);
    for (int64_t k = k0 + threadIdx.x; k < k1; k += FB) {
        const double diff = w[k] - mx.v;
        const double dk = diff * ((x[k] > 1.0e-8) ? 1.0 : 0.0);
        dm = better_min(dm, ValIdx{dk, k});
    }
    const ValIdx mn = block_reduce_vi_n<false, FB>(dm, sh);
    if (threadIdx.x == 0) {
        part2[blockIdx.x] = mn;
        if (blockIdx.x == 0) *mxslot = mx;
    }
}
__global__ __launch_bounds__(FB) void fw_probe_away_final_kernel(const ValIdx* __restrict__ part2, int nb2,
                                                                const ValIdx* __restrict__ mxslot,
                                                                const double* __restrict__ w,
                                                                const double* __restrict__ x, int64_t n,
                                                                double* __restrict__ dout, int64_t* __restrict__ iout,
                                                                const double* __restrict__ qsrc) {
    __shared__ ValIdx sh[FB / 64];
    const double inf = __builtin_inf();
    ValIdx dm{inf, INT64_MAX};
    for (int b = threadIdx.x; b < nb2; b += FB) dm = better_min(dm, part2[b]);
    const ValIdx mn = block_reduce_vi_n<false, FB>(dm, sh);
    if (threadIdx.x == 0) {
        const ValIdx mx = *mxslot;
        const int64_t j = mn.i;
        iout[0] = mx.i;
        iout[1] = j;
        dout[0] = mx.v;
        const bool ok = j >= 0 && j < n;
        dout[1] = ok ? w[j] : inf;
        dout[2] = ok ? x[j] : 0.0;
        dout[6] = *qsrc;
    }
}

// x <- x*xscale; x[p] += xadd   (D_opt_alg.py:76-77,164-165,173-174); vp <- V[:,p]
__global__ __launch_bounds__(FB) void fw_xupdate_gather_kernel(double* __restrict__ x, int64_t n, int64_t p,
                                                              double xscale, double xadd,
                                                              const double* __restrict__ V, int64_t ldv, int64_t m,
                                                              double* __restrict__ vp) {
    const int64_t stride = (int64_t)gridDim.x * FB;
    for (int64_t k = (int64_t)blockIdx.x * FB + threadIdx.x; k < n; k += stride) {
        double v = x[k] * xscale;
        if (k == p) v += xadd;
        x[k] = v;
    }
    for (int64_t r = (int64_t)blockIdx.x * FB + threadIdx.x; r < m; r += stride) vp[r] = V[r * ldv + p];
}

// hv = H vp, one wave per row of H          (np.dot(H, V[:,i]), D_opt_alg.py:78,166,175)
__global__ __launch_bounds__(FB) void fw_gemv_h_kernel(const double* __restrict__ H, int64_t m,
                                                      const double* __restrict__ vp, double* __restrict__ hv) {
    // one wave per PAIR of rows (twice the loads in flight per wave; the per-row summation order is that of one
    // row per wave)
    const int lane = threadIdx.x & 63;
    const int64_t row = 2 * ((int64_t)blockIdx.x * (FB / 64) + (threadIdx.x >> 6));
    if (row >= m) return;
    const bool two = row + 1 < m;
    const double* hr = H + row * m;
    const double* hq = two ? hr + m : hr;
    double s = 0.0, t = 0.0;
    if ((m & 1) == 0) {
        // 16-byte loads, four independent accumulators per row (H rows are 16-byte aligned when m is even)
        const double2* h2 = reinterpret_cast<const double2*>(hr);
        const double2* q2 = reinterpret_cast<const double2*>(hq);
        const double2* v2 = reinterpret_cast<const double2*>(vp);
        const int64_t n2 = m >> 1;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
        int64_t c = lane;
        for (; c + 64 < n2; c += 128) {
            const double2 a = h2[c], b = h2[c + 64], p = q2[c], q = q2[c + 64], x = v2[c], y = v2[c + 64];
            s0 = fma(a.x, x.x, s0); s1 = fma(a.y, x.y, s1);
            s2 = fma(b.x, y.x, s2); s3 = fma(b.y, y.y, s3);
            t0 = fma(p.x, x.x, t0); t1 = fma(p.y, x.y, t1);
            t2 = fma(q.x, y.x, t2); t3 = fma(q.y, y.y, t3);
        }
        for (; c < n2; c += 64) {
            const double2 a = h2[c], p = q2[c], x = v2[c];
            s0 = fma(a.x, x.x, s0); s1 = fma(a.y, x.y, s1);
            t0 = fma(p.x, x.x, t0); t1 = fma(p.y, x.y, t1);
        }
        s = (s0 + s1) + (s2 + s3);
        t = (t0 + t1) + (t2 + t3);
    } else {
        for (int64_t c = lane; c < m; c += 64) {
            s = fma(hr[c], vp[c], s);
            t = fma(hq[c], vp[c], t);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off);
        t += __shfl_down(t, off);
    }
    if (lane == 0) {
        hv[row] = s;
        if (two) hv[row + 1] = t;
    }
}

// H <- (H + hcoef*outer(hv,hv)) / hdiv       (D_opt_alg.py:79,167,176); a row per workgroup pass
// Workgroup 0 also leaves q = v_p^T (H v_p), the quadratic form of the pivot column in the inverse AS MAINTAINED (the
// tracked w_p drifts away from it: w is never refreshed, D_opt_alg.py:82): by the matrix determinant lemma
// det(H+) = det(H) (1 + hcoef q) / hdiv^m holds for exactly this q, which is what the log-space advance of log det(H)
// between two factorisations uses (accbpg_fw_probe: q_prev).
__global__ __launch_bounds__(FB) void fw_rank1_kernel(double* __restrict__ H, int64_t m,
                                                     const double* __restrict__ hv, double hcoef, double hdiv,
                                                     const double* __restrict__ vp, double* __restrict__ qout) {
    if (blockIdx.x == 0) {
        __shared__ double qs[FB / 64];
        double q = 0.0;
        for (int64_t c = threadIdx.x; c < m; c += FB) q = fma(vp[c], hv[c], q);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off);
        if ((threadIdx.x & 63) == 0) qs[threadIdx.x >> 6] = q;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = qs[0];
            for (int i = 1; i < FB / 64; ++i) t += qs[i];
            *qout = t;
        }
    }
    for (int64_t r = blockIdx.x; r < m; r += gridDim.x) {
        const double hr = hv[r];
        double* row = H + r * m;
        if ((m & 1) == 0) {
            double2* row2 = reinterpret_cast<double2*>(row);
            const double2* hv2 = reinterpret_cast<const double2*>(hv);
            for (int64_t c = threadIdx.x; c < (m >> 1); c += FB) {
                double2 a = row2[c];
                const double2 b = hv2[c];
                const double o0 = hr * b.x, o1 = hr * b.y;
                const double t0 = hcoef * o0, t1 = hcoef * o1;
                a.x = (a.x + t0) / hdiv;
                a.y = (a.y + t1) / hdiv;
                row2[c] = a;
            }
        } else {
            for (int64_t c = threadIdx.x; c < m; c += FB) {
                const double o = hr * hv[c];
                const double t = hcoef * o;
                row[c] = (row[c] + t) / hdiv;
            }
        }
    }
}

// partial u[s][k] = sum over the rows of split s of hv[r]*V[r][k]; two columns per thread
__global__ __launch_bounds__(FB) void fw_vgemv_partial_kernel(const double* __restrict__ V, int64_t ldv, int64_t m,
                                                             int64_t n, const double* __restrict__ hv, int nsplit,
                                                             double* __restrict__ upart, bool vec_ok) {
    const int64_t k = (int64_t)blockIdx.x * VG_COLS + 2 * threadIdx.x;
    const int s = blockIdx.y;
    const int64_t rows_per = (m + nsplit - 1) / nsplit;
    const int64_t r0 = (int64_t)s * rows_per, r1 = min(m, r0 + rows_per);
    if (k >= n) return;
    double a0 = 0.0, a1 = 0.0;
    const bool pair = vec_ok && (k + 1 < n);
    if (pair) {
        const double* vp = V + r0 * ldv + k;
#pragma unroll 8
        for (int64_t r = r0; r < r1; ++r) {
            // V is streamed once per pass (512 MiB at config 3): non-temporal loads keep it out of the caches
            const double2 v = double2{__builtin_nontemporal_load(vp), __builtin_nontemporal_load(vp + 1)};
            const double h = hv[r];
            a0 = fma(h, v.x, a0);
            a1 = fma(h, v.y, a1);
            vp += ldv;
        }
    } else {
        for (int64_t r = r0; r < r1; ++r) {
            const double h = hv[r];
            a0 = fma(h, V[r * ldv + k], a0);
            if (k + 1 < n) a1 = fma(h, V[r * ldv + k + 1], a1);
        }
    }
    upart[(int64_t)s * n + k] = a0;
    if (k + 1 < n) upart[(int64_t)s * n + k + 1] = a1;
}

// w <- (w + hcoef * u^2) / hdiv with u = sum of the partials     (D_opt_alg.py:82,170,179)
__global__ __launch_bounds__(FB) void fw_wupdate_kernel(double* __restrict__ w, int64_t n,
                                                       const double* __restrict__ upart, int nsplit, double hcoef,
                                                       double hdiv) {
    const int64_t stride = (int64_t)gridDim.x * FB;
    for (int64_t k = (int64_t)blockIdx.x * FB + threadIdx.x; k < n; k += stride) {
        double u = 0.0;
        for (int s = 0; s < nsplit; ++s) u += upart[(int64_t)s * n + k];
        const double sq = u * u;
        const double t = hcoef * sq;
        w[k] = (w[k] + t) / hdiv;
    }
}

// The same update fused with stage 1 of the NEXT iteration's probe (fw_probe_partial_kernel): the new w
// is in registers anyway, x was updated earlier in this step.
__global__ __launch_bounds__(FB) void fw_wupdate_probe_kernel(double* __restrict__ w, int64_t n,
                                                             const double* __restrict__ upart, int nsplit,
                                                             double hcoef, double hdiv,
                                                             const double* __restrict__ x, int away,
                                                             ValIdx* __restrict__ part) {
    __shared__ ValIdx sh[FB / 64];
    const double inf = __builtin_inf();
    const double thr = away ? 1.0e-8 : 0.0;
    ValIdx best{-inf, INT64_MAX}, lo{inf, INT64_MAX};
    const int64_t stride = (int64_t)gridDim.x * FB;
    for (int64_t k = (int64_t)blockIdx.x * FB + threadIdx.x; k < n; k += stride) {
        double u = 0.0;
        for (int s = 0; s < nsplit; ++s) u += upart[(int64_t)s * n + k];
        const double sq = u * u;
        const double t = hcoef * sq;
        const double wk = (w[k] + t) / hdiv;
        w[k] = wk;
        best = better_max(best, ValIdx{wk, k});
        if (x[k] > thr) lo = better_min(lo, ValIdx{wk, k});
    }
    const ValIdx mx = block_reduce_vi_n<true, FB>(best, sh);
    const ValIdx mn = block_reduce_vi_n<false, FB>(lo, sh);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = mx;
        part[2 * blockIdx.x + 1] = mn;
    }
}

// u = sum of the row-split partials (u = V^T q)
__global__ __launch_bounds__(FB) void fw_usum_kernel(const double* __restrict__ upart, int nsplit, int64_t n,
                                                    double* __restrict__ u) {
    const int64_t stride = (int64_t)gridDim.x * FB;
    for (int64_t k = (int64_t)blockIdx.x * FB + threadIdx.x; k < n; k += stride) {
        double a = 0.0;
        for (int s = 0; s < nsplit; ++s) a += upart[(int64_t)s * n + k];
        u[k] = a;
    }
}
__global__ __launch_bounds__(FB) void column_kernel(const double* __restrict__ V, int64_t ldv, int64_t m, int64_t j,
                                                   double* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * FB;
    for (int64_t r = (int64_t)blockIdx.x * FB + threadIdx.x; r < m; r += stride) out[r] = V[r * ldv + j];
}

// out = in^T for an m x m matrix (used once per fw_init to form H = W^T W on the MFMA engine)
__global__ __launch_bounds__(FB) void transpose_kernel(const double* __restrict__ in, double* __restrict__ out,
                                                      int64_t m) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8)
        tile[j][tx] = (r0 + j < m && c0 + tx < m) ? in[(r0 + j) * m + c0 + tx] : 0.0;
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < m && r0 + tx < m) out[(c0 + j) * m + r0 + tx] = tile[tx][j];
}

// symmetrise: copy the lower triangle of H onto the upper one
__global__ __launch_bounds__(FB) void mirror_lower_kernel(double* __restrict__ H, int64_t m) {
    const int64_t total = m * m;
    const int64_t stride = (int64_t)gridDim.x * FB;
    for (int64_t e = (int64_t)blockIdx.x * FB + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / m, c = e - r * m;
        if (c > r) H[e] = H[c * m + r];
    }
}

}  // namespace accbpg

using namespace accbpg;

static int fw_alloc(accbpg_dopt* h) {
    if (h->fw_x) return ACCBPG_OK;
    ACC_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->fw_hpin_dev), h->hpin, 0));   // the pinned record, as the device addresses it
    ACC_HIP(hipMalloc(&h->fw_x, sizeof(double) * h->n));
    ACC_HIP(hipMalloc(&h->fw_w, sizeof(double) * h->n));
    ACC_HIP(hipMalloc(&h->fw_H, sizeof(double) * h->m * h->m));
    // Hv, vp, 2 * 512 stage-1 probe records, AWAY_NB stage-2 records + (w_i, i) of the away variant
    ACC_HIP(hipMalloc(&h->fw_hv, sizeof(double) * (2 * h->m + 2 * 1024 + 16 + 2 * AWAY_NB + 4)));
    return ACCBPG_OK;
}

namespace accbpg { int vt_nsplit(int64_t m, int64_t n, int num_cu); }
static int fw_nsplit(const accbpg_dopt* h) { return vt_nsplit(h->m, h->n, h->num_cu); }

static int read_scalars(accbpg_dopt* h, int nd, int ni) {
    (void)nd; (void)ni;
    ACC_HIP(hipMemcpyAsync(h->hpin, h->dscal, sizeof(double) * STATUS_DOUBLES, hipMemcpyDeviceToHost, h->stream));   // scalars + flags
    ACC_HIP(hipStreamSynchronize(h->stream));
    return ACCBPG_OK;
}

static int fw_ring_drain(accbpg_dopt* h);

extern "C" int accbpg_fw_init(accbpg_dopt* h, const double* x0_dev, double* logdet_gram_host) {
    if (!h || !x0_dev) return ACCBPG_ERR_ARG;
    ACC_TRY(fw_ring_drain(h));                                 // factorisations left in flight by an earlier run
    ACC_TRY(fw_alloc(h));
    const int64_t m = h->m;
    ACC_TRY(device_copy(h->fw_x, x0_dev, (size_t)h->n, h->stream));
    double* gram = chol_tiles_usable(h) ? h->Gbuf : h->Lbuf;
    ACC_TRY(launch_gram(h, h->fw_x, gram));                    // D_opt_alg.py:40
    ACC_TRY(launch_cholesky(h, h->Lbuf, nullptr, nullptr, gram));   // det / inv via the Cholesky factor (:41-42)
    ACC_TRY(read_scalars(h, 1, 4));
    const int* fl = reinterpret_cast<const int*>(h->hpin + 16);
    if (fl[FLAG_ABORT] && !h->chol_tiles_off) {                // one-launch factorisation abandoned: launch per block column
        h->chol_tiles_off = true;
        return accbpg_fw_init(h, x0_dev, logdet_gram_host);
    }
    if (fl[FLAG_NOT_PD]) {
        set_last_error("accbpg_fw_init: V diag(x0) V^T is singular or not positive definite");
        return ACCBPG_ERR_NOT_PD;
    }
    if (logdet_gram_host) *logdet_gram_host = h->hpin[0];
    ACC_TRY(launch_trtri(h));                                  // W = L^-1
    ACC_TRY(launch_colnorm(h, h->Wbuf, h->fw_w, 1.0));         // w_i = |W v_i|^2 = v_i^T H v_i (:45)
    // H = W^T W: transpose W into Tbuf (rows of W^T are k-contiguous), then one product
    dim3 tg((unsigned)((m + 31) / 32), (unsigned)((m + 31) / 32));
    transpose_kernel<<<tg, FB, 0, h->stream>>>(h->Wbuf, h->Tbuf, m);
    GemmOp op{};
    op.A = h->Tbuf; op.lda = m; op.B = h->Wbuf; op.ldb = m; op.C = h->fw_H; op.ldc = m;
    op.M = (int)m; op.N = (int)m; op.K = (int)m; op.lower_only = 1; op.alpha = 1.0; op.beta = 0.0;
    ACC_HIP(hipMemcpyAsync(h->chol_op, &op, sizeof(GemmOp), hipMemcpyHostToDevice, h->stream));
    ACC_TRY(launch_gemm_ops(h->chol_op, 1, (int)m, (int)m, true, h->stream));
    mirror_lower_kernel<<<1024, FB, 0, h->stream>>>(h->fw_H, m);
    ACC_HIP(hipMemsetAsync(h->dscal + 10, 0, sizeof(double), h->stream));   // q of "the previous update": none yet
    ACC_HIP(hipGetLastError());
    ACC_HIP(hipStreamSynchronize(h->stream));
    h->fw_ready = true;
    h->fw_part_nblk = 0;
    return ACCBPG_OK;
}

// ---- log det(H) beside the steps (refresh_logdet = 2) ----------------------------------------------------------
// F[k] = log det(H_k) of the away-step variant (D_opt_alg.py:136) is a LOGGED value: no decision of the iteration
// reads it.  So its O(m^3) factorisation need not sit between two HBM-bound steps: H_k is copied into a slot of a ring
// (one copy kernel on the main stream, in order with the updates) and factored on that slot's own stream by that
// slot's own auxiliary handle -- own buffers, own hand-off flags, own scalars -- while the main stream probes, decides
// and applies the updates that follow; the value is collected when the slot comes round again, `depth` such calls
// later (or by accbpg_fw_logdet_flush).  Same kernels on the same matrix as the synchronous form (refresh_logdet =
// 1): the numbers are those.  Nothing of the slots is shared with the main handle, so an evaluation on the main
// handle while factorisations are in flight is safe.
static int fw_ring_setup(accbpg_dopt* h) {
    while ((int)h->fw_ring.size() < h->fw_ring_depth) {
        accbpg_dopt::FwSlot sl;
        ACC_HIP(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        ACC_HIP(hipEventCreateWithFlags(&sl.ev_snap, hipEventDisableTiming));
        const int rc = accbpg_dopt_create(h->V, h->m, h->n, h->ldv, sl.stream, &sl.aux, 1);
        if (rc != ACCBPG_OK) {
            hipStreamDestroy(sl.stream);
            hipEventDestroy(sl.ev_snap);
            return rc;
        }
        h->fw_ring.push_back(sl);
    }
    for (auto& sl : h->fw_ring) {                              // (development switches of the main handle apply to its slots)
        sl.aux->chol_stall_test = h->chol_stall_test;
        sl.aux->chol_spin_limit = h->chol_spin_limit;
    }
    return ACCBPG_OK;
}

static int fw_slot_factor(accbpg_dopt::FwSlot& sl, const double* snap) {
    accbpg_dopt* a = sl.aux;
    ACC_TRY(launch_cholesky(a, a->Lbuf, nullptr, nullptr, snap));
    ACC_HIP(hipMemcpyAsync(a->hpin, a->dscal, sizeof(double) * STATUS_DOUBLES, hipMemcpyDeviceToHost, a->stream));
    ACC_HIP(hipEventRecord(a->ev_done, a->stream));
    return ACCBPG_OK;
}

// wait for the slot's factorisation and return its log det (NaN if the matrix was not positive definite)
static int fw_slot_collect(accbpg_dopt::FwSlot& sl, double* logdet) {
    *logdet = __builtin_nan("");
    if (!sl.pending) return ACCBPG_OK;
    accbpg_dopt* a = sl.aux;
    ACC_HIP(hipEventSynchronize(a->ev_done));
    const int* fl = reinterpret_cast<const int*>(a->hpin + 16);
    if (fl[FLAG_ABORT] && !a->chol_tiles_off) {
        // the one-launch factorisation gave up a wait: the snapshot (in Gbuf) is intact, factor it with a launch per
        // block column, and keep this slot on those
        a->chol_tiles_off = true;
        note_tiles_fallback("accbpg_fw_probe_step (side factorisation)");
        ACC_TRY(fw_slot_factor(sl, a->Gbuf));
        ACC_HIP(hipEventSynchronize(a->ev_done));
    }
    sl.pending = false;
    if (!fl[FLAG_NOT_PD]) *logdet = a->hpin[0];
    return ACCBPG_OK;
}

static int fw_ring_drain(accbpg_dopt* h) {
    for (auto& sl : h->fw_ring) {
        double unused;
        ACC_TRY(fw_slot_collect(sl, &unused));
    }
    h->fw_ring_collected = h->fw_ring_issued;
    return ACCBPG_OK;
}

extern "C" int accbpg_fw_logdet_ring(accbpg_dopt* h, int depth, int small_launches) {
    if (!h || depth < 1 || depth > 16 || small_launches < 0 || small_launches > 2) return ACCBPG_ERR_ARG;
    ACC_TRY(fw_ring_drain(h));
    h->fw_ring_depth = depth;
    h->fw_ring_small = small_launches;
    for (auto& sl : h->fw_ring) sl.aux->chol_tiles_off = false;   // (a slot that had to give up the one launch finds out again)
    h->fw_ring_issued = h->fw_ring_collected = 0;              // slot = issue count modulo depth
    return ACCBPG_OK;
}

extern "C" int accbpg_fw_logdet_pending(accbpg_dopt* h) {
    return h ? (int)(h->fw_ring_issued - h->fw_ring_collected) : 0;
}

// the oldest factorisation in flight (NaN when there is none)
extern "C" int accbpg_fw_logdet_flush(accbpg_dopt* h, double* logdet_host) {
    if (!h || !logdet_host) return ACCBPG_ERR_ARG;
    *logdet_host = __builtin_nan("");
    if (h->fw_ring_collected >= h->fw_ring_issued) return ACCBPG_OK;
    accbpg_dopt::FwSlot& sl = h->fw_ring[(size_t)(h->fw_ring_collected % h->fw_ring_depth)];
    ACC_TRY(fw_slot_collect(sl, logdet_host));
    ++h->fw_ring_collected;
    return ACCBPG_OK;
}

extern "C" int accbpg_fw_probe_step(accbpg_dopt* h, int away, int refresh_logdet, accbpg_fw_probe* out) {
    if (!h || !out || !h->fw_ready) return ACCBPG_ERR_ARG;
    double logdet = __builtin_nan("");
    if (refresh_logdet == 2) {
        ACC_TRY(fw_ring_setup(h));
        // the slot that comes round: collect what it holds -- the value of the call made `depth` such calls ago
        if (h->fw_ring_issued - h->fw_ring_collected >= h->fw_ring_depth) ACC_TRY(accbpg_fw_logdet_flush(h, &logdet));
        accbpg_dopt::FwSlot& sl = h->fw_ring[(size_t)(h->fw_ring_issued % h->fw_ring_depth)];
        accbpg_dopt* a = sl.aux;
        const int64_t T = (h->m + NB - 1) / NB;
        if (!a->chol_tiles_off)
            a->chol_tiles_off = h->fw_ring_small == 1 || (h->fw_ring_small == 2 && T * (T + 1) / 2 >= h->num_cu);
        double* snap = chol_tiles_usable(a) ? a->Gbuf : a->Lbuf;
        ACC_TRY(device_copy(snap, h->fw_H, (size_t)h->m * h->m, h->stream));   // H_k, in order with the updates
        ACC_HIP(hipEventRecord(sl.ev_snap, h->stream));
        ACC_HIP(hipStreamWaitEvent(sl.stream, sl.ev_snap, 0));
        ACC_TRY(fw_slot_factor(sl, snap));
        sl.pending = true;
        ++h->fw_ring_issued;
    } else if (refresh_logdet) {
        // F[k] = log det(H) from a fresh factorisation of the maintained inverse (D_opt_alg.py:136)
        // (the factor goes to Lbuf; H itself is read in place by the one-launch kernel, copied otherwise)
        ACC_TRY(launch_cholesky(h, h->Lbuf, nullptr, nullptr, h->fw_H));
        ACC_TRY(read_scalars(h, 1, 4));
        const int* fl = reinterpret_cast<const int*>(h->hpin + 16);
        if (fl[FLAG_ABORT] && !h->chol_tiles_off) {
            h->chol_tiles_off = true;
            ACC_TRY(launch_cholesky(h, h->Lbuf, nullptr, nullptr, h->fw_H));
            ACC_TRY(read_scalars(h, 1, 4));
        }
        logdet = h->hpin[0];
        if (fl[FLAG_NOT_PD]) logdet = __builtin_nan("");
    }
    double* dout = h->fw_hpin_dev + 4;
    int64_t* iout = reinterpret_cast<int64_t*>(h->fw_hpin_dev + 8);
    int nblk = (int)((h->n + (int64_t)FB * 8 - 1) / ((int64_t)FB * 8));
    if (nblk < 1) nblk = 1;
    if (nblk > 512) nblk = 512;
    ValIdx* part = reinterpret_cast<ValIdx*>(h->fw_hv + 2 * h->m);      // 2*512 records behind Hv / vp
    if (h->fw_part_nblk > 0 && h->fw_part_away == (away != 0)) {
        nblk = h->fw_part_nblk;                                 // stage 1 came with the last update of w
    } else {
        fw_probe_partial_kernel<<<nblk, FB, 0, h->stream>>>(h->fw_w, h->fw_x, h->n, away, part);
    }
    h->fw_part_nblk = 0;
    h->fw_part_away = (away != 0);
    if (away && h->n >= 4096) {
        ValIdx* part2 = part + 2 * 512;
        ValIdx* mxslot = part2 + AWAY_NB;
        int nb2 = (int)((h->n + (int64_t)FB * 8 - 1) / ((int64_t)FB * 8));
        if (nb2 > AWAY_NB) nb2 = AWAY_NB;
        fw_probe_away_partial_kernel<<<nb2, FB, 0, h->stream>>>(part, nblk, h->fw_w, h->fw_x, h->n, part2, mxslot);
        fw_probe_away_final_kernel<<<1, FB, 0, h->stream>>>(part2, nb2, mxslot, h->fw_w, h->fw_x, h->n, dout, iout, h->dscal + 10);
    } else {
        fw_probe_final_kernel<<<1, FB, 0, h->stream>>>(part, nblk, h->fw_w, h->fw_x, h->n, away, dout, iout, h->dscal + 10);
    }
    ACC_HIP(hipGetLastError());
    // (the final stage wrote its record straight into the pinned host buffer: the copy operation that used to follow it
    //  was 4 us of every 0.12 ms step)
    ACC_HIP(hipStreamSynchronize(h->stream));
    const int64_t* ih = reinterpret_cast<const int64_t*>(h->hpin + 8);
    out->i = ih[0];
    out->j = ih[1];
    out->w_i = h->hpin[4];
    out->w_j = h->hpin[5];
    out->x_j = h->hpin[6];
    out->logdet_H = logdet;
    out->q_prev = h->hpin[10];
    return ACCBPG_OK;
}

extern "C" int accbpg_fw_update(accbpg_dopt* h, int64_t p, double xscale, double xadd, double hcoef, double hdiv) {
    if (!h || !h->fw_ready || p < 0 || p >= h->n) {
        set_last_error("accbpg_fw_update: no Frank-Wolfe state (call accbpg_fw_init) or pivot index %lld outside [0, n)",
                       (long long)p);
        return ACCBPG_ERR_ARG;
    }
    const int64_t m = h->m, n = h->n;
    double* vp = h->fw_hv + m;
    int64_t gb = (std::max(n, m) + FB - 1) / FB;
    if (gb > 1024) gb = 1024;
    fw_xupdate_gather_kernel<<<(int)gb, FB, 0, h->stream>>>(h->fw_x, n, p, xscale, xadd, h->V, h->ldv, m, vp);
    const int64_t pairs = (m + 1) / 2;
    fw_gemv_h_kernel<<<(int)((pairs + FB / 64 - 1) / (FB / 64)), FB, 0, h->stream>>>(h->fw_H, m, vp, h->fw_hv);
    int64_t rb = m;
    if (rb > 4096) rb = 4096;
    fw_rank1_kernel<<<(int)rb, FB, 0, h->stream>>>(h->fw_H, m, h->fw_hv, hcoef, hdiv, vp, h->dscal + 10);
    const int ns = fw_nsplit(h);
    dim3 vg((unsigned)((n + VG_COLS - 1) / VG_COLS), (unsigned)ns);
    prof_begin(h, PROF_FWV);
    fw_vgemv_partial_kernel<<<vg, FB, 0, h->stream>>>(h->V, h->ldv, m, n, h->fw_hv, ns, h->vws, h->vec_ok);
    prof_end(h, PROF_FWV);
    // w update fused with stage 1 of the next probe (same support threshold as the last probe call)
    int64_t wb = (n + FB - 1) / FB;
    if (wb > 512) wb = 512;                                     // 2*512 probe records behind Hv / vp
    ValIdx* part = reinterpret_cast<ValIdx*>(h->fw_hv + 2 * h->m);
    fw_wupdate_probe_kernel<<<(int)wb, FB, 0, h->stream>>>(h->fw_w, n, h->vws, ns, hcoef, hdiv, h->fw_x,
                                                          h->fw_part_away ? 1 : 0, part);
    h->fw_part_nblk = (int)wb;
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

extern "C" int accbpg_fw_get_state(accbpg_dopt* h, double* x_dev, double* w_dev, double* H_dev) {
    if (!h || !h->fw_ready) return ACCBPG_ERR_ARG;
    if (x_dev) ACC_TRY(device_copy(x_dev, h->fw_x, (size_t)h->n, h->stream));
    if (w_dev) ACC_TRY(device_copy(w_dev, h->fw_w, (size_t)h->n, h->stream));
    if (H_dev)
        ACC_TRY(device_copy(H_dev, h->fw_H, (size_t)h->m * h->m, h->stream));
    ACC_HIP(hipStreamSynchronize(h->stream));
    return ACCBPG_OK;
}

namespace accbpg {
int vt_nsplit(int64_t m, int64_t n, int num_cu) {
    const int64_t colblocks = (n + VG_COLS - 1) / VG_COLS;
    // About 0.8 workgroups per CU in total, NOT several per CU: measured on MI355X, the pass streams fastest
    // when every CU follows one row range (config 3, 64 column blocks: 3 splits = 192 workgroups 124 us per
    // FW step, 4 splits 130, 16 splits 143; Poisson (8192,65536), 128 column blocks: 2 splits 0.835 of the
    // HBM peak, 8 splits 0.815, 1 split 0.76) -- fewer concurrent DRAM streams, and fewer partials to sum.
    int64_t s = (4 * (int64_t)num_cu / 5 + colblocks / 2) / colblocks;
    if (s > VG_MAXSPLIT) s = VG_MAXSPLIT;
    if (s > m / 8) s = m / 8;
    if (s < 1) s = 1;
    return (int)s;
}

// u = V^T q for a row-major m x n matrix: split-row partial sums into upart (nsplit*n doubles), then their sum
int launch_vt_times(const double* V, int64_t ldv, int64_t m, int64_t n, const double* q, double* upart, int nsplit,
                    double* u, bool vec_ok, hipStream_t s) {
    dim3 vg((unsigned)((n + VG_COLS - 1) / VG_COLS), (unsigned)nsplit);
    fw_vgemv_partial_kernel<<<vg, FB, 0, s>>>(V, ldv, m, n, q, nsplit, upart, vec_ok);
    int64_t wb = (n + FB - 1) / FB;
    if (wb > 2048) wb = 2048;
    fw_usum_kernel<<<(int)wb, FB, 0, s>>>(upart, nsplit, n, u);
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}
}  // namespace accbpg

/* u = V^T q  (np.dot(q, V), applications.py:79): one pass over V on the split-row kernel */
extern "C" int accbpg_dopt_vt_times(accbpg_dopt* h, const double* q_dev, double* u_dev) {
    if (!h || !q_dev || !u_dev) return ACCBPG_ERR_ARG;
    return launch_vt_times(h->V, h->ldv, h->m, h->n, q_dev, h->vws, fw_nsplit(h), u_dev, h->vec_ok, h->stream);
}

extern "C" int accbpg_dopt_get_column(accbpg_dopt* h, int64_t j, double* out_dev) {
    if (!h || !out_dev || j < 0 || j >= h->n) return ACCBPG_ERR_ARG;
    int64_t gb = (h->m + FB - 1) / FB;
    if (gb > 1024) gb = 1024;
    column_kernel<<<(int)gb, FB, 0, h->stream>>>(h->V, h->ldv, h->m, j, out_dev);
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}
