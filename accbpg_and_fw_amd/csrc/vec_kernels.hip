// Length-n kernels of the Bregman proximal gradient loop on gfx950: Burg-entropy simplex prox
// (the whole bisection + Newton scalar solve in one launch), Bregman divergences, the line-search
// inner product, and the a*x + b*z combination.  HBM/latency-bound: coalesced 8-byte-per-lane
// streams, wavefront (64-lane) shuffle reductions, a fixed reduction tree so that results are
// reproducible run to run.
//
// This translation unit is compiled with -ffp-contract=off: the reference evaluates these
// expressions with separate NumPy ufuncs (one rounding per operation), so no multiply-add here
// may be fused.  Replaces accbpg/functions.py:246-271 and 336-356.
#include <mutex>

#include "internal.h"

namespace accbpg {

constexpr int PB = 1024;   // threads of the single-workgroup prox kernel
constexpr int RB = 256;    // threads of the streaming reductions
constexpr int RMAXBLK = 1024;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}
// minimum that keeps a NaN (np.min does; fmin would drop it and a NaN would slip through the reference's
// `x.min() > 0` assertions, functions.py:252)
__device__ __forceinline__ double min_nan(double a, double b) { return (b < a || b != b) ? b : a; }
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min_nan(v, __shfl_down(v, off));
    return v;
}

// sum over the PB threads of the prox workgroup, result broadcast to every thread
__device__ __forceinline__ double block_sum_bcast(double v, double* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < PB / 64; ++i) s += sh[i];
    return s;
}
__device__ __forceinline__ double block_min_bcast(double v, double* sh) {
    v = wave_min(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double s = sh[0];
#pragma unroll
    for (int i = 1; i < PB / 64; ++i) s = min_nan(s, sh[i]);
    return s;
}

// Burg simplex prox, functions.py:336-356 (and :264-271 when y != NULL).
// EPT > 0: gg lives in registers (n <= PB*EPT); EPT == 0: gg is kept in `ggbuf` and re-read.
template <int EPT>
__device__ __forceinline__ void burg_prox_body(const double* __restrict__ y, const double* __restrict__ g,
                                               double L, double eps, int64_t n, double* __restrict__ xout,
                                               double* __restrict__ ggbuf, int* __restrict__ info,
                                               int* __restrict__ flags) {
    __shared__ double sh[PB / 64];
    const int tid = threadIdx.x;
    constexpr int R = EPT > 0 ? EPT : 1;
    double gg[R];
    const double inf = __builtin_inf();
    double lmin = inf;
    bool bad = false;

    auto make_gg = [&](int64_t i) -> double {
        double a = g[i];
        if (y != nullptr) {
            const double yi = y[i];
            if (!(yi > 0.0)) bad = true;
            const double t = -1.0 / yi;       // h.gradient(y)      functions.py:248
            a = a - L * t;                    // g - L*grad         functions.py:271
        }
        return a / L;                         // gg = g / L         functions.py:341
    };
    if constexpr (EPT > 0) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int64_t i = tid + (int64_t)PB * e;
            gg[e] = (i < n) ? make_gg(i) : inf;
            lmin = min_nan(lmin, gg[e]);
        }
    } else {
        for (int64_t i = tid; i < n; i += PB) {
            const double v = make_gg(i);
            ggbuf[i] = v;
            lmin = min_nan(lmin, v);
        }
    }
    const double cmin = -block_min_bcast(lmin, sh);       // functions.py:342

    auto phi = [&](double c) -> double {                   // sum(1/(gg+c)) - 1
        double s = 0.0;
        if constexpr (EPT > 0) {
#pragma unroll
            for (int e = 0; e < EPT; ++e) s += 1.0 / (gg[e] + c);
        } else {
            for (int64_t i = tid; i < n; i += PB) s += 1.0 / (ggbuf[i] + c);
        }
        return block_sum_bcast(s, sh) - 1.0;
    };
    auto dphi = [&](double c) -> double {                  // sum(-1/(gg+c)^2)
        double s = 0.0;
        if constexpr (EPT > 0) {
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const double r = gg[e] + c;
                s += -1.0 / (r * r);
            }
        } else {
            for (int64_t i = tid; i < n; i += PB) {
                const double r = ggbuf[i] + c;
                s += -1.0 / (r * r);
            }
        }
        return block_sum_bcast(s, sh);
    };

    double c = cmin + 1.0;                                 // functions.py:344
    int nb = 0, nn = 0;
    double fc = phi(c);
    while (fc < 0.0 && nb < 4096) {                        // functions.py:345-346
        c = (cmin + c) / 2.0;
        fc = phi(c);
        ++nb;
    }
    while (fabs(fc) > eps && nn < 4096) {                  // functions.py:349
        const double fpc = dphi(c);                        // :350
        const double step = fc / fpc;
        if ((c - (c - step)) == 0.0) break;                // :351-352
        c = c - step;                                      // :353
        fc = phi(c);                                       // :354
        ++nn;
    }
    if constexpr (EPT > 0) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int64_t i = tid + (int64_t)PB * e;
            if (i < n) xout[i] = 1.0 / (gg[e] + c);        // :355
        }
    } else {
        for (int64_t i = tid; i < n; i += PB) xout[i] = 1.0 / (ggbuf[i] + c);
    }
    // the status is written unconditionally (no memset in front of the launch)
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    if (tid == 0) {
        flags[FLAG_NOT_PD] = 0; flags[FLAG_NEG_X] = 0; flags[FLAG_NONPOS] = any_bad; flags[FLAG_BAD_G] = 0;
        info[0] = nb;
        info[1] = nn;
    }
}

template <int EPT>
__global__ __launch_bounds__(PB) void burg_prox_kernel(const double* __restrict__ y, const double* __restrict__ g,
                                                      double L, double eps, int64_t n, double* __restrict__ xout,
                                                      double* __restrict__ ggbuf, int* __restrict__ info,
                                                      int* __restrict__ flags) {
    burg_prox_body<EPT>(y, g, L, eps, n, xout, ggbuf, info, flags);
}
// the prox of every active instance of a batch in one launch: a workgroup per instance (row `instance` of the K x n
// arrays), its own constant L, its own status words
template <int EPT>
__global__ __launch_bounds__(PB) void burg_prox_batch_kernel(BatchAct act, const double* __restrict__ ybase,
                                                            const double* __restrict__ gbase, int64_t ld, BatchVals Ls,
                                                            double eps, int64_t n, double* __restrict__ obase,
                                                            double* __restrict__ ggbase, int* __restrict__ flags) {
    const int inst = act.idx[blockIdx.x];
    burg_prox_body<EPT>(ybase ? ybase + (int64_t)inst * ld : nullptr, gbase + (int64_t)inst * ld, Ls.v[inst], eps, n,
                        obase + (int64_t)inst * ld, ggbase ? ggbase + (int64_t)inst * n : nullptr, flags + 8 * inst + 4,
                        flags + 8 * inst);
}

// ---------------------------------------------------------------------------------------------------------------
// The same prox over SEVERAL workgroups for long vectors (n > 32768: BASELINE config 5 has n = 262144, replicated on
// every rank): each workgroup keeps its slice of gg in registers and runs the scalar loop itself; the two sums of a
// Newton step (phi and phi' at the same c: the reference evaluates phi at the new c and then phi' at that c) travel in
// ONE exchange: every workgroup publishes its two partial sums (write-through stores, then an epoch flag), waits for the
// flags of all others, and adds the partials in workgroup order -- so every workgroup obtains bit-identical totals and
// takes identical decisions, with no further communication.  Exchange r uses slot set r & 1 (a workgroup can only be
// one exchange ahead of the slowest).  Every wait is bounded: on a timeout the launch raises flags[4] and the host
// reruns the single-workgroup kernel.  All workgroups must be resident together (at most 128 of 1024 threads).
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) double pm_gdouble;
typedef __attribute__((address_space(1))) int pm_gint;
struct ProxMultiShared {
    double* part;      // [2][G][2] partial sums
    int* epoch;        // [G] last exchange each workgroup has published (1-based)
    int* abortw;
};
template <int OP>      // 0: sums of both values, 1: minimum of the first
__device__ __forceinline__ bool pm_exchange(const ProxMultiShared sh, int G, int round, double& a, double& b, double* lds,
                                            long long limit) {
    const int w = blockIdx.x;
    double* slot = sh.part + ((size_t)(round & 1) * G + w) * 2;
    if (threadIdx.x == 0) {
        __hip_atomic_store((pm_gdouble*)slot, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((pm_gdouble*)(slot + 1), b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store((pm_gint*)(sh.epoch + w), round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int ok = 1;
    if (threadIdx.x < 64) {
        // lane l waits for the workgroups l, l + 64, ...
        const long long t0 = wall_clock64();
        unsigned spins = 0;
        for (int v = threadIdx.x; v < G; v += 64) {
            while (__hip_atomic_load((const pm_gint*)(sh.epoch + v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round + 1) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 31u) == 0u) {
                    if (__hip_atomic_load((const pm_gint*)sh.abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
                    if (wall_clock64() - t0 > limit) {
                        __hip_atomic_store((pm_gint*)sh.abortw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = 0;
                        break;
                    }
                }
            }
            if (!ok) break;
        }
        ok = __all(ok);
        if (ok && threadIdx.x == 0) {
            // one lane adds the partials in workgroup order (every workgroup does the same sum)
            const double* base = sh.part + (size_t)(round & 1) * G * 2;
            double sa = __hip_atomic_load((const pm_gdouble*)base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            double sb = __hip_atomic_load((const pm_gdouble*)(base + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int v = 1; v < G; ++v) {
                const double pa = __hip_atomic_load((const pm_gdouble*)(base + 2 * v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double pb = __hip_atomic_load((const pm_gdouble*)(base + 2 * v + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (OP == 0) { sa += pa; sb += pb; }
                else { sa = min_nan(sa, pa); }
            }
            lds[0] = sa; lds[1] = sb;
        }
        if (threadIdx.x == 0) lds[2] = ok ? 1.0 : 0.0;
    }
    __syncthreads();
    a = lds[0]; b = lds[1];
    const bool good = lds[2] != 0.0;
    __syncthreads();
    return good;
}

template <int EPT>
__global__ __launch_bounds__(PB) void burg_prox_multi_kernel(const double* __restrict__ y, const double* __restrict__ g,
                                                            double L, double eps, int64_t n, double* __restrict__ xout,
                                                            ProxMultiShared shm, int G, int* __restrict__ info,
                                                            int* __restrict__ flags, long long limit) {
    __shared__ double sh[PB / 64];
    __shared__ double xch[4];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * PB * EPT;
    double gg[EPT];
    const double inf = __builtin_inf();
    double lmin = inf;
    bool bad = false;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int64_t i = base + tid + (int64_t)PB * e;
        double v = inf;
        if (i < n) {
            double a = g[i];
            if (y != nullptr) {
                const double yi = y[i];
                if (!(yi > 0.0)) bad = true;
                const double t = -1.0 / yi;       // h.gradient(y)      functions.py:248
                a = a - L * t;                    // g - L*grad         functions.py:271
            }
            v = a / L;                            // gg = g / L         functions.py:341
        }
        gg[e] = v;
        lmin = min_nan(lmin, v);
    }
    int round = 0;
    double m0 = block_min_bcast(lmin, sh), m1 = 0.0;
    if (!pm_exchange<1>(shm, G, round++, m0, m1, xch, limit)) return;
    const double cmin = -m0;                                   // functions.py:342
    // phi(c) = sum 1/(gg+c) - 1 and phi'(c) = sum -1/(gg+c)^2, both at the same c, in one exchange
    auto both = [&](double c, double& fc, double& fpc) -> bool {
        double s = 0.0, d = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const double r = gg[e] + c;
            s += 1.0 / r;
            d += -1.0 / (r * r);
        }
        s = block_sum_bcast(s, sh);
        d = block_sum_bcast(d, sh);
        if (!pm_exchange<0>(shm, G, round++, s, d, xch, limit)) return false;
        fc = s - 1.0;
        fpc = d;
        return true;
    };
    double c = cmin + 1.0;                                     // functions.py:344
    int nb = 0, nn = 0;
    double fc, fpc;
    if (!both(c, fc, fpc)) return;
    while (fc < 0.0 && nb < 4096) {                            // functions.py:345-346
        c = (cmin + c) / 2.0;
        if (!both(c, fc, fpc)) return;
        ++nb;
    }
    while (fabs(fc) > eps && nn < 4096) {                      // functions.py:349
        const double step = fc / fpc;                          // :350 (phi' at the current c came with phi)
        if ((c - (c - step)) == 0.0) break;                    // :351-352
        c = c - step;                                          // :353
        if (!both(c, fc, fpc)) return;                         // :354
        ++nn;
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int64_t i = base + tid + (int64_t)PB * e;
        if (i < n) xout[i] = 1.0 / (gg[e] + c);                // :355
    }
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    if (tid == 0) {
        if (any_bad) atomicOr(flags + FLAG_NONPOS, 1);
        if (blockIdx.x == 0) { info[0] = nb; info[1] = nn; }
    }
}

// Streaming reduction, stage 1.  Up to four sums / minima per pass:
//   q0 = sum g*(x-y)                                  (np.dot(g, x1-x), algorithms.py:53)
//   q1 = sum x/y - log(x/y) - 1                       (functions.py:253)
//   q2 = sum z/z1 - log(z/z1) - 1
//   q3 = min over every vector that enters a divergence (positivity assert, functions.py:252)
__device__ __forceinline__ void ls_terms_partial_body(const double* __restrict__ g,
                                                      const double* __restrict__ x,
                                                      const double* __restrict__ y,
                                                      const double* __restrict__ z,
                                                      const double* __restrict__ z1, int64_t n,
                                                      int want_div_xy, double* __restrict__ part) {
    __shared__ double sh[4][RB / 64];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, mn = __builtin_inf();
    const int64_t stride = (int64_t)gridDim.x * RB;
    for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += stride) {
        if (x != nullptr && y != nullptr) {
            const double xi = x[i], yi = y[i];
            if (g != nullptr) {
                const double d = xi - yi;
                s0 += g[i] * d;
            }
            if (want_div_xy) {
                const double r = xi / yi;
                s1 += r - log(r) - 1.0;
                mn = min_nan(mn, min_nan(xi, yi));
            }
        }
        if (z != nullptr) {
            const double zi = z[i], wi = z1[i];
            const double r = zi / wi;
            s2 += r - log(r) - 1.0;
            mn = min_nan(mn, min_nan(zi, wi));
        }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); mn = wave_min(mn);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sh[0][w] = s0; sh[1][w] = s1; sh[2][w] = s2; sh[3][w] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0, c = 0.0, d = sh[3][0];
        for (int i = 0; i < RB / 64; ++i) { a += sh[0][i]; b += sh[1][i]; c += sh[2][i]; d = min_nan(d, sh[3][i]); }
        part[blockIdx.x * 4 + 0] = a; part[blockIdx.x * 4 + 1] = b;
        part[blockIdx.x * 4 + 2] = c; part[blockIdx.x * 4 + 3] = d;
    }
}

__global__ __launch_bounds__(RB) void ls_terms_partial_kernel(const double* __restrict__ g,
                                                             const double* __restrict__ x,
                                                             const double* __restrict__ y,
                                                             const double* __restrict__ z,
                                                             const double* __restrict__ z1, int64_t n,
                                                             int want_div_xy, double* __restrict__ part) {
    ls_terms_partial_body(g, x, y, z, z1, n, want_div_xy, part);
}
// the same pass for every active instance of a batch (blockIdx.y); partials of instance i at part + i * pstride
__global__ __launch_bounds__(RB) void ls_terms_partial_batch_kernel(BatchAct act, const double* __restrict__ g,
                                                                   const double* __restrict__ x,
                                                                   const double* __restrict__ y,
                                                                   const double* __restrict__ z,
                                                                   const double* __restrict__ z1, int64_t ld, int64_t n,
                                                                   double* __restrict__ part, int64_t pstride) {
    const int64_t o = (int64_t)act.idx[blockIdx.y] * ld;
    ls_terms_partial_body(g ? g + o : nullptr, x + o, y + o, z ? z + o : nullptr, z1 ? z1 + o : nullptr, n, 1,
                          part + (int64_t)act.idx[blockIdx.y] * pstride);
}

// stage 2: one workgroup adds the partials in block order
__device__ __forceinline__ void ls_terms_final_body(const double* __restrict__ part, int nblk,
                                                    double* __restrict__ out) {
    __shared__ double sh[4][RB / 64];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, mn = __builtin_inf();
    for (int b = threadIdx.x; b < nblk; b += RB) {
        s0 += part[b * 4 + 0]; s1 += part[b * 4 + 1]; s2 += part[b * 4 + 2];
        mn = min_nan(mn, part[b * 4 + 3]);
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); mn = wave_min(mn);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sh[0][w] = s0; sh[1][w] = s1; sh[2][w] = s2; sh[3][w] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0, c = 0.0, d = sh[3][0];
        for (int i = 0; i < RB / 64; ++i) { a += sh[0][i]; b += sh[1][i]; c += sh[2][i]; d = min_nan(d, sh[3][i]); }
        out[0] = a; out[1] = b; out[2] = c; out[3] = d;
    }
}

__global__ __launch_bounds__(RB) void ls_terms_final_kernel(const double* __restrict__ part, int nblk,
                                                           double* __restrict__ out) {
    ls_terms_final_body(part, nblk, out);
}
__global__ __launch_bounds__(RB) void ls_terms_final_batch_kernel(BatchAct act, const double* __restrict__ part,
                                                                 int64_t pstride, int nblk, double* __restrict__ out) {
    const int inst = act.idx[blockIdx.x];
    ls_terms_final_body(part + (int64_t)inst * pstride, nblk, out + 4 * inst);
}

// out = a*x + b*z for every active instance of a batch (its own a and b), NumPy's rounding as in axpby_kernel
__global__ __launch_bounds__(RB) void axpby_batch_kernel(BatchAct act, BatchVals a, const double* __restrict__ x, BatchVals b,
                                                        const double* __restrict__ z, int64_t ld, int64_t n,
                                                        double* __restrict__ out) {
    const int inst = act.idx[blockIdx.y];
    const double ai = a.v[inst], bi = b.v[inst];
    const int64_t o = (int64_t)inst * ld;
    const int64_t stride = (int64_t)gridDim.x * RB;
    for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += stride) {
        const double p = ai * x[o + i];
        const double q = bi * z[o + i];
        out[o + i] = p + q;
    }
}

__global__ __launch_bounds__(RB) void min_sum_partial_kernel(const double* __restrict__ x, int64_t n,
                                                            double* __restrict__ part) {
    __shared__ double sh[2][RB / 64];
    double s = 0.0, mn = __builtin_inf();
    const int64_t stride = (int64_t)gridDim.x * RB;
    for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += stride) {
        const double v = x[i];
        s += v;
        mn = min_nan(mn, v);
    }
    s = wave_sum(s); mn = wave_min(mn);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { sh[0][w] = s; sh[1][w] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, d = sh[1][0];
        for (int i = 0; i < RB / 64; ++i) { a += sh[0][i]; d = min_nan(d, sh[1][i]); }
        part[blockIdx.x * 4 + 0] = a; part[blockIdx.x * 4 + 1] = 0.0;
        part[blockIdx.x * 4 + 2] = 0.0; part[blockIdx.x * 4 + 3] = d;
    }
}

// out = a*x + b*z with NumPy's rounding of (1-theta)*x + theta*z   (algorithms.py:147,150,369,374)
__global__ __launch_bounds__(RB) void axpby_kernel(double a, const double* __restrict__ x, double b,
                                                  const double* __restrict__ z, int64_t n,
                                                  double* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * RB;
    for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += stride) {
        const double p = a * x[i];
        const double q = b * z[i];
        out[i] = p + q;
    }
}

// out = x / d elementwise (NumPy true division: gavg/csum, algorithms.py:485)
__global__ __launch_bounds__(RB) void div_scalar_kernel(const double* __restrict__ x, double d, int64_t n,
                                                       double* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * RB;
    for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += stride) out[i] = x[i] / d;
}

// out[i] = fill, out[idx] = value  (lmo_simplex vertex, functions_lmo.py:152-157)
__global__ __launch_bounds__(RB) void vertex_kernel(int64_t idx, double value, double fill, int64_t n,
                                                   double* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * RB;
    for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += stride) out[i] = (i == idx) ? value : fill;
}

// sum x*y  (np.dot(x, x) of BurgEntropyL2.extra_Psi, functions.py:314)
__global__ __launch_bounds__(RB) void dot_partial_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                        int64_t n, double* __restrict__ part) {
    __shared__ double sh[RB / 64];
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * RB;
    for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += stride) s += x[i] * y[i];
    s = wave_sum(s);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int i = 0; i < RB / 64; ++i) a += sh[i];
        part[blockIdx.x * 4 + 0] = a; part[blockIdx.x * 4 + 1] = 0.0;
        part[blockIdx.x * 4 + 2] = 0.0; part[blockIdx.x * 4 + 3] = 0.0;
    }
}

// Closed-form Burg-entropy prox maps on x > 0 with NumPy's operation order:
//   gt = g - L*(-1/y)  when y != NULL (BurgEntropy.div_prox_map, functions.py:264-271), else gt = g
//   kind 0: L / gt                                     (BurgEntropy.prox_map,   :255-262, needs gt > 0)
//   kind 1: L / (lamda + gt)                           (BurgEntropyL1.prox_map, :290-298, needs gt > -lamda)
//   kind 2: (sqrt(gg*gg + 4*lamda_L) - gg)/(2*lamda_L) (BurgEntropyL2.prox_map, :316-323), gg = gt/L, lamda_L = lamda/L
__global__ __launch_bounds__(RB) void burg_reg_prox_kernel(int kind, const double* __restrict__ y,
                                                          const double* __restrict__ g, double L, double lamda,
                                                          int64_t n, double* __restrict__ out,
                                                          int* __restrict__ flags) {
    const int64_t stride = (int64_t)gridDim.x * RB;
    const double lamda_L = lamda / L;
    const double four_l = 4 * lamda_L, two_l = 2 * lamda_L;
    bool bad_y = false, bad_g = false;
    for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += stride) {
        double gt = g[i];
        if (y != nullptr) {
            const double yi = y[i];
            bad_y |= !(yi > 0.0);
            const double hy = -1.0 / yi;
            const double t = L * hy;
            gt = gt - t;
        }
        double r;
        if (kind == 0) {
            bad_g |= !(gt > 0.0);
            r = L / gt;
        } else if (kind == 1) {
            bad_g |= !(gt > -lamda);
            r = L / (lamda + gt);
        } else {
            const double gg = gt / L;
            const double sq = gg * gg;
            r = (sqrt(sq + four_l) - gg) / two_l;
        }
        out[i] = r;
    }
    if (bad_y) flags[FLAG_NONPOS] = 1;
    if (bad_g) flags[FLAG_BAD_G] = 1;
}

struct MinMaxRec {
    double vmin, vmax;
    int64_t imin, imax;
};
__device__ __forceinline__ MinMaxRec mm_merge(MinMaxRec a, MinMaxRec b) {
    // np.argmin / np.argmax: a NaN is the extremum; among equals (or among NaNs) the first index wins
    MinMaxRec r = a;
    const bool an = a.vmin != a.vmin, bn = b.vmin != b.vmin;
    if ((bn && !an) || (!an && b.vmin < a.vmin) || ((b.vmin == a.vmin || (an && bn)) && b.imin < a.imin)) {
        r.vmin = b.vmin; r.imin = b.imin;
    }
    const bool ax = a.vmax != a.vmax, bx = b.vmax != b.vmax;
    if ((bx && !ax) || (!ax && b.vmax > a.vmax) || ((b.vmax == a.vmax || (ax && bx)) && b.imax < a.imax)) {
        r.vmax = b.vmax; r.imax = b.imax;
    }
    return r;
}
__device__ __forceinline__ MinMaxRec mm_block(MinMaxRec a, MinMaxRec* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        MinMaxRec o;
        o.vmin = __shfl_down(a.vmin, off); o.vmax = __shfl_down(a.vmax, off);
        o.imin = __shfl_down((long long)a.imin, off); o.imax = __shfl_down((long long)a.imax, off);
        a = mm_merge(a, o);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = a;
    __syncthreads();
    MinMaxRec r = sh[0];
    for (int i = 1; i < RB / 64; ++i) r = mm_merge(r, sh[i]);
    return r;
}
// first-index argmin and argmax (np.argmin / np.argmax / np.where(g == g.min())[0][0])
__global__ __launch_bounds__(RB) void minmax_partial_kernel(const double* __restrict__ x, int64_t n,
                                                           MinMaxRec* __restrict__ part) {
    __shared__ MinMaxRec sh[RB / 64];
    const double inf = __builtin_inf();
    MinMaxRec a{inf, -inf, INT64_MAX, INT64_MAX};
    const int64_t stride = (int64_t)gridDim.x * RB;
    for (int64_t i = (int64_t)blockIdx.x * RB + threadIdx.x; i < n; i += stride) {
        const double v = x[i];
        a = mm_merge(a, MinMaxRec{v, v, i, i});
    }
    a = mm_block(a, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = a;
}
__global__ __launch_bounds__(RB) void minmax_final_kernel(const MinMaxRec* __restrict__ part, int nblk,
                                                         MinMaxRec* __restrict__ out) {
    __shared__ MinMaxRec sh[RB / 64];
    const double inf = __builtin_inf();
    MinMaxRec a{inf, -inf, INT64_MAX, INT64_MAX};
    for (int b = threadIdx.x; b < nblk; b += RB) a = mm_merge(a, part[b]);
    a = mm_block(a, sh);
    if (threadIdx.x == 0) *out = a;
}

// -----------------------------------------------------------------------------------------
// Scratch of the handle-free entry points: one block per (host thread, device) -- instances of a batch are
// driven from separate threads on separate streams, and one thread may drive several GPUs.  A block goes back
// to a per-device pool when its thread ends and is handed to the next thread that asks, so generations of
// worker threads do not grow the allocation; the pool itself lives as long as the process.
struct ScratchBlock {
    double* pin = nullptr;       // pinned host scratch (32 doubles)
    int* flags = nullptr;        // device flags (8 ints)
    double* out = nullptr;       // device result scalars (16 doubles)
};
constexpr int MAX_DEV = 64;
static std::mutex g_pool_mu;
static std::vector<ScratchBlock> g_pool[MAX_DEV];
struct ThreadScratch {
    ScratchBlock blk[MAX_DEV];
    ~ThreadScratch() {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (int d = 0; d < MAX_DEV; ++d)
            if (blk[d].pin) g_pool[d].push_back(blk[d]);
    }
};
static thread_local ThreadScratch g_tls;
static thread_local double* g_pin = nullptr;       // the block of the device current at the last ensure_scratch()
static thread_local int* g_flags = nullptr;
static thread_local double* g_out = nullptr;

static int ensure_scratch() {
    int dev = 0;
    ACC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEV) return ACCBPG_ERR_ARG;
    ScratchBlock& b = g_tls.blk[dev];
    if (!b.pin) {
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            if (!g_pool[dev].empty()) { b = g_pool[dev].back(); g_pool[dev].pop_back(); }
        }
        if (!b.pin) {
            ScratchBlock nb;
            ACC_HIP(hipHostMalloc(&nb.pin, 32 * sizeof(double), hipHostMallocDefault));
            if (hipMalloc(&nb.flags, 8 * sizeof(int)) != hipSuccess || hipMalloc(&nb.out, 16 * sizeof(double)) != hipSuccess) {
                hipHostFree(nb.pin); hipFree(nb.flags); hipFree(nb.out);
                set_last_error("accbpg: out of device memory for the vector-kernel scratch");
                return ACCBPG_ERR_HIP;
            }
            b = nb;
        }
    }
    g_pin = b.pin; g_flags = b.flags; g_out = b.out;
    return ACCBPG_OK;
}

static bool g_prox_multi_off = false;    // set when the multi-workgroup prox had to give up a wait (then: one workgroup)

int64_t vec_ws_doubles(int64_t n) { return n + 4 * RMAXBLK + 64; }

static int red_blocks(int64_t n) {
    int64_t b = (n + (int64_t)RB * 4 - 1) / ((int64_t)RB * 4);
    if (b < 1) b = 1;
    if (b > RMAXBLK) b = RMAXBLK;
    return (int)b;
}

}  // namespace accbpg

using namespace accbpg;

extern "C" int64_t accbpg_vec_workspace_doubles(int64_t n) { return vec_ws_doubles(n); }

extern "C" int accbpg_burg_simplex_div_prox(const double* y_dev, const double* g_dev, double L, double eps,
                                            int64_t n, double* x_out_dev, double* ws_dev, int* info_host,
                                            void* stream) {
    if (!g_dev || !x_out_dev || n <= 0 || !ws_dev) return ACCBPG_ERR_ARG;
    if (!(L > 0.0)) return ACCBPG_ERR_ASSERT;                 // functions.py:270 / :340
    ACC_TRY(ensure_scratch());
    hipStream_t s = (hipStream_t)stream;
    int* g_info = g_flags + 4;                                  // {bisection, newton} right behind the flags
    if (n > (int64_t)PB * 32 && n <= (int64_t)PB * 32 * 128 && !g_prox_multi_off) {
        // long vectors: several workgroups, each with its slice in registers (ws_dev: exchange slots, then the flags)
        const bool wide = n > (int64_t)PB * 8 * 128;
        const int ept = wide ? 32 : 8;
        const int G = (int)((n + (int64_t)PB * ept - 1) / ((int64_t)PB * ept));
        ProxMultiShared shm;
        shm.part = ws_dev;
        shm.epoch = reinterpret_cast<int*>(ws_dev + 4 * G);
        shm.abortw = shm.epoch + G;
        ACC_HIP(hipMemsetAsync(ws_dev, 0, sizeof(double) * 4 * G + sizeof(int) * (G + 4), s));
        ACC_HIP(hipMemsetAsync(g_flags, 0, 8 * sizeof(int), s));
        if (wide)
            burg_prox_multi_kernel<32><<<G, PB, 0, s>>>(y_dev, g_dev, L, eps, n, x_out_dev, shm, G, g_info, g_flags, 20000000LL);
        else
            burg_prox_multi_kernel<8><<<G, PB, 0, s>>>(y_dev, g_dev, L, eps, n, x_out_dev, shm, G, g_info, g_flags, 20000000LL);
        ACC_HIP(hipGetLastError());
        int* pin_m = reinterpret_cast<int*>(g_pin);
        ACC_HIP(hipMemcpyAsync(pin_m, g_flags, 6 * sizeof(int), hipMemcpyDeviceToHost, s));
        ACC_HIP(hipMemcpyAsync(pin_m + 8, shm.abortw, sizeof(int), hipMemcpyDeviceToHost, s));
        ACC_HIP(hipStreamSynchronize(s));
        if (pin_m[8] == 0) {
            if (info_host) { info_host[0] = pin_m[4]; info_host[1] = pin_m[5]; }
            if (pin_m[FLAG_NONPOS]) return ACCBPG_ERR_ASSERT;     // y.min() > 0, functions.py:270
            return ACCBPG_OK;
        }
        g_prox_multi_off = true;                                // a wait timed out: one workgroup from here on
    }
    if (n <= (int64_t)PB * 2)
        burg_prox_kernel<2><<<1, PB, 0, s>>>(y_dev, g_dev, L, eps, n, x_out_dev, ws_dev, g_info, g_flags);
    else if (n <= (int64_t)PB * 8)
        burg_prox_kernel<8><<<1, PB, 0, s>>>(y_dev, g_dev, L, eps, n, x_out_dev, ws_dev, g_info, g_flags);
    else if (n <= (int64_t)PB * 32)
        burg_prox_kernel<32><<<1, PB, 0, s>>>(y_dev, g_dev, L, eps, n, x_out_dev, ws_dev, g_info, g_flags);
    else
        burg_prox_kernel<0><<<1, PB, 0, s>>>(y_dev, g_dev, L, eps, n, x_out_dev, ws_dev, g_info, g_flags);
    ACC_HIP(hipGetLastError());
    int* pin_i = reinterpret_cast<int*>(g_pin);
    ACC_HIP(hipMemcpyAsync(pin_i, g_flags, 6 * sizeof(int), hipMemcpyDeviceToHost, s));   // flags + info, one copy
    ACC_HIP(hipStreamSynchronize(s));
    if (info_host) { info_host[0] = pin_i[4]; info_host[1] = pin_i[5]; }
    if (pin_i[FLAG_NONPOS]) return ACCBPG_ERR_ASSERT;         // y.min() > 0, functions.py:270
    return ACCBPG_OK;
}

extern "C" int accbpg_ls_terms(const double* g_dev, const double* x_dev, const double* y_dev, const double* z_dev,
                               const double* z1_dev, int64_t n, double* out_host, double* ws_dev, void* stream) {
    if (n <= 0 || !out_host || !ws_dev) return ACCBPG_ERR_ARG;
    if ((z_dev == nullptr) != (z1_dev == nullptr)) return ACCBPG_ERR_ARG;
    ACC_TRY(ensure_scratch());
    hipStream_t s = (hipStream_t)stream;
    const int nb = red_blocks(n);
    double* part = ws_dev + n;
    ls_terms_partial_kernel<<<nb, RB, 0, s>>>(g_dev, x_dev, y_dev, z_dev, z1_dev, n, 1, part);
    ls_terms_final_kernel<<<1, RB, 0, s>>>(part, nb, g_out);
    ACC_HIP(hipGetLastError());
    ACC_HIP(hipMemcpyAsync(g_pin + 8, g_out, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
    ACC_HIP(hipStreamSynchronize(s));
    out_host[0] = g_pin[8]; out_host[1] = g_pin[9]; out_host[2] = g_pin[10];
    if (!(g_pin[11] > 0.0)) return ACCBPG_ERR_ASSERT;         // functions.py:252
    return ACCBPG_OK;
}

extern "C" int accbpg_burg_divergence(const double* x_dev, const double* y_dev, int64_t n, double* out_host,
                                      double* ws_dev, void* stream) {
    if (!x_dev || !y_dev) return ACCBPG_ERR_ARG;
    double o[3];
    int rc = accbpg_ls_terms(nullptr, x_dev, y_dev, nullptr, nullptr, n, o, ws_dev, stream);
    if (out_host) out_host[0] = o[1];
    return rc;
}

extern "C" int accbpg_vec_dot_diff(const double* g_dev, const double* x_dev, const double* y_dev, int64_t n,
                                   double* out_host, double* ws_dev, void* stream) {
    if (!g_dev || !x_dev || !y_dev || n <= 0 || !out_host || !ws_dev) return ACCBPG_ERR_ARG;
    ACC_TRY(ensure_scratch());
    hipStream_t s = (hipStream_t)stream;
    const int nb = red_blocks(n);
    double* part = ws_dev + n;
    ls_terms_partial_kernel<<<nb, RB, 0, s>>>(g_dev, x_dev, y_dev, nullptr, nullptr, n, 0, part);
    ls_terms_final_kernel<<<1, RB, 0, s>>>(part, nb, g_out);
    ACC_HIP(hipGetLastError());
    ACC_HIP(hipMemcpyAsync(g_pin + 8, g_out, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
    ACC_HIP(hipStreamSynchronize(s));
    out_host[0] = g_pin[8];
    return ACCBPG_OK;
}

extern "C" int accbpg_vec_min_sum(const double* x_dev, int64_t n, double* out_host, double* ws_dev, void* stream) {
    if (!x_dev || n <= 0 || !out_host || !ws_dev) return ACCBPG_ERR_ARG;
    ACC_TRY(ensure_scratch());
    hipStream_t s = (hipStream_t)stream;
    const int nb = red_blocks(n);
    double* part = ws_dev + n;
    min_sum_partial_kernel<<<nb, RB, 0, s>>>(x_dev, n, part);
    ls_terms_final_kernel<<<1, RB, 0, s>>>(part, nb, g_out);
    ACC_HIP(hipGetLastError());
    ACC_HIP(hipMemcpyAsync(g_pin + 8, g_out, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
    ACC_HIP(hipStreamSynchronize(s));
    out_host[0] = g_pin[11];
    out_host[1] = g_pin[8];
    return ACCBPG_OK;
}

extern "C" int accbpg_vec_axpby(double a, const double* x_dev, double b, const double* z_dev, int64_t n,
                                double* out_dev, void* stream) {
    if (!x_dev || !z_dev || !out_dev || n <= 0) return ACCBPG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int64_t nb = (n + RB - 1) / RB;
    if (nb > 2048) nb = 2048;
    axpby_kernel<<<(int)nb, RB, 0, s>>>(a, x_dev, b, z_dev, n, out_dev);
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

extern "C" int accbpg_vec_div_scalar(const double* x_dev, double d, int64_t n, double* out_dev, void* stream) {
    if (!x_dev || !out_dev || n <= 0) return ACCBPG_ERR_ARG;
    int64_t nb = (n + RB - 1) / RB;
    if (nb > 2048) nb = 2048;
    div_scalar_kernel<<<(int)nb, RB, 0, (hipStream_t)stream>>>(x_dev, d, n, out_dev);
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

extern "C" int accbpg_vec_vertex(int64_t idx, double value, double fill, int64_t n, double* out_dev, void* stream) {
    if (!out_dev || n <= 0 || idx < 0 || idx >= n) return ACCBPG_ERR_ARG;
    int64_t nb = (n + RB - 1) / RB;
    if (nb > 2048) nb = 2048;
    vertex_kernel<<<(int)nb, RB, 0, (hipStream_t)stream>>>(idx, value, fill, n, out_dev);
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

extern "C" int accbpg_vec_argminmax(const double* x_dev, int64_t n, int64_t* idx_host, double* val_host,
                                    double* ws_dev, void* stream) {
    if (!x_dev || n <= 0 || !idx_host || !ws_dev) return ACCBPG_ERR_ARG;
    ACC_TRY(ensure_scratch());
    hipStream_t s = (hipStream_t)stream;
    int nb = red_blocks(n);
    if (nb > 128) nb = 128;                                     // 128 records of 32 B = the 4096-double tail of ws
    MinMaxRec* part = reinterpret_cast<MinMaxRec*>(ws_dev + n + (n & 1));
    MinMaxRec* out = reinterpret_cast<MinMaxRec*>(g_out + 8);
    minmax_partial_kernel<<<nb, RB, 0, s>>>(x_dev, n, part);
    minmax_final_kernel<<<1, RB, 0, s>>>(part, nb, out);
    ACC_HIP(hipGetLastError());
    ACC_HIP(hipMemcpyAsync(g_pin + 16, out, sizeof(MinMaxRec), hipMemcpyDeviceToHost, s));
    ACC_HIP(hipStreamSynchronize(s));
    const MinMaxRec* r = reinterpret_cast<const MinMaxRec*>(g_pin + 16);
    idx_host[0] = r->imin;
    idx_host[1] = r->imax;
    if (val_host) { val_host[0] = r->vmin; val_host[1] = r->vmax; }
    return ACCBPG_OK;
}

extern "C" int accbpg_vec_dot(const double* x_dev, const double* y_dev, int64_t n, double* out_host, double* ws_dev,
                              void* stream) {
    if (!x_dev || !y_dev || n <= 0 || !out_host || !ws_dev) return ACCBPG_ERR_ARG;
    ACC_TRY(ensure_scratch());
    hipStream_t s = (hipStream_t)stream;
    const int nb = red_blocks(n);
    double* part = ws_dev + n;
    dot_partial_kernel<<<nb, RB, 0, s>>>(x_dev, y_dev, n, part);
    ls_terms_final_kernel<<<1, RB, 0, s>>>(part, nb, g_out);
    ACC_HIP(hipGetLastError());
    ACC_HIP(hipMemcpyAsync(g_pin + 8, g_out, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
    ACC_HIP(hipStreamSynchronize(s));
    out_host[0] = g_pin[8];
    return ACCBPG_OK;
}

extern "C" int accbpg_burg_reg_div_prox(int kind, const double* y_dev, const double* g_dev, double L, double lamda,
                                        int64_t n, double* x_out_dev, void* stream) {
    if (!g_dev || !x_out_dev || n <= 0 || kind < 0 || kind > 2) return ACCBPG_ERR_ARG;
    if (!(L > 0.0)) {                                           // functions.py:260, :270, :295, :320
        set_last_error("prox_map only takes positive L");
        return ACCBPG_ERR_ASSERT;
    }
    if (kind != 0 && !(lamda >= 0.0)) return ACCBPG_ERR_ARG;
    ACC_TRY(ensure_scratch());
    hipStream_t s = (hipStream_t)stream;
    ACC_HIP(hipMemsetAsync(g_flags, 0, 8 * sizeof(int), s));
    int64_t nb = (n + RB - 1) / RB;
    if (nb > 2048) nb = 2048;
    burg_reg_prox_kernel<<<(int)nb, RB, 0, s>>>(kind, y_dev, g_dev, L, lamda, n, x_out_dev, g_flags);
    ACC_HIP(hipGetLastError());
    int* pin_i = reinterpret_cast<int*>(g_pin);
    ACC_HIP(hipMemcpyAsync(pin_i, g_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    ACC_HIP(hipStreamSynchronize(s));
    if (pin_i[FLAG_NONPOS]) {                                   // y.min() > 0, functions.py:270
        set_last_error("Either y or L is not positive.");
        return ACCBPG_ERR_ASSERT;
    }
    if (pin_i[FLAG_BAD_G]) {                                    // functions.py:261 / :296
        set_last_error(kind == 1 ? "Not getting positive solution." : "BurgEntropy prox_map only takes positive value.");
        return ACCBPG_ERR_ASSERT;
    }
    return ACCBPG_OK;
}


// ------------------------------------------------------------------------------------------------------------
// Length-n kernels over the active instances of a batch (accbpg_dopt_batch_*): K x n arrays, row i = instance i.
// ------------------------------------------------------------------------------------------------------------
static int batch_vec_scratch(accbpg_dopt_batch* b) {
    if (b->vflags) return ACCBPG_OK;
    const int64_t n = b->inst[0]->n;
    ACC_HIP(hipMalloc(&b->vflags, sizeof(int) * 8 * (size_t)b->K));
    ACC_HIP(hipMemset(b->vflags, 0, sizeof(int) * 8 * (size_t)b->K));
    ACC_HIP(hipMalloc(&b->vout, sizeof(double) * 4 * (size_t)b->K));
    ACC_HIP(hipMalloc(&b->vpart, sizeof(double) * 4 * RMAXBLK * (size_t)b->K));
    if (n > (int64_t)PB * 32) ACC_HIP(hipMalloc(&b->vgg, sizeof(double) * (size_t)n * (size_t)b->K));
    ACC_HIP(hipHostMalloc(&b->vpin, sizeof(double) * 8 * (size_t)b->K, hipHostMallocDefault));
    return ACCBPG_OK;
}

static BatchAct batch_active(const accbpg_dopt_batch* b, const int* active_host) {
    BatchAct act;
    for (int i = 0; i < b->K && i < BATCH_MAX; ++i)
        if (!active_host || active_host[i]) act.idx[act.n++] = i;
    return act;
}

/* x_out[i] <- BurgEntropySimplex.div_prox_map(y[i], g[i], L_host[i]) for the active instances (y_dev NULL: prox_map).
 * status_host[i]: ACCBPG_OK or ACCBPG_ERR_ASSERT (L <= 0, or min(y) <= 0).  info_host (optional, 2 ints per instance)
 * receives {bisection steps, Newton steps}. */
extern "C" int accbpg_dopt_batch_burg_simplex_div_prox(accbpg_dopt_batch* b, const double* y_dev, const double* g_dev,
                                                       int64_t ld, const double* L_host, double eps, double* x_out_dev,
                                                       const int* active_host, int* status_host, int* info_host) {
    if (!b || !g_dev || !x_out_dev || !L_host || !status_host || b->K > BATCH_MAX) return ACCBPG_ERR_ARG;
    const int64_t n = b->inst[0]->n;
    if (ld < n) return ACCBPG_ERR_ARG;
    ACC_TRY(batch_vec_scratch(b));
    BatchAct all = batch_active(b, active_host), act;
    BatchVals Ls;
    for (int a = 0; a < all.n; ++a) {                           // functions.py:270 / :340: L > 0 is checked before anything runs
        const int i = all.idx[a];
        Ls.v[i] = L_host[i];
        if (!(L_host[i] > 0.0)) status_host[i] = ACCBPG_ERR_ASSERT;
        else { status_host[i] = ACCBPG_OK; act.idx[act.n++] = i; }
    }
    if (act.n == 0) return ACCBPG_OK;
    hipStream_t s = b->stream;
    if (n <= (int64_t)PB * 2)
        burg_prox_batch_kernel<2><<<act.n, PB, 0, s>>>(act, y_dev, g_dev, ld, Ls, eps, n, x_out_dev, nullptr, b->vflags);
    else if (n <= (int64_t)PB * 8)
        burg_prox_batch_kernel<8><<<act.n, PB, 0, s>>>(act, y_dev, g_dev, ld, Ls, eps, n, x_out_dev, nullptr, b->vflags);
    else if (n <= (int64_t)PB * 32)
        burg_prox_batch_kernel<32><<<act.n, PB, 0, s>>>(act, y_dev, g_dev, ld, Ls, eps, n, x_out_dev, nullptr, b->vflags);
    else
        burg_prox_batch_kernel<0><<<act.n, PB, 0, s>>>(act, y_dev, g_dev, ld, Ls, eps, n, x_out_dev, b->vgg, b->vflags);
    ACC_HIP(hipGetLastError());
    int* pin_i = reinterpret_cast<int*>(b->vpin);
    ACC_HIP(hipMemcpyAsync(pin_i, b->vflags, sizeof(int) * 8 * (size_t)b->K, hipMemcpyDeviceToHost, s));
    ACC_HIP(hipStreamSynchronize(s));
    for (int a = 0; a < act.n; ++a) {
        const int i = act.idx[a];
        if (pin_i[8 * i + FLAG_NONPOS]) status_host[i] = ACCBPG_ERR_ASSERT;     // y.min() > 0, functions.py:270
        if (info_host) { info_host[2 * i] = pin_i[8 * i + 4]; info_host[2 * i + 1] = pin_i[8 * i + 5]; }
    }
    return ACCBPG_OK;
}

/* out_host[3 i ..] <- { <g_i, x_i - y_i>, D_h(x_i, y_i), D_h(z_i, z1_i) } for the active instances (g, and z with z1,
 * may be NULL); status_host[i] = ACCBPG_ERR_ASSERT where an entry that enters a divergence is not positive. */
extern "C" int accbpg_dopt_batch_ls_terms(accbpg_dopt_batch* b, const double* g_dev, const double* x_dev, const double* y_dev,
                                          const double* z_dev, const double* z1_dev, int64_t ld, const int* active_host,
                                          double* out_host, int* status_host) {
    if (!b || !x_dev || !y_dev || !out_host || !status_host || b->K > BATCH_MAX) return ACCBPG_ERR_ARG;
    if ((z_dev == nullptr) != (z1_dev == nullptr)) return ACCBPG_ERR_ARG;
    const int64_t n = b->inst[0]->n;
    if (ld < n) return ACCBPG_ERR_ARG;
    ACC_TRY(batch_vec_scratch(b));
    const BatchAct act = batch_active(b, active_host);
    if (act.n == 0) return ACCBPG_OK;
    hipStream_t s = b->stream;
    const int nb = red_blocks(n);
    ls_terms_partial_batch_kernel<<<dim3(nb, act.n), RB, 0, s>>>(act, g_dev, x_dev, y_dev, z_dev, z1_dev, ld, n, b->vpart,
                                                                4 * RMAXBLK);
    ls_terms_final_batch_kernel<<<act.n, RB, 0, s>>>(act, b->vpart, 4 * RMAXBLK, nb, b->vout);
    ACC_HIP(hipGetLastError());
    ACC_HIP(hipMemcpyAsync(b->vpin, b->vout, sizeof(double) * 4 * (size_t)b->K, hipMemcpyDeviceToHost, s));
    ACC_HIP(hipStreamSynchronize(s));
    for (int a = 0; a < act.n; ++a) {
        const int i = act.idx[a];
        out_host[3 * i] = b->vpin[4 * i]; out_host[3 * i + 1] = b->vpin[4 * i + 1]; out_host[3 * i + 2] = b->vpin[4 * i + 2];
        status_host[i] = (b->vpin[4 * i + 3] > 0.0) ? ACCBPG_OK : ACCBPG_ERR_ASSERT;     // functions.py:252
    }
    return ACCBPG_OK;
}

/* out[i] <- a_host[i] * x[i] + b_host[i] * z[i] for the active instances, NumPy's rounding (algorithms.py:147,150). */
extern "C" int accbpg_dopt_batch_axpby(accbpg_dopt_batch* b, const double* a_host, const double* x_dev, const double* b_host,
                                       const double* z_dev, int64_t ld, const int* active_host, double* out_dev) {
    if (!b || !a_host || !b_host || !x_dev || !z_dev || !out_dev || b->K > BATCH_MAX) return ACCBPG_ERR_ARG;
    const int64_t n = b->inst[0]->n;
    if (ld < n) return ACCBPG_ERR_ARG;
    const BatchAct act = batch_active(b, active_host);
    if (act.n == 0) return ACCBPG_OK;
    BatchVals av, bv;
    for (int i = 0; i < b->K; ++i) { av.v[i] = a_host[i]; bv.v[i] = b_host[i]; }
    int64_t nb = (n + RB - 1) / RB;
    if (nb > 256) nb = 256;
    axpby_batch_kernel<<<dim3((unsigned)nb, act.n), RB, 0, b->stream>>>(act, av, x_dev, bv, z_dev, ld, n, out_dev);
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}
