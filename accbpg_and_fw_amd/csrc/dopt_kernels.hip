// D-optimal objective kernels for gfx950: weighted Gram matrix (stream-K over the MFMA tile
// engine), blocked Cholesky + log-determinant, blocked triangular inverse, and the triangular
// product with fused column norms that yields the gradient.
//
// Replaces the NumPy/LAPACK calls of DOptimalObj.func_grad (accbpg/functions.py:43-59):
//   :46  np.dot(H*x, H.T)            -> gram_streamk_kernel + gram_fixup_kernel
//   :48  np.linalg.slogdet           -> Cholesky (potrf_diag / trsm_panel / trailing GEMM), logdet = 2*sum log L_ii
//   :57  np.linalg.solve(HXHT, H)    -> W = L^-1 (trtri_diag + merge GEMMs), Y = W V never stored
//   :58  -sum(H * HXHTinvH, axis=0)  -> -colsum(Y*Y) fused into the product (colnorm_kernel)
#include "internal.h"
#include "mfma_tile.hpp"

namespace accbpg {

// =========================================================================================
// Weighted Gram matrix  G = (V diag(x)) V^T, lower tiles only, stream-K.
// The flattened (tile, k-step) space is cut into equal contiguous ranges, one per workgroup;
// a range that does not cover a whole tile leaves its raw accumulators in a slab, and the
// fix-up kernel adds the slabs of a tile in workgroup order (deterministic, no atomics).
// =========================================================================================
template <class T>
__global__ __launch_bounds__(NTHREADS, 1) void gram_streamk_kernel(
    const double* __restrict__ V, int64_t ldv, int64_t m, int64_t n, const double* __restrict__ x,
    const TileRC* __restrict__ tiles, int ntiles, int64_t kiters, int64_t per, double* __restrict__ slabs,
    double* __restrict__ G, int64_t ldg, bool vec_ok) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int64_t total = (int64_t)ntiles * kiters;
    const int64_t it0 = min((int64_t)blockIdx.x * per, total);
    const int64_t it1 = min(it0 + per, total);
    T t;
    int64_t it = it0;
    int seg = 0;
    while (it < it1) {
        const int tile = (int)(it / kiters);
        const int64_t kb = it - (int64_t)tile * kiters;
        const int64_t ke = min(kiters, kb + (it1 - it));
        const int64_t row0 = (int64_t)tiles[tile].rb * T::BM;
        const int64_t col0 = (int64_t)tiles[tile].cb * T::BN;
        t.zero();
        // prologue: stage k-step kb into buffer 0
        t.gload_A(V, ldv, row0, m, kb * BK, n, vec_ok);
        t.gload_B_kc(V, ldv, col0, m, kb * BK, n, vec_ok);
        t.scale_A(x, kb * BK, n);
        __syncthreads();              // previous segment's readers are done with the LDS buffers
        t.sstore(lds);
        __syncthreads();
        int cur = 0;
        for (int64_t ks = kb; ks < ke; ++ks) {
            const bool more = ks + 1 < ke;
            if (more) {
                t.gload_A(V, ldv, row0, m, (ks + 1) * BK, n, vec_ok);
                t.gload_B_kc(V, ldv, col0, m, (ks + 1) * BK, n, vec_ok);
            }
            t.compute(lds + cur * T::STAGE_ELEMS);
            if (more) {
                t.scale_A(x, (ks + 1) * BK, n);
                t.sstore(lds + (cur ^ 1) * T::STAGE_ELEMS);
            }
            __syncthreads();
            cur ^= 1;
        }
        if (kb == 0 && ke == kiters) {
            t.store_C(G, ldg, row0, col0, m, m, 1.0, 0.0, true);
        } else {
            t.store_slab(slabs + ((int64_t)blockIdx.x * 2 + (seg == 0 ? 0 : 1)) * T::SLAB_DOUBLES);
        }
        it += ke - kb;
        ++seg;
    }
}

template <class T>
__global__ __launch_bounds__(NTHREADS, 1) void gram_fixup_kernel(
    const TileRC* __restrict__ tiles, int ntiles, int64_t kiters, int64_t per, const double* __restrict__ slabs,
    double* __restrict__ G, int64_t ldg, int64_t m) {
    const int tile = blockIdx.x;
    const int64_t first_it = (int64_t)tile * kiters, last_it = first_it + kiters - 1;
    const int64_t w0 = first_it / per, w1 = last_it / per;
    // both ends of the tile inside one workgroup's range: that workgroup ran it as one whole
    // segment (kb == 0, ke == kiters) and stored it directly
    if (w0 == w1) return;
    T t;
    t.zero();
    for (int64_t w = w0; w <= w1; ++w) {
        const int64_t wfirst_tile = (w * per) / kiters;
        const int slot = (wfirst_tile == tile) ? 0 : 1;
        t.add_slab(slabs + (w * 2 + slot) * T::SLAB_DOUBLES);
    }
    t.store_C(G, ldg, (int64_t)tiles[tile].rb * T::BM, (int64_t)tiles[tile].cb * T::BN, m, m, 1.0, 0.0, true);
}

// =========================================================================================
// Gradient: out[c] = sign * sum_r Y[r][c]^2 with Y = W V, W = L^-1 lower triangular (m x m),
// V m x n.  One workgroup owns a block of BN design points (columns of V) and walks the row
// blocks of W; Y only ever exists as accumulators.  (functions.py:57-58; D_opt_alg.py:45)
// =========================================================================================
template <class T>
__global__ __launch_bounds__(NTHREADS, 1) void colnorm_kernel(
    const double* __restrict__ W, int64_t ldw, const double* __restrict__ V, int64_t ldv, int64_t m, int64_t n,
    double* __restrict__ out, double sign, bool vec_ok_w, bool vec_ok_v) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ double red[4][T::BN];
    const int64_t col0 = (int64_t)blockIdx.x * T::BN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / T::WAVES_N, wn = wave % T::WAVES_N;
    const int lr = lane & 15;
    double colsum[T::NI];
#pragma unroll
    for (int j = 0; j < T::NI; ++j) colsum[j] = 0.0;
    T t;
    const int nrb = (int)((m + T::BM - 1) / T::BM);
    for (int rb = 0; rb < nrb; ++rb) {
        const int64_t row0 = (int64_t)rb * T::BM;
        const int64_t Kend = min(row0 + T::BM, m);        // W is lower triangular
        const int64_t ksteps = (Kend + BK - 1) / BK;
        t.zero();
        t.gload_A(W, ldw, row0, m, 0, Kend, vec_ok_w);
        t.gload_B_km(V, ldv, col0, n, 0, Kend, vec_ok_v);
        __syncthreads();
        t.sstore(lds);
        __syncthreads();
        int cur = 0;
        for (int64_t ks = 0; ks < ksteps; ++ks) {
            const bool more = ks + 1 < ksteps;
            if (more) {
                t.gload_A(W, ldw, row0, m, (ks + 1) * BK, Kend, vec_ok_w);
                t.gload_B_km(V, ldv, col0, n, (ks + 1) * BK, Kend, vec_ok_v);
            }
            // rows of this wave's fragment i are 16*(i*WAVES_M+wm)..+15; they are all zero in W
            // for this k-step when their last row is above the step's first column.
            const int64_t kd = ks * BK - row0;
            int mi_lo = 0;
            if (kd > 0) {
                const int64_t num = kd - 15 - 16 * wm;
                mi_lo = num > 0 ? (int)((num + 16 * T::WAVES_M - 1) / (16 * T::WAVES_M)) : 0;
            }
            t.compute(lds + cur * T::STAGE_ELEMS, mi_lo);
            if (more) t.sstore(lds + (cur ^ 1) * T::STAGE_ELEMS);
            __syncthreads();
            cur ^= 1;
        }
        // square and add this row block's Y into the per-column sums (rows beyond m are zero)
#pragma unroll
        for (int j = 0; j < T::NI; ++j) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < T::MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) s += t.acc[i][j][r] * t.acc[i][j][r];
            colsum[j] += s;
        }
    }
    // lanes l, l+16, l+32, l+48 hold different rows of the same column
#pragma unroll
    for (int j = 0; j < T::NI; ++j) {
        double s = colsum[j];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        colsum[j] = s;
    }
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j < T::NI; ++j) red[wm][wn * T::WN + 16 * j + lr] = colsum[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < T::BN; c += NTHREADS) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < T::WAVES_M; ++w) s += red[w][c];
        if (col0 + c < n) out[col0 + c] = sign * s;
    }
}

// =========================================================================================
// Batched small GEMM on the 64x64 tile: C = alpha*A*op(B) + beta*C for a table of products.
// grid = (tiles_n, tiles_m, nops).  Used by the Cholesky trailing update and the inverse merges.
// =========================================================================================
template <class T>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_ops_kernel(const GemmOp* __restrict__ ops) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const GemmOp op = ops[blockIdx.z];
    const int64_t row0 = (int64_t)blockIdx.y * T::BM, col0 = (int64_t)blockIdx.x * T::BN;
    if (row0 >= op.M || col0 >= op.N) return;
    if (op.lower_only && col0 > row0 + T::BM - 1) return;
    const bool va = ((reinterpret_cast<uintptr_t>(op.A) & 15) == 0) && ((op.lda & 1) == 0);
    const bool vb = ((reinterpret_cast<uintptr_t>(op.B) & 15) == 0) && ((op.ldb & 1) == 0);
    T t;
    t.zero();
    const int64_t ksteps = (op.K + BK - 1) / BK;
    auto gload = [&](int64_t ks) {
        t.gload_A(op.A, op.lda, row0, op.M, ks * BK, op.K, va);
        if constexpr (T::BKM) t.gload_B_km(op.B, op.ldb, col0, op.N, ks * BK, op.K, vb);
        else t.gload_B_kc(op.B, op.ldb, col0, op.N, ks * BK, op.K, vb);
    };
    gload(0);
    t.sstore(lds);
    __syncthreads();
    int cur = 0;
    for (int64_t ks = 0; ks < ksteps; ++ks) {
        const bool more = ks + 1 < ksteps;
        if (more) gload(ks + 1);
        t.compute(lds + cur * T::STAGE_ELEMS);
        if (more) t.sstore(lds + (cur ^ 1) * T::STAGE_ELEMS);
        __syncthreads();
        cur ^= 1;
    }
    t.store_C(op.C, op.ldc, row0, col0, op.M, op.N, op.alpha, op.beta, op.lower_only != 0);
}

// same loop on the big tile, single product (unit-test hook and large merges)
template <class T>
__global__ __launch_bounds__(NTHREADS, 1) void gemm_big_kernel(GemmOp op) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int64_t row0 = (int64_t)blockIdx.y * T::BM, col0 = (int64_t)blockIdx.x * T::BN;
    if (row0 >= op.M || col0 >= op.N) return;
    const bool va = ((reinterpret_cast<uintptr_t>(op.A) & 15) == 0) && ((op.lda & 1) == 0);
    const bool vb = ((reinterpret_cast<uintptr_t>(op.B) & 15) == 0) && ((op.ldb & 1) == 0);
    T t;
    t.zero();
    const int64_t ksteps = (op.K + BK - 1) / BK;
    auto gload = [&](int64_t ks) {
        t.gload_A(op.A, op.lda, row0, op.M, ks * BK, op.K, va);
        if constexpr (T::BKM) t.gload_B_km(op.B, op.ldb, col0, op.N, ks * BK, op.K, vb);
        else t.gload_B_kc(op.B, op.ldb, col0, op.N, ks * BK, op.K, vb);
    };
    gload(0);
    t.sstore(lds);
    __syncthreads();
    int cur = 0;
    for (int64_t ks = 0; ks < ksteps; ++ks) {
        const bool more = ks + 1 < ksteps;
        if (more) gload(ks + 1);
        t.compute(lds + cur * T::STAGE_ELEMS);
        if (more) t.sstore(lds + (cur ^ 1) * T::STAGE_ELEMS);
        __syncthreads();
        cur ^= 1;
    }
    t.store_C(op.C, op.ldc, row0, col0, op.M, op.N, op.alpha, op.beta, op.lower_only != 0);
}

// =========================================================================================
// Cholesky, right-looking, block NB = 64.
// =========================================================================================
constexpr int SP = NB + 1;   // LDS stride of a 64x64 block image

// Factor the bs x bs diagonal block at A[k0][k0] in LDS.  One barrier per column: the update of
// step c touches columns > c only while everybody reads column c, and the finished column goes
// to a second image.  Adds 2*sum(log L_ii) to *logdet (stream order fixes the summation order),
// raises flags[FLAG_NOT_PD] on a non-positive (or NaN) pivot.
__global__ __launch_bounds__(NTHREADS) void potrf_diag_kernel(double* __restrict__ A, int64_t lda, int64_t k0,
                                                             int bs, double* __restrict__ logdet,
                                                             int* __restrict__ flags) {
    __shared__ double S[NB * SP];
    __shared__ double Lo[NB * SP];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    double* Ab = A + k0 * lda + k0;
    for (int e = tid; e < NB * NB; e += NTHREADS) {
        const int r = e / NB, c = e % NB;
        S[r * SP + c] = (r < bs && c <= r) ? Ab[(int64_t)r * lda + c] : ((r == c) ? 1.0 : 0.0);
        Lo[r * SP + c] = 0.0;
    }
    const int r = tid & 63, q = tid >> 6;
    bool bad = false;
    for (int c = 0; c < bs; ++c) {
        __syncthreads();
        double d = S[c * SP + c];
        if (!(d > 0.0)) { bad = true; d = 1.0; }
        const double piv = sqrt(d);
        const double rpiv = 1.0 / piv;
        if (r >= c) {
            const double lrc = (r == c) ? piv : S[r * SP + c] * rpiv;
            if (q == 0) Lo[r * SP + c] = lrc;
            if (r > c) {
                for (int cc = c + 1 + ((q - (c + 1)) & 3); cc <= r; cc += 4)
                    S[r * SP + cc] -= lrc * (S[cc * SP + c] * rpiv);
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < NB * NB; e += NTHREADS) {
        const int rr = e / NB, c = e % NB;
        if (rr < bs && c <= rr) Ab[(int64_t)rr * lda + c] = Lo[rr * SP + c];
    }
    // log-determinant contribution, fixed reduction tree
    double lg = (tid < bs) ? log(Lo[tid * SP + tid]) : 0.0;
    for (int off = 32; off > 0; off >>= 1) lg += __shfl_down(lg, off);
    if ((tid & 63) == 0) red[tid >> 6] = lg;
    __syncthreads();
    if (tid == 0) {
        *logdet += 2.0 * (red[0] + red[1] + red[2] + red[3]);
        if (bad) flags[FLAG_NOT_PD] = 1;
    }
}

// Panel solve below the diagonal block: X * L11^T = A21, one thread per row, 64 rows per
// workgroup; the row lives in registers, L11 is read from LDS by broadcast.
__global__ __launch_bounds__(64) void trsm_panel_kernel(double* __restrict__ A, int64_t lda, int64_t k0, int bs,
                                                       int64_t m) {
    __shared__ double Ls[NB * SP];
    __shared__ double Xs[NB * SP];
    __shared__ double rinv[NB];
    const int tid = threadIdx.x;
    const double* L11 = A + k0 * lda + k0;
    const int64_t prow0 = k0 + bs + (int64_t)blockIdx.x * NB;
    const int nrows = (int)min((int64_t)NB, m - prow0);
    double* P = A + prow0 * lda + k0;
    for (int e = tid; e < NB * NB; e += 64) {
        const int r = e / NB, c = e % NB;
        Ls[r * SP + c] = (r < bs && c <= r) ? L11[(int64_t)r * lda + c] : ((r == c) ? 1.0 : 0.0);
        Xs[r * SP + c] = (r < nrows && c < bs) ? P[(int64_t)r * lda + c] : 0.0;
    }
    __syncthreads();
    rinv[tid] = 1.0 / Ls[tid * SP + tid];
    __syncthreads();
    double xr[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) xr[c] = Xs[tid * SP + c];
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        double s = xr[c];
#pragma unroll
        for (int cc = 0; cc < c; ++cc) s -= xr[cc] * Ls[c * SP + cc];
        xr[c] = s * rinv[c];
    }
#pragma unroll
    for (int c = 0; c < NB; ++c) Xs[tid * SP + c] = xr[c];
    __syncthreads();
    for (int e = tid; e < NB * NB; e += 64) {
        const int r = e / NB, c = e % NB;
        if (r < nrows && c < bs) P[(int64_t)r * lda + c] = Xs[r * SP + c];
    }
}

// Inverse of every 64x64 diagonal block of L: thread j solves L11 w = e_j (column j of the
// inverse).  The block is written whole, zeros above the diagonal included.
__global__ __launch_bounds__(64) void trtri_diag_kernel(const double* __restrict__ L, int64_t ldl,
                                                       double* __restrict__ W, int64_t ldw, int64_t m) {
    __shared__ double Ls[NB * SP];
    __shared__ double Ws[NB * SP];
    __shared__ double rinv[NB];
    const int tid = threadIdx.x;
    const int64_t k0 = (int64_t)blockIdx.x * NB;
    const int bs = (int)min((int64_t)NB, m - k0);
    const double* L11 = L + k0 * ldl + k0;
    for (int e = tid; e < NB * NB; e += 64) {
        const int r = e / NB, c = e % NB;
        Ls[r * SP + c] = (r < bs && c <= r) ? L11[(int64_t)r * ldl + c] : ((r == c) ? 1.0 : 0.0);
    }
    __syncthreads();
    rinv[tid] = 1.0 / Ls[tid * SP + tid];
    __syncthreads();
    double w[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        double s = (r == tid) ? 1.0 : 0.0;
#pragma unroll
        for (int c = 0; c < r; ++c) s -= Ls[r * SP + c] * w[c];
        w[r] = (r >= tid) ? s * rinv[r] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < NB; ++r) Ws[r * SP + tid] = w[r];
    __syncthreads();
    double* W11 = W + k0 * ldw + k0;
    for (int e = tid; e < NB * NB; e += 64) {
        const int r = e / NB, c = e % NB;
        if (r < bs && c < bs) W11[(int64_t)r * ldw + c] = Ws[r * SP + c];
    }
}

__global__ void zero_scalars_kernel(double* dscal, int* dflag) {
    if (threadIdx.x < 8) dscal[threadIdx.x] = 0.0;
    if (threadIdx.x < 8) dflag[threadIdx.x] = 0;
}

__global__ void set_op_kernel(GemmOp* slot, GemmOp op) { *slot = op; }

// fp64 MFMA peak: every wave issues `iters` x 8 independent-accumulator MFMAs
__global__ __launch_bounds__(NTHREADS, 1) void mfma_peak_kernel(int iters, double* sink) {
    d4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456) sink[0] = s;
}

// =========================================================================================
// host side
// =========================================================================================
template <class K>
static int set_lds(K kernel, int bytes) {
    ACC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                bytes));
    return ACCBPG_OK;
}

void prof_begin(accbpg_dopt* h, ProfKind k) {
    if (!h->prof_on) return;
    ProfSlot& p = h->prof[k];
    if (p.used + 2 > p.ev.size()) {
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        p.ev.push_back(a);
        p.ev.push_back(b);
    }
    hipEventRecord(p.ev[p.used], h->stream);
}
void prof_end(accbpg_dopt* h, ProfKind k) {
    if (!h->prof_on) return;
    ProfSlot& p = h->prof[k];
    hipEventRecord(p.ev[p.used + 1], h->stream);
    p.used += 2;
    p.launches += 1;
}

int build_plans(accbpg_dopt* h) {
    const int64_t m = h->m;
    // ---- Gram tile list (lower tiles) and stream-K partition
    const int BM = h->big ? TileBig<false>::BM : TileSmall<false>::BM;
    const int BN = h->big ? TileBig<false>::BN : TileSmall<false>::BN;
    std::vector<TileRC> tl;
    const int nrb = (int)((m + BM - 1) / BM), ncb = (int)((m + BN - 1) / BN);
    for (int rb = 0; rb < nrb; ++rb)
        for (int cb = 0; cb < ncb; ++cb)
            if ((int64_t)cb * BN <= (int64_t)rb * BM + BM - 1) tl.push_back(TileRC{rb, cb});
    h->ntiles = (int)tl.size();
    h->kiters = (h->n + BK - 1) / BK;
    const int64_t total = (int64_t)h->ntiles * h->kiters;
    int grid = h->big ? h->num_cu : 2 * h->num_cu;
    if (grid > total) grid = (int)total;
    if (grid < h->ntiles && total / h->ntiles < 8) grid = h->ntiles;   // tiny K: one tile per workgroup
    h->gram_per = 0;
    int64_t per = (total + grid - 1) / grid;
    grid = (int)((total + per - 1) / per);
    h->gram_grid = grid;
    h->gram_per = (int)per;
    ACC_HIP(hipMalloc(&h->tiles, sizeof(TileRC) * tl.size()));
    ACC_HIP(hipMemcpy(h->tiles, tl.data(), sizeof(TileRC) * tl.size(), hipMemcpyHostToDevice));
    ACC_HIP(hipMalloc(&h->slabs, sizeof(double) * (size_t)grid * 2 * BM * BN));

    // ---- inverse merge plan: binary tree over the NB-blocks of L
    const int T = (int)((m + NB - 1) / NB);
    h->ops_host.clear();
    h->level_begin.clear();
    h->level_maxm.clear();
    h->level_maxn.clear();
    for (int span = 1; span < T; span *= 2) {
        // two op lists per level: first T1 = L21 * W11, then W21 = -W22 * T1
        std::vector<GemmOp> first, second;
        int maxm = 0, maxn = 0;
        for (int g = 0; g + span < T; g += 2 * span) {
            const int64_t r1 = (int64_t)g * NB;                         // left group rows/cols start
            const int64_t s1 = (int64_t)span * NB;                      // left size (always full)
            const int64_t r2 = r1 + s1;
            const int64_t s2 = std::min<int64_t>((int64_t)span * NB, m - r2);
            GemmOp a{};
            a.A = h->Lbuf + r2 * m + r1; a.lda = m;                     // L21 (s2 x s1)
            a.B = h->Wbuf + r1 * m + r1; a.ldb = m;                     // W11 (s1 x s1), B[k][col]
            a.C = h->Tbuf + r2 * m + r1; a.ldc = m;                     // T1  (s2 x s1)
            a.M = (int)s2; a.N = (int)s1; a.K = (int)s1; a.lower_only = 0; a.alpha = 1.0; a.beta = 0.0;
            GemmOp b{};
            b.A = h->Wbuf + r2 * m + r2; b.lda = m;                     // W22 (s2 x s2)
            b.B = h->Tbuf + r2 * m + r1; b.ldb = m;                     // T1, B[k][col]
            b.C = h->Wbuf + r2 * m + r1; b.ldc = m;                     // W21
            b.M = (int)s2; b.N = (int)s1; b.K = (int)s2; b.lower_only = 0; b.alpha = -1.0; b.beta = 0.0;
            first.push_back(a);
            second.push_back(b);
            maxm = std::max(maxm, (int)s2);
            maxn = std::max(maxn, (int)s1);
        }
        h->level_begin.push_back((int)h->ops_host.size());
        h->ops_host.insert(h->ops_host.end(), first.begin(), first.end());
        h->level_maxm.push_back(maxm); h->level_maxn.push_back(maxn);
        h->level_begin.push_back((int)h->ops_host.size());
        h->ops_host.insert(h->ops_host.end(), second.begin(), second.end());
        h->level_maxm.push_back(maxm); h->level_maxn.push_back(maxn);
    }
    h->level_begin.push_back((int)h->ops_host.size());
    if (!h->ops_host.empty()) {
        ACC_HIP(hipMalloc(&h->ops, sizeof(GemmOp) * h->ops_host.size()));
        ACC_HIP(hipMemcpy(h->ops, h->ops_host.data(), sizeof(GemmOp) * h->ops_host.size(), hipMemcpyHostToDevice));
    }
    ACC_HIP(hipMalloc(&h->chol_op, sizeof(GemmOp) * (size_t)(T + 1)));

    ACC_TRY(set_lds(gram_streamk_kernel<TileBig<false, true>>, TileBig<false>::LDS_BYTES));
    ACC_TRY(set_lds(gram_streamk_kernel<TileBig<false, false>>, TileBig<false>::LDS_BYTES));
    ACC_TRY(set_lds(gram_streamk_kernel<TileSmall<false, true>>, TileSmall<false>::LDS_BYTES));
    ACC_TRY(set_lds(gram_streamk_kernel<TileSmall<false, false>>, TileSmall<false>::LDS_BYTES));
    ACC_TRY(set_lds(colnorm_kernel<TileBig<true, true>>, TileBig<true>::LDS_BYTES));
    ACC_TRY(set_lds(colnorm_kernel<TileBig<true, false>>, TileBig<true>::LDS_BYTES));
    ACC_TRY(set_lds(colnorm_kernel<TileSmall<true, true>>, TileSmall<true>::LDS_BYTES));
    ACC_TRY(set_lds(colnorm_kernel<TileSmall<true, false>>, TileSmall<true>::LDS_BYTES));
    ACC_TRY(set_lds(gemm_ops_kernel<TileSmall<true>>, TileSmall<true>::LDS_BYTES));
    ACC_TRY(set_lds(gemm_ops_kernel<TileSmall<false>>, TileSmall<false>::LDS_BYTES));
    ACC_TRY(set_lds(gemm_big_kernel<TileBig<true>>, TileBig<true>::LDS_BYTES));
    ACC_TRY(set_lds(gemm_big_kernel<TileBig<false>>, TileBig<false>::LDS_BYTES));
    return ACCBPG_OK;
}

template <class T>
static void gram_launch_t(accbpg_dopt* h, const double* x, double* gram) {
    prof_begin(h, PROF_GRAM);
    gram_streamk_kernel<T><<<h->gram_grid, NTHREADS, T::LDS_BYTES, h->stream>>>(
        h->V, h->ldv, h->m, h->n, x, h->tiles, h->ntiles, h->kiters, h->gram_per, h->slabs, gram, h->m, h->vec_ok);
    prof_end(h, PROF_GRAM);
    prof_begin(h, PROF_GRAMFIX);
    gram_fixup_kernel<T><<<h->ntiles, NTHREADS, 0, h->stream>>>(h->tiles, h->ntiles, h->kiters, h->gram_per,
                                                                h->slabs, gram, h->m, h->m);
    prof_end(h, PROF_GRAMFIX);
}

int launch_gram(accbpg_dopt* h, const double* x, double* gram) {
    const bool xal = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (h->big) {
        const bool interior = h->vec_ok && xal && (h->m % 256 == 0) && (h->n % BK == 0);
        if (interior) gram_launch_t<TileBig<false, false>>(h, x, gram);
        else gram_launch_t<TileBig<false, true>>(h, x, gram);
    } else {
        const bool interior = h->vec_ok && xal && (h->m % 64 == 0) && (h->n % BK == 0);
        if (interior) gram_launch_t<TileSmall<false, false>>(h, x, gram);
        else gram_launch_t<TileSmall<false, true>>(h, x, gram);
    }
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

int launch_gemm_ops(const GemmOp* ops_dev, int nops, int maxM, int maxN, bool b_kmajor, hipStream_t s) {
    if (nops <= 0 || maxM <= 0 || maxN <= 0) return ACCBPG_OK;
    dim3 grid((maxN + 63) / 64, (maxM + 63) / 64, nops);
    if (b_kmajor)
        gemm_ops_kernel<TileSmall<true>><<<grid, NTHREADS, TileSmall<true>::LDS_BYTES, s>>>(ops_dev);
    else
        gemm_ops_kernel<TileSmall<false>><<<grid, NTHREADS, TileSmall<false>::LDS_BYTES, s>>>(ops_dev);
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

// In-place Cholesky of the lower triangle of A (m x m, ld m).  Resets and fills dscal[0] (log det)
// and dflag[FLAG_NOT_PD].
int launch_cholesky(accbpg_dopt* h, double* A) {
    const int64_t m = h->m;
    const int T = (int)((m + NB - 1) / NB);
    prof_begin(h, PROF_CHOL);
    zero_scalars_kernel<<<1, 64, 0, h->stream>>>(h->dscal, h->dflag);
    for (int k = 0; k < T; ++k) {
        const int64_t k0 = (int64_t)k * NB;
        const int bs = (int)std::min<int64_t>(NB, m - k0);
        potrf_diag_kernel<<<1, NTHREADS, 0, h->stream>>>(A, m, k0, bs, h->dscal, h->dflag);
        const int64_t rem = m - k0 - bs;
        if (rem <= 0) break;
        trsm_panel_kernel<<<(int)((rem + NB - 1) / NB), 64, 0, h->stream>>>(A, m, k0, bs, m);
        GemmOp op{};
        op.A = A + (k0 + bs) * m + k0; op.lda = m;      // L21 (rem x bs)
        op.B = op.A; op.ldb = m;                        // L21 again, k-contiguous -> L21 L21^T
        op.C = A + (k0 + bs) * m + (k0 + bs); op.ldc = m;
        op.M = (int)rem; op.N = (int)rem; op.K = bs; op.lower_only = 1; op.alpha = -1.0; op.beta = 1.0;
        set_op_kernel<<<1, 1, 0, h->stream>>>(h->chol_op + k, op);
        ACC_TRY(launch_gemm_ops(h->chol_op + k, 1, (int)rem, (int)rem, false, h->stream));
    }
    prof_end(h, PROF_CHOL);
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

// W = L^-1 for the factor in h->Lbuf.
int launch_trtri(accbpg_dopt* h) {
    const int64_t m = h->m;
    const int T = (int)((m + NB - 1) / NB);
    prof_begin(h, PROF_TRTRI);
    trtri_diag_kernel<<<T, 64, 0, h->stream>>>(h->Lbuf, m, h->Wbuf, m, m);
    const int nlev = (int)h->level_maxm.size();
    for (int l = 0; l < nlev; ++l) {
        const int b = h->level_begin[l], e = h->level_begin[l + 1];
        ACC_TRY(launch_gemm_ops(h->ops + b, e - b, h->level_maxm[l], h->level_maxn[l], true, h->stream));
    }
    prof_end(h, PROF_TRTRI);
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

template <class T>
static void colnorm_launch_t(accbpg_dopt* h, const double* W, double* out, double sign, bool vw) {
    const int grid = (int)((h->n + T::BN - 1) / T::BN);
    colnorm_kernel<T><<<grid, NTHREADS, T::LDS_BYTES, h->stream>>>(W, h->m, h->V, h->ldv, h->m, h->n, out, sign, vw,
                                                                  h->vec_ok);
}

int launch_colnorm(accbpg_dopt* h, const double* W, double* out, double sign) {
    prof_begin(h, PROF_GRAD);
    const bool vw = ((reinterpret_cast<uintptr_t>(W) & 15) == 0) && ((h->m & 1) == 0);
    if (h->big) {
        const bool interior = vw && h->vec_ok && (h->m % 256 == 0) && (h->n % 128 == 0);
        if (interior) colnorm_launch_t<TileBig<true, false>>(h, W, out, sign, vw);
        else colnorm_launch_t<TileBig<true, true>>(h, W, out, sign, vw);
    } else {
        const bool interior = vw && h->vec_ok && (h->m % 64 == 0) && (h->n % 64 == 0);
        if (interior) colnorm_launch_t<TileSmall<true, false>>(h, W, out, sign, vw);
        else colnorm_launch_t<TileSmall<true, true>>(h, W, out, sign, vw);
    }
    prof_end(h, PROF_GRAD);
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

int launch_test_gemm(const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, int64_t M,
                     int64_t N, int64_t K, int b_kmajor, double alpha, double beta, int config, hipStream_t s) {
    GemmOp op{};
    op.A = A; op.B = B; op.C = C; op.lda = lda; op.ldb = ldb; op.ldc = ldc;
    op.M = (int)M; op.N = (int)N; op.K = (int)K; op.lower_only = 0; op.alpha = alpha; op.beta = beta;
    if (config == 1) {
        if (b_kmajor) {
            using T = TileBig<true>;
            ACC_TRY(set_lds(gemm_big_kernel<T>, T::LDS_BYTES));
            dim3 grid((unsigned)((N + T::BN - 1) / T::BN), (unsigned)((M + T::BM - 1) / T::BM));
            gemm_big_kernel<T><<<grid, NTHREADS, T::LDS_BYTES, s>>>(op);
        } else {
            using T = TileBig<false>;
            ACC_TRY(set_lds(gemm_big_kernel<T>, T::LDS_BYTES));
            dim3 grid((unsigned)((N + T::BN - 1) / T::BN), (unsigned)((M + T::BM - 1) / T::BM));
            gemm_big_kernel<T><<<grid, NTHREADS, T::LDS_BYTES, s>>>(op);
        }
    } else {
        GemmOp* d = nullptr;
        ACC_HIP(hipMalloc(&d, sizeof(GemmOp)));
        ACC_HIP(hipMemcpyAsync(d, &op, sizeof(GemmOp), hipMemcpyHostToDevice, s));
        ACC_TRY(set_lds(gemm_ops_kernel<TileSmall<true>>, TileSmall<true>::LDS_BYTES));
        ACC_TRY(set_lds(gemm_ops_kernel<TileSmall<false>>, TileSmall<false>::LDS_BYTES));
        int rc = launch_gemm_ops(d, 1, (int)M, (int)N, b_kmajor != 0, s);
        hipStreamSynchronize(s);
        hipFree(d);
        return rc;
    }
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

int mfma_peak(int iters, double* tflops, hipStream_t s) {
    int dev = 0;
    hipDeviceProp_t prop;
    ACC_HIP(hipGetDevice(&dev));
    ACC_HIP(hipGetDeviceProperties(&prop, dev));
    const int blocks = prop.multiProcessorCount;
    double* sink = nullptr;
    ACC_HIP(hipMalloc(&sink, 8));
    hipEvent_t a, b;
    ACC_HIP(hipEventCreate(&a));
    ACC_HIP(hipEventCreate(&b));
    mfma_peak_kernel<<<blocks, NTHREADS, 0, s>>>(iters / 10 + 1, sink);   // warm-up
    ACC_HIP(hipEventRecord(a, s));
    mfma_peak_kernel<<<blocks, NTHREADS, 0, s>>>(iters, sink);
    ACC_HIP(hipEventRecord(b, s));
    ACC_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    ACC_HIP(hipEventElapsedTime(&ms, a, b));
    const double flops = (double)blocks * 4.0 * (double)iters * 8.0 * 2048.0;
    *tflops = flops / (ms * 1e-3) * 1e-12;
    hipEventDestroy(a);
    hipEventDestroy(b);
    hipFree(sink);
    return ACCBPG_OK;
}

}  // namespace accbpg
