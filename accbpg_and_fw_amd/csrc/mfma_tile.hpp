// fp64 MFMA tile engine for gfx950 (CDNA4).  One workgroup = 256 threads = 4 wave64, one wave
// per SIMD.  Every product in the D-optimal hot path (weighted Gram matrix, triangular product
// with column norms, the trailing updates of the Cholesky factorisation, the merges of the
// triangular inverse) runs through this engine on v_mfma_f64_16x16x4_f64.
//
// Operand maps of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md section 3): lane l supplies
// A[row = l&15][k = l>>4] and B[k = l>>4][col = l&15]; result register r of lane l is
// D[row = (l>>4) + 4r][col = l&15].
//
// Row fragments are interleaved over the wave rows (fragment i of wave-row wm covers tile rows
// 16*(i*WAVES_M + wm) .. +15) so that a triangular operand loads every wave equally.
//
// LDS images (BK = 16 doubles deep):
//   k-contiguous operand  : [rows][BK+2]   -- stride 18 doubles makes the 32-lane group of a
//                           ds_read_b64 (16 rows x 2 k) hit 64 distinct banks;
//   k-major (B[k][col])   : [BK][cols+16]  -- lanes 0-15 read one 128-B row segment, lanes 16-31
//                           the next k, offset by 2*(cols+16) = 32 (mod 64) banks.
// Staging is global -> registers -> LDS, double buffered, one barrier per 16-deep step: the
// fp64 MFMA takes 64 cycles per instruction per SIMD, so a 64x128 wave tile spends 8192 cycles
// of matrix work per step and a one-step-ahead register prefetch covers an HBM miss.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace accbpg {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int BK = 16;
constexpr int SK = BK + 2;          // LDS row stride (doubles) of a k-contiguous image
constexpr int NTHREADS = 256;

__device__ __forceinline__ d2 load2_guard(const double* __restrict__ p, bool row_ok, int64_t k, int64_t K,
                                          bool vec_ok) {
    d2 v;
    if (row_ok && vec_ok && k + 1 < K) {
        v = *reinterpret_cast<const d2*>(p);
    } else {
        v.x = (row_ok && k < K) ? p[0] : 0.0;
        v.y = (row_ok && k + 1 < K) ? p[1] : 0.0;
    }
    return v;
}

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E-1
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// BM x BN workgroup tile, WM x WN wave tile, B either k-major (B[k][col], "NN") or
// k-contiguous (B[col][k], "NT").
// EDGE = false promises: every tile is interior (row/col/k extents are multiples of the tile and
// of BK) and every operand row is 16-byte aligned, so staging loads are unguarded 16-byte loads.
template <int BM_, int BN_, int WM_, int WN_, bool BKM_, bool EDGE_ = true>
struct Tile {
    static constexpr bool EDGE = EDGE_;
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
    static constexpr bool BKM = BKM_;
    static constexpr int WAVES_M = BM / WM, WAVES_N = BN / WN;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    static constexpr int MI = WM / 16, NI = WN / 16;
    static constexpr int SBN = BN + 16;                         // stride of a k-major B image
    static constexpr int A_ELEMS = BM * SK;
    static constexpr int B_ELEMS = BKM ? BK * SBN : BN * SK;
    static constexpr int STAGE_ELEMS = A_ELEMS + B_ELEMS;
    static constexpr int LDS_BYTES = 2 * STAGE_ELEMS * 8;
    static constexpr int A_PASS = BM / 32;                      // 16-B chunks per thread, A
    static constexpr int BKM_TPR = BN / 2;                      // threads per k-row, k-major B
    static constexpr int BKM_RPP = NTHREADS / BKM_TPR;          // k-rows per pass
    static constexpr int B_PASS = BKM ? BK / BKM_RPP : BN / 32;
    static_assert(!BKM || (BK % BKM_RPP == 0), "k-major staging shape");

    d4 acc[MI][NI];
    d2 ra[A_PASS];
    d2 rb[B_PASS];

    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
    }

    // ---- global -> registers -------------------------------------------------------------
    // A: rows [row0, row0+BM) of a k-contiguous matrix (lda), columns [k0, k0+16).
    __device__ __forceinline__ void gload_A(const double* __restrict__ A, int64_t lda, int64_t row0, int64_t M,
                                            int64_t k0, int64_t K, bool vec_ok) {
        const int tid = threadIdx.x;
        const int c = tid & 7, r = tid >> 3;
        const int64_t k = k0 + 2 * c;
        if constexpr (!EDGE) {
            const double* __restrict__ base = A + (row0 + r) * lda + k;
#pragma unroll
            for (int p = 0; p < A_PASS; ++p) ra[p] = *reinterpret_cast<const d2*>(base + (int64_t)(32 * p) * lda);
        } else {
#pragma unroll
            for (int p = 0; p < A_PASS; ++p) {
                const int64_t row = row0 + r + 32 * p;
                ra[p] = load2_guard(A + row * lda + k, row < M, k, K, vec_ok);
            }
        }
    }
    // scale the staged A chunk by x[k], x[k+1] (weighted Gram matrix: (V*x) of functions.py:46)
    __device__ __forceinline__ void scale_A(const double* __restrict__ x, int64_t k0, int64_t K) {
        const int c = threadIdx.x & 7;
        const int64_t k = k0 + 2 * c;
        double x0, x1;
        if constexpr (!EDGE) {
            const d2 xv = *reinterpret_cast<const d2*>(x + k);
            x0 = xv.x; x1 = xv.y;
        } else {
            x0 = (k < K) ? x[k] : 0.0;
            x1 = (k + 1 < K) ? x[k + 1] : 0.0;
        }
#pragma unroll
        for (int p = 0; p < A_PASS; ++p) {
            ra[p].x *= x0;
            ra[p].y *= x1;
        }
    }
    // B k-contiguous: rows (= output columns) [col0, col0+BN) of B[col][k].
    __device__ __forceinline__ void gload_B_kc(const double* __restrict__ B, int64_t ldb, int64_t col0, int64_t N,
                                               int64_t k0, int64_t K, bool vec_ok) {
        const int tid = threadIdx.x;
        const int c = tid & 7, r = tid >> 3;
        const int64_t k = k0 + 2 * c;
        if constexpr (!EDGE) {
            const double* __restrict__ base = B + (col0 + r) * ldb + k;
#pragma unroll
            for (int p = 0; p < B_PASS; ++p) rb[p] = *reinterpret_cast<const d2*>(base + (int64_t)(32 * p) * ldb);
        } else {
#pragma unroll
            for (int p = 0; p < B_PASS; ++p) {
                const int64_t row = col0 + r + 32 * p;
                rb[p] = load2_guard(B + row * ldb + k, row < N, k, K, vec_ok);
            }
        }
    }
    // B k-major: rows [k0, k0+16) of B[k][col], columns [col0, col0+BN).
    __device__ __forceinline__ void gload_B_km(const double* __restrict__ B, int64_t ldb, int64_t col0, int64_t N,
                                               int64_t k0, int64_t K, bool vec_ok) {
        const int tid = threadIdx.x;
        const int c = 2 * (tid % BKM_TPR), r = tid / BKM_TPR;
        const int64_t col = col0 + c;
        if constexpr (!EDGE) {
            const double* __restrict__ base = B + (k0 + r) * ldb + col;
#pragma unroll
            for (int p = 0; p < B_PASS; ++p)
                rb[p] = *reinterpret_cast<const d2*>(base + (int64_t)(BKM_RPP * p) * ldb);
        } else {
#pragma unroll
            for (int p = 0; p < B_PASS; ++p) {
                const int64_t k = k0 + r + BKM_RPP * p;
                rb[p] = load2_guard(B + k * ldb + col, k < K, col, N, vec_ok);
            }
        }
    }

    // ---- registers -> LDS ----------------------------------------------------------------
    __device__ __forceinline__ void sstore(double* __restrict__ stage) {
        const int tid = threadIdx.x;
        {
            const int c = tid & 7, r = tid >> 3;
#pragma unroll
            for (int p = 0; p < A_PASS; ++p)
                *reinterpret_cast<d2*>(stage + (r + 32 * p) * SK + 2 * c) = ra[p];
        }
        double* bs = stage + A_ELEMS;
        if constexpr (BKM) {
            const int c = 2 * (tid % BKM_TPR), r = tid / BKM_TPR;
#pragma unroll
            for (int p = 0; p < B_PASS; ++p)
                *reinterpret_cast<d2*>(bs + (r + BKM_RPP * p) * SBN + c) = rb[p];
        } else {
            const int c = tid & 7, r = tid >> 3;
#pragma unroll
            for (int p = 0; p < B_PASS; ++p)
                *reinterpret_cast<d2*>(bs + (r + 32 * p) * SK + 2 * c) = rb[p];
        }
    }

    // ---- LDS -> MFMA ---------------------------------------------------------------------
    // Fragments of one 4-deep k-group live in a register set; two sets let the reads of group
    // kk+1 fly while the 32 (big tile) MFMAs of group kk occupy the matrix pipe.
    double fa[2][MI], fb[2][NI];

    template <int SET>
    __device__ __forceinline__ void read_frag(const double* __restrict__ stage, int kk) {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int wm = wave / WAVES_N, wn = wave % WAVES_N;
        const int lr = lane & 15, lq = lane >> 4;
        const double* as = stage + (16 * wm + lr) * SK + lq;   // fragment i of wave-row wm = rows 16*(i*WAVES_M+wm)..+15
        const double* bs = stage + A_ELEMS + (BKM ? (lq * SBN + wn * WN + lr) : ((wn * WN + lr) * SK + lq));
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[SET][i] = as[i * 16 * WAVES_M * SK + 4 * kk];
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[SET][j] = BKM ? bs[4 * kk * SBN + 16 * j] : bs[j * 16 * SK + 4 * kk];
    }
    // mi_lo: first 16-row fragment of this wave that can be non-zero (triangular skip), [mi_lo, MI).
    template <int SET>
    __device__ __forceinline__ void mma_frag(int mi_lo = 0) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (i >= mi_lo) {
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[SET][i], fb[SET][j], acc[i][j], 0, 0, 0);
            }
        }
    }
    // one fragment row of a group (NI MFMAs): lets the caller place other instructions between rows
    template <int SET>
    __device__ __forceinline__ void mma_row(int i_rt, int mi_lo = 0) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
            if (i == i_rt && i >= mi_lo) {
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[SET][i], fb[SET][j], acc[i][j], 0, 0, 0);
            }
    }
    // x[k] of the fragment set (weighted Gram matrix), applied by scale_frag right before the set is used:
    // the multiply then waits for data that was requested a whole MFMA group earlier
    double fx[2];
    template <int SET>
    __device__ __forceinline__ void scale_frag() {
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[SET][i] *= fx[SET];
    }
    // whole 16-deep stage, no overlap with staging (small / latency-bound users)
    __device__ __forceinline__ void compute(const double* __restrict__ stage, int mi_lo = 0) {
        read_frag<0>(stage, 0);
        read_frag<1>(stage, 1);
        mma_frag<0>(mi_lo);
        read_frag<0>(stage, 2);
        mma_frag<1>(mi_lo);
        read_frag<1>(stage, 3);
        mma_frag<0>(mi_lo);
        mma_frag<1>(mi_lo);
    }

    // ---- direct-to-LDS staging (interior tiles only) --------------------------------------
    // global_load_lds_dwordx4 writes one wave-instruction's 64 x 16 bytes LINEARLY into LDS, so the
    // images are unpadded and the bank-conflict fix is an XOR swizzle applied to the per-lane SOURCE
    // address and, identically, to the fragment read address:
    //   k-contiguous image [rows][16]: 16-byte chunk c of row r sits at chunk c ^ ((r >> 1) & 7)
    //     (the 32-lane group of a ds_read_b64 -- 16 rows x 2 k -- then covers all 64 banks);
    //   k-major image [16][BN]       : column col of row k sits at col ^ (16 * (k & 1)).
    // Three stages of (A image, B image, 32 doubles of x) rotate; loads run two k-steps ahead and
    // are retired with a counted s_waitcnt vmcnt, never drained inside the loop.
    static constexpr int G_A = BM * BK;
    static constexpr int G_B = BN * BK;
    static constexpr int G_X = 32;
    static constexpr int G_STAGE = G_A + G_B + G_X;
    static constexpr int G_LDS_BYTES = 3 * G_STAGE * 8;
    static constexpr int G_NA = BM / 32;                 // glds per wave, A image
    static constexpr int G_NB = BKM ? BK / 4 : BN / 32;  // glds per wave, B image

    static __device__ __forceinline__ void glds16(const double* __restrict__ src, double* __restrict__ dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    // NPK pieces that are neighbours in the LDS image behind ONE write of M0: the instruction's immediate offset moves
    // the LDS destination AND the global source, so piece j of such a run has its lane offset set up j KiB low
    // (glds_setup_*<NPK>).  Takes the scalar work per load (M0 write + its wait state) and the copy of the lane offset
    // that the builtin form needs out of the k-step.  `l0` = LDS byte address of the wave's first piece, OFF = bytes from
    // there.  The "m0" clobber is REQUIRED: without it the compiler moves the M0 write of one of its own LDS-DMA loads (the
    // x piece) above such a statement and its load then lands where this statement pointed M0.  (clang warns that a
    // clobber of a reserved register is not saved and restored -- nothing needs it to be: the compiler writes M0 before
    // every use; with the clobber it also orders those writes behind the statement.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    template <int OFF>
    static __device__ __forceinline__ void glds16_run2(const char* __restrict__ base, uint32_t o0, uint32_t o1, uint32_t l0) {
        asm volatile("s_add_u32 m0, %2, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024"
                     :: "v"(o0), "v"(o1), "s"(l0), "s"(base), "i"(OFF) : "memory", "scc", "m0");
    }
    template <int OFF>
    static __device__ __forceinline__ void glds16_run4(const char* __restrict__ base, uint32_t o0, uint32_t o1, uint32_t o2,
                                                       uint32_t o3, uint32_t l0) {
        asm volatile("s_add_u32 m0, %4, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %5\n\tglobal_load_lds_dwordx4 %1, %5 offset:1024\n\t"
                     "global_load_lds_dwordx4 %2, %5 offset:2048\n\tglobal_load_lds_dwordx4 %3, %5 offset:3072"
                     :: "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(l0), "s"(base), "i"(OFF) : "memory", "scc", "m0");
    }
#pragma clang diagnostic pop
    static __device__ __forceinline__ uint32_t lds_addr(const double* p) {
        return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
    }
    static constexpr int G_PAIR_LOW = 1024;
    // 1-KiB piece of the image that piece p of wave `wave` fills: dealt round the waves, or (runs) np consecutive ones
    template <int NPK>
    static __device__ __forceinline__ int glds_piece(int wave, int p, int np) { return NPK > 1 ? wave * np + p : wave + 4 * p; }
    // per-lane BYTE offsets of the pieces this wave moves, relative to the uniform base pointers of k-step 0
    // (set once per tile segment).  A k-step adds a scalar to the base only, so the load takes the
    // "scalar base + 32-bit lane offset" addressing form and needs no vector address arithmetic.
    uint32_t goa[G_NA];
    uint32_t gob[G_NB];
    const char* gbase_a;
    const char* gbase_b;
    template <int NPK = 1>
    __device__ __forceinline__ void glds_setup_A(const double* __restrict__ A, int64_t lda, int64_t row0) {
        static_assert(G_NA % NPK == 0, "pieces go out in whole runs");
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        gbase_a = reinterpret_cast<const char*>(A + row0 * lda);
#pragma unroll
        for (int p = 0; p < G_NA; ++p) {
            const int q = glds_piece<NPK>(wave, p, G_NA);        // 1-KiB piece = rows 8q .. 8q+7
            const int row = 8 * q + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            // (piece j of a run has q >= j, so its offset is at least 8j rows = 64j * lda bytes >= j KiB: lda >= BK)
            goa[p] = (uint32_t)(((int64_t)row * lda + 2 * c) * 8) - (p % NPK) * G_PAIR_LOW;
        }
    }
    template <int NPK = 1>
    __device__ __forceinline__ void glds_setup_B_kc(const double* __restrict__ B, int64_t ldb, int64_t col0) {
        static_assert(G_NB % NPK == 0, "pieces go out in whole runs");
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        gbase_b = reinterpret_cast<const char*>(B + col0 * ldb);
#pragma unroll
        for (int p = 0; p < G_NB; ++p) {
            const int q = glds_piece<NPK>(wave, p, G_NB);
            const int row = 8 * q + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            gob[p] = (uint32_t)(((int64_t)row * ldb + 2 * c) * 8) - (p % NPK) * G_PAIR_LOW;
        }
    }
    template <int NPK = 1>
    __device__ __forceinline__ void glds_setup_B_km(const double* __restrict__ B, int64_t ldb, int64_t col0) {
        static_assert(!BKM || BN == 128, "one 1-KiB piece per k-row");
        static_assert(G_NB % NPK == 0, "pieces go out in whole runs");
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        gbase_b = reinterpret_cast<const char*>(B + col0);
#pragma unroll
        for (int p = 0; p < G_NB; ++p) {
            const int kr = glds_piece<NPK>(wave, p, G_NB);       // k-row of the tile
            const int c = lane ^ (8 * (kr & 1));                 // source chunk (2 columns) for LDS chunk `lane`
            // (piece j of a run has kr >= j, so its offset is at least j rows = 8j * ldb bytes >= j KiB: ldb >= BN)
            gob[p] = (uint32_t)(((int64_t)kr * ldb + 2 * c) * 8) - (p % NPK) * G_PAIR_LOW;
        }
    }
    // issue the pieces of one k-step: ka / kb = element offsets of that step in A / B
    template <int NPK = 1>
    __device__ __forceinline__ void glds_issue(int64_t ka, int64_t kb, double* __restrict__ stage) const {
        if constexpr (NPK > 1) {
            glds_issue_range<NPK>(ka, kb, stage, 0, G_NA + G_NB);
            return;
        }
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const char* ba = gbase_a + ka * 8;
        const char* bb = gbase_b + kb * 8;
        // (the empty asm keeps the 32-bit offset's zero-extension next to its use: hoisted out of the k-loop
        //  it would turn into a 64-bit vector add per load instead of the scalar-base form)
#pragma unroll
        for (int p = 0; p < G_NA; ++p) {
            uint32_t o = goa[p];
            asm volatile("" : "+v"(o));
            glds16(reinterpret_cast<const double*>(ba + o), stage + (wave + 4 * p) * 128);
        }
#pragma unroll
        for (int p = 0; p < G_NB; ++p) {
            uint32_t o = gob[p];
            asm volatile("" : "+v"(o));
            glds16(reinterpret_cast<const double*>(bb + o), stage + G_A + (wave + 4 * p) * 128);
        }
    }
    // pieces [p0, p1) of the combined list (A pieces first, then B pieces) of one k-step
    template <int NPK = 1>
    __device__ __forceinline__ void glds_issue_range(int64_t ka, int64_t kb, double* __restrict__ stage, int p0,
                                                     int p1) const {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const char* ba = gbase_a + ka * 8;
        const char* bb = gbase_b + kb * 8;
        if constexpr (NPK > 1) {
            // (p0, p1 multiples of NPK: a range is whole runs)
            const uint32_t la = lds_addr(stage + wave * G_NA * 128), lb = lds_addr(stage + G_A + wave * G_NB * 128);
            static_for<0, (G_NA + G_NB) / NPK>([&](auto rtag) {
                constexpr int p = decltype(rtag)::value * NPK;
                if (p >= p0 && p + NPK <= p1) {
                    if constexpr (p < G_NA) {
                        if constexpr (NPK == 2) glds16_run2<p * 1024>(ba, goa[p], goa[p + 1], la);
                        else glds16_run4<p * 1024>(ba, goa[p], goa[p + 1], goa[p + 2], goa[p + 3], la);
                    } else {
                        constexpr int pb = p - G_NA;
                        if constexpr (NPK == 2) glds16_run2<pb * 1024>(bb, gob[pb], gob[pb + 1], lb);
                        else glds16_run4<pb * 1024>(bb, gob[pb], gob[pb + 1], gob[pb + 2], gob[pb + 3], lb);
                    }
                }
            });
            return;
        }
#pragma unroll
        for (int p = 0; p < G_NA + G_NB; ++p)
            if (p >= p0 && p < p1) {
                uint32_t o = (p < G_NA) ? goa[p < G_NA ? p : 0] : gob[p >= G_NA ? p - G_NA : 0];
                asm volatile("" : "+v"(o));
                const char* b = (p < G_NA) ? ba : bb;
                glds16(reinterpret_cast<const double*>(b + o), stage + (p < G_NA ? 0 : G_A) + (wave + 4 * (p < G_NA ? p : p - G_NA)) * 128);
            }
    }
    // 32 doubles of x (16 used) behind the images; every wave writes the same bytes so that all
    // waves carry the same number of outstanding loads
    __device__ __forceinline__ void glds_x(const double* __restrict__ x, int64_t k0, double* __restrict__ stage) const {
        const int lane = threadIdx.x & 63;
        const float* src = reinterpret_cast<const float*>(x + k0) + (lane & 31);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stage + G_A + G_B), 4, 0, 0);
    }
    // fragment group kk of a swizzled stage -> register set SET; SCALE multiplies the A fragments by
    // x[k] (the weighted Gram matrix: (V*x) of functions.py:46)
    template <int SET, bool SCALE>
    __device__ __forceinline__ void read_frag_g(const double* __restrict__ stage, int kk) {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int wm = wave / WAVES_N, wn = wave % WAVES_N;
        const int lr = lane & 15, lq = lane >> 4;
        const int sw = (lr >> 1) & 7;
        const int koff = 2 * ((2 * kk + (lq >> 1)) ^ sw) + (lq & 1);      // swizzled position of k = 4kk+lq
        const double* as = stage + (16 * wm + lr) * BK + koff;
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[SET][i] = as[i * 16 * WAVES_M * BK];
        if constexpr (BKM) {
            const double* bs = stage + G_A + (4 * kk + lq) * BN + wn * WN + lr;
#pragma unroll
            for (int j = 0; j < NI; ++j) fb[SET][j] = bs[16 * (j ^ (lq & 1))];
        } else {
            const double* bs = stage + G_A + (wn * WN + lr) * BK + koff;
#pragma unroll
            for (int j = 0; j < NI; ++j) fb[SET][j] = bs[j * 16 * BK];
        }
        if constexpr (SCALE) fx[SET] = stage[G_A + G_B + 4 * kk + lq];   // applied by scale_frag<SET>()
    }

    // The same fragment group in seven parts, so that the caller can deal the LDS reads out between MFMAs instead of
    // issuing them as one block at a group boundary (where the matrix pipe would sit idle behind its last MFMA while
    // the block issues).  Parts in the order the next group needs them: 0 = x (SCALE), 1 = A rows 0,1, 2..5 = B
    // column fragments 0,1 / 2,3 / 4,5 / 6,7, 6 = A rows 2,3.
    template <int SET, bool SCALE, int PART>
    __device__ __forceinline__ void read_part_g(const double* __restrict__ stage, int kk) {
        static_assert(MI == 4 && NI == 8, "parts are laid out for the 64 x 128 wave tile");
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int wm = wave / WAVES_N, wn = wave % WAVES_N;
        const int lr = lane & 15, lq = lane >> 4;
        const int sw = (lr >> 1) & 7;
        const int koff = 2 * ((2 * kk + (lq >> 1)) ^ sw) + (lq & 1);
        if constexpr (PART == 0) {
            if constexpr (SCALE) fx[SET] = stage[G_A + G_B + 4 * kk + lq];
        } else if constexpr (PART == 1 || PART == 6) {
            const double* as = stage + (16 * wm + lr) * BK + koff;
            constexpr int i0 = (PART == 1) ? 0 : 2;
            fa[SET][i0] = as[i0 * 16 * WAVES_M * BK];
            fa[SET][i0 + 1] = as[(i0 + 1) * 16 * WAVES_M * BK];
        } else {
            constexpr int j0 = 2 * (PART - 2);
            if constexpr (BKM) {
                const double* bs = stage + G_A + (4 * kk + lq) * BN + wn * WN + lr;
                fb[SET][j0] = bs[16 * (j0 ^ (lq & 1))];
                fb[SET][j0 + 1] = bs[16 * ((j0 + 1) ^ (lq & 1))];
            } else {
                const double* bs = stage + G_A + (wn * WN + lr) * BK + koff;
                fb[SET][j0] = bs[j0 * 16 * BK];
                fb[SET][j0 + 1] = bs[(j0 + 1) * 16 * BK];
            }
        }
    }
    template <int SET, int I>
    __device__ __forceinline__ void scale_row() { fa[SET][I] *= fx[SET]; }
    // MFMAs (I, 2P) and (I, 2P+1) of a fragment group
    template <int SET, int I, int P>
    __device__ __forceinline__ void mma_pair(int mi_lo = 0) {
        if (I >= mi_lo) {
            acc[I][2 * P] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[SET][I], fb[SET][2 * P], acc[I][2 * P], 0, 0, 0);
            acc[I][2 * P + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[SET][I], fb[SET][2 * P + 1], acc[I][2 * P + 1], 0, 0, 0);
        }
    }

    // ---- "dual" diagonal tile (Gram matrix only) ---------------------------------------------
    // A 128 x 128 block ON the diagonal, G[band, band] = sum_k (V x)[band, k] V[band, k]^T, uses the same
    // rows of V for both operands.  The 256-row A image then holds the band twice, at k-step ks (rows
    // 0..127) and at k-step ks + K/2 (rows 128..255); the B fragments are read from the A image itself
    // (rows 16j+lr of the matching half), no B image is loaded, and the two halves of the accumulators
    // are partial sums over the two halves of K of the SAME 128 x 128 block -- all 32 MFMAs per fragment
    // group do useful work for K/2 steps instead of half of them for K steps.
    double fb2[2][NI];
    __device__ __forceinline__ void glds_setup_A_dual(const double* __restrict__ V, int64_t ldv, int64_t band0,
                                                      int64_t khalf) {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        gbase_a = reinterpret_cast<const char*>(V + band0 * ldv);
#pragma unroll
        for (int p = 0; p < G_NA; ++p) {
            const int q = wave + 4 * p;
            const int row = 8 * q + (lane >> 3);                 // image row
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            goa[p] = (uint32_t)(((int64_t)(row & (BM / 2 - 1)) * ldv + 2 * c + ((row >= BM / 2) ? khalf : 0)) * 8);
        }
    }
    __device__ __forceinline__ void glds_issue_dual(int64_t ka, const double* __restrict__ x, int64_t khalf,
                                                    double* __restrict__ stage) const {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const char* ba = gbase_a + ka * 8;
#pragma unroll
        for (int p = 0; p < G_NA; ++p) {
            uint32_t o = goa[p];
            asm volatile("" : "+v"(o));
            glds16(reinterpret_cast<const double*>(ba + o), stage + (wave + 4 * p) * 128);
        }
        // one 256-byte piece: lanes 0..31 bring x[ka .. ka+15], lanes 32..63 x[ka+khalf .. +15]
        const float* sx = reinterpret_cast<const float*>(x + ka + ((lane & 32) ? khalf : 0)) + (lane & 31);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sx,
                                         (__attribute__((address_space(3))) void*)(stage + G_A + G_B), 4, 0, 0);
    }
    static constexpr int G_NLD_DUAL = G_NA + 1;
    // Which 16-row block of the 128-row band fragment row i2 (= i mod MI/2) of wave `w` covers in a dual tile.  Row
    // block b of the lower triangle holds b + 1 fragments, so a wave takes the blocks w and 7 - w: 9 fragments for
    // every wave (with the blocks w and 4 + w of the ordinary layout wave 3 would carry 12 and wave 0 only 6, and the
    // slowest wave sets the time of the step).  Which wave forms a fragment does not enter its arithmetic.
    static __device__ __forceinline__ int dual_row_block(int i2, int w) {
        return i2 ? (2 * WAVES_M - 1 - w) : w;                  // (dual tiles exist for the 256 x 128 tile only: MI / 2 = 2, WAVES_M = 4)
    }
    template <int SET>
    __device__ __forceinline__ void read_frag_dual(const double* __restrict__ stage, int kk) {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int lr = lane & 15, lq = lane >> 4;
        const int sw = (lr >> 1) & 7;
        const int koff = 2 * ((2 * kk + (lq >> 1)) ^ sw) + (lq & 1);
#pragma unroll
        for (int i = 0; i < MI; ++i)
            fa[SET][i] = stage[((i / (MI / 2)) * (BM / 2) + 16 * dual_row_block(i % (MI / 2), wave) + lr) * BK + koff];
        const double* bs = stage + lr * BK + koff;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            fb[SET][j] = bs[j * 16 * BK];
            fb2[SET][j] = bs[(BM / 2 + j * 16) * BK];
        }
        fx[SET] = stage[G_A + G_B + 4 * kk + lq];
        fx2[SET] = stage[G_A + G_B + 16 + 4 * kk + lq];
    }
    double fx2[2];
    template <int SET>
    __device__ __forceinline__ void scale_dual() {
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[SET][i] *= (i < MI / 2) ? fx[SET] : fx2[SET];
    }
    // Fragment (i, j) of the diagonal block covers the 16-row block dual_row_block(i mod MI/2, wave) and columns
    // 16j .. 16j+15: it lies strictly above the diagonal, and is skipped, when j > that row block.
    template <int SET>
    __device__ __forceinline__ void mma_dual() {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int lim = dual_row_block(i % (MI / 2), wave);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if (j > lim) break;
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[SET][i], (i < MI / 2) ? fb[SET][j] : fb2[SET][j],
                                                                 acc[i][j], 0, 0, 0);
            }
        }
    }
    // ---- epilogues -----------------------------------------------------------------------
    // C[row][col] = alpha*acc + beta*C, guarded; lower_only drops elements with col > row.
    __device__ __forceinline__ void store_C(double* __restrict__ C, int64_t ldc, int64_t row0, int64_t col0,
                                            int64_t M, int64_t N, double alpha, double beta, bool lower_only) const {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int wm = wave / WAVES_N, wn = wave % WAVES_N;
        const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = row0 + 16 * (i * WAVES_M + wm) + lq + 4 * r;
                    const int64_t col = col0 + wn * WN + 16 * j + lr;
                    if ((!EDGE || (row < M && col < N)) && !(lower_only && col > row)) {
                        double* p = C + row * ldc + col;
                        double v = alpha * acc[i][j][r];
                        if (beta != 0.0) v += beta * *p;
                        *p = v;
                    }
                }
    }
    // visit every accumulator element: f(row_in_tile, col_in_tile, value)
    template <class F>
    __device__ __forceinline__ void for_each(F&& f) {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int wm = wave / WAVES_N, wn = wave % WAVES_N;
        const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    f(16 * (i * WAVES_M + wm) + lq + 4 * r, wn * WN + 16 * j + lr, (double)acc[i][j][r]);
    }
    // raw accumulator slab (stream-K partials): [MI*NI*4][256] doubles, coalesced per register.
    __device__ __forceinline__ void store_slab(double* __restrict__ slab) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[((i * NI + j) * 4 + r) * NTHREADS + tid] = acc[i][j][r];
    }
    __device__ __forceinline__ void add_slab(const double* __restrict__ slab) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += slab[((i * NI + j) * 4 + r) * NTHREADS + tid];
    }
    static constexpr int SLAB_DOUBLES = MI * NI * 4 * NTHREADS;   // == BM*BN
    // fix-up on a piece of a tile: fragment row `part` (of MI), column fragments [j0, j1)
    __device__ __forceinline__ void add_slab_part(const double* __restrict__ slab, int part, int j0 = 0,
                                                  int j1 = NI) {
        add_slab_part_to(slab, part, part, j0, j1);
    }
    // dual tiles: fragment row `src` of a slab is a partial sum of output fragment row `dst`
    __device__ __forceinline__ void add_slab_part_to(const double* __restrict__ slab, int src, int dst, int j0 = 0,
                                                     int j1 = NI) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < MI; ++i)
            if (i == dst) {
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    if (j >= j0 && j < j1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][r] += slab[((src * NI + j) * 4 + r) * NTHREADS + tid];
                    }
            }
    }
    __device__ __forceinline__ void store_C_part(double* __restrict__ C, int64_t ldc, int64_t row0, int64_t col0,
                                                 int64_t M, int64_t N, bool lower_only, int part, int j0 = 0,
                                                 int j1 = NI) const {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int wm = wave / WAVES_N, wn = wave % WAVES_N;
        const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
        for (int i = 0; i < MI; ++i)
            if (i == part) {
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    if (j >= j0 && j < j1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int64_t row = row0 + 16 * (i * WAVES_M + wm) + lq + 4 * r;
                            const int64_t col = col0 + wn * WN + 16 * j + lr;
                            if (row < M && col < N && !(lower_only && col > row)) C[row * ldc + col] = acc[i][j][r];
                        }
                    }
            }
    }
};

// Scheduling pipelines (LLVM sched_group_barrier) for the software-pipelined main loops of the big
// tile.  Masks: VALU 0x2, MFMA 0x8, VMEM_READ 0x20, DS_READ 0x100, DS_WRITE 0x200.
// One k-step = four groups of NMFMA MFMAs:
//   group 0: + LDS reads of fragment group 1
//   group 1: + LDS reads of group 2
//   group 2: + LDS reads of group 3, the scaling multiplies and the LDS writes of the next tile
//   -- barrier --
//   group 3: + global loads of the tile after next, LDS reads of group 0 of the next tile
// (measured alternatives: staging in group 1 with the loads behind it was 6 % slower)
template <int NMFMA, int NDSR, int NVALU, int NDSW, int NVMEM>
__device__ __forceinline__ void sched_pre_barrier() {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (g == 0 && NVMEM > 0) __builtin_amdgcn_sched_group_barrier(0x020, NVMEM, 0);
#pragma unroll
        for (int i = 0; i < NDSR; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - NDSR, 0);
    }
#pragma unroll
    for (int i = 0; i < NDSR; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    constexpr int REST = NMFMA - NDSR;                 // MFMAs left in group 2
    constexpr int NV = NVALU > 0 ? (NVALU + 1) / 2 : 0; // two multiplies per MFMA
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
    }
#pragma unroll
    for (int i = 0; i < NDSW; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
    if (REST - NV - NDSW > 0) __builtin_amdgcn_sched_group_barrier(0x008, REST - NV - NDSW, 0);
}
template <int NMFMA, int NVMEM, int NDSR>
__device__ __forceinline__ void sched_post_barrier() {
#pragma unroll
    for (int i = 0; i < NVMEM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < NDSR; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - NVMEM - NDSR, 0);
}

// The two instantiated shapes: "big" fills one CU with a 256x128 tile (wave tile 64x128, 256
// accumulator VGPRs, one workgroup per CU); "small" is a 64x64 tile for the latency-bound
// pieces (Cholesky trailing updates, inverse merges, small instances).
template <bool BKM, bool EDGE = true> using TileBig = Tile<256, 128, 64, 128, BKM, EDGE>;
template <bool BKM, bool EDGE = true> using TileSmall = Tile<64, 64, 32, 32, BKM, EDGE>;

}  // namespace accbpg
