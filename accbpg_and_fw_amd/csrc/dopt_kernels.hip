// D-optimal objective kernels for gfx950: weighted Gram matrix (stream-K over the MFMA tile
// engine), blocked Cholesky + log-determinant, blocked triangular inverse, and the triangular
// product with fused column norms that yields the gradient.
//
// Replaces the NumPy/LAPACK calls of DOptimalObj.func_grad (accbpg/functions.py:43-59):
//   :46  np.dot(H*x, H.T)            -> gram_streamk_kernel + gram_fixup_kernel
//   :48  np.linalg.slogdet           -> Cholesky (potrf_diag / trsm_panel / trailing GEMM), logdet = 2*sum log L_ii
//   :57  np.linalg.solve(HXHT, H)    -> W = L^-1 (trtri_diag + merge GEMMs), Y = W V never stored
//   :58  -sum(H * HXHTinvH, axis=0)  -> -colsum(Y*Y) fused into the product (colnorm_kernel)
#include <mutex>
#include <unordered_map>
#include <type_traits>

#include "internal.h"
#include "mfma_tile.hpp"

namespace accbpg {

// One stream-K segment: the part of a tile a workgroup covers with the units [it, it1) it still owns.
// The unit space has `kiters` units per ENTRY of the tile list and one k-step per unit.  An entry is a
// BM x BN tile, or a pair of 128 x 128 diagonal blocks run as dual tiles (mfma_tile.hpp): a dual tile
// covers K in kiters/2 steps of full MFMA work, so two of them fill one entry and equal unit ranges
// stay equal work.
struct GramSeg {
    int64_t row0, col0, kb, ke;
    bool dual, whole;
};
constexpr int GRAM_RMAX = 4;    // unit ranges a workgroup can own (wg_ranges[w][r] = {first, end})
template <class T>
__device__ __forceinline__ GramSeg gram_segment(const TileRC* __restrict__ tiles, int64_t kiters, int64_t it,
                                                int64_t it1) {
    const int e = (int)(it / kiters);
    const int64_t off = it - (int64_t)e * kiters;
    const TileRC tr = tiles[e];
    GramSeg g;
    if (tr.d1 < 0) {
        g.row0 = (int64_t)tr.rb * T::BM; g.col0 = (int64_t)tr.cb * T::BN;
        g.kb = off; g.ke = min(kiters, off + (it1 - it));
        g.dual = false;
        g.whole = (g.kb == 0 && g.ke == kiters);
    } else {
        const int64_t half = kiters >> 1;
        const bool second = off >= half;
        g.kb = second ? off - half : off;
        g.ke = min(half, g.kb + (it1 - it));
        g.row0 = g.col0 = (int64_t)(second ? tr.d2 : tr.d1) * (T::BM / 2);
        g.dual = true;
        g.whole = (g.kb == 0 && g.ke == half);
    }
    return g;
}

// =========================================================================================
// Weighted Gram matrix  G = (V diag(x)) V^T, lower tiles only, stream-K.
// The flattened (tile, k-step) space is cut into equal contiguous ranges, one per workgroup;
// a range that does not cover a whole tile leaves its raw accumulators in a slab, and the
// fix-up kernel adds the slabs of a tile in workgroup order (deterministic, no atomics).
// =========================================================================================
// VAR (timing ablations only, wrong results unless 0): 1 = no global loads in the steady state,
// 2 = no scaling / LDS staging writes, 3 = neither and no barrier, 4 = additionally no fragment reads.
template <class T, int VAR = 0>
__global__ __launch_bounds__(NTHREADS, 1) void gram_streamk_kernel(
    const double* __restrict__ V, int64_t ldv, int64_t m, int64_t n, const double* __restrict__ x,
    const TileRC* __restrict__ tiles, const int64_t* __restrict__ wg_ranges, int64_t kiters, int nslot,
    double* __restrict__ slabs,
    double* __restrict__ G, int64_t ldg, bool vec_ok) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    T t;
    int seg = 0;
    for (int rg = 0; rg < GRAM_RMAX; ++rg) {
    int64_t it = wg_ranges[((int64_t)blockIdx.x * GRAM_RMAX + rg) * 2];
    const int64_t it1 = wg_ranges[((int64_t)blockIdx.x * GRAM_RMAX + rg) * 2 + 1];
    while (it < it1) {
        const GramSeg sg = gram_segment<T>(tiles, kiters, it, it1);   // never dual: pairs exist on the glds path only
        const int64_t kb = sg.kb, ke = sg.ke, row0 = sg.row0, col0 = sg.col0;
        t.zero();
        // prologue: stage k-step kb into buffer 0, start the loads of k-step kb+1
        t.gload_A(V, ldv, row0, m, kb * BK, n, vec_ok);
        t.gload_B_kc(V, ldv, col0, m, kb * BK, n, vec_ok);
        t.scale_A(x, kb * BK, n);
        __syncthreads();              // previous segment's readers are done with the LDS buffers
        t.sstore(lds);
        __syncthreads();
        // (the tail steps reload / restage the last tile redundantly: no branch in the steady state,
        //  so the staging instructions can be scheduled between the MFMAs)
        const int64_t klast = ke - 1;
        {
            const int64_t k1 = min(kb + 1, klast);
            t.gload_A(V, ldv, row0, m, k1 * BK, n, vec_ok);
            t.gload_B_kc(V, ldv, col0, m, k1 * BK, n, vec_ok);
        }
        t.template read_frag<0>(lds, 0);
        int cur = 0;
        // Steady state of k-step ks (its tile is complete in buffer `cur`, the loads of tile ks+1 are
        // in flight since the previous step, fragment group 0 is in register set 0):
        //   groups 0..2 run with the next group's LDS reads in flight; tile ks+1 is scaled and
        //   written to the other buffer in the shadow of group 2; ONE barrier; group 3 runs while
        //   the loads of tile ks+2 are issued and group 0 of tile ks+1 is read.
#pragma unroll 1
        for (int64_t ks = kb; ks < ke; ++ks) {
            const int64_t k1 = min(ks + 1, klast), k2 = min(ks + 2, klast);
            const double* st = lds + cur * T::STAGE_ELEMS;
            double* nx = lds + (cur ^ 1) * T::STAGE_ELEMS;
            if constexpr (VAR < 4) t.template read_frag<1>(st, 1);
            t.template mma_frag<0>();
            if constexpr (VAR < 4) t.template read_frag<0>(st, 2);
            t.template mma_frag<1>();
            if constexpr (VAR < 4) t.template read_frag<1>(st, 3);
            if constexpr (VAR == 0 || VAR == 1) {
                t.scale_A(x, k1 * BK, n);
                t.sstore(nx);
            }
            t.template mma_frag<0>();
            if constexpr (T::MI * T::NI == 32 && !T::EDGE && VAR == 0)
                sched_pre_barrier<32, T::MI + T::NI, 2 * T::A_PASS, T::A_PASS + T::B_PASS, 1>();
            if constexpr (VAR < 3) __syncthreads();
            if constexpr (VAR == 0 || VAR == 2) {
                t.gload_A(V, ldv, row0, m, k2 * BK, n, vec_ok);
                t.gload_B_kc(V, ldv, col0, m, k2 * BK, n, vec_ok);
            }
            if constexpr (VAR == 2) {
#pragma unroll
                for (int p2 = 0; p2 < T::A_PASS; ++p2) asm volatile("" ::"v"(t.ra[p2]));
#pragma unroll
                for (int p2 = 0; p2 < T::B_PASS; ++p2) asm volatile("" ::"v"(t.rb[p2]));
            }
            if constexpr (VAR < 4) t.template read_frag<0>(nx, 0);
            t.template mma_frag<1>();
            if constexpr (T::MI * T::NI == 32 && !T::EDGE && VAR == 0)
                sched_post_barrier<32, T::A_PASS + T::B_PASS, T::MI + T::NI>();
            cur ^= 1;
        }
        if (sg.whole) {
            t.store_C(G, ldg, row0, col0, m, m, 1.0, 0.0, true);
        } else {
            t.store_slab(slabs + ((int64_t)blockIdx.x * nslot + seg) * T::SLAB_DOUBLES);
        }
        it += ke - kb;
        ++seg;
    }
    }
}

// Direct-to-LDS version of the Gram kernel for interior tiles (same stream-K decomposition and
// slabs): global_load_lds_dwordx4 into three rotating swizzled stages, loads two k-steps ahead,
// counted vmcnt + raw barrier, x applied to the A fragments after the LDS read.
// GV (timing ablations only): bit 0 = no loads inside the k-loop, bit 1 = no wait + barrier,
// bit 2 = no fragment reads, bit 3 = barrier without the vmcnt wait, bit 4 = half of the fragment reads.
template <class T, int GV = 0>
__device__ __forceinline__ void gram_streamk_glds_body(
    const double* __restrict__ V, int64_t ldv, int64_t m, int64_t n, const double* __restrict__ x,
    const TileRC* __restrict__ tiles, const int64_t* __restrict__ wg_ranges, int64_t kiters, int nslot,
    double* __restrict__ slabs,
    double* __restrict__ G, int64_t ldg, int all_slabs) {
    // all_slabs: every segment leaves its accumulators in a slab, also one that covers its whole tile -- the fix-up
    // launch then ADDS the tile to G (a launch over one column block of V behind the first; the plan lists every
    // segment as a contributor).  (Adding in this kernel's own epilogue cost it its register allocation.)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NLD = T::G_NA + T::G_NB + 1;                  // loads per wave per stage
    T t;
    int seg = 0;
    for (int rg = 0; rg < GRAM_RMAX; ++rg) {
    int64_t it = wg_ranges[((int64_t)blockIdx.x * GRAM_RMAX + rg) * 2];
    const int64_t it1 = wg_ranges[((int64_t)blockIdx.x * GRAM_RMAX + rg) * 2 + 1];
    while (it < it1) {
        const GramSeg sg = gram_segment<T>(tiles, kiters, it, it1);
        const int64_t kb = sg.kb, ke = sg.ke, row0 = sg.row0, col0 = sg.col0;
        const int64_t klast = ke - 1;
        t.zero();
        if constexpr (T::WAVES_N == 1) {
            if (sg.dual) {
                // dual diagonal tile: A image = the band at k-step ks and at ks + K/2, no B image
                constexpr int NLDD = T::G_NLD_DUAL;
                const int64_t khalf = (kiters >> 1) * BK;
                t.glds_setup_A_dual(V, ldv, row0, khalf);
                __builtin_amdgcn_s_barrier();
                t.glds_issue_dual(kb * BK, x, khalf, lds);
                t.glds_issue_dual(min(kb + 1, klast) * BK, x, khalf, lds + T::G_STAGE);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLDD) : "memory");
                __builtin_amdgcn_s_barrier();
                int cur = 0;
                t.template read_frag_dual<0>(lds, 0);
#pragma unroll 1
                for (int64_t ks = kb; ks < ke; ++ks) {
                    int nx2 = cur + 2;
                    if (nx2 >= 3) nx2 -= 3;
                    const double* st = lds + cur * T::G_STAGE;
                    t.template scale_dual<0>();
                    __builtin_amdgcn_sched_barrier(0);
                    t.template read_frag_dual<1>(st, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    t.template mma_dual<0>();
                    t.glds_issue_dual(min(ks + 2, klast) * BK, x, khalf, lds + nx2 * T::G_STAGE);
                    __builtin_amdgcn_sched_barrier(0);
                    t.template scale_dual<1>();
                    __builtin_amdgcn_sched_barrier(0);
                    t.template read_frag_dual<0>(st, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    t.template mma_dual<1>();
                    __builtin_amdgcn_sched_barrier(0);
                    t.template scale_dual<0>();
                    __builtin_amdgcn_sched_barrier(0);
                    t.template read_frag_dual<1>(st, 3);
                    __builtin_amdgcn_sched_barrier(0);
                    t.template mma_dual<0>();
                    __builtin_amdgcn_sched_barrier(0);
                    t.template scale_dual<1>();
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(NLDD) : "memory");
                    __builtin_amdgcn_s_barrier();
                    cur = (cur + 1 == 3) ? 0 : cur + 1;
                    t.template read_frag_dual<0>(lds + cur * T::G_STAGE, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    t.template mma_dual<1>();
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
                // always through a slab: the fix-up adds the two K halves (fragment rows i and i + MI/2)
                t.store_slab(slabs + ((int64_t)blockIdx.x * nslot + seg) * T::SLAB_DOUBLES);
                it += ke - kb;
                ++seg;
                continue;
            }
        }
        constexpr int PAIR = (GV & 256) ? 4 : (GV & 128) ? 2 : 1;   // loads of the k-loop per M0 write (mfma_tile.hpp: glds16_run*)
        t.template glds_setup_A<PAIR>(V, ldv, row0);
        t.template glds_setup_B_kc<PAIR>(V, ldv, col0);
        auto issue = [&](int64_t ks, int buf) {
            double* st = lds + buf * T::G_STAGE;
            t.template glds_issue<PAIR>(ks * BK, ks * BK, st);
            t.glds_x(x, ks * BK, st);
        };
        __builtin_amdgcn_s_barrier();                           // previous segment's readers are done
        issue(kb, 0);
        issue(min(kb + 1, klast), 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");   // stage 0 landed (this wave's share)
        __builtin_amdgcn_s_barrier();
        int cur = 0;
        // The barrier sits BEFORE the last fragment group's MFMAs: the first fragments of the next
        // stage are read right behind it and land while those 32 MFMAs occupy the matrix pipe, so
        // the pipe has work queued across the barrier instead of waiting out an LDS round trip.
        t.template read_frag_g<0, true>(lds, 0);
        // The schedule is pinned (sched_barrier): fragment reads of the next group first, then the
        // x-scaling of the set that is about to be used (requested a whole group earlier), then the
        // MFMAs; the loads of stage ks+2 are dealt out between the fragment rows of group 0.
        static_assert(T::MI == 4, "group 0 is dealt out over four fragment rows");
        constexpr int NP = T::G_NA + T::G_NB;
        constexpr int P1 = (NP + 2) / 3, P2 = 2 * P1 < NP ? 2 * P1 : NP;
        if constexpr (GV & 32) {
            // Dealt-out schedule: nothing is issued as a block at a group boundary.  A fragment group is sixteen pairs
            // of MFMAs; behind every pair goes ONE small thing -- a part of the next group's LDS reads, two of the
            // loads of stage ks+2, the multiply that scales the next fragment row -- so that each of them issues in
            // the shadow of the 128 cycles the pair keeps the matrix pipe busy, and the pipe never waits behind a
            // block of reads (seven LDS instructions, four multiplies and their address arithmetic per group before:
            // about 8 % of the kernel by the read ablations).  The barrier sits in the MIDDLE of the last group, and
            // the first fragments of the next stage are read behind it under that group's second half.
#pragma unroll 1
            for (int64_t ks = kb; ks < ke; ++ks) {
                int nx2 = cur + 2;
                if (nx2 >= 3) nx2 -= 3;
                int nx1 = cur + 1;
                if (nx1 >= 3) nx1 -= 3;
                const double* st = lds + cur * T::G_STAGE;
                const double* st1 = lds + nx1 * T::G_STAGE;
                double* nst = lds + nx2 * T::G_STAGE;            // last read in step ks-1
                const int64_t k2 = min(ks + 2, klast) * BK;
                // group g (fragment set S = g & 1): its MFMAs, and the reads of group g+1 into the other set
                auto group = [&](auto gtag) {
                    constexpr int g = decltype(gtag)::value;
                    constexpr int S = g & 1, NS = S ^ 1;
                    static_for<0, 16>([&](auto qtag) {
                        constexpr int q = decltype(qtag)::value;
                        constexpr int I = q >> 2, P = q & 3;
                        if constexpr (P == 0) t.template scale_row<S, I>();
                        __builtin_amdgcn_sched_barrier(0);
                        // placement B (GV & 64): reads early in the group (behind pairs 1..7), the loads of group 0
                        // behind pairs 8..14, the barrier behind the first fragment row of the last group
                        constexpr bool PB = (GV & 64) != 0;
                        constexpr int QBAR = PB ? 4 : 8;
                        if constexpr (g == 3 && q == QBAR && !(GV & 2)) {
                            // stage ks+1 (issued one step ago) has landed: all but the newest NLD loads done;
                            // lgkmcnt(0): this wave's reads of stage ks are complete before others may overwrite it
                            if constexpr (GV & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(NLD) : "memory");
                            __builtin_amdgcn_s_barrier();
                        }
                        t.template mma_pair<S, I, P>();
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (g < 3) {
                            constexpr int rp = PB ? q - 1 : (((q & 1) == 1) ? (q >> 1) : -1);   // read part behind this pair
                            if constexpr (rp >= 0 && rp < 7 && !(GV & 4)) t.template read_part_g<NS, true, rp>(st, g + 1);
                            constexpr int lp = PB ? q - 8 : (((q & 1) == 0) ? (q >> 1) : -1);   // load slot behind this pair
                            if constexpr (g == 0 && lp >= 0 && !(GV & 1)) {
                                constexpr int p0 = 2 * lp;
                                if constexpr (PAIR == 4) {
                                    if constexpr (p0 < NP && p0 % 4 == 0) t.template glds_issue_range<4>(k2, k2, nst, p0, p0 + 4);
                                } else if constexpr (p0 < NP) t.template glds_issue_range<PAIR>(k2, k2, nst, p0, p0 + 2 < NP ? p0 + 2 : NP);
                                if constexpr (p0 == NP || (p0 + 1 == NP)) t.glds_x(x, k2, nst);
                            }
                        } else {
                            // (after the last step this reads the redundant, already landed copy of stage klast)
                            constexpr int rp = q - QBAR - (PB ? 1 : 0);
                            if constexpr (rp >= 0 && rp < 7 && !(GV & 4)) t.template read_part_g<NS, true, rp>(st1, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                };
                group(std::integral_constant<int, 0>{});
                group(std::integral_constant<int, 1>{});
                group(std::integral_constant<int, 2>{});
                group(std::integral_constant<int, 3>{});
                cur = nx1;
            }
        } else {
#pragma unroll 1
        for (int64_t ks = kb; ks < ke; ++ks) {
            int nx2 = cur + 2;
            if (nx2 >= 3) nx2 -= 3;
            const double* st = lds + cur * T::G_STAGE;
            double* nst = lds + nx2 * T::G_STAGE;            // last read in step ks-1
            const int64_t k2 = min(ks + 2, klast) * BK;
            // (the scaling multiplies come BEFORE the next reads are issued: the wait in front of
            //  them then covers only reads that were issued a whole group ago)
            t.template scale_frag<0>();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(GV & 4)) t.template read_frag_g<1, true>(st, 1);
            __builtin_amdgcn_sched_barrier(0);
            t.template mma_row<0>(0);
            if constexpr (!(GV & 1)) t.glds_issue_range(k2, k2, nst, 0, P1);
            t.template mma_row<0>(1);
            if constexpr (!(GV & 1)) t.glds_issue_range(k2, k2, nst, P1, P2);
            t.template mma_row<0>(2);
            if constexpr (!(GV & 1)) { t.glds_issue_range(k2, k2, nst, P2, NP); t.glds_x(x, k2, nst); }
            t.template mma_row<0>(3);
            __builtin_amdgcn_sched_barrier(0);
            t.template scale_frag<1>();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(GV & 4) && !(GV & 16)) t.template read_frag_g<0, true>(st, 2);
            __builtin_amdgcn_sched_barrier(0);
            t.template mma_frag<1>();
            __builtin_amdgcn_sched_barrier(0);
            t.template scale_frag<0>();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(GV & 4)) t.template read_frag_g<1, true>(st, 3);
            __builtin_amdgcn_sched_barrier(0);
            t.template mma_frag<0>();
            __builtin_amdgcn_sched_barrier(0);
            t.template scale_frag<1>();
            __builtin_amdgcn_sched_barrier(0);
            // stage ks+1 (issued one step ago) must have landed: all but the newest NLD loads done;
            // lgkmcnt(0): this wave's reads of stage ks are complete before others may overwrite it
            if constexpr (!(GV & 2)) {
                if constexpr (GV & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(NLD) : "memory");
                __builtin_amdgcn_s_barrier();
            }
            cur = (cur + 1 == 3) ? 0 : cur + 1;
            // (after the last step this reads the redundant, already landed copy of stage klast)
            if constexpr (!(GV & 4) && !(GV & 16)) t.template read_frag_g<0, true>(lds + cur * T::G_STAGE, 0);
            __builtin_amdgcn_sched_barrier(0);
            t.template mma_frag<1>();
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // tail loads / reads retire before LDS is reused
        if (sg.whole && !all_slabs) {
            t.store_C(G, ldg, row0, col0, m, m, 1.0, 0.0, true);
        } else {
            t.store_slab(slabs + ((int64_t)blockIdx.x * nslot + seg) * T::SLAB_DOUBLES);
        }
        it += ke - kb;
        ++seg;
    }
    }
}

constexpr int GRAM_GV = 224;    // schedule of the production Gram kernel: dealt out (32), placement B (64), loads two to an M0 write (128)
template <class T, int GV = 0>
__global__ __launch_bounds__(NTHREADS, 1) void gram_streamk_glds_kernel(
    const double* __restrict__ V, int64_t ldv, int64_t m, int64_t n, const double* __restrict__ x,
    const TileRC* __restrict__ tiles, const int64_t* __restrict__ wg_ranges, int64_t kiters, int nslot,
    double* __restrict__ slabs,
    double* __restrict__ G, int64_t ldg, int all_slabs) {
    gram_streamk_glds_body<T, GV>(V, ldv, m, n, x, tiles, wg_ranges, kiters, nslot, slabs, G, ldg, all_slabs);
}
// The same kernel over the ACTIVE instances of a batch of same-shaped problems (blockIdx.y picks the instance;
// the tile list and the stream-K ranges are shared, matrix / slabs / result come from the instance table, x from
// row `instance` of a K x n array).
template <class T>
__global__ __launch_bounds__(NTHREADS, 1) void gram_streamk_glds_batch_kernel(
    const BatchInst* __restrict__ bt, BatchAct act, int64_t ldv, int64_t m, int64_t n, const double* __restrict__ xbase,
    int64_t ldx, const TileRC* __restrict__ tiles, const int64_t* __restrict__ wg_ranges, int64_t kiters, int nslot,
    int64_t ldg) {
    const int inst = act.idx[blockIdx.y];
    const BatchInst bi = bt[inst];
    gram_streamk_glds_body<T, GRAM_GV>(bi.V, ldv, m, n, xbase + (int64_t)inst * ldx, tiles, wg_ranges, kiters, nslot, bi.slabs,
                                       bi.gram, ldg, 0);
}

// grid = ntiles * 2 * T::MI * FIX_PJ: workgroup (entry, half, part, jq) sums column-fragment group jq of
// fragment row `part` of the slabs of one tile (entry = a tile, or half 0/1 of a pair of dual diagonal
// tiles): many small workgroups, because the pass is bound by HBM latency, not by 72 CUs' worth of adds
constexpr int FIX_PJ = 4;
template <class T>
__device__ __forceinline__ void gram_fixup_body(
    const TileRC* __restrict__ tiles, const int32_t* __restrict__ cstart, const int32_t* __restrict__ contrib,
    int nslot, const double* __restrict__ slabs, double* __restrict__ G, int64_t ldg, int64_t m, double beta) {
    constexpr int JW = (T::NI + FIX_PJ - 1) / FIX_PJ;
    const int jq = blockIdx.x % FIX_PJ;
    const int b = blockIdx.x / FIX_PJ;
    const int e = b / (2 * T::MI), hs = (b / T::MI) & 1, part = b % T::MI;
    const int j0 = jq * JW, j1 = min(T::NI, j0 + JW);
    if (j0 >= T::NI) return;
    const TileRC tr = tiles[e];
    const bool dual = tr.d1 >= 0;
    if (dual && part >= T::MI / 2) return;                  // rows i + MI/2 hold the second K half of rows i
    // contributors of this tile (entry e, half hs): (workgroup, slot) pairs in k order; none when one
    // workgroup ran the whole tile and stored it directly
    const int c0 = cstart[2 * e + hs], c1 = cstart[2 * e + hs + 1];
    if (c0 == c1) return;
    int64_t row0 = (int64_t)tr.rb * T::BM, col0 = (int64_t)tr.cb * T::BN;
    if (dual) row0 = col0 = (int64_t)(hs ? tr.d2 : tr.d1) * (T::BM / 2);
    // column fragments entirely above the diagonal are never stored
    if (col0 + 16 * j0 > row0 + 16 * (part * T::WAVES_M + T::WAVES_M - 1) + 15) return;
    // (no Tile object here: a handful of accumulators per thread instead of the 256-register tile)
    double a[JW][4];
#pragma unroll
    for (int jj = 0; jj < JW; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r) a[jj][r] = 0.0;
    const int tid = threadIdx.x;
    for (int c = c0; c < c1; ++c) {
        const int64_t w = contrib[2 * c];
        const int slot = contrib[2 * c + 1];
        const double* sl = slabs + (w * nslot + slot) * T::SLAB_DOUBLES;
#pragma unroll
        for (int jj = 0; jj < JW; ++jj) {
            const int j = j0 + jj;
            if (j < j1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a[jj][r] += sl[((part * T::NI + j) * 4 + r) * NTHREADS + tid];
                    if (dual) a[jj][r] += sl[(((part + T::MI / 2) * T::NI + j) * 4 + r) * NTHREADS + tid];
                }
            }
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / T::WAVES_N, wn = wave % T::WAVES_N;
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
        const int j = j0 + jj;
        if (j < j1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // (a dual tile's fragment row `part` of wave wm covers row block dual_row_block(part, wm): the same set
                //  of four blocks as part * WAVES_M + wm, dealt to the waves so that each carries 9 fragments)
                const int rblk = dual ? T::dual_row_block(part, wm) : part * T::WAVES_M + wm;
                const int64_t row = row0 + 16 * rblk + lq + 4 * r;
                const int64_t col = col0 + wn * T::WN + 16 * j + lr;
                if (row < m && col < m && col <= row) {
                    double* gp = G + row * ldg + col;
                    *gp = (beta != 0.0) ? a[jj][r] + beta * *gp : a[jj][r];
                }
            }
        }
    }
}

template <class T>
__global__ __launch_bounds__(NTHREADS, 2) void gram_fixup_kernel(
    const TileRC* __restrict__ tiles, const int32_t* __restrict__ cstart, const int32_t* __restrict__ contrib,
    int nslot, const double* __restrict__ slabs, double* __restrict__ G, int64_t ldg, int64_t m, double beta) {
    gram_fixup_body<T>(tiles, cstart, contrib, nslot, slabs, G, ldg, m, beta);
}
template <class T>
__global__ __launch_bounds__(NTHREADS, 2) void gram_fixup_batch_kernel(
    const BatchInst* __restrict__ bt, BatchAct act, const TileRC* __restrict__ tiles, const int32_t* __restrict__ cstart,
    const int32_t* __restrict__ contrib, int nslot, int64_t ldg, int64_t m) {
    const BatchInst bi = bt[act.idx[blockIdx.y]];
    gram_fixup_body<T>(tiles, cstart, contrib, nslot, bi.slabs, bi.gram, ldg, m, 0.0);
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
        const int64_t klast = ksteps - 1;
        {
            const int64_t k1 = min((int64_t)1, klast);
            t.gload_A(W, ldw, row0, m, k1 * BK, Kend, vec_ok_w);
            t.gload_B_km(V, ldv, col0, n, k1 * BK, Kend, vec_ok_v);
        }
        t.template read_frag<0>(lds, 0);
        int cur = 0;
#pragma unroll 1
        for (int64_t ks = 0; ks < ksteps; ++ks) {
            const int64_t k2 = min(ks + 2, klast);
            const double* st = lds + cur * T::STAGE_ELEMS;
            double* nx = lds + (cur ^ 1) * T::STAGE_ELEMS;
            // rows of this wave's fragment i are 16*(i*WAVES_M+wm)..+15; they are all zero in W
            // for this k-step when their last row is above the step's first column.
            const int64_t kd = ks * BK - row0;
            int mi_lo = 0;
            if (kd > 0) {
                const int64_t num = kd - 15 - 16 * wm;
                mi_lo = num > 0 ? (int)((num + 16 * T::WAVES_M - 1) / (16 * T::WAVES_M)) : 0;
            }
            t.template read_frag<1>(st, 1);
            t.template mma_frag<0>(mi_lo);
            t.template read_frag<0>(st, 2);
            t.template mma_frag<1>(mi_lo);
            t.template read_frag<1>(st, 3);
            t.sstore(nx);
            t.template mma_frag<0>(mi_lo);
            __syncthreads();
            t.gload_A(W, ldw, row0, m, k2 * BK, Kend, vec_ok_w);
            t.gload_B_km(V, ldv, col0, n, k2 * BK, Kend, vec_ok_v);
            t.template read_frag<0>(nx, 0);
            t.template mma_frag<1>(mi_lo);
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

// Direct-to-LDS version of the gradient kernel for interior sizes.
// CV: 32 = dealt-out schedule, 64 = its placement B (both development variants), 256 = loads four to an M0 write.
constexpr int COLNORM_CV = 256; // the production gradient kernel: block schedule, loads four to an M0 write
template <class T, int CV = 0>
__device__ __forceinline__ void colnorm_glds_body(
    const double* __restrict__ W, int64_t ldw, const double* __restrict__ V, int64_t ldv, int64_t m, int64_t n,
    double* __restrict__ out, double sign) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* red = lds + 3 * T::G_STAGE;                         // [4][BN] behind the stages
    constexpr int NLD = T::G_NA + T::G_NB;
    const int64_t col0 = (int64_t)blockIdx.x * T::BN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / T::WAVES_N, wn = wave % T::WAVES_N;
    const int lr = lane & 15;
    double colsum[T::NI];
#pragma unroll
    for (int j = 0; j < T::NI; ++j) colsum[j] = 0.0;
    T t;
    const int nrb = (int)(m / T::BM);
    for (int rb = 0; rb < nrb; ++rb) {
        const int64_t row0 = (int64_t)rb * T::BM;
        const int64_t ksteps = (row0 + T::BM) / BK;             // W is lower triangular
        const int64_t klast = ksteps - 1;
        t.zero();
        constexpr int RUN = (CV & 256) ? 4 : 1;                 // loads per M0 write (mfma_tile.hpp: glds16_run4)
        t.template glds_setup_A<RUN>(W, ldw, row0);
        t.template glds_setup_B_km<RUN>(V, ldv, col0);
        auto issue = [&](int64_t ks, int buf) {
            t.template glds_issue<RUN>(ks * BK, ks * BK * ldv, lds + buf * T::G_STAGE);
        };
        __builtin_amdgcn_s_barrier();
        issue(0, 0);
        issue(min((int64_t)1, klast), 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        __builtin_amdgcn_s_barrier();
        int cur = 0;
        t.template read_frag_g<0, false>(lds, 0);
        if constexpr (CV & 32) {
            // dealt-out schedule, as in the Gram kernel (gram_streamk_glds_body, GV & 32)
            constexpr int NP = T::G_NA + T::G_NB;
#pragma unroll 1
            for (int64_t ks = 0; ks < ksteps; ++ks) {
                int nx2 = cur + 2;
                if (nx2 >= 3) nx2 -= 3;
                int nx1 = cur + 1;
                if (nx1 >= 3) nx1 -= 3;
                const double* st = lds + cur * T::G_STAGE;
                const double* st1 = lds + nx1 * T::G_STAGE;
                double* nst = lds + nx2 * T::G_STAGE;
                const int64_t kd = ks * BK - row0;
                int mi_lo = 0;
                if (kd > 0) {
                    const int64_t num = kd - 15 - 16 * wm;
                    mi_lo = num > 0 ? (int)((num + 16 * T::WAVES_M - 1) / (16 * T::WAVES_M)) : 0;
                }
                const int64_t k2 = min(ks + 2, klast) * BK;
                auto group = [&](auto gtag) {
                    constexpr int g = decltype(gtag)::value;
                    constexpr int S = g & 1, NS = S ^ 1;
                    static_for<0, 16>([&](auto qtag) {
                        constexpr int q = decltype(qtag)::value;
                        constexpr int I = q >> 2, P = q & 3;
                        constexpr bool PB = (CV & 64) != 0;
                        constexpr int QBAR = PB ? 4 : 8;
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (g == 3 && q == QBAR) {
                            asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(NLD) : "memory");
                            __builtin_amdgcn_s_barrier();
                        }
                        t.template mma_pair<S, I, P>(mi_lo);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (g < 3) {
                            constexpr int rp = PB ? q - 1 : (((q & 1) == 1) ? (q >> 1) : -1);
                            if constexpr (rp >= 1 && rp < 7) t.template read_part_g<NS, false, rp>(st, g + 1);
                            constexpr int lp = PB ? q - 8 : (((q & 1) == 0) ? (q >> 1) : -1);
                            if constexpr (g == 0 && lp >= 0) {
                                constexpr int p0 = 2 * lp;
                                if constexpr (p0 < NP) t.glds_issue_range(k2, k2 * ldv, nst, p0, p0 + 2 < NP ? p0 + 2 : NP);
                            }
                        } else {
                            constexpr int rp = q - QBAR - (PB ? 1 : 0);
                            if constexpr (rp >= 1 && rp < 7) t.template read_part_g<NS, false, rp>(st1, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                };
                group(std::integral_constant<int, 0>{});
                group(std::integral_constant<int, 1>{});
                group(std::integral_constant<int, 2>{});
                group(std::integral_constant<int, 3>{});
                cur = nx1;
            }
        } else {
#pragma unroll 1
        for (int64_t ks = 0; ks < ksteps; ++ks) {
            int nx2 = cur + 2;
            if (nx2 >= 3) nx2 -= 3;
            const double* st = lds + cur * T::G_STAGE;
            const int64_t kd = ks * BK - row0;
            int mi_lo = 0;
            if (kd > 0) {
                const int64_t num = kd - 15 - 16 * wm;
                mi_lo = num > 0 ? (int)((num + 16 * T::WAVES_M - 1) / (16 * T::WAVES_M)) : 0;
            }
            // pinned schedule as in the Gram kernel: reads of the next group, then the MFMAs, the loads
            // of stage ks+2 dealt out between the fragment rows of group 0
            constexpr int NP = T::G_NA + T::G_NB;
            constexpr int P1 = (NP + 2) / 3, P2 = 2 * P1 < NP ? 2 * P1 : NP;
            static_assert(T::MI == 4, "group 0 is dealt out over four fragment rows");
            double* nst = lds + nx2 * T::G_STAGE;
            const int64_t k2 = min(ks + 2, klast) * BK;
            t.template read_frag_g<1, false>(st, 1);
            __builtin_amdgcn_sched_barrier(0);
            t.template mma_row<0>(0, mi_lo);
            t.template glds_issue_range<RUN>(k2, k2 * ldv, nst, 0, P1);
            t.template mma_row<0>(1, mi_lo);
            t.template glds_issue_range<RUN>(k2, k2 * ldv, nst, P1, P2);
            t.template mma_row<0>(2, mi_lo);
            t.template glds_issue_range<RUN>(k2, k2 * ldv, nst, P2, NP);
            t.template mma_row<0>(3, mi_lo);
            __builtin_amdgcn_sched_barrier(0);
            t.template read_frag_g<0, false>(st, 2);
            __builtin_amdgcn_sched_barrier(0);
            t.template mma_frag<1>(mi_lo);
            __builtin_amdgcn_sched_barrier(0);
            t.template read_frag_g<1, false>(st, 3);
            __builtin_amdgcn_sched_barrier(0);
            t.template mma_frag<0>(mi_lo);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(NLD) : "memory");
            __builtin_amdgcn_s_barrier();
            cur = (cur + 1 == 3) ? 0 : cur + 1;
            // first fragments of the next stage land under the last group's MFMAs
            t.template read_frag_g<0, false>(lds + cur * T::G_STAGE, 0);
            __builtin_amdgcn_sched_barrier(0);
            t.template mma_frag<1>(mi_lo);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
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
#pragma unroll
    for (int j = 0; j < T::NI; ++j) {
        double s = colsum[j];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        colsum[j] = s;
    }
    __syncthreads();
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j < T::NI; ++j) red[wm * T::BN + wn * T::WN + 16 * j + lr] = colsum[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < T::BN; c += NTHREADS) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < T::WAVES_M; ++w) s += red[w * T::BN + c];
        if (col0 + c < n) out[col0 + c] = sign * s;
    }
}

template <class T, int CV = 0>
__global__ __launch_bounds__(NTHREADS, 1) void colnorm_glds_kernel(
    const double* __restrict__ W, int64_t ldw, const double* __restrict__ V, int64_t ldv, int64_t m, int64_t n,
    double* __restrict__ out, double sign) {
    colnorm_glds_body<T, CV>(W, ldw, V, ldv, m, n, out, sign);
}
template <class T>
__global__ __launch_bounds__(NTHREADS, 1) void colnorm_glds_batch_kernel(
    const BatchInst* __restrict__ bt, BatchAct act, int64_t ldw, int64_t ldv, int64_t m, int64_t n,
    double* __restrict__ outbase, int64_t ldo, double sign) {
    const int inst = act.idx[blockIdx.y];
    const BatchInst bi = bt[inst];
    colnorm_glds_body<T, COLNORM_CV>(bi.Wbuf, ldw, bi.V, ldv, m, n, outbase + (int64_t)inst * ldo, sign);
}

// =========================================================================================
// Batched small GEMM on the 64x64 tile: C = alpha*A*op(B) + beta*C for a table of products.
// grid = (tiles_n, tiles_m, nops).  Used by the Cholesky trailing update and the inverse merges.
// =========================================================================================
template <class T>
__device__ __forceinline__ void gemm_ops_body(const GemmOp op) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int64_t row0 = (int64_t)blockIdx.y * T::BM, col0 = (int64_t)blockIdx.x * T::BN;
    if (row0 >= op.M || col0 >= op.N) return;
    if (op.lower_only && col0 > row0 + T::BM - 1) return;
    const bool va = ((reinterpret_cast<uintptr_t>(op.A) & 15) == 0) && ((op.lda & 1) == 0);
    const bool vb = ((reinterpret_cast<uintptr_t>(op.B) & 15) == 0) && ((op.ldb & 1) == 0);
    T t;
    t.zero();
    int64_t ksteps = (op.K + BK - 1) / BK;
    int64_t ks0 = 0;
    if (op.tri & 1) ks0 = max((int64_t)0, col0 - op.koff) / BK;             // triangular B: leading zeros
    if (op.tri & 2) ksteps = min(ksteps, max((int64_t)0, row0 + T::BM - op.koff + BK - 1) / BK);   // triangular A: trailing zeros
    if (ks0 >= ksteps) {                                                    // nothing but zeros in this piece
        t.store_C(op.C, op.ldc, row0, col0, op.M, op.N, op.alpha, op.beta, op.lower_only != 0);
        return;
    }
    auto gload = [&](int64_t ks) {
        t.gload_A(op.A, op.lda, row0, op.M, ks * BK, op.K, va);
        if constexpr (T::BKM) t.gload_B_km(op.B, op.ldb, col0, op.N, ks * BK, op.K, vb);
        else t.gload_B_kc(op.B, op.ldb, col0, op.N, ks * BK, op.K, vb);
    };
    gload(ks0);
    t.sstore(lds);
    __syncthreads();
    int cur = 0;
    for (int64_t ks = ks0; ks < ksteps; ++ks) {
        const bool more = ks + 1 < ksteps;
        if (more) gload(ks + 1);
        t.compute(lds + cur * T::STAGE_ELEMS);
        if (more) t.sstore(lds + (cur ^ 1) * T::STAGE_ELEMS);
        __syncthreads();
        cur ^= 1;
    }
    t.store_C(op.C, op.ldc, row0, col0, op.M, op.N, op.alpha, op.beta, op.lower_only != 0);
}

template <class T>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_ops_kernel(const GemmOp* __restrict__ ops) {
    gemm_ops_body<T>(ops[blockIdx.z]);
}
// products of the ACTIVE instances of a batch: `ops` holds the op tables of all instances one after the other
// (`per_inst` ops each, the same stage of every instance at the same offset)
template <class T>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_ops_batch_kernel(const GemmOp* __restrict__ ops, int per_inst, int begin,
                                                                    int count, BatchAct act) {
    const int a = blockIdx.z / count, o = blockIdx.z - a * count;
    gemm_ops_body<T>(ops[(int64_t)act.idx[a] * per_inst + begin + o]);
}

// split-K reduction: C = sum of the ns partial products (fixed order)
__device__ __forceinline__ void gemm_reduce_body(const RedOp r) {
    const int64_t total = (int64_t)r.M * r.N;
    const int64_t stride = (int64_t)gridDim.x * 256 * 2;
    for (int64_t e = 2 * ((int64_t)blockIdx.x * 256 + threadIdx.x); e < total; e += stride) {
        double2 acc = *reinterpret_cast<const double2*>(r.P + e);
        for (int p = 1; p < r.ns; ++p) {
            const double2 v = *reinterpret_cast<const double2*>(r.P + (int64_t)p * total + e);
            acc.x += v.x;
            acc.y += v.y;
        }
        const int64_t row = e / r.N, col = e - row * r.N;
        *reinterpret_cast<double2*>(r.C + row * r.ldc + col) = acc;
    }
}
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const RedOp* __restrict__ reds) { gemm_reduce_body(reds[blockIdx.y]); }
__global__ __launch_bounds__(256) void gemm_reduce_batch_kernel(const RedOp* __restrict__ reds, int per_inst, int begin, int count,
                                                               BatchAct act) {
    const int a = blockIdx.y / count, o = blockIdx.y - a * count;
    gemm_reduce_body(reds[(int64_t)act.idx[a] * per_inst + begin + o]);
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
// Cholesky, right-looking, block NB = 64, ONE launch per block column.
//
// Launch kc (= 0 .. T-1) receives block column kprev = kc - 1 finished (L(:,kprev) in place) and
//   * panel workgroups (one per block row i >= kc): bring the diagonal block
//     D = A(kc,kc) - L(kc,kprev) L(kc,kprev)^T up to date (each redundantly: 64^3 MFMA work),
//     factor it in LDS, then either publish it (i == kc: factor block to Ldiag -- NOT over A(kc,kc), which
//     the other panel workgroups of the launch still read -- its log-det share, and on request its inverse)
//     or solve their own panel block  L(i,kc) = (A(i,kc) - L(i,kprev) L(kc,kprev)^T) L(kc,kc)^-T;
//   * update workgroups: A(i,j) -= L(i,kprev) L(j,kprev)^T for the remaining tiles j > kc.
// Everything a launch reads was written by earlier launches, so there is no hand-off between
// workgroups inside a launch; T launches factor the matrix.
//
// From T = 64 block columns on (m > 4032) the rank-64 update of the whole trailing matrix per launch is
// bound by HBM (the trailing matrix is read and written once per block column: 8 flop per byte), and the
// factorisation runs in two levels: the step launches update only the eight block columns of their outer
// panel, and chol_syrk_kernel applies the finished panel to everything behind it in one pass (rank 512,
// an eighth of the traffic).
// =========================================================================================
constexpr int SP = NB + 1;   // LDS stride of a 64x64 block image (row reads by one lane per row)
constexpr int SQ = NB + 2;   // LDS stride of a 64x64 MFMA operand image (ds_read_b64 conflict-free)
constexpr int CHOL_LDS_DOUBLES = 2 * NB * SQ + 2 * NB * SP + 6 * NB;
constexpr int CHOL_LDS_BYTES = CHOL_LDS_DOUBLES * 8;
constexpr int SYRK_LDS_BYTES = 2 * NB * SQ * 8;

typedef d4 acc64_t[2][2];   // 64x64 product on 4 waves (2x2), wave tile 32x32 = 2x2 MFMA fragments

// acc = As * Bs^T for two 64x64 operand images (k-contiguous, stride SQ) resident in LDS
__device__ __forceinline__ void mma64(acc64_t& acc, const double* __restrict__ As, const double* __restrict__ Bs) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lq = lane >> 4;
    const double* ap = As + (16 * wm + lr) * SQ + lq;          // fragment i: rows 16*(2i+wm)
    const double* bp = Bs + (32 * wn + lr) * SQ + lq;          // fragment j: cols 32*wn + 16j
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int kk = 0; kk < NB / 4; ++kk) {
        const double a0 = ap[4 * kk], a1 = ap[32 * SQ + 4 * kk];
        const double b0 = bp[4 * kk], b1 = bp[16 * SQ + 4 * kk];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
}

// img[row][col] -= acc (image stride SP), guarded by (row < mr && col < mc) and optionally col <= row
__device__ __forceinline__ void sub_acc64(double* __restrict__ img, const acc64_t& acc, int mr, int mc, bool lower) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * (2 * i + wm) + lq + 4 * r, col = 32 * wn + 16 * j + lr;
                if (row < mr && col < mc && (!lower || col <= row)) img[row * SP + col] -= acc[i][j][r];
            }
}

// 64x64 block of a row-major matrix -> registers (16 doubles per thread, 16-byte pieces), zero padded
struct Blk64 {
    d2 v[8];
    __device__ __forceinline__ void load(const double* __restrict__ G, int64_t ld, int mr, int mc, bool vec_ok) {
        const int tid = threadIdx.x;
        const int c = 2 * (tid & 31), r0 = tid >> 5;           // 32 threads cover a 64-double row
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int r = r0 + 8 * p;
            v[p] = load2_guard(G + (int64_t)r * ld + c, r < mr, c, mc, vec_ok);
        }
    }
    // into an image with row stride `st`
    __device__ __forceinline__ void to_lds(double* __restrict__ img, int st) const {
        const int tid = threadIdx.x;
        const int c = 2 * (tid & 31), r0 = tid >> 5;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            img[(r0 + 8 * p) * st + c] = v[p].x;
            img[(r0 + 8 * p) * st + c + 1] = v[p].y;
        }
    }
    __device__ __forceinline__ void from_lds(const double* __restrict__ img, int st) {
        const int tid = threadIdx.x;
        const int c = 2 * (tid & 31), r0 = tid >> 5;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            v[p].x = img[(r0 + 8 * p) * st + c];
            v[p].y = img[(r0 + 8 * p) * st + c + 1];
        }
    }
    __device__ __forceinline__ void store(double* __restrict__ G, int64_t ld, int mr, int mc, bool lower) const {
        const int tid = threadIdx.x;
        const int c = 2 * (tid & 31), r0 = tid >> 5;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int r = r0 + 8 * p;
            if (r < mr) {
                if (c < mc && (!lower || c <= r)) G[(int64_t)r * ld + c] = v[p].x;
                if (c + 1 < mc && (!lower || c + 1 <= r)) G[(int64_t)r * ld + c + 1] = v[p].y;
            }
        }
    }
};

// 1/sqrt(d) to fp64 accuracy: v_rsq_f64 plus two Newton steps (the pivot chain of the factorisation
// is latency-bound; this is a third of the dependent depth of sqrt followed by a divide).
__device__ __forceinline__ double rsqrt_newton(double d) {
    double y = __builtin_amdgcn_rsq(d);
    const double h = -0.5 * d;
    y = y * fma(h * y, y, 1.5);
    y = y * fma(h * y, y, 1.5);
    return y;
}

// quad broadcast on DPP: every lane of a quad gets lane `SRC`'s value, no LDS round trip
template <int SRC>
__device__ __forceinline__ double quad_bcast(double v) {
    constexpr int ctrl = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, ctrl, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, ctrl, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// x L16^T = p for one row of 16 entries P[row][k0 .. k0+15] -> X[row][k0 .. k0+15], four lanes per row: lane q of the quad keeps
// the entries of the columns 4t+q.  Per column one multiply, one quad broadcast and the updates of the
// columns to the right; lt[c*16 + cc] = L16[cc][c] (zero above the diagonal), rv[c] = 1/L16[c][c].
__device__ __forceinline__ void solve16_quad(const double* P, double* X, int row, int k0,
                                             const double* __restrict__ lt, const double* __restrict__ rv) {
    const int q = threadIdx.x & 3;
    const double* prow = P + row * SP + k0;
    double* xr = X + row * SP + k0;                               // may be the same image as P
    double pr[4], xs[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) pr[t] = prow[4 * t + q];
    // The factor entries this lane needs are fetched four columns at a time, one group ahead of the dependent
    // chain (the stores of the results come after it: LDS stores in between would pin the loads behind them).
    // Holding all sixteen columns' entries at once costs 110 registers, which the one-launch Cholesky -- two
    // workgroups per CU -- does not have.
    double lv[2][4][4], rvv[2][4];
    auto group = [&](auto gtag) {
        constexpr int g = decltype(gtag)::value;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int c = 4 * g + cc;
            rvv[g & 1][cc] = rv[c];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (4 * t + 3 > c) lv[g & 1][cc][t] = lt[c * 16 + 4 * t + q];
        }
    };
    auto column = [&](auto ctag) {
        constexpr int c = decltype(ctag)::value;
        constexpr int qc = c & 3, tc = c >> 2, g = c >> 2;
        const double x = quad_bcast<qc>(pr[tc] * rvv[g & 1][c & 3]);
        xs[tc] = (q == qc) ? x : xs[tc];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (4 * t + 3 > c) pr[t] = fma(-x, lv[g & 1][c & 3][t], pr[t]);      // (a finished entry may take junk)
    };
    group(std::integral_constant<int, 0>{});
    group(std::integral_constant<int, 1>{});
    column(std::integral_constant<int, 0>{});  column(std::integral_constant<int, 1>{});
    column(std::integral_constant<int, 2>{});  column(std::integral_constant<int, 3>{});
    group(std::integral_constant<int, 2>{});
    column(std::integral_constant<int, 4>{});  column(std::integral_constant<int, 5>{});
    column(std::integral_constant<int, 6>{});  column(std::integral_constant<int, 7>{});
    group(std::integral_constant<int, 3>{});
    column(std::integral_constant<int, 8>{});  column(std::integral_constant<int, 9>{});
    column(std::integral_constant<int, 10>{}); column(std::integral_constant<int, 11>{});
    column(std::integral_constant<int, 12>{}); column(std::integral_constant<int, 13>{});
    column(std::integral_constant<int, 14>{}); column(std::integral_constant<int, 15>{});
#pragma unroll
    for (int t = 0; t < 4; ++t) xr[4 * t + q] = xs[t];
}

// value of lane `SRC` of each 16-lane row, in every lane of that row: one 64-bit DPP move (row_newbcast is the
// DPP control the fp64 ALU ops of gfx90a+ accept), no SGPR or LDS round trip
template <int SRC>
__device__ __forceinline__ double row_bcast(double v) {
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + SRC, 0xF, 0xF, true);
}
// acc += l[lane C of this row] * nl as ONE instruction: v_fmac_f64 with the row broadcast applied to its
// first source.  hipcc keeps the broadcast and the FMA apart (three instructions with 32-bit moves, two with
// the 64-bit move), and the 16 x 16 factorisation below is bound by its instruction count.  Inline asm is
// opaque to the hazard recogniser: NOPS adds the two wait states a DPP read needs after a VALU write of its
// source (first update of a step) or that a following DPP read of `acc` needs (last one).
#ifndef ACCBPG_DPP_ASM
#define ACCBPG_DPP_ASM 1
#endif
template <int C, int NOPS_BEFORE, int NOPS_AFTER>
__device__ __forceinline__ void fmac_row_bcast(double& acc, double l, double nl) {
#if ACCBPG_DPP_ASM
    if constexpr (NOPS_BEFORE) asm volatile("s_nop 1");
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc)
                 : "v"(l), "v"(nl), "n"(C));
    if constexpr (NOPS_AFTER) asm volatile("s_nop 1");
#else
    acc = fma(row_bcast<C>(l), nl, acc);       // compiler-managed hazards (two instructions)
#endif
}
// out = y[lane C of this row] * v as one instruction (v_mul_f64 has no DPP form on gfx950: v_fmac_f64 onto a
// zero, same rounding)
template <int C>
__device__ __forceinline__ double mul_row_bcast(double y, double v) {
#if ACCBPG_DPP_ASM
    double out = 0.0;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(out)
                 : "v"(y), "v"(v), "n"(C));
    return out;
#else
    return fma(row_bcast<C>(y), v, 0.0);
#endif
}
// rank-1 updates of the columns [C0, C1) of step J (clipped to 15)
template <int C0, int C1>
__device__ __forceinline__ void potrf16_updates(double (&a)[16], double l, double nl) {
    if constexpr (C0 < C1 && C0 < 16) {
        fmac_row_bcast<C0, 0, (C0 == 15)>(a[C0], l, nl);
        potrf16_updates<C0 + 1, C1>(a, l, nl);
    }
}
// Step J of the 16x16 factorisation: `l` holds column J of the factor (entry of this lane's row).  One
// wavefront, in-order issue: the step is bound by its dependent chain and its instruction count, so
//   * every rank-1 update is a single v_fmac_f64 with a DPP row broadcast;
//   * every lane takes the reciprocal square root of ITS entry of column J+1 (only lane J+1's is the
//     pivot's; the others are never used), so the pivot is not broadcast before the rsq and the scaled
//     column is one v_mul_f64 with the broadcast of lane J+1's factor: six dependent operations per column;
//   * that factor is v_rsq_f64 refined by ONE Newton step (2^-26 -> about 2^-51 relative; it only scales
//     the column, the diagonal entry and the log-determinant use a correctly rounded sqrt of the pivot).
// Lane r keeps its own pivot (dsv), its factor (myrs) and the smallest pivot it saw (minp).
template <int J>
__device__ __forceinline__ void potrf16_step(double (&a)[16], double& l, int r, int k0, int bs, double& dsv,
                                             double& myrs, double& minp) {
    a[J] = l;
    if constexpr (J + 1 < 16) {
        const double nl = -l;
        fmac_row_bcast<J + 1, 1, 0>(a[J + 1], l, nl);                // lane J+1: the next pivot
        const double p = a[J + 1];
        // the independent updates are dealt out between the links of the dependent chain (pinned: left
        // alone they are issued as one block in the middle of it)
        constexpr int N = 14 - J, Q = (N + 3) / 4;                   // updates left, per gap
        double y = __builtin_amdgcn_rsq(p);                          // (a non-positive pivot spreads NaNs; minp reports it)
        const double h = -0.5 * p;
        __builtin_amdgcn_sched_barrier(0);
        potrf16_updates<J + 2, J + 2 + Q>(a, l, nl);
        __builtin_amdgcn_sched_barrier(0);
        const double t = h * y;
        __builtin_amdgcn_sched_barrier(0);
        potrf16_updates<J + 2 + Q, J + 2 + 2 * Q>(a, l, nl);
        __builtin_amdgcn_sched_barrier(0);
        const double u = fma(t, y, 1.5);
        __builtin_amdgcn_sched_barrier(0);
        potrf16_updates<J + 2 + 2 * Q, J + 2 + 3 * Q>(a, l, nl);
        __builtin_amdgcn_sched_barrier(0);
        y = y * u;
        __builtin_amdgcn_sched_barrier(0);
        potrf16_updates<J + 2 + 3 * Q, 16>(a, l, nl);
        __builtin_amdgcn_sched_barrier(0);
        const bool mine = (r == J + 1);
        dsv = mine ? p : dsv;
        myrs = mine ? y : myrs;
        minp = (mine && (k0 + J + 1 < bs)) ? p : minp;
        l = mul_row_bcast<J + 1>(y, p);
    }
}
// Blocked factorisation of the 64x64 block image S (lower, identity-padded beyond bs) into Lo, in four
// panels of 16 columns:
//   A  one wavefront factors the 16x16 diagonal block with a row per lane in registers -- the pivot and
//      the column entries travel by DPP row broadcasts, no LDS round trip and no barrier in the 16 steps;
//   B  the rows below solve against it, four lanes per row (solve16_quad);
//   C  the trailing blocks take their rank-16 update on the MFMA pipe.
// Three barriers per panel instead of one per column; the serial chain is 64 register-resident steps.
// `tbuf`: 4 * POTRF_TB doubles of scratch.  Lo gets the lower triangle, zeros above it.
constexpr int POTRF_TB = 16 * 16 + 16;
// `hook(pnl, where)` lets the one-launch Cholesky publish the factor panel by panel while it is being formed:
// where = 0 just before and 1 just after the barrier that ends phase A of panel pnl, 3 on the wavefronts 1..3 while
// wavefront 0 factors the 16x16 block of panel pnl (the 16 columns of panel pnl - 1 are final by then).  The
// launch-per-column kernel passes nothing.
struct PotrfNoHook { __device__ __forceinline__ void operator()(int, int) const {} };
template <class Hook = PotrfNoHook>
__device__ __forceinline__ void potrf64_blk(double* __restrict__ S, double* __restrict__ Lo,
                                            double* __restrict__ tbuf, int* __restrict__ badflag, int bs,
                                            int dbg = 0, Hook hook = Hook()) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // per panel: lt[c*16 + cc] = L16[cc][c] (column c of the panel's diagonal factor), rv[c] = 1 / L16[c][c];
    // all four panels' copies stay in tbuf for the panel solve that follows (trsm64_blk)
    // zero the six 16x16 blocks of Lo above the diagonal blocks (nothing below writes them)
    for (int e = tid; e < 6 * 256; e += NTHREADS) {
        const int b = e >> 8, w = e & 255;                       // b: (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
        const int br = (b < 3) ? 0 : ((b < 5) ? 1 : 2);
        const int bc = (b < 3) ? b + 1 : ((b < 5) ? b - 1 : 3);
        Lo[(16 * br + (w >> 4)) * SP + 16 * bc + (w & 15)] = 0.0;
    }
#pragma unroll 1
    for (int pnl = 0; pnl < NB / 16; ++pnl) {
        const int k0 = 16 * pnl;
        double* lt = tbuf + pnl * POTRF_TB;
        double* rv = lt + 256;
        // ---- A: 16x16 diagonal block on wavefront 0 (every lane l works on row l & 15)
        if (wave == 0 && !(dbg & 32)) {
            const int r = lane & 15;
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = S[(k0 + r) * SP + k0 + c];
            double dsv = 1.0, myrs = 1.0, minp = 1.0;
            double l;
            {
                const double piv = row_bcast<0>(a[0]);
                const double rs = rsqrt_newton(piv);
                dsv = (r == 0) ? piv : dsv;
                myrs = (r == 0) ? rs : myrs;
                minp = (r == 0 && k0 < bs) ? piv : 1.0;
                l = a[0] * rs;
            }
            potrf16_step<0>(a, l, r, k0, bs, dsv, myrs, minp);
            potrf16_step<1>(a, l, r, k0, bs, dsv, myrs, minp);
            potrf16_step<2>(a, l, r, k0, bs, dsv, myrs, minp);
            potrf16_step<3>(a, l, r, k0, bs, dsv, myrs, minp);
            potrf16_step<4>(a, l, r, k0, bs, dsv, myrs, minp);
            potrf16_step<5>(a, l, r, k0, bs, dsv, myrs, minp);
            potrf16_step<6>(a, l, r, k0, bs, dsv, myrs, minp);
            potrf16_step<7>(a, l, r, k0, bs, dsv, myrs, minp);
            potrf16_step<8>(a, l, r, k0, bs, dsv, myrs, minp);
            potrf16_step<9>(a, l, r, k0, bs, dsv, myrs, minp);
            potrf16_step<10>(a, l, r, k0, bs, dsv, myrs, minp);
            potrf16_step<11>(a, l, r, k0, bs, dsv, myrs, minp);
            potrf16_step<12>(a, l, r, k0, bs, dsv, myrs, minp);
            potrf16_step<13>(a, l, r, k0, bs, dsv, myrs, minp);
            potrf16_step<14>(a, l, r, k0, bs, dsv, myrs, minp);
            potrf16_step<15>(a, l, r, k0, bs, dsv, myrs, minp);
            const bool bad = __any(!(minp > 0.0));               // some row's pivot was not positive (or NaN)
            const double dg = sqrt(dsv);
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const double v = (c < r) ? a[c] : ((c == r) ? dg : 0.0);
                    Lo[(k0 + r) * SP + k0 + c] = v;
                    lt[c * 16 + r] = v;
                }
                rv[r] = myrs;
                if (bad) *badflag = 1;
            }
        } else {
            hook(pnl, 3);                                        // the three wavefronts that idle during phase A
        }
        hook(pnl, 0);
        __syncthreads();
        hook(pnl, 1);
        if (pnl == NB / 16 - 1) break;
        // ---- B: rows below the diagonal block: x L16^T = p, four lanes per row
        const int nrows = NB - k0 - 16;
        if (tid < 4 * nrows && !(dbg & 8)) {
            const int row = k0 + 16 + (tid >> 2);
            solve16_quad(S, Lo, row, k0, lt, rv);
        }
        __syncthreads();
        // ---- C: trailing update S(bi,bj) -= X_bi X_bj^T on the MFMA pipe, 16x16 blocks bi >= bj > pnl
        {
            const int nb = NB / 16 - 1 - pnl;                      // remaining 16-blocks per dimension
            const int nblk = nb * (nb + 1) / 2;
            const int lr = lane & 15, lq = lane >> 4;
            for (int b = wave; b < nblk && !(dbg & 16); b += NTHREADS / 64) {
                int bi = 0;
                while ((bi + 1) * (bi + 2) / 2 <= b) ++bi;
                const int bj = b - bi * (bi + 1) / 2;
                const int ri = 16 * (pnl + 1 + bi), rj = 16 * (pnl + 1 + bj);
                d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double av = Lo[(ri + lr) * SP + k0 + 4 * kk + lq];
                    const double bv = Lo[(rj + lr) * SP + k0 + 4 * kk + lq];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) S[(ri + lq + 4 * rr) * SP + rj + lr] -= acc[rr];
            }
        }
        __syncthreads();
    }
}

// Solve X * L^T = P for the 64 rows of Xs in place with the factor potrf64_blk left in Lo / tbuf, in four
// panels of 16 columns: a thread per row substitutes against the 16x16 diagonal factor (LDS broadcasts),
// then the columns to the right take their rank-16 update on the MFMA pipe.  Two barriers per panel.
__device__ __forceinline__ void trsm64_blk(double* __restrict__ Xs, const double* __restrict__ Lo,
                                           const double* __restrict__ tbuf) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll 1
    for (int pnl = 0; pnl < NB / 16; ++pnl) {
        const int k0 = 16 * pnl;
        const double* lt = tbuf + pnl * POTRF_TB;
        const double* rv = lt + 256;
        solve16_quad(Xs, Xs, tid >> 2, k0, lt, rv);
        if (pnl == NB / 16 - 1) break;
        __syncthreads();
        {
            // P(:, 16q ..) -= X_p L(16q .., k0 .. k0+15)^T for the column blocks q > pnl, 16 x 16 pieces
            const int ncb = NB / 16 - 1 - pnl;
            const int lr = lane & 15, lq = lane >> 4;
            for (int b = wave; b < 4 * ncb; b += NTHREADS / 64) {
                const int bi = b & 3, q = pnl + 1 + (b >> 2);
                d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double av = Xs[(16 * bi + lr) * SP + k0 + 4 * kk + lq];
                    const double bv = Lo[(16 * q + lr) * SP + k0 + 4 * kk + lq];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) Xs[(16 * bi + lq + 4 * rr) * SP + 16 * q + lr] -= acc[rr];
            }
        }
        __syncthreads();
    }
}

// trsm64_blk for the one-launch Cholesky: the factor of the block column arrives panel by panel (16 columns at a
// time) while its owner is still factoring the panels to the right.  `fetch(pnl, 0)` waits for panel pnl and
// brings its piece (the 16x16 factor copy + reciprocals into tbuf, the rows of the factor below it into Lo) into
// LDS; it returns false when the launch is being abandoned.  `fetch(pnl, 1)` asks whether the piece has been flagged
// and `fetch(pnl, 2)` acts on the answer: if so it starts the piece's loads, which then land behind the work on the
// current panel.  `done(pnl)` runs once the 16 columns of panel pnl of the solution are final (every thread, behind a
// barrier): the owner publishes them and folds them into what follows while the next piece is on its way.
// Same arithmetic as trsm64_blk.  (Issuing the fold's matrix-pipe work in front of the NEXT panel's solve instead, to
// run under its dependent vector chain, was measured slower: its LDS reads queue ahead of the solve's.)
template <class Fetch, class Done>
__device__ __forceinline__ bool trsm64_stream(double* __restrict__ Xs, const double* __restrict__ Lo,
                                              const double* __restrict__ tbuf, Fetch fetch, Done done) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll 1
    for (int pnl = 0; pnl < NB / 16; ++pnl) {
        const int k0 = 16 * pnl;
        const double* lt = tbuf + pnl * POTRF_TB;
        const double* rv = lt + 256;
        if (!fetch(pnl, 0)) return false;                         // ends with a barrier: the piece is in LDS
        if (pnl + 1 < NB / 16) fetch(pnl + 1, 1);                 // look for the next piece (the answer lands behind the solve)
        solve16_quad(Xs, Xs, tid >> 2, k0, lt, rv);
        if (pnl + 1 < NB / 16) fetch(pnl + 1, 2);                 // if it is there: its loads in flight behind what follows
        __syncthreads();
        if (!done(pnl)) return false;
        if (pnl == NB / 16 - 1) break;
        {
            // P(:, 16q ..) -= X_p L(16q .., k0 .. k0+15)^T for the column blocks q > pnl, 16 x 16 pieces
            const int ncb = NB / 16 - 1 - pnl;
            const int lr = lane & 15, lq = lane >> 4;
            for (int b = wave; b < 4 * ncb; b += NTHREADS / 64) {
                const int bi = b & 3, q = pnl + 1 + (b >> 2);
                d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double av = Xs[(16 * bi + lr) * SP + k0 + 4 * kk + lq];
                    const double bv = Lo[(16 * q + lr) * SP + k0 + 4 * kk + lq];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) Xs[(16 * bi + lq + 4 * rr) * SP + 16 * q + lr] -= acc[rr];
            }
        }
        // (the barrier inside the next fetch separates these writes from the next panel's solve)
    }
    return true;
}

// kc: block column to factor; pend: the rank-64 update from block column kc - 1 is still owed to the
// columns kc .. jmax (false for the first column of the matrix and, in the two-level scheme, for the first
// column of an outer panel, which chol_syrk_kernel has brought up to date); jmax: last block column the
// update workgroups touch (T - 1 in the one-level scheme, the end of the outer panel otherwise).
__global__ __launch_bounds__(NTHREADS, 1) void chol_step_kernel(double* __restrict__ A, int64_t lda, int64_t m,
                                                               int kc, int pend, int jmax, int T,
                                                               double* __restrict__ logdet,
                                                               int* __restrict__ flags, int dbg,
                                                               double* __restrict__ Winv, double* __restrict__ Ldiag) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* Ak = lds;                       // 64 x SQ : L(i,kprev) operand image
    double* Bk = Ak + NB * SQ;              // 64 x SQ : L(kc,kprev) (or L(j,kprev)) operand image
    double* S = Bk + NB * SQ;               // 64 x SP : diagonal block image, later the panel block
    double* Lo = S + NB * SP;               // 64 x SP (+ pad rows read by the solve): factor
    double* misc = Lo + NB * SP;            // small scratch behind the images
    int* badflag = reinterpret_cast<int*>(misc);
    double* red = misc + 8;
    const int tid = threadIdx.x;
    const int kprev = pend ? kc - 1 : -1;
    const int npanel = T - kc;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((lda & 1) == 0);
    const double* Lk = A + (int64_t)(kc - 1) * NB;  // block column kc - 1 (unused unless pend)
    acc64_t acc;

    if ((int)blockIdx.x >= npanel) {
        // ---- update role: tile (i, j), kc < j <= min(i, jmax):  A(i,j) -= L(i,kc-1) L(j,kc-1)^T
        const int u = blockIdx.x - npanel;
        int ii, jj;
        if (jmax >= T - 1) {
            // the whole trailing triangle, row by row
            ii = (int)((sqrt(8.0 * u + 1.0) - 1.0) * 0.5);
            while ((ii + 1) * (ii + 2) / 2 <= u) ++ii;
            while (ii * (ii + 1) / 2 > u) --ii;
            jj = u - ii * (ii + 1) / 2;
        } else {
            // the (jmax - kc) block columns left in the outer panel: a rectangle less its upper corner
            const int J = jmax - kc;
            ii = u / J; jj = u - ii * J;
            if (jj > ii) return;
        }
        const int i = kc + 1 + ii, j = kc + 1 + jj;
        const int mi = (int)min((int64_t)NB, m - (int64_t)i * NB), mj = (int)min((int64_t)NB, m - (int64_t)j * NB);
        Blk64 bi, bj, bc;
        bi.load(Lk + (int64_t)i * NB * lda, lda, mi, NB, vec_ok);
        bj.load(Lk + (int64_t)j * NB * lda, lda, mj, NB, vec_ok);
        double* Cij = A + (int64_t)i * NB * lda + (int64_t)j * NB;
        bc.load(Cij, lda, mi, mj, vec_ok);
        bi.to_lds(Ak, SQ);
        bj.to_lds(Bk, SQ);
        bc.to_lds(S, SP);
        __syncthreads();
        mma64(acc, Ak, Bk);
        sub_acc64(S, acc, mi, mj, i == j);
        __syncthreads();
        bc.from_lds(S, SP);
        bc.store(Cij, lda, mi, mj, i == j);
        return;
    }

    // ---- panel role: block row i of block column kc
    const int i = kc + blockIdx.x;
    const bool diag = (i == kc);
    const int bs = (int)min((int64_t)NB, m - (int64_t)kc * NB);
    const int mi = (int)min((int64_t)NB, m - (int64_t)i * NB);
    double* Akk = A + (int64_t)kc * NB * lda + (int64_t)kc * NB;
    double* Pik = A + (int64_t)i * NB * lda + (int64_t)kc * NB;
    Blk64 bkk, bik, bd, bp;
    // every global load of this workgroup is issued before anything waits
    bd.load(Akk, lda, bs, bs, vec_ok);
    if (kprev >= 0) bkk.load(Lk + (int64_t)kc * NB * lda, lda, bs, NB, vec_ok);
    if (!diag) {
        bp.load(Pik, lda, mi, bs, vec_ok);
        if (kprev >= 0) bik.load(Lk + (int64_t)i * NB * lda, lda, mi, NB, vec_ok);
    }
    // diagonal image: lower part, identity padding
    {
        const int c = 2 * (tid & 31), r0 = tid >> 5;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int r = r0 + 8 * p;
            const bool in = r < bs;
            S[r * SP + c] = (in && c <= r) ? bd.v[p].x : ((r == c) ? 1.0 : 0.0);
            S[r * SP + c + 1] = (in && c + 1 <= r && c + 1 < bs) ? bd.v[p].y : ((r == c + 1) ? 1.0 : 0.0);
        }
    }
    if (tid == 0) *badflag = 0;
    acc64_t pacc;
    if (kprev >= 0) {
        bkk.to_lds(Bk, SQ);
        if (!diag) bik.to_lds(Ak, SQ);
        __syncthreads();
        if (!(dbg & 4)) {
            mma64(acc, Bk, Bk);                          // L(kc,kprev) L(kc,kprev)^T
            if (!diag) mma64(pacc, Ak, Bk);              // L(i,kprev) L(kc,kprev)^T
            sub_acc64(S, acc, bs, bs, true);
        }
    }
    __syncthreads();
    if (!(dbg & 1)) potrf64_blk(S, Lo, Ak, badflag, bs, dbg);            // ends with a barrier
    else { __syncthreads(); for (int e = tid; e < NB * SP; e += NTHREADS) Lo[e] = S[e]; __syncthreads(); }
    const bool bad = (*badflag != 0);
    if (diag) {
        // The factor's diagonal block goes to Ldiag, NOT over A(kc,kc): the other panel workgroups of this
        // launch read A(kc,kc) for their own (redundant) factorisation, and one that is dispatched late
        // must still find the unfactored block there.
        Blk64 bl;
        bl.from_lds(Lo, SP);
        bl.store(Ldiag + (int64_t)kc * NB * lda + (int64_t)kc * NB, lda, bs, bs, true);
        double lg = (tid < bs) ? log(Lo[tid * SP + tid]) : 0.0;
        for (int off = 32; off > 0; off >>= 1) lg += __shfl_down(lg, off);
        if ((tid & 63) == 0) red[tid >> 6] = lg;
        __syncthreads();
        if (tid == 0) {
            // consecutive launches add to one scalar from different XCDs: device-scope atomic, not a plain
            // read-modify-write through whichever L2 the workgroup sits behind
#if defined(ACCBPG_PLAIN_LOGDET) && ACCBPG_PLAIN_LOGDET
            *logdet += 2.0 * (red[0] + red[1] + red[2] + red[3]);     // (experiment build: see DESIGN.md section 4)
#else
            __hip_atomic_fetch_add(logdet, 2.0 * (red[0] + red[1] + red[2] + red[3]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
#endif
            if (bad) flags[FLAG_NOT_PD] = 1;
        }
        if (Winv != nullptr) {
            // inverse of the diagonal block for the gradient's W = L^-1, off the critical path (the other
            // panel workgroups are in their panel solve meanwhile): X L^T = I gives X = W^T
            __syncthreads();
            for (int e = tid; e < NB * NB; e += NTHREADS) S[(e >> 6) * SP + (e & 63)] = ((e >> 6) == (e & 63)) ? 1.0 : 0.0;
            __syncthreads();
            trsm64_blk(S, Lo, Ak);
            __syncthreads();
            double* Wkk = Winv + (int64_t)kc * NB * lda + (int64_t)kc * NB;
            for (int e = tid; e < NB * NB; e += NTHREADS) {
                const int r = e >> 6, c = e & 63;
                if (r < bs && c < bs) Wkk[(int64_t)r * lda + c] = (c <= r) ? S[c * SP + r] : 0.0;
            }
        }
        return;
    }
    // panel block: P = A(i,kc) - L(i,kprev) L(kc,kprev)^T, then X L^T = P
    double* Xs = S;                                      // the diagonal image is dead after the factorisation
    bp.to_lds(Xs, SP);
    __syncthreads();
    if (kprev >= 0 && !(dbg & 4)) sub_acc64(Xs, pacc, mi, bs, false);
    __syncthreads();
    if (!(dbg & 2)) trsm64_blk(Xs, Lo, Ak);
    __syncthreads();
    bp.from_lds(Xs, SP);
    bp.store(Pik, lda, mi, bs, false);
}

// acc += As * Bs^T (mma64 without the reset)
__device__ __forceinline__ void mma64_acc(acc64_t& acc, const double* __restrict__ As, const double* __restrict__ Bs) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lq = lane >> 4;
    const double* ap = As + (16 * wm + lr) * SQ + lq;
    const double* bp = Bs + (32 * wn + lr) * SQ + lq;
#pragma unroll 4
    for (int kk = 0; kk < NB / 4; ++kk) {
        const double a0 = ap[4 * kk], a1 = ap[32 * SQ + 4 * kk];
        const double b0 = bp[4 * kk], b1 = bp[16 * SQ + 4 * kk];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
}

// Two-level Cholesky for large m (where the rank-64 trailing update of chol_step_kernel is bound by the HBM
// traffic of the trailing matrix, read and written once per block column): the trailing matrix behind an
// outer panel of nk block columns k0 .. k0+nk-1 takes all their updates in one pass,
//   A(i,j) -= sum_k L(i,k) L(j,k)^T,   k0 + nk <= j <= i,
// one workgroup per 64x64 tile; the operand blocks of the next k are requested while the current pair is
// multiplied, the tile of A is requested first and consumed last.
__global__ __launch_bounds__(NTHREADS, 2) void chol_syrk_kernel(double* __restrict__ A, int64_t lda, int64_t m,
                                                               int k0, int nk, int T) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* Ak = lds;
    double* Bk = Ak + NB * SQ;
    double* S = Ak;                            // the tile image reuses the operand image after the last product:
                                               // 66 KB of LDS, two workgroups per CU
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((lda & 1) == 0);
    const int u = blockIdx.x;
    int ii = (int)((sqrt(8.0 * u + 1.0) - 1.0) * 0.5);
    while ((ii + 1) * (ii + 2) / 2 <= u) ++ii;
    while (ii * (ii + 1) / 2 > u) --ii;
    const int jj = u - ii * (ii + 1) / 2;
    const int i = k0 + nk + ii, j = k0 + nk + jj;
    if (i >= T) return;
    const int mi = (int)min((int64_t)NB, m - (int64_t)i * NB), mj = (int)min((int64_t)NB, m - (int64_t)j * NB);
    double* Cij = A + (int64_t)i * NB * lda + (int64_t)j * NB;
    const double* Li = A + (int64_t)i * NB * lda + (int64_t)k0 * NB;
    const double* Lj = A + (int64_t)j * NB * lda + (int64_t)k0 * NB;
    Blk64 bi, bj, bc;
    bc.load(Cij, lda, mi, mj, vec_ok);
    bi.load(Li, lda, mi, NB, vec_ok);
    bj.load(Lj, lda, mj, NB, vec_ok);
    acc64_t acc;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int k = 0; k < nk; ++k) {
        bi.to_lds(Ak, SQ);
        bj.to_lds(Bk, SQ);
        if (k + 1 < nk) {
            bi.load(Li + (int64_t)(k + 1) * NB, lda, mi, NB, vec_ok);
            bj.load(Lj + (int64_t)(k + 1) * NB, lda, mj, NB, vec_ok);
        }
        __syncthreads();
        mma64_acc(acc, Ak, Bk);
        __syncthreads();                       // the operand images are rewritten by the next k
    }
    bc.to_lds(S, SP);
    __syncthreads();
    sub_acc64(S, acc, mi, mj, i == j);
    __syncthreads();
    bc.from_lds(S, SP);
    bc.store(Cij, lda, mi, mj, i == j);
}

// =========================================================================================
// Cholesky in ONE launch for T <= 32 block columns (m <= 2048): every 64x64 tile of the lower triangle has an
// OWNER workgroup that keeps it in registers (MFMA accumulator layout) from the start of the launch until the
// tile is final, applies the rank-64 updates of the block columns to its left as their panels are published,
// then finishes the tile (panel solve, or factorisation for a diagonal tile) and publishes it for the tiles
// that need it.  The trailing matrix never goes back to memory between block columns, and nothing waits for a
// whole block column: a tile moves as soon as ITS operands exist.
//
//   owner of diagonal tile (d,d): also owns (d,d-1).  Critical chain per block column:
//        L(d-1,d-1) published -> [hand-off] -> solve (d,d-1) -> subtract its square from (d,d) -> factor (d,d)
//   every other tile (i,j), i > j+1, has a workgroup of its own.
//
// Same arithmetic in the same order as chol_step_kernel (every update is a 64-deep product formed from zero
// and subtracted; the same potrf64_blk / trsm64_blk), so the factor, the log-determinant and the diagonal-block
// inverses are BIT-IDENTICAL to the launch-per-block-column scheme (test_tile_cholesky_matches_step_kernels).
//
// Hand-off between workgroups inside the launch (MI355X_MICROARCH.md, visibility table, first row; Guideline
// 16 form with sc1 loads): the payload is stored write-through (agent-scope relaxed atomic stores = sc1), every
// storing wave drains its stores, the workgroup meets at a barrier and ONE lane sets the tile's flag (agent-scope
// store); a consumer polls that ONE word from one wave (relaxed agent-scope loads, s_sleep between polls), the
// workgroup meets at a barrier, then EVERY load of handed-off bytes is an agent-scope (sc1) load into registers.
// All workgroups must be resident at once (2 per CU: 76.8 KB of LDS each, <= 256 VGPRs): the host checks the
// grid against the occupancy the runtime reports, serialises these launches across streams of the process, and
// every spin is bounded: a wait that outlasts `spin_limit` ticks of the 100 MHz wall clock raises
// flags[FLAG_ABORT], every workgroup leaves at its next wait, and the host redoes the factorisation with the
// launch-per-column kernels (the source matrix is intact unless the factorisation ran in place, in which case
// the caller regenerates it).
// =========================================================================================
constexpr int CT_TMAX = 32;                       // block columns this scheme handles
constexpr int CT_AUX = 4 * POTRF_TB + 8;          // per block column: the four 16x16 factor copies + reciprocals, running log det, bad-pivot mark
constexpr int CT_LDS_DOUBLES = 2 * NB * SQ + 4 * POTRF_TB + 16;
constexpr int CT_LDS_BYTES = CT_LDS_DOUBLES * 8;  // 76.9 KB: two workgroups per CU

constexpr int CT_NSTAMP = 32;   // 8 stage stamps, then 4 per piece of the streamed panel solve (enter, landed, solved, folded)

// Every shared word is accessed as a GLOBAL agent-scope access (global_load / global_store ... sc1), never through
// a flat pointer: the pointers arrive inside a struct, which hides their address space from the compiler.
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) int gint;
__device__ __forceinline__ double ct_ld(const double* p) {
    return __hip_atomic_load((const gdouble*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ct_st(double* p, double v) {
    __hip_atomic_store((gdouble*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ct_ldi(const int* p) {
    return __hip_atomic_load((const gint*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ct_sti(int* p, int v) {
    __hip_atomic_store((gint*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one wave, uniformly: wait until *flag != 0; false on abort / timeout
__device__ __forceinline__ bool ct_spin(const int* flag, int* abortw, long long limit) {
    if (ct_ldi(flag) != 0) return true;
    const long long t0 = wall_clock64();
    unsigned n = 0;
    for (;;) {
        __builtin_amdgcn_s_sleep(1);
        if (ct_ldi(flag) != 0) return true;
        if ((++n & 15u) == 0u) {
            if (ct_ldi(abortw) != 0) return false;
            if (wall_clock64() - t0 > limit) {
                ct_sti(abortw, 1);
                return false;
            }
        }
    }
}
// whole workgroup: wave 0 polls the (one or two) flags, everybody learns the outcome at a barrier
__device__ __forceinline__ bool ct_wait(const int* f0, const int* f1, int* abortw, long long limit, int* word) {
    if (threadIdx.x < 64) {
        bool ok = ct_spin(f0, abortw, limit);
        if (ok && f1 != nullptr) ok = ct_spin(f1, abortw, limit);
        if (threadIdx.x == 0) *word = ok ? 1 : 0;
    }
    __syncthreads();
    return *word != 0;
}
// published 64x64 block (row-major, ld) -> LDS image with row stride ST; rows >= mr / columns >= mc read as zero;
// LOWER: entries above the diagonal read as zero too (a diagonal tile of the factor is stored lower-only)
template <int ST, bool LOWER>
__device__ __forceinline__ void ct_fetch(double* __restrict__ img, const double* G, int64_t ld, int mr, int mc) {
    const int tid = threadIdx.x;
    const int c = 2 * (tid & 31), r0 = tid >> 5;
    double v[16];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int r = r0 + 8 * p;
        const bool in0 = r < mr && c < mc && (!LOWER || c <= r);
        const bool in1 = r < mr && c + 1 < mc && (!LOWER || c + 1 <= r);
        v[2 * p] = in0 ? ct_ld(G + (int64_t)r * ld + c) : 0.0;
        v[2 * p + 1] = in1 ? ct_ld(G + (int64_t)r * ld + c + 1) : 0.0;
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        img[(r0 + 8 * p) * ST + c] = v[2 * p];
        img[(r0 + 8 * p) * ST + c + 1] = v[2 * p + 1];
    }
}
// LDS image (stride SP) -> global, write-through; the caller drains and flags
template <bool LOWER>
__device__ __forceinline__ void ct_publish(const double* __restrict__ img, double* G, int64_t ld, int mr, int mc) {
    const int tid = threadIdx.x;
    const int c = 2 * (tid & 31), r0 = tid >> 5;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int r = r0 + 8 * p;
        if (r < mr) {
            if (c < mc && (!LOWER || c <= r)) ct_st(G + (int64_t)r * ld + c, img[r * SP + c]);
            if (c + 1 < mc && (!LOWER || c + 1 <= r)) ct_st(G + (int64_t)r * ld + c + 1, img[r * SP + c + 1]);
        }
    }
}
// every storing wave has drained; one lane raises the flag
__device__ __forceinline__ void ct_signal(int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) ct_sti(flag, 1);
}
// tile of the source matrix -> accumulator layout (rows 16*(2i+wm)+lq+4r, columns 32*wn+16j+lr), zero padded
__device__ __forceinline__ void ct_load_acc(acc64_t& acc, const double* __restrict__ G, int64_t ld, int mr, int mc) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * (2 * i + wm) + lq + 4 * r, col = 32 * wn + 16 * j + lr;
                acc[i][j][r] = (row < mr && col < mc) ? G[(int64_t)row * ld + col] : 0.0;
            }
}
// accumulators -> LDS image (stride SP).  DIAG: lower part, zeros above, identity beyond bs (what potrf64_blk reads)
template <bool DIAG>
__device__ __forceinline__ void ct_acc_to_img(double* __restrict__ img, const acc64_t& acc, int bs) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * (2 * i + wm) + lq + 4 * r, col = 32 * wn + 16 * j + lr;
                double v = acc[i][j][r];
                if (DIAG) v = (row < bs && col < bs && col <= row) ? v : ((row == col) ? 1.0 : 0.0);
                img[row * SP + col] = v;
            }
}
// P = As * Bs^T for two 64x64 images with row strides SA, SB, then acc -= P (the product is formed from zero and
// subtracted, as chol_step_kernel does through sub_acc64)
template <int SA, int SB>
__device__ __forceinline__ void ct_update(acc64_t& acc, const double* __restrict__ As, const double* __restrict__ Bs) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lq = lane >> 4;
    const double* ap = As + (16 * wm + lr) * SA + lq;
    const double* bp = Bs + (32 * wn + lr) * SB + lq;
    d4 p00 = d4{0.0, 0.0, 0.0, 0.0}, p01 = p00, p10 = p00, p11 = p00;
#pragma unroll 4
    for (int kk = 0; kk < NB / 4; ++kk) {
        const double a0 = ap[4 * kk], a1 = ap[32 * SA + 4 * kk];
        const double b0 = bp[4 * kk], b1 = bp[16 * SB + 4 * kk];
        p00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, p00, 0, 0, 0);
        p01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, p01, 0, 0, 0);
        p10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, p10, 0, 0, 0);
        p11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, p11, 0, 0, 0);
    }
    acc[0][0] -= p00; acc[0][1] -= p01; acc[1][0] -= p10; acc[1][1] -= p11;
}

__global__ __launch_bounds__(NTHREADS, 2) void chol_tiles_kernel(CholInst one, const CholInst* __restrict__ table,
                                                                const CholJob* __restrict__ jobs, int64_t ld, int64_t m,
                                                                int T, long long spin_limit, int stall_test, BatchAct act) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* R0 = lds;                           // operand image (stride SQ) / tile image (stride SP)
    double* R1 = R0 + NB * SQ;                  // operand image / factor image
    double* tbuf = R1 + NB * SQ;                // 4 * POTRF_TB: 16x16 factor copies + reciprocals of a block column
    int* word = reinterpret_cast<int*>(tbuf + 4 * POTRF_TB);      // outcome of the last wait
    int* badflag = word + 2;
    double* red = tbuf + 4 * POTRF_TB + 8;

    const CholInst ci = (table != nullptr) ? table[act.idx[blockIdx.y]] : one;
    const CholJob job = jobs[blockIdx.x];
    const int tid = threadIdx.x;
    const bool diagrole = (job.i == job.j);
    const int d = job.i;
    const bool has_off = !diagrole || d > 0;
    const int ti = job.i, tj = diagrole ? d - 1 : job.j;          // the off-diagonal tile of this workgroup
    int* abortw = ci.flags + FLAG_ABORT;
    int* pflags = ci.ready + T * T;                               // 4 per block column: the 16-column pieces of its factor
    int* hflags = pflags + 4 * T;                                 // 1 per block row: its diagonal tile, left updates applied
    auto blk = [&](int b) { return (int)min((int64_t)NB, m - (int64_t)b * NB); };
    auto at = [&](auto* base, int bi, int bj) { return base + (int64_t)bi * NB * ld + (int64_t)bj * NB; };
    if (stall_test && blockIdx.x == 0) return;                    // test hook: block column 0 is never published
    auto stamp = [&](int slot) {
        if (ci.trace != nullptr && diagrole && tid == 0) ci.trace[d * CT_NSTAMP + slot] = wall_clock64();
    };
    stamp(0);                                                     // workgroup started

    // The diagonal tile (d,d) is brought up to date through block column d-2 by ANOTHER workgroup -- the owner of
    // tile (d,0), which is done with its own tile after the first block column -- and handed over through memory
    // (hand[d], accumulator layout); the chain workgroup adds only the last update, its own panel block times
    // itself.  This keeps the chain workgroup's way from "L(d-1,d-2) is there" to its panel solve short.
    acc64_t accO, accD;
    const int mi = blk(ti);
    if (has_off) ct_load_acc(accO, at(ci.src, ti, tj), ld, mi, blk(tj));
    if (diagrole && d < 2) ct_load_acc(accD, at(ci.src, d, d), ld, blk(d), blk(d));

    double run_logdet = 0.0, run_bad = 0.0;
    if (has_off) {
        // ---- rank-64 updates from the block columns left of the tile
        for (int k = 0; k < tj; ++k) {
            // (the block of our own row comes from a workgroup with no diagonal tile to look after, usually the
            //  earlier of the two: it is on its way into LDS while the other one is still awaited)
            if (!ct_wait(ci.ready + ti * T + k, nullptr, abortw, spin_limit, word)) return;
            ct_fetch<SQ, false>(R0, at(ci.L, ti, k), ld, mi, NB);
            if (!ct_wait(ci.ready + tj * T + k, nullptr, abortw, spin_limit, word)) return;
            ct_fetch<SQ, false>(R1, at(ci.L, tj, k), ld, blk(tj), NB);
            __syncthreads();
            ct_update<SQ, SQ>(accO, R0, R1);
        }
        // ---- panel solve against the factor of block column tj:  X L(tj,tj)^T = tile
        // (the tile is staged while the factor is still on its way: R0 is free once every wave is past its products)
        __syncthreads();
        ct_acc_to_img<false>(R0, accO, NB);
        stamp(1);                                                 // updates from the columns further left are in
        // the factor of block column tj comes in four pieces of 16 columns, each flagged as its owner finishes it
        const double* ax = ci.aux + (int64_t)tj * CT_AUX;
        const double* Ljj = at(ci.Ldiag, tj, tj);
        int* pflag = pflags + 4 * tj;
        // one piece in registers: its 16x16 factor copy + reciprocals (272 doubles) and the rows of the factor below
        // it (at most 48 x 16); a wave that found the piece flagged ahead of time already holds its share
        double pc_t[2], pc_l[3], pc_s[2];
        bool have = false;
        int looked = 0;
        auto load_piece = [&](int pnl) {
            const int k0 = 16 * pnl;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = tid + q * NTHREADS;
                pc_t[q] = (e < POTRF_TB) ? ct_ld(ax + pnl * POTRF_TB + e) : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int e = tid + q * NTHREADS;
                pc_l[q] = (e < (NB - k0 - 16) * 16) ? ct_ld(Ljj + (int64_t)(k0 + 16 + (e >> 4)) * ld + k0 + (e & 15)) : 0.0;
            }
            if (pnl == 3) {
                pc_s[0] = ct_ld(ax + 4 * POTRF_TB);
                pc_s[1] = ct_ld(ax + 4 * POTRF_TB + 1);
            }
        };
        auto land_piece = [&](int pnl) {
            const int k0 = 16 * pnl;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = tid + q * NTHREADS;
                if (e < POTRF_TB) tbuf[pnl * POTRF_TB + e] = pc_t[q];
            }
            // rows 16(pnl+1) .. 63 of the factor, columns k0 .. k0+15 (what the update of the columns to the right reads)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int e = tid + q * NTHREADS;
                if (e < (NB - k0 - 16) * 16) R1[(k0 + 16 + (e >> 4)) * SP + k0 + (e & 15)] = pc_l[q];
            }
            if (pnl == 3) { run_logdet = pc_s[0]; run_bad = pc_s[1]; }
        };
        auto fetch = [&](int pnl, int mode) -> bool {
            if (mode == 1) { looked = ct_ldi(pflag + pnl); return true; }            // (uniform within a wave)
            if (mode == 2) { if (looked != 0) { load_piece(pnl); have = true; } return true; }
            // (a piece that is already in registers goes to LDS in front of the barrier: nobody reads those
            //  columns of the images before it)
            stamp(8 + 4 * pnl + (have ? 0 : 0));
            if (have) land_piece(pnl);
            if (__syncthreads_or(have ? 0 : 1)) {
                if (!ct_wait(pflag + pnl, nullptr, abortw, spin_limit, word)) return false;
                if (!have) { load_piece(pnl); land_piece(pnl); }
                __syncthreads();
            }
            have = false;
            stamp(9 + 4 * pnl);
            if (pnl == 0) stamp(2);                               // the first piece of the previous block column is here
            if (pnl == 3) stamp(3);                               // ... and the last
            return true;
        };
        // As soon as 16 columns of the solved tile are final they are published, and the owner of a diagonal tile
        // folds them into the product that is the LAST update of that tile (its own panel block times itself: the
        // same 64-deep MFMA chain as ct_update, formed from zero in k order and subtracted once, 16 of its 64
        // steps per panel) -- both while the next piece of the factor is still on its way.
        double* Lij = at(ci.L, ti, tj);
        const int mj = blk(tj);
        d4 p00 = d4{0.0, 0.0, 0.0, 0.0}, p01 = p00, p10 = p00, p11 = p00;
        auto done = [&](int pnl) -> bool {
            const int k0 = 16 * pnl;
            stamp(10 + 4 * pnl);
            for (int e = tid; e < NB * 16; e += NTHREADS) {
                const int r = e >> 4, c = k0 + (e & 15);
                if (r < mi && c < mj) ct_st(Lij + (int64_t)r * ld + c, R0[r * SP + c]);
            }
            if (diagrole) {
                const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
                const int wm = wave >> 1, wn = wave & 1;
                const double* ap = R0 + (16 * wm + (lane & 15)) * SP + (lane >> 4);
                const double* bp = R0 + (32 * wn + (lane & 15)) * SP + (lane >> 4);
#pragma unroll
                for (int kk = 4 * pnl; kk < 4 * pnl + 4; ++kk) {
                    const double a0 = ap[4 * kk], a1 = ap[32 * SP + 4 * kk];
                    const double b0 = bp[4 * kk], b1 = bp[16 * SP + 4 * kk];
                    p00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, p00, 0, 0, 0);
                    p01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, p01, 0, 0, 0);
                    p10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, p10, 0, 0, 0);
                    p11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, p11, 0, 0, 0);
                }
                if (pnl == 2 && d >= 2) {
                    // the diagonal tile as its accumulator left it (long there by now): requested here, it lands
                    // while the last piece of the factor is awaited
                    if (!ct_wait(hflags + d, nullptr, abortw, spin_limit, word)) return false;
                    const double* hd = ci.hand + (int64_t)d * NB * NB;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) accD[i][j][r] = ct_ld(hd + ((i * 2 + j) * 4 + r) * NTHREADS + tid);
                }
            }
            stamp(11 + 4 * pnl);
            if (pnl == 3) stamp(4);                               // panel solve done, last columns on their way
            return true;
        };
        if (!trsm64_stream(R0, R1, tbuf, fetch, done)) return;
        if (!diagrole) {
            ct_signal(ci.ready + ti * T + tj);
            if (tj != 0) return;
            // ---- second life of the owner of tile (i,0): accumulate the diagonal tile (i,i) of its block row
            // through block column i-2 and hand it to the chain workgroup of that row
            acc64_t accH;
            ct_load_acc(accH, at(ci.src, ti, ti), ld, mi, mi);
            for (int k = 0; k + 1 < ti; ++k) {
                if (!ct_wait(ci.ready + ti * T + k, nullptr, abortw, spin_limit, word)) return;
                ct_fetch<SQ, false>(R0, at(ci.L, ti, k), ld, mi, NB);
                __syncthreads();
                ct_update<SQ, SQ>(accH, R0, R0);
            }
            double* hd = ci.hand + (int64_t)ti * NB * NB;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ct_st(hd + ((i * 2 + j) * 4 + r) * NTHREADS + tid, accH[i][j][r]);
            ct_signal(hflags + ti);
            return;
        }
        accD[0][0] -= p00; accD[0][1] -= p01; accD[1][0] -= p10; accD[1][1] -= p11;
    }
    // ---- diagonal tile: factor, publish, log-determinant, inverse
    const int bs = blk(d);
    ct_acc_to_img<true>(R1, accD, bs);
    if (tid == 0) *badflag = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the panel block's stores have drained on this wave ...
    __syncthreads();
    if (has_off && tid == 0) ct_sti(ci.ready + ti * T + tj, 1);   // ... and on every wave: flag it
    stamp(5);                                                     // diagonal tile up to date and staged
    // The factor is published in four pieces of 16 columns as potrf64_blk finishes them (rows k0.. of the factor,
    // lower part, plus that panel's 16x16 copy and reciprocals): piece pnl-1 goes out from the wavefronts that have
    // nothing to do while wavefront 0 factors the next 16x16 block, and is flagged when that block is done -- by
    // then its write-through stores have drained -- so the consumers' panel solves run while the panels to the
    // right are still being factored.
    double* axd = ci.aux + (int64_t)d * CT_AUX;
    double* Ldd = at(ci.Ldiag, d, d);
    int* pflag_d = pflags + 4 * d;
    auto publish_piece = [&](int pnl, int first, int nthr) {
        const int k0 = 16 * pnl;
        for (int e = tid - first; e < (NB - k0) * 16; e += nthr) {
            const int r = k0 + (e >> 4), c = k0 + (e & 15);
            if (r < bs && c <= r) ct_st(Ldd + (int64_t)r * ld + c, R0[r * SP + c]);
        }
        for (int e = tid - first; e < POTRF_TB; e += nthr) ct_st(axd + pnl * POTRF_TB + e, tbuf[pnl * POTRF_TB + e]);
    };
    auto hook = [&](int pnl, int where) {
        if (where == 3 && pnl > 0) publish_piece(pnl - 1, 64, NTHREADS - 64);
        else if (where == 0 && pnl > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (where == 1 && pnl > 0 && tid == 0) ct_sti(pflag_d + pnl - 1, 1);          // every wave has drained: flag it
    };
    potrf64_blk(R1, R0, tbuf, badflag, bs, 0, hook);              // factor in R0; ends with a barrier
    stamp(6);                                                     // factored
    double lg = (tid < bs) ? log(R0[tid * SP + tid]) : 0.0;
    for (int off = 32; off > 0; off >>= 1) lg += __shfl_down(lg, off);
    if ((tid & 63) == 0) red[tid >> 6] = lg;
    publish_piece(3, 0, NTHREADS);
    __syncthreads();
    const double logdet = run_logdet + 2.0 * (red[0] + red[1] + red[2] + red[3]);     // block columns in order
    const double bad = (*badflag != 0 || run_bad != 0.0) ? 1.0 : 0.0;
    if (tid == 0) { ct_st(axd + 4 * POTRF_TB, logdet); ct_st(axd + 4 * POTRF_TB + 1, bad); }
    ct_signal(pflag_d + 3);
    stamp(7);                                                     // factor published
    if (d == T - 1 && tid == 0) {
        *ci.logdet = logdet;
        if (bad != 0.0) ci.flags[FLAG_NOT_PD] = 1;
    }
    if (ci.Winv != nullptr) {
        // inverse of the diagonal block for W = L^-1 (off every chain): X L^T = I gives X = W^T
        for (int e = tid; e < NB * NB; e += NTHREADS) R1[(e >> 6) * SP + (e & 63)] = ((e >> 6) == (e & 63)) ? 1.0 : 0.0;
        __syncthreads();
        trsm64_blk(R1, R0, tbuf);
        __syncthreads();
        double* Wkk = at(ci.Winv, d, d);
        for (int e = tid; e < NB * NB; e += NTHREADS) {
            const int r = e >> 6, c = e & 63;
            if (r < bs && c < bs) Wkk[(int64_t)r * ld + c] = (c <= r) ? R1[c * SP + r] : 0.0;
        }
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

// any entry of x that is not >= 0 (NaN counts)?  One workgroup, eight independent loads in flight per thread: the
// check sits between the Gram kernel and the factorisation on every evaluation, so its latency is on the path.
__device__ __forceinline__ bool xcheck_bad(const double* __restrict__ x, int64_t n) {
    bool bad = false;
    const int64_t step = (int64_t)blockDim.x * 8;
    int64_t i = threadIdx.x;
    for (; i + 7 * (int64_t)blockDim.x < n; i += step) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x[i + u * (int64_t)blockDim.x];
#pragma unroll
        for (int u = 0; u < 8; ++u) bad |= !(v[u] >= 0.0);
    }
    for (; i < n; i += blockDim.x) bad |= !(x[i] >= 0.0);
    return __syncthreads_or(bad ? 1 : 0) != 0;
}

// resets the scalars and the status flags; with x != NULL also the x >= 0 check of functions.py:45
__global__ __launch_bounds__(1024) void zero_scalars_kernel(double* dscal, int* dflag, const double* __restrict__ x,
                                                          int64_t n, int* __restrict__ ready, int nready) {
#if defined(ACCBPG_PLAIN_LOGDET) && ACCBPG_PLAIN_LOGDET
    if (threadIdx.x < 8) dscal[threadIdx.x] = 0.0;
#else
    if (threadIdx.x < 8) __hip_atomic_store(dscal + threadIdx.x, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    if (threadIdx.x < 8) dflag[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < nready; i += blockDim.x) ready[i] = 0;     // hand-off flags of the one-launch Cholesky
    if (x == nullptr) return;
    if (xcheck_bad(x, n) && threadIdx.x == 0) dflag[FLAG_NEG_X] = 1;
}

// the same reset for the active instances of a batch (one workgroup per instance)
__global__ __launch_bounds__(1024) void zero_scalars_batch_kernel(const BatchInst* __restrict__ bt, BatchAct act,
                                                                const double* __restrict__ xbase, int64_t ldx, int64_t n,
                                                                int nready) {
    const int inst = act.idx[blockIdx.x];
    const BatchInst bi = bt[inst];
    if (threadIdx.x < 8) __hip_atomic_store(bi.dscal + threadIdx.x, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < 8) bi.dflag[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < nready; i += blockDim.x) bi.chol_ready[i] = 0;
    if (xbase == nullptr) return;
    if (xcheck_bad(xbase + (int64_t)inst * ldx, n) && threadIdx.x == 0) bi.dflag[FLAG_NEG_X] = 1;
}

__global__ void set_op_kernel(GemmOp* slot, GemmOp op) { *slot = op; }

// fp64 MFMA peak: every wave issues `iters` x 8 independent-accumulator MFMAs (inline asm keeps the
// accumulators where they are; the builtin form made hipcc shuttle them through the AGPR file).
__global__ __launch_bounds__(NTHREADS, 1) void mfma_peak_kernel(int iters, double* sink) {
    d4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456) sink[0] = s;
}

// Development probe: do the fp64 matrix pipe and the fp64 vector pipe run at the same time?  Per loop trip every wave
// issues `nm` independent MFMAs and `nv` independent v_fma_f64 (mode bit 0: MFMAs on, bit 1: vector FMAs on).
__global__ __launch_bounds__(NTHREADS, 1) void pipe_probe_kernel(int iters, int mode, double* sink) {
    d4 acc[8];
    double va[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 16; ++i) va[i] = threadIdx.x * 1e-3 + i;
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
        if (mode & 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
        if (mode & 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(va[i]) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += va[i];
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

// The kernels with variants (schedules, ablations) raise their limit at the launch site, the first time the instantiation
// is launched: on this runtime a launch that asks for more dynamic LDS than the kernel's limit reports NO error -- not from
// the launch, not from hipGetLastError, not from a synchronize -- and leaves garbage (measured; a Gram variant that had
// been left out of a list of set_lds calls went unnoticed until its results were compared).
static int ensure_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::unordered_map<const void*, int> limit;
    std::lock_guard<std::mutex> guard(mu);
    auto it = limit.find(kernel);
    if (it != limit.end() && it->second >= bytes) return ACCBPG_OK;
    ACC_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    limit[kernel] = bytes;
    return ACCBPG_OK;
}
// (a kernel name with template commas goes in parentheses)
#define ACC_LAUNCH_LDS(kernel, grid, block, lds, stream, ...)                        \
    do {                                                                             \
        auto k__ = kernel;                                                           \
        ACC_TRY(ensure_lds(reinterpret_cast<const void*>(k__), (int)(lds)));         \
        k__<<<(grid), (block), (lds), (stream)>>>(__VA_ARGS__);                      \
    } while (0)

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

// development switch read when a handle is created: bit 0 = plain stream-K ranges also where there are more tiles than
// workgroups (A/B of the whole-tile partition below)
static int g_plan_flags = 0;
void set_plan_flags(int flags) { g_plan_flags = flags; }

int build_plans(accbpg_dopt* h) {
    const int64_t m = h->m;
    // ---- Gram tile list (lower tiles) and stream-K partition
    const bool interior256 = h->vec_ok && (m % 256 == 0) && (h->n % BK == 0);
    const int BM = h->big ? TileBig<false>::BM : TileSmall<false>::BM;
    const int BN = h->big ? TileBig<false>::BN : TileSmall<false>::BN;
    std::vector<TileRC> tl;
    const int nrb = (int)((m + BM - 1) / BM), ncb = (int)((m + BN - 1) / BN);
    // Long rows: the Gram matrix is formed column block by column block of V (one launch each, added up in G).  Over a
    // pass of 16384 k-steps the workgroups of a launch drift apart and stop sharing their panels of V through L2: at
    // (8192,262144) one launch reached 0.68 of the MFMA peak, the (8192,32768) shape 0.86 -- so rows of 65536 columns or
    // more are cut into blocks of 32768 (which is also what the eight ranks of BASELINE config 5 hold each).
    h->gram_chunks = 1;
    h->gram_nc = h->n;
    constexpr int64_t GRAM_NC = 32768;
    if (!(g_plan_flags & 4) && h->big && interior256 && h->use_glds && h->n >= 2 * GRAM_NC && h->n % GRAM_NC == 0) {
        h->gram_chunks = (int)(h->n / GRAM_NC);
        h->gram_nc = GRAM_NC;
        // Rows a megabyte or more apart: a tile of the Gram kernel streams 384 rows of V at once, and with that stride
        // every one of them sits in a page of its own -- measured on one (8192, 32768) column block: 32.5 ms with rows
        // 256 or 512 KiB apart, 34.6 ms at 1 MiB, 40.4 ms at 2 MiB (`tools/gram_stride.py`,
        // profiles/r03_gram_stride.json).  The handle then keeps a copy of V stored block by block ([block][row][32768
        // columns], made once, here) and the Gram launches read that; everything else keeps reading the caller's V,
        // which must not change while the handle lives.
        if (!(g_plan_flags & 8) && h->ldv * (int64_t)sizeof(double) >= (1 << 20)) {
            const size_t bytes = sizeof(double) * (size_t)m * (size_t)h->n;
            if (hipMalloc(&h->Vblk, bytes) == hipSuccess) {
                for (int c = 0; c < h->gram_chunks; ++c)
                    ACC_HIP(hipMemcpy2DAsync(h->Vblk + (size_t)c * m * GRAM_NC, sizeof(double) * GRAM_NC,
                                             h->V + (size_t)c * GRAM_NC, sizeof(double) * (size_t)h->ldv,
                                             sizeof(double) * GRAM_NC, (size_t)m, hipMemcpyDeviceToDevice, h->stream));
                ACC_HIP(hipStreamSynchronize(h->stream));
            } else {
                (void)hipGetLastError();                        // no room for the copy: read V where it lies
                h->Vblk = nullptr;
            }
        }
    }
    h->kiters = (h->gram_nc + BK - 1) / BK;
    // 256x128 tiles: the tile (rb, 2rb+1) next to the diagonal holds one useful 128 x 128 block, the odd
    // diagonal block 2rb+1, under 128 rows that lie strictly above the diagonal.  On the direct-to-LDS
    // path those blocks run as dual tiles instead (half of K each at full MFMA work), two per entry.
    const bool duals = h->big && interior256 && h->use_glds && (h->kiters % 2 == 0) && BM == 2 * BN;
    std::vector<int> lone;
    for (int rb = 0; rb < nrb; ++rb)
        for (int cb = 0; cb < ncb; ++cb)
            if ((int64_t)cb * BN <= (int64_t)rb * BM + BM - 1) {
                if (duals && cb == 2 * rb + 1) lone.push_back(cb);
                else tl.push_back(TileRC{rb, cb});
            }
    for (size_t i = 0; i + 1 < lone.size(); i += 2)
        tl.push_back(TileRC{lone[i] / 2, lone[i], lone[i], lone[i + 1]});
    if (lone.size() & 1) tl.push_back(TileRC{lone.back() / 2, lone.back()});   // odd one out: ordinary tile
    h->has_duals = lone.size() >= 2;
    h->ntiles = (int)tl.size();
    const int64_t total = (int64_t)h->ntiles * h->kiters;
    int grid = h->big ? h->num_cu : 2 * h->num_cu;
    if (h->gram_grid_cap > 0 && grid > h->gram_grid_cap) grid = h->gram_grid_cap;   // one instance of a batch: its share of the chip
    if (grid > total) grid = (int)total;
    if (grid < h->ntiles && total / h->ntiles < 8) grid = h->ntiles;   // tiny K: one tile per workgroup
    int64_t per = (total + grid - 1) / grid;
    grid = (int)((total + per - 1) / per);
    h->gram_grid = grid;
    h->gram_per = (int)per;
    const int64_t kit = h->kiters;
    // ---- unit ranges per workgroup.
    // Plain stream-K hands workgroup w the contiguous range [w*per, (w+1)*per).  When a tile holds several
    // whole ranges (q = kiters / per >= 1, remainder rem > 0) the ranges are instead ALIGNED to the
    // tiles: every tile is cut at 0, per, .., q*per, so that all workgroups of "class" j start at k-step
    // j*per of their tile and stream the same columns of V at the same time; the q*ntiles body pieces
    // are dealt to the XCDs (workgroup id mod 8, round-robin placement -- speed only, never correctness)
    // 32 at a time in an order in which 32 consecutive tiles form a compact block (few distinct row and
    // column panels -> shared through that XCD's L2), and the remainders [q*per, kiters) are walked by
    // the last workgroups as one contiguous stream.
    std::vector<int64_t> ranges((size_t)grid * GRAM_RMAX * 2, 0);
    const int64_t q = kit / per, rem = kit % per;
    // (a tail workgroup crosses at most per/rem + 2 remainders)
    const bool aligned = h->big && q >= 1 && rem > 0 && (int64_t)h->ntiles * q < grid &&
                         per / rem + 2 <= GRAM_RMAX;
    if (aligned) {
        // compact order of the entries: 4 x 8 blocks of tiles
        std::stable_sort(tl.begin(), tl.end(), [](const TileRC& a, const TileRC& b) {
            const int ka[4] = {a.rb / 4, a.cb / 8, a.rb, a.cb}, kb[4] = {b.rb / 4, b.cb / 8, b.rb, b.cb};
            for (int i = 0; i < 4; ++i)
                if (ka[i] != kb[i]) return ka[i] < kb[i];
            return false;
        });
        const int nbody = (int)(h->ntiles * q);
        const int px = (grid % 8 == 0) ? grid / 8 : 0;          // workgroups per XCD
        auto wg_of = [&](int i) { return px ? (i / px) + 8 * (i % px) : i; };
        int idx = 0;
        for (int64_t j = 0; j < q; ++j)
            for (int e = 0; e < h->ntiles; ++e, ++idx) {
                const int w = wg_of(idx);
                ranges[((size_t)w * GRAM_RMAX) * 2] = (int64_t)e * kit + j * per;
                ranges[((size_t)w * GRAM_RMAX) * 2 + 1] = (int64_t)e * kit + (j + 1) * per;
            }
        // remainders: tail-workgroup t covers the concatenated tails' units [t*per, (t+1)*per)
        const int64_t tail_total = (int64_t)h->ntiles * rem;
        for (int t = 0; nbody + t < grid; ++t) {
            const int w = wg_of(nbody + t);
            int64_t u = (int64_t)t * per;
            const int64_t u1 = std::min(u + per, tail_total);
            int r = 0;
            while (u < u1) {
                const int64_t e = u / rem, off = u - e * rem;
                const int64_t len = std::min(rem - off, u1 - u);
                if (r >= GRAM_RMAX) return ACCBPG_ERR_ARG;      // cannot happen: per / rem + 2 <= GRAM_RMAX
                ranges[((size_t)w * GRAM_RMAX + r) * 2] = e * kit + q * per + off;
                ranges[((size_t)w * GRAM_RMAX + r) * 2 + 1] = e * kit + q * per + off + len;
                ++r;
                u += len;
            }
        }
    }
    // More tiles than workgroups (m >= 4096 on 256 CUs; BASELINE config 5: 1056 entries): plain stream-K would hand
    // every workgroup a contiguous range of 4.1 tiles that starts somewhere inside a tile, so no two workgroups ever
    // stream the same columns of V at the same time and every k-step of every workgroup comes from HBM (measured at
    // (8192,262144): 0.67 of the MFMA peak, against 0.84 at (8192,32768) where the Infinity Cache still covers the
    // offsets).  Instead every workgroup gets `wt` WHOLE tiles, which it walks from k = 0 in step with all the others,
    // and the workgroups of one XCD (id mod 8 under round-robin placement -- speed only, never correctness) hold, at
    // every one of the wt phases, a compact 4 x 8 block of tiles: 4 row panels + 8 column panels feed 32 tiles through
    // that XCD's L2.  The ntiles - wt*grid entries left over are cut into equal pieces, one per workgroup (the
    // stream-K tail: a few percent of the work).
    const bool whole_tiles = !(g_plan_flags & 1) && !aligned && h->big && per >= kit && h->ntiles >= grid && grid % 8 == 0 && grid >= 8;
    if (whole_tiles) {
        std::stable_sort(tl.begin(), tl.end(), [](const TileRC& a, const TileRC& b) {
            const int ka[4] = {a.rb / 4, a.cb / 8, a.rb, a.cb}, kb[4] = {b.rb / 4, b.cb / 8, b.rb, b.cb};
            for (int i = 0; i < 4; ++i)
                if (ka[i] != kb[i]) return ka[i] < kb[i];
            return false;
        });
        const int wt = h->ntiles / grid, B = grid / 8;
        const int nbody = wt * grid;
        std::vector<TileRC> perm(tl);
        for (int pph = 0; pph < wt; ++pph)
            for (int xcd = 0; xcd < 8; ++xcd)
                for (int sl = 0; sl < B; ++sl) {
                    const int w = 8 * sl + xcd;
                    perm[(size_t)wt * w + pph] = tl[((size_t)pph * 8 + xcd) * B + sl];
                }
        tl.swap(perm);
        const int64_t tail_total = (int64_t)(h->ntiles - nbody) * kit;
        const int64_t tail_per = (tail_total + grid - 1) / grid;
        for (int w = 0; w < grid; ++w) {
            ranges[((size_t)w * GRAM_RMAX) * 2] = (int64_t)wt * w * kit;
            ranges[((size_t)w * GRAM_RMAX) * 2 + 1] = (int64_t)wt * (w + 1) * kit;
            const int64_t t0 = std::min((int64_t)w * tail_per, tail_total), t1 = std::min((int64_t)(w + 1) * tail_per, tail_total);
            ranges[((size_t)w * GRAM_RMAX + 1) * 2] = (int64_t)nbody * kit + t0;
            ranges[((size_t)w * GRAM_RMAX + 1) * 2 + 1] = (int64_t)nbody * kit + t1;
        }
    }
    if (!aligned && !whole_tiles) {
        for (int w = 0; w < grid; ++w) {
            ranges[((size_t)w * GRAM_RMAX) * 2] = std::min((int64_t)w * per, total);
            ranges[((size_t)w * GRAM_RMAX) * 2 + 1] = std::min((int64_t)(w + 1) * per, total);
        }
    }
    // XCD-aware order.  Workgroups whose ranges start a whole number of tiles apart run the same
    // k-step at the same time; with `per` steps per workgroup those are workgroups dw = kiters/g apart
    // (g = gcd(per, kiters)), D = per/g tiles apart, and when dw is a multiple of 8 they share an XCD
    // (round-robin placement: speed only, never correctness).  Arrange the list so that positions
    // i, i+D, i+2D, ... hold a compact 2 x 4 block of tiles (2 A panels + 4 B panels feed 8 tiles),
    // which those workgroups then stream through the same L2 together.
    {
        auto gcd64 = [](int64_t a, int64_t b) { while (b) { int64_t t = a % b; a = b; b = t; } return a; };
        const int64_t g = gcd64(per, h->kiters);
        const int64_t D = per / g, dw = h->kiters / g;
        const int nt = (int)tl.size();
        if (!aligned && !whole_tiles && h->big && D > 1 && D < nt && nt % D == 0 && dw % 8 == 0) {
            const int gs = (int)(nt / D);
            // sequence in which consecutive runs are compact: row-block pairs, then 4 column blocks at a time
            std::vector<TileRC> seq;
            std::vector<char> used(tl.size(), 0);
            auto find = [&](int rb, int cb) {
                for (size_t i = 0; i < tl.size(); ++i)
                    if (!used[i] && tl[i].rb == rb && tl[i].cb == cb) return (int)i;
                return -1;
            };
            const int bcols = (gs >= 2 && gs % 2 == 0) ? gs / 2 : 4;      // 2 x (gs/2) blocks fill one group
            for (int rp = nrb - 2 + (nrb & 1); rp >= -1; rp -= 2)
                for (int c4 = 0; c4 < ncb; c4 += bcols)
                    for (int dr = 0; dr < 2; ++dr)
                        for (int dc = 0; dc < bcols; ++dc) {
                            const int rb = rp + dr, cb = c4 + dc;
                            if (rb < 0 || rb >= nrb || cb >= ncb) continue;
                            // only full 2x4 blocks first; ragged remainders are appended below
                            const int i = find(rb, cb);
                            if (i >= 0 && find(rp + (1 - dr), cb) != -2) { used[i] = 1; seq.push_back(tl[i]); }
                        }
            for (size_t i = 0; i < tl.size(); ++i)
                if (!used[i]) seq.push_back(tl[i]);
            std::vector<TileRC> perm(tl.size());
            for (int i = 0; i < (int)D; ++i)
                for (int j = 0; j < gs; ++j) perm[(size_t)D * j + i] = seq[(size_t)i * gs + j];
            tl.swap(perm);
        }
    }
    ACC_HIP(hipMalloc(&h->tiles, sizeof(TileRC) * tl.size()));
    ACC_HIP(hipMemcpy(h->tiles, tl.data(), sizeof(TileRC) * tl.size(), hipMemcpyHostToDevice));
    // ---- replay every workgroup's walk (the device splits a range at tile and dual-half boundaries the
    // same way): slab slot of each segment = its ordinal in the walk; contributors of each tile in k order
    {
        struct Contrib { int64_t kb; int w, slot; };
        std::vector<std::vector<Contrib>> cl((size_t)h->ntiles * 2);
        const int64_t half = kit / 2;
        int nslot = 1;
        for (int w = 0; w < grid; ++w) {
            int seg = 0;
            for (int r = 0; r < GRAM_RMAX; ++r) {
                int64_t it = ranges[((size_t)w * GRAM_RMAX + r) * 2];
                const int64_t it1 = ranges[((size_t)w * GRAM_RMAX + r) * 2 + 1];
                while (it < it1) {
                    const int64_t e = it / kit, off = it - e * kit;
                    const bool dual = tl[(size_t)e].d1 >= 0;
                    const int hs = (dual && off >= half) ? 1 : 0;
                    const int64_t lo = dual ? (hs ? half : 0) : 0, hi = dual ? (hs ? kit : half) : kit;
                    const int64_t ue = std::min(hi, off + (it1 - it));
                    const bool whole = (off == lo && ue == hi);
                    if (dual || !whole || h->gram_chunks > 1) cl[(size_t)e * 2 + hs].push_back(Contrib{off, w, seg});
                    it += ue - off;
                    ++seg;
                }
            }
            nslot = std::max(nslot, seg);
        }
        h->gram_nslot = nslot;
        std::vector<int32_t> cstart(cl.size() + 1, 0), contrib;
        for (size_t i = 0; i < cl.size(); ++i) {
            std::sort(cl[i].begin(), cl[i].end(), [](const Contrib& a, const Contrib& b) { return a.kb < b.kb; });
            for (const Contrib& c : cl[i]) { contrib.push_back(c.w); contrib.push_back(c.slot); }
            cstart[i + 1] = (int32_t)(contrib.size() / 2);
        }
        if (contrib.empty()) contrib.push_back(0);
        ACC_HIP(hipMalloc(&h->wg_ranges, sizeof(int64_t) * ranges.size()));
        ACC_HIP(hipMemcpy(h->wg_ranges, ranges.data(), sizeof(int64_t) * ranges.size(), hipMemcpyHostToDevice));
        ACC_HIP(hipMalloc(&h->gram_cstart, sizeof(int32_t) * cstart.size()));
        ACC_HIP(hipMemcpy(h->gram_cstart, cstart.data(), sizeof(int32_t) * cstart.size(), hipMemcpyHostToDevice));
        ACC_HIP(hipMalloc(&h->gram_contrib, sizeof(int32_t) * contrib.size()));
        ACC_HIP(hipMemcpy(h->gram_contrib, contrib.data(), sizeof(int32_t) * contrib.size(), hipMemcpyHostToDevice));
    }
    ACC_HIP(hipMalloc(&h->slabs, sizeof(double) * (size_t)grid * h->gram_nslot * BM * BN));

    // ---- inverse merge plan: binary tree over the NB-blocks of L.
    // Per level two products: T1 = L21 * W11, then W21 = -W22 * T1.  The 64 x 64-tile kernel is bound by
    // the latency of its k-steps, and the top levels have few tiles with long K: products with K >= 512
    // are cut into pieces of 256 along K (written to Pbuf, zero ranges of the triangular operands
    // skipped per piece) and summed by a reduction launch in a fixed order.
    const int T = (int)((m + NB - 1) / NB);
    h->ops_host.clear();
    h->merge_stages.clear();
    std::vector<RedOp> reds;
    constexpr int64_t KP = 256, KSPLIT_MIN = 512;
    ACC_HIP(hipMalloc(&h->Pbuf, sizeof(double) * (size_t)m * m));
    for (int span = 1; span < T; span *= 2) {
        std::vector<GemmOp> first, second;
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
            a.tri = 1;                                                  // W11 is lower triangular
            GemmOp b{};
            b.A = h->Wbuf + r2 * m + r2; b.lda = m;                     // W22 (s2 x s2)
            b.B = h->Tbuf + r2 * m + r1; b.ldb = m;                     // T1, B[k][col]
            b.C = h->Wbuf + r2 * m + r1; b.ldc = m;                     // W21
            b.M = (int)s2; b.N = (int)s1; b.K = (int)s2; b.lower_only = 0; b.alpha = -1.0; b.beta = 0.0;
            b.tri = 2;                                                  // W22 is lower triangular
            first.push_back(a);
            second.push_back(b);
        }
        for (std::vector<GemmOp>* list : {&first, &second}) {
            // split when every product of the list is a whole number of K pieces (full-size groups; N even)
            bool split = !list->empty();
            size_t pdoubles = 0;
            for (const GemmOp& o : *list) {
                split = split && o.K >= KSPLIT_MIN && o.K % KP == 0 && (o.N % 2 == 0) && (o.ldc % 2 == 0);
                pdoubles += (size_t)(o.K / KP) * o.M * o.N;
            }
            split = split && pdoubles <= (size_t)m * m;
            accbpg_dopt::MergeStage st{0, (int)h->ops_host.size(), 0, 0, 0};
            accbpg_dopt::MergeStage rs{1, (int)reds.size(), 0, 0, 0};
            double* pb = h->Pbuf;
            for (const GemmOp& o : *list) {
                st.maxm = std::max(st.maxm, o.M); st.maxn = std::max(st.maxn, o.N);
                if (!split) { h->ops_host.push_back(o); continue; }
                const int ns = (int)(o.K / KP);
                RedOp r{o.C, pb, o.ldc, o.M, o.N, ns, 0};
                reds.push_back(r);
                rs.maxm = std::max(rs.maxm, o.M * o.N);
                for (int p = 0; p < ns; ++p) {
                    GemmOp q = o;
                    q.A = o.A + p * KP;                                 // A[row][k]: k is the fast index
                    q.B = o.B + p * KP * o.ldb;                         // B[k][col]
                    q.K = (int)KP;
                    q.koff = (int)(p * KP);
                    q.C = pb; q.ldc = o.N;
                    pb += (size_t)o.M * o.N;
                    h->ops_host.push_back(q);
                }
            }
            st.end = (int)h->ops_host.size();
            h->merge_stages.push_back(st);
            if (split) { rs.end = (int)reds.size(); h->merge_stages.push_back(rs); }
        }
    }
    if (!h->ops_host.empty()) {
        ACC_HIP(hipMalloc(&h->ops, sizeof(GemmOp) * h->ops_host.size()));
        ACC_HIP(hipMemcpy(h->ops, h->ops_host.data(), sizeof(GemmOp) * h->ops_host.size(), hipMemcpyHostToDevice));
    }
    h->red_host = reds;
    if (!reds.empty()) {
        ACC_HIP(hipMalloc(&h->red, sizeof(RedOp) * reds.size()));
        ACC_HIP(hipMemcpy(h->red, reds.data(), sizeof(RedOp) * reds.size(), hipMemcpyHostToDevice));
    }
    ACC_HIP(hipMalloc(&h->chol_op, sizeof(GemmOp) * (size_t)(T + 1)));

    // ---- one-launch Cholesky: a workgroup per tile (the owner of a diagonal tile also owns the tile left of it)
    if (T <= CT_TMAX) {
        std::vector<CholJob> jobs;
        for (int d2 = 0; d2 < T; ++d2) jobs.push_back(CholJob{d2, d2});            // the chain first
        for (int j = 0; j < T; ++j)                                               // then by urgency: leftmost columns
            for (int i = j + 2; i < T; ++i) jobs.push_back(CholJob{i, j});
        ACC_TRY(set_lds(chol_tiles_kernel, CT_LDS_BYTES));
        int per_cu = 0;
        ACC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chol_tiles_kernel, NTHREADS, CT_LDS_BYTES));
        h->chol_tiles_grid = (int)jobs.size();
        h->chol_slots = std::min(per_cu, 2) * h->num_cu;         // workgroups of this kernel the chip holds at once
        h->chol_tiles_ok = per_cu >= 1 && (int64_t)jobs.size() <= (int64_t)h->chol_slots;
        if (h->chol_tiles_ok) {
            ACC_HIP(hipMalloc(&h->chol_jobs, sizeof(CholJob) * jobs.size()));
            ACC_HIP(hipMemcpy(h->chol_jobs, jobs.data(), sizeof(CholJob) * jobs.size(), hipMemcpyHostToDevice));
            ACC_HIP(hipMalloc(&h->chol_ready, sizeof(int) * (size_t)(T * T + 5 * T)));
            ACC_HIP(hipMemset(h->chol_ready, 0, sizeof(int) * (size_t)(T * T + 5 * T)));
            ACC_HIP(hipMalloc(&h->chol_hand, sizeof(double) * (size_t)T * NB * NB));
            ACC_HIP(hipMalloc(&h->chol_aux, sizeof(double) * (size_t)T * CT_AUX));
            ACC_HIP(hipMalloc(&h->Gbuf, sizeof(double) * (size_t)m * m));
            ACC_HIP(hipMemset(h->Gbuf, 0, sizeof(double) * (size_t)m * m));
        }
    }

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
    ACC_TRY(set_lds(chol_step_kernel, CHOL_LDS_BYTES));
    ACC_TRY(set_lds(chol_syrk_kernel, SYRK_LDS_BYTES));
    ACC_TRY(set_lds(gemm_big_kernel<TileBig<true>>, TileBig<true>::LDS_BYTES));
    ACC_TRY(set_lds(gemm_big_kernel<TileBig<false>>, TileBig<false>::LDS_BYTES));
    return ACCBPG_OK;
}

template <class T>
static int gram_launch_t(accbpg_dopt* h, const double* x, double* gram) {
    if constexpr (!T::EDGE && T::BM == 256) {
        if (h->use_glds || h->has_duals) {
            // production: the dealt-out schedule, placement B, loads two to an M0 write (GRAM_GV); the development switch
            // selects one load per M0 write through the builtin (1: round 3's first form), four to an M0 write (2), or
            // the block schedule of round 2 (3) -- all give bit-identical results.
            // Long rows (gram_chunks > 1): one launch per column block of V, each adding its Gram matrix to G.
            const int64_t nc = h->gram_chunks > 1 ? h->gram_nc : h->n;
            for (int c = 0; c < h->gram_chunks; ++c) {
                const double* Vc = h->Vblk ? h->Vblk + (int64_t)c * h->m * nc : h->V + (int64_t)c * nc;
                const int64_t ldc = h->Vblk ? nc : h->ldv;
                const double* xc = x + (int64_t)c * nc;
                const double beta = c == 0 ? 0.0 : 1.0;
                const int all_slabs = h->gram_chunks > 1 ? 1 : 0;   // (the plan of a chunked handle lists every segment)
                prof_begin(h, PROF_GRAM);
#define ACC_GRAM_ARGS Vc, ldc, h->m, nc, xc, h->tiles, h->wg_ranges, h->kiters, h->gram_nslot, h->slabs, gram, h->m, all_slabs
                if (h->kern_variant == 1)
                    ACC_LAUNCH_LDS((gram_streamk_glds_kernel<T, GRAM_GV ^ 128>), h->gram_grid, NTHREADS, T::G_LDS_BYTES, h->stream, ACC_GRAM_ARGS);
                else if (h->kern_variant == 3)
                    ACC_LAUNCH_LDS((gram_streamk_glds_kernel<T, 0>), h->gram_grid, NTHREADS, T::G_LDS_BYTES, h->stream, ACC_GRAM_ARGS);
                else if (h->kern_variant == 2)
                    ACC_LAUNCH_LDS((gram_streamk_glds_kernel<T, (GRAM_GV ^ 128) | 256>), h->gram_grid, NTHREADS, T::G_LDS_BYTES, h->stream, ACC_GRAM_ARGS);
                else
                    ACC_LAUNCH_LDS((gram_streamk_glds_kernel<T, GRAM_GV>), h->gram_grid, NTHREADS, T::G_LDS_BYTES, h->stream, ACC_GRAM_ARGS);
#undef ACC_GRAM_ARGS
                ACC_HIP(hipGetLastError());
                prof_end(h, PROF_GRAM);
                prof_begin(h, PROF_GRAMFIX);
                gram_fixup_kernel<T><<<h->ntiles * 2 * T::MI * FIX_PJ, NTHREADS, 0, h->stream>>>(
                    h->tiles, h->gram_cstart, h->gram_contrib, h->gram_nslot, h->slabs, gram, h->m, h->m, beta);
                prof_end(h, PROF_GRAMFIX);
            }
            return ACCBPG_OK;
        }
    }
    prof_begin(h, PROF_GRAM);
    gram_streamk_kernel<T><<<h->gram_grid, NTHREADS, T::LDS_BYTES, h->stream>>>(
        h->V, h->ldv, h->m, h->n, x, h->tiles, h->wg_ranges, h->kiters, h->gram_nslot, h->slabs, gram, h->m, h->vec_ok);
    ACC_HIP(hipGetLastError());
    prof_end(h, PROF_GRAM);
    prof_begin(h, PROF_GRAMFIX);
    gram_fixup_kernel<T><<<h->ntiles * 2 * T::MI * FIX_PJ, NTHREADS, 0, h->stream>>>(h->tiles, h->gram_cstart, h->gram_contrib, h->gram_nslot,
                                                                h->slabs, gram, h->m, h->m, 0.0);
    prof_end(h, PROF_GRAMFIX);
    return ACCBPG_OK;
}

// timing ablations of the Gram kernel (big interior tile only); returns average ms over `iters`
int debug_gram_variant(accbpg_dopt* h, const double* x, int var, int iters, double* ms_out) {
    using T = TileBig<false, false>;
    if (!h->big) return ACCBPG_ERR_ARG;
    if (h->has_duals && var < 10) return ACCBPG_ERR_ARG;     // the register-staged variants know no dual tiles
    hipEvent_t a, b;
    ACC_HIP(hipEventCreate(&a));
    ACC_HIP(hipEventCreate(&b));
    auto launch = [&]() -> int {
#define ACC_LAUNCH_VAR(VV)                                                                                       \
    gram_streamk_kernel<T, VV><<<h->gram_grid, NTHREADS, T::LDS_BYTES, h->stream>>>(                             \
        h->V, h->ldv, h->m, h->n, x, h->tiles, h->wg_ranges, h->kiters, h->gram_nslot, h->slabs, h->Tbuf, h->m, h->vec_ok)
#define ACC_LAUNCH_G(GG)                                                                                         \
    ACC_LAUNCH_LDS((gram_streamk_glds_kernel<T, GG>), h->gram_grid, NTHREADS, T::G_LDS_BYTES, h->stream,          \
        h->V, h->ldv, h->m, h->n, x, h->tiles, h->wg_ranges, h->kiters, h->gram_nslot, h->slabs, h->Tbuf, h->m, 0)
        switch (var) {
            case 10: ACC_LAUNCH_G(0); break;
            case 11: ACC_LAUNCH_G(1); break;
            case 12: ACC_LAUNCH_G(2); break;
            case 13: ACC_LAUNCH_G(3); break;
            case 14: ACC_LAUNCH_G(4); break;
            case 15: ACC_LAUNCH_G(8); break;
            case 16: ACC_LAUNCH_G(7); break;
            case 17: ACC_LAUNCH_G(16); break;
            case 18: ACC_LAUNCH_G(32); break;
            case 19: ACC_LAUNCH_G(96); break;
            case 20: ACC_LAUNCH_G(97); break;     /* dealt-out, no loads in the loop */
            case 21: ACC_LAUNCH_G(98); break;     /* dealt-out, no wait + barrier */
            case 22: ACC_LAUNCH_G(100); break;    /* dealt-out, no fragment reads */
            case 23: ACC_LAUNCH_G(104); break;    /* dealt-out, barrier without the vmcnt wait */
            case 24: ACC_LAUNCH_G(103); break;    /* dealt-out, MFMA only */
            case 25: ACC_LAUNCH_G(102); break;    /* dealt-out, loads only (no barrier, no reads) */
            case 26: ACC_LAUNCH_G(101); break;    /* dealt-out, barrier only (no loads, no reads) */
            case 27: ACC_LAUNCH_G(99); break;     /* dealt-out, reads only (no loads, no barrier) */
            case 28: ACC_LAUNCH_G(224); break;    /* dealt-out, loads two to an M0 write */
            case 29: ACC_LAUNCH_G(352); break;    /* dealt-out, loads four to an M0 write */
            case 0: ACC_LAUNCH_VAR(0); break;
            case 1: ACC_LAUNCH_VAR(1); break;
            case 2: ACC_LAUNCH_VAR(2); break;
            case 3: ACC_LAUNCH_VAR(3); break;
            default: ACC_LAUNCH_VAR(4); break;
        }
#undef ACC_LAUNCH_VAR
        return ACCBPG_OK;
    };
    ACC_TRY(set_lds(gram_streamk_kernel<T, 1>, T::LDS_BYTES));
    ACC_TRY(set_lds(gram_streamk_kernel<T, 2>, T::LDS_BYTES));
    ACC_TRY(set_lds(gram_streamk_kernel<T, 3>, T::LDS_BYTES));
    ACC_TRY(set_lds(gram_streamk_kernel<T, 4>, T::LDS_BYTES));
    ACC_TRY(launch());
    ACC_HIP(hipEventRecord(a, h->stream));
    for (int i = 0; i < iters; ++i) ACC_TRY(launch());
    ACC_HIP(hipEventRecord(b, h->stream));
    ACC_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    ACC_HIP(hipEventElapsedTime(&ms, a, b));
    *ms_out = ms / iters;
    hipEventDestroy(a);
    hipEventDestroy(b);
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

int launch_gram(accbpg_dopt* h, const double* x, double* gram) {
    bool xal = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (!xal && (h->has_duals || h->gram_chunks > 1)) {
        // the tile list was built for the direct-to-LDS kernel, which reads x in 16-byte pieces
        if (!h->xbuf) ACC_HIP(hipMalloc(&h->xbuf, sizeof(double) * (size_t)h->n));
        ACC_TRY(device_copy(h->xbuf, x, (size_t)h->n, h->stream));
        x = h->xbuf;
        xal = true;
    }
    if (h->big) {
        const bool interior = h->vec_ok && xal && (h->m % 256 == 0) && (h->n % BK == 0);
        if (interior) ACC_TRY((gram_launch_t<TileBig<false, false>>(h, x, gram)));
        else ACC_TRY((gram_launch_t<TileBig<false, true>>(h, x, gram)));
    } else {
        const bool interior = h->vec_ok && xal && (h->m % 64 == 0) && (h->n % BK == 0);
        if (interior) ACC_TRY((gram_launch_t<TileSmall<false, false>>(h, x, gram)));
        else ACC_TRY((gram_launch_t<TileSmall<false, true>>(h, x, gram)));
    }
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

__global__ __launch_bounds__(256) void copy_kernel(double* __restrict__ dst, const double* __restrict__ src, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
int device_copy(double* dst, const double* src, size_t ndoubles, hipStream_t s) {
    if (ndoubles == 0 || dst == src) return ACCBPG_OK;
    size_t nb = (ndoubles + 255) / 256;
    if (nb > 4096) nb = 4096;
    copy_kernel<<<(unsigned)nb, 256, 0, s>>>(dst, src, ndoubles);
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
// and dflag[FLAG_NOT_PD].  T = ceil(m/64) launches.
// The one-launch kernel needs all of its workgroups resident together, so two of them must never share the chip:
// launches of this process are chained through one event per device (the second waits for the first to finish;
// everything else on the streams still overlaps).  Another PROCESS on the same GPU is what the bounded spins are for.
// Small factorisations (few workgroups each) may share the chip as long as ALL the launches in flight fit together:
// a launch waits for every launch of the process that was issued `together` or more launches before it, where
// `together` = workgroup slots of the chip / the largest grid among the recent launches.  Because each launch waits
// for that whole window, everything older has finished too, so at most `together` launches are ever in flight.
static std::mutex g_tiles_mu;
constexpr int TILES_RING = 16;
struct TilesRing {
    hipEvent_t ev[TILES_RING] = {};
    hipStream_t stream[TILES_RING] = {};
    int grid[TILES_RING] = {};
    long long issued = 0;
};
static TilesRing g_tiles[64];

bool chol_tiles_usable(const accbpg_dopt* h) {
    return h->chol_tiles_ok && !h->chol_tiles_off && (h->chol_dbg & 63) == 0;
}

static int launch_chol_tiles(accbpg_dopt* h, const double* src, double* A, double* Winv) {
    const int64_t m = h->m;
    const int T = (int)((m + NB - 1) / NB);
    CholInst ci;
    ci.src = src; ci.L = A; ci.Ldiag = h->Tbuf; ci.Winv = Winv; ci.logdet = h->dscal; ci.flags = h->dflag;
    ci.ready = h->chol_ready; ci.aux = h->chol_aux; ci.hand = h->chol_hand; ci.trace = h->chol_trace;
    const int dev = (h->device >= 0 && h->device < 64) ? h->device : 0;
    std::lock_guard<std::mutex> lk(g_tiles_mu);
    TilesRing& ring = g_tiles[dev];
    {
        int gmax = h->chol_tiles_grid;
        for (int i = 0; i < TILES_RING; ++i) gmax = std::max(gmax, ring.grid[i]);
        int together = std::max(1, h->chol_slots) / std::max(1, gmax);
        if (together > TILES_RING - 1) together = TILES_RING - 1;   // (so that the waits of successive launches chain)
        if (together < 1) together = 1;
        for (long long back = together; back <= TILES_RING && back <= ring.issued; ++back) {
            const int slot = (int)((ring.issued - back) % TILES_RING);
            if (ring.ev[slot] && ring.stream[slot] != h->stream) ACC_HIP(hipStreamWaitEvent(h->stream, ring.ev[slot], 0));
        }
    }
    chol_tiles_kernel<<<dim3(h->chol_tiles_grid, 1), NTHREADS, CT_LDS_BYTES, h->stream>>>(
        ci, nullptr, reinterpret_cast<const CholJob*>(h->chol_jobs), m, m, T, h->chol_spin_limit, h->chol_stall_test, BatchAct{});
    ACC_HIP(hipGetLastError());
    {
        const int slot = (int)(ring.issued % TILES_RING);
        if (!ring.ev[slot]) ACC_HIP(hipEventCreateWithFlags(&ring.ev[slot], hipEventDisableTiming));
        ACC_HIP(hipEventRecord(ring.ev[slot], h->stream));
        ring.stream[slot] = h->stream;
        ring.grid[slot] = h->chol_tiles_grid;
        ++ring.issued;
    }
    return ACCBPG_OK;
}

int launch_cholesky(accbpg_dopt* h, double* A, double* Winv, const double* xcheck, const double* src) {
    const int64_t m = h->m;
    const int T = (int)((m + NB - 1) / NB);
    if (src == nullptr) src = A;
    const bool tiles = chol_tiles_usable(h);
    prof_begin(h, PROF_CHOL);
    // 512 threads, not 1024: two waves of 32 registers per SIMD fit beside a wave of the Gram kernel (392 of the 512
    // registers of a SIMD lane), four do not -- and this launch often meets the OTHER stream's Gram launch, which holds
    // every compute unit for 2 ms (its rocprofv3 average with 1024 threads in the steady state of ABPG_gain: 328 us,
    // nearly all of it waiting for a compute unit)
    zero_scalars_kernel<<<1, (xcheck || tiles) ? 512 : 64, 0, h->stream>>>(h->dscal, h->dflag, xcheck, h->n,
                                                                         tiles ? h->chol_ready : nullptr, tiles ? T * T + 5 * T : 0);
    if (tiles) {
        ACC_TRY(launch_chol_tiles(h, src, A, Winv));
        h->diag_inv_ready = (Winv != nullptr);
        prof_end(h, PROF_CHOL);
        return ACCBPG_OK;
    }
    if (src != A) ACC_TRY(device_copy(A, src, (size_t)m * m, h->stream));
    // One level (every launch updates the whole trailing matrix) up to chol_two_level_T block columns; beyond,
    // outer panels of chol_nk (8) block columns: the step launches stay inside the panel, chol_syrk_kernel
    // applies the panel to the rest in one pass.
    const int nkp = (T >= h->chol_two_level_T) ? h->chol_nk : T;
    for (int k0 = 0; k0 < T; k0 += nkp) {
        const int kend = std::min(k0 + nkp, T) - 1;             // last block column of this outer panel
        for (int kc = k0; kc <= kend; ++kc) {
            const int pend = (kc > k0) ? 1 : 0;
            const int npanel = T - kc;
            const int R = T - (kc + 1);
            int nupd = 0;
            if (pend) nupd = (kend >= T - 1) ? R * (R + 1) / 2 : R * (kend - kc);
            chol_step_kernel<<<npanel + nupd, NTHREADS, CHOL_LDS_BYTES, h->stream>>>(
                A, m, m, kc, pend, kend, T, h->dscal, h->dflag, h->chol_dbg, Winv, h->Tbuf);
        }
        const int Rr = T - (kend + 1);
        if (Rr > 0)
            chol_syrk_kernel<<<Rr * (Rr + 1) / 2, NTHREADS, SYRK_LDS_BYTES, h->stream>>>(A, m, m, k0, kend - k0 + 1, T);
    }
    h->diag_inv_ready = (Winv != nullptr);
    prof_end(h, PROF_CHOL);
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

// W = L^-1 for the factor in h->Lbuf.
int launch_trtri(accbpg_dopt* h) {
    const int64_t m = h->m;
    const int T = (int)((m + NB - 1) / NB);
    prof_begin(h, PROF_TRTRI);
    // (the factorisation leaves the inverses of the diagonal blocks in Wbuf when it was asked to)
    // (the diagonal blocks of the factor live in Tbuf's diagonal blocks, which the merges do not use)
    if (!h->diag_inv_ready) trtri_diag_kernel<<<T, 64, 0, h->stream>>>(h->Tbuf, m, h->Wbuf, m, m);
    h->diag_inv_ready = false;
    for (const accbpg_dopt::MergeStage& st : h->merge_stages) {
        if (st.kind == 0) {
            ACC_TRY(launch_gemm_ops(h->ops + st.begin, st.end - st.begin, st.maxm, st.maxn, true, h->stream));
        } else {
            int gx = (int)std::min<int64_t>(1024, ((int64_t)st.maxm / 2 + 255) / 256);
            gemm_reduce_kernel<<<dim3(gx, st.end - st.begin), 256, 0, h->stream>>>(h->red + st.begin);
        }
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
        if (interior && h->use_glds) {
            using T = TileBig<true, false>;
            const int grid = (int)(h->n / T::BN), lds = T::G_LDS_BYTES + 4 * T::BN * 8;
            if (h->kern_variant == 1)
                ACC_LAUNCH_LDS((colnorm_glds_kernel<T, COLNORM_CV ^ 256>), grid, NTHREADS, lds, h->stream, W, h->m, h->V, h->ldv, h->m, h->n, out, sign);
            else if (h->kern_variant == 2)
                ACC_LAUNCH_LDS((colnorm_glds_kernel<T, 96>), grid, NTHREADS, lds, h->stream, W, h->m, h->V, h->ldv, h->m, h->n, out, sign);
            else
                ACC_LAUNCH_LDS((colnorm_glds_kernel<T, COLNORM_CV>), grid, NTHREADS, lds, h->stream, W, h->m, h->V, h->ldv, h->m, h->n, out, sign);
        } else if (interior) colnorm_launch_t<TileBig<true, false>>(h, W, out, sign, vw);
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

// ---------------------------------------------------------------------------------------------------------
// Batched launches: the active instances of a batch of same-shaped problems, one launch per kernel family.
// Only the interior big-tile path (m a multiple of 256, n of 128, 16-byte aligned rows) is batched; the caller
// evaluates other shapes instance by instance.
// ---------------------------------------------------------------------------------------------------------
int launch_gram_batch(accbpg_dopt_batch* b, const BatchAct& act, const double* xbase, int64_t ldx) {
    using T = TileBig<false, false>;
    accbpg_dopt* h0 = b->inst[0];
    static bool lds_set = false;
    if (!lds_set) {
        ACC_TRY(set_lds(gram_streamk_glds_batch_kernel<T>, T::G_LDS_BYTES));
        lds_set = true;
    }
    // (kernel-time accounting of a batch rides on instance 0's slots: accbpg_dopt_profile_* on accbpg_dopt_batch_instance(b, 0))
    prof_begin(h0, PROF_GRAM);
    gram_streamk_glds_batch_kernel<T><<<dim3(h0->gram_grid, act.n), NTHREADS, T::G_LDS_BYTES, b->stream>>>(
        b->table, act, h0->ldv, h0->m, h0->n, xbase, ldx, h0->tiles, h0->wg_ranges, h0->kiters, h0->gram_nslot, h0->m);
    prof_end(h0, PROF_GRAM);
    prof_begin(h0, PROF_GRAMFIX);
    gram_fixup_batch_kernel<T><<<dim3(h0->ntiles * 2 * T::MI * FIX_PJ, act.n), NTHREADS, 0, b->stream>>>(
        b->table, act, h0->tiles, h0->gram_cstart, h0->gram_contrib, h0->gram_nslot, h0->m, h0->m);
    prof_end(h0, PROF_GRAMFIX);
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

int launch_cholesky_batch(accbpg_dopt_batch* b, const BatchAct& act, bool with_inverse, const double* xbase, int64_t ldx) {
    accbpg_dopt* h0 = b->inst[0];
    const int64_t m = h0->m;
    const int T = (int)((m + NB - 1) / NB);
    prof_begin(h0, PROF_CHOL);
    zero_scalars_batch_kernel<<<act.n, 512, 0, b->stream>>>(b->table, act, xbase, ldx, h0->n, T * T + 5 * T);
    const int dev = (b->device >= 0 && b->device < 64) ? b->device : 0;
    const int grid_all = h0->chol_tiles_grid * act.n;
    std::lock_guard<std::mutex> lk(g_tiles_mu);
    TilesRing& ring = g_tiles[dev];
    {
        int gmax = grid_all;
        for (int i = 0; i < TILES_RING; ++i) gmax = std::max(gmax, ring.grid[i]);
        int together = std::max(1, h0->chol_slots) / std::max(1, gmax);
        if (together > TILES_RING - 1) together = TILES_RING - 1;
        if (together < 1) together = 1;
        for (long long back = together; back <= TILES_RING && back <= ring.issued; ++back) {
            const int slot = (int)((ring.issued - back) % TILES_RING);
            if (ring.ev[slot] && ring.stream[slot] != b->stream) ACC_HIP(hipStreamWaitEvent(b->stream, ring.ev[slot], 0));
        }
    }
    chol_tiles_kernel<<<dim3(h0->chol_tiles_grid, act.n), NTHREADS, CT_LDS_BYTES, b->stream>>>(
        CholInst{}, b->chol_table[with_inverse ? 1 : 0], reinterpret_cast<const CholJob*>(h0->chol_jobs), m, m, T,
        h0->chol_spin_limit, 0, act);
    prof_end(h0, PROF_CHOL);
    ACC_HIP(hipGetLastError());
    {
        const int slot = (int)(ring.issued % TILES_RING);
        if (!ring.ev[slot]) ACC_HIP(hipEventCreateWithFlags(&ring.ev[slot], hipEventDisableTiming));
        ACC_HIP(hipEventRecord(ring.ev[slot], b->stream));
        ring.stream[slot] = b->stream;
        ring.grid[slot] = grid_all;
        ++ring.issued;
    }
    return ACCBPG_OK;
}

int launch_trtri_batch(accbpg_dopt_batch* b, const BatchAct& act) {
    accbpg_dopt* h0 = b->inst[0];
    prof_begin(h0, PROF_TRTRI);
    for (const accbpg_dopt::MergeStage& st : h0->merge_stages) {
        const int count = st.end - st.begin;
        if (count <= 0) continue;
        if (st.kind == 0) {
            dim3 grid((st.maxn + 63) / 64, (st.maxm + 63) / 64, count * act.n);
            gemm_ops_batch_kernel<TileSmall<true>><<<grid, NTHREADS, TileSmall<true>::LDS_BYTES, b->stream>>>(
                b->ops_all, b->ops_per_inst, st.begin, count, act);
        } else {
            int gx = (int)std::min<int64_t>(1024, ((int64_t)st.maxm / 2 + 255) / 256);
            gemm_reduce_batch_kernel<<<dim3(gx, count * act.n), 256, 0, b->stream>>>(b->red_all, b->red_per_inst, st.begin, count, act);
        }
    }
    prof_end(h0, PROF_TRTRI);
    ACC_HIP(hipGetLastError());
    return ACCBPG_OK;
}

int launch_colnorm_batch(accbpg_dopt_batch* b, const BatchAct& act, double* gbase, int64_t ldg, double sign) {
    using T = TileBig<true, false>;
    accbpg_dopt* h0 = b->inst[0];
    static bool lds_set = false;
    if (!lds_set) {
        ACC_TRY(set_lds(colnorm_glds_batch_kernel<T>, T::G_LDS_BYTES + 4 * T::BN * 8));
        lds_set = true;
    }
    prof_begin(h0, PROF_GRAD);
    colnorm_glds_batch_kernel<T><<<dim3((unsigned)(h0->n / T::BN), act.n), NTHREADS, T::G_LDS_BYTES + 4 * T::BN * 8, b->stream>>>(
        b->table, act, h0->m, h0->ldv, h0->m, h0->n, gbase, ldg, sign);
    prof_end(h0, PROF_GRAD);
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

int pipe_probe(int iters, int mode, double* ms_out, hipStream_t s) {
    int dev = 0;
    hipDeviceProp_t prop;
    ACC_HIP(hipGetDevice(&dev));
    ACC_HIP(hipGetDeviceProperties(&prop, dev));
    double* sink = nullptr;
    ACC_HIP(hipMalloc(&sink, 8));
    hipEvent_t a, b;
    ACC_HIP(hipEventCreate(&a));
    ACC_HIP(hipEventCreate(&b));
    pipe_probe_kernel<<<prop.multiProcessorCount, NTHREADS, 0, s>>>(iters / 10 + 1, mode, sink);
    ACC_HIP(hipEventRecord(a, s));
    pipe_probe_kernel<<<prop.multiProcessorCount, NTHREADS, 0, s>>>(iters, mode, sink);
    ACC_HIP(hipEventRecord(b, s));
    ACC_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    ACC_HIP(hipEventElapsedTime(&ms, a, b));
    *ms_out = ms;
    hipEventDestroy(a); hipEventDestroy(b); hipFree(sink);
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
