// Internal declarations shared by the translation units of libaccbpg_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#include "../../include/accbpg_hip.h"

namespace accbpg {

void set_last_error(const char* fmt, ...);
void note_tiles_fallback(const char* where);

#define ACC_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            ::accbpg::set_last_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call,            \
                                     hipGetErrorString(e__));                                 \
            return ACCBPG_ERR_HIP;                                                            \
        }                                                                                     \
    } while (0)

#define ACC_TRY(call)                       \
    do {                                    \
        int rc__ = (call);                  \
        if (rc__ != ACCBPG_OK) return rc__; \
    } while (0)

// one product of the batched small-GEMM kernel: C = alpha * A * op(B) + beta * C
struct GemmOp {
    const double* A;
    const double* B;
    double* C;
    int64_t lda, ldb, ldc;
    int32_t M, N, K;
    int32_t lower_only;     // skip tiles / elements strictly above the diagonal of C
    int32_t tri;            // bit 0: B[k][col] is zero for k < col; bit 1: A[row][k] is zero for k > row (k ranges are cut)
    int32_t koff;           // k index of this op's first k-step in the coordinates of those triangles (split-K pieces)
    double alpha, beta;
};

// sum of split-K partial products: C[r][c] = sum_p P[p*M*N + r*N + c]
struct RedOp {
    double* C;
    const double* P;
    int64_t ldc;
    int32_t M, N, ns, pad;
};

// entry of the Gram tile list: one BM x BN tile (d1 < 0), or a pair of 128 x 128 diagonal blocks d1, d2
// (128-block indices) that run as "dual" tiles over half of K each (mfma_tile.hpp)
struct TileRC {
    int32_t rb, cb;
    int32_t d1 = -1, d2 = -1;
};

// A batch of same-shaped instances evaluated by ONE launch per kernel family (blockIdx.y = position among the
// active instances): the per-instance pointers live in a device table, the active set travels as a kernel argument.
constexpr int BATCH_MAX = ACCBPG_BATCH_MAX;
struct BatchAct {
    int n = 0;
    int idx[BATCH_MAX] = {};
};
struct BatchInst {
    const double* V;        // design matrix of the instance
    double* slabs;          // stream-K slabs
    double* gram;           // Gram matrix target of func_grad
    double* Lbuf;
    double* Wbuf;
    double* Tbuf;
    double* dscal;          // scalars + status flags of the instance (slice of one array for the whole batch)
    int* dflag;
    int* chol_ready;
};

struct CholJob { int i, j; };                     // i == j: owner of the diagonal tile (and of (i, i-1))
struct CholInst {                                  // one factorisation (one entry per instance of a batched launch)
    const double* src;      // matrix to factor (lower triangle significant), leading dimension ld
    double* L;              // off-diagonal tiles of the factor (may be src: in place)
    double* Ldiag;          // diagonal tiles of the factor
    double* Winv;           // inverses of the diagonal tiles (or null)
    double* logdet;         // scalar result
    int* flags;             // status flags (FLAG_NOT_PD, FLAG_ABORT)
    int* ready;             // hand-off flags, zero at launch: T*T tile flags, 4 per block column (the 16-column pieces of
                            // its factor), 1 per block row (its diagonal tile with the left updates applied)
    double* aux;            // T * CT_AUX doubles
    double* hand;           // T * 64*64 doubles: diagonal tiles on their way from their accumulators to the chain
    long long* trace;       // development aid: CT_NSTAMP wall-clock stamps per block column from the chain workgroups (or null)
};

constexpr int NB = 64;          // Cholesky / inverse block size
constexpr int FLAG_NOT_PD = 0;  // index into the device flag array
constexpr int FLAG_NEG_X = 1;
constexpr int FLAG_NONPOS = 2;
constexpr int FLAG_BAD_G = 3;
constexpr int FLAG_ABORT = 4;   // the one-launch Cholesky gave up a wait (co-residency not reached in time)
constexpr int STATUS_DOUBLES = 20;   // 16 scalars + 8 status flags, copied to the host in one piece
constexpr int ACCBPG_RETRY = 100;    // internal: redo the evaluation with the launch-per-column Cholesky

enum ProfKind { PROF_GRAM = 0, PROF_CHOL = 1, PROF_TRTRI = 2, PROF_GRAD = 3, PROF_GRAMFIX = 4, PROF_FWV = 5, PROF_COUNT = 6 };

struct ProfSlot {
    std::vector<hipEvent_t> ev;   // pairs (start, stop)
    size_t used = 0;
    double total_ms = 0.0;
    int64_t launches = 0;
};

}  // namespace accbpg

struct accbpg_dopt {
    const double* V = nullptr;
    int64_t m = 0, n = 0, ldv = 0;
    hipStream_t stream = nullptr;
    int device = 0;
    int num_cu = 256;
    bool big = false;           // use the 256x128 tile for Gram / gradient products
    // set before the plans are built when the handle is one instance of a batch (accbpg_dopt_batch_create)
    bool force_big = false;     // 256x128 tiles also below m = 768 (interior shapes only)
    int gram_grid_cap = 0;      // stream-K workgroups of this instance's Gram launch (0: the whole chip)
    double* dscal_ext = nullptr;   // scalars + flags live in this slice of the batch's array instead of an own allocation
    bool vec_ok = false;        // V rows are 16-byte aligned

    double* Lbuf = nullptr;     // m*m : Gram matrix, then its Cholesky factor (lower)
    double* Wbuf = nullptr;     // m*m : inverse of the factor (lower, upper stays zero)
    double* Tbuf = nullptr;     // m*m : scratch of the inverse merges / FW refactorisation
    double* slabs = nullptr;    // stream-K partial accumulators
    accbpg::TileRC* tiles = nullptr;
    int64_t* wg_ranges = nullptr;     // [gram_grid][GRAM_RMAX][2] unit ranges of the Gram stream-K walk
    int32_t* gram_cstart = nullptr;   // per (entry, half): first index into gram_contrib
    int32_t* gram_contrib = nullptr;  // (workgroup, slab slot) pairs in k order
    int ntiles = 0, gram_grid = 0, gram_per = 0, gram_nslot = 2;
    int64_t kiters = 0;
    int gram_chunks = 1;        // column blocks of V the Gram matrix is formed over (one launch each; rows >= 65536 long)
    int64_t gram_nc = 0;        // columns per block
    double* Vblk = nullptr;     // owned copy of V stored block by block (rows >= 1 MiB apart only), read by the Gram launches
    double* dscal = nullptr;    // device scalars
    int* dflag = nullptr;       // device status flags
    double* hpin = nullptr;     // pinned host mirror (scalars then flags)
    double* vws = nullptr;      // vector-kernel workspace
    double* xbuf = nullptr;     // 16-byte aligned copy of an unaligned x (lazy)

    accbpg::GemmOp* ops = nullptr;            // device op table of the inverse merges
    std::vector<accbpg::GemmOp> ops_host;
    std::vector<accbpg::RedOp> red_host;
    // launch list of the inverse merges: kind 0 = products ops[begin, end), kind 1 = partial-sum reductions red[begin, end)
    struct MergeStage { int kind, begin, end, maxm, maxn; };
    std::vector<MergeStage> merge_stages;
    accbpg::RedOp* red = nullptr;
    double* Pbuf = nullptr;     // m*m: split-K partial products of the top merge levels
    accbpg::GemmOp* chol_op = nullptr;        // device slot for the trailing-update op

    // Frank-Wolfe state
    double *fw_x = nullptr, *fw_w = nullptr, *fw_H = nullptr, *fw_hv = nullptr;
    double* fw_hpin_dev = nullptr;  // device address of the pinned host record hpin (the probe's final stage writes there)
    bool fw_ready = false;
    // log det(H) of the away-step variant (D_opt_alg.py:136) factored beside the steps: a ring of snapshot slots, each an
    // auxiliary handle (own buffers, own stream) that factors a copy of H_k while the main stream goes on
    struct FwSlot {
        accbpg_dopt* aux = nullptr;     // owns snapshot / factor buffers, scalars, pinned mirror, ev_done
        hipStream_t stream = nullptr;   // created here
        hipEvent_t ev_snap = nullptr;   // recorded on the main stream behind the snapshot copy
        bool pending = false;
    };
    std::vector<FwSlot> fw_ring;
    int fw_ring_depth = 1;          // slots in use (factorisations in flight)
    int fw_ring_small = 2;          // how the slots factor: 0 one launch, 1 a launch per block column, 2 by size (as
                                    // accbpg_dopt_factor_in_small_launches)
    long long fw_ring_issued = 0, fw_ring_collected = 0;
    int fw_part_nblk = 0;       // probe stage-1 records left behind by the last w update (0: none)
    bool fw_part_away = false;  // support threshold they were computed for

    bool use_glds = true;       // direct-to-LDS staging for interior big tiles (debug switch)
    int kern_variant = 0;       // schedule of the direct-to-LDS Gram / gradient kernels (development switch)
    bool diag_inv_ready = false;  // the last factorisation wrote the inverses of the diagonal blocks into Wbuf
    bool has_duals = false;     // the Gram tile list holds dual diagonal tiles (direct-to-LDS kernel only)
    int chol_nk = 8;            // block columns per outer panel of the two-level scheme
    int chol_two_level_T = 64;  // block columns from which the Cholesky runs its two-level scheme (m > 4032)
    int chol_dbg = 0;           // timing ablation bits for chol_step_kernel (0 in production)
    // one-launch Cholesky (tile owners, T <= 32 block columns)
    bool chol_tiles_ok = false;     // the grid fits the chip (checked against the occupancy query at creation)
    bool chol_tiles_off = false;    // switched off (debug bit, or after a wait timed out)
    int chol_stall_test = 0;        // debug: make the launch time out
    long long chol_spin_limit = 20000000;   // 0.2 s of the 100 MHz wall clock
    int chol_tiles_grid = 0;
    int chol_slots = 0;             // workgroups of the one-launch kernel the chip holds at once (occupancy query x CUs, at most 2 per CU)
    void* chol_jobs = nullptr;      // CholJob[chol_tiles_grid]
    int* chol_ready = nullptr;      // T*T hand-off flags
    double* chol_aux = nullptr;     // T * CT_AUX doubles
    double* chol_hand = nullptr;    // T * 64*64 doubles: diagonal tiles handed from their accumulators to the chain
    long long* chol_trace = nullptr;   // development aid: stage stamps of the chain workgroups (null in production)
    double* Gbuf = nullptr;         // m*m: Gram matrix of func_grad when the one-launch Cholesky is in use (kept intact for a redo)
    const double* last_x = nullptr; // arguments of the evaluation in flight (for a redo)
    double* last_g = nullptr;
    int last_flag = 0;
    hipEvent_t ev_done = nullptr;   // recorded behind the result copy of every begin/end evaluation
    bool prof_on = false;
    accbpg::ProfSlot prof[accbpg::PROF_COUNT];
};

// A batch of same-shaped D-optimal instances evaluated together (accbpg_dopt_batch_*)
struct accbpg_dopt_batch {
    int K = 0;
    std::vector<accbpg_dopt*> inst;
    hipStream_t stream = nullptr;
    int device = 0;
    bool fast = false;                      // one launch per kernel family over the active instances
    int chunk = 0;                          // ... at most this many instances per launch (their one-launch
                                            // factorisations must fit the chip together)
    accbpg::BatchInst* table = nullptr;     // device, K entries
    accbpg::CholInst* chol_table[2] = {nullptr, nullptr};   // device, K entries each: without / with diagonal-block inverses
    accbpg::GemmOp* ops_all = nullptr;      // device: the merge op tables of all instances, instance after instance
    accbpg::RedOp* red_all = nullptr;
    int ops_per_inst = 0, red_per_inst = 0;
    double* dscal_all = nullptr;            // device, K * 24 doubles (scalars + flags of every instance)
    double* hpin = nullptr;                 // pinned mirror
    // scratch of the batched length-n kernels (vec_kernels.hip), allocated on first use
    int* vflags = nullptr;                  // K * 8 ints: status flags + {bisection, newton} of the prox
    double* vout = nullptr;                 // K * 4 doubles: reduction results
    double* vpart = nullptr;                // K * 4 * 1024 doubles: reduction partials
    double* vgg = nullptr;                  // K * n doubles: gg of the prox when it does not fit in registers
    double* vpin = nullptr;                 // pinned mirror (K * 8 doubles)
    // the evaluation in flight between _begin and _end
    accbpg::BatchAct pend_act;
    const double* pend_x = nullptr;
    double* pend_g = nullptr;
    int64_t pend_ldx = 0, pend_ldg = 0;
    int pend_flag = 0;
    bool pend_fused = false;
};
struct BatchVals { double v[accbpg::BATCH_MAX]; };          // one scalar per instance, as a kernel argument

namespace accbpg {

// batched launches over the active instances (dopt_kernels.hip)
int launch_gram_batch(accbpg_dopt_batch* b, const BatchAct& act, const double* xbase, int64_t ldx);
int launch_cholesky_batch(accbpg_dopt_batch* b, const BatchAct& act, bool with_inverse, const double* xbase, int64_t ldx);
int launch_trtri_batch(accbpg_dopt_batch* b, const BatchAct& act);
int launch_colnorm_batch(accbpg_dopt_batch* b, const BatchAct& act, double* gbase, int64_t ldg, double sign);

// dopt_kernels.hip
int dopt_init(accbpg_dopt* h);
int launch_gram(accbpg_dopt* h, const double* x, double* gram);
int launch_cholesky(accbpg_dopt* h, double* A /* m*m, factor goes here */, double* Winv = nullptr /* diagonal-block inverses */,
                    const double* xcheck = nullptr /* x >= 0 check folded into the reset launch */,
                    const double* src = nullptr /* matrix to factor when it is not A itself */);
bool chol_tiles_usable(const accbpg_dopt* h);
int launch_trtri(accbpg_dopt* h);
int launch_colnorm(accbpg_dopt* h, const double* W, double* out, double sign);
int launch_gemm_ops(const GemmOp* ops_dev, int nops, int maxM, int maxN, bool b_kmajor, hipStream_t s);
int launch_test_gemm(const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, int b_kmajor, double alpha, double beta, int config,
                     hipStream_t s);
int build_plans(accbpg_dopt* h);
void set_plan_flags(int flags);
int debug_gram_variant(accbpg_dopt* h, const double* x, int var, int iters, double* ms_out);
int mfma_peak(int iters, double* tflops, hipStream_t s);
int pipe_probe(int iters, int mode, double* ms_out, hipStream_t s);

// vec_kernels.hip
int64_t vec_ws_doubles(int64_t n);

// fw_kernels.hip
int vt_nsplit(int64_t m, int64_t n, int num_cu);
int launch_vt_times(const double* V, int64_t ldv, int64_t m, int64_t n, const double* q, double* upart, int nsplit,
                    double* u, bool vec_ok, hipStream_t s);
constexpr int VT_MAXSPLIT = 64;

// device-to-device copy as a kernel on the stream (no copy-engine hand-off between producer and consumer kernels)
int device_copy(double* dst, const double* src, size_t ndoubles, hipStream_t s);

// prof helpers
void prof_begin(accbpg_dopt* h, ProfKind k);
void prof_end(accbpg_dopt* h, ProfKind k);

}  // namespace accbpg
