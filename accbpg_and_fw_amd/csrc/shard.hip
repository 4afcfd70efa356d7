// One large D-optimal instance with the design points (columns of V) dealt to ranks, one process per GPU
// (BASELINE config 5; SURVEY.md section 8(e) -- no counterpart in the single-process reference).
//
// Per evaluation of f(x) = -log det(V diag(x) V^T) (accbpg/functions.py:40-60): each rank forms the Gram
// contribution of its columns, ONE RCCL all-reduce sums them over xGMI as packed lower triangles
// (m(m+1)/2 doubles, the count of negative entries of the local x riding as one more element), every rank
// factors the replicated sum, evaluates the gradient entries of its own columns, and one all-gather of the
// slices assembles the length-n gradient.  RCCL is opened at run time (librccl.so.1, the copy already in the
// process if there is one), so the library itself loads on machines without it.
#include "internal.h"

#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>

using namespace accbpg;

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[256] = "";      // what dlopen said about the last name tried
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            r.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
            const char* why = dlerror();                        // (valid right behind the failed dlopen only)
            snprintf(r.why, sizeof(r.why), "%s", why ? why : "dlopen failed without a message");
        }
        if (!r.lib) return;
        bool all = true;
        auto sym = [&](const char* nm) { void* p = dlsym(r.lib, nm); all = all && p != nullptr; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.ok = all;
    });
    return &r;
}

int need_rccl(Rccl** out) {
    Rccl* r = rccl();
    if (!r->ok) {
        set_last_error("RCCL (librccl.so.1) could not be opened: %s", r->lib ? "a symbol is missing" : r->why);
        return ACCBPG_ERR_HIP;
    }
    *out = r;
    return ACCBPG_OK;
}

#define ACC_RCCL(call)                                                                            \
    do {                                                                                          \
        ncclResult_t res_ = (call);                                                               \
        if (res_ != ncclSuccess) {                                                                \
            set_last_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, R->GetErrorString(res_)); \
            return ACCBPG_ERR_HIP;                                                                \
        }                                                                                         \
    } while (0)

}  // namespace

struct accbpg_dopt_shard {
    accbpg_dopt* local = nullptr;      // the rank's own columns V[:, lo:hi] (borrowed)
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    int world = 1, rank = 0;
    int64_t n = 0, lo = 0, hi = 0;
    int64_t tri = 0, msg_len = 0;      // m(m+1)/2, and the message length (triangle + count, padded to 16 bytes)
    int64_t piece = 0;                 // gradient entries every rank contributes to the all-gather (the longest slice)
    double* gram = nullptr;            // m*m: local contribution, then the replicated sum
    double* msg = nullptr;             // the one all-reduce message
    double* gath = nullptr;            // world * piece: gathered slices when they are not all of one length
    double* hcount = nullptr;          // pinned: the summed count of negative entries
};

extern "C" int accbpg_dopt_shard_bounds(int64_t n, int world, int rank, int64_t* lo, int64_t* hi) {
    if (n <= 0 || world <= 0 || rank < 0 || rank >= world || !lo || !hi) return ACCBPG_ERR_ARG;
    const int64_t base = n / world, extra = n % world;
    *lo = rank * base + (rank < extra ? rank : extra);
    *hi = *lo + base + (rank < extra ? 1 : 0);
    return ACCBPG_OK;
}

extern "C" int accbpg_shard_unique_id(void* id_out) {
    if (!id_out) return ACCBPG_ERR_ARG;
    Rccl* R;
    ACC_TRY(need_rccl(&R));
    ncclUniqueId id;
    ACC_RCCL(R->GetUniqueId(&id));
    static_assert(sizeof(ncclUniqueId) == ACCBPG_SHARD_ID_BYTES, "rendezvous token size");
    memcpy(id_out, &id, sizeof id);
    return ACCBPG_OK;
}

extern "C" int accbpg_dopt_shard_destroy(accbpg_dopt_shard* s) {
    if (!s) return ACCBPG_OK;
    if (s->local) hipSetDevice(s->local->device);
    if (s->own_comm && s->comm) {
        Rccl* r = rccl();
        if (r->ok) r->CommDestroy(s->comm);
    }
    hipFree(s->gram); hipFree(s->msg); hipFree(s->gath);
    if (s->hcount) hipHostFree(s->hcount);
    delete s;
    return ACCBPG_OK;
}

/* `local` is an accbpg_dopt handle over this rank's columns V[:, lo:hi], with lo, hi from accbpg_dopt_shard_bounds.
 * Either `comm` is an RCCL communicator the caller made (ncclComm_t; it is used, not owned), or it is null and
 * the communicator is made here from the 128-byte token of accbpg_shard_unique_id that rank 0 handed round. */
extern "C" int accbpg_dopt_shard_create(accbpg_dopt* local, int64_t n, int world, int rank, const void* unique_id,
                                        void* comm, accbpg_dopt_shard** out) {
    if (!local || !out || world <= 0 || rank < 0 || rank >= world || n <= 0) return ACCBPG_ERR_ARG;
    if (!comm && !unique_id) return ACCBPG_ERR_ARG;
    Rccl* R;
    ACC_TRY(need_rccl(&R));
    int64_t lo, hi;
    ACC_TRY(accbpg_dopt_shard_bounds(n, world, rank, &lo, &hi));
    if (local->n != hi - lo) {
        set_last_error("accbpg_dopt_shard_create: rank %d of %d holds columns [%lld, %lld) of %lld, the handle has %lld",
                       rank, world, (long long)lo, (long long)hi, (long long)n, (long long)local->n);
        return ACCBPG_ERR_ARG;
    }
    ACC_HIP(hipSetDevice(local->device));
    accbpg_dopt_shard* s = new accbpg_dopt_shard();
    s->local = local; s->world = world; s->rank = rank; s->n = n; s->lo = lo; s->hi = hi;
    const int64_t m = local->m;
    s->tri = m * (m + 1) / 2;
    s->msg_len = s->tri + 2 - (s->tri & 1);
    s->piece = (n + world - 1) / world;
    auto fail = [&](int rc) { accbpg_dopt_shard_destroy(s); return rc; };
    if (hipMalloc(&s->gram, sizeof(double) * (size_t)m * m) != hipSuccess ||
        hipMalloc(&s->msg, sizeof(double) * (size_t)s->msg_len) != hipSuccess ||
        hipMemset(s->msg, 0, sizeof(double) * (size_t)s->msg_len) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&s->hcount), sizeof(double)) != hipSuccess) {
        set_last_error("accbpg_dopt_shard_create: out of device memory");
        return fail(ACCBPG_ERR_HIP);
    }
    if (n % world != 0 &&
        hipMalloc(&s->gath, sizeof(double) * (size_t)world * s->piece) != hipSuccess) {
        set_last_error("accbpg_dopt_shard_create: out of device memory");
        return fail(ACCBPG_ERR_HIP);
    }
    if (comm) {
        s->comm = static_cast<ncclComm_t>(comm);
        int cw = -1, cr = -1;
        if (R->CommCount(s->comm, &cw) != ncclSuccess || R->CommUserRank(s->comm, &cr) != ncclSuccess ||
            cw != world || cr != rank) {
            set_last_error("accbpg_dopt_shard_create: the communicator is rank %d of %d, the call says %d of %d", cr, cw, rank, world);
            return fail(ACCBPG_ERR_ARG);
        }
    } else {
        ncclUniqueId id;
        memcpy(&id, unique_id, sizeof id);
        ncclResult_t res = R->CommInitRank(&s->comm, world, id, rank);
        if (res != ncclSuccess) {
            set_last_error("ncclCommInitRank -> %s", R->GetErrorString(res));
            s->comm = nullptr;
            return fail(ACCBPG_ERR_HIP);
        }
        s->own_comm = true;
    }
    *out = s;
    return ACCBPG_OK;
}

/* f (flag 0), gradient (flag 1) or both (flag 2) at the length-n device vector x, which every rank holds whole.
 * g_dev receives the whole length-n gradient on every rank.  Return codes as accbpg_dopt_func_grad; a negative entry
 * anywhere in x is reported by every rank together, before anything is factored. */
extern "C" int accbpg_dopt_shard_func_grad(accbpg_dopt_shard* s, const double* x_dev, int flag, double* f_host,
                                           double* g_dev) {
    if (!s || !x_dev || flag < 0 || flag > 2) return ACCBPG_ERR_ARG;
    if (flag != 0 && !g_dev) return ACCBPG_ERR_ARG;
    Rccl* R;
    ACC_TRY(need_rccl(&R));
    accbpg_dopt* h = s->local;
    ACC_HIP(hipSetDevice(h->device));
    const int64_t m = h->m, nloc = s->hi - s->lo;
    ACC_TRY(accbpg_dopt_gram(h, x_dev + s->lo, s->gram));
    ACC_TRY(accbpg_tri_pack(s->gram, m, s->msg, h->stream));
    ACC_TRY(accbpg_vec_count_bad(x_dev + s->lo, nloc, s->msg + s->tri, h->stream));
    ACC_RCCL(R->AllReduce(s->msg, s->msg, (size_t)s->msg_len, ncclDouble, ncclSum, s->comm, h->stream));   // the one Gram all-reduce
    ACC_HIP(hipMemcpyAsync(s->hcount, s->msg + s->tri, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    ACC_TRY(accbpg_tri_unpack(s->msg, m, s->gram, h->stream));
    ACC_HIP(hipStreamSynchronize(h->stream));
    if (*s->hcount != 0.0) {                                   // accbpg/functions.py:45 on the whole vector
        set_last_error("DOptimalObj: x needs to be nonnegative");
        return ACCBPG_ERR_ASSERT;
    }
    ACC_TRY(accbpg_dopt_factor(h, s->gram, f_host));           // replicated Cholesky + log det
    if (flag == 0) return ACCBPG_OK;
    if (s->gath == nullptr) {                                  // equal slices: gathered in place
        ACC_TRY(accbpg_dopt_grad(h, g_dev + s->lo));
        ACC_RCCL(R->AllGather(g_dev + s->lo, g_dev, (size_t)nloc, ncclDouble, s->comm, h->stream));
    } else {
        double* mine = s->gath + (size_t)s->rank * s->piece;
        if (nloc < s->piece) ACC_HIP(hipMemsetAsync(mine + nloc, 0, sizeof(double) * (size_t)(s->piece - nloc), h->stream));
        ACC_TRY(accbpg_dopt_grad(h, mine));
        ACC_RCCL(R->AllGather(mine, s->gath, (size_t)s->piece, ncclDouble, s->comm, h->stream));
        for (int r = 0; r < s->world; ++r) {
            int64_t lo, hi;
            ACC_TRY(accbpg_dopt_shard_bounds(s->n, s->world, r, &lo, &hi));
            ACC_HIP(hipMemcpyAsync(g_dev + lo, s->gath + (size_t)r * s->piece, sizeof(double) * (size_t)(hi - lo),
                                   hipMemcpyDeviceToDevice, h->stream));
        }
    }
    return ACCBPG_OK;
}

/* Test hook: gather the gradient slices through the padded staging buffer (the path of slice lengths that differ
 * between ranks) with `extra` more entries per rank than the longest slice needs, whatever n and the world size are. */
extern "C" int accbpg_debug_shard_pad(accbpg_dopt_shard* s, int64_t extra) {
    if (!s || extra < 0) return ACCBPG_ERR_ARG;
    ACC_HIP(hipSetDevice(s->local->device));
    ACC_HIP(hipStreamSynchronize(s->local->stream));
    hipFree(s->gath);
    s->gath = nullptr;
    s->piece = (s->n + s->world - 1) / s->world + extra;
    ACC_HIP(hipMalloc(&s->gath, sizeof(double) * (size_t)s->world * s->piece));
    return ACCBPG_OK;
}
