// Batches of same-shaped D-optimal instances on one GPU (BASELINE config 4: many D_opt_design(512,8192) problems per
// device; SURVEY.md 8(b) "*_batched", 8(e).1).  A small instance cannot fill the chip and its evaluation is a chain of
// short launches, so the instances of a batch advance in lock-step: ONE launch per kernel family covers all active
// instances (blockIdx.y = position among them), one readback returns every instance's value and status.  The host
// keeps the per-instance decisions (stopping, line search) and passes the set of instances that take part in a call.
#include "internal.h"

using namespace accbpg;

extern "C" int accbpg_dopt_batch_destroy(accbpg_dopt_batch* b) {
    if (!b) return ACCBPG_OK;
    for (accbpg_dopt* h : b->inst) accbpg_dopt_destroy(h);
    hipFree(b->table); hipFree(b->chol_table[0]); hipFree(b->chol_table[1]); hipFree(b->ops_all); hipFree(b->red_all);
    hipFree(b->dscal_all); hipFree(b->vflags); hipFree(b->vout); hipFree(b->vpart); hipFree(b->vgg);
    if (b->hpin) hipHostFree(b->hpin);
    if (b->vpin) hipHostFree(b->vpin);
    delete b;
    return ACCBPG_OK;
}

static int batch_init(accbpg_dopt_batch* b, const double* const* V_host, int K, int64_t m, int64_t n, int64_t ldv) {
    ACC_HIP(hipGetDevice(&b->device));
    hipDeviceProp_t prop;
    ACC_HIP(hipGetDeviceProperties(&prop, b->device));
    ACC_HIP(hipMalloc(&b->dscal_all, sizeof(double) * 24 * (size_t)K));
    ACC_HIP(hipMemset(b->dscal_all, 0, sizeof(double) * 24 * (size_t)K));
    ACC_HIP(hipHostMalloc(&b->hpin, sizeof(double) * 24 * (size_t)K, hipHostMallocDefault));
    // instances that share a launch: as many as have their one-launch factorisations resident together.  The first
    // guess (T(T+1)/2 - (T-1) workgroups each, two per CU) sizes the per-instance Gram grids; it is replaced below by
    // what the plan of instance 0 says (its grid, and the slots the occupancy query reports).
    const int T = (int)((m + NB - 1) / NB);
    const int per_inst = std::max(1, T * (T + 1) / 2 - (T - 1));
    b->chunk = std::max(1, std::min(K, (2 * prop.multiProcessorCount) / per_inst));
    for (int i = 0; i < K; ++i) {
        accbpg_dopt* h = new accbpg_dopt();
        b->inst.push_back(h);                                   // (destroyed with the batch from here on)
        h->V = V_host[i]; h->m = m; h->n = n; h->ldv = ldv; h->stream = b->stream;
        h->force_big = true;                                    // the tuned 256 x 128 tile also at m = 512
        h->gram_grid_cap = std::max(1, prop.multiProcessorCount / b->chunk);   // the instances of a launch share the chip
        h->dscal_ext = b->dscal_all + 24 * (size_t)i;
        ACC_TRY(dopt_init(h));
    }
    accbpg_dopt* h0 = b->inst[0];
    if (h0->chol_tiles_ok && h0->chol_tiles_grid > 0)
        b->chunk = std::max(1, std::min(b->chunk, h0->chol_slots / h0->chol_tiles_grid));
    // one launch per kernel family needs: the interior big-tile path, the one-launch Cholesky for every instance at
    // once, identical plans (same shape and alignment give identical plans)
    bool fast = h0->big && h0->use_glds && (m % 256 == 0) && (n % 128 == 0) && h0->chol_tiles_ok && K <= BATCH_MAX;
    for (accbpg_dopt* h : b->inst) fast = fast && h->vec_ok && h->big && h->chol_tiles_ok;
    b->fast = fast;
    if (!fast) return ACCBPG_OK;
    std::vector<BatchInst> tab(K);
    std::vector<CholInst> c0(K), c1(K);
    std::vector<GemmOp> ops;
    std::vector<RedOp> reds;
    b->ops_per_inst = (int)h0->ops_host.size();
    b->red_per_inst = (int)h0->red_host.size();
    for (int i = 0; i < K; ++i) {
        accbpg_dopt* h = b->inst[i];
        if ((int)h->ops_host.size() != b->ops_per_inst || (int)h->red_host.size() != b->red_per_inst ||
            h->gram_grid != h0->gram_grid || h->ntiles != h0->ntiles || h->gram_nslot != h0->gram_nslot) {
            b->fast = false;
            return ACCBPG_OK;
        }
        tab[i] = BatchInst{h->V, h->slabs, h->Gbuf, h->Lbuf, h->Wbuf, h->Tbuf, h->dscal, h->dflag, h->chol_ready};
        CholInst ci{};
        ci.src = h->Gbuf; ci.L = h->Lbuf; ci.Ldiag = h->Tbuf; ci.Winv = nullptr; ci.logdet = h->dscal; ci.flags = h->dflag;
        ci.ready = h->chol_ready; ci.aux = h->chol_aux; ci.hand = h->chol_hand; ci.trace = nullptr;
        c0[i] = ci;
        ci.Winv = h->Wbuf;
        c1[i] = ci;
        ops.insert(ops.end(), h->ops_host.begin(), h->ops_host.end());
        reds.insert(reds.end(), h->red_host.begin(), h->red_host.end());
    }
    ACC_HIP(hipMalloc(&b->table, sizeof(BatchInst) * K));
    ACC_HIP(hipMemcpy(b->table, tab.data(), sizeof(BatchInst) * K, hipMemcpyHostToDevice));
    ACC_HIP(hipMalloc(&b->chol_table[0], sizeof(CholInst) * K));
    ACC_HIP(hipMemcpy(b->chol_table[0], c0.data(), sizeof(CholInst) * K, hipMemcpyHostToDevice));
    ACC_HIP(hipMalloc(&b->chol_table[1], sizeof(CholInst) * K));
    ACC_HIP(hipMemcpy(b->chol_table[1], c1.data(), sizeof(CholInst) * K, hipMemcpyHostToDevice));
    if (!ops.empty()) {
        ACC_HIP(hipMalloc(&b->ops_all, sizeof(GemmOp) * ops.size()));
        ACC_HIP(hipMemcpy(b->ops_all, ops.data(), sizeof(GemmOp) * ops.size(), hipMemcpyHostToDevice));
    }
    if (!reds.empty()) {
        ACC_HIP(hipMalloc(&b->red_all, sizeof(RedOp) * reds.size()));
        ACC_HIP(hipMemcpy(b->red_all, reds.data(), sizeof(RedOp) * reds.size(), hipMemcpyHostToDevice));
    }
    return ACCBPG_OK;
}

extern "C" int accbpg_dopt_batch_create(const double* const* V_dev_host, int K, int64_t m, int64_t n, int64_t ldv,
                                        void* stream, accbpg_dopt_batch** out) {
    if (!V_dev_host || !out || K <= 0 || m <= 0 || n <= 0 || ldv < n) {
        set_last_error("accbpg_dopt_batch_create: bad arguments (K=%d m=%lld n=%lld ldv=%lld)", K, (long long)m,
                       (long long)n, (long long)ldv);
        return ACCBPG_ERR_ARG;
    }
    if (K > BATCH_MAX) {
        // the active set travels as a fixed-size kernel argument (BatchAct) and the length-n batch kernels refuse
        // more: one limit for the whole family, stated here instead of an index past the table later
        set_last_error("accbpg_dopt_batch_create: K=%d instances, at most ACCBPG_BATCH_MAX=%d per batch (use several batches)",
                       K, BATCH_MAX);
        return ACCBPG_ERR_ARG;
    }
    if (!(m < n)) {                                             // DOptimalObj: need m < n   (functions.py:35)
        set_last_error("DOptimalObj: need m < n");
        return ACCBPG_ERR_ASSERT;
    }
    for (int i = 0; i < K; ++i)
        if (!V_dev_host[i]) return ACCBPG_ERR_ARG;
    accbpg_dopt_batch* b = new accbpg_dopt_batch();
    b->K = K;
    b->stream = (hipStream_t)stream;
    const int rc = batch_init(b, V_dev_host, K, m, n, ldv);
    if (rc != ACCBPG_OK) {
        accbpg_dopt_batch_destroy(b);
        return rc;
    }
    *out = b;
    return ACCBPG_OK;
}

extern "C" int accbpg_dopt_batch_size(accbpg_dopt_batch* b) { return b ? b->K : 0; }
extern "C" int accbpg_dopt_batch_is_fused(accbpg_dopt_batch* b) { return (b && b->fast) ? 1 : 0; }
extern "C" int accbpg_dopt_batch_chunk(accbpg_dopt_batch* b) { return b ? (b->fast ? std::min(b->chunk, b->K) : 1) : 0; }

extern "C" accbpg_dopt* accbpg_dopt_batch_instance(accbpg_dopt_batch* b, int i) {
    if (!b || i < 0 || i >= b->K) return nullptr;
    return b->inst[i];
}

extern "C" int accbpg_dopt_batch_set_stream(accbpg_dopt_batch* b, void* stream) {
    if (!b) return ACCBPG_ERR_ARG;
    b->stream = (hipStream_t)stream;
    for (accbpg_dopt* h : b->inst) h->stream = b->stream;
    return ACCBPG_OK;
}

static int status_of(const double* st) {
    const int* fl = reinterpret_cast<const int*>(st + 16);
    if (fl[FLAG_NEG_X]) return ACCBPG_ERR_ASSERT;
    if (fl[FLAG_NOT_PD]) return ACCBPG_ERR_NOT_PD;
    return ACCBPG_OK;
}

/* func_grad for the instances with active_host[i] != 0 (NULL: all).  x_dev / g_dev: row i = instance i, leading
 * dimensions ldx / ldg.  f_host[i] and status_host[i] (ACCBPG_OK, ACCBPG_ERR_ASSERT for min(x_i) < 0, ACCBPG_ERR_NOT_PD)
 * are written for the active instances only; the return value reports failures of the call itself.
 * _begin queues the whole evaluation on the batch's stream and returns, _end waits for it: two batches over the same
 * matrices on different streams let INDEPENDENT evaluations overlap (F[k] = f(x) beside the gradient at y). */
extern "C" int accbpg_dopt_batch_func_grad_begin(accbpg_dopt_batch* b, const double* x_dev, int64_t ldx,
                                                 const int* active_host, int flag, double* g_dev, int64_t ldg) {
    if (!b || !x_dev || flag < 0 || flag > 2 || ldx < b->inst[0]->n) return ACCBPG_ERR_ARG;
    if (flag != 0 && (!g_dev || ldg < b->inst[0]->n)) return ACCBPG_ERR_ARG;
    BatchAct act;
    for (int i = 0; i < b->K; ++i)
        if (!active_host || active_host[i]) {
            if (act.n < BATCH_MAX) act.idx[act.n] = i;
            ++act.n;
        }
    b->pend_act = act; b->pend_x = x_dev; b->pend_ldx = ldx; b->pend_flag = flag; b->pend_g = g_dev; b->pend_ldg = ldg;
    b->pend_fused = false;
    if (act.n == 0) return ACCBPG_OK;
    const bool aligned = ((reinterpret_cast<uintptr_t>(x_dev) & 15) == 0) && ((ldx & 1) == 0);
    if (b->fast && aligned && act.n <= BATCH_MAX) {
        for (int c0 = 0; c0 < act.n; c0 += b->chunk) {          // (one chunk unless the batch is larger than the chip takes at once)
            BatchAct part;
            part.n = std::min(b->chunk, act.n - c0);
            for (int a = 0; a < part.n; ++a) part.idx[a] = act.idx[c0 + a];
            ACC_TRY(launch_gram_batch(b, part, x_dev, ldx));
            ACC_TRY(launch_cholesky_batch(b, part, flag != 0, x_dev, ldx));
            if (flag != 0) {
                ACC_TRY(launch_trtri_batch(b, part));
                ACC_TRY(launch_colnorm_batch(b, part, g_dev, ldg, -1.0));
            }
        }
        ACC_HIP(hipMemcpyAsync(b->hpin, b->dscal_all, sizeof(double) * 24 * (size_t)b->K, hipMemcpyDeviceToHost, b->stream));
        b->pend_fused = true;
    }
    return ACCBPG_OK;
}

extern "C" int accbpg_dopt_batch_func_grad_end(accbpg_dopt_batch* b, double* f_host, int* status_host) {
    if (!b || !status_host) return ACCBPG_ERR_ARG;
    const BatchAct& act = b->pend_act;
    if (act.n == 0) return ACCBPG_OK;
    if (b->pend_fused) {
        ACC_HIP(hipStreamSynchronize(b->stream));
        bool aborted = false;
        for (int a = 0; a < act.n; ++a) {
            const double* st = b->hpin + 24 * (size_t)act.idx[a];
            if (reinterpret_cast<const int*>(st + 16)[FLAG_ABORT]) aborted = true;
        }
        if (!aborted) {
            for (int a = 0; a < act.n; ++a) {
                const int i = act.idx[a];
                const double* st = b->hpin + 24 * (size_t)i;
                status_host[i] = status_of(st);
                if (f_host) f_host[i] = -st[0];                 // f = -logdet   (functions.py:51)
            }
            return ACCBPG_OK;
        }
        // the one-launch Cholesky gave up a wait: evaluate instance by instance with the launch-per-column kernels
        b->fast = false;
        note_tiles_fallback("accbpg_dopt_batch_func_grad");
        for (accbpg_dopt* h : b->inst) h->chol_tiles_off = true;
    }
    for (int a = 0; a < act.n; ++a) {
        const int i = act.idx[a];
        accbpg_dopt* h = b->inst[i];
        h->stream = b->stream;
        double fv = 0.0;
        const int rc = accbpg_dopt_func_grad(h, b->pend_x + (size_t)i * b->pend_ldx, b->pend_flag, &fv,
                                             b->pend_flag != 0 ? b->pend_g + (size_t)i * b->pend_ldg : nullptr);
        if (rc == ACCBPG_ERR_HIP || rc == ACCBPG_ERR_ARG) return rc;
        status_host[i] = rc;
        if (f_host) f_host[i] = fv;
    }
    return ACCBPG_OK;
}

extern "C" int accbpg_dopt_batch_func_grad(accbpg_dopt_batch* b, const double* x_dev, int64_t ldx, const int* active_host,
                                           int flag, double* f_host, double* g_dev, int64_t ldg, int* status_host) {
    if (flag != 1 && !f_host) return ACCBPG_ERR_ARG;
    if (!status_host) return ACCBPG_ERR_ARG;
    ACC_TRY(accbpg_dopt_batch_func_grad_begin(b, x_dev, ldx, active_host, flag, g_dev, ldg));
    return accbpg_dopt_batch_func_grad_end(b, f_host, status_host);
}
