// shard.hip — ONE conjugate-gradient solve over several GPUs, inside the library (include/elph_gpu.h: elph_shard_*).
//
// Decomposition (the north_star's): slabs of rows of cells along the slowest spatial index, one process per GPU.
// The caller creates the rank's handle on its SLAB lattice = own rows + the ghost rows the fused M^T M needs (the dependency
// closure of the checkerboard, elphdynamics_amd/sharded.py: SpatialSlabs) with the bonds, exp(-dtau V) and — bond-phonon
// models — the per-(tau, bond) cosh/sinh tables of that slab (bonds sharded by owner, SSHModels.jl:581-701).
//
// Transport: no collective and no host in the iteration.  Every rank owns a MAILBOX in its device memory (uncached
// fine-grained allocation), exports it with hipIpcGetMemHandle and maps every other rank's with hipIpcOpenMemHandle; the
// resident CG kernel (cg_wg.hip, SHARD form) stores its partial sums and the checkerboard boundary rows of the residual
// straight into the neighbours' mailboxes (xGMI peer stores between GPUs) as self-tagged 8-byte granules and polls its own.
// The only host-side step is the caller's barrier between elph_shard_prepare (mailbox zeroed) and elph_shard_solve — the
// 64-byte IPC handles and that barrier travel by whatever the host language has (torch.distributed / MPI.jl).
// RCCL is not used on this path on purpose: its smallest collective costs more than a whole iteration at these sizes
// (DESIGN.md §6), and it refuses two ranks on one GPU — the only multi-rank set-up the test box offers.

#include <cstring>

#include "elph_internal.h"

struct ShardState {
    ElphShardCtl ctl;
    unsigned long long *mail = nullptr;       // own mailbox
    size_t mail_bytes = 0;
    void *opened[ELPH_SHARD_MAXRANKS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool connected = false, prepared = false;
    int G = 0;
};

static size_t mailbox_words(int64_t L, int cap) { return 2 * (size_t)ELPH_SHARD_MAXREC * 2 + 2 * (size_t)L * (size_t)cap * 2; }

void elph_shard_free(elph_handle_s *h) {
    ShardState *S = static_cast<ShardState *>(h->shard);
    if (!S) return;
    for (int q = 0; q < ELPH_SHARD_MAXRANKS; ++q) if (S->opened[q]) (void)hipIpcCloseMemHandle(S->opened[q]);
    if (S->mail) (void)hipFree(S->mail);
    delete S;
    h->shard = nullptr;
}

extern "C" int elph_shard_create(elph_handle h, int rank, int world, int64_t own_lo, int64_t own_n, int64_t n_to_prev,
                                 int64_t n_to_next, int64_t cap_ghost, void *ipc_handle_out) {
    if (!h) { elph_set_error("null handle"); return ELPH_E_ARG; }
    HIPCHK(hipSetDevice(h->device));
    if (world < 1 || world > ELPH_SHARD_MAXRANKS || rank < 0 || rank >= world) { elph_set_error("bad rank %d of %d (at most %d ranks)", rank, world, ELPH_SHARD_MAXRANKS); return ELPH_E_ARG; }
    if (own_lo < 0 || own_n < 1 || own_lo + own_n > h->N || n_to_prev < 0 || n_to_next < 0 || n_to_prev > own_n || n_to_next > own_n ||
        cap_ghost < 0 || own_lo > cap_ghost || h->N - (own_lo + own_n) > cap_ghost || n_to_prev > cap_ghost || n_to_next > cap_ghost || !ipc_handle_out) {
        elph_set_error("bad shard geometry: own [%lld, +%lld) of %lld sites, sends %lld / %lld, ghost capacity %lld", (long long)own_lo,
                       (long long)own_n, (long long)h->N, (long long)n_to_prev, (long long)n_to_next, (long long)cap_ghost);
        return ELPH_E_ARG;
    }
    if (world == 1 && (own_lo != 0 || own_n != h->N)) { elph_set_error("one rank owns the whole lattice"); return ELPH_E_ARG; }
    elph_shard_free(h);
    ShardState *S = new ShardState();
    h->shard = S;
    S->ctl.rank = rank; S->ctl.P = world;
    S->ctl.own_lo = (int)own_lo; S->ctl.own_hi = (int)(own_lo + own_n);
    S->ctl.n_to_prev = (int)n_to_prev; S->ctl.n_to_next = (int)n_to_next;
    S->ctl.cap_ghost = (int)cap_ghost;
    S->mail_bytes = mailbox_words(h->L, (int)cap_ghost) * sizeof(unsigned long long);
    // uncached, fine-grained: stores from another GPU become visible to this GPU's (system-scope) polls while its kernel runs
    HIPCHK(hipExtMallocWithFlags((void **)&S->mail, S->mail_bytes, hipDeviceMallocUncached));
    HIPCHK(hipMemset(S->mail, 0, S->mail_bytes));
    HIPCHK(hipDeviceSynchronize());
    hipIpcMemHandle_t mh;
    HIPCHK(hipIpcGetMemHandle(&mh, S->mail));
    static_assert(sizeof(hipIpcMemHandle_t) == ELPH_SHARD_IPC_BYTES, "IPC handle size");
    memcpy(ipc_handle_out, &mh, sizeof(mh));
    S->ctl.mail[rank] = S->mail;
    return ELPH_OK;
}

extern "C" int elph_shard_connect(elph_handle h, const void *all_ipc_handles) {
    if (!h || !h->shard) { elph_set_error("elph_shard_create has not been called"); return ELPH_E_STATE; }
    HIPCHK(hipSetDevice(h->device));
    ShardState *S = static_cast<ShardState *>(h->shard);
    if (!all_ipc_handles) { elph_set_error("null argument"); return ELPH_E_ARG; }
    const char *p = static_cast<const char *>(all_ipc_handles);
    for (int q = 0; q < S->ctl.P; ++q) {
        if (q == S->ctl.rank) continue;
        hipIpcMemHandle_t mh;
        memcpy(&mh, p + (size_t)q * ELPH_SHARD_IPC_BYTES, sizeof(mh));
        void *ptr = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&ptr, mh, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { elph_set_error("hipIpcOpenMemHandle(rank %d): %s", q, hipGetErrorString(e)); return ELPH_E_HIP; }
        S->opened[q] = ptr;
        S->ctl.mail[q] = static_cast<unsigned long long *>(ptr);
    }
    S->connected = true;
    return ELPH_OK;
}

// zero the own mailbox; the CALLER then synchronises all ranks (barrier) before any of them calls elph_shard_solve
extern "C" int elph_shard_prepare(elph_handle h) {
    if (!h || !h->shard) { elph_set_error("elph_shard_create has not been called"); return ELPH_E_STATE; }
    HIPCHK(hipSetDevice(h->device));
    ShardState *S = static_cast<ShardState *>(h->shard);
    if (!S->connected && S->ctl.P > 1) { elph_set_error("elph_shard_connect has not been called"); return ELPH_E_STATE; }
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemsetAsync(S->mail, 0, S->mail_bytes, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    S->prepared = true;
    return ELPH_OK;
}

static int shard_run(elph_handle_s *h, const double *b_slab, double tol, int64_t maxiter, double kappa_max, long long fixed_iters,
                     double *ms_out) {
    ShardState *S = static_cast<ShardState *>(h->shard);
    if (!S || !S->prepared) { elph_set_error("elph_shard_prepare (and the caller's barrier) must precede every sharded solve"); return ELPH_E_STATE; }
    S->prepared = false;
    if (!h->have_E) { elph_set_error("update_model has not been called on this handle"); return ELPH_E_STATE; }
    int rc = elph_i_ensure_capacity(h, 1);
    if (rc) return rc;
    CgParams P;
    P.tol = tol; P.kmax = (kappa_max > 0.0) ? kappa_max : h->kmax; P.maxiter = maxiter; P.use_prec = 0; P.record_hist = 0; P.hist_stride = 0;
    h->cur_params = P;
    const size_t bytes = (size_t)h->ndim * sizeof(double);
    if (b_slab) {
        HIPCHK(hipMemcpyAsync(h->d_stage_in, b_slab, bytes, hipMemcpyHostToDevice, h->stream));
        rc = elph_launch_r2s(h, h->d_b, h->d_stage_in, 1);
        if (rc) return rc;
    }
    // x0 = 0, r0 = p0 = b (IterativeSolvers.jl:259-274 with a zero initial guess, as every caller passes: HMC.jl:854)
    HIPCHK(hipMemsetAsync(h->d_x, 0, bytes, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_r, h->d_b, bytes, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_p, h->d_b, bytes, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemsetAsync(h->d_state, 0, 2 * sizeof(CgState), h->stream));
    CgBufs B = elph_make_bufs(h, 1);
    B.params = P;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ms_out) { HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e0, h->stream)); }
    rc = elph_wg_cg_shard(h, B, fixed_iters, S->ctl, &S->G);
    if (rc == ELPH_OK && ms_out) {
        hipError_t er = hipEventRecord(e1, h->stream);
        if (er == hipSuccess) er = hipEventSynchronize(e1);
        float ms = 0.f;
        if (er == hipSuccess) er = hipEventElapsedTime(&ms, e0, e1);
        if (er != hipSuccess) { elph_set_error("sharded solve: %s", hipGetErrorString(er)); rc = ELPH_E_HIP; }
        *ms_out = (double)ms;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h->h_state, h->d_state, sizeof(CgState) * 2, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    bool aborted = false;
    rc = elph_wg_aborted(h, &aborted);
    if (rc) return rc;
    if (aborted) return ELPH_E_HIP;
    return ELPH_OK;
}

extern "C" int elph_shard_solve(elph_handle h, double *x_slab, const double *b_slab, double tol, int64_t maxiter, double kappa_max,
                                int64_t *iters, int *done, double *eps) {
    if (!h) { elph_set_error("null handle"); return ELPH_E_ARG; }
    HIPCHK(hipSetDevice(h->device));
    if (!x_slab || !b_slab || !(tol >= 0.0) || maxiter < 1) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    int rc = shard_run(h, b_slab, tol, maxiter, kappa_max, 0, nullptr);
    if (rc) return rc;
    const CgState &s = h->h_state[0];
    if (!s.done) { elph_set_error("sharded CG ended without a terminal state (internal error)"); return ELPH_E_STATE; }
    if (iters) *iters = s.iters;
    if (done) *done = s.done;
    if (eps) *eps = s.eps;
    rc = elph_launch_s2r(h, h->d_stage_out, h->d_x, 1);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(x_slab, h->d_stage_out, (size_t)h->ndim * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}

// measurement: exactly `iters` iterations of the sharded solve (no stop test) on the right-hand side of the last solve / b_slab;
// *ms = HIP-event time of the launch on this rank
extern "C" int elph_shard_iterate(elph_handle h, const double *b_slab, int64_t iters, double *ms) {
    if (!h) { elph_set_error("null handle"); return ELPH_E_ARG; }
    HIPCHK(hipSetDevice(h->device));
    if (iters < 1 || !ms) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    return shard_run(h, b_slab, 0.0, (int64_t)1 << 40, 1e300, iters, ms);
}

extern "C" int elph_shard_destroy(elph_handle h) {
    if (!h) return ELPH_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    elph_shard_free(h);
    return ELPH_OK;
}
