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

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "elph_internal.h"

struct ShardState {
    ElphShardCtl ctl;
    unsigned long long *mail = nullptr;       // own mailbox
    size_t mail_bytes = 0;
    void *opened[ELPH_SHARD_MAXRANKS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool connected = false, prepared = false;
    int G = 0;
    // preconditioned solve (streaming form): global sites of the slab's sites, the spectrum area of the mailbox, scratch
    int64_t n_global = 0, own_gstart = 0;
    size_t nu_off = 0, ext_off = 0;           // u64 word offsets into the mailbox: spectrum [Lo2][n_global] complex; records + flags of the streaming form
    int *d_gsites = nullptr;                  // [N_loc] global site of every slab site
    int *d_counter = nullptr, *d_abort = nullptr;
    unsigned epoch = 0;
    unsigned selftest_calls = 0;
    // the sharded CALLERS (elph_shard_ldiv, elph_shard_fermion_force_*, elph_hmc_update on a sharded handle): what they need from the host
    elph_shard_barrier_fn barrier = nullptr;
    elph_shard_allreduce_fn allreduce = nullptr;
    void *coll_ctx = nullptr;
    std::vector<int> gsites_host;             // [N_loc] global site of every slab site (host copy of d_gsites)
    size_t xch_words = 0;                     // words of the spectrum area = the exchange area of the ghost-row pushes between solves
    int *d_xcol = nullptr;                    // ghost exchange of vectors whose columns are not sites: global column, owner weight (device copies)
    double *d_xown = nullptr;
    size_t xcol_cap = 0;
    std::vector<int> gbond;                   // bond phonons: global bond (checkerboard position on the whole lattice) of every slab bond, owner weight
    std::vector<double> bown;
    int64_t n_gbonds = 0;
    double *d_csbar = nullptr;                // [2][slab bonds]: tau-means of the slab's hopping tables
    uint64_t ghost_dev = 0, ghost_host = 0;   // ghost exchanges that went through the mailboxes / were staged through the host collectives
    elph_handle_s *full = nullptr;            // the full-lattice handle of the preconditioned callers (elph_shard_set_full_lattice; not owned)
};

// mailbox layout (u64 words), identical on all ranks:
//   [2][ELPH_SHARD_MAXREC][2]   records of the resident kernel's two meetings
//   [2][2][Ltau][cap][2]        ghost rows from below / from above, once per parity of the iteration
//   [8][8][2] + [8]             streaming form: records of up to 8 named all-sums (one 2-granule record per rank) + spectrum flags
//   [Lo2][n_global][2]          spectrum nu of the whole lattice (16-byte complex), written by all ranks (KPM apply) — and, BETWEEN solves,
//                               the exchange area of the callers' ghost rows (elph_i_shard_ghost_sync: up to four vectors of the whole
//                               lattice, [vector][tau][global column] doubles), hence at least 4 L n_global words
static size_t mailbox_words(int64_t L, int cap, int64_t n_global, size_t *ext_off, size_t *nu_off, size_t *xch_words = nullptr) {
    size_t w = 2 * (size_t)ELPH_SHARD_MAXREC * 2 + 2 * 2 * (size_t)L * (size_t)cap * 2;      // (ghost regions twice: by the parity of the iteration)
    if (ext_off) *ext_off = w;
    w += 8 * 8 * 2 + 8;
    if (nu_off) *nu_off = w;
    const size_t area = std::max((size_t)((L + 1) / 2) * (size_t)n_global * 2, (size_t)4 * (size_t)L * (size_t)n_global);
    if (xch_words) *xch_words = area;
    w += area;
    return w;
}

// Mailboxes created by THIS process, keyed by their IPC handle: a host that drives several ranks from one process (one thread
// per GPU — or the test box's 8 ranks in fewer processes than the box admits on its card) hands elph_shard_connect the same
// all-gathered handle list as everybody else; a handle found here is mapped by its device pointer (hipIpcOpenMemHandle refuses
// memory of the calling process), with peer access enabled when the mailbox lives on another device.
struct LocalMailbox { unsigned char key[ELPH_SHARD_IPC_BYTES]; unsigned long long *ptr; int device; };
static std::mutex g_local_mu;
static std::vector<LocalMailbox> g_local_mail;

static void local_mail_register(const void *key, unsigned long long *ptr, int device) {
    LocalMailbox m;
    memcpy(m.key, key, ELPH_SHARD_IPC_BYTES);
    m.ptr = ptr; m.device = device;
    std::lock_guard<std::mutex> lk(g_local_mu);
    g_local_mail.push_back(m);
}

static void local_mail_forget(const unsigned long long *ptr) {
    std::lock_guard<std::mutex> lk(g_local_mu);
    for (size_t i = 0; i < g_local_mail.size(); ++i)
        if (g_local_mail[i].ptr == ptr) { g_local_mail.erase(g_local_mail.begin() + (long)i); return; }
}

static bool local_mail_find(const void *key, unsigned long long **ptr, int *device) {
    std::lock_guard<std::mutex> lk(g_local_mu);
    for (const LocalMailbox &m : g_local_mail)
        if (memcmp(m.key, key, ELPH_SHARD_IPC_BYTES) == 0) { *ptr = m.ptr; *device = m.device; return true; }
    return false;
}

void elph_shard_free(elph_handle_s *h) {
    ShardState *S = static_cast<ShardState *>(h->shard);
    if (!S) return;
    if (S->mail) local_mail_forget(S->mail);
    for (int q = 0; q < ELPH_SHARD_MAXRANKS; ++q) if (S->opened[q]) (void)hipIpcCloseMemHandle(S->opened[q]);
    if (S->mail) (void)hipFree(S->mail);
    if (S->d_gsites) (void)hipFree(S->d_gsites);
    if (S->d_counter) (void)hipFree(S->d_counter);
    if (S->d_xcol) (void)hipFree(S->d_xcol);
    if (S->d_xown) (void)hipFree(S->d_xown);
    if (S->d_csbar) (void)hipFree(S->d_csbar);
    delete S;
    h->shard = nullptr;
}

// Team shape of the sharded resident kernel and whether its meetings hold the records of `world` ranks — host arithmetic only (no
// device needed): one slice per wave, W = the largest divisor of Ltau that is <= 8 waves, G = Ltau / W workgroups per rank.
extern "C" int elph_shard_shape(int64_t ltau, int world, int *waves, int *groups, int *records, int *max_records) {
    if (ltau < 1 || world < 1 || world > ELPH_SHARD_MAXRANKS) { elph_set_error("bad argument: ltau %lld, world %d (at most %d ranks)", (long long)ltau, world, ELPH_SHARD_MAXRANKS); return ELPH_E_ARG; }
    int W = 0;
    for (int w = (int)std::min<int64_t>(8, ltau); w >= 1; --w) if (ltau % w == 0) { W = w; break; }
    const int64_t G = ltau / W;
    if (waves) *waves = W;
    if (groups) *groups = (int)G;
    if (records) *records = (int)(world * G);
    if (max_records) *max_records = ELPH_SHARD_MAXREC;
    if (world * G > ELPH_SHARD_MAXREC || (G > 1 && W < 2)) {
        elph_set_error("sharded solve: %d ranks x %lld workgroups exceed the %d records of a meeting", world, (long long)G, ELPH_SHARD_MAXREC);
        return ELPH_E_UNSUPPORTED;
    }
    return ELPH_OK;
}

// local_only (slabs.hip): ALL ranks of this solve are slabs on this handle's device and run as one launch (k_cg_wg<..., RANKS>, agent-scope mailbox
// accesses) — the mailbox is ordinary device memory, never exported (the fine-grained uncached allocation and its IPC handle are what make a
// mailbox visible to another GPU / process; an exported allocation freed early was also seen to stay reserved: 2 MB per slab and model),
// and the "handle" handed back is a process-local key for elph_shard_connect's registry.
static int shard_create(elph_handle_s *h, int rank, int world, int64_t own_lo, int64_t own_n, int64_t n_to_prev, int64_t n_to_next,
                        int64_t cap_ghost, int64_t n_global, int64_t own_global_start, const int64_t *global_sites, void *ipc_handle_out,
                        bool local_only);

extern "C" int elph_shard_create(elph_handle h, int rank, int world, int64_t own_lo, int64_t own_n, int64_t n_to_prev,
                                 int64_t n_to_next, int64_t cap_ghost, int64_t n_global, int64_t own_global_start,
                                 const int64_t *global_sites, void *ipc_handle_out) {
    return shard_create(h, rank, world, own_lo, own_n, n_to_prev, n_to_next, cap_ghost, n_global, own_global_start, global_sites, ipc_handle_out, false);
}

int elph_i_shard_create_local(elph_handle_s *h, int rank, int world, int64_t own_lo, int64_t own_n, int64_t n_to_prev, int64_t n_to_next,
                              int64_t cap_ghost, void *key_out) {
    const int64_t zero = 0;
    return shard_create(h, rank, world, own_lo, own_n, n_to_prev, n_to_next, cap_ghost, 0, 0, &zero, key_out, true);
}

static int shard_create(elph_handle_s *h, int rank, int world, int64_t own_lo, int64_t own_n, int64_t n_to_prev, int64_t n_to_next,
                        int64_t cap_ghost, int64_t n_global, int64_t own_global_start, const int64_t *global_sites, void *ipc_handle_out,
                        bool local_only) {
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
    if (n_global < 0 || (n_global > 0 && (!global_sites || own_global_start < 0 || own_global_start + own_n > n_global))) {
        elph_set_error("bad global geometry: %lld sites, own rows start at %lld", (long long)n_global, (long long)own_global_start);
        return ELPH_E_ARG;
    }
    S->n_global = n_global; S->own_gstart = own_global_start;
    S->mail_bytes = mailbox_words(h->L, (int)cap_ghost, n_global, &S->ext_off, &S->nu_off, &S->xch_words) * sizeof(unsigned long long);
    if (n_global > 0) {
        std::vector<int> gs((size_t)h->N);
        for (int64_t i = 0; i < h->N; ++i) {
            if (global_sites[i] < 0 || global_sites[i] >= n_global) { elph_set_error("global_sites[%lld] out of range", (long long)i); return ELPH_E_ARG; }
            gs[(size_t)i] = (int)global_sites[i];
        }
        S->gsites_host = gs;
        HIPCHK(hipMalloc((void **)&S->d_gsites, gs.size() * sizeof(int)));
        HIPCHK(hipMemcpy(S->d_gsites, gs.data(), gs.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    HIPCHK(hipMalloc((void **)&S->d_counter, 64));
    HIPCHK(hipMemset(S->d_counter, 0, 64));
    S->d_abort = S->d_counter + 8;
    static_assert(sizeof(hipIpcMemHandle_t) == ELPH_SHARD_IPC_BYTES, "IPC handle size");
    if (local_only) {
        HIPCHK(hipMalloc((void **)&S->mail, S->mail_bytes));
        HIPCHK(hipMemset(S->mail, 0, S->mail_bytes));
        unsigned char key[ELPH_SHARD_IPC_BYTES];
        memset(key, 0, sizeof(key));
        memcpy(key, "ELPHLOCL", 8);
        const unsigned long long pv = (unsigned long long)(uintptr_t)S->mail;
        memcpy(key + 8, &pv, sizeof(pv));
        memcpy(ipc_handle_out, key, sizeof(key));
        local_mail_register(key, S->mail, h->device);
    } else {
        // uncached, fine-grained: stores from another GPU become visible to this GPU's (system-scope) polls while its kernel runs
        HIPCHK(hipExtMallocWithFlags((void **)&S->mail, S->mail_bytes, hipDeviceMallocUncached));
        HIPCHK(hipMemset(S->mail, 0, S->mail_bytes));
        HIPCHK(hipDeviceSynchronize());
        hipIpcMemHandle_t mh;
        HIPCHK(hipIpcGetMemHandle(&mh, S->mail));
        memcpy(ipc_handle_out, &mh, sizeof(mh));
        local_mail_register(&mh, S->mail, h->device);
    }
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
        unsigned long long *lp = nullptr;
        int ldev = -1;
        if (local_mail_find(&mh, &lp, &ldev)) {               // rank q lives in this process
            if (ldev != h->device) {
                hipError_t pe = hipDeviceEnablePeerAccess(ldev, 0);
                if (pe == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                else if (pe != hipSuccess) { elph_set_error("hipDeviceEnablePeerAccess(device %d -> %d): %s", h->device, ldev, hipGetErrorString(pe)); return ELPH_E_HIP; }
            }
            S->ctl.mail[q] = lp;
            if (getenv("ELPH_SHARD_DEBUG")) fprintf(stderr, "[shard] rank %d: rank %d is local (device %d, mailbox %p; own %p)\n", S->ctl.rank, q, ldev, (void *)lp, (void *)S->mail);
            continue;
        }
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

// ALL RANKS OF A SOLVE WHOSE SLABS LIVE ON ONE DEVICE (slabs.hip), one launch: every slab's right-hand side is in its d_b already (layout S);
// the solutions stay in the slabs' d_x.  Everything is queued on the ONE stream the slab handles share — the mailboxes are zeroed, the
// Krylov vectors seeded and the kernel launched in stream order, so no barrier is needed between "prepare" and "solve".
int elph_i_shard_run_ranks(elph_handle_s *const *hs, int P, int nsets, void *h_args, void *d_args, double tol, int64_t maxiter, double kmax,
                           long long fixed_iters, long long timeout_ms, CgState *state_out, double *ms_out, double *const *hist_dev,
                           long long hist_stride) {
    // hist_dev[set] (optional): device array of hist_stride values that takes the eps history of that set's solve (entry 0, eps0, is the caller's)
    // hs[set * P + q]: `nsets` independent solves (each over its own P slabs and their mailboxes) in the one launch
    if (P < 2 || P > ELPH_SHARD_MAXRANKS || nsets < 1 || nsets > 2) { elph_set_error("bad rank count %d x %d", nsets, P); return ELPH_E_ARG; }
    const int PT = P * nsets;
    hipStream_t st = hs[0]->stream;
    std::vector<CgBufs> Bs((size_t)PT);
    std::vector<ElphShardCtl> ctls((size_t)PT);
    CgParams Pm;
    Pm.tol = tol; Pm.kmax = kmax; Pm.maxiter = maxiter; Pm.use_prec = 0; Pm.record_hist = hist_dev ? 1 : 0; Pm.hist_stride = hist_dev ? hist_stride : 0;
    for (int q = 0; q < PT; ++q) {
        elph_handle_s *h = hs[q];
        ShardState *S = static_cast<ShardState *>(h->shard);
        if (!S || !S->connected || h->stream != st) { elph_set_error("slab %d is not connected / runs on another stream", q); return ELPH_E_STATE; }
        if (!h->have_E) { elph_set_error("slab %d has no exp(-dtau V)", q); return ELPH_E_STATE; }
        int rc = elph_i_ensure_capacity(h, 1);
        if (rc) return rc;
        h->cur_params = Pm;
        const size_t bytes = (size_t)h->ndim * sizeof(double);
        HIPCHK(hipMemsetAsync(S->mail, 0, S->mail_bytes, st));
        HIPCHK(hipMemsetAsync(h->d_x, 0, bytes, st));
        HIPCHK(hipMemcpyAsync(h->d_r, h->d_b, bytes, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(h->d_p, h->d_b, bytes, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemsetAsync(h->d_state, 0, 2 * sizeof(CgState), st));
        Bs[(size_t)q] = elph_make_bufs(h, 1);
        Bs[(size_t)q].params = Pm;
        if (hist_dev) Bs[(size_t)q].hist = hist_dev[q / P];      // (workgroup 0 of rank 0 writes; every rank runs the exact stop arithmetic)
        ctls[(size_t)q] = S->ctl;
        S->prepared = false;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ms_out) { HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventRecord(e0, st)); }
    int G = 0;
    int rc = elph_wg_cg_ranks(hs, PT, Bs.data(), fixed_iters, ctls.data(), h_args, d_args, st, timeout_ms, &G);
    if (rc == ELPH_OK && ms_out) {
        hipError_t er = hipEventRecord(e1, st);
        if (er == hipSuccess) er = hipEventSynchronize(e1);
        float ms = 0.f;
        if (er == hipSuccess) er = hipEventElapsedTime(&ms, e0, e1);
        if (er != hipSuccess) { elph_set_error("slab solve: %s", hipGetErrorString(er)); rc = ELPH_E_HIP; }
        *ms_out = (double)ms;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (rc) return rc;
    for (int k = 0; k < nsets; ++k) HIPCHK(hipMemcpyAsync(hs[k * P]->h_state, hs[k * P]->d_state, sizeof(CgState) * 2, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int q = 0; q < PT; ++q) {
        bool aborted = false;
        rc = elph_wg_aborted(hs[q], &aborted);
        if (rc) return rc;
        if (aborted) { hs[q]->wg_broken = false; hs[q]->wg_cooldown = 0; return ELPH_I_ABORTED; }      // (the caller owns the fallback and its cool-down; every OTHER failure keeps its own code)
    }
    if (state_out) for (int k = 0; k < nsets; ++k) state_out[k] = hs[k * P]->h_state[0];
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

// =========================================================================================================================
// KPM-preconditioned solve under sharding (IterativeSolvers.jl:153-234 with P = SymmetricKPMPreconditioner).
// Streaming form: the two-kernel iteration of the slab handle (generic family, inner products over the own rows) with the three
// inner products combined across ranks by a one-wave kernel through the mailbox, and the preconditioner applied as
//     forward tau-DFT of r on the own sites  ->  every rank stores its columns of the spectrum into EVERY rank's mailbox
//     (one all-gather by peer stores, 16 B per site and frequency)  ->  Chebyshev recursion on the WHOLE lattice (a second handle on
//     the full lattice; every rank computes all frequency blocks — the apply is bounded by its longest recursion, which would
//     sit on one rank under any omega-sharding as well, so the redundant work costs no time and the all-to-all back disappears)
//     ->  inverse tau-DFT for the slab's own AND ghost sites straight from the full spectrum: P^-1 r arrives on the ghost rows
//     without an exchange of its own.
// No ghost-row exchange of r is needed at all in this form (r enters only through own-site quantities).
// =========================================================================================================================

typedef unsigned long long u64s;
__device__ __forceinline__ void sst(u64s *p, u64s v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ u64s sld(const u64s *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// part[0 .. n) -> [sum over ALL ranks, 0, 0, ...]: one wave; the local sum is taken in index order, the ranks' sums in rank order,
// so every rank holds the same bits.  slot: which of the 8 named all-sums (its records are single-buffered: between two uses of a
// slot every rank passes another all-sum that needs this rank's record, which it publishes only after it has read this one).
__global__ void __launch_bounds__(64) k_shard_allsum(double *part, int n, ElphShardCtl Sh, size_t ext_off, int slot, unsigned epoch,
                                                     int *abort_flag, long long timeout_ticks) {
    const int lane = threadIdx.x;
    if (*(volatile int *)abort_flag) return;                 // a peer was given up on earlier in this solve: do not wait again
    double a = 0.0;
    for (int i = lane; i < n; i += 64) a += part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (lane < 2 * Sh.P) {
        const u64s bits = (u64s)__double_as_longlong(a);
        const int dest = lane >> 1, half = lane & 1;
        sst(Sh.mail[dest] + ext_off + ((size_t)slot * 8 + Sh.rank) * 2 + half, ((u64s)epoch << 32) | (half ? (bits >> 32) : (bits & 0xFFFFFFFFull)));
    }
    const u64s *rec = Sh.mail[Sh.rank] + ext_off + (size_t)slot * 8 * 2;
    u64s v = 0;
    long long t0 = 0;
    for (int spin = 0;; ++spin) {
        bool ok = true;
        if (lane < 2 * Sh.P) { v = sld(rec + lane); ok = ((unsigned)(v >> 32) == epoch); }
        if (__all(ok)) break;
        if ((spin & 31) == 31) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > timeout_ticks) { if (lane == 0) *abort_flag = 1; return; }
        }
        __builtin_amdgcn_s_sleep(2);
    }
    const int half = (int)(unsigned)v;
    double tot = 0.0;
    for (int q = 0; q < Sh.P; ++q) tot += __hiloint2double(__builtin_amdgcn_readlane(half, 2 * q + 1), __builtin_amdgcn_readlane(half, 2 * q));
    for (int i = lane; i < n; i += 64) part[i] = (i == 0) ? tot : 0.0;
}

// nu_slab[w][s] of the own sites -> nu_full[w][global site] in the mailbox of every rank; the block that finishes last raises this
// rank's flag (epoch) in every mailbox
__global__ void __launch_bounds__(256) k_shard_push_nu(const double2 *__restrict__ nu_slab, ElphShardCtl Sh, int Lo2, int N_loc,
                                                       int own_lo, int own_n, int gstart, int n_global, size_t nu_off, size_t ext_off,
                                                       unsigned epoch, int *counter) {
    const int w = blockIdx.x;
    for (int i = threadIdx.x; i < own_n; i += 256) {
        const double2 v = nu_slab[(size_t)w * N_loc + own_lo + i];
        const u64s re = (u64s)__double_as_longlong(v.x), im = (u64s)__double_as_longlong(v.y);
        for (int q = 0; q < Sh.P; ++q) {
            u64s *d = Sh.mail[q] + nu_off + ((size_t)w * n_global + gstart + i) * 2;
            sst(d, re); sst(d + 1, im);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave drains, then the workgroup's barrier, then ONE signal
    __syncthreads();
    if (threadIdx.x == 0) {
        const int done = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (int)gridDim.x - 1) {
            __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int q = 0; q < Sh.P; ++q) sst(Sh.mail[q] + ext_off + 8 * 8 * 2 + Sh.rank, (u64s)epoch);
        }
    }
}

__global__ void __launch_bounds__(64) k_shard_wait_nu(ElphShardCtl Sh, size_t ext_off, unsigned epoch, int *abort_flag, long long timeout_ticks) {
    const int lane = threadIdx.x;
    if (*(volatile int *)abort_flag) return;
    const u64s *fl = Sh.mail[Sh.rank] + ext_off + 8 * 8 * 2;
    long long t0 = 0;
    for (int spin = 0;; ++spin) {
        bool ok = true;
        if (lane < Sh.P) ok = (sld(fl + lane) == (u64s)epoch);
        if (__all(ok)) return;
        if ((spin & 31) == 31) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > timeout_ticks) { if (lane == 0) *abort_flag = 1; return; }
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// full spectrum out of the (uncached) mailbox into the full-lattice handle's working buffer
__global__ void __launch_bounds__(256) k_shard_copy_nu(double2 *__restrict__ dst, const u64s *__restrict__ src, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    dst[i] = make_double2(__longlong_as_double((long long)sld(src + 2 * i)), __longlong_as_double((long long)sld(src + 2 * i + 1)));
}

// nu_slab[w][s] = nu_full[w][gsites[s]] for every slab site (own and ghost)
__global__ void __launch_bounds__(256) k_shard_gather_nu(double2 *__restrict__ nu_slab, const double2 *__restrict__ nu_full,
                                                         const int *__restrict__ gsites, int Lo2, int N_loc, int n_global) {
    const int w = blockIdx.x;
    for (int s = threadIdx.x; s < N_loc; s += 256) nu_slab[(size_t)w * N_loc + s] = nu_full[(size_t)w * n_global + gsites[s]];
}

// per-slice partial r.z over the own sites -> rz[t] (slots >= L zeroed), the layout k_cg_ap / k_cg_state0 reduce
__global__ void __launch_bounds__(64) k_shard_rz_own(const double *__restrict__ r, const double *__restrict__ zp, double *__restrict__ rz,
                                                     int nrz, int N, int L, int lo, int hi) {
    const int t = blockIdx.x;
    double a = 0.0;
    for (int s = lo + threadIdx.x; s < hi; s += 64) a += r[(size_t)t * N + s] * zp[(size_t)t * N + s];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (threadIdx.x == 0) {
        rz[t] = a;
        for (int q = L + t; q < nrz; q += L) rz[q] = 0.0;
    }
}

static int launch_ok(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { elph_set_error("launch %s failed: %s", what, hipGetErrorString(e)); return ELPH_E_HIP; }
    return ELPH_OK;
}

// Wait bound of the sharded solves (ms): ELPH_SHARD_TIMEOUT_MS, else ELPH_WG_TIMEOUT_MS, else 20 s.  A sharded solve has no fallback
// behind it (a time-out is ELPH_E_HIP), so it keeps the long bound; the 2 s default of ELPH_WG_TIMEOUT_MS belongs to the un-sharded
// resident kernels, which fall back to the streaming iteration.
long long elph_shard_timeout_ms() {
    const char *es = getenv("ELPH_SHARD_TIMEOUT_MS"), *ew = getenv("ELPH_WG_TIMEOUT_MS");
    const long long ms = es ? atoll(es) : (ew ? atoll(ew) : 20000);
    return ms > 0 ? ms : 20000;
}
static long long shard_timeout_ticks() { return elph_shard_timeout_ms() * 100000LL; }      // wall_clock64 runs at 100 MHz

// ---- preflight of the mailbox protocol (elph_shard_selftest) ---------------------------------------------------------------------------
// `rounds` lock-step rounds between ALL ranks, one wave per rank: in round k every rank stores the granule {seq, k} into slot [rank] of
// all-sum record 7 in every peer's mailbox (the same sc1 / system-scope stores the solve uses: hipIpc- or pointer-mapped memory, xGMI
// between GPUs) and polls its own mailbox until every peer's granule shows round >= k.  lat[q] = mean ticks between this rank's store
// of round k and seeing rank q's; status[q] = 1 when rank q never answered within the bound.  Nothing else in the library uses record 7.
__global__ void __launch_bounds__(64) k_shard_selftest(ElphShardCtl Sh, size_t ext_off, unsigned seq, int rounds, long long timeout_ticks,
                                                       long long *lat, int *status) {
    const int lane = threadIdx.x, P = Sh.P;
    const size_t base = ext_off + (size_t)7 * 8 * 2;
    long long acc = 0;
    bool dead = false;
    for (int k = 1; k <= rounds; ++k) {
        const u64s g = ((u64s)seq << 32) | (u64s)(unsigned)k;
        if (lane < P) sst(Sh.mail[lane] + base + (size_t)Sh.rank * 2, g);
        const long long t_store = wall_clock64();
        bool seen = (lane >= P);
        long long t_seen = t_store;
        for (int spin = 0;; ++spin) {
            if (!seen) {
                const u64s v = sld(Sh.mail[Sh.rank] + base + (size_t)lane * 2);
                if ((unsigned)(v >> 32) == seq && (unsigned)v >= (unsigned)k) { seen = true; t_seen = wall_clock64(); }
            }
            if (__all(seen)) break;
            if ((spin & 31) == 31 && wall_clock64() - t_store > timeout_ticks) { dead = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (dead) { if (lane < P) status[lane] = seen ? 0 : 1; break; }
        acc += t_seen - t_store;
    }
    if (lane < P) { lat[lane] = dead ? -1 : acc; if (!dead) status[lane] = 0; }
}

// Collective over the ranks of a connected shard, under the protocol of a solve: elph_shard_prepare on every rank, the caller's barrier,
// then this call on every rank.  us_per_round[P] (may be NULL): mean time between this rank's store and the sight of rank q's granule
// (own entry: the local round trip); *slowest_us: the largest of them.  A peer that does not answer within ELPH_SHARD_SELFTEST_MS
// (default 10 s: first contact includes code loading and whatever skew the host barrier leaves) gives ELPH_E_HIP with the silent
// ranks named — at set-up, instead of a time-out inside the first solve.
extern "C" int elph_shard_selftest(elph_handle h, int rounds, double *us_per_round, double *slowest_us) {
    if (!h || !h->shard) { elph_set_error("elph_shard_create has not been called"); return ELPH_E_STATE; }
    HIPCHK(hipSetDevice(h->device));
    ShardState *S = static_cast<ShardState *>(h->shard);
    // arguments first: a bad call must not consume the prepare + barrier the ranks have just paid for
    if (rounds < 1 || rounds > (1 << 20)) { elph_set_error("bad number of rounds"); return ELPH_E_ARG; }
    if (!S->prepared) { elph_set_error("elph_shard_prepare (and the caller's barrier) must precede elph_shard_selftest"); return ELPH_E_STATE; }
    S->prepared = false;
    const int P = S->ctl.P;
    static_assert(ELPH_SHARD_MAXRANKS <= 64, "the self-test's device buffers hold 64 ranks");
    long long *d_lat = nullptr;
    HIPCHK(hipMalloc((void **)&d_lat, 64 * sizeof(long long) + 64 * sizeof(int)));
    int *d_status = reinterpret_cast<int *>(d_lat + 64);
    {
        hipError_t em = hipMemsetAsync(d_lat, 0xFF, 64 * sizeof(long long) + 64 * sizeof(int), h->stream);
        if (em != hipSuccess) { (void)hipFree(d_lat); elph_set_error("elph_shard_selftest: %s", hipGetErrorString(em)); return ELPH_E_HIP; }
    }
    const char *eb = getenv("ELPH_SHARD_SELFTEST_MS");
    const long long bound_ms = (eb && atoll(eb) > 0) ? atoll(eb) : 10000;
    const unsigned seq = 0x5E1F0000u + (++S->selftest_calls & 0xFFFFu);       // (a collective: every rank's n-th call carries the same number)
    hipLaunchKernelGGL(k_shard_selftest, dim3(1), dim3(64), 0, h->stream, S->ctl, S->ext_off, seq, rounds, bound_ms * 100000LL, d_lat, d_status);
    int rc = launch_ok("k_shard_selftest");
    long long lat[ELPH_SHARD_MAXRANKS];
    int status[ELPH_SHARD_MAXRANKS];
    if (rc == ELPH_OK) {
        hipError_t e = hipMemcpyAsync(lat, d_lat, sizeof(long long) * (size_t)P, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(status, d_status, sizeof(int) * (size_t)P, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { elph_set_error("elph_shard_selftest: %s", hipGetErrorString(e)); rc = ELPH_E_HIP; }
    }
    (void)hipFree(d_lat);
    if (rc) return rc;
    std::string silent;
    double worst = 0.0;
    for (int q = 0; q < P; ++q) {
        if (status[q] != 0) { silent += (silent.empty() ? "" : ", ") + std::to_string(q); continue; }
        const double us = (double)lat[q] / (double)rounds / 100.0;       // wall_clock64: 100 MHz
        if (us_per_round) us_per_round[q] = us;
        worst = std::max(worst, us);
    }
    if (!silent.empty()) {
        elph_set_error("elph_shard_selftest: rank %d saw no mailbox granule from rank(s) %s within %lld ms (peer mapping, peer access or the "
                       "prepare/barrier order is broken)", S->ctl.rank, silent.c_str(), bound_ms);
        return ELPH_E_HIP;
    }
    if (slowest_us) *slowest_us = worst;
    return ELPH_OK;
}

// Whether device `dev_a` can map memory of device `dev_b` (hipDeviceCanAccessPeer; 1 for dev_a == dev_b): what the mailbox stores of a
// solve sharded over the GPUs of a node need.  No handle, no context switch.
extern "C" int elph_peer_access(int dev_a, int dev_b, int *can) {
    if (!can) { elph_set_error("null argument"); return ELPH_E_ARG; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { elph_set_error("no HIP device visible"); return ELPH_E_NOGPU; }
    if (dev_a < 0 || dev_b < 0 || dev_a >= n || dev_b >= n) { elph_set_error("device index out of range (%d devices)", n); return ELPH_E_ARG; }
    if (dev_a == dev_b) { *can = 1; return ELPH_OK; }
    int c = 0;
    HIPCHK(hipDeviceCanAccessPeer(&c, dev_a, dev_b));
    *can = c;
    return ELPH_OK;
}

static int allsum(elph_handle_s *h, ShardState *S, double *part, int n, int slot) {
    hipLaunchKernelGGL(k_shard_allsum, dim3(1), dim3(64), 0, h->stream, part, n, S->ctl, S->ext_off, slot, ++S->epoch, S->d_abort,
                       shard_timeout_ticks());
    return launch_ok("k_shard_allsum");
}

// z = P^-1 r on the slab (own + ghost sites) through the full-lattice handle hf; r.z partials (own sites) -> B.rz, combined over ranks
static int shard_kpm_apply(elph_handle_s *h, elph_handle_s *hf, ShardState *S) {
    const int N = (int)h->N, L = (int)h->L, Lo2 = (L + 1) / 2;
    CgBufs B = elph_make_bufs(h, 1);
    int rc = elph_dft_fwd_twisted(h, h->d_nu, h->d_r, N, 1, nullptr);
    if (rc) return rc;
    const unsigned ep = ++S->epoch;
    hipLaunchKernelGGL(k_shard_push_nu, dim3((unsigned)Lo2), dim3(256), 0, h->stream, h->d_nu, S->ctl, Lo2, N, S->ctl.own_lo,
                       S->ctl.own_hi - S->ctl.own_lo, (int)S->own_gstart, (int)S->n_global, S->nu_off, S->ext_off, ep, S->d_counter);
    if ((rc = launch_ok("k_shard_push_nu"))) return rc;
    hipLaunchKernelGGL(k_shard_wait_nu, dim3(1), dim3(64), 0, h->stream, S->ctl, S->ext_off, ep, S->d_abort, shard_timeout_ticks());
    if ((rc = launch_ok("k_shard_wait_nu"))) return rc;
    const long long nfull = (long long)Lo2 * S->n_global;
    hipLaunchKernelGGL(k_shard_copy_nu, dim3((unsigned)((nfull + 255) / 256)), dim3(256), 0, h->stream, hf->d_nu, S->mail + S->nu_off, nfull);
    if ((rc = launch_ok("k_shard_copy_nu"))) return rc;
    // Chebyshev recursion alone (parts = 2) on the full lattice, in place on hf->d_nu; hf shares this stream.  An inactive expansion
    // (KPMPreconditioners.jl:475-478) is the identity: the spectrum goes back as it came.
    if (hf->kpm_active && (rc = elph_launch_kpm_apply(hf, hf->d_zp, hf->d_r, 1, 0, 2))) return rc;
    hipLaunchKernelGGL(k_shard_gather_nu, dim3((unsigned)Lo2), dim3(256), 0, h->stream, h->d_nu, hf->d_nu, S->d_gsites, Lo2, N, (int)S->n_global);
    if ((rc = launch_ok("k_shard_gather_nu"))) return rc;
    if ((rc = elph_dft_inv_twisted(h, h->d_zp, h->d_nu, N, 1, nullptr, nullptr, nullptr, 0))) return rc;
    hipLaunchKernelGGL(k_shard_rz_own, dim3((unsigned)L), dim3(64), 0, h->stream, h->d_r, h->d_zp, B.rz, B.nrz, N, L, S->ctl.own_lo, S->ctl.own_hi);
    if ((rc = launch_ok("k_shard_rz_own"))) return rc;
    return allsum(h, S, B.rz, B.nrz, 2);
}

// the preconditioned sharded solve on DEVICE-resident vectors: right-hand side in h->d_b (layout S, slab with its ghost entries), solution
// left in h->d_x; needs S->prepared (elph_shard_prepare + the ranks' barrier)
static int shard_kpm_core(elph_handle_s *h, elph_handle_s *hfull, ShardState *S, double tol, int64_t maxiter, double kappa_max,
                          int64_t *iters, int *done, double *eps) {
    if (!S->prepared) { elph_set_error("elph_shard_prepare (and the caller's barrier) must precede every sharded solve"); return ELPH_E_STATE; }
    S->prepared = false;
    if (S->n_global <= 0 || hfull->N != S->n_global || hfull->L != h->L) { elph_set_error("the full-lattice handle does not match the shard's global geometry"); return ELPH_E_ARG; }
    if (!hfull->kpm_ready) { elph_set_error("elph_kpm_setup has not been called on the full-lattice handle"); return ELPH_E_STATE; }
    if (!h->have_E) { elph_set_error("update_model has not been called on this handle"); return ELPH_E_STATE; }
    int rc;
    if ((rc = elph_i_ensure_capacity(h, 1)) || (rc = elph_i_ensure_capacity(hfull, 1))) return rc;
    // inner products over the own rows: the generic kernel family (elph_set_dot_range)
    if ((rc = elph_i_set_dot_range(h, S->ctl.own_lo, S->ctl.own_hi))) return rc;
    hipStream_t saved = hfull->stream;
    HIPCHK(hipStreamSynchronize(hfull->stream));
    hfull->stream = h->stream;
    S->epoch = 0;
    HIPCHK(hipMemsetAsync(S->d_counter, 0, 64, h->stream));
    CgParams P;
    P.tol = tol; P.kmax = (kappa_max > 0.0) ? kappa_max : h->kmax; P.maxiter = maxiter; P.use_prec = 1; P.record_hist = 0; P.hist_stride = 0;
    h->cur_params = P;
    const size_t bytes = (size_t)h->ndim * sizeof(double);
    auto body = [&]() -> int {
        int r;
        HIPCHK(hipMemsetAsync(h->d_x, 0, bytes, h->stream));
        HIPCHK(hipMemsetAsync(h->d_tmp, 0, bytes, h->stream));                      // A x0 = 0
        if ((r = elph_launch_cg_init_only(h, 1))) return r;                         // r0 = p0 = b, partial r.r and b.b over the own rows
        CgBufs B = elph_make_bufs(h, 1);
        const size_t Pst = (size_t)h->cap_rhs * (size_t)h->L * (size_t)h->npl;
        if ((r = allsum(h, S, B.rr, (int)h->L, 1))) return r;
        if ((r = allsum(h, S, h->d_part + 3 * Pst, (int)h->L, 3))) return r;         // b.b
        if ((r = shard_kpm_apply(h, hfull, S))) return r;                           // z0 = P^-1 r0, rho0 = r0.z0 (IterativeSolvers.jl:182-189)
        if ((r = elph_launch_cg_init_prec_only(h, 1))) return r;                    // p0 = z0
        if ((r = elph_launch_cg_state0_only(h, 1))) return r;
        const int check_every = 4;
        for (int64_t launched = 0; launched <= maxiter + 1;) {
            for (int k = 0; k < check_every; ++k, ++launched) {
                if ((r = elph_launch_cg_kernel(h, 1, 0))) return r;                 // p = z + beta p, A p, partial p.Ap
                if ((r = allsum(h, S, B.pap, B.npap, 0))) return r;
                if ((r = elph_launch_cg_kernel(h, 1, 1))) return r;                 // x += alpha p, r -= alpha A p, partial r.r
                if ((r = allsum(h, S, B.rr, (int)h->L, 1))) return r;
                if ((r = shard_kpm_apply(h, hfull, S))) return r;                   // z = P^-1 r, partial r.z
            }
            HIPCHK(hipMemcpyAsync(h->h_state, h->d_state, sizeof(CgState) * 2, hipMemcpyDeviceToHost, h->stream));
            int ab = 0;
            HIPCHK(hipMemcpyAsync(&ab, S->d_abort, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            if (ab) { elph_set_error("sharded preconditioned CG: a rank timed out waiting for its peers"); return ELPH_E_HIP; }
            const CgState &s = h->h_state[h->ap_count & 1];
            if (s.done) {
                if (iters) *iters = s.iters;
                if (done) *done = s.done;
                if (eps) *eps = s.eps;
                return ELPH_OK;
            }
        }
        elph_set_error("sharded preconditioned CG ended without a terminal state (internal error)");
        return ELPH_E_STATE;
    };
    rc = body();
    (void)hipStreamSynchronize(h->stream);
    hfull->stream = saved;
    (void)elph_i_set_dot_range(h, 0, h->N);
    return rc;
}

// hfull: a handle on the WHOLE lattice with the model set and elph_kpm_setup done (identically on every rank); it is switched to the
// slab handle's stream for the duration of the call.  Needs elph_shard_prepare + the caller's barrier like every sharded solve.
extern "C" int elph_shard_solve_kpm(elph_handle h, elph_handle hfull, double *x_slab, const double *b_slab, double tol, int64_t maxiter,
                                    double kappa_max, int64_t *iters, int *done, double *eps) {
    if (!h || !hfull) { elph_set_error("null handle"); return ELPH_E_ARG; }
    HIPCHK(hipSetDevice(h->device));
    ShardState *S = static_cast<ShardState *>(h->shard);
    if (!S) { elph_set_error("elph_shard_create has not been called"); return ELPH_E_STATE; }
    if (!x_slab || !b_slab || !(tol >= 0.0) || maxiter < 1) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    int rc;
    if ((rc = elph_i_ensure_capacity(h, 1))) return rc;
    const size_t bytes = (size_t)h->ndim * sizeof(double);
    HIPCHK(hipMemcpyAsync(h->d_stage_in, b_slab, bytes, hipMemcpyHostToDevice, h->stream));
    if ((rc = elph_launch_r2s(h, h->d_b, h->d_stage_in, 1))) return rc;
    if ((rc = shard_kpm_core(h, hfull, S, tol, maxiter, kappa_max, iters, done, eps))) return rc;
    if ((rc = elph_launch_s2r(h, h->d_stage_out, h->d_x, 1))) return rc;
    HIPCHK(hipMemcpyAsync(x_slab, h->d_stage_out, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}


// =========================================================================================================================
// The CALLERS of the solve on a sharded lattice (BASELINE configs "HMC ... spatial-sharded"): ldiv!'s wrapper (Models.jl:74-186), the
// fermion force (HMC.jl:790-915) and — hmc.hip, through the elph_i_shard_* hooks below — one HMC update, on DEVICE-resident slab vectors.
// Everything but the solve is pointwise in the site index or reaches no further than the MᵀM closure the slab already holds, so it runs
// on the slab handle unchanged; what crosses ranks besides the solve are a few scalars (the true residual, the energies) and, once per
// force evaluation, the ghost rows of a vector — through two host collectives the caller provides once (elph_shard_set_collectives: a
// barrier, and an in-place sum over the ranks; MPI.Barrier / MPI.Allreduce!, torch.distributed, or a thread barrier for ranks that
// share a process).  The conjugate-gradient iteration itself stays as it was: device-initiated mailbox stores, no host.
// =========================================================================================================================

extern "C" int elph_shard_set_collectives(elph_handle h, elph_shard_barrier_fn barrier, elph_shard_allreduce_fn allreduce_sum, void *ctx) {
    if (!h || !h->shard) { elph_set_error("elph_shard_create has not been called"); return ELPH_E_STATE; }
    ShardState *S = static_cast<ShardState *>(h->shard);
    if ((!barrier || !allreduce_sum) && S->ctl.P > 1) { elph_set_error("a barrier and an all-reduce are needed for more than one rank"); return ELPH_E_ARG; }
    S->barrier = barrier; S->allreduce = allreduce_sum; S->coll_ctx = ctx;
    return ELPH_OK;
}

// The full-lattice handle that carries the KPM expansion for preconditioned solves inside elph_hmc_update on a sharded handle (the other
// sharded callers take it as an argument).  elph_kpm_create must have been called on it; it needs no field of its own: every force evaluation
// injects the τ-averaged exp(−ΔτV) of the whole lattice, summed over the ranks' own rows.
extern "C" int elph_shard_set_full_lattice(elph_handle h, elph_handle hfull) {
    if (!h || !h->shard) { elph_set_error("elph_shard_create has not been called"); return ELPH_E_STATE; }
    ShardState *S = static_cast<ShardState *>(h->shard);
    if (hfull && (hfull->N != S->n_global || hfull->L != h->L || hfull->kind != h->kind)) { elph_set_error("the full-lattice handle does not match the shard's global geometry"); return ELPH_E_ARG; }
    S->full = hfull;
    return ELPH_OK;
}

elph_handle_s *elph_i_shard_full(const elph_handle_s *h) {
    const ShardState *S = h ? static_cast<const ShardState *>(h->shard) : nullptr;
    return S ? S->full : nullptr;
}

static ShardState *shard_callers(elph_handle_s *h) {
    ShardState *S = h ? static_cast<ShardState *>(h->shard) : nullptr;
    if (!S) { elph_set_error("elph_shard_create has not been called"); return nullptr; }
    if (S->ctl.P > 1 && (!S->barrier || !S->allreduce)) { elph_set_error("elph_shard_set_collectives has not been called"); return nullptr; }
    return S;
}

bool elph_i_shard_active(const elph_handle_s *h) {
    const ShardState *S = h ? static_cast<const ShardState *>(h->shard) : nullptr;
    return S && (S->ctl.P == 1 || (S->barrier && S->allreduce));
}

void elph_i_shard_own_range(const elph_handle_s *h, int *lo, int *hi) {
    const ShardState *S = static_cast<const ShardState *>(h->shard);
    *lo = S->ctl.own_lo; *hi = S->ctl.own_hi;
}

// in-place sum over the ranks of n doubles (identical result on every rank: the caller's all-reduce)
int elph_i_shard_allreduce(elph_handle_s *h, double *buf, int n) {
    ShardState *S = shard_callers(h);
    if (!S) return ELPH_E_STATE;
    if (S->ctl.P == 1 || n == 0) return ELPH_OK;
    if (S->allreduce(S->coll_ctx, buf, n) != 0) { elph_set_error("the caller's all-reduce failed"); return ELPH_E_HIP; }
    return ELPH_OK;
}

static int shard_arm(elph_handle_s *h, ShardState *S) {      // mailbox zeroed on every rank before any of them stores into one
    int rc = elph_shard_prepare(h);
    if (rc) return rc;
    if (S->ctl.P > 1 && S->barrier(S->coll_ctx) != 0) { elph_set_error("the caller's barrier failed"); return ELPH_E_HIP; }
    return ELPH_OK;
}

// ---- ghost rows (ghost columns) of a slab vector from their owners, through the mailboxes ------------------------------------------
// Between two solves the spectrum area of every mailbox is idle: it is the exchange area.  Every rank stores the columns it OWNS of
// vec[vt][column] (vt = vector x time slice) into the area of every other rank at [vt][global column] (peer stores, 8 bytes each), the
// ranks pass the caller's barrier, and every rank takes the columns it does NOT own out of its own area.  No host copy of the vector, no
// all-reduce of a whole-lattice array: what crosses between the GPUs is what the owners hold, once.
// own == nullptr: site vectors — column c is owned when lo <= c < hi.
__global__ void __launch_bounds__(256) k_shard_push_cols(const double *__restrict__ vec, ElphShardCtl Sh, size_t area_off, int ncols, int ngcol,
                                                         const int *__restrict__ gcol, const double *__restrict__ own, int lo, int hi) {
    const size_t vt = blockIdx.x;
    for (int c = threadIdx.x; c < ncols; c += 256) {
        const bool mine = own ? own[c] != 0.0 : (c >= lo && c < hi);
        if (!mine) continue;
        const u64s bits = (u64s)__double_as_longlong(vec[vt * ncols + c]);
        const size_t at = area_off + vt * ngcol + gcol[c];
        for (int q = 0; q < Sh.P; ++q) if (q != Sh.rank) sst(Sh.mail[q] + at, bits);
    }
}

__global__ void __launch_bounds__(256) k_shard_pull_cols(double *__restrict__ vec, const u64s *__restrict__ area, int ncols, int ngcol,
                                                         const int *__restrict__ gcol, const double *__restrict__ own, int lo, int hi) {
    const size_t vt = blockIdx.x;
    for (int c = threadIdx.x; c < ncols; c += 256) {
        const bool mine = own ? own[c] != 0.0 : (c >= lo && c < hi);
        if (mine) continue;
        vec[vt * ncols + c] = __longlong_as_double((long long)sld(area + vt * ngcol + gcol[c]));
    }
}

// *done = false: not taken (one rank, ELPH_SHARD_GHOST_HOST=1, a vector that does not fit the exchange area) — the caller stages through the host
static int ghost_sync_dev(elph_handle_s *h, ShardState *S, double *vec, int nvec, int ncols, int ngcol, const int *d_gcol, const double *d_own,
                          bool *done) {
    *done = false;
    const char *e = getenv("ELPH_SHARD_GHOST_HOST");
    if ((e && e[0] == '1') || !S->connected || nvec < 1) return ELPH_OK;
    const size_t rows = (size_t)nvec * (size_t)h->L;
    if (rows * (size_t)ngcol > S->xch_words) return ELPH_OK;
    // (1) nobody still reads this area — a spectrum of the last solve, the pull of an earlier exchange: those are stream-ordered on their rank
    HIPCHK(hipStreamSynchronize(h->stream));
    if (S->barrier(S->coll_ctx) != 0) { elph_set_error("the caller's barrier failed"); return ELPH_E_HIP; }
    hipLaunchKernelGGL(k_shard_push_cols, dim3((unsigned)rows), dim3(256), 0, h->stream, (const double *)vec, S->ctl, S->nu_off, ncols, ngcol, d_gcol,
                       d_own, S->ctl.own_lo, S->ctl.own_hi);
    { const int rcl = launch_ok("k_shard_push_cols"); if (rcl) return rcl; }
    // (2) every rank's stores have landed (a kernel's system-scope stores are visible once its stream has drained)
    HIPCHK(hipStreamSynchronize(h->stream));
    if (S->barrier(S->coll_ctx) != 0) { elph_set_error("the caller's barrier failed"); return ELPH_E_HIP; }
    hipLaunchKernelGGL(k_shard_pull_cols, dim3((unsigned)rows), dim3(256), 0, h->stream, vec, (const u64s *)(S->mail + S->nu_off), ncols, ngcol, d_gcol,
                       d_own, S->ctl.own_lo, S->ctl.own_hi);
    { const int rcl = launch_ok("k_shard_pull_cols"); if (rcl) return rcl; }
    ++S->ghost_dev;
    *done = true;
    return ELPH_OK;
}

// how many ghost exchanges of this handle went through the mailboxes / through the host collectives (tests, bench)
extern "C" int elph_shard_ghost_stats(elph_handle h, int64_t *through_mailboxes, int64_t *through_host) {
    if (!h || !h->shard) { elph_set_error("elph_shard_create has not been called"); return ELPH_E_STATE; }
    const ShardState *S = static_cast<const ShardState *>(h->shard);
    if (through_mailboxes) *through_mailboxes = (int64_t)S->ghost_dev;
    if (through_host) *through_host = (int64_t)S->ghost_host;
    return ELPH_OK;
}

// Ghost rows of a site vector (layout S on the slab, nvec vectors) from their owners; the own rows are not touched.  Through the mailboxes
// (above); the fall-back stages through the host: every rank contributes its own rows to a vector on the WHOLE lattice, the ranks sum it,
// and each takes its ghost rows from the sum.  Once per force evaluation / refresh, not per CG iteration.
int elph_i_shard_ghost_sync(elph_handle_s *h, double *vecS, int nvec) {
    ShardState *S = shard_callers(h);
    if (!S) return ELPH_E_STATE;
    if (S->ctl.P == 1) return ELPH_OK;
    if (S->n_global <= 0 || S->gsites_host.empty()) { elph_set_error("the shard was created without its global geometry"); return ELPH_E_STATE; }
    {
        bool done = false;
        const int rcd = ghost_sync_dev(h, S, vecS, nvec, (int)h->N, (int)S->n_global, S->d_gsites, nullptr, &done);
        if (rcd) return rcd;
        if (done) return ELPH_OK;
        ++S->ghost_host;
    }
    const size_t N = (size_t)h->N, L = (size_t)h->L, NG = (size_t)S->n_global;
    std::vector<double> loc(N * L * (size_t)nvec), glob(NG * L * (size_t)nvec, 0.0);
    HIPCHK(hipMemcpyAsync(loc.data(), vecS, loc.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t lo = (size_t)S->ctl.own_lo, hi = (size_t)S->ctl.own_hi;
    for (size_t v = 0; v < (size_t)nvec; ++v)
        for (size_t t = 0; t < L; ++t)
            for (size_t s2 = lo; s2 < hi; ++s2) glob[(v * L + t) * NG + (size_t)S->gsites_host[s2]] = loc[(v * L + t) * N + s2];
    int rc = elph_i_shard_allreduce(h, glob.data(), (int)glob.size());
    if (rc) return rc;
    for (size_t v = 0; v < (size_t)nvec; ++v)
        for (size_t t = 0; t < L; ++t)
            for (size_t s2 = 0; s2 < N; ++s2)
                if (s2 < lo || s2 >= hi) loc[(v * L + t) * N + s2] = glob[(v * L + t) * NG + (size_t)S->gsites_host[s2]];
    HIPCHK(hipMemcpyAsync(vecS, loc.data(), loc.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}

// Ē = τ-mean of exp(−ΔτV) on the WHOLE lattice (update_A!, KPMPreconditioners.jl:332-349): every rank averages its own rows (the slab
// handle's d_E), the ranks sum the disjoint pieces.
__global__ void __launch_bounds__(64) k_shard_ebar_own(double *__restrict__ out, const double *__restrict__ E, int N, int L, int lo, int hi) {
    const int s2 = lo + (int)(blockIdx.x * 64 + threadIdx.x);
    if (s2 >= hi) return;
    double a = 0.0;
    for (int t = 0; t < L; ++t) a += E[(size_t)t * N + s2];
    out[s2 - lo] = a / L;
}

int elph_i_shard_global_ebar(elph_handle_s *h, std::vector<double> &Eg) {
    ShardState *S = shard_callers(h);
    if (!S) return ELPH_E_STATE;
    if (S->n_global <= 0) { elph_set_error("the shard was created without its global geometry"); return ELPH_E_STATE; }
    const int lo = S->ctl.own_lo, hi = S->ctl.own_hi, n = hi - lo;
    hipLaunchKernelGGL(k_shard_ebar_own, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, h->stream, h->d_tmp, h->d_E, (int)h->N, (int)h->L, lo, hi);
    int rc = launch_ok("k_shard_ebar_own");
    if (rc) return rc;
    std::vector<double> own((size_t)n);
    HIPCHK(hipMemcpyAsync(own.data(), h->d_tmp, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    Eg.assign((size_t)S->n_global, 0.0);
    for (int i = 0; i < n; ++i) Eg[(size_t)S->gsites_host[(size_t)(lo + i)]] = own[(size_t)i];
    return elph_i_shard_allreduce(h, Eg.data(), (int)Eg.size());
}

// Bond phonons under a preconditioner: the expansion on the full-lattice handle takes the tau-means of cosh / sinh of EVERY bond of the lattice
// (update_A!, KPMPreconditioners.jl:355-381); a rank holds the tables of its slab's bonds.  global_bond[b] = the bond's position in the
// checkerboard order of the whole lattice, own_weight[b] = 1 for the bonds this rank owns (every bond of the lattice has exactly one owner).
extern "C" int elph_shard_set_bonds(elph_handle h, const int64_t *global_bond, int64_t n_global_bonds, const double *own_weight) {
    if (!h || !h->shard) { elph_set_error("elph_shard_create has not been called"); return ELPH_E_STATE; }
    ShardState *S = static_cast<ShardState *>(h->shard);
    if (!global_bond || !own_weight || n_global_bonds < h->nb) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    std::vector<int> gb((size_t)h->nb);
    std::vector<double> w((size_t)h->nb);
    for (int64_t b = 0; b < h->nb; ++b) {
        if (global_bond[b] < 0 || global_bond[b] >= n_global_bonds) { elph_set_error("global_bond[%lld] out of range", (long long)b); return ELPH_E_ARG; }
        gb[(size_t)b] = (int)global_bond[b];
        w[(size_t)b] = own_weight[b] != 0.0 ? 1.0 : 0.0;
    }
    HIPCHK(hipSetDevice(h->device));
    if (S->d_csbar) HIPCHK(hipFree(S->d_csbar));
    S->d_csbar = nullptr;
    if (h->nb > 0) HIPCHK(hipMalloc((void **)&S->d_csbar, 2 * (size_t)h->nb * sizeof(double)));
    S->gbond.swap(gb); S->bown.swap(w); S->n_gbonds = n_global_bonds;
    return ELPH_OK;
}

bool elph_i_shard_has_bonds(const elph_handle_s *h) {
    const ShardState *S = h ? static_cast<const ShardState *>(h->shard) : nullptr;
    return S && S->n_gbonds > 0 && (int64_t)S->gbond.size() == h->nb;
}

// cs = [c̄ of every bond of the lattice | s̄ of every bond]: every rank takes the tau-means of its slab's tables on the device, contributes
// the bonds it owns, the ranks sum the disjoint pieces (2 x bonds doubles through the caller's all-reduce — no vector of the lattice)
int elph_i_shard_global_csbar(elph_handle_s *h, std::vector<double> &cs, int64_t *n_bonds) {
    ShardState *S = shard_callers(h);
    if (!S) return ELPH_E_STATE;
    if (!elph_i_shard_has_bonds(h)) { elph_set_error("elph_shard_set_bonds has not been called"); return ELPH_E_STATE; }
    const size_t nb = (size_t)h->nb, NB = (size_t)S->n_gbonds;
    if (h->kind != ELPH_MODEL_SSH || !h->have_E) { elph_set_error("the tau-means of the hopping tables: a bond-phonon handle after update_model"); return ELPH_E_STATE; }
    int rc = elph_launch_cs_bar(h, S->d_csbar, S->d_csbar + nb, 1);
    if (rc) return rc;
    std::vector<double> cl(nb), sl(nb);
    HIPCHK(hipMemcpyAsync(cl.data(), S->d_csbar, nb * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(sl.data(), S->d_csbar + nb, nb * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    cs.assign(2 * NB, 0.0);
    for (size_t b = 0; b < nb; ++b)
        if (S->bown[b] != 0.0) { cs[(size_t)S->gbond[b]] = cl[b]; cs[NB + (size_t)S->gbond[b]] = sl[b]; }
    *n_bonds = S->n_gbonds;
    return elph_i_shard_allreduce(h, cs.data(), (int)cs.size());
}

// The same for vectors whose columns are not sites (bond-phonon fields): gcol[c] = the column's number on the whole lattice, own[c] = 1 when
// this rank owns it; owned columns are contributed, the others taken from the sum.
int elph_i_shard_ghost_sync_cols(elph_handle_s *h, double *vecS, int nvec, int ncols, const int *gcol, int ngcol, const double *own) {
    ShardState *S = shard_callers(h);
    if (!S) return ELPH_E_STATE;
    if (S->ctl.P == 1) return ELPH_OK;
    {
        if ((size_t)ncols > S->xcol_cap) {
            HIPCHK(hipStreamSynchronize(h->stream));
            if (S->d_xcol) HIPCHK(hipFree(S->d_xcol));
            if (S->d_xown) HIPCHK(hipFree(S->d_xown));
            S->d_xcol = nullptr; S->d_xown = nullptr; S->xcol_cap = 0;
            HIPCHK(hipMalloc((void **)&S->d_xcol, (size_t)ncols * sizeof(int)));
            HIPCHK(hipMalloc((void **)&S->d_xown, (size_t)ncols * sizeof(double)));
            S->xcol_cap = (size_t)ncols;
        }
        for (int c = 0; c < ncols; ++c)
            if (gcol[c] < 0 || gcol[c] >= ngcol) { elph_set_error("ghost exchange: column %d maps to %d of %d", c, gcol[c], ngcol); return ELPH_E_ARG; }
        HIPCHK(hipMemcpyAsync(S->d_xcol, gcol, (size_t)ncols * sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(S->d_xown, own, (size_t)ncols * sizeof(double), hipMemcpyHostToDevice, h->stream));
        bool done = false;
        const int rcd = ghost_sync_dev(h, S, vecS, nvec, ncols, ngcol, S->d_xcol, S->d_xown, &done);
        if (rcd) return rcd;
        if (done) return ELPH_OK;
        ++S->ghost_host;
    }
    const size_t N = (size_t)ncols, L = (size_t)h->L, NG = (size_t)ngcol;
    std::vector<double> loc(N * L * (size_t)nvec), glob(NG * L * (size_t)nvec, 0.0);
    HIPCHK(hipMemcpyAsync(loc.data(), vecS, loc.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (size_t v = 0; v < (size_t)nvec; ++v)
        for (size_t t = 0; t < L; ++t)
            for (size_t c = 0; c < N; ++c)
                if (own[c] != 0.0) glob[(v * L + t) * NG + (size_t)gcol[c]] = loc[(v * L + t) * N + c];
    int rc = elph_i_shard_allreduce(h, glob.data(), (int)glob.size());
    if (rc) return rc;
    for (size_t v = 0; v < (size_t)nvec; ++v)
        for (size_t t = 0; t < L; ++t)
            for (size_t c = 0; c < N; ++c)
                if (own[c] == 0.0) loc[(v * L + t) * N + c] = glob[(v * L + t) * NG + (size_t)gcol[c]];
    HIPCHK(hipMemcpyAsync(vecS, loc.data(), loc.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}

// partial sums over the OWN sites of one slice each: part[0][t] = sum (a - b)^2, part[1][t] = sum b^2
__global__ void __launch_bounds__(64) k_shard_resid_own(const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ part,
                                                        int N, int L, int lo, int hi) {
    const int t = blockIdx.x;
    double num = 0.0, den = 0.0;
    for (int s2 = lo + (int)threadIdx.x; s2 < hi; s2 += 64) {
        const size_t i = (size_t)t * N + s2;
        const double d = a[i] - b[i];
        num += d * d; den += b[i] * b[i];
    }
    for (int o = 32; o > 0; o >>= 1) { num += __shfl_xor(num, o, 64); den += __shfl_xor(den, o, 64); }
    if (threadIdx.x == 0) { part[t] = num; part[L + t] = den; }
}

// ldiv!(x, model, b[, P]) on the slab, device-resident: right-hand side in h->d_b (slot 0, layout S, ghost entries filled), solution in
// h->d_x (slot 0; exact on own AND ghost rows: the solve keeps r, p, x of the ghost rows current through the boundary exchange).
// Models.jl:74-137,139-186 line for line: solve, true residual |MᵀM x − b| / |b| (own rows of every rank, summed), flag 1 (hit maxiter) / 2
// (false convergence), zero-fill, and with a preconditioner the un-preconditioned retry with 10 maxiter.  iters / resid / flag come out
// identical on every rank (the sums are).
int elph_i_shard_ldiv_dev(elph_handle_s *h, elph_handle_s *hfull, int use_prec, int64_t maxiter, int64_t *iters, double *resid, int *flag) {
    ShardState *S = shard_callers(h);
    if (!S) return ELPH_E_STATE;
    if (use_prec && !hfull) { elph_set_error("a preconditioned sharded solve needs the full-lattice handle"); return ELPH_E_ARG; }
    if (maxiter == 0) maxiter = h->maxiter;                                  // Models.jl:78-80,143-145
    const size_t bytes = (size_t)h->ndim * sizeof(double);
    auto solve = [&](int prec, int64_t mi, int64_t *it) -> int {
        int rc = shard_arm(h, S);
        if (rc) return rc;
        if (prec) {
            int done = 0; double eps = 0.0;
            return shard_kpm_core(h, hfull, S, h->tol, mi, h->kmax, it, &done, &eps);
        }
        rc = shard_run(h, nullptr, h->tol, mi, h->kmax, 0, nullptr);
        if (rc) return rc;
        const CgState &st = h->h_state[0];
        if (!st.done) { elph_set_error("sharded CG ended without a terminal state (internal error)"); return ELPH_E_STATE; }
        *it = st.iters;
        return ELPH_OK;
    };
    auto residual_flag = [&](int64_t it, int64_t cmp_maxiter, double *res, int *fl) -> int {
        int rc = elph_launch_mul(h, 2, h->d_tmp, h->d_x, 1);                 // mul!(v, model, x) on the slab: exact on the own rows
        if (rc) return rc;
        hipLaunchKernelGGL(k_shard_resid_own, dim3((unsigned)h->L), dim3(64), 0, h->stream, h->d_tmp, h->d_b, h->d_part, (int)h->N, (int)h->L,
                           S->ctl.own_lo, S->ctl.own_hi);
        if ((rc = launch_ok("k_shard_resid_own"))) return rc;
        std::vector<double> p(2 * (size_t)h->L);
        HIPCHK(hipMemcpyAsync(p.data(), h->d_part, p.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        double nd[2] = {0.0, 0.0};
        for (int64_t t = 0; t < h->L; ++t) { nd[0] += p[(size_t)t]; nd[1] += p[(size_t)(h->L + t)]; }
        if ((rc = elph_i_shard_allreduce(h, nd, 2))) return rc;
        *res = sqrt(nd[0]) / sqrt(nd[1]);
        if (*res > sqrt(h->tol)) {                                           // Models.jl:100,157 (NaN compares false, as in the reference)
            *fl = (it == cmp_maxiter) ? 1 : 2;
            HIPCHK(hipMemsetAsync(h->d_x, 0, bytes, h->stream));             // fill!(x, 0)
        } else {
            *fl = 0;
        }
        return ELPH_OK;
    };
    int rc;
    if ((rc = solve(use_prec, maxiter, iters))) return rc;
    if ((rc = residual_flag(*iters, use_prec ? maxiter : h->maxiter, resid, flag))) return rc;      // :103 compares maxiter, :160 solver.maxiter
    if (use_prec && *flag > 0) {                                             // :129-133
        if ((rc = solve(0, 10 * maxiter, iters))) return rc;
        if ((rc = residual_flag(*iters, h->maxiter, resid, flag))) return rc;
    }
    return ELPH_OK;
}

extern "C" int elph_shard_ldiv(elph_handle h, elph_handle hfull, double *x_slab, const double *b_slab, int use_precond, int64_t maxiter,
                               int64_t *iters, double *residual_error, int *flag) {
    if (!h) { elph_set_error("null handle"); return ELPH_E_ARG; }
    HIPCHK(hipSetDevice(h->device));
    if (!x_slab || !b_slab || !iters || !residual_error || !flag || maxiter < 0) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    if (!h->have_E) { elph_set_error("update_model has not been called on this handle"); return ELPH_E_STATE; }
    int rc;
    if ((rc = elph_i_ensure_capacity(h, 1))) return rc;
    const size_t bytes = (size_t)h->ndim * sizeof(double);
    HIPCHK(hipMemcpyAsync(h->d_stage_in, b_slab, bytes, hipMemcpyHostToDevice, h->stream));
    if ((rc = elph_launch_r2s(h, h->d_b, h->d_stage_in, 1))) return rc;
    if ((rc = elph_i_shard_ldiv_dev(h, hfull, use_precond, maxiter, iters, residual_error, flag))) return rc;
    if ((rc = elph_launch_s2r(h, h->d_stage_out, h->d_x, 1))) return rc;
    HIPCHK(hipMemcpyAsync(x_slab, h->d_stage_out, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}

// The two pseudofermion solves of calc_O⁻¹Λϕ! (HMC.jl:851-886) on the slab: right-hand sides in h->d_b[0], h->d_b[1], solutions into
// h->d_x[0], h->d_x[1] (capacity >= 2), at tol^power; iters = cld(total, 2) and the flag as the reference returns them (:907-909); a
// failed first solve suppresses the second (:880).  One sharded solve at a time (the SHARD kernel carries one right-hand side).
int elph_i_shard_solve_pair(elph_handle_s *h, elph_handle_s *hfull, int use_prec, double tol_power, int64_t *iters, int *flag) {
    const size_t nd = (size_t)h->ndim, bytes = nd * sizeof(double);
    const double tol0 = h->tol;
    h->tol = pow(tol0, tol_power);
    int64_t it1 = 0, it2 = 0;
    double res = 0.0;
    int fl = 0;
    // (+): slot 0 as it lies; keep b(−) aside, it moves into slot 0 for its own solve
    int rc = elph_i_shard_ldiv_dev(h, hfull, use_prec, 0, &it1, &res, &fl);
    int64_t tot = it1;
    if (rc == ELPH_OK && fl == 0) {
        // X₊ waits in slot 1 of d_x for the length of the second solve (a sharded solve carries ONE right-hand side: slot 0 of every per-right-hand-side
        // array; the handle's capacity is two) — on the device, no host round trip; b₊ aside in the staging buffer, b₋ into slot 0
        hipError_t e = hipMemcpyAsync(h->d_x + nd, h->d_x, bytes, hipMemcpyDeviceToDevice, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(h->d_stage_out, h->d_b, bytes, hipMemcpyDeviceToDevice, h->stream);   // b₊ aside
        if (e == hipSuccess) e = hipMemcpyAsync(h->d_b, h->d_b + nd, bytes, hipMemcpyDeviceToDevice, h->stream);
        if (e != hipSuccess) { h->tol = tol0; elph_set_error("sharded pair: %s", hipGetErrorString(e)); return ELPH_E_HIP; }
        rc = elph_i_shard_ldiv_dev(h, hfull, use_prec, 0, &it2, &res, &fl);
        if (rc == ELPH_OK) {
            tot += it2;
            // slot 0 holds X₋, slot 1 X₊: swap through slot 1 of d_z (free: the solves are over)
            e = hipMemcpyAsync(h->d_z + nd, h->d_x, bytes, hipMemcpyDeviceToDevice, h->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(h->d_x, h->d_x + nd, bytes, hipMemcpyDeviceToDevice, h->stream);           // X₊ -> slot 0
            if (e == hipSuccess) e = hipMemcpyAsync(h->d_x + nd, h->d_z + nd, bytes, hipMemcpyDeviceToDevice, h->stream);      // X₋ -> slot 1
            if (e == hipSuccess) e = hipMemcpyAsync(h->d_b + nd, h->d_b, bytes, hipMemcpyDeviceToDevice, h->stream);   // b₋ back to slot 1
            if (e == hipSuccess) e = hipMemcpyAsync(h->d_b, h->d_stage_out, bytes, hipMemcpyDeviceToDevice, h->stream); // b₊ back to slot 0
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) { h->tol = tol0; elph_set_error("sharded pair: %s", hipGetErrorString(e)); return ELPH_E_HIP; }
        }
    } else if (rc == ELPH_OK) {
        rc = elph_launch_zero(h, h->d_x + nd, (int64_t)nd);
    }
    h->tol = tol0;
    if (rc) return rc;
    if (fl == 0) tot = (tot + 1) / 2;                                         // cld(iters, 2)
    *iters = tot;
    *flag = fl;
    return ELPH_OK;
}

// One fermion-force evaluation of the Holstein model on a sharded lattice: elph_fermion_force_holstein's steps (update_model!,
// calc_O⁻¹Λϕ!, calc_dSfdx!) on the slab.  All vectors are slab vectors (own + ghost rows, ghost entries filled from the global arrays);
// the force is exact on the OWN rows (the same closure argument as for z = Mᵀ(M p): CB, then CBᵀ) and only those are accumulated into.
extern "C" int elph_shard_fermion_force_holstein(elph_handle h, elph_handle hfull, const double *x, const double *lambda, const double *lambda2,
                                                 const double *mu, double dtau, const double *phi_plus, const double *phi_minus, int use_precond,
                                                 double tol_power, double *dSfdx, double *Xp_out, double *Xm_out, int64_t *iters, int *flag) {
    if (!h) { elph_set_error("null handle"); return ELPH_E_ARG; }
    HIPCHK(hipSetDevice(h->device));
    ShardState *S = shard_callers(h);
    if (!S) return ELPH_E_STATE;
    if (h->kind != ELPH_MODEL_HOLSTEIN) { elph_set_error("not a Holstein handle"); return ELPH_E_ARG; }
    if (!x || !lambda || !lambda2 || !mu || !phi_plus || !phi_minus || !dSfdx || !iters || !flag) { elph_set_error("null argument"); return ELPH_E_ARG; }
    int rc;
    if ((rc = elph_i_ensure_capacity(h, 2))) return rc;
    const size_t nd = (size_t)h->ndim, N = (size_t)h->N, bytes = nd * sizeof(double);
    HIPCHK(hipMemcpyAsync(h->d_lam, lambda, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_lam + N, lambda2, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_lam + 2 * N, mu, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_stage_in, x, bytes, hipMemcpyHostToDevice, h->stream));
    if ((rc = elph_launch_expV(h, h->d_stage_in, dtau))) return rc;
    if ((rc = elph_launch_r2s(h, h->d_xfield, h->d_stage_in, 1))) return rc;
    h->have_E = true;
    HIPCHK(hipMemcpyAsync(h->d_stage_in, phi_plus, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_stage_in + nd, phi_minus, bytes, hipMemcpyHostToDevice, h->stream));
    if ((rc = elph_launch_r2s(h, h->d_phi, h->d_stage_in, 2))) return rc;
    if ((rc = elph_launch_lambda_rhs(h, h->d_b, h->d_phi, h->d_xfield, dtau))) return rc;
    if ((rc = elph_i_shard_solve_pair(h, hfull, use_precond, tol_power, iters, flag))) return rc;
    if ((rc = elph_launch_force_holstein(h, h->d_tmp, h->d_x, h->d_phi, h->d_xfield, dtau))) return rc;
    if ((rc = elph_launch_s2r(h, h->d_stage_out, h->d_tmp, 1))) return rc;
    std::vector<double> F(nd);
    HIPCHK(hipMemcpyAsync(F.data(), h->d_stage_out, bytes, hipMemcpyDeviceToHost, h->stream));
    if (Xp_out || Xm_out) {
        if ((rc = elph_launch_s2r(h, h->d_stage_in, h->d_x, 2))) return rc;
        if (Xp_out) HIPCHK(hipMemcpyAsync(Xp_out, h->d_stage_in, bytes, hipMemcpyDeviceToHost, h->stream));
        if (Xm_out) HIPCHK(hipMemcpyAsync(Xm_out, h->d_stage_in + nd, bytes, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t L = (size_t)h->L;
    for (size_t s2 = (size_t)S->ctl.own_lo; s2 < (size_t)S->ctl.own_hi; ++s2)
        for (size_t t = 0; t < L; ++t) dSfdx[s2 * L + t] += F[s2 * L + t];      // own rows only ("@. dSfdx += ...", HMC.jl:803-811)
    return ELPH_OK;
}

// The same for the bond-phonon model (elph_fermion_force_ssh): the two solves on the given right-hand sides and the bond brackets
// q_out[n Ltau + tau] of the slab's bonds (local checkerboard order).  A bracket is exact on the rank that owns its bond (both ends in
// the own rows, or one end in the first ghost row across the slab's edge: the crossing colour is the last factor of the sweep); the
// caller keeps the brackets of its own bonds and scatters them onto the phonon fields (sharded.py: ShardedSolver.force_ssh).
extern "C" int elph_shard_fermion_force_ssh(elph_handle h, elph_handle hfull, const double *rhs_plus, const double *rhs_minus, int use_precond,
                                            double tol_power, double *q_out, double *Xp_out, double *Xm_out, int64_t *iters, int *flag) {
    if (!h) { elph_set_error("null handle"); return ELPH_E_ARG; }
    HIPCHK(hipSetDevice(h->device));
    ShardState *S = shard_callers(h);
    if (!S) return ELPH_E_STATE;
    if (h->kind != ELPH_MODEL_SSH) { elph_set_error("not an SSH handle"); return ELPH_E_ARG; }
    if (!h->have_E) { elph_set_error("update_model has not been called on this handle"); return ELPH_E_STATE; }
    if (!rhs_plus || !rhs_minus || !q_out || !iters || !flag) { elph_set_error("null argument"); return ELPH_E_ARG; }
    int rc;
    if ((rc = elph_i_ensure_capacity(h, 2))) return rc;
    const size_t nd = (size_t)h->ndim, bytes = nd * sizeof(double), L = (size_t)h->L, nb = (size_t)h->nb, nq = L * nb;
    if (nq > 2 * nd) { elph_set_error("more bonds than 2*nsites: scratch too small"); return ELPH_E_UNSUPPORTED; }
    HIPCHK(hipMemcpyAsync(h->d_stage_in, rhs_plus, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_stage_in + nd, rhs_minus, bytes, hipMemcpyHostToDevice, h->stream));
    if ((rc = elph_launch_r2s(h, h->d_b, h->d_stage_in, 2))) return rc;
    if ((rc = elph_i_shard_solve_pair(h, hfull, use_precond, tol_power, iters, flag))) return rc;
    if ((rc = elph_launch_force_ssh(h, h->d_p, h->d_x))) return rc;
    std::vector<double> qt(nq);
    if (nq) HIPCHK(hipMemcpyAsync(qt.data(), h->d_p, nq * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (Xp_out || Xm_out) {
        if ((rc = elph_launch_s2r(h, h->d_stage_in, h->d_x, 2))) return rc;
        if (Xp_out) HIPCHK(hipMemcpyAsync(Xp_out, h->d_stage_in, bytes, hipMemcpyDeviceToHost, h->stream));
        if (Xm_out) HIPCHK(hipMemcpyAsync(Xm_out, h->d_stage_in + nd, bytes, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    for (size_t t = 0; t < L; ++t)
        for (size_t n = 0; n < nb; ++n) q_out[n * L + t] = qt[t * nb + n];
    return ELPH_OK;
}
