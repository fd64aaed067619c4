// slabs.hip — the RESIDENT un-preconditioned solve for lattices beyond one wave's slice (N > 320 sites) on ONE GPU.
//
// The workgroup-resident kernel (cg_wg.hip) keeps a time slice of a right-hand side in one wavefront: at most 320 sites.  Larger
// lattices ran the streaming two-kernel iteration only — 25 us per iteration for the reference's call shape (one or two right-hand
// sides, IterativeSolvers.jl:239-314) whatever the arithmetic.  The sharded solve (shard.hip) already cuts a lattice into
// slabs of rows with the ghost rows the fused M^T M needs and runs the resident kernel per slab, the slabs meeting through mailboxes:
// here the same decomposition runs with ALL slabs on the device of the handle, as ONE launch (k_cg_wg<..., RANKS>: the ranks wait for
// each other, so one grid must carry them all — P streams would share the process's few hardware queues).
//
// Geometry without geometry: the slabs are P equal, contiguous ranges of the site index (the reference numbers sites row by row:
// site = orbit + norbits (l1 + L1 l2), so a range of N / P sites is a band of rows for every P the library accepts); the ghost sites
// of a range are the contiguous hull of its M^T M dependency closure (Checkerboard.jl:57-141 walked backwards, as
// elphdynamics_amd/sharded.py: mtm_dependency_closure) — lo sites below, hi sites above, the same for every range.  Sites of the
// hull that the closure does not need carry garbage that nothing reads.  Everything is checked, nothing assumed: a bond the own sites
// depend on that leaves the slab, ghost sites that reach beyond the neighbouring range, a slab that does not compile to a 4-colour
// lane program of <= 320 sites, slabs of different shapes — any of these and the handle keeps the streaming iteration.
//
// Holstein handles, one chain, x0 = 0 (the callers' fill!(x, 0), HMC.jl:854), one right-hand side per launch.

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "elph_internal.h"

namespace {

struct SlabPtrs { double *p[ELPH_SHARD_MAXRANKS]; const int *g[ELPH_SHARD_MAXRANKS]; };

struct SlabSet {
    int P = 0, Nloc = 0, own_lo = 0, own_n = 0, hi = 0;
    int nsets = 0;                          // sets of P slab handles: set k solves right-hand side k of a pair in the same launch (own mailboxes)
    int G = 0;                              // workgroups per slab
    int ring_row = 0;                       // > 0: sites per row of the recognised square lattice — the slabs are closed into rings (add_set)
    std::vector<elph_handle_s *> hs;        // [nsets][P]
    std::vector<int *> d_g;                 // [P][Nloc] site of the parent lattice of every slab site
    void *h_args = nullptr, *d_args = nullptr;
    hipStream_t stream = nullptr;           // the stream the slab handles were last bound to
};

// dst_q[t][s] = src[t][g_q[s]] for every slab q (grid: x = site blocks, y = tau, z = slab)
__global__ void __launch_bounds__(256) k_slab_gather(SlabPtrs S, const double *__restrict__ src, int N, int Nloc) {
    const int s = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y, q = blockIdx.z;
    if (s < Nloc) S.p[q][(size_t)t * Nloc + s] = src[(size_t)t * N + S.g[q][s]];
}
// dst[t][g_q[own_lo + k]] = src_q[t][own_lo + k]
__global__ void __launch_bounds__(256) k_slab_scatter(SlabPtrs S, double *__restrict__ dst, int N, int Nloc, int own_lo, int own_n) {
    const int k = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y, q = blockIdx.z;
    if (k < own_n) dst[(size_t)t * N + S.g[q][own_lo + k]] = S.p[q][(size_t)t * Nloc + own_lo + k];
}

// sites of p that z = M^T (M p) on [start, start + n) depends on, and the bonds that carry the dependency (sharded.py: mtm_dependency_closure)
void closure(const elph_handle_s *h, int start, int n, std::vector<char> &S, std::vector<char> &need) {
    const int N = (int)h->N, nb = (int)h->nb;
    S.assign((size_t)N, 0);
    need.assign((size_t)nb, 0);
    for (int k = 0; k < n; ++k) S[(size_t)((start + k) % N)] = 1;
    auto visit = [&](int b) {
        const int i = h->h_bi[(size_t)b], j = h->h_bj[(size_t)b];
        if (S[(size_t)i] || S[(size_t)j]) { S[(size_t)i] = S[(size_t)j] = 1; need[(size_t)b] = 1; }
    };
    for (int b = 0; b < nb; ++b) visit(b);             // M^T backwards (bond 0 was applied last) ...
    for (int b = nb - 1; b >= 0; --b) visit(b);        // ... then M backwards
}

bool uniform_hop(const elph_handle_s *h) {
    for (int64_t n = 1; n < h->nb; ++n) if (h->h_c[(size_t)n] != h->h_c[0] || h->h_s[(size_t)n] != h->h_s[0]) return false;
    return h->nb > 0;
}

void free_set(SlabSet *S) {
    if (!S) return;
    for (elph_handle_s *s : S->hs) if (s) (void)elph_destroy(s);
    for (int *g : S->d_g) if (g) (void)hipFree(g);
    if (S->h_args) (void)hipHostFree(S->h_args);
    if (S->d_args) (void)hipFree(S->d_args);
    delete S;
}

// the decomposition into P ranges; false: it does not apply (reason in `why`)
bool plan(const elph_handle_s *h, int P, int &lo, int &hi, const char *&why) {
    const int N = (int)h->N;
    if (N % P) { why = "N is no multiple of the slab count"; return false; }
    const int n = N / P;
    lo = hi = 0;
    std::vector<char> S, need;
    for (int q = 0; q < P; ++q) {
        closure(h, q * n, n, S, need);
        for (int s = 0; s < N; ++s) {
            if (!S[(size_t)s]) continue;
            const int d = ((s - q * n) % N + N) % N;
            if (d < n) continue;
            const int up = d - (n - 1), down = N - d;      // distance above the last / below the first own site
            if (up <= down) hi = std::max(hi, up); else lo = std::max(lo, down);
        }
    }
    if (lo > n || hi > n) { why = "ghost sites reach beyond the neighbouring slab"; return false; }
    if (lo + n + hi >= N) { why = "a slab with its ghost sites covers the lattice"; return false; }
    if (lo + n + hi > 5 * ELPH_WAVE) { why = "slab beyond 320 sites"; return false; }
    return true;
}

// one more set of P slab handles (its own mailboxes); d_g is made with the first
int add_set(elph_handle_s *h, SlabSet *S) {
    const int N = (int)h->N, nb = (int)h->nb, P = S->P, n = S->own_n, lo = S->own_lo, hi = S->hi, Nloc = S->Nloc;
    const size_t base = S->hs.size();
    S->hs.resize(base + (size_t)P, nullptr);
    if (S->d_g.empty()) S->d_g.assign((size_t)P, nullptr);
    std::vector<unsigned char> ipc((size_t)P * ELPH_SHARD_IPC_BYTES);
    std::vector<char> Cs, need;
    int rc = ELPH_OK;
    for (int q = 0; q < P && rc == ELPH_OK; ++q) {
        std::vector<int> g((size_t)Nloc), loc((size_t)N, -1);
        for (int k = 0; k < Nloc; ++k) { g[(size_t)k] = ((q * n - lo + k) % N + N) % N; loc[(size_t)g[(size_t)k]] = k; }
        closure(h, q * n, n, Cs, need);
        std::vector<int64_t> tab;
        std::vector<double> c, s;
        // THE RING (as sharded.py: SpatialSlabs(ring=True)): on a square lattice the library recognised (row = L sites) the bonds that leave
        // the slab through its last row re-enter at its first row — the own sites do not see the difference (the ring bond lies beyond the
        // closure that fixed the ghost rows; it only stirs the outermost ghost rows, whose values nothing reads), but the slab becomes a
        // periodic rectangle in the reference's colouring, which the GRID form of the resident kernel takes (2 x 2 patches in registers
        // instead of the lane program's LDS slabs).  Kept only if it leaves the colouring alone.
        for (int pass = (S->ring_row > 0 ? 0 : 1); pass < 2; ++pass) {
            const bool ring = pass == 0;
            tab.clear(); c.clear(); s.clear();
            const int row = S->ring_row, top = g[(size_t)Nloc - 1];          // (the slab's last site)
            for (int b = 0; b < nb; ++b) {
                int i = loc[(size_t)h->h_bi[(size_t)b]], j = loc[(size_t)h->h_bj[(size_t)b]];
                if (ring && (i < 0) != (j < 0)) {
                    const int in = (i < 0) ? j : i, out = (i < 0) ? h->h_bi[(size_t)b] : h->h_bj[(size_t)b];
                    const int k = ((out - top - 1) % N + N) % N;              // the outer end is site k of the row above the slab
                    if (in >= Nloc - row && k < row) { if (i < 0) i = k; else j = k; }
                }
                if (i < 0 || j < 0) {
                    if (need[(size_t)b]) { elph_set_error("slabs: a bond the own sites depend on leaves the slab"); rc = ELPH_E_UNSUPPORTED; break; }
                    continue;
                }
                tab.push_back(i + 1); tab.push_back(j + 1);
                c.push_back(h->h_c[(size_t)b]); s.push_back(h->h_s[(size_t)b]);
            }
            if (rc || !ring) break;
            // colours = maximal runs of site-disjoint bonds (elph_create): the ring must not add one
            auto ncolours = [&](const std::vector<int64_t> &t) {
                std::vector<char> used((size_t)Nloc, 0);
                int nc = t.empty() ? 0 : 1;
                for (size_t b2 = 0; b2 + 1 < t.size(); b2 += 2) {
                    const size_t a = (size_t)t[b2] - 1, d = (size_t)t[b2 + 1] - 1;
                    if (used[a] || used[d]) { ++nc; std::fill(used.begin(), used.end(), 0); }
                    used[a] = used[d] = 1;
                }
                return nc;
            };
            std::vector<int64_t> open_tab;
            for (int b = 0; b < nb; ++b) {
                const int i = loc[(size_t)h->h_bi[(size_t)b]], j = loc[(size_t)h->h_bj[(size_t)b]];
                if (i >= 0 && j >= 0) { open_tab.push_back(i + 1); open_tab.push_back(j + 1); }
            }
            if (ncolours(tab) == ncolours(open_tab)) break;      // the ring stands
        }
        if (rc) break;
        elph_handle sh = nullptr;
        rc = elph_create(&sh, ELPH_MODEL_HOLSTEIN, Nloc, h->L, (int64_t)(tab.size() / 2), tab.data(), c.data(), s.data(), h->device);
        if (rc) break;
        S->hs[base + (size_t)q] = sh;
        sh->is_slab = true;
        rc = elph_set_stream(sh, h->stream);
        if (rc) break;
        rc = elph_i_shard_create_local(sh, q, P, lo, n, /* to prev = its ghosts above */ hi, /* to next = its ghosts below */ lo, std::max(lo, hi),
                                       ipc.data() + (size_t)q * ELPH_SHARD_IPC_BYTES);
        if (rc) break;
        if (!S->d_g[(size_t)q] && (hipMalloc((void **)&S->d_g[(size_t)q], (size_t)Nloc * sizeof(int)) != hipSuccess ||
                                   hipMemcpy(S->d_g[(size_t)q], g.data(), (size_t)Nloc * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)) {
            elph_set_error("slabs: allocation failed"); rc = ELPH_E_HIP; break;
        }
    }
    for (int q = 0; q < P && rc == ELPH_OK; ++q) rc = elph_shard_connect(S->hs[base + (size_t)q], ipc.data());
    if (rc) {
        for (size_t k = base; k < S->hs.size(); ++k) if (S->hs[k]) (void)elph_destroy(S->hs[k]);
        S->hs.resize(base);
        return rc;
    }
    ++S->nsets;
    return ELPH_OK;
}

int build(elph_handle_s *h, int P, int lo, int hi, int G, SlabSet **out) {
    SlabSet *S = new SlabSet();
    S->P = P; S->Nloc = lo + (int)h->N / P + hi; S->own_lo = lo; S->own_n = (int)h->N / P; S->hi = hi; S->G = G;
    {   // rings on a recognised square lattice whose slabs are whole rows: an even number of them, at least four, at most 64 lanes of 2 x 2 patches.
        // OPT-IN (ELPH_SLABS_RING=1): measured slower — the GRID form of the sharded kernel sits at 256 registers with scratch: 15.5 us per
        // iteration against 11.0 (24 x 24) / 12.5 (32 x 32) in the lane program (profiles/r05/slabs_resident_large_lattices.log)
        const char *er = getenv("ELPH_SLABS_RING");
        const int row = (h->pg_kind == 1) ? h->pg_L : 0;
        if ((er && er[0] == '1') && row > 0 && (int64_t)row * row == h->N && S->Nloc % row == 0 && lo % row == 0 && S->own_n % row == 0 && (row & 1) == 0) {
            const int rows = S->Nloc / row;
            if (rows >= 4 && (rows & 1) == 0 && (row / 2) * (rows / 2) <= 64 && uniform_hop(h)) S->ring_row = row;
        }
    }
    int rc = add_set(h, S);
    if (rc == ELPH_OK) {
        const size_t bytes = 2 * (size_t)P * elph_wg_rank_args_bytes();
        if (hipHostMalloc(&S->h_args, bytes, hipHostMallocDefault) != hipSuccess || hipMalloc(&S->d_args, bytes) != hipSuccess) {
            elph_set_error("slabs: allocation failed"); rc = ELPH_E_HIP;
        }
    }
    if (rc) { free_set(S); return rc; }
    S->stream = h->stream;
    *out = S;
    return ELPH_OK;
}

SlabPtrs ptrs_of(const SlabSet *S, int set, int which /* 0 d_b, 1 d_E, 2 d_x */) {
    SlabPtrs T;
    for (int q = 0; q < ELPH_SHARD_MAXRANKS; ++q) {
        T.p[q] = nullptr; T.g[q] = nullptr;
        if (q < S->P) {
            elph_handle_s *s = S->hs[(size_t)set * (size_t)S->P + (size_t)q];
            T.p[q] = which == 0 ? s->d_b : (which == 1 ? s->d_E : s->d_x);
            T.g[q] = S->d_g[(size_t)q];
        }
    }
    return T;
}

}  // namespace

void elph_i_slabs_free(elph_handle_s *h) {
    free_set(static_cast<SlabSet *>(h->slabs));
    h->slabs = nullptr;
}

// Does the resident slab form serve an un-preconditioned solve of nrhs right-hand sides on this handle?  Decided by rule from measurements
// (profiles/r05/slabs_resident_large_lattices.log; Ltau = 160, one right-hand side, us per iteration, slab form / streaming pair):
//     slab of 144..192 sites 10.6-11.0, 240..256 sites 12.5-12.7, 270..300 sites (five sites per lane) 16.9-18.5, odd slab counts 14.2-14.4;
//     streaming: 9.3 (18 x 18), 9.9 (20 x 20), 13.5 (24 x 24), 14.0 (28 x 28), 16.2 (30 x 30), 14.2 (32 x 32)
// so: lattices from 576 sites, an EVEN number of slabs of at most 256 sites each (own + ghost), the count with the smallest slab; one
// right-hand side, or two where two SETS of slabs fit the chip together (2 P G <= 240 workgroups: both solves in one launch — one launch
// after the other two right-hand sides cost 22-25 us against 15-18 streaming).
// ELPH_SLABS=0: never; =1: wherever the decomposition exists (any count, slabs up to 320 sites, up to 8 right-hand sides: the tests);
// ELPH_SLABS_P forces the slab count.
bool elph_i_slabs_usable(elph_handle_s *h, int nrhs) {
    const char *e = getenv("ELPH_SLABS");
    const int force = e ? atoi(e) : -1;
    if (force == 0 || nrhs < 1 || nrhs > (force == 1 ? 8 : 2)) return false;
    if (h->kind != ELPH_MODEL_HOLSTEIN || h->is_slab || h->shard || h->nchains != 1 || h->solo_chain >= 0 || h->dot_hi != 0 || h->wg_broken) return false;
    if (h->N <= 5 * ELPH_WAVE || !h->have_E) return false;
    if (h->slabs) {
        // two right-hand sides (the pseudofermion pair): only as ONE launch of two sets of slabs — all 2 P G workgroups resident at once;
        // one after the other they lose to the streaming pair
        const SlabSet *S = static_cast<const SlabSet *>(h->slabs);
        return nrhs == 1 || force == 1 || 2LL * S->P * S->G <= elph_i_resident_wg_limit(h);
    }
    if (h->slabs_tried) return false;
    h->slabs_tried = true;
    const char *ep = getenv("ELPH_SLABS_P");
    const int fp = ep ? atoi(ep) : 0;
    const char *why = "no slab count from 2 to 8 fits";
    if (force != 1 && fp <= 0 && h->N < 576) why = "below 576 sites the streaming iteration is faster";
    else {
        // candidates by slab size
        struct Cand { int P, lo, hi, nloc; };
        std::vector<Cand> cands;
        for (int P = (fp > 0 ? fp : 2); P <= (fp > 0 ? fp : ELPH_SHARD_MAXRANKS); ++P) {
            int lo = 0, hi = 0;
            const char *w2 = nullptr;
            if (!plan(h, P, lo, hi, w2)) { if (fp > 0 && w2) why = w2; continue; }
            const int nloc = lo + (int)h->N / P + hi;
            if (force != 1 && fp <= 0 && ((P & 1) || nloc > 4 * ELPH_WAVE)) continue;
            int W = 0, G = 0;
            if (elph_shard_shape(h->L, P, &W, &G, nullptr, nullptr) != ELPH_OK || (long long)P * G > elph_i_resident_wg_limit(h)) { why = "the slabs' workgroups cannot all be resident"; continue; }
            cands.push_back({P, lo, hi, nloc});
        }
        std::stable_sort(cands.begin(), cands.end(), [](const Cand &a, const Cand &b) { return a.nloc < b.nloc; });
        for (const Cand &c : cands) {
            SlabSet *S = nullptr;
            int W = 0, G = 0;
            (void)elph_shard_shape(h->L, c.P, &W, &G, nullptr, nullptr);
            if (build(h, c.P, c.lo, c.hi, G, &S) != ELPH_OK) { why = "a slab handle could not be made"; continue; }
            // (that every slab takes the lane-program form of the sharded kernel is checked by the launch set-up of the first solve)
            h->slabs = S;
            if (getenv("ELPH_SLABS_DEBUG")) fprintf(stderr, "[slabs] N = %lld: %d slabs of %d own + %d / %d ghost sites, %d workgroups each\n", (long long)h->N, c.P, S->own_n, c.lo, c.hi, G);
            return nrhs == 1 || force == 1 || 2LL * S->P * S->G <= elph_i_resident_wg_limit(h);
        }
    }
    if (getenv("ELPH_SLABS_DEBUG")) fprintf(stderr, "[slabs] N = %lld: not decomposed (%s)\n", (long long)h->N, why);
    return false;
}

// x = (M^T M)^-1 b for the nrhs right-hand sides in h->d_b (layout S), x0 = 0, into h->d_x; CG states into h->h_state / d_state.
// *ran = false (and ELPH_OK): the slab form gave up (a time-out: the handle cools down as after any resident kernel's) or does not apply
// after all — h->d_x is zero again and the caller runs the streaming iteration.
int elph_i_slabs_solve(elph_handle_s *h, int nrhs, const CgParams &P, long long fixed_iters, int64_t *iters, bool *ran, double *ms_out) {
    *ran = false;
    SlabSet *S = static_cast<SlabSet *>(h->slabs);
    if (!S) return ELPH_OK;
    HIPCHK(hipSetDevice(h->device));
    if (S->stream != h->stream) {                       // (elph_set_stream on the parent since the last solve)
        for (elph_handle_s *s : S->hs) { const int rc = elph_set_stream(s, h->stream); if (rc) return rc; }
        S->stream = h->stream;
    }
    const int N = (int)h->N, L = (int)h->L, Nloc = S->Nloc, Pq = S->P;
    // a pair of right-hand sides runs as two sets of slabs in ONE launch where all their workgroups are resident together
    const bool pairs = nrhs >= 2 && 2LL * Pq * S->G <= elph_i_resident_wg_limit(h);
    if (pairs && S->nsets < 2) {
        const int rc = add_set(h, S);
        if (rc) return rc;
    }
    for (elph_handle_s *s : S->hs) { const int rc = elph_i_ensure_capacity(s, 1); if (rc) return rc; }
    const dim3 gg((unsigned)((Nloc + 255) / 256), (unsigned)L, (unsigned)Pq), gs((unsigned)((S->own_n + 255) / 256), (unsigned)L, (unsigned)Pq);
    for (int k = 0; k < (pairs ? 2 : 1); ++k) {
        hipLaunchKernelGGL(k_slab_gather, gg, dim3(256), 0, h->stream, ptrs_of(S, k, 1), (const double *)h->d_E, N, Nloc);
        for (int q = 0; q < Pq; ++q) S->hs[(size_t)k * Pq + q]->have_E = true;
    }
    const char *et = getenv("ELPH_WG_TIMEOUT_MS");
    const long long timeout_ms = et ? std::max(1, atoi(et)) : 2000;
    const char *etest = getenv("ELPH_SLABS_TEST_TIMEOUT");
    const bool test_give_up = etest && etest[0] == '1';      // (tests: 1 = the launch is taken to have given up — the host side of a time-out; 2 = a real one, cg_wg.hip)
    double ms_sum = 0.0;
    for (int r = 0; r < nrhs;) {
        const int ns = (pairs && r + 1 < nrhs) ? 2 : 1;        // right-hand sides of this launch
        for (int k = 0; k < ns; ++k)
            hipLaunchKernelGGL(k_slab_gather, gg, dim3(256), 0, h->stream, ptrs_of(S, k, 0), (const double *)(h->d_b + (size_t)(r + k) * h->ndim), N, Nloc);
        HIPCHK(hipGetLastError());
        CgState st[2];
        double ms = 0.0;
        double *hist[2] = {nullptr, nullptr};
        if (P.record_hist) for (int k = 0; k < ns; ++k) hist[k] = h->d_hist + (size_t)(r + k) * (size_t)P.hist_stride;
        int rc = elph_i_shard_run_ranks(S->hs.data(), Pq, ns, S->h_args, S->d_args, P.tol, P.maxiter, P.kmax, fixed_iters, timeout_ms, st,
                                        ms_out ? &ms : nullptr, P.record_hist ? hist : nullptr, P.hist_stride);
        if (test_give_up && rc == ELPH_OK) rc = ELPH_I_ABORTED;
        if (rc == ELPH_E_UNSUPPORTED) {                  // the slabs do not take the sharded kernel's lane-program form: never again
            elph_i_slabs_free(h);
            HIPCHK(hipMemsetAsync(h->d_x, 0, (size_t)nrhs * (size_t)h->ndim * sizeof(double), h->stream));
            return ELPH_OK;
        }
        if (rc == ELPH_I_ABORTED) {
            // a time-out inside the launch (the abort word was raised — and nothing else: an event, copy or launch failure comes back as
            // ELPH_E_HIP and is returned below): cool down like the other resident kernels, streaming takes over
            h->wg_broken = true;
            const char *ec = getenv("ELPH_WG_COOLDOWN");
            h->wg_cooldown = ec ? std::max(1, atoi(ec)) : 16;
            ++h->wg_fallbacks;
            HIPCHK(hipMemsetAsync(h->d_x, 0, (size_t)nrhs * (size_t)h->ndim * sizeof(double), h->stream));
            return ELPH_OK;
        }
        if (rc) return rc;
        ms_sum += ms;
        for (int k = 0; k < ns; ++k) {
            if (!st[k].done && fixed_iters <= 0) { elph_set_error("slab CG ended without a terminal state (internal error)"); return ELPH_E_STATE; }
            hipLaunchKernelGGL(k_slab_scatter, gs, dim3(256), 0, h->stream, ptrs_of(S, k, 2), h->d_x + (size_t)(r + k) * h->ndim, N, Nloc, S->own_lo, S->own_n);
            st[k].seq = st[k].iters + 1;
            h->h_state[2 * (r + k)] = st[k];
            h->h_state[2 * (r + k) + 1] = st[k];
            if (iters) iters[r + k] = st[k].iters;
        }
        HIPCHK(hipGetLastError());
        r += ns;
    }
    HIPCHK(hipMemcpyAsync(h->d_state, h->h_state, sizeof(CgState) * 2 * (size_t)nrhs, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (ms_out) *ms_out = ms_sum;
    *ran = true;
    return ELPH_OK;
}

// (elph_bench.h) would an un-preconditioned solve of nrhs right-hand sides from x = 0 run in the slab form, and its shape
extern "C" int elph_bench_slabs_info(elph_handle h, int nrhs, int *usable, int *slabs, int *sites_per_slab, int *own_sites) {
    if (!h || !usable) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    HIPCHK(hipSetDevice(h->device));
    *usable = elph_i_slabs_usable(h, nrhs) ? 1 : 0;
    const SlabSet *S = static_cast<const SlabSet *>(h->slabs);
    if (slabs) *slabs = (*usable && S) ? S->P : 0;
    if (sites_per_slab) *sites_per_slab = (*usable && S) ? S->Nloc : 0;
    if (own_sites) *own_sites = (*usable && S) ? S->own_n : 0;
    return ELPH_OK;
}
