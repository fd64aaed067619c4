// elph_api.hip — extern "C" entry points of libelphgpu.so (include/elph_gpu.h).
// Host orchestration only: argument checks, layout staging, the CG chunk loop and the ldiv!
// flag logic (Models.jl:74-186).  All arithmetic on lattice vectors happens in kernels.hip.
// There is NO CPU fallback: without a gfx950 device every compute entry point fails.

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <numeric>


#include "elph_internal.h"
#include "pgrid_dev.h"

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------

static thread_local char g_err[512] = "";

void elph_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *elph_last_error(void) { return g_err; }
extern "C" int elph_abi_version(void) { return ELPH_ABI_VERSION; }
// the build record: written by elphdynamics_amd/build.py into a translation unit of its own at every link
extern "C" const char elph_build_info_text[];
extern "C" const char *elph_build_info(void) { return elph_build_info_text; }

extern "C" int elph_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

#define CHECK_H(h)                                    \
    do {                                              \
        if (!(h)) {                                   \
            elph_set_error("null handle");            \
            return ELPH_E_ARG;                        \
        }                                             \
        HIPCHK(hipSetDevice((h)->device));            \
    } while (0)

#define RC(call)                \
    do {                        \
        int _rc = (call);       \
        if (_rc) return _rc;    \
    } while (0)

template <class T>
static int dev_alloc(T **p, size_t n) {
    if (*p) { HIPCHK(hipFree(*p)); *p = nullptr; }
    if (n == 0) n = 1;
    HIPCHK(hipMalloc((void **)p, n * sizeof(T)));
    return ELPH_OK;
}

static void drop_graphs(elph_handle_s *h) {
    for (auto &g : h->graphs) (void)hipGraphExecDestroy(g.exec);
    h->graphs.clear();
}

// ------------------------------------------------------------------------------------------
// lane program (cg_fast.hip): bonds re-packed [colour][pass][lane]
// ------------------------------------------------------------------------------------------

void elph_lp_pack(const elph_handle_s *h, const double *per_bond, double *out, double fill) {
    const int PP = (h->npl + 1) / 2, NE = h->lp_mc * PP;
    for (int i = 0; i < NE * ELPH_WAVE; ++i) out[i] = fill;
    for (int col = 0; col < h->ncol && col < h->lp_mc; ++col) {
        const int b0 = h->h_coloff[col], b1 = h->h_coloff[col + 1];
        for (int n = b0; n < b1; ++n) {
            const int k = n - b0;
            out[(col * PP + k / ELPH_WAVE) * ELPH_WAVE + (k % ELPH_WAVE)] = per_bond[n];
        }
    }
}

static void detect_square(elph_handle_s *h);
static void detect_honeycomb12(elph_handle_s *h);
static void detect_triangular(elph_handle_s *h);

static int build_lane_program(elph_handle_s *h) {
    h->lp_mc = (h->ncol <= 4) ? 4 : 6;      // kernels exist for 4-colour (square, honeycomb, chain) and 6-colour (triangular) programs
    const int PP = (h->npl + 1) / 2, NE = h->lp_mc * PP;
    h->lp_ne = NE;
    const char *ci = getenv("ELPH_CHUNK_ITERS");
    h->chunk = ci ? atoi(ci) : ELPH_CG_CHUNK;
    if (h->chunk < 2 || (h->chunk & 1)) h->chunk = ELPH_CG_CHUNK;   // must be even (ping-pong parity)
    const char *co = getenv("ELPH_DBG_COPY_OUTSIDE");
    h->dbg_copy_outside = (co && co[0] == '1');
    const char *ct = getenv("ELPH_CHUNK_T");
    h->force_T = ct ? atoi(ct) : 0;
    const char *nf = getenv("ELPH_NO_FAST");
    h->fast = (h->ncol <= 6) && (h->npl <= ELPH_MAX_NPL) && !(nf && nf[0] == '1');
    // idle slots (ragged colours / fewer than 4 colours): each lane owns two padding slots of the LDS slab,
    // paired with (cosh, sinh) = (1, 0) by elph_lp_pack => a no-op bond, no predicate in the kernels
    h->h_lp_ij.resize((size_t)NE * ELPH_WAVE);
    for (int e = 0; e < NE; ++e)
        for (int l = 0; l < ELPH_WAVE; ++l) {
            const unsigned i = (unsigned)(h->npl * ELPH_WAVE + 2 * l);
            h->h_lp_ij[(size_t)e * ELPH_WAVE + l] = i | ((i + 1) << 16);
        }
    if (h->fast) {
        for (int col = 0; col < h->ncol; ++col) {
            const int b0 = h->h_coloff[col], b1 = h->h_coloff[col + 1];
            if (b1 - b0 > PP * ELPH_WAVE) { h->fast = false; break; }
            for (int n = b0; n < b1; ++n) {
                const int k = n - b0;
                h->h_lp_ij[(size_t)(col * PP + k / ELPH_WAVE) * ELPH_WAVE + (k % ELPH_WAVE)] =
                    (unsigned)h->h_bi[n] | ((unsigned)h->h_bj[n] << 16);
            }
        }
    }
    h->fast_capable = h->fast;
    RC(dev_alloc(&h->d_lp_ij, (size_t)NE * ELPH_WAVE));
    HIPCHK(hipMemcpy(h->d_lp_ij, h->h_lp_ij.data(), sizeof(unsigned) * NE * ELPH_WAVE, hipMemcpyHostToDevice));
    const size_t ntau = (h->kind == ELPH_MODEL_SSH) ? (size_t)h->L : 1;
    RC(dev_alloc(&h->d_lp_c, ntau * NE * ELPH_WAVE));
    RC(dev_alloc(&h->d_lp_s, ntau * NE * ELPH_WAVE));
    RC(dev_alloc(&h->d_lp_cbar, (size_t)NE * ELPH_WAVE));
    RC(dev_alloc(&h->d_lp_sbar, (size_t)NE * ELPH_WAVE));
    detect_honeycomb12(h);
    detect_square(h);
    detect_triangular(h);
    if (h->sq_L > 0) {
        RC(dev_alloc(&h->d_sq_cbar, (size_t)4 * h->N));
        RC(dev_alloc(&h->d_sq_sbar, (size_t)4 * h->N));
        RC(dev_alloc(&h->d_sq_bond, (size_t)4 * h->N));
        HIPCHK(hipMemcpy(h->d_sq_bond, h->sq_bond.data(), sizeof(int) * 4 * h->N, hipMemcpyHostToDevice));
    }
    if (h->pg_kind == 1 && h->pg_bond.size() == (size_t)4 * h->N) {      // the site -> bond map of the colouring: hopping disorder in the patch layout (pgrid.hip)
        RC(dev_alloc(&h->d_pg_bond, (size_t)4 * h->N));
        HIPCHK(hipMemcpy(h->d_pg_bond, h->pg_bond.data(), sizeof(int) * 4 * h->N, hipMemcpyHostToDevice));
    }
    return ELPH_OK;
}

// Recognise the even-L square lattice with the reference's colouring [x-even | x-odd | y-even | y-odd]
// (even L from 4 to 16, site = x + L*y).  Only then may the register-exchange forms run (sq_L; sq_P = L / 8 for L = 8, 16, the sizes with
// DPP layouts of their own; the GRID layout of cg_fast_common.h serves the others); any deviation (other lattice, other bond order,
// disordered table) leaves sq_L = sq_P = 0 and the LDS kernels are used.
static bool match_square(elph_handle_s *h, int LX, int LY) {
    h->sq_bond.assign((size_t)4 * h->N, -1);
    for (int col = 0; col < 4; ++col) {
        const int b0 = h->h_coloff[col], b1 = h->h_coloff[col + 1];
        if (b1 - b0 != h->N / 2) return false;
        for (int n = b0; n < b1; ++n) {
            const int i = h->h_bi[n], j = h->h_bj[n];
            const int xi = i % LX, yi = i / LX, xj = j % LX, yj = j / LX;
            bool ok = false;
            // expected partner of site (x,y): col 0: x^1 ; col 1: x odd -> x+1, even -> x-1 ; cols 2,3 the same along y
            auto partner = [](int x, int colpar, int L) { return colpar == 0 ? (x ^ 1) : ((x & 1) ? (x + 1) % L : (x + L - 1) % L); };
            if (col < 2) ok = (yi == yj) && (partner(xi, col, LX) == xj) && (partner(xj, col, LX) == xi);
            else ok = (xi == xj) && (partner(yi, col - 2, LY) == yj) && (partner(yj, col - 2, LY) == yi);
            if (!ok) return false;
            h->sq_bond[(size_t)col * h->N + i] = n;
            h->sq_bond[(size_t)col * h->N + j] = n;
        }
    }
    for (int v : h->sq_bond) if (v < 0) return false;
    return true;
}

// Recognise an even-L triangular lattice (site = x + L y; bonds (1,0), (0,1), (1,-1): examples/holstein_hmc_triangular.toml) in the colouring
// the checkerboard gives them: [x-even | x-odd | y-even | diagonal from even y | y-odd | diagonal from odd y], the diagonal of (x, y)
// ending at (x - 1, y + 1).  Only then may the patch-layout kernels run on it (pgrid.hip, pgrid::Tri).
static void detect_triangular(elph_handle_s *h) {
    if (h->ncol != 6 || h->nb != 3 * h->N || h->N < 16) return;
    int L = 0;
    for (int l = 4; l <= 64; l += 2) if ((int64_t)l * l == h->N) L = l;
    int px = 0, py = 0;
    if (!L || !pgrid::pick_tpatch(L, &px, &py)) return;
    auto partner = [L](int col, int x, int y, int *px_, int *py_) {
        const int xp = (x + 1) % L, xm = (x + L - 1) % L, yp = (y + 1) % L, ym = (y + L - 1) % L;
        switch (col) {
            case 0: *px_ = x ^ 1; *py_ = y; break;
            case 1: *px_ = (x & 1) ? xp : xm; *py_ = y; break;
            case 2: *px_ = x; *py_ = y ^ 1; break;
            case 3: if (!(y & 1)) { *px_ = xm; *py_ = yp; } else { *px_ = xp; *py_ = ym; } break;
            case 4: *px_ = x; *py_ = (y & 1) ? yp : ym; break;
            default: if (y & 1) { *px_ = xm; *py_ = yp; } else { *px_ = xp; *py_ = ym; } break;
        }
    };
    std::vector<char> seen((size_t)6 * h->N, 0);
    for (int col = 0; col < 6; ++col) {
        const int b0 = h->h_coloff[col], b1 = h->h_coloff[col + 1];
        if (b1 - b0 != h->N / 2) return;
        for (int n = b0; n < b1; ++n) {
            const int i = h->h_bi[n], j = h->h_bj[n];
            int qx, qy;
            partner(col, i % L, i / L, &qx, &qy);
            if (qx + L * qy != j) return;
            partner(col, j % L, j / L, &qx, &qy);
            if (qx + L * qy != i) return;
            if (seen[(size_t)col * h->N + i] || seen[(size_t)col * h->N + j]) return;
            seen[(size_t)col * h->N + i] = seen[(size_t)col * h->N + j] = 1;
        }
    }
    h->pg_L = L; h->pg_PX = px; h->pg_PY = py; h->pg_kind = 3;
}

static void detect_square(elph_handle_s *h) {
    h->sq_P = 0;
    h->sq_L = 0;
    h->sq_LX = h->sq_LY = 0;
    if (h->ncol != 4 || h->nb != 2 * h->N || h->N < 16) return;
    // candidates: the square first, then every even LX x LY with LX LY = N whose 2 x 2 patches fit the 64 lanes of a wave (the slab of a
    // sharded solve: its rows closed into a ring)
    std::vector<std::pair<int, int>> cand;
    for (int l = 4; l <= 16; l += 2) if ((int64_t)l * l == h->N) cand.push_back({l, l});
    for (int lx = 4; lx <= 32; lx += 2)
        if (h->N % lx == 0) { const int ly = (int)(h->N / lx); if (ly >= 4 && ly % 2 == 0 && lx != ly && (lx / 2) * (ly / 2) <= 64) cand.push_back({lx, ly}); }
    for (auto &c : cand) {
        if (!match_square(h, c.first, c.second)) continue;
        h->sq_LX = c.first; h->sq_LY = c.second;
        if (c.first == c.second) { h->sq_L = c.first; h->sq_P = (c.first == 8 || c.first == 16) ? c.first / 8 : 0; }
        return;
    }
    // a larger square lattice: PX x PY patches per lane (pgrid_dev.h) — the Chebyshev recursion of the preconditioner in registers
    h->pg_L = h->pg_PX = h->pg_PY = h->pg_kind = 0;
    h->pg_NW = 0;
    for (int l = 18; l <= 64; l += 2) {
        int px = 0, py = 0, nw = 1;
        if ((int64_t)l * l != h->N) continue;
        if (!pgrid::pick_patch(l, &px, &py)) {
            // no patch that fits one wavefront: several wavefronts per slice, the patch edges through LDS (ELPH_PG_MW=0: the generic kernels, A/B)
            const char *em = getenv("ELPH_PG_MW");
            if ((em && em[0] == '0') || !pgrid::pick_patch_mw(l, &px, &py, &nw)) continue;
        }
        {   // hopping disorder on 28 x 28 / 32 x 32: 2 x 2 patches on four wavefronts instead of 4 x 4 on one — 12 table entries per thread instead of 40
            // (measured, profiles/r06/hopping_disorder_patch_kernels_with_tables.log); ELPH_PG_MW=0 keeps the one-wavefront shape
            const char *em = getenv("ELPH_PG_MW");
            bool uni = true;
            if (h->kind == ELPH_MODEL_HOLSTEIN)
                for (int64_t n = 1; n < h->nb && uni; ++n) uni = (h->h_c[(size_t)n] == h->h_c[0] && h->h_s[(size_t)n] == h->h_s[0]);
            // (... and 30 x 30, whose 2 x 10 patches have no table variant: 15 x 15 threads on four wavefronts)
            if (!uni && nw == 1 && ((px == 4 && py == 4) || (px == 2 && py == 10)) && !(em && em[0] == '0')) { px = 2; py = 2; nw = ((l / 2) * (l / 2) + 63) / 64; }
            h->pg_uniform_c = uni && h->kind == ELPH_MODEL_HOLSTEIN;
        }
        if (match_square(h, l, l)) { h->pg_L = l; h->pg_PX = px; h->pg_PY = py; h->pg_kind = 1; h->pg_NW = nw; h->pg_bond = h->sq_bond; }
    }
    h->sq_bond.clear();
}

// Recognise a honeycomb lattice of L x L two-site cells (site = 2 (x + L y) + orbital; hc_L, and hc12 for L = 12) with the reference's colouring
// [A-B of a cell | B(x,y)-A(x+1,y) | B(x,y)-A(x,y+1)] (the bond definitions of examples/holstein_hmc_honeycomb.toml through
// Checkerboard.jl:471-515).  Only then may the register-exchange form of the resident CG run (cg_wg_dev.h, HcCtx).
static bool match_honeycomb(elph_handle_s *h, int LX, int LY) {
    std::vector<char> seen((size_t)3 * h->N, 0);
    for (int col = 0; col < 3; ++col) {
        const int b0 = h->h_coloff[col], b1 = h->h_coloff[col + 1];
        if (b1 - b0 != LX * LY) return false;
        for (int n = b0; n < b1; ++n) {
            int i = h->h_bi[n], j = h->h_bj[n];
            if (i & 1) std::swap(i, j);                      // i: the A site (orbital 0), j: the B site
            if ((i & 1) != 0 || (j & 1) != 1) return false;
            const int ca = i >> 1, cb = j >> 1, xa = ca % LX, ya = ca / LX, xb = cb % LX, yb = cb / LX;
            bool ok = false;
            if (col == 0) ok = (ca == cb);
            else if (col == 1) ok = (ya == yb) && (xa == (xb + 1) % LX);
            else ok = (xa == xb) && (ya == (yb + 1) % LY);
            if (!ok || seen[(size_t)col * h->N + i] || seen[(size_t)col * h->N + j]) return false;
            seen[(size_t)col * h->N + i] = seen[(size_t)col * h->N + j] = 1;
        }
    }
    return true;
}

static void detect_honeycomb12(elph_handle_s *h) {
    h->hc12 = false;
    h->hc_L = 0;
    h->hc_LX = h->hc_LY = 0;
    if (h->ncol != 3 || (h->N & 1) || h->nb != 3 * (h->N / 2) || h->N < 8) return;
    const int cells = (int)(h->N / 2);
    std::vector<std::pair<int, int>> cand;
    for (int l = 2; l <= 64; ++l) if (l * l == cells) cand.push_back({l, l});
    for (int lx = 2; lx <= 32; ++lx)
        if (cells % lx == 0) { const int ly = cells / lx; if (ly >= 2 && lx != ly) cand.push_back({lx, ly}); }
    for (auto &c : cand) {
        if (!match_honeycomb(h, c.first, c.second)) continue;
        h->hc_LX = c.first; h->hc_LY = c.second;
        if (c.first == c.second) {
            h->hc_L = c.first; h->hc12 = (c.first == 12);
            int px = 0, py = 0, nw = 1;
            const char *em = getenv("ELPH_PG_MW");
            if (c.first > 16 && pgrid::pick_hpatch(c.first, &px, &py)) { h->pg_L = c.first; h->pg_PX = px; h->pg_PY = py; h->pg_kind = 2; h->pg_NW = 1; }     // (pgrid.hip: PX x PY cells per lane)
            else if (c.first > 16 && !(em && em[0] == '0') && pgrid::pick_hpatch_mw(c.first, &px, &py, &nw)) { h->pg_L = c.first; h->pg_PX = px; h->pg_PY = py; h->pg_kind = 2; h->pg_NW = nw; }     // (several wavefronts per slice)
        }
        return;
    }
}

// uploads the lane-program copy of the per-bond cosh/sinh tables (h_c/h_s)
static int upload_lp_cs(elph_handle_s *h) {
    if (!h->fast) return ELPH_OK;
    const int NE = h->lp_ne;
    const size_t ntau = (h->kind == ELPH_MODEL_SSH) ? (size_t)h->L : 1, per = (size_t)NE * ELPH_WAVE;
    if (h->h_c.size() < ntau * (size_t)h->nb) return ELPH_OK;   // SSH before the first update_model
    std::vector<double> c(ntau * per), s(ntau * per);
    for (size_t t = 0; t < ntau; ++t) {
        elph_lp_pack(h, h->h_c.data() + t * (size_t)h->nb, c.data() + t * per, 1.0);
        elph_lp_pack(h, h->h_s.data() + t * (size_t)h->nb, s.data() + t * per, 0.0);
    }
    HIPCHK(hipMemcpy(h->d_lp_c, c.data(), sizeof(double) * c.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_lp_s, s.data(), sizeof(double) * s.size(), hipMemcpyHostToDevice));
    return ELPH_OK;
}

// ------------------------------------------------------------------------------------------
// life cycle
// ------------------------------------------------------------------------------------------

static int ensure_capacity(elph_handle_s *h, int nrhs) {
    if (nrhs <= h->cap_rhs) return ELPH_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    drop_graphs(h);
    const size_t nd = (size_t)h->ndim, c = (size_t)nrhs;
    RC(dev_alloc(&h->d_stage_in, c * nd));
    RC(dev_alloc(&h->d_stage_out, c * nd));
    RC(dev_alloc(&h->d_b, c * nd));
    RC(dev_alloc(&h->d_x, c * nd));
    RC(dev_alloc(&h->d_r, c * nd));
    RC(dev_alloc(&h->d_z, c * nd));
    RC(dev_alloc(&h->d_zp, c * nd));
    RC(dev_alloc(&h->d_tmp, c * nd));
    RC(dev_alloc(&h->d_p, 2 * c * nd));
    RC(dev_alloc(&h->d_phi, 2 * nd));
    RC(dev_alloc(&h->d_xfield, nd));
    RC(dev_alloc(&h->d_part, 4 * c * (size_t)h->L * (size_t)h->npl));
    RC(dev_alloc(&h->d_state, 2 * c));
    RC(dev_alloc(&h->d_scal, 4 * c));
    RC(dev_alloc(&h->d_alpha, c));
    const size_t Lo2 = (size_t)(h->L + 1) / 2, Lh = (size_t)h->L / 2 + 1;
    RC(dev_alloc(&h->d_nu, c * std::max(std::max(Lo2, Lh) * (size_t)h->N, nd)));
    if (h->h_state) HIPCHK(hipHostFree(h->h_state));
    if (h->h_scal) HIPCHK(hipHostFree(h->h_scal));
    HIPCHK(hipHostMalloc((void **)&h->h_state, 2 * c * sizeof(CgState), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&h->h_scal, 4 * c * sizeof(double), hipHostMallocDefault));
    HIPCHK(hipMemsetAsync(h->d_part, 0, 4 * c * (size_t)h->L * (size_t)h->npl * sizeof(double), h->stream));
    h->cap_rhs = nrhs;
    return ELPH_OK;
}

extern "C" int elph_create(elph_handle *out, int kind, int64_t nsites, int64_t ltau, int64_t nbonds,
                           const int64_t *neighbor_table, const double *cosht, const double *sinht, int device) {
    if (!out) { elph_set_error("out is null"); return ELPH_E_ARG; }
    *out = nullptr;
    if (kind != ELPH_MODEL_HOLSTEIN && kind != ELPH_MODEL_SSH) { elph_set_error("bad model kind %d", kind); return ELPH_E_ARG; }
    if (nsites < 1 || ltau < 1 || nbonds < 0) { elph_set_error("bad sizes N=%lld L=%lld nb=%lld", (long long)nsites, (long long)ltau, (long long)nbonds); return ELPH_E_ARG; }
    if (nsites > (int64_t)ELPH_MAX_SITES) {
        elph_set_error("nsites=%lld exceeds the %d sites a one-workgroup-per-slice kernel supports", (long long)nsites, ELPH_MAX_SITES);
        return ELPH_E_UNSUPPORTED;
    }
    if (nsites * ltau > (int64_t)1 << 30) { elph_set_error("ndim too large"); return ELPH_E_UNSUPPORTED; }
    // (beyond 1024 slices the tau-transforms run as dft_big.hip's two-step form, whose launches carry the slice index in gridDim.y)
    if (ltau > 65535) { elph_set_error("ltau=%lld: the time axis is limited to 65535 slices (launch geometry of the long-axis transform)", (long long)ltau); return ELPH_E_UNSUPPORTED; }
    if (nbonds > 0 && !neighbor_table) { elph_set_error("neighbor_table is null"); return ELPH_E_ARG; }
    if (kind == ELPH_MODEL_HOLSTEIN && nbonds > 0 && (!cosht || !sinht)) { elph_set_error("cosht/sinht null"); return ELPH_E_ARG; }
    for (int64_t n = 0; n < nbonds; ++n) {
        const int64_t i = neighbor_table[2 * n], j = neighbor_table[2 * n + 1];
        if (i < 1 || i > nsites || j < 1 || j > nsites || i == j) {
            elph_set_error("neighbor_table[:,%lld] = (%lld,%lld) out of range 1..%lld", (long long)n + 1, (long long)i, (long long)j, (long long)nsites);
            return ELPH_E_ARG;
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { elph_set_error("no HIP device visible"); return ELPH_E_NOGPU; }
    if (device < 0 || device >= ndev) { elph_set_error("device %d not in 0..%d", device, ndev - 1); return ELPH_E_ARG; }
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        elph_set_error("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return ELPH_E_NOGPU;
    }

    elph_handle_s *h = new elph_handle_s();
    h->kind = kind; h->device = device;
    h->N = nsites; h->L = ltau; h->nb = nbonds; h->ndim = nsites * ltau;
    h->npl = (int)((nsites + ELPH_WAVE - 1) / ELPH_WAVE);
    h->maxiter = h->ndim;   // ConjugateGradient ctor default (IterativeSolvers.jl:49-51)
    // hipGraph replay of CG chunks is opt-in (ELPH_USE_GRAPH=1).  Eager launches are within ~3 % of the replay
    // speed here (the stream stays queued ahead of the GPU: kernels take 3-6 us, a launch ~2 us), rocprofv3
    // --kernel-trace cannot trace graph replays on this image, and eager keeps bench == profile.
    const char *ug = getenv("ELPH_USE_GRAPH");
    h->use_graph = (ug && ug[0] == '1');

    // bond tables, 0-based; colours = maximal runs of site-disjoint bonds (reproduces the groups of
    // checkerboard_groups!, Checkerboard.jl:471-515, for any table in checkerboard order, and stays
    // correct — only slower — for an arbitrary bond order)
    h->h_bi.resize(nbonds); h->h_bj.resize(nbonds);
    h->h_coloff.clear(); h->h_coloff.push_back(0);
    {
        std::vector<char> used(nsites, 0);
        for (int64_t n = 0; n < nbonds; ++n) {
            const int i = (int)(neighbor_table[2 * n] - 1), j = (int)(neighbor_table[2 * n + 1] - 1);
            if (used[i] || used[j]) {
                h->h_coloff.push_back((int)n);
                std::fill(used.begin(), used.end(), 0);
            }
            used[i] = used[j] = 1;
            h->h_bi[n] = i; h->h_bj[n] = j;
        }
        if (nbonds > 0) h->h_coloff.push_back((int)nbonds);
    }
    h->ncol = (int)h->h_coloff.size() - 1;

    int rc = ELPH_OK;
    auto fail = [&](int code) { elph_destroy(h); return code; };
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { elph_set_error("stream create failed"); return fail(ELPH_E_HIP); }
    h->own_stream = true;
    if ((rc = dev_alloc(&h->d_bi, (size_t)nbonds))) return fail(rc);
    if ((rc = dev_alloc(&h->d_bj, (size_t)nbonds))) return fail(rc);
    if ((rc = dev_alloc(&h->d_coloff, h->h_coloff.size()))) return fail(rc);
    const size_t ncs = (kind == ELPH_MODEL_SSH) ? (size_t)ltau * (size_t)nbonds : (size_t)nbonds;
    if ((rc = dev_alloc(&h->d_c, ncs))) return fail(rc);
    if ((rc = dev_alloc(&h->d_s, ncs))) return fail(rc);
    const size_t nE = (kind == ELPH_MODEL_SSH) ? (size_t)nsites : (size_t)h->ndim;
    if ((rc = dev_alloc(&h->d_E, nE))) return fail(rc);
    h->E_cap = (int64_t)nE;
    if ((rc = dev_alloc(&h->d_lam, 3 * (size_t)nsites))) return fail(rc);
    if (nbonds > 0) {
        if (hipMemcpy(h->d_bi, h->h_bi.data(), sizeof(int) * nbonds, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->d_bj, h->h_bj.data(), sizeof(int) * nbonds, hipMemcpyHostToDevice) != hipSuccess) {
            elph_set_error("table upload failed");
            return fail(ELPH_E_HIP);
        }
    }
    if (hipMemcpy(h->d_coloff, h->h_coloff.data(), sizeof(int) * h->h_coloff.size(), hipMemcpyHostToDevice) != hipSuccess) {
        elph_set_error("table upload failed");
        return fail(ELPH_E_HIP);
    }
    if (kind == ELPH_MODEL_HOLSTEIN && nbonds > 0) {
        h->h_c.assign(cosht, cosht + nbonds);
        h->h_s.assign(sinht, sinht + nbonds);
        if (hipMemcpy(h->d_c, cosht, sizeof(double) * nbonds, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->d_s, sinht, sizeof(double) * nbonds, hipMemcpyHostToDevice) != hipSuccess) {
            elph_set_error("cosh/sinh upload failed");
            return fail(ELPH_E_HIP);
        }
    }
    // twiddles: tw1[m] = exp(-2 pi i m/L), m<L ; tw2[m] = exp(-i pi m/L), m<2L (exact index reduction on host)
    {
        std::vector<double2> tw1((size_t)ltau), tw2((size_t)2 * ltau);
        for (int64_t m = 0; m < ltau; ++m) {
            const double a = 2.0 * M_PI * (double)m / (double)ltau;
            tw1[m] = make_double2(cos(a), -sin(a));
        }
        for (int64_t m = 0; m < 2 * ltau; ++m) {
            const double a = M_PI * (double)m / (double)ltau;
            tw2[m] = make_double2(cos(a), -sin(a));
        }
        if ((rc = dev_alloc(&h->d_tw, (size_t)ltau))) return fail(rc);
        if ((rc = dev_alloc(&h->d_theta, (size_t)2 * ltau))) return fail(rc);
        if (hipMemcpy(h->d_tw, tw1.data(), sizeof(double2) * tw1.size(), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->d_theta, tw2.data(), sizeof(double2) * tw2.size(), hipMemcpyHostToDevice) != hipSuccess) {
            elph_set_error("twiddle upload failed");
            return fail(ELPH_E_HIP);
        }
    }
    if ((rc = elph_dft_build_tables(h))) return fail(rc);
    if ((rc = build_lane_program(h))) return fail(rc);
    if ((rc = upload_lp_cs(h))) return fail(rc);
    if ((rc = ensure_capacity(h, 1))) return fail(rc);
    if (hipStreamSynchronize(h->stream) != hipSuccess) { elph_set_error("sync failed"); return fail(ELPH_E_HIP); }
    *out = h;
    return ELPH_OK;
}

extern "C" int elph_destroy(elph_handle h) {
    if (!h) return ELPH_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    drop_graphs(h);
    elph_i_slabs_free(h);                             // (the slab handles of a large lattice: each is destroyed through here again)
    elph_shard_free(h);
    elph_hmc_free(h);
    elph_greens_free(h);
    elph_dft_mfma_free(h);
    elph_dft_big_free(h);
    void *ptrs[] = {h->d_bi, h->d_bj, h->d_coloff, h->d_c, h->d_s, h->d_E, h->d_lam, h->d_stage_in, h->d_stage_out,
                    h->d_b, h->d_x, h->d_r, h->d_z, h->d_zp, h->d_p, h->d_tmp, h->d_part, h->d_state, h->d_phi, h->d_xfield,
                    h->d_hist, h->d_scal, h->d_alpha, h->d_Ebar, h->d_cbar, h->d_sbar, h->d_order, h->d_coff, h->d_wsched, h->d_kdesc, h->d_kfold,
                    h->d_coeff, h->d_klam, h->d_ssh_x, h->d_ssh_par, h->d_ssh_tbare, h->d_ssh_bar, h->d_ssh_cb, h->d_ssh_slot, h->d_nu, h->d_tw, h->d_theta, h->d_diag, h->d_lp_ij, h->d_lp_c, h->d_lp_s, h->d_lp_cbar,
                    h->d_lp_sbar, h->d_Tk, h->d_Tt, h->d_Pk, h->d_Pt, h->d_sq_cbar, h->d_sq_sbar, h->d_sq_bond, h->d_pg_bond, h->d_res, h->d_mu_ch, h->d_kpm_start};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (h->h_state) (void)hipHostFree(h->h_state);
    if (h->h_scal) (void)hipHostFree(h->h_scal);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    for (int k = 1; k < ELPH_SPLIT_MAX; ++k)
        if (h->split_stream[k]) (void)hipStreamDestroy(h->split_stream[k]);
    if (h->split_ev) (void)hipEventDestroy(h->split_ev);
    delete h;
    return ELPH_OK;
}

extern "C" int elph_set_stream(elph_handle h, void *hip_stream) {
    CHECK_H(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    drop_graphs(h);
    if (hip_stream) {
        if (h->own_stream) { HIPCHK(hipStreamDestroy(h->stream)); h->own_stream = false; }
        h->stream = (hipStream_t)hip_stream;
    } else if (!h->own_stream) {
        HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    return ELPH_OK;
}

extern "C" int elph_synchronize(elph_handle h) {
    CHECK_H(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}

// ------------------------------------------------------------------------------------------
// update_model!
// ------------------------------------------------------------------------------------------

extern "C" int elph_update_model_holstein(elph_handle h, const double *x, const double *lambda, const double *lambda2,
                                          const double *mu, double dtau) {
    CHECK_H(h);
    if (h->kind != ELPH_MODEL_HOLSTEIN) { elph_set_error("not a Holstein handle"); return ELPH_E_ARG; }
    if (!x || !lambda || !lambda2 || !mu) { elph_set_error("null argument"); return ELPH_E_ARG; }
    const size_t N = (size_t)h->N;
    HIPCHK(hipMemcpyAsync(h->d_lam, lambda, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_lam + N, lambda2, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_lam + 2 * N, mu, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_stage_in, x, (size_t)h->ndim * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (h->nchains != 1) { h->nchains = 1; drop_graphs(h); h->kpm_ready = false; }   // expansions were per chain
    RC(elph_launch_expV(h, h->d_stage_in, dtau));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->have_E = true;
    return ELPH_OK;
}

// several independent phonon configurations (chains) resident at once: right-hand side r of a batched solve
// uses chain r % nchains.  X: nchains * ndim (reference layout, chain-major).
extern "C" int elph_update_model_holstein_chains(elph_handle h, int nchains, const double *X, const double *lambda,
                                                 const double *lambda2, const double *mu, double dtau) {
    CHECK_H(h);
    if (h->kind != ELPH_MODEL_HOLSTEIN) { elph_set_error("not a Holstein handle"); return ELPH_E_ARG; }
    if (nchains < 1 || !X || !lambda || !lambda2 || !mu) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    const size_t N = (size_t)h->N, nd = (size_t)h->ndim;
    HIPCHK(hipStreamSynchronize(h->stream));
    if ((int64_t)nchains * (int64_t)nd > h->E_cap) {
        RC(dev_alloc(&h->d_E, (size_t)nchains * nd));
        h->E_cap = (int64_t)nchains * (int64_t)nd;
    }
    drop_graphs(h);
    h->nchains = nchains;
    HIPCHK(hipMemcpyAsync(h->d_lam, lambda, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_lam + N, lambda2, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_lam + 2 * N, mu, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    // all configurations in one transfer (the staging buffer holds cap_rhs vectors), one exp kernel per chain, one sync
    RC(ensure_capacity(h, nchains));
    HIPCHK(hipMemcpyAsync(h->d_stage_in, X, (size_t)nchains * nd * sizeof(double), hipMemcpyHostToDevice, h->stream));
    for (int c = 0; c < nchains; ++c) RC(elph_launch_expV(h, h->d_stage_in + (size_t)c * nd, dtau, c));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->have_E = true;
    h->kpm_ready = false;   // the preconditioner belongs to ONE configuration
    return ELPH_OK;
}

extern "C" int elph_set_expV(elph_handle h, const double *expnDtauV) {
    CHECK_H(h);
    if (h->kind != ELPH_MODEL_HOLSTEIN) { elph_set_error("not a Holstein handle"); return ELPH_E_ARG; }
    if (!expnDtauV) { elph_set_error("null argument"); return ELPH_E_ARG; }
    HIPCHK(hipMemcpyAsync(h->d_stage_in, expnDtauV, (size_t)h->ndim * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (h->nchains != 1) { h->nchains = 1; drop_graphs(h); h->kpm_ready = false; }   // expansions were per chain
    RC(elph_launch_r2s(h, h->d_E, h->d_stage_in, 1));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->have_E = true;
    return ELPH_OK;
}

extern "C" int elph_update_model_ssh(elph_handle h, const double *cosht, const double *sinht, const double *expDtauMu) {
    CHECK_H(h);
    if (h->kind != ELPH_MODEL_SSH) { elph_set_error("not an SSH handle"); return ELPH_E_ARG; }
    if (!expDtauMu || (h->nb > 0 && (!cosht || !sinht))) { elph_set_error("null argument"); return ELPH_E_ARG; }
    const size_t L = (size_t)h->L, nb = (size_t)h->nb;
    // reference: (Ltau x Nbonds) column-major = [bond][tau]; device: tau-major [tau][bond]
    h->h_c.resize(L * nb); h->h_s.resize(L * nb);
    for (size_t n = 0; n < nb; ++n)
        for (size_t t = 0; t < L; ++t) {
            h->h_c[t * nb + n] = cosht[n * L + t];
            h->h_s[t * nb + n] = sinht[n * L + t];
        }
    if (nb > 0) {
        HIPCHK(hipMemcpy(h->d_c, h->h_c.data(), L * nb * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_s, h->h_s.data(), L * nb * sizeof(double), hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpy(h->d_E, expDtauMu, (size_t)h->N * sizeof(double), hipMemcpyHostToDevice));
    RC(upload_lp_cs(h));
    h->mu_per_chain = false;
    if (h->nchains != 1) { h->nchains = 1; drop_graphs(h); h->kpm_ready = false; }      // host tables describe ONE configuration
    h->cs_host_stale = false;
    h->have_E = true;
    return ELPH_OK;
}

// update_model!(ssh) computed ON THE DEVICE from the phonon fields (SSHModels.jl:510-562): no cosh/sinh on the host, no
// (Ltau x Nbonds) tables over PCIe.
//   x          double[nph * ltau]   ssh.x, field = (phonon-1) Ltau + tau
//   cb_index   int64[nph]           1-based checkerboard position of each phonon's bond = checkerboard_perm[phonon_to_bond[p]]
//   t_ph, alpha, alpha2  double[nph] bare hopping of that bond (ssh.t[bond]) and the couplings
//   t_bare_cb  double[nbonds]       bare hopping of EVERY bond in checkerboard order (bonds without a phonon keep it)
//   mu         double[nsites]
// A bond is driven by at most one field per tau (equivalent fields carry equal x, :548-558).
// per-phonon tables, bare hoppings, lane-program slot map and mu of an SSH handle -> device (d_ssh_*, d_lam)
int elph_i_ssh_upload_params(elph_handle_s *h, int64_t nph, const int64_t *cb_index, const double *t_ph, const double *alpha,
                             const double *alpha2, const double *t_bare_cb, const double *mu) {
    if (h->kind != ELPH_MODEL_SSH) { elph_set_error("not an SSH handle"); return ELPH_E_ARG; }
    if (nph < 0 || !mu || (h->nb > 0 && !t_bare_cb) || (nph > 0 && (!cb_index || !t_ph || !alpha || !alpha2))) {
        elph_set_error("null argument");
        return ELPH_E_ARG;
    }
    const size_t L = (size_t)h->L, nb = (size_t)h->nb, np = (size_t)nph;
    std::vector<int> cb0(np);
    for (size_t p = 0; p < np; ++p) {
        if (cb_index[p] < 1 || cb_index[p] > (int64_t)nb) { elph_set_error("cb_index[%zu] = %lld outside 1..%zu", p, (long long)cb_index[p], nb); return ELPH_E_ARG; }
        cb0[p] = (int)(cb_index[p] - 1);
    }
    if (!h->d_ssh_slot) {     // once: lane-program slot of each checkerboard bond (the layout of elph_lp_pack)
        std::vector<int> slot(std::max<size_t>(nb, 1), -1);
        if (h->fast_capable) {
            const int PP = (h->npl + 1) / 2;
            for (int col = 0; col < h->ncol && col < h->lp_mc; ++col)
                for (int n = h->h_coloff[col]; n < h->h_coloff[col + 1]; ++n) {
                    const int k = n - h->h_coloff[col];
                    slot[(size_t)n] = (col * PP + k / ELPH_WAVE) * ELPH_WAVE + (k % ELPH_WAVE);
                }
            // idle lane-program slots: the identity bond (cosh, sinh) = (1, 0) on every slice
            const size_t per = (size_t)h->lp_ne * ELPH_WAVE;
            std::vector<double> one(L * per, 1.0), zero(L * per, 0.0);
            HIPCHK(hipMemcpy(h->d_lp_c, one.data(), one.size() * sizeof(double), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(h->d_lp_s, zero.data(), zero.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        RC(dev_alloc(&h->d_ssh_slot, slot.size()));
        HIPCHK(hipMemcpy(h->d_ssh_slot, slot.data(), slot.size() * sizeof(int), hipMemcpyHostToDevice));
        RC(dev_alloc(&h->d_ssh_tbare, std::max<size_t>(nb, 1)));
    }
    if ((int64_t)np > h->ssh_nph_cap) {
        HIPCHK(hipStreamSynchronize(h->stream));
        RC(dev_alloc(&h->d_ssh_x, std::max<size_t>((size_t)h->ssh_chain_cap * np * L, 1)));
        RC(dev_alloc(&h->d_ssh_par, std::max<size_t>(3 * np, 1)));
        RC(dev_alloc(&h->d_ssh_cb, std::max<size_t>(np, 1)));
        h->ssh_nph_cap = (int64_t)np;
    }
    if (np > 0) {
        HIPCHK(hipMemcpyAsync(h->d_ssh_par, t_ph, np * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_ssh_par + np, alpha, np * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_ssh_par + 2 * np, alpha2, np * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_ssh_cb, cb0.data(), np * sizeof(int), hipMemcpyHostToDevice, h->stream));
    }
    if (nb > 0) HIPCHK(hipMemcpyAsync(h->d_ssh_tbare, t_bare_cb, nb * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_lam, mu, (size_t)h->N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));       // cb0 / the caller's arrays may go away
    h->ssh_nph = (int)np;
    return ELPH_OK;
}

extern "C" int elph_update_model_ssh_fields(elph_handle h, const double *x, int64_t nph, const int64_t *cb_index, const double *t_ph,
                                            const double *alpha, const double *alpha2, const double *t_bare_cb, const double *mu,
                                            double dtau) {
    CHECK_H(h);
    if (nph > 0 && !x) { elph_set_error("null argument"); return ELPH_E_ARG; }
    RC(elph_i_ssh_upload_params(h, nph, cb_index, t_ph, alpha, alpha2, t_bare_cb, mu));
    h->mu_per_chain = false;
    if (h->nchains != 1) { h->nchains = 1; drop_graphs(h); h->kpm_ready = false; }
    const size_t np = (size_t)nph;
    if (np > 0) HIPCHK(hipMemcpyAsync(h->d_ssh_x, x, np * (size_t)h->L * sizeof(double), hipMemcpyHostToDevice, h->stream));
    RC(elph_launch_ssh_update(h, h->d_ssh_x, (int)np, h->d_ssh_cb, h->d_ssh_par, h->d_ssh_tbare, h->d_ssh_slot, dtau));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->ssh_dtau = dtau;
    h->cs_host_stale = true;
    h->have_E = true;
    return ELPH_OK;
}

// Several independent phonon configurations (chains) of one SSH deck in one handle: X[nchains][nph*ltau]; in a batched call
// right-hand side r then uses the hopping tables of chain r % nchains (exp(dtau mu) and the couplings are the deck's, shared).
extern "C" int elph_update_model_ssh_fields_chains(elph_handle h, int nchains, const double *X, int64_t nph, const int64_t *cb_index,
                                                   const double *t_ph, const double *alpha, const double *alpha2,
                                                   const double *t_bare_cb, const double *mu, double dtau) {
    CHECK_H(h);
    if (h->kind != ELPH_MODEL_SSH) { elph_set_error("not an SSH handle"); return ELPH_E_ARG; }
    if (nchains < 1 || (nph > 0 && !X)) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    RC(elph_i_reserve_chains(h, nchains));
    RC(elph_i_ssh_upload_params(h, nph, cb_index, t_ph, alpha, alpha2, t_bare_cb, mu));
    const size_t np = (size_t)nph;
    if (np > 0) HIPCHK(hipMemcpyAsync(h->d_ssh_x, X, (size_t)nchains * np * (size_t)h->L * sizeof(double), hipMemcpyHostToDevice, h->stream));
    RC(elph_launch_ssh_update(h, h->d_ssh_x, (int)np, h->d_ssh_cb, h->d_ssh_par, h->d_ssh_tbare, h->d_ssh_slot, dtau, 0, nchains));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->ssh_dtau = dtau;
    h->cs_host_stale = true;
    h->have_E = true;
    h->kpm_ready = false;
    return ELPH_OK;
}

// model.cosht / model.sinht as the reference stores them, (Ltau x Nbonds) column-major = [bond][tau] — for callers that
// reach into those fields (KPMPreconditioners.jl:362-378) after a device-side update
extern "C" int elph_get_cosh_sinh(elph_handle h, double *cosht, double *sinht) {
    CHECK_H(h);
    if (!cosht || !sinht) { elph_set_error("null argument"); return ELPH_E_ARG; }
    const size_t L = (h->kind == ELPH_MODEL_SSH) ? (size_t)h->L : 1, nb = (size_t)h->nb;
    std::vector<double> c(L * nb), s(L * nb);
    if (nb > 0) {
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(c.data(), h->d_c, c.size() * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(s.data(), h->d_s, s.size() * sizeof(double), hipMemcpyDeviceToHost));
    }
    for (size_t n = 0; n < nb; ++n)
        for (size_t t = 0; t < L; ++t) { cosht[n * L + t] = c[t * nb + n]; sinht[n * L + t] = s[t * nb + n]; }
    return ELPH_OK;
}

// ------------------------------------------------------------------------------------------
// mul! family
// ------------------------------------------------------------------------------------------

static int need_model(elph_handle_s *h) {
    if (!h->have_E) { elph_set_error("update_model has not been called on this handle"); return ELPH_E_STATE; }
    return ELPH_OK;
}

static int mul_dev(elph_handle_s *h, int which, double *y_dev, const double *v_dev) {
    RC(need_model(h));
    if (!y_dev || !v_dev) { elph_set_error("null argument"); return ELPH_E_ARG; }
    RC(elph_launch_r2s(h, h->d_b, v_dev, 1));
    if (which == 3) {                                    // M Mᵀ v (Models.jl:229-238): two launches, not on the CG path
        RC(elph_launch_mul(h, 1, h->d_tmp, h->d_b, 1));
        RC(elph_launch_mul(h, 0, h->d_x, h->d_tmp, 1));
    } else {
        RC(elph_launch_mul(h, which, h->d_x, h->d_b, 1));
    }
    RC(elph_launch_s2r(h, y_dev, h->d_x, 1));
    return ELPH_OK;
}

static int mul_host(elph_handle_s *h, int which, double *y, const double *v) {
    if (!y || !v) { elph_set_error("null argument"); return ELPH_E_ARG; }
    const size_t bytes = (size_t)h->ndim * sizeof(double);
    HIPCHK(hipMemcpyAsync(h->d_stage_in, v, bytes, hipMemcpyHostToDevice, h->stream));
    RC(mul_dev(h, which, h->d_stage_out, h->d_stage_in));
    HIPCHK(hipMemcpyAsync(y, h->d_stage_out, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}

extern "C" int elph_mulM(elph_handle h, double *y, const double *v) { CHECK_H(h); return mul_host(h, 0, y, v); }
extern "C" int elph_mulMT(elph_handle h, double *y, const double *v) { CHECK_H(h); return mul_host(h, 1, y, v); }
extern "C" int elph_mulMTM(elph_handle h, double *y, const double *v) { CHECK_H(h); return mul_host(h, 2, y, v); }
extern "C" int elph_mulMMT(elph_handle h, double *y, const double *v) { CHECK_H(h); return mul_host(h, 3, y, v); }
extern "C" int elph_mulM_dev(elph_handle h, double *y, const double *v) { CHECK_H(h); return mul_dev(h, 0, y, v); }
extern "C" int elph_mulMT_dev(elph_handle h, double *y, const double *v) { CHECK_H(h); return mul_dev(h, 1, y, v); }
extern "C" int elph_mulMTM_dev(elph_handle h, double *y, const double *v) { CHECK_H(h); return mul_dev(h, 2, y, v); }

// ------------------------------------------------------------------------------------------
// CG driver
// ------------------------------------------------------------------------------------------

extern "C" int elph_solver_set(elph_handle h, double tol, int64_t maxiter, double kappa_max) {
    CHECK_H(h);
    if (!(tol > 0.0) || maxiter < 0 || !(kappa_max > 0.0)) { elph_set_error("bad solver parameters"); return ELPH_E_ARG; }
    h->tol = tol;
    h->maxiter = (maxiter < 1) ? h->ndim : maxiter;   // IterativeSolvers.jl:49-51
    h->kmax = kappa_max;
    return ELPH_OK;
}

static int get_chunk_graph(elph_handle_s *h, int nrhs, int use_prec, hipGraphExec_t *out) {
    for (auto &g : h->graphs)
        if (g.nrhs == nrhs && g.use_prec == use_prec) { *out = g.exec; return ELPH_OK; }
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamSynchronize(h->stream));   // drain the eager init kernels before the stream goes into capture mode
    HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    int rc = ELPH_OK;
    for (int it = 0; it < h->chunk && rc == ELPH_OK; ++it) rc = elph_launch_cg_iteration(h, nrhs, use_prec);
    if (rc == ELPH_OK && !h->dbg_copy_outside) {
        hipError_t e = hipMemcpyAsync(h->h_state, h->d_state, sizeof(CgState) * 2 * (size_t)nrhs, hipMemcpyDeviceToHost, h->stream);
        if (e != hipSuccess) { elph_set_error("capture memcpy: %s", hipGetErrorString(e)); rc = ELPH_E_HIP; }
    }
    hipError_t e = hipStreamEndCapture(h->stream, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) { elph_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return ELPH_E_HIP; }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { elph_set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return ELPH_E_HIP; }
    h->graphs.push_back({nrhs, use_prec, exec});
    *out = exec;
    return ELPH_OK;
}

// ------------------------------------------------------------------------------------------
// A large KPM-preconditioned batch as TWO half-batches on two streams.  One iteration is four dependent kernels (k_cg_ap, forward
// transform + residual update, Chebyshev recursion, inverse transform + p/x-update), each ending in a drain of the whole chip before the
// next may start, the Chebyshev one latency-bound and the transforms quantised in rounds of waves (288 right-hand sides: 2.25 rounds).
// Two independent halves, each with its own chain of kernels, fill one another's tails and run the latency-bound kernel of one under
// the HBM-bound kernels of the other.  The second half is a VIEW of the handle: a copy whose per-right-hand-side device arrays start at
// right-hand side n/2 and whose stream is the second stream — every launcher reads buffers and stream from the handle it is given, so
// nothing else changes, and each right-hand side sees exactly the arithmetic of the single-stream form (same kernels, same partial-sum
// layout per right-hand side).  Needs the p/x-fused iteration (the unfused one ping-pongs p between two slots whose distance depends on
// the batch size) and halves that hold whole groups of chains.  ELPH_SPLIT_STREAMS=0 off, =1 wherever legal; default from 192 right-hand sides.
// ------------------------------------------------------------------------------------------
__global__ void k_spin_us(long long ticks, long long *sink) {      // wall_clock64: 100 MHz
    const long long t0 = wall_clock64();
    long long t = t0;
    while (t - t0 < ticks) t = wall_clock64();
    if (sink) *sink = t;
}

// do kernels on `a` and `b` run at the same time?  (two 40-us spins: together ~40 us, one after the other ~80)
static int streams_overlap(hipStream_t a, hipStream_t b, bool *yes) {
    hipEvent_t e0 = nullptr, e1 = nullptr, eb = nullptr;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
    float best = 1e9f;
    hipError_t er = hipSuccess;
    for (int rep = 0; rep < 3 && er == hipSuccess; ++rep) {
        er = hipStreamSynchronize(a);
        if (er == hipSuccess) er = hipStreamSynchronize(b);
        if (er == hipSuccess) er = hipEventRecord(e0, a);
        if (er == hipSuccess) {
            hipLaunchKernelGGL(k_spin_us, dim3(1), dim3(1), 0, a, 4000LL, (long long *)nullptr);
            hipLaunchKernelGGL(k_spin_us, dim3(1), dim3(1), 0, b, 4000LL, (long long *)nullptr);
            er = hipEventRecord(eb, b);
        }
        if (er == hipSuccess) er = hipStreamWaitEvent(a, eb, 0);
        if (er == hipSuccess) er = hipEventRecord(e1, a);
        if (er == hipSuccess) er = hipEventSynchronize(e1);
        float ms = 0.f;
        if (er == hipSuccess) er = hipEventElapsedTime(&ms, e0, e1);
        if (er == hipSuccess && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(eb);
    if (er != hipSuccess) { elph_set_error("stream pairing probe: %s", hipGetErrorString(er)); return ELPH_E_HIP; }
    *yes = best < 0.062f;      // (40 us each: < 62 us = they overlapped)
    return ELPH_OK;
}

static int make_overlapping_stream(elph_handle_s *h, hipStream_t *out) {
    const char *ep = getenv("ELPH_SPLIT_PROBE");      // (0: take the first stream HIP hands out — the A/B)
    std::vector<hipStream_t> aside;
    hipStream_t chosen = nullptr;
    int rc = ELPH_OK;
    for (int attempt = 0; attempt < 8 && !chosen && rc == ELPH_OK; ++attempt) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { elph_set_error("hipStreamCreate failed"); rc = ELPH_E_HIP; break; }
        bool ok = true;
        if (!(ep && ep[0] == '0')) rc = streams_overlap(h->stream, s, &ok);
        if (rc == ELPH_OK && (ok || attempt == 7)) chosen = s;      // (eight in a row on the main stream's queue: take it, the form still works)
        else aside.push_back(s);
    }
    for (hipStream_t s : aside) (void)hipStreamDestroy(s);
    if (rc) { if (chosen) (void)hipStreamDestroy(chosen); return rc; }
    *out = chosen;
    return ELPH_OK;
}

struct SplitRun {
    bool on = false, ok = false, px_before = false;
    int ways = 0, n1 = 0;                                 // `ways` parts of n1 right-hand sides each; part 0 is the handle itself on its own stream
    elph_handle_s *main = nullptr, *view[ELPH_SPLIT_MAX] = {};
    // A solve that leaves through an error return must not leave kernels of the other streams in flight behind it (they write d_x, d_p, d_r,
    // d_state of their parts while the caller's next step — ldiv's zero-fill, a retry, a new solve — runs on the main stream), nor the handle
    // believing in a p/x-fused solve that never happened.
    ~SplitRun() {
        if (main) main->T_rhs_hint = 0;
        for (int k = 1; k < ways; ++k)
            if (view[k]) {
                if (on) (void)hipStreamSynchronize(view[k]->stream);
                delete view[k];
            }
        if (main && on && !ok) main->px_solve = px_before;
    }
};

// parts of the split form: ELPH_SPLIT_WAYS (2 … 8) [2]
static int split_ways() {
    const char *e = getenv("ELPH_SPLIT_WAYS");
    const int w = e ? atoi(e) : 2;
    return w < 2 ? 2 : (w > ELPH_SPLIT_MAX ? ELPH_SPLIT_MAX : w);
}

static bool split_legal(elph_handle_s *h, int nrhs, int use_prec, bool hist) {
    const int ways = split_ways();
    if (!use_prec || hist || h->use_graph || nrhs < 2 * ways || (nrhs % ways) || h->solo_chain >= 0 || h->dot_hi > 0) return false;
    const int n1 = nrhs / ways;
    if (n1 % std::max(1, h->nchains) || n1 % std::max(1, h->kpm_nch)) return false;
    const int keep = h->T_rhs_hint;
    h->T_rhs_hint = nrhs;                 // the parts choose their slices per wave for the whole batch in flight
    const bool ok = elph_px_plan(h, n1);
    h->T_rhs_hint = keep;
    return ok;
}

static bool split_wanted(elph_handle_s *h, int nrhs, int use_prec, bool hist) {
    const char *e = getenv("ELPH_SPLIT_STREAMS");
    if (e && e[0] == '0') return false;
    if (!split_legal(h, nrhs, use_prec, hist)) return false;
    // (config C, 128 right-hand sides: 130 us either way, profiles/r05/px_chunk_T.log; five sites per lane — the honeycomb lattice of config D,
    //  whose fused kernel stays at two waves per SIMD — gains from 64 on: D 128: 131 -> 116 us, 256: 262 -> 230, profiles/r05/px_fused_five_sites_per_lane.log;
    //  honeycomb 16 x 16 cells, eight sites per lane: 64 right-hand sides 64 -> 55 us, 256: 159 -> 145; D at 64: 89 -> 87)
    // (hopping disorder on 4 x 4 patches: the table variants of the patch kernels hold 40 KB of LDS per wavefront — two half-batches side by side lose to one
    //  stream: 32 x 32 at 96 right-hand sides 763 against 732 us, 28 x 28 695 against 663; profiles/r06/hopping_disorder_patch_kernels_with_tables.log)
    if (!(e && e[0] == '1') && h->pg_L > 0 && h->pg_PX * h->pg_PY >= 16 && !h->pg_uniform && elph_pg_disorder_ok(h)) return false;
    return (e && e[0] == '1') || nrhs >= (h->npl >= 5 ? 64 : 192);
}

// after elph_launch_cg_init(h, nrhs, 1, …) on the main stream
static int split_begin(elph_handle_s *h, int nrhs, SplitRun &S) {
    S.ways = split_ways();
    S.n1 = nrhs / S.ways;
    S.main = h;
    S.view[0] = h;
    S.px_before = h->px_solve;
    // The parts must run on DIFFERENT hardware queues.  HIP maps its streams onto a few hardware queues per process (four by default,
    // GPU_MAX_HW_QUEUES) by a rule of its own: in a process that holds several handles a part's stream can share the queue of the handle's main
    // stream — the parts then run one after the other, slower than one stream (round 6: the second handle of a process, 32 x 32 at 72 right-hand
    // sides: 437 us instead of 353; a stream priority of its own did not change the mapping).  So the pairing is MEASURED once per stream: a spin
    // kernel on each of the two streams at the same time — two that overlap finish in the time of one; a candidate that does not is set aside
    // (kept alive until the choice is made, so that the next candidate does not inherit its queue) and the next one is tried.
    for (int k = 1; k < S.ways; ++k)
        if (!h->split_stream[k]) RC(make_overlapping_stream(h, &h->split_stream[k]));
    if (!h->split_ev) HIPCHK(hipEventCreateWithFlags(&h->split_ev, hipEventDisableTiming));
    h->px_solve = true;                                   // (split_legal: the parts run p/x-fused whatever the whole batch would have run)
    h->px_via_pg = h->fast && h->lp_mc != 4;              // (six-colour lane programs: the patch-form pair, kernels.hip: px_plan)
    h->T_rhs_hint = nrhs;                                 // slices per wave for the right-hand sides in flight = all parts
    HIPCHK(hipEventRecord(h->split_ev, h->stream));       // the start state (x0, r0, p0, rho0 of every right-hand side) is on the main stream
    const size_t nd = (size_t)h->ndim, Lo2 = (size_t)(h->L + 1) / 2, nrz = (size_t)h->L * (size_t)h->npl;
    for (int k = 1; k < S.ways; ++k) {
        elph_handle_s *v = S.view[k] = new elph_handle_s(*h);
        const size_t r0 = (size_t)k * (size_t)S.n1;
        v->d_x += r0 * nd; v->d_r += r0 * nd; v->d_z += r0 * nd; v->d_zp += r0 * nd; v->d_b += r0 * nd; v->d_tmp += r0 * nd; v->d_p += r0 * nd;
        v->d_nu += r0 * Lo2 * (size_t)h->N;
        v->d_state += 2 * r0; v->h_state += 2 * r0; v->d_alpha += r0;
        v->d_part += r0 * nrz;    // the three partial-sum arrays lie cap_rhs * nrz apart and are indexed [rhs][<= nrz]: one offset serves all
        v->stream = h->split_stream[k];
        v->graphs.clear();
        HIPCHK(hipStreamWaitEvent(v->stream, h->split_ev, 0));
    }
    S.on = true;
    return ELPH_OK;
}

// one iteration of every part, each on its stream
static int split_iteration(SplitRun &S, int use_prec) {
    for (int k = 0; k < S.ways; ++k) RC(elph_launch_cg_iteration(S.view[k], S.n1, use_prec));
    return ELPH_OK;
}

// the main stream continues only after the other parts have finished
static int split_join(SplitRun &S) {
    elph_handle_s *h = S.main;
    for (int k = 1; k < S.ways; ++k) {
        HIPCHK(hipEventRecord(h->split_ev, S.view[k]->stream));
        HIPCHK(hipStreamWaitEvent(h->stream, h->split_ev, 0));
    }
    return ELPH_OK;
}

// Runs CG on d_b / d_x (layout S) for nrhs right-hand sides.  Returns per-rhs iteration counts.
static int run_cg(elph_handle_s *h, int nrhs, int use_prec, double tol, int64_t maxiter, double kmax, int64_t *iters,
                  double *eps_hist /* host, optional, nrhs*(maxiter+1) */) {
    const bool x0_zero = h->x_zero;        // the hint belongs to THIS solve: consumed before anything can return
    h->x_zero = false;
    if (use_prec && !h->kpm_ready) { elph_set_error("preconditioned solve requested before elph_kpm_setup"); return ELPH_E_STATE; }
    CgParams P;
    P.tol = tol; P.kmax = kmax; P.maxiter = maxiter; P.use_prec = use_prec;
    P.record_hist = eps_hist ? 1 : 0;
    P.hist_stride = maxiter + 1;
    if (eps_hist) {
        const int64_t need = (int64_t)nrhs * (maxiter + 1);
        if (need > h->hist_cap) {
            HIPCHK(hipStreamSynchronize(h->stream));
            drop_graphs(h);
            RC(dev_alloc(&h->d_hist, (size_t)need));
            h->hist_cap = need;
        }
    }
    if (memcmp(&P, &h->cur_params, sizeof(P)) != 0) drop_graphs(h);   // parameters are baked into captured launches
    h->cur_params = P;
    // one solve of the cool-down after a resident kernel timed out — counted only for solves that WOULD have taken a resident kernel (either
    // kernel, either kind of solve): a stream of large streaming batches in between does not bring the retry forward, and a solve that
    // never launches a resident kernel does not clear the abort word
    if (h->wg_broken) {
        bool eligible = false;
        if (!use_prec) eligible = h->fast && maxiter >= 1 && elph_wg_usable(h, nullptr, nullptr, nullptr, nrhs);
        else { h->wg_broken = false; eligible = maxiter >= 1 && elph_pcg_wg_usable(h, nrhs); h->wg_broken = true; }      // (its shape test reads the flag itself)
        if (!use_prec && !eligible && h->slabs && x0_zero && maxiter >= 1) {      // (the slab form of a large lattice, slabs.hip)
            h->wg_broken = false; eligible = elph_i_slabs_usable(h, nrhs); h->wg_broken = true;
        }
        if (eligible) RC(elph_wg_cooldown_step(h));
    }
    RC(elph_launch_cg_init(h, nrhs, use_prec, x0_zero));      // (x0 = 0: A x0 = 0 without the mat-vec)
    h->wg_x0_zero = h->x_zero_seen;

    // whole solve in one launch with the Krylov vectors in registers (cg_wg.hip: k_cg_wg) when it applies
    if (h->fast && !use_prec && maxiter >= 1 && elph_wg_usable(h, nullptr, nullptr, nullptr, nrhs)) {
        bool ran = false;
        CgBufs B = elph_make_bufs(h, nrhs);
        B.params = P;
        // (elph_wg_cg saves the caller's initial guess in d_zp — unused by an un-preconditioned solve — once it has decided to launch)
        RC(elph_wg_cg(h, B, nrhs, 0, &ran));
        if (ran) {
            HIPCHK(hipMemcpyAsync(h->h_state, h->d_state, sizeof(CgState) * 2 * (size_t)nrhs, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            bool aborted = false;
            RC(elph_wg_aborted(h, &aborted));
            if (!aborted) {
                for (int r = 0; r < nrhs; ++r) {
                    if (!h->h_state[2 * r].done) { elph_set_error("workgroup-resident CG ended without a terminal state (internal error)"); return ELPH_E_STATE; }
                    iters[r] = h->h_state[2 * r].iters;
                }
                if (eps_hist) {
                    HIPCHK(hipMemcpyAsync(eps_hist, h->d_hist, sizeof(double) * (size_t)nrhs * (size_t)(maxiter + 1), hipMemcpyDeviceToHost, h->stream));
                    HIPCHK(hipStreamSynchronize(h->stream));
                }
                return ELPH_OK;
            }
            // a team gave up (x, r of the right-hand sides that had finished were overwritten): start every right-hand side again
            // from the caller's initial guess; the two-kernel iteration below does the work
            HIPCHK(hipMemcpyAsync(h->d_x, h->d_zp, (size_t)nrhs * (size_t)h->ndim * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            RC(elph_launch_cg_init(h, nrhs, use_prec));
        }
    }

    // a lattice beyond one wave's slice: the resident kernel on slabs of rows of the lattice, all on this device, one launch per right-hand
    // side (slabs.hip).  x0 = 0 only (the slab kernel starts from it): ldiv!'s zero-fill.
    if (!use_prec && maxiter >= 1 && x0_zero && elph_i_slabs_usable(h, nrhs)) {
        bool ran = false;
        RC(elph_i_slabs_solve(h, nrhs, P, 0, iters, &ran, nullptr));
        if (ran) {
            if (eps_hist) {
                HIPCHK(hipMemcpyAsync(eps_hist, h->d_hist, sizeof(double) * (size_t)nrhs * (size_t)(maxiter + 1), hipMemcpyDeviceToHost, h->stream));
                HIPCHK(hipStreamSynchronize(h->stream));
            }
            return ELPH_OK;
        }
        RC(elph_launch_cg_init(h, nrhs, use_prec, true));      // (x is zero again: elph_i_slabs_solve)
    }

    // the whole PRECONDITIONED solve in one launch (pcg_wg.hip: k_pcg_wg) for one to eight right-hand sides on the 16 x 16 square lattice
    if (use_prec && maxiter >= 1 && elph_pcg_wg_usable(h, nrhs)) {
        bool ran = false;
        CgBufs B = elph_make_bufs(h, nrhs);
        B.params = P;
        const size_t xbytes = (size_t)nrhs * (size_t)h->ndim * sizeof(double);
        HIPCHK(hipMemcpyAsync(h->d_tmp, h->d_x, xbytes, hipMemcpyDeviceToDevice, h->stream));     // the initial guess, for the fallback (A x0 in d_tmp has been consumed)
        RC(elph_pcg_wg(h, B, nrhs, 0, &ran));
        if (ran) {
            HIPCHK(hipMemcpyAsync(h->h_state, h->d_state, sizeof(CgState) * 2 * (size_t)nrhs, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            bool aborted = false;
            RC(elph_wg_aborted(h, &aborted));
            if (!aborted) {
                for (int r = 0; r < nrhs; ++r) {
                    if (!h->h_state[2 * r].done) { elph_set_error("resident preconditioned CG ended without a terminal state (internal error)"); return ELPH_E_STATE; }
                    iters[r] = h->h_state[2 * r].iters;
                }
                if (eps_hist) {
                    HIPCHK(hipMemcpyAsync(eps_hist, h->d_hist, sizeof(double) * (size_t)nrhs * (size_t)(maxiter + 1), hipMemcpyDeviceToHost, h->stream));
                    HIPCHK(hipStreamSynchronize(h->stream));
                }
                return ELPH_OK;
            }
            HIPCHK(hipMemcpyAsync(h->d_x, h->d_tmp, xbytes, hipMemcpyDeviceToDevice, h->stream));
            RC(elph_launch_cg_init(h, nrhs, use_prec));
        }
    }

    const int64_t max_chunks = (maxiter + 1 + h->chunk - 1) / h->chunk + 1;
    bool all_done = false;
    if (split_wanted(h, nrhs, use_prec, eps_hist != nullptr)) {
        SplitRun S;
        RC(split_begin(h, nrhs, S));
        for (int64_t c = 0; c < max_chunks && !all_done; ++c) {
            for (int it = 0; it < h->chunk; ++it) RC(split_iteration(S, use_prec));
            for (int k = 0; k < S.ways; ++k)
                HIPCHK(hipMemcpyAsync(S.view[k]->h_state, S.view[k]->d_state, sizeof(CgState) * 2 * (size_t)S.n1, hipMemcpyDeviceToHost, S.view[k]->stream));
            for (int k = 0; k < S.ways; ++k) HIPCHK(hipStreamSynchronize(S.view[k]->stream));
            all_done = true;
            for (int r = 0; r < nrhs; ++r) {
                const CgState &a = h->h_state[2 * r], &b = h->h_state[2 * r + 1];
                const CgState &s = (b.seq > a.seq) ? b : a;
                if (!s.done) all_done = false;
                else iters[r] = s.iters;
            }
        }
        h->ap_count = S.view[S.ways - 1]->ap_count;
        S.ok = all_done;
        if (!all_done) { elph_set_error("CG chunk loop (two streams) ended without a terminal state (internal error)"); return ELPH_E_STATE; }
        return ELPH_OK;
    }
    for (int64_t c = 0; c < max_chunks && !all_done; ++c) {
        if (h->use_graph) {
            hipGraphExec_t exec;
            RC(get_chunk_graph(h, nrhs, use_prec, &exec));
            HIPCHK(hipGraphLaunch(exec, h->stream));
            h->ap_count += h->chunk;   // even: the captured launch parities stay aligned with seq
            if (h->dbg_copy_outside)
                HIPCHK(hipMemcpyAsync(h->h_state, h->d_state, sizeof(CgState) * 2 * (size_t)nrhs, hipMemcpyDeviceToHost, h->stream));
        } else {
            for (int it = 0; it < h->chunk; ++it) RC(elph_launch_cg_iteration(h, nrhs, use_prec));
            HIPCHK(hipMemcpyAsync(h->h_state, h->d_state, sizeof(CgState) * 2 * (size_t)nrhs, hipMemcpyDeviceToHost, h->stream));
        }
        HIPCHK(hipStreamSynchronize(h->stream));
        all_done = true;
        for (int r = 0; r < nrhs; ++r) {
            const CgState &a = h->h_state[2 * r], &b = h->h_state[2 * r + 1];
            const CgState &s = (b.seq > a.seq) ? b : a;
            if (!s.done) all_done = false;
            else iters[r] = s.iters;
        }
    }
    if (!all_done) { elph_set_error("CG chunk loop ended without a terminal state (internal error)"); return ELPH_E_STATE; }
    if (eps_hist) {
        HIPCHK(hipMemcpyAsync(eps_hist, h->d_hist, sizeof(double) * (size_t)nrhs * (size_t)(maxiter + 1), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return ELPH_OK;
}

// residual + flag logic of ldiv! for rhs already solved into d_x; zeroes x where flag > 0
static int residual_and_flags(elph_handle_s *h, int nrhs, const int64_t *iters, int64_t cmp_maxiter, double *resid, int *flag) {
    RC(elph_launch_residual(h, nrhs));
    HIPCHK(hipMemcpyAsync(h->h_scal, h->d_scal, sizeof(double) * (size_t)nrhs, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int r = 0; r < nrhs; ++r) {
        resid[r] = h->h_scal[r];
        if (resid[r] > sqrt(h->tol)) {           // Models.jl:100,157 (NaN compares false, as in the reference)
            flag[r] = (iters[r] == cmp_maxiter) ? 1 : 2;
            RC(elph_launch_zero(h, h->d_x + (size_t)r * (size_t)h->ndim, h->ndim));   // fill!(x,0)
        } else {
            flag[r] = 0;
        }
    }
    return ELPH_OK;
}

// ldiv! on device-resident layout-S d_b/d_x
static int ldiv_core(elph_handle_s *h, int nrhs, int use_prec, int64_t maxiter, int64_t *iters, double *resid, int *flag) {
    RC(need_model(h));
    if (maxiter == 0) maxiter = h->maxiter;                                  // Models.jl:78-80,143-145
    if (!use_prec) {
        RC(run_cg(h, nrhs, 0, h->tol, maxiter, h->kmax, iters, nullptr));
        RC(residual_and_flags(h, nrhs, iters, h->maxiter, resid, flag));     // Models.jl:160 compares solver.maxiter
        return ELPH_OK;
    }
    RC(run_cg(h, nrhs, 1, h->tol, maxiter, h->kmax, iters, nullptr));
    RC(residual_and_flags(h, nrhs, iters, maxiter, resid, flag));            // Models.jl:103 compares maxiter
    // failed right-hand sides: retry without preconditioner, 10x maxiter, from x = 0 (Models.jl:129-133)
    for (int r = 0; r < nrhs; ++r) {
        if (flag[r] == 0) continue;
        const size_t nd = (size_t)h->ndim;
        if (r != 0) {
            // park rhs 0, move rhs r into slot 0, solve single, move back
            HIPCHK(hipMemcpyAsync(h->d_tmp, h->d_x, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->d_z, h->d_b, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->d_x, h->d_x + r * nd, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->d_b, h->d_b + r * nd, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->d_stage_out, h->d_tmp, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->d_stage_in, h->d_z, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        }
        int64_t it1 = 0; double rs1 = 0; int fl1 = 0;
        if (h->nchains > 1) { h->solo_chain = r % h->nchains; drop_graphs(h); }   // slot 0 must keep rhs r's fermion matrix
        int rc1 = run_cg(h, 1, 0, h->tol, 10 * maxiter, h->kmax, &it1, nullptr);
        if (rc1 == ELPH_OK) rc1 = residual_and_flags(h, 1, &it1, h->maxiter, &rs1, &fl1);
        if (h->solo_chain >= 0) { h->solo_chain = -1; drop_graphs(h); }
        RC(rc1);
        iters[r] = it1; resid[r] = rs1; flag[r] = fl1;
        if (r != 0) {
            HIPCHK(hipMemcpyAsync(h->d_x + r * nd, h->d_x, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->d_x, h->d_stage_out, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->d_b, h->d_stage_in, nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        }
    }
    return ELPH_OK;
}

int elph_i_ldiv_core(elph_handle_s *h, int nrhs, int use_prec, int64_t maxiter, int64_t *iters, double *resid, int *flag) {
    return ldiv_core(h, nrhs, use_prec, maxiter, iters, resid, flag);
}
int elph_i_ensure_capacity(elph_handle_s *h, int nrhs) { return ensure_capacity(h, nrhs); }
// SSH: one set of hopping tables (tau-major cosh/sinh + their lane-program copies) and one field buffer per chain
static int ssh_reserve_chains(elph_handle_s *h, int nchains) {
    if (nchains <= h->ssh_chain_cap) return ELPH_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    drop_graphs(h);
    const size_t L = (size_t)h->L, nb = (size_t)h->nb, per = (size_t)h->lp_ne * ELPH_WAVE, nc = (size_t)nchains;
    RC(dev_alloc(&h->d_c, nc * L * nb));
    RC(dev_alloc(&h->d_s, nc * L * nb));
    if (h->fast_capable) {
        RC(dev_alloc(&h->d_lp_c, nc * L * per));
        RC(dev_alloc(&h->d_lp_s, nc * L * per));
        // idle lane-program slots: the identity bond (cosh, sinh) = (1, 0) on every slice of every chain
        std::vector<double> one(nc * L * per, 1.0);
        HIPCHK(hipMemcpy(h->d_lp_c, one.data(), one.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemset(h->d_lp_s, 0, one.size() * sizeof(double)));
    }
    if (h->ssh_nph_cap > 0) RC(dev_alloc(&h->d_ssh_x, nc * (size_t)h->ssh_nph_cap * L));
    RC(dev_alloc(&h->d_E, nc * (size_t)h->N));          // exp(dtau mu) per chain
    h->E_cap = (int64_t)(nc * (size_t)h->N);
    h->ssh_chain_cap = nchains;
    h->have_E = false;          // the tables are empty until the next update_model!
    h->cs_host_stale = true;
    return ELPH_OK;
}

int elph_i_reserve_chains(elph_handle_s *h, int nchains) {
    if (nchains < 1) { elph_set_error("nchains < 1"); return ELPH_E_ARG; }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->kind == ELPH_MODEL_SSH) {
        RC(ssh_reserve_chains(h, nchains));
        if (h->nchains != nchains) { h->nchains = nchains; drop_graphs(h); h->kpm_ready = false; }
        return ELPH_OK;
    }
    const int64_t need = (int64_t)nchains * h->ndim;
    if (need > h->E_cap) {
        RC(dev_alloc(&h->d_E, (size_t)need));
        h->E_cap = need;
        h->have_E = false;
    }
    if (h->nchains != nchains) { h->nchains = nchains; drop_graphs(h); h->kpm_ready = false; }
    return ELPH_OK;
}
void elph_i_drop_graphs(elph_handle_s *h) { drop_graphs(h); }

static int stage_in_dev(elph_handle_s *h, int nrhs, const double *X_dev, const double *B_dev) {
    RC(ensure_capacity(h, nrhs));
    RC(elph_launch_r2s(h, h->d_b, B_dev, nrhs));
    RC(elph_launch_r2s(h, h->d_x, X_dev, nrhs));
    h->x_zero = false;                              // (a caller's initial guess)
    return ELPH_OK;
}

// Is the caller's initial guess all zeros?  Looked at only where it decides something: a lattice beyond one wave's slice whose
// un-preconditioned solve of one or two right-hand sides the slab form (slabs.hip; it starts from x = 0) could take.  Stops at the first
// non-zero; a zero guess costs one pass over the vector on the host (1.3 MB at 32 x 32 sites, 160 slices) — against a solve of milliseconds.
static bool x0_is_zero(elph_handle_s *h, int nrhs, int use_prec, const double *X) {
    if (use_prec || h->have_E == false) return false;
    const bool prev = h->wg_broken;
    h->wg_broken = false;                            // (cooling down: the hint still lets run_cg count the solve)
    const bool cand = elph_i_slabs_usable(h, nrhs);
    h->wg_broken = prev;
    if (!cand) return false;
    const size_t n = (size_t)nrhs * (size_t)h->ndim;
    for (size_t i = 0; i < n; ++i) if (X[i] != 0.0) return false;
    return true;
}

extern "C" int elph_ldiv_batched_dev(elph_handle h, int nrhs, double *X_dev, const double *B_dev, int use_prec,
                                     int64_t maxiter, int64_t *iters, double *residual_error, int *flag) {
    CHECK_H(h);
    if (nrhs < 1 || !X_dev || !B_dev || !iters || !residual_error || !flag || maxiter < 0) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    RC(stage_in_dev(h, nrhs, X_dev, B_dev));
    RC(ldiv_core(h, nrhs, use_prec, maxiter, iters, residual_error, flag));
    RC(elph_launch_s2r(h, X_dev, h->d_x, nrhs));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}

extern "C" int elph_ldiv_dev(elph_handle h, double *x_dev, const double *b_dev, int use_prec, int64_t maxiter,
                             int64_t *iters, double *residual_error, int *flag) {
    return elph_ldiv_batched_dev(h, 1, x_dev, b_dev, use_prec, maxiter, iters, residual_error, flag);
}

extern "C" int elph_ldiv_batched(elph_handle h, int nrhs, double *X, const double *B, int use_prec, int64_t maxiter,
                                 int64_t *iters, double *residual_error, int *flag) {
    CHECK_H(h);
    if (nrhs < 1 || !X || !B || !iters || !residual_error || !flag || maxiter < 0) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    RC(ensure_capacity(h, nrhs));
    const size_t bytes = (size_t)nrhs * (size_t)h->ndim * sizeof(double);
    HIPCHK(hipMemcpyAsync(h->d_stage_in, B, bytes, hipMemcpyHostToDevice, h->stream));
    RC(elph_launch_r2s(h, h->d_b, h->d_stage_in, nrhs));
    if (x0_is_zero(h, nrhs, use_prec, X)) {          // fill!(x, 0) of the callers (HMC.jl:854, GreensFunctions.jl:334): seen on the host, the slab form may take the solve
        HIPCHK(hipMemsetAsync(h->d_x, 0, bytes, h->stream));
        h->x_zero = true;
    } else {
        HIPCHK(hipMemcpyAsync(h->d_stage_in, X, bytes, hipMemcpyHostToDevice, h->stream));
        RC(elph_launch_r2s(h, h->d_x, h->d_stage_in, nrhs));
        h->x_zero = false;                          // (a caller's initial guess)
    }
    RC(ldiv_core(h, nrhs, use_prec, maxiter, iters, residual_error, flag));
    RC(elph_launch_s2r(h, h->d_stage_out, h->d_x, nrhs));
    HIPCHK(hipMemcpyAsync(X, h->d_stage_out, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}

extern "C" int elph_ldiv(elph_handle h, double *x, const double *b, int use_prec, int64_t maxiter, int64_t *iters,
                         double *residual_error, int *flag) {
    return elph_ldiv_batched(h, 1, x, b, use_prec, maxiter, iters, residual_error, flag);
}

extern "C" int elph_cg_solve(elph_handle h, double *x, const double *b, double tol, int64_t maxiter, double kappa_max,
                             int use_precond, int64_t *iters, double *eps_hist) {
    CHECK_H(h);
    RC(need_model(h));
    if (!x || !b || !iters || maxiter < 0) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    if (maxiter == 0) maxiter = h->maxiter;       // iszero() defaults, IterativeSolvers.jl:160-173
    if (tol == 0.0) tol = h->tol;
    if (kappa_max == 0.0) kappa_max = h->kmax;
    const size_t bytes = (size_t)h->ndim * sizeof(double);
    HIPCHK(hipMemcpyAsync(h->d_stage_in, b, bytes, hipMemcpyHostToDevice, h->stream));
    RC(elph_launch_r2s(h, h->d_b, h->d_stage_in, 1));
    if (x0_is_zero(h, 1, use_precond, x)) {
        HIPCHK(hipMemsetAsync(h->d_x, 0, bytes, h->stream));
        h->x_zero = true;
    } else {
        HIPCHK(hipMemcpyAsync(h->d_stage_in, x, bytes, hipMemcpyHostToDevice, h->stream));
        RC(elph_launch_r2s(h, h->d_x, h->d_stage_in, 1));
        h->x_zero = false;                          // (a caller's initial guess)
    }
    RC(run_cg(h, 1, use_precond ? 1 : 0, tol, maxiter, kappa_max, iters, eps_hist));
    RC(elph_launch_s2r(h, h->d_stage_out, h->d_x, 1));
    HIPCHK(hipMemcpyAsync(x, h->d_stage_out, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}

// (internal: shard.hip) Sites [site_lo, site_hi) (0-based) enter the inner products of the streaming CG (p.z, r.r, b.b); the other
// sites of the handle's lattice are ghost sites of a spatial shard: they take part in the mat-vec, their values come from the
// neighbouring ranks.  A restricted range runs the generic kernel family.  (0, nsites) restores the default.
int elph_i_set_dot_range(elph_handle_s *h, int64_t site_lo, int64_t site_hi) {
    if (site_lo < 0 || site_hi > h->N || site_lo >= site_hi) { elph_set_error("bad site range [%lld, %lld)", (long long)site_lo, (long long)site_hi); return ELPH_E_ARG; }
    HIPCHK(hipStreamSynchronize(h->stream));
    drop_graphs(h);
    const bool all = (site_lo == 0 && site_hi == h->N);
    h->dot_lo = all ? 0 : (int)site_lo;
    h->dot_hi = all ? 0 : (int)site_hi;
    h->fast = all ? h->fast_capable : false;
    return ELPH_OK;
}

// ------------------------------------------------------------------------------------------
// fermion force: update_model! + calc_O⁻¹Λϕ! + calc_dSfdx! with x, phi± , X± resident
// ------------------------------------------------------------------------------------------

extern "C" int elph_fermion_force_holstein(elph_handle h, const double *x, const double *lambda, const double *lambda2,
                                           const double *mu, double dtau, const double *phi_plus, const double *phi_minus,
                                           int use_precond, double tol_power, double *dSfdx, double *Xp_out, double *Xm_out,
                                           int64_t *iters, int *flag) {
    CHECK_H(h);
    if (h->kind != ELPH_MODEL_HOLSTEIN) { elph_set_error("not a Holstein handle"); return ELPH_E_ARG; }
    if (!x || !lambda || !lambda2 || !mu || !phi_plus || !phi_minus || !dSfdx || !iters || !flag) { elph_set_error("null argument"); return ELPH_E_ARG; }
    RC(ensure_capacity(h, 2));
    if (h->nchains != 1) { h->nchains = 1; drop_graphs(h); h->kpm_ready = false; }   // expansions were per chain
    const size_t nd = (size_t)h->ndim, N = (size_t)h->N, bytes = nd * sizeof(double);
    // update_model! (HolsteinModels.jl:526-549) and x in layout S
    HIPCHK(hipMemcpyAsync(h->d_lam, lambda, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_lam + N, lambda2, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_lam + 2 * N, mu, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_stage_in, x, bytes, hipMemcpyHostToDevice, h->stream));
    RC(elph_launch_expV(h, h->d_stage_in, dtau));
    RC(elph_launch_r2s(h, h->d_xfield, h->d_stage_in, 1));
    h->have_E = true;
    // phi± -> layout S; b± = Λ phi± (HMC.jl:840-842)
    HIPCHK(hipMemcpyAsync(h->d_stage_in, phi_plus, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_stage_in + nd, phi_minus, bytes, hipMemcpyHostToDevice, h->stream));
    RC(elph_launch_r2s(h, h->d_phi, h->d_stage_in, 2));
    RC(elph_launch_lambda_rhs(h, h->d_b, h->d_phi, h->d_xfield, dtau));
    HIPCHK(hipMemsetAsync(h->d_x, 0, 2 * bytes, h->stream));                      // fill!(O⁻¹Λϕ, 0)  (HMC.jl:854,883)
    h->x_zero = true;
    // the two solves as one batch at tol^power (HMC.jl:827-828, restored below as at :912)
    const double tol0 = h->tol;
    h->tol = pow(tol0, tol_power);
    int64_t it2[2] = {0, 0};
    double res2[2];
    int fl2[2] = {0, 0};
    int rc = ldiv_core(h, 2, use_precond, 0, it2, res2, fl2);
    h->tol = tol0;
    if (rc) return rc;
    int64_t tot = it2[0];
    int fl = fl2[0];
    if (fl == 0) { tot += it2[1]; fl = fl2[1]; }                                  // a failed first solve suppresses the second (:880)
    else RC(elph_launch_zero(h, h->d_x + nd, (int64_t)nd));
    if (fl == 0) tot = (tot + 1) / 2;                                             // cld(iters, 2)  (:907-909)
    *iters = tot;
    *flag = fl;
    // dSf/dx (HMC.jl:790-814)
    RC(elph_launch_force_holstein(h, h->d_tmp, h->d_x, h->d_phi, h->d_xfield, dtau));
    RC(elph_launch_s2r(h, h->d_stage_out, h->d_tmp, 1));
    std::vector<double> F(nd);
    HIPCHK(hipMemcpyAsync(F.data(), h->d_stage_out, bytes, hipMemcpyDeviceToHost, h->stream));
    if (Xp_out || Xm_out) {
        RC(elph_launch_s2r(h, h->d_stage_in, h->d_x, 2));
        if (Xp_out) HIPCHK(hipMemcpyAsync(Xp_out, h->d_stage_in, bytes, hipMemcpyDeviceToHost, h->stream));
        if (Xm_out) HIPCHK(hipMemcpyAsync(Xm_out, h->d_stage_in + nd, bytes, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < nd; ++i) dSfdx[i] += F[i];                             // "@. dSfdx += ..." accumulates (:803-811)
    return ELPH_OK;
}

// the two solves of calc_O⁻¹Λϕ! for an SSH handle (Λ = identity) + the bond brackets q[tau][n] left in h->d_p
static int ssh_force_core(elph_handle_s *h, const double *rhs_plus, const double *rhs_minus, int use_precond, double tol_power,
                          int64_t *iters, int *flag) {
    RC(ensure_capacity(h, 2));
    const size_t nd = (size_t)h->ndim, bytes = nd * sizeof(double), nq = (size_t)h->L * (size_t)h->nb;
    HIPCHK(hipMemcpyAsync(h->d_stage_in, rhs_plus, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_stage_in + nd, rhs_minus, bytes, hipMemcpyHostToDevice, h->stream));
    RC(elph_launch_r2s(h, h->d_b, h->d_stage_in, 2));
    HIPCHK(hipMemsetAsync(h->d_x, 0, 2 * bytes, h->stream));
    h->x_zero = true;
    const double tol0 = h->tol;
    h->tol = pow(tol0, tol_power);
    int64_t it2[2] = {0, 0};
    double res2[2];
    int fl2[2] = {0, 0};
    int rc = ldiv_core(h, 2, use_precond, 0, it2, res2, fl2);
    h->tol = tol0;
    if (rc) return rc;
    int64_t tot = it2[0];
    int fl = fl2[0];
    if (fl == 0) { tot += it2[1]; fl = fl2[1]; }
    else RC(elph_launch_zero(h, h->d_x + nd, (int64_t)nd));
    if (fl == 0) tot = (tot + 1) / 2;
    *iters = tot;
    *flag = fl;
    if (nq > 2 * nd) { elph_set_error("more bonds than 2*nsites: scratch too small"); return ELPH_E_UNSUPPORTED; }
    return elph_launch_force_ssh(h, h->d_p, h->d_x);     // d_p: 2*cap*ndim doubles of scratch, free once the solves are done
}

static int ssh_solutions_out(elph_handle_s *h, double *Xp_out, double *Xm_out) {
    const size_t nd = (size_t)h->ndim, bytes = nd * sizeof(double);
    if (Xp_out || Xm_out) {
        RC(elph_launch_s2r(h, h->d_stage_in, h->d_x, 2));
        if (Xp_out) HIPCHK(hipMemcpyAsync(Xp_out, h->d_stage_in, bytes, hipMemcpyDeviceToHost, h->stream));
        if (Xm_out) HIPCHK(hipMemcpyAsync(Xm_out, h->d_stage_in + nd, bytes, hipMemcpyDeviceToHost, h->stream));
    }
    return ELPH_OK;
}

extern "C" int elph_fermion_force_ssh(elph_handle h, const double *rhs_plus, const double *rhs_minus, int use_precond,
                                      double tol_power, double *q_out, double *Xp_out, double *Xm_out, int64_t *iters,
                                      int *flag) {
    CHECK_H(h);
    if (h->kind != ELPH_MODEL_SSH) { elph_set_error("not an SSH handle"); return ELPH_E_ARG; }
    RC(need_model(h));
    if (!rhs_plus || !rhs_minus || !q_out || !iters || !flag) { elph_set_error("null argument"); return ELPH_E_ARG; }
    RC(ssh_force_core(h, rhs_plus, rhs_minus, use_precond, tol_power, iters, flag));
    // q[tau][n] on the device (tau-major) -> q_out[n*L + tau] (tau fastest, like the reference's (Ltau x Nbonds) arrays)
    const size_t L = (size_t)h->L, nb = (size_t)h->nb, nq = L * nb;
    std::vector<double> qt(nq);
    HIPCHK(hipMemcpyAsync(qt.data(), h->d_p, nq * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    RC(ssh_solutions_out(h, Xp_out, Xm_out));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (size_t t = 0; t < L; ++t)
        for (size_t n = 0; n < nb; ++n) q_out[n * L + t] = qt[t * nb + n];
    return ELPH_OK;
}

// The same with the scatter onto the phonon fields done on the device (SSHModels.jl:797-823 with one field per bond):
//   dSdx[(p-1) Ltau + tau] -= sg(tau) dtau (alpha_p + 2 alpha2_p x) q[tau][bond(p)],   sg(1) = -1   (HMC.jl:803,808)
// for the fields, couplings and checkerboard positions given to the last elph_update_model_ssh_fields call.
extern "C" int elph_fermion_force_ssh_fields(elph_handle h, const double *rhs_plus, const double *rhs_minus, int use_precond,
                                             double tol_power, double *dSdx, double *Xp_out, double *Xm_out, int64_t *iters,
                                             int *flag) {
    CHECK_H(h);
    if (h->kind != ELPH_MODEL_SSH) { elph_set_error("not an SSH handle"); return ELPH_E_ARG; }
    RC(need_model(h));
    if (!rhs_plus || !rhs_minus || !dSdx || !iters || !flag) { elph_set_error("null argument"); return ELPH_E_ARG; }
    if (!h->cs_host_stale || h->ssh_nph < 0) { elph_set_error("elph_update_model_ssh_fields has not been called for the current field"); return ELPH_E_STATE; }
    RC(ssh_force_core(h, rhs_plus, rhs_minus, use_precond, tol_power, iters, flag));
    const size_t nf = (size_t)h->ssh_nph * (size_t)h->L;
    double *dF = h->d_tmp;                                   // cap*ndim doubles
    if (nf > (size_t)h->cap_rhs * (size_t)h->ndim) { elph_set_error("more phonon fields than scratch"); return ELPH_E_UNSUPPORTED; }
    RC(elph_launch_ssh_scatter(h, dF, h->d_p, h->d_ssh_x, h->d_ssh_par, h->d_ssh_cb, h->ssh_nph, h->ssh_dtau));
    std::vector<double> F(nf);
    if (nf) HIPCHK(hipMemcpyAsync(F.data(), dF, nf * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    RC(ssh_solutions_out(h, Xp_out, Xm_out));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < nf; ++i) dSdx[i] -= F[i];         // the reference accumulates into hmc.dSdx (HMC.jl:808)
    return ELPH_OK;
}

// ------------------------------------------------------------------------------------------
// muldMdx!(dMdx, u, model, v): ⟨∂M/∂x⟩ = uᵀ(∂M/∂x)v per field, for caller-given u and v — the operator calc_dSfdx! (HMC.jl:799,804)
// and LangevinDynamics.calc_dSfdx! (:378) call; HolsteinModels.jl:691-755, SSHModels.jl:707-829
// ------------------------------------------------------------------------------------------

// u, v, x, dMdx: device pointers, reference layout; lambda, lambda2: host double[nsites]
extern "C" int elph_muldMdx_holstein_dev(elph_handle h, double *dMdx_dev, const double *u_dev, const double *v_dev, const double *x_dev,
                                         const double *lambda, const double *lambda2, double dtau) {
    CHECK_H(h);
    if (h->kind != ELPH_MODEL_HOLSTEIN) { elph_set_error("not a Holstein handle"); return ELPH_E_ARG; }
    RC(need_model(h));
    if (!dMdx_dev || !u_dev || !v_dev || !x_dev || !lambda || !lambda2) { elph_set_error("null argument"); return ELPH_E_ARG; }
    if (h->nchains != 1) { elph_set_error("muldMdx! acts on one phonon configuration; the handle holds %d chains", h->nchains); return ELPH_E_STATE; }
    RC(ensure_capacity(h, 2));
    const size_t nd = (size_t)h->ndim, N = (size_t)h->N;
    HIPCHK(hipMemcpyAsync(h->d_lam, lambda, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_lam + N, lambda2, N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    RC(elph_launch_r2s(h, h->d_xfield, x_dev, 1));
    RC(elph_launch_r2s(h, h->d_b, u_dev, 1));
    RC(elph_launch_r2s(h, h->d_b + nd, v_dev, 1));
    RC(elph_launch_dmdx_holstein(h, h->d_tmp, h->d_b, h->d_b + nd, h->d_xfield, dtau, 1.0));
    RC(elph_launch_s2r(h, dMdx_dev, h->d_tmp, 1));
    HIPCHK(hipStreamSynchronize(h->stream));           // lambda / lambda2 are the caller's host arrays
    return ELPH_OK;
}

extern "C" int elph_muldMdx_holstein(elph_handle h, double *dMdx, const double *u, const double *v, const double *x,
                                     const double *lambda, const double *lambda2, double dtau) {
    CHECK_H(h);
    if (!dMdx || !u || !v || !x) { elph_set_error("null argument"); return ELPH_E_ARG; }
    RC(ensure_capacity(h, 3));                         // stage_in: u, v, x
    const size_t nd = (size_t)h->ndim, bytes = nd * sizeof(double);
    HIPCHK(hipMemcpyAsync(h->d_stage_in, u, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_stage_in + nd, v, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_stage_in + 2 * nd, x, bytes, hipMemcpyHostToDevice, h->stream));
    RC(elph_muldMdx_holstein_dev(h, h->d_stage_out, h->d_stage_in, h->d_stage_in + nd, h->d_stage_in + 2 * nd, lambda, lambda2, dtau));
    HIPCHK(hipMemcpyAsync(dMdx, h->d_stage_out, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}

// the bond brackets q[tau][n] of muldMdx!(·, u, ssh, v) left in h->d_p (u, v: device, reference layout)
static int ssh_dmdx_core(elph_handle_s *h, const double *u_dev, const double *v_dev) {
    if (h->kind != ELPH_MODEL_SSH) { elph_set_error("not an SSH handle"); return ELPH_E_ARG; }
    RC(need_model(h));
    if (!u_dev || !v_dev) { elph_set_error("null argument"); return ELPH_E_ARG; }
    if (h->nchains != 1) { elph_set_error("muldMdx! acts on one phonon configuration; the handle holds %d chains", h->nchains); return ELPH_E_STATE; }
    RC(ensure_capacity(h, 2));
    const size_t nd = (size_t)h->ndim, nq = (size_t)h->L * (size_t)h->nb;
    if (nq > 2 * nd) { elph_set_error("more bonds than 2*nsites: scratch too small"); return ELPH_E_UNSUPPORTED; }
    RC(elph_launch_r2s(h, h->d_b, u_dev, 1));
    RC(elph_launch_r2s(h, h->d_b + nd, v_dev, 1));
    return elph_launch_force_ssh(h, h->d_p, h->d_b + nd, h->d_b, 1);      // b0 = e^{Δτμ} v(τ−1), c0 = CBᵀ u
}

static int ssh_dmdx_stage(elph_handle_s *h, const double *u, const double *v) {
    if (!u || !v) { elph_set_error("null argument"); return ELPH_E_ARG; }
    RC(ensure_capacity(h, 2));
    const size_t nd = (size_t)h->ndim, bytes = nd * sizeof(double);
    HIPCHK(hipMemcpyAsync(h->d_stage_in, u, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_stage_in + nd, v, bytes, hipMemcpyHostToDevice, h->stream));
    return ssh_dmdx_core(h, h->d_stage_in, h->d_stage_in + nd);
}

extern "C" int elph_muldMdx_ssh(elph_handle h, double *q_out, const double *u, const double *v) {
    CHECK_H(h);
    if (!q_out) { elph_set_error("null argument"); return ELPH_E_ARG; }
    RC(ssh_dmdx_stage(h, u, v));
    const size_t L = (size_t)h->L, nb = (size_t)h->nb, nq = L * nb;
    std::vector<double> qt(nq);
    if (nq) HIPCHK(hipMemcpyAsync(qt.data(), h->d_p, nq * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (size_t t = 0; t < L; ++t)
        for (size_t n = 0; n < nb; ++n) q_out[n * L + t] = qt[t * nb + n];
    return ELPH_OK;
}

static int ssh_dmdx_fields(elph_handle_s *h, double **dF, size_t *nf) {
    if (!h->cs_host_stale || h->ssh_nph < 0) { elph_set_error("elph_update_model_ssh_fields has not been called for the current field"); return ELPH_E_STATE; }
    *nf = (size_t)h->ssh_nph * (size_t)h->L;
    *dF = h->d_tmp;
    if (*nf > (size_t)h->cap_rhs * (size_t)h->ndim) { elph_set_error("more phonon fields than scratch"); return ELPH_E_UNSUPPORTED; }
    return elph_launch_ssh_scatter(h, *dF, h->d_p, h->d_ssh_x, h->d_ssh_par, h->d_ssh_cb, h->ssh_nph, h->ssh_dtau);
}

extern "C" int elph_muldMdx_ssh_fields(elph_handle h, double *dMdx, const double *u, const double *v) {
    CHECK_H(h);
    if (!dMdx) { elph_set_error("null argument"); return ELPH_E_ARG; }
    RC(ssh_dmdx_stage(h, u, v));
    double *dF = nullptr; size_t nf = 0;
    RC(ssh_dmdx_fields(h, &dF, &nf));
    if (nf) HIPCHK(hipMemcpyAsync(dMdx, dF, nf * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}

extern "C" int elph_muldMdx_ssh_fields_dev(elph_handle h, double *dMdx_dev, const double *u_dev, const double *v_dev) {
    CHECK_H(h);
    if (!dMdx_dev) { elph_set_error("null argument"); return ELPH_E_ARG; }
    RC(ssh_dmdx_core(h, u_dev, v_dev));
    double *dF = nullptr; size_t nf = 0;
    RC(ssh_dmdx_fields(h, &dF, &nf));
    if (nf) HIPCHK(hipMemcpyAsync(dMdx_dev, dF, nf * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    return ELPH_OK;
}

// ------------------------------------------------------------------------------------------
// KPM preconditioner
// ------------------------------------------------------------------------------------------

extern "C" int elph_kpm_create(elph_handle h, int n, double buf, double c1, double c2) {
    CHECK_H(h);
    if (n < 1 || !(buf >= 0.0)) { elph_set_error("bad KPM parameters"); return ELPH_E_ARG; }
    h->kpm_n = n; h->kpm_buf = buf; h->kpm_c1 = c1; h->kpm_c2 = c2;
    // KPMExpansion ctor, KPMPreconditioners.jl:101-146: λ_lo = 0, λ_hi = 2, order 1 everywhere (one per chain, made on demand)
    h->lam_lo = 0.0; h->lam_hi = 2.0; h->lam_avg = 1.0; h->lam_mag = 1.0;
    h->kpm_chain.clear();
    h->kpm_nch = 1;
    h->kpm_hop_uploaded = false;
    h->h_cbar.assign((size_t)h->nb, 0.0);
    h->h_sbar.assign((size_t)h->nb, 0.0);
    RC(dev_alloc(&h->d_cbar, (size_t)h->nb));
    RC(dev_alloc(&h->d_sbar, (size_t)h->nb));
    h->kpm_tab_cap = 0;
    h->kpm_created = true;
    h->kpm_ready = false;
    h->kpm_active = 1;
    return ELPH_OK;
}

// (re)size the per-chain host state and the device tables for nch chains
static int kpm_reserve(elph_handle_s *h, int nch) {
    const int Lo2 = (int)((h->L + 1) / 2);
    if ((int)h->kpm_chain.size() != nch) {
        // a different number of configurations: every expansion starts from the constructor state again
        h->kpm_chain.assign((size_t)nch, elph_handle_s::KpmChainHost());
        for (auto &c : h->kpm_chain) { c.order.assign(Lo2, 1); c.coeff.assign(2 * (size_t)Lo2, 0.0); for (int w = 0; w < Lo2; ++w) c.coeff[2 * w] = 1.0; }
    }
    h->kpm_nch = nch;
    h->h_Ebar.resize((size_t)nch * h->N);
    if (nch > h->kpm_tab_cap) {
        RC(dev_alloc(&h->d_Ebar, (size_t)nch * h->N));
        RC(dev_alloc(&h->d_order, (size_t)nch * Lo2));
        RC(dev_alloc(&h->d_coff, (size_t)nch * (Lo2 + 1)));
        RC(dev_alloc(&h->d_wsched, (size_t)nch * Lo2));
        RC(dev_alloc(&h->d_kdesc, (size_t)nch * Lo2));
        RC(dev_alloc(&h->d_kfold, (size_t)nch * Lo2 * 2));
        RC(dev_alloc(&h->d_klam, (size_t)nch * 2));
        h->kpm_tab_cap = nch;
    }
    return ELPH_OK;
}

// flatten the per-chain expansions into the device tables.  A chain whose expansion is inactive (while others are
// active) gets the identity expansion: order 1, c₀ = 1 — ldiv! copies for it (KPMPreconditioners.jl:475-478).
static int kpm_upload(elph_handle_s *h) {
    const int Lo2 = (int)((h->L + 1) / 2), nch = h->kpm_nch;
    h->h_order.assign((size_t)nch * Lo2, 1);
    h->h_coff.assign((size_t)nch * (Lo2 + 1), 0);
    h->h_wsched.assign((size_t)nch * Lo2, 0);
    h->h_lam.assign((size_t)nch * 2, 1.0);
    h->h_coeff.clear();
    h->h_kdesc.assign((size_t)nch * Lo2, KpmDesc());
    h->h_kfold.assign((size_t)nch * Lo2 * 2, 0.0);
    int off = 0;
    for (int c = 0; c < nch; ++c) {
        const auto &C = h->kpm_chain[(size_t)c];
        int *ord = h->h_order.data() + (size_t)c * Lo2, *cof = h->h_coff.data() + (size_t)c * (Lo2 + 1);
        int *ws = h->h_wsched.data() + (size_t)c * Lo2;
        int loc = 0;
        for (int w = 0; w < Lo2; ++w) {
            const int o = C.active ? C.order[w] : 1;
            ord[w] = o;
            cof[w] = off;
            if (C.active) {
                h->h_coeff.insert(h->h_coeff.end(), C.coeff.begin() + 2 * (size_t)loc, C.coeff.begin() + 2 * (size_t)(loc + o));
                loc += o;
            } else {
                h->h_coeff.push_back(1.0); h->h_coeff.push_back(0.0);
                loc += C.order[w];
            }
            off += o;
        }
        cof[Lo2] = off;
        // schedule: frequency blocks by decreasing order (the low frequencies carry the long recursions)
        std::iota(ws, ws + Lo2, 0);
        std::stable_sort(ws, ws + Lo2, [&](int a, int b) { return ord[a] > ord[b]; });
        for (int y = 0; y < Lo2; ++y) {
            const int w = ws[y];
            KpmDesc d;
            d.w = w; d.order = ord[w]; d.coff = cof[w]; d.pad = 0;
            d.c0x = h->h_coeff[2 * (size_t)cof[w]]; d.c0y = h->h_coeff[2 * (size_t)cof[w] + 1];
            h->h_kdesc[(size_t)c * Lo2 + y] = d;
            // order-1 fold (active chains only: an identity expansion hands over the r.r partial sums instead, bit for bit)
            const bool fold = C.active && d.order == 1;
            const double s1 = fold ? d.c0x * d.c0x + d.c0y * d.c0y : 1.0;
            const double wgt = ((h->L & 1) && w == Lo2 - 1) ? 1.0 : 2.0;
            h->h_kfold[2 * ((size_t)c * Lo2 + w)] = s1;
            h->h_kfold[2 * ((size_t)c * Lo2 + w) + 1] = fold ? wgt * s1 / (double)h->L : 0.0;
        }
        h->h_lam[2 * c] = (C.lam_hi + C.lam_lo) / 2;
        h->h_lam[2 * c + 1] = C.active ? (C.lam_hi - C.lam_lo) / 2 : -1.0;      // < 0 marks the identity (KpmChainView::active)
    }
    const size_t ntot = (size_t)off;
    if ((int64_t)ntot > h->coeff_cap) {
        RC(dev_alloc(&h->d_coeff, ntot));
        h->coeff_cap = (int64_t)ntot;
    }
    HIPCHK(hipMemcpy(h->d_coeff, h->h_coeff.data(), ntot * sizeof(double2), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_order, h->h_order.data(), sizeof(int) * h->h_order.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_coff, h->h_coff.data(), sizeof(int) * h->h_coff.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_wsched, h->h_wsched.data(), sizeof(int) * h->h_wsched.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_kdesc, h->h_kdesc.data(), sizeof(KpmDesc) * h->h_kdesc.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_kfold, h->h_kfold.data(), sizeof(double) * h->h_kfold.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_klam, h->h_lam.data(), sizeof(double) * h->h_lam.size(), hipMemcpyHostToDevice));
    h->lam_lo = h->kpm_chain[0].lam_lo; h->lam_hi = h->kpm_chain[0].lam_hi;
    h->lam_avg = h->h_lam[0]; h->lam_mag = h->h_lam[1];
    return ELPH_OK;
}

static bool jl_isapprox(double x, double y, double rtol) {
    return x == y || (std::isfinite(x) && std::isfinite(y) && fabs(x - y) <= rtol * std::max(fabs(x), fabs(y)));
}

// setup!(P) for every chain resident in the handle (h->nchains of them).  All arrays are per chain.
static int kpm_setup_core(elph_handle_s *h, const double *b_max, const double *b_min, const double *e_min_in,
                          const double *e_max_in, int *active, double *lam_lo, double *lam_hi) {
    if (!h->kpm_created) { elph_set_error("elph_kpm_create has not been called"); return ELPH_E_STATE; }
    RC(need_model(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int N = (int)h->N, L = (int)h->L, Lo2 = (L + 1) / 2, nch = h->nchains;
    const bool resized = ((int)h->kpm_chain.size() != nch);
    RC(kpm_reserve(h, nch));
    // update_A!  (KPMPreconditioners.jl:332-349 Holstein; :355-381 SSH)
    if (h->kind == ELPH_MODEL_HOLSTEIN) {
        if (!h->ebar_external) RC(elph_launch_ebar(h, nch));      // (Ē stays on the device: the Arnoldi kernel and the apply read it there;
                                                                  //  ebar_external: d_Ebar was filled by the caller — elph_i_kpm_setup_ebar)
        h->h_cbar = h->h_c;
        h->h_sbar = h->h_s;
        h->kpm_hop_per_chain = false;
    } else {
        // Ebar = exp(dtau mu), held per chain (equal unless the chemical potential is tuned per chain)
        HIPCHK(hipMemcpyAsync(h->d_Ebar, h->d_E, sizeof(double) * (size_t)nch * N, hipMemcpyDeviceToDevice, h->stream));
        h->kpm_hop_per_chain = nch > 1;
        if (nch > h->kpm_hop_cap) {          // averaged hopping tables per chain (and their lane-program / register-exchange images)
            RC(dev_alloc(&h->d_cbar, (size_t)nch * h->nb));
            RC(dev_alloc(&h->d_sbar, (size_t)nch * h->nb));
            if (h->fast_capable) {
                RC(dev_alloc(&h->d_lp_cbar, (size_t)nch * h->lp_ne * ELPH_WAVE));
                RC(dev_alloc(&h->d_lp_sbar, (size_t)nch * h->lp_ne * ELPH_WAVE));
            }
            if (h->sq_L > 0) {
                RC(dev_alloc(&h->d_sq_cbar, (size_t)nch * 4 * h->N));
                RC(dev_alloc(&h->d_sq_sbar, (size_t)nch * 4 * h->N));
            }
            h->kpm_hop_cap = nch;
        }
        h->h_cbar.resize((size_t)nch * h->nb);
        h->h_sbar.resize((size_t)nch * h->nb);
        if (h->nb > 0 && h->csbar_external) {      // (the caller's: the full-lattice handle of a sharded update — elph_i_kpm_setup_csbar filled h_cbar / h_sbar)
            HIPCHK(hipMemcpyAsync(h->d_cbar, h->h_cbar.data(), sizeof(double) * (size_t)nch * h->nb, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->d_sbar, h->h_sbar.data(), sizeof(double) * (size_t)nch * h->nb, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
        } else if (h->nb > 0) {   // tau-means of cosht, sinht from the device tables (they may have been produced there)
            RC(elph_launch_cs_bar(h, h->d_cbar, h->d_sbar, nch));
            HIPCHK(hipMemcpyAsync(h->h_cbar.data(), h->d_cbar, sizeof(double) * (size_t)nch * h->nb, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipMemcpyAsync(h->h_sbar.data(), h->d_sbar, sizeof(double) * (size_t)nch * h->nb, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
        }
    }
    // Holstein: c̄ = cosh(Δτ t), s̄ = sinh(Δτ t) never change after elph_create — upload their three device images once
    const bool hop_fresh = !(h->kind == ELPH_MODEL_HOLSTEIN && h->kpm_hop_uploaded);
    const int hch = h->kpm_hop_per_chain ? nch : 1;                       // hopping tables: one per chain (SSH chains) or shared
    if (hop_fresh && h->nb > 0 && h->kind == ELPH_MODEL_HOLSTEIN) {
        HIPCHK(hipMemcpy(h->d_cbar, h->h_cbar.data(), sizeof(double) * h->nb, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_sbar, h->h_sbar.data(), sizeof(double) * h->nb, hipMemcpyHostToDevice));
    }
    if (hop_fresh && h->fast_capable) {
        const size_t per = (size_t)h->lp_ne * ELPH_WAVE;
        std::vector<double> lc((size_t)hch * per), ls((size_t)hch * per);
        for (int c = 0; c < hch; ++c) {
            elph_lp_pack(h, h->h_cbar.data() + (size_t)c * h->nb, lc.data() + (size_t)c * per, 1.0);
            elph_lp_pack(h, h->h_sbar.data() + (size_t)c * h->nb, ls.data() + (size_t)c * per, 0.0);
        }
        HIPCHK(hipMemcpy(h->d_lp_cbar, lc.data(), sizeof(double) * lc.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_lp_sbar, ls.data(), sizeof(double) * ls.size(), hipMemcpyHostToDevice));
    }
    if (hop_fresh && h->sq_L > 0) {
        const size_t per = (size_t)4 * h->N;
        std::vector<double> qc((size_t)hch * per), qs((size_t)hch * per);
        bool uni = true;
        for (int c = 0; c < hch; ++c) {
            double *q0 = qc.data() + (size_t)c * per, *q1 = qs.data() + (size_t)c * per;
            for (size_t k = 0; k < per; ++k) { q0[k] = h->h_cbar[(size_t)c * h->nb + h->sq_bond[k]]; q1[k] = h->h_sbar[(size_t)c * h->nb + h->sq_bond[k]]; }
            for (size_t k = 1; k < per; ++k) uni = uni && q0[k] == q0[0] && q1[k] == q1[0];       // uniform within the chain
        }
        if (uni != h->sq_uniform) drop_graphs(h);
        h->sq_uniform = uni;
        HIPCHK(hipMemcpy(h->d_sq_cbar, qc.data(), sizeof(double) * qc.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_sq_sbar, qs.data(), sizeof(double) * qs.size(), hipMemcpyHostToDevice));
    }
    if (hop_fresh && h->hc_LX > 0) {
        bool uni = h->nb > 0 && !h->kpm_hop_per_chain;
        for (size_t k = 1; k < (size_t)h->nb && uni; ++k) uni = h->h_cbar[k] == h->h_cbar[0] && h->h_sbar[k] == h->h_sbar[0];
        if (uni != h->hc_uniform) drop_graphs(h);
        h->hc_uniform = uni;
    }
    if (hop_fresh && h->pg_L > 0) {
        bool uni = h->nb > 0 && !h->kpm_hop_per_chain;
        for (size_t k = 1; k < (size_t)h->nb && uni; ++k) uni = h->h_cbar[k] == h->h_cbar[0] && h->h_sbar[k] == h->h_sbar[0];
        if (uni != h->pg_uniform) drop_graphs(h);       // (a captured chunk holds the Chebyshev kernel chosen under the old flag)
        h->pg_uniform = uni;
    }
    h->kpm_hop_uploaded = true;
    const int was_active = h->kpm_active;
    bool changed = !h->kpm_ready || resized;
    if (!(b_max && b_min))
        for (int c = 0; c < nch; ++c)
            if (!(e_min_in && e_max_in && std::isfinite(e_min_in[c]) && std::isfinite(e_max_in[c]))) {
                elph_set_error("Arnoldi start vectors required when bounds are not injected");
                return ELPH_E_ARG;
            }
    // eigenvalue bounds (:272-273): injected, or the Arnoldi process with the caller's start vectors — on the device for all chains at
    // once (kpm_dev.hip: one wavefront per chain and per operator, Ritz values by a wave-parallel Hessenberg QR), on the host for
    // lattices beyond one wave
    std::vector<double> eb((size_t)2 * nch, NAN);
    bool need_arnoldi = false;
    for (int c = 0; c < nch; ++c)
        if (!(e_min_in && e_max_in && std::isfinite(e_min_in[c]) && std::isfinite(e_max_in[c]))) need_arnoldi = true;
    bool on_device = false;
    if (need_arnoldi) {
        const char *eh = getenv("ELPH_KPM_HOST"), *ed = getenv("ELPH_KPM_DEVICE");     // read per call: tests pin one path
        const bool host_only = eh && eh[0] == '1', dev_always = ed && ed[0] == '1';
        const size_t nst = (size_t)2 * nch * N;
        // (one or two chains: the host's scalar Arnoldi + LAPACK-style QR, 0.1 ms per chain, beats a kernel whose one wave spends
        //  ~0.25 ms on the same sequential work; from three chains on all of them run side by side on the device)
        if (!host_only && N <= 512 && (nch >= 3 || dev_always)) {
            if ((int64_t)(nst + 2 * nch) > h->kpm_start_cap) {
                RC(dev_alloc(&h->d_kpm_start, nst + 2 * (size_t)nch));
                h->kpm_start_cap = (int64_t)(nst + 2 * nch);
            }
            HIPCHK(hipMemcpyAsync(h->d_kpm_start, b_max, sizeof(double) * (size_t)nch * N, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->d_kpm_start + (size_t)nch * N, b_min, sizeof(double) * (size_t)nch * N, hipMemcpyHostToDevice, h->stream));
            const int rcd = elph_kpm_bounds_dev(h, nch, h->d_kpm_start, h->d_kpm_start + nst);
            if (rcd == ELPH_OK) {
                HIPCHK(hipMemcpyAsync(eb.data(), h->d_kpm_start + nst, sizeof(double) * 2 * (size_t)nch, hipMemcpyDeviceToHost, h->stream));
                HIPCHK(hipStreamSynchronize(h->stream));
                on_device = true;
            } else if (rcd != ELPH_E_UNSUPPORTED) {
                return rcd;
            }
        }
        if (!on_device) {
            // host path needs Ē (and the averaged hoppings) on the host
            HIPCHK(hipMemcpy(h->h_Ebar.data(), h->d_Ebar, sizeof(double) * (size_t)nch * N, hipMemcpyDeviceToHost));
        }
    }
    // per chain: acceptance window, and new orders + coefficients when the bounds moved by more than buf (host, ~0.1 ms per chain,
    // only when they moved)
    std::vector<char> moved((size_t)nch, 0);
    std::vector<int> was((size_t)nch, 0);
    auto one_chain = [&](int c) {
        auto &C = h->kpm_chain[(size_t)c];
        was[(size_t)c] = C.active;
        double e_min = e_min_in ? e_min_in[c] : NAN, e_max = e_max_in ? e_max_in[c] : NAN;
        if (!(std::isfinite(e_min) && std::isfinite(e_max))) {
            if (on_device) { e_min = eb[2 * (size_t)c]; e_max = eb[2 * (size_t)c + 1]; }
            else (void)elph_kpm_arnoldi(h, c, b_max + (size_t)c * N, b_min + (size_t)c * N, &e_min, &e_max);
        }
        if ((0.0 < e_min && e_min < 1.0) && (1.0 < e_max) && (e_max - e_min) < 2.0) {       // :280
            const double lo = std::max(0.0, (1 - 2 * h->kpm_buf) * e_min), hi = (1 + 2 * h->kpm_buf) * e_max;
            if (!jl_isapprox(lo, C.lam_lo, h->kpm_buf) || !jl_isapprox(hi, C.lam_hi, h->kpm_buf)) {   // :288
                C.lam_lo = lo; C.lam_hi = hi;
                int off = 0;
                std::vector<double> coeff;
                for (int w = 0; w < Lo2; ++w) {
                    const double phi = 2.0 * M_PI / (double)L * (w + 0.5);                   // ctor :117
                    int order = (int)floor((hi - lo) * (h->kpm_c1 / phi + h->kpm_c2));       // :300
                    order = std::max(1, order);
                    C.order[w] = order;
                    coeff.resize(2 * (size_t)(off + order));
                    elph_kpm_coefficients(coeff.data() + 2 * (size_t)off, order, lo, hi, phi);
                    off += order;
                }
                C.coeff.swap(coeff);
                moved[(size_t)c] = 1;
            }
            C.active = 1;
        } else {
            C.active = 0;                                                                    // :312-318
        }
    };
    for (int c = 0; c < nch; ++c) one_chain(c);
    int any_active = 0;
    for (int c = 0; c < nch; ++c) {
        auto &C = h->kpm_chain[(size_t)c];
        if (moved[(size_t)c] || was[(size_t)c] != C.active || C.fresh) changed = true;
        C.fresh = false;
        any_active |= C.active;
        if (active) active[c] = C.active;
        if (lam_lo) lam_lo[c] = C.lam_lo;
        if (lam_hi) lam_hi[c] = C.lam_hi;
    }
    h->kpm_active = any_active;
    if (changed || was_active != h->kpm_active) {
        RC(kpm_upload(h));
        drop_graphs(h);   // KpmDev (lam_avg, lam_mag, active) is baked into captured kernel arguments
    }
    h->kpm_ready = true;
    return ELPH_OK;
}

// setup!(P) of a Holstein handle whose τ-averaged exp(−ΔτV) comes from OUTSIDE: the full-lattice handle of a sharded HMC update holds no
// field of its own — every rank contributes the Ē of its own rows and the sum is injected here (hmc.hip).  One chain.
int elph_i_kpm_setup_ebar(elph_handle_s *h, const double *Ebar_host, const double *b_max, const double *b_min) {
    if (h->kind != ELPH_MODEL_HOLSTEIN || !h->kpm_created) { elph_set_error("Ē injection: a Holstein handle with elph_kpm_create done"); return ELPH_E_STATE; }
    if (h->nchains != 1) { h->nchains = 1; drop_graphs(h); h->kpm_ready = false; }
    RC(kpm_reserve(h, 1));
    HIPCHK(hipMemcpy(h->d_Ebar, Ebar_host, sizeof(double) * (size_t)h->N, hipMemcpyHostToDevice));
    const bool had_E = h->have_E;
    h->have_E = true;                  // (the expansion needs Ē and the hopping only; this handle never multiplies by M)
    h->ebar_external = true;
    const int rc = kpm_setup_core(h, b_max, b_min, nullptr, nullptr, nullptr, nullptr, nullptr);
    h->ebar_external = false;
    h->have_E = had_E;
    return rc;
}

// setup!(P) of a bond-phonon handle whose τ-averaged hopping tables come from OUTSIDE (update_A!, KPMPreconditioners.jl:355-381): the
// full-lattice handle of a sharded HMC update — the hoppings move on the ranks' slabs, every rank contributes the τ-means of the bonds it
// owns and the sum is injected here (hmc.hip).  exp(Δτμ) is the handle's own (elph_update_model_ssh once; μ does not move).  One chain.
int elph_i_kpm_setup_csbar(elph_handle_s *h, const double *cbar_host, const double *sbar_host, const double *b_max, const double *b_min) {
    if (h->kind != ELPH_MODEL_SSH || !h->kpm_created || !h->have_E) {
        elph_set_error("c̄ / s̄ injection: a bond-phonon handle with elph_kpm_create and one elph_update_model_ssh done");
        return ELPH_E_STATE;
    }
    if (h->nchains != 1) { h->nchains = 1; drop_graphs(h); h->kpm_ready = false; }
    h->h_cbar.assign(cbar_host, cbar_host + h->nb);
    h->h_sbar.assign(sbar_host, sbar_host + h->nb);
    h->csbar_external = true;
    const int rc = kpm_setup_core(h, b_max, b_min, nullptr, nullptr, nullptr, nullptr, nullptr);
    h->csbar_external = false;
    return rc;
}

extern "C" int elph_kpm_setup(elph_handle h, const double *b_max, const double *b_min, double e_min, double e_max,
                              int *active, double *lam_lo, double *lam_hi) {
    CHECK_H(h);
    if (h->nchains != 1) { elph_set_error("%d phonon configurations are resident: use elph_kpm_setup_chains", h->nchains); return ELPH_E_STATE; }
    return kpm_setup_core(h, b_max, b_min, &e_min, &e_max, active, lam_lo, lam_hi);
}

extern "C" int elph_kpm_setup_chains(elph_handle h, const double *b_max, const double *b_min, const double *e_min,
                                     const double *e_max, int *active, double *lam_lo, double *lam_hi) {
    CHECK_H(h);
    return kpm_setup_core(h, b_max, b_min, e_min, e_max, active, lam_lo, lam_hi);
}

extern "C" int elph_kpm_orders(elph_handle h, int64_t *orders, int64_t *total) {
    CHECK_H(h);
    if (!h->kpm_created) { elph_set_error("elph_kpm_create has not been called"); return ELPH_E_STATE; }
    const int Lo2 = (int)((h->L + 1) / 2);
    int64_t tot = 0;
    for (int w = 0; w < Lo2; ++w) {
        const int o = h->kpm_chain.empty() ? 1 : h->kpm_chain[0].order[w];
        if (orders) orders[w] = o;
        tot += o;
    }
    if (total) *total = tot;
    return ELPH_OK;
}

extern "C" int elph_kpm_apply_dev(elph_handle h, double *z_dev, const double *r_dev) {
    CHECK_H(h);
    if (!h->kpm_ready) { elph_set_error("elph_kpm_setup has not been called"); return ELPH_E_STATE; }
    if (!z_dev || !r_dev) { elph_set_error("null argument"); return ELPH_E_ARG; }
    RC(elph_launch_r2s(h, h->d_r, r_dev, 1));
    RC(elph_launch_kpm_apply(h, h->d_zp, h->d_r, 1, 0));
    RC(elph_launch_s2r(h, z_dev, h->d_zp, 1));
    return ELPH_OK;
}

extern "C" int elph_kpm_apply(elph_handle h, double *z, const double *r) {
    CHECK_H(h);
    if (!z || !r) { elph_set_error("null argument"); return ELPH_E_ARG; }
    const size_t bytes = (size_t)h->ndim * sizeof(double);
    HIPCHK(hipMemcpyAsync(h->d_stage_in, r, bytes, hipMemcpyHostToDevice, h->stream));
    RC(elph_kpm_apply_dev(h, h->d_stage_out, h->d_stage_in));
    HIPCHK(hipMemcpyAsync(z, h->d_stage_out, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}

// ------------------------------------------------------------------------------------------
// Fourier acceleration / twisted FFT
// ------------------------------------------------------------------------------------------

extern "C" int elph_fourier_accelerate(elph_handle h, double *vout, const double *vin, const double *diag, double power,
                                       int64_t nph) {
    CHECK_H(h);
    if (!vout || !vin || !diag || nph < 1) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    const int64_t L = h->L, n = nph * L;
    // scratch: 3 real vectors of n + complex half spectrum; grown on demand (nph may exceed nsites for SSH)
    const int64_t need = 3 * n + 2 * (L / 2 + 1) * nph;
    if (need > h->diag_cap) {
        HIPCHK(hipStreamSynchronize(h->stream));
        RC(dev_alloc(&h->d_diag, (size_t)need));
        h->diag_cap = need;
    }
    double *dR = h->d_diag, *aS = h->d_diag + n, *bS = h->d_diag + 2 * n;
    double2 *u = reinterpret_cast<double2 *>(h->d_diag + 3 * n);
    // stage (layout R -> S for "nph" columns), transform with the handle's tables, stage back
    HIPCHK(hipMemcpyAsync(dR, diag, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    RC(elph_launch_r2s(h, bS, dR, 1, (int)nph));          // bS = diag in layout S  (diag[k][s])
    HIPCHK(hipMemcpyAsync(dR, vin, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    RC(elph_launch_r2s(h, aS, dR, 1, (int)nph));          // aS = v in layout S
    RC(elph_dft_accel(h, dR, aS, bS, power, (int)nph, u, 1));   // dR = result in layout S
    RC(elph_launch_s2r(h, aS, dR, 1, (int)nph));          // aS = result in layout R
    HIPCHK(hipMemcpyAsync(vout, aS, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}

extern "C" int elph_tau_to_omega(elph_handle h, double *nu_complex, const double *v) {
    CHECK_H(h);
    if (!nu_complex || !v) { elph_set_error("null argument"); return ELPH_E_ARG; }
    const size_t nd = (size_t)h->ndim;
    RC(ensure_capacity(h, 2));   // complex scratch = 2 real vectors
    HIPCHK(hipMemcpyAsync(h->d_stage_in, v, nd * sizeof(double), hipMemcpyHostToDevice, h->stream));
    RC(elph_launch_r2s(h, h->d_b, h->d_stage_in, 1));
    double2 *fullS = reinterpret_cast<double2 *>(h->d_p);           // [L][N] complex
    RC(elph_launch_tau_to_omega(h, fullS, h->d_b));
    // complex layout S -> complex layout R: transpose re and im planes separately via a strided copy
    // (API convenience path; the solver itself never leaves layout S)
    std::vector<double> tmp(2 * nd);
    HIPCHK(hipMemcpyAsync(tmp.data(), fullS, 2 * nd * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t N = (size_t)h->N, L = (size_t)h->L;
    for (size_t k = 0; k < L; ++k)
        for (size_t s = 0; s < N; ++s) {
            nu_complex[2 * (s * L + k)] = tmp[2 * (k * N + s)];
            nu_complex[2 * (s * L + k) + 1] = tmp[2 * (k * N + s) + 1];
        }
    return ELPH_OK;
}

extern "C" int elph_omega_to_tau(elph_handle h, double *v, const double *nu_complex) {
    CHECK_H(h);
    if (!nu_complex || !v) { elph_set_error("null argument"); return ELPH_E_ARG; }
    const size_t nd = (size_t)h->ndim, N = (size_t)h->N, L = (size_t)h->L;
    RC(ensure_capacity(h, 2));
    std::vector<double> tmp(2 * nd);
    for (size_t k = 0; k < L; ++k)
        for (size_t s = 0; s < N; ++s) {
            tmp[2 * (k * N + s)] = nu_complex[2 * (s * L + k)];
            tmp[2 * (k * N + s) + 1] = nu_complex[2 * (s * L + k) + 1];
        }
    double2 *fullS = reinterpret_cast<double2 *>(h->d_p);
    HIPCHK(hipMemcpy(fullS, tmp.data(), 2 * nd * sizeof(double), hipMemcpyHostToDevice));
    RC(elph_launch_omega_to_tau(h, h->d_x, fullS));
    RC(elph_launch_s2r(h, h->d_stage_out, h->d_x, 1));
    HIPCHK(hipMemcpyAsync(v, h->d_stage_out, nd * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ELPH_OK;
}

// ------------------------------------------------------------------------------------------
// measurement hooks (bench.py)
// ------------------------------------------------------------------------------------------

static int bench_launch_unit(elph_handle_s *h, int what, int nrhs) {
    switch (what) {
        case 0: return elph_launch_mul(h, 2, h->d_z, h->d_b, nrhs);
        case 1: return elph_launch_cg_iteration(h, nrhs, 0);
        case 2: return elph_launch_kpm_apply(h, h->d_zp, h->d_b, nrhs, 0);
        case 3: return elph_launch_cg_iteration(h, nrhs, 1);
        case 11: return ELPH_E_ARG;      // (two half-batches on two streams: elph_bench_run drives both handles itself)
        case 4: return elph_launch_cg_kernel(h, nrhs, 0);
        case 5: return elph_launch_cg_kernel(h, nrhs, 1);
        default: {
            // 6, 7, 8: the three kernels of the KPM apply as the preconditioned iteration (case 3) launches them, one at a time
            const bool xr_fused = h->fast && h->kpm_active && h->dot_hi == 0 && elph_dft_mfma_xr_usable(h, (int)h->N, nrhs);
            return elph_launch_kpm_apply(h, h->d_zp, h->d_r, nrhs, xr_fused ? 2 : 1, 1 << (what - 6));
        }
    }
}

extern "C" int elph_bench_prepare(elph_handle h, int what, int nrhs, const double *B) {
    CHECK_H(h);
    RC(need_model(h));
    if (nrhs < 1 || what < 0 || what > 12) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    if ((what == 2 || what == 3 || what == 10 || what == 11 || (what >= 6 && what <= 8)) && !h->kpm_ready) { elph_set_error("KPM not set up"); return ELPH_E_STATE; }
    RC(ensure_capacity(h, nrhs));
    if (B) {
        const size_t bytes = (size_t)nrhs * (size_t)h->ndim * sizeof(double);
        HIPCHK(hipMemcpyAsync(h->d_stage_in, B, bytes, hipMemcpyHostToDevice, h->stream));
        RC(elph_launch_r2s(h, h->d_b, h->d_stage_in, nrhs));
    }
    // fixed-count CG: tol = 0 never converges, kmax = inf, x0 = 0
    CgParams P;
    P.tol = 0.0; P.kmax = INFINITY; P.maxiter = (long long)1 << 40; P.use_prec = (what == 3 || what == 10 || what == 11 || (what >= 6 && what <= 8)); P.record_hist = 0; P.hist_stride = 0;
    h->cur_params = P;
    HIPCHK(hipMemsetAsync(h->d_x, 0, (size_t)nrhs * (size_t)h->ndim * sizeof(double), h->stream));
    h->x_zero = false;
    RC(elph_launch_cg_init(h, nrhs, P.use_prec, true));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->bench_fresh = true;
    return ELPH_OK;
}

extern "C" int elph_bench_info(elph_handle h, int nrhs, int *slices_per_wave) {
    CHECK_H(h);
    if (nrhs < 1 || !slices_per_wave) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    *slices_per_wave = (h->px_solve && h->cur_params.use_prec) ? elph_choose_T_px(h, nrhs) : elph_choose_T(h, nrhs);
    return ELPH_OK;
}

// Health of the workgroup-resident kernels on this handle: *cooling_down = solves left on the streaming iteration after a team timed
// out (0: the resident kernel is in use), *fallbacks = how many launches were given up and re-solved by the streaming iteration.
extern "C" int elph_wg_status(elph_handle h, int *cooling_down, int64_t *fallbacks) {
    CHECK_H(h);
    if (cooling_down) *cooling_down = h->wg_broken ? h->wg_cooldown : 0;
    if (fallbacks) *fallbacks = h->wg_fallbacks;
    return ELPH_OK;
}

extern "C" int elph_bench_wg_info(elph_handle h, int nrhs, int *usable, int *T, int *W, int *G) {
    CHECK_H(h);
    if (!usable || nrhs < 1) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    int t = 0, w = 0, g = 0;
    *usable = (!h->wg_broken && elph_wg_usable(h, &t, &w, &g, nrhs)) ? 1 : 0;
    if (T) *T = t;
    if (W) *W = w;
    if (G) *G = g;
    return ELPH_OK;
}

extern "C" int elph_bench_px_info(elph_handle h, int *fused) {
    CHECK_H(h);
    if (!fused) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    *fused = h->px_solve ? (h->sq16_ap_ran ? 2 : 1) : 0;      // 2: with the register-exchange k_cg_ap of the 16 x 16 lattice (cg_sq16.hip)
    return ELPH_OK;
}

extern "C" int elph_bench_pg_info(elph_handle h, int *kind, int *px, int *py, int *nw, int *tables) {
    CHECK_H(h);
    if (kind) *kind = h->pg_L > 0 ? h->pg_kind : 0;
    if (px) *px = h->pg_PX;
    if (py) *py = h->pg_PY;
    if (nw) *nw = h->pg_NW > 1 ? h->pg_NW : 1;
    if (tables) {
        ModelDev m = elph_model_dev(h);
        *tables = (h->pg_L > 0 && h->kind == ELPH_MODEL_HOLSTEIN && !m.uniform && elph_pg_disorder_ok(h)) ? 1 : 0;
    }
    return ELPH_OK;
}

extern "C" int elph_bench_run(elph_handle h, int what, int nrhs, int reps, int use_graph, double *ms_total) {
    CHECK_H(h);
    if (nrhs < 1 || nrhs > h->cap_rhs || reps < 1 || !ms_total || what < 0 || what > 12) { elph_set_error("bad argument"); return ELPH_E_ARG; }
    if (what == 12) {        // `reps` un-preconditioned iterations of every right-hand side in the slab form of a large lattice (slabs.hip): sum of the launches' event times
        const int prev_broken = h->wg_broken;
        h->wg_broken = false;
        const bool ok = elph_i_slabs_usable(h, nrhs);
        h->wg_broken = prev_broken;
        if (!ok) { elph_set_error("the slab form does not apply to this handle / batch"); return ELPH_E_UNSUPPORTED; }
        bool ran = false;
        std::vector<int64_t> its((size_t)nrhs);
        RC(elph_i_slabs_solve(h, nrhs, h->cur_params, reps, its.data(), &ran, ms_total));
        if (!ran) { elph_set_error("the slab form gave up (time-out)"); return ELPH_E_HIP; }
        return ELPH_OK;
    }
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    int rc = ELPH_OK;
    if (what == 9 || what == 10) {        // `reps` iterations of the whole batch in one launch of the workgroup-resident kernel (10: the preconditioned one)
        bool ran = false, aborted = false;
        CgBufs B = elph_make_bufs(h, nrhs);
        B.params = h->cur_params;
        hipError_t er = hipStreamSynchronize(h->stream);
        if (er == hipSuccess) er = hipEventRecord(e0, h->stream);
        if (er == hipSuccess) {
            rc = elph_wg_cooldown_step(h);
            h->wg_x0_zero = h->bench_fresh && h->x_zero_seen;      // (x is known to be zero only right after elph_bench_prepare)
            h->bench_fresh = false;
            if (rc == ELPH_OK) rc = (what == 9) ? elph_wg_cg(h, B, nrhs, reps, &ran) : elph_pcg_wg(h, B, nrhs, reps, &ran);
            if (rc == ELPH_OK && !ran) { elph_set_error("the workgroup-resident kernel does not apply to this handle"); rc = ELPH_E_UNSUPPORTED; }
        }
        if (er == hipSuccess && rc == ELPH_OK) er = hipEventRecord(e1, h->stream);
        if (er == hipSuccess && rc == ELPH_OK) er = hipEventSynchronize(e1);
        float ms = 0.f;
        if (er == hipSuccess && rc == ELPH_OK) er = hipEventElapsedTime(&ms, e0, e1);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (er != hipSuccess) { elph_set_error("bench (workgroup-resident): %s", hipGetErrorString(er)); return ELPH_E_HIP; }
        if (rc) return rc;
        RC(elph_wg_aborted(h, &aborted));
        if (aborted) return ELPH_E_HIP;
        *ms_total = (double)ms;
        return ELPH_OK;
    }
    if (what == 11) {        // `reps` preconditioned iterations of the batch as two half-batches on two streams (run_cg's form from 192 right-hand sides)
        if (!split_legal(h, nrhs, 1, false)) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); elph_set_error("the two-stream form does not apply to this batch"); return ELPH_E_UNSUPPORTED; }
        SplitRun S;
        hipError_t er = hipStreamSynchronize(h->stream);
        if (er == hipSuccess) er = hipEventRecord(e0, h->stream);
        if (er == hipSuccess) rc = split_begin(h, nrhs, S);
        for (int r = 0; r < reps && rc == ELPH_OK && er == hipSuccess; ++r) rc = split_iteration(S, 1);
        if (rc == ELPH_OK && er == hipSuccess) rc = split_join(S);
        if (S.on) h->ap_count = S.view[S.ways - 1]->ap_count;
        S.ok = (rc == ELPH_OK && er == hipSuccess);
        if (er == hipSuccess && rc == ELPH_OK) er = hipEventRecord(e1, h->stream);
        if (er == hipSuccess && rc == ELPH_OK) er = hipEventSynchronize(e1);
        float ms = 0.f;
        if (er == hipSuccess && rc == ELPH_OK) er = hipEventElapsedTime(&ms, e0, e1);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (er != hipSuccess) { elph_set_error("bench (two streams): %s", hipGetErrorString(er)); return ELPH_E_HIP; }
        if (rc) return rc;
        *ms_total = (double)ms;
        return ELPH_OK;
    }
    hipGraphExec_t exec = nullptr;
    const int chunk = ELPH_CG_CHUNK;
    if (use_graph && h->use_graph && reps % chunk == 0) {
        // capture `chunk` units once (not cached: the bench owns it)
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            elph_set_error("elph_bench_run: stream capture did not start");
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            return ELPH_E_HIP;
        }
        for (int i = 0; i < chunk && rc == ELPH_OK; ++i) rc = bench_launch_unit(h, what, nrhs);
        hipError_t e = hipStreamEndCapture(h->stream, &graph);
        if (rc == ELPH_OK && e != hipSuccess) { elph_set_error("capture: %s", hipGetErrorString(e)); rc = ELPH_E_HIP; }
        if (rc == ELPH_OK) {
            e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            if (e != hipSuccess) { elph_set_error("instantiate: %s", hipGetErrorString(e)); rc = ELPH_E_HIP; }
        }
        if (graph) (void)hipGraphDestroy(graph);
    }
    if (rc == ELPH_OK) {
        hipError_t er = hipStreamSynchronize(h->stream);
        if (er == hipSuccess) er = hipEventRecord(e0, h->stream);
        if (er == hipSuccess) {
            if (exec) {
                for (int r = 0; r < reps / chunk && er == hipSuccess; ++r) er = hipGraphLaunch(exec, h->stream);
            } else {
                for (int r = 0; r < reps && rc == ELPH_OK; ++r) rc = bench_launch_unit(h, what, nrhs);
            }
        }
        if (er == hipSuccess) er = hipEventRecord(e1, h->stream);
        if (er == hipSuccess) er = hipEventSynchronize(e1);
        float ms = 0.f;
        if (er == hipSuccess) er = hipEventElapsedTime(&ms, e0, e1);
        if (er != hipSuccess) { elph_set_error("elph_bench_run: %s", hipGetErrorString(er)); rc = ELPH_E_HIP; }
        else *ms_total = (double)ms;
    }
    if (exec) (void)hipGraphExecDestroy(exec);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}
