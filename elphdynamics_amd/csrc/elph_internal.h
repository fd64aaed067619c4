// elph_internal.h — internal declarations of libelphgpu (gfx950 only).
//
// Device-side data layout ("layout S", slice-major): a lattice vector is stored as
//     v_S[tau * N + site]
// i.e. one imaginary-time slice (all N sites) is contiguous.  The reference / C-ABI layout
// ("layout R", Utilities.jl:12-15) is v_R[site * L + tau].  Host and `_dev` entry points
// convert with a tiled transpose kernel; everything between (CG vectors, expV, KPM scratch)
// stays in layout S, so that
//   * one wavefront owns one tau-slice: its loads/stores are contiguous 8*N bytes,
//   * the checkerboard sweep (couples sites at fixed tau) runs in that wave's LDS with no
//     cross-wave synchronisation,
//   * the tau-FFT output nu[omega][site] is already the omega-major layout the per-omega
//     Chebyshev recursion of the KPM preconditioner wants (no transposes).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/elph_gpu.h"
#include "elph_bench.h"

#define ELPH_WAVE 64
#define ELPH_MAX_NPL 8          // sites per thread
#define ELPH_MAX_SITES 8192      // generic kernels: workgroups of up to 1024 threads x 8 sites
#define ELPH_CG_CHUNK 16        // CG iterations per captured graph launch

void elph_set_error(const char *fmt, ...);

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            elph_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return ELPH_E_HIP;                                                               \
        }                                                                                    \
    } while (0)

// Model description handed to kernels by value.
struct ModelDev {
    int N, L, nb, ncol;
    int cs_tau_stride;   // 0 (Holstein: c,s per bond) or nb (SSH: c,s per (tau,bond))
    int E_tau_stride;    // N (Holstein: E per (tau,site)) or 0 (SSH: E per site)
    int nchains;         // independent phonon configurations resident in E (right-hand side r uses chain r % nchains)
    long long E_chain_stride;   // ndim (Holstein) between chains; unused when nchains == 1
    const int *bi;       // [nb] 0-based first site of bond, checkerboard order
    const int *bj;       // [nb]
    const int *coloff;   // [ncol+1] colour boundaries into the bond list
    const double *c;     // cosh table
    const double *s;     // sinh table
    const double *E;     // exp(-dtau V) (Holstein, layout S) or exp(dtau mu) (SSH)
    // "lane program" (cg_fast.hip): bonds re-packed [colour][pass][lane], PP = ceil(npl/2) passes per colour,
    // NE = MC*PP entries (MC = 4 or 6 colours); idle slots point at the lane's padding pair.  Only valid when ncol <= 6.
    const unsigned *lp_ij;   // [NE][64]  i | j << 16
    const double *lp_c;      // [1 or L][NE][64]
    const double *lp_s;
    int lp_tau_stride;       // 0 (Holstein) or NE*64 (SSH)
    // SSH with several resident phonon configurations: one set of hopping tables per chain (0: shared tables)
    long long cs_chain_stride, lp_chain_stride;
    // every bond carries the same (cosh, sinh) (Holstein without hopping disorder): kernels may keep them in two scalars
    int uniform;
    double c_uni, s_uni;
    // 16 x 16 square lattice in the reference's colouring: the bond that covers site i in colour col ([4][N]; nullptr otherwise)
    const int *sq_bond;
    // a periodic square lattice of (2 grid_GX) x (2 grid_GY) sites in that colouring (square: GX = GY = L / 2; the slab of a sharded solve:
    // 16 x rows): the lane grid of the GRID layout (cg_fast_common.h); 0 otherwise
    int grid_GX, grid_GY;
    // a periodic honeycomb lattice of hc_LX x hc_LY two-site cells in the reference's colouring (detect_honeycomb); 0 otherwise
    int hc_LX, hc_LY;
};
#ifdef __HIPCC__
// the hopping tables of the chain right-hand side `rhs` belongs to (SSH chains; no-op otherwise)
__device__ __forceinline__ void ssh_chain_select(ModelDev &m, int rhs) {
    if (m.cs_chain_stride) {
        const long long c = rhs % m.nchains;
        m.c += c * m.cs_chain_stride; m.s += c * m.cs_chain_stride;
        m.lp_c += c * m.lp_chain_stride; m.lp_s += c * m.lp_chain_stride;
    }
}
#endif

// internal only (never crosses the C ABI): a resident launch gave up at its time-out — the caller owns the fallback.  Distinct from ELPH_E_HIP so
// that a real device error (event, copy, launch) is never mistaken for a time-out and hidden behind the streaming fallback.
#define ELPH_I_ABORTED (-100)
int elph_i_resident_wg_limit(const elph_handle_s *h);      // cg_wg.hip: workgroups of a resident kernel that can be co-resident on the handle's device

// Solver parameters travel BY VALUE in the kernel arguments (never through a small H2D copy + scalar load).
#define ELPH_SPLIT_MAX 8      // parts of a preconditioned batch on streams of their own (elph_api.hip: SplitRun)

struct CgParams {
    double tol, kmax;
    long long maxiter;
    int use_prec;
    int record_hist;
    long long hist_stride;
};

// Buffers of the CG iteration kernels (all layout S).
struct CgState;
struct CgBufs {
    double *x, *r, *z, *zp;     // [nrhs][ndim]; zp = P^-1 r (preconditioned only)
    double *p;                  // [2][nrhs][ndim] ping-pong by (seq & 1)
    double *pap, *rr, *rz;      // partial sums [nrhs][npart]
    double *alpha;              // [nrhs] step length of the last k_cg_xr (fast family: x += alpha p is applied by the NEXT k_cg_ap)
    CgState *state;             // [nrhs][2]
    CgParams params;
    double *hist;               // optional eps history
    int dot_lo, dot_hi;         // sites [dot_lo, dot_hi) enter the inner products (whole slice unless a spatial shard)
    int nrz;                    // number of r.z partials per rhs
    int npap;                   // number of p.z partials per rhs (L, or L/T for the chunked kernel)
    int nrhs;
};

// CG state of one right-hand side; two copies, the newer one has the larger seq.
struct CgState {
    double rho;       // r.z (or r.r) belonging to the current search direction
    double kmin;      // running lower bound of the condition number (IterativeSolvers.jl:289)
    double eps0;      // |r0|/|b|
    double normb;     // |b|
    double eps;       // last evaluated |r|/|b|
    long long seq;    // number of k_cg_ap launches that advanced this state
    long long iters;  // completed CG iterations
    int done;         // 0 running, 1 eps<tol, 2 kmin>kmax, 3 maxiter reached
    int pad;
};


// one frequency block of a chain's schedule, packed so that a Chebyshev block learns what it has to do from ONE 32-byte load
// (frequency, order, offset of its coefficients, leading coefficient) instead of a chain of four dependent ones
struct KpmDesc { int w, order, coff, pad; double c0x, c0y; };

struct KpmDev {
    int active;
    int Lo2;
    int nchains;              // one expansion per phonon configuration (chain); right-hand side r uses chain r % nchains
    double lam_avg, lam_mag;  // chain 0 by value (the single-chain kernels never wait for a load of them)
    const double *lam;        // [nchains][2] lam_avg, lam_mag
    const double *Ebar;       // [nchains][N]
    const double *cbar;       // [nb]
    const double *sbar;       // [nb]
    const int *order;         // [nchains][Lo2]
    const int *coff;          // [nchains][Lo2+1] offsets into coeff (absolute: a chain's base is included)
    const double2 *coeff;     // [sum over chains of sum order]
    const int *wsched;        // [nchains][Lo2] omega indices sorted by decreasing order (longest first)
    const KpmDesc *desc;      // [nchains][Lo2] the same schedule, packed (entry y: frequency wsched[y])
    const double *lp_cbar;    // lane-program copies of cbar/sbar [NE][64]
    const double *lp_sbar;
    // SSH chains: every chain has its own tau-averaged hopping (cbar, sbar) — strides between chains, 0 when shared (Holstein)
    long long hop_stride, lp_hop_stride, sq_stride;
};

// One chain's view of the expansion (device side).
struct KpmChainView {
    const int *order, *coff, *wsched;
    const KpmDesc *desc;
    const double *Ebar;
    double a, b;              // 1/lam_mag, lam_avg/lam_mag
    const double *cbar, *sbar, *lp_cbar, *lp_sbar;    // this chain's averaged hopping tables
    bool active;              // false: this chain's preconditioner is the identity (order 1, coefficient 1 everywhere)
};
#ifdef __HIPCC__
__device__ __forceinline__ KpmChainView kpm_chain_view(const KpmDev &K, int rhs, int N) {
    KpmChainView V;
    if (K.nchains > 1) {
        const int c = rhs % K.nchains;
        V.order = K.order + (size_t)c * K.Lo2;
        V.coff = K.coff + (size_t)c * (K.Lo2 + 1);
        V.wsched = K.wsched + (size_t)c * K.Lo2;
        V.desc = K.desc + (size_t)c * K.Lo2;
        V.Ebar = K.Ebar + (size_t)c * N;
        const double avg = K.lam[2 * c], mag = K.lam[2 * c + 1];
        V.a = 1.0 / mag;
        V.b = avg / mag;
        V.active = mag > 0.0;                                  // the host uploads a negative magnitude for an inactive chain
        V.cbar = K.cbar + c * K.hop_stride; V.sbar = K.sbar + c * K.hop_stride;
        V.lp_cbar = K.lp_cbar + c * K.lp_hop_stride; V.lp_sbar = K.lp_sbar + c * K.lp_hop_stride;
    } else {
        V.cbar = K.cbar; V.sbar = K.sbar; V.lp_cbar = K.lp_cbar; V.lp_sbar = K.lp_sbar;
        V.order = K.order; V.coff = K.coff; V.wsched = K.wsched; V.desc = K.desc; V.Ebar = K.Ebar;
        V.a = 1.0 / K.lam_mag;
        V.b = K.lam_avg / K.lam_mag;
        V.active = K.active != 0;
    }
    return V;
}
#endif

struct elph_handle_s {
    int kind = 0, device = 0;
    int64_t N = 0, L = 0, nb = 0, ndim = 0;
    int npl = 0;               // ceil(N/64)
    hipStream_t stream = nullptr;
    bool own_stream = false;

    // model
    int ncol = 0;
    std::vector<int> h_bi, h_bj, h_coloff;
    std::vector<double> h_c, h_s;          // Holstein: [nb]; SSH: tau-major [L][nb]
    int *d_bi = nullptr, *d_bj = nullptr, *d_coloff = nullptr;
    double *d_c = nullptr, *d_s = nullptr, *d_E = nullptr;
    bool have_E = false;
    int nchains = 1;                       // Holstein: independent chains sharing one handle / one batch
    int64_t E_cap = 0;                     // doubles allocated for d_E
    int dot_lo = 0, dot_hi = 0;            // sites counted in the CG inner products (elph_set_dot_range); hi = 0: all
    bool fast_capable = false;             // lane-program kernels possible for this bond table (fast may be switched off)
    int solo_chain = -1;                   // >= 0: kernels see only this chain (single re-solve of one RHS of a chains batch)
    double *d_lam = nullptr;               // [3N] lambda, lambda2, mu staging
    hipStream_t split_stream[8] = {};      // streams 1 … ways-1 + event of the split form of a preconditioned batch (elph_api.hip: SplitRun; [0] unused: the handle's own stream)
    hipEvent_t split_ev = nullptr;
    int T_rhs_hint = 0;                    // > 0: right-hand sides in flight when the slices per wave are chosen (two-stream batches: both halves)
    bool csbar_external = false;           // kpm_setup_core: h_cbar / h_sbar were filled by the caller (elph_i_kpm_setup_csbar)
    bool ebar_external = false;            // kpm_setup_core: d_Ebar was filled by the caller (elph_i_kpm_setup_ebar)
    bool px_solve = false;                 // the current solve's preconditioned iteration is p/x-fused (kernels.hip: px_plan)
    bool px_via_pg = false;                // this solve's p/x-fused iteration takes the patch-form k_cg_ap_pg although the handle is of the lane-program family (six-colour lane programs: triangular lattices up to 16 x 16)
    bool sq16_ap_ran = false;              // the latest p/x-fused k_cg_ap ran in the register-exchange form (cg_sq16.hip)
    // SSH update_model! on the device (elph_update_model_ssh_fields): staging of x, per-phonon tables, slot map
    double *d_ssh_x = nullptr, *d_ssh_par = nullptr, *d_ssh_tbare = nullptr, *d_ssh_bar = nullptr;
    int *d_ssh_cb = nullptr, *d_ssh_slot = nullptr;
    int64_t ssh_nph_cap = 0;
    double *d_mu_ch = nullptr;             // [mu_ch_cap][N] chemical potential per chain (the tuner with chains in lockstep)
    int mu_ch_cap = 0;
    bool mu_per_chain = false;
    int ssh_chain_cap = 1;                 // SSH: chains whose hopping tables (d_c, d_s, d_lp_c, d_lp_s) and fields (d_ssh_x) are allocated
    int ssh_nph = -1;                      // fields of the last device-side update_model!
    double ssh_dtau = 0.0;
    bool cs_host_stale = false;            // SSH: d_c/d_s were produced on the device; h_c/h_s are not current
    // lane program (fast path, ncol <= 4)
    bool fast = false;
    int lp_mc = 4;                         // colours of the lane program the kernels were compiled for: 4 (lp4) or 6 (lp6)
    int lp_ne = 0;
    std::vector<unsigned> h_lp_ij;
    unsigned *d_lp_ij = nullptr;
    double *d_lp_c = nullptr, *d_lp_s = nullptr, *d_lp_cbar = nullptr, *d_lp_sbar = nullptr;
    // even-L square lattice (L = 8 or 16) recognised in the bond table: P = L/8, per-site per-colour coefficients
    int sq_P = 0;
    int sq_L = 0;                          // even-L square lattice (4 <= L <= 16) recognised in the bond table: L (sq_P = L / 8 for L = 8, 16, the sizes with DPP forms)
    int sq_LX = 0, sq_LY = 0;              // periodic LX x LY square lattice (both even, LX LY / 4 <= 64 lanes) recognised: sq_L = LX when LX == LY
    bool sq_uniform = false;               // every bond has the same (cbar, sbar): the Chebyshev kernel keeps them in scalars
    std::vector<int> sq_bond;              // [4][N] bond index touching site s in colour c
    std::vector<int> pg_bond;              // ... of a square lattice in the patch layout (pg_kind 1)
    bool hc_uniform = false;               // ... and its tau-averaged hopping tables are one (cosh, sinh) for every bond (the register-exchange Chebyshev recursion)
    int pg_NW = 0;                         // wavefronts per time slice of the patch kernels (0 / 1: one; round 6: L = 22, 26, 34, 38 and 40 ... 64 take several)
    int pg_L = 0, pg_PX = 0, pg_PY = 0;    // even-L square lattice beyond 16 x 16 in the reference's colouring (detect_square): PX x PY sites per lane (pgrid_dev.h)
    int pg_kind = 0;                       // 1: that square lattice; 2: a honeycomb lattice beyond 16 x 16 cells, PX x PY CELLS per lane (detect_honeycomb);
                                           // 3: an even-L triangular lattice of any size (detect_triangular)
    bool pg_uniform_c = false;             // the model's own hopping table is uniform (known at elph_create; pg_uniform: the averaged one of the KPM set-up)
    bool pg_uniform = false;               // ... and one (cbar, sbar) for every bond
    int hc_L = 0;                          // honeycomb lattice of hc_L x hc_L cells in the reference's colouring (detect_honeycomb); hc12: hc_L == 12
    int hc_LX = 0, hc_LY = 0;              // periodic honeycomb lattice of LX x LY cells recognised: hc_L = LX when LX == LY
    bool hc12 = false;                     // honeycomb lattice of 12 x 12 cells in the reference's colouring (detect_honeycomb12): the DPP form of k_cg_wg
    double *d_sq_cbar = nullptr, *d_sq_sbar = nullptr;   // [4][N]
    int *d_pg_bond = nullptr;                            // [4][N] the same for a square lattice in the patch layout (pg_kind 1): hopping disorder there
    int *d_sq_bond = nullptr;                            // [4][N] device copy of sq_bond (sq_P > 0)
    void *shard = nullptr;                 // ShardState (shard.hip), owned
    void *slabs = nullptr;                 // SlabSet (slabs.hip), owned: slab handles of this lattice on the same device
    bool slabs_tried = false;              // the slab decomposition was attempted (slabs == nullptr: it does not apply)
    bool is_slab = false;                  // this handle IS a slab of another handle (never decomposed again)
    void *hmc = nullptr;                   // HmcState (hmc.hip), owned
    void *greens = nullptr;                // GreensState (greens.hip), owned
    void *d_res = nullptr;                 // control block of the workgroup-resident CG (cg_wg.hip): meeting records, abort word, boundary slices
    size_t res_cap = 0;
    // x = 0 hint: set by the library right after it zeroes d_x for a solve it is about to start (fill!(x, 0) of the callers, HMC.jl:854);
    // run_cg reads AND clears it first thing (an early error return cannot leave it behind for the next solve) and hands it to
    // elph_launch_cg_init (A x0 = 0 needs no mat-vec; x_zero_seen tells the resident kernel not to read x0 either)
    bool x_zero = false, x_zero_seen = false, wg_x0_zero = false;
    bool bench_fresh = false;              // elph_bench_prepare ran and no elph_bench_run(9 | 10) has consumed its zeroed x yet
    bool wg_broken = false;                // a workgroup-resident launch timed out: streaming iteration for the next wg_cooldown solves, then retry
    int wg_cooldown = 0;
    long long wg_fallbacks = 0;            // how many times that happened (elph_wg_status)
    unsigned wg_epoch = 0;                 // next free tag of the meeting records (cg_wg.hip: WgCtl::epoch0)
    size_t wg_abort_off = 0;               // byte offset of the abort word in d_res
    int wg_T = 0, wg_W = 0, wg_G = 0;      // shape of the last workgroup-resident solve (0: none yet)
    long long ap_count = 0;                // k_cg_ap launches since the last cg_init (ping-pong parity)
    int force_T = 0;                       // ELPH_CHUNK_T: 0 auto, 1 never chunk, n > 1 force n slices per wave (template sizes 2/4/5/8/10/16/20 dividing Ltau: the unrolled kernel; any other: k_cg_ap_chunk_rt, ragged last chunk)

    // solver defaults (model.solver)
    double tol = 1e-4, kmax = 1e12;
    int64_t maxiter = 0;

    // workspace
    int cap_rhs = 0;
    double *d_stage_in = nullptr, *d_stage_out = nullptr;  // layout R staging, cap_rhs*ndim
    double *d_b = nullptr, *d_x = nullptr, *d_r = nullptr, *d_z = nullptr, *d_zp = nullptr;
    double *d_p = nullptr;                 // 2 * cap_rhs * ndim (ping-pong)
    double *d_tmp = nullptr;               // cap_rhs*ndim scratch (v''')
    double *d_phi = nullptr, *d_xfield = nullptr;   // force assembly: phi+- (2*ndim), x in layout S (ndim)
    double *d_part = nullptr;              // partial sums: 4 arrays of cap_rhs * L
    CgState *d_state = nullptr;            // cap_rhs * 2
    CgState *h_state = nullptr;            // pinned, cap_rhs * 2
    CgParams cur_params;                   // parameters of the solve in progress (copied into every launch)
    double *d_hist = nullptr;
    int64_t hist_cap = 0;
    double *d_scal = nullptr;              // small scalar scratch (residual norms), 4*cap_rhs
    double *d_alpha = nullptr;             // cap_rhs: CG step length handed from k_cg_xr to the next k_cg_ap
    double *h_scal = nullptr;              // pinned

    // captured CG chunk graphs, keyed by (nrhs, use_prec)
    struct GraphEntry { int nrhs, use_prec; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
    bool use_graph = false;                // hipGraph replay of CG chunks: opt-in (ELPH_USE_GRAPH=1), see DESIGN.md §3
    int chunk = ELPH_CG_CHUNK;             // CG iterations per graph launch (even)
    bool dbg_copy_outside = false;

    // KPM
    bool kpm_created = false, kpm_ready = false;
    int kpm_n = 20;
    double kpm_buf = 0.05, kpm_c1 = 1.0, kpm_c2 = 1.0;
    double lam_lo = 0.0, lam_hi = 2.0, lam_avg = 1.0, lam_mag = 1.0;
    int kpm_active = 1;                    // 0: every chain's expansion is inactive (identity copy path)
    int kpm_nch = 1;                       // chains the expansion tables are built for
    bool kpm_hop_uploaded = false;         // Holstein: c̄, s̄ (= cosh, sinh of the fixed hoppings) are on the device already
    struct KpmChainHost { double lam_lo = 0.0, lam_hi = 2.0; int active = 1; bool fresh = true;
                          std::vector<int> order; std::vector<double> coeff; };
    std::vector<KpmChainHost> kpm_chain;   // per chain: bounds, orders, coefficients (complex interleaved)
    double *d_kpm_start = nullptr;         // Arnoldi start vectors [2][nch][N] + the bounds [nch][2] coming back (kpm_dev.hip)
    int64_t kpm_start_cap = 0;
    std::vector<double> h_Ebar, h_cbar, h_sbar;   // h_Ebar: [kpm_nch][N]; h_cbar, h_sbar: [nb], or [kpm_nch][nb] for SSH chains
    bool kpm_hop_per_chain = false;        // SSH with several chains: averaged hopping tables per chain
    int kpm_hop_cap = 1;                   // chains the device copies of the averaged hopping tables are allocated for
    std::vector<int> h_order, h_coff, h_wsched;   // flattened [kpm_nch][...] images of the device tables
    std::vector<double> h_coeff;           // complex interleaved, all chains
    std::vector<double> h_lam;             // [kpm_nch][2]
    double *d_klam = nullptr;
    int64_t kpm_tab_cap = 0;               // chains the device tables are allocated for
    double *d_Ebar = nullptr, *d_cbar = nullptr, *d_sbar = nullptr;
    int *d_order = nullptr, *d_coff = nullptr, *d_wsched = nullptr;
    KpmDesc *d_kdesc = nullptr;
    std::vector<KpmDesc> h_kdesc;
    double *d_kfold = nullptr;             // [kpm_nch][Lo2][2] order-1 fold of the forward transform: {scale, r.z weight} (dft_mfma.hip: XrFuse)
    std::vector<double> h_kfold;
    double2 *d_coeff = nullptr;
    int64_t coeff_cap = 0;
    double2 *d_nu = nullptr;               // cap_rhs * Lo2 * N complex (half spectrum, omega-major)

    // FFT twiddles
    double2 *d_tw = nullptr;               // [L] exp(-2 pi i k / L)
    double2 *d_theta = nullptr;            // [L] exp(-i pi t / L)
    double2 *d_Tk = nullptr, *d_Tt = nullptr;   // twisted DFT tables [Lo2][L] / [L][Lo2] (dft.hip)
    double2 *d_Pk = nullptr, *d_Pt = nullptr;   // plain DFT tables   [Lh][L]  / [L][Lh]
    // batched tau-DFT on the matrix cores (dft_mfma.hip): W in A-tile order, [which: twisted/plain][fwd/inv]
    struct MfmaTab { double *W = nullptr; int nt = 0, groups = 0; };
    MfmaTab mf[2][2];
    // twisted transform split once over even/odd tau (L % 4 == 0): half-length tables [fwd/inv] + the L/2 twiddles
    MfmaTab mf_r2[2];
    double *d_r2_tw = nullptr;             // [L/2] (cos, -sin) of pi (2k+1) / L
    double *d_diag = nullptr;              // fourier-acceleration diagonal staging
    int64_t diag_cap = 0;
    // long time axes (L > 1024, dft_big.hip): one Cooley-Tukey split L = big_L1 * big_L2, tables and two complex work vectors
    int big_L1 = 0, big_L2 = 0;
    double2 *d_big_W1 = nullptr, *d_big_W2 = nullptr, *d_big_TW = nullptr, *d_big_TH = nullptr, *d_big_a = nullptr, *d_big_b = nullptr;
    size_t big_cap = 0;
};

// ---- launchers implemented in kernels.hip -------------------------------------------------
ModelDev elph_model_dev(const elph_handle_s *h);
KpmDev elph_kpm_dev(const elph_handle_s *h);

// elph_api.hip internals used by hmc.hip
int elph_i_ldiv_core(elph_handle_s *h, int nrhs, int use_prec, int64_t maxiter, int64_t *iters, double *resid, int *flag);
int elph_i_ensure_capacity(elph_handle_s *h, int nrhs);
int elph_i_set_dot_range(elph_handle_s *h, int64_t site_lo, int64_t site_hi);   // inner products over [lo, hi) only (a shard's own sites)
int elph_i_reserve_chains(elph_handle_s *h, int nchains);   // d_E for nchains configurations, h->nchains = nchains
void elph_i_drop_graphs(elph_handle_s *h);
void elph_hmc_free(elph_handle_s *h);
void elph_shard_free(elph_handle_s *h);
// sharded callers (shard.hip): hooks for hmc.hip
bool elph_i_shard_active(const elph_handle_s *h);                           // a shard with its collectives set (or a single rank)
void elph_i_shard_own_range(const elph_handle_s *h, int *lo, int *hi);      // own sites [lo, hi) of the slab
int elph_i_shard_allreduce(elph_handle_s *h, double *buf, int n);
int elph_i_shard_ghost_sync(elph_handle_s *h, double *vecS, int nvec);
int elph_i_shard_ghost_sync_cols(elph_handle_s *h, double *vecS, int nvec, int ncols, const int *gcol, int ngcol, const double *own);
int elph_i_shard_ldiv_dev(elph_handle_s *h, elph_handle_s *hfull, int use_prec, int64_t maxiter, int64_t *iters, double *resid, int *flag);
int elph_i_kpm_setup_ebar(elph_handle_s *h, const double *Ebar_host, const double *b_max, const double *b_min);      // elph_api.hip
elph_handle_s *elph_i_shard_full(const elph_handle_s *h);                  // the full-lattice handle registered with elph_shard_set_full_lattice (or nullptr)
int elph_i_shard_global_ebar(elph_handle_s *h, std::vector<double> &Ebar_global);      // Ē of the whole lattice from every rank's own rows
bool elph_i_shard_has_bonds(const elph_handle_s *h);                                    // elph_shard_set_bonds has been called for this slab's bonds
int elph_i_shard_global_csbar(elph_handle_s *h, std::vector<double> &cs, int64_t *n_bonds);   // [c̄ | s̄] of every bond of the lattice from the owners' tables
int elph_i_kpm_setup_csbar(elph_handle_s *h, const double *cbar_host, const double *sbar_host, const double *b_max, const double *b_min);      // elph_api.hip
int elph_i_shard_solve_pair(elph_handle_s *h, elph_handle_s *hfull, int use_prec, double tol_power, int64_t *iters, int *flag);
void elph_greens_free(elph_handle_s *h);
int elph_launch_r2s(elph_handle_s *h, double *dstS, const double *srcR, int nvec, int ncols = 0);
int elph_launch_s2r(elph_handle_s *h, double *dstR, const double *srcS, int nvec, int ncols = 0);
int elph_launch_expV(elph_handle_s *h, const double *xR, double dtau, int chain = 0);
int elph_launch_ssh_update(elph_handle_s *h, const double *x_dev, int nph, const int *cb0_dev, const double *par_dev,
                           const double *tbare_dev, const int *slot_dev, double dtau, int x_tau_major = 0, int nch = 1);
int elph_launch_cs_bar(elph_handle_s *h, double *cbar_dev, double *sbar_dev, int nch = 1);
int elph_launch_ssh_scatter(elph_handle_s *h, double *F_dev, const double *q_dev, const double *x_dev, const double *par_dev,
                            const int *cb0_dev, int nph, double dtau, int tau_major = 0, double scale = 1.0, int nch = 1);
int elph_i_ssh_upload_params(elph_handle_s *h, int64_t nph, const int64_t *cb_index, const double *t_ph, const double *alpha,
                             const double *alpha2, const double *t_bare_cb, const double *mu);
int elph_launch_mul(elph_handle_s *h, int which /*0 M, 1 MT, 2 MTM*/, double *yS, const double *vS, int nvec);
int elph_launch_cg_init(elph_handle_s *h, int nrhs, int use_prec, bool x_zero = false);   // x_zero: d_x was zeroed by the library for THIS solve
int elph_launch_cg_iteration(elph_handle_s *h, int nrhs, int use_prec);
int elph_launch_cg_kernel(elph_handle_s *h, int nrhs, int which /*0 = k_cg_ap, 1 = k_cg_xr*/);
int elph_launch_residual(elph_handle_s *h, int nrhs);
int elph_launch_cg_init_only(elph_handle_s *h, int nrhs);
int elph_launch_cg_state0_only(elph_handle_s *h, int nrhs);
int elph_launch_cg_init_prec_only(elph_handle_s *h, int nrhs);
int elph_launch_kpm_apply(elph_handle_s *h, double *zS, const double *rS, int nrhs, int cg_mode, int parts = 7);
int elph_launch_ebar(elph_handle_s *h, int nch = 1);
int elph_launch_fft_accel(elph_handle_s *h, double *outS, const double *inS, const double *diagS, double power, int64_t ncol,
                          int nvec = 1);
int elph_launch_tau_to_omega(elph_handle_s *h, double2 *nuS, const double *vS);
int elph_launch_omega_to_tau(elph_handle_s *h, double *vS, const double2 *nuS);
int elph_launch_zero(elph_handle_s *h, double *p, int64_t n);
int elph_launch_lambda_rhs(elph_handle_s *h, double *bS, const double *phiS, const double *xS, double dtau, int nch = 1);
int elph_launch_force_ssh(elph_handle_s *h, double *q, const double *XS, const double *US = nullptr, int nch = 1);
int elph_launch_dmdx_holstein(elph_handle_s *h, double *FS, const double *uS, const double *vS, const double *xS, double dtau,
                              double scale, int nch = 1);
int elph_launch_force_holstein(elph_handle_s *h, double *FS, const double *XS, const double *phiS, const double *xS, double dtau,
                               int nch = 1);

// ---- fast path (cg_fast.hip) ----------------------------------------------------------------
int elph_fast_mul(elph_handle_s *h, int which, double *yS, const double *vS, int nvec);
int elph_fast_cg_ap(elph_handle_s *h, const CgBufs &B, int nrhs, int parity, bool fused = false);      // fused: the p/x-fused iteration (reads the ready p)
int elph_fast_cg_xr(elph_handle_s *h, const CgBufs &B, int nrhs, int parity);
// cg_sq16.hip: the p/x-fused k_cg_ap of the 16 x 16 square lattice with the checkerboard in registers (no LDS slabs)
bool elph_sq16_ap_usable(const elph_handle_s *h, int T);
int elph_sq16_cg_ap_px(elph_handle_s *h, const CgBufs &B, int nrhs, int parity);
// ---- one solve over several GPUs (cg_wg.hip, shard.hip): by-value description of this rank's shard for the resident kernel
#define ELPH_SHARD_MAXREC 256      // records of a meeting: ranks x workgroups per rank (8 x 20 at Ltau = 160; polled as 8 x 64 granules)
#define ELPH_SHARD_MAXRANKS 8
struct ElphShardCtl {
    int rank = 0, P = 1;
    int own_lo = 0, own_hi = 0;       // own sites [own_lo, own_hi) of the slab lattice; [0, own_lo) and [own_hi, N) are ghosts
    int n_to_prev = 0, n_to_next = 0; // own sites (from the bottom / from the top) that the previous / next rank holds as ghosts
    int cap_ghost = 0;                // capacity (sites) of a ghost region of the mailbox, the same on all ranks
    unsigned long long *mail[ELPH_SHARD_MAXRANKS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};
extern "C" int elph_shard_shape(int64_t ltau, int world, int *waves, int *groups, int *records, int *max_records);   // shard.hip
int elph_wg_cg_shard(elph_handle_s *h, const CgBufs &B, long long fixed_iters, const ElphShardCtl &Sh, int *G_out);
// all ranks of a sharded solve whose slabs live on ONE device in one launch (cg_wg.hip; h_args pinned, d_args device: P x elph_wg_rank_args_bytes())
size_t elph_wg_rank_args_bytes();
int elph_wg_cg_ranks(elph_handle_s *const *hs, int P, const CgBufs *Bs, long long fixed_iters, const ElphShardCtl *ctls, void *h_args,
                     void *d_args, hipStream_t stream, long long timeout_ms, int *G_out);
int elph_i_shard_run_ranks(elph_handle_s *const *hs, int P, int nsets, void *h_args, void *d_args, double tol, int64_t maxiter, double kmax,
                           long long fixed_iters, long long timeout_ms, CgState *state_out, double *ms_out, double *const *hist_dev = nullptr,
                           long long hist_stride = 0);      // shard.hip
int elph_i_shard_create_local(elph_handle_s *h, int rank, int world, int64_t own_lo, int64_t own_n, int64_t n_to_prev, int64_t n_to_next,
                              int64_t cap_ghost, void *key_out);      // shard.hip: elph_shard_create for slabs that all live on one device (mailbox = ordinary device memory, not exported)
// ---- slabs.hip: the resident un-preconditioned solve of a lattice BEYOND one wave's slice (N > 320) as slabs of rows on the same device
bool elph_i_slabs_usable(elph_handle_s *h, int nrhs);      // builds the slabs on first use
int elph_i_slabs_solve(elph_handle_s *h, int nrhs, const CgParams &P, long long fixed_iters, int64_t *iters, bool *ran, double *ms_out);      // (P.record_hist: the eps histories go to h->d_hist)
void elph_i_slabs_free(elph_handle_s *h);

// ---- workgroup-resident CG (cg_wg.hip): the whole un-preconditioned solve in one launch
bool elph_wg_usable(const elph_handle_s *h, int *T, int *W, int *G, int nrhs = 1);
int elph_wg_cg(elph_handle_s *h, const CgBufs &B, int nrhs, long long fixed_iters, bool *ran);
int elph_wg_aborted(elph_handle_s *h, bool *aborted);       // after the stream has drained
bool elph_pg_cheb_usable(const elph_handle_s *h);
bool elph_pg_disorder_ok(const elph_handle_s *h);      // hopping disorder on this handle's patch shape (pgrid.hip)                       // pgrid.hip
int elph_pg_kpm_cheb(elph_handle_s *h, int nrhs, const CgState *st, double *rz_part = nullptr, int nrz = 0, const double *rr_part = nullptr);
bool elph_pg_ap_usable(const elph_handle_s *h);
bool elph_pg_mul_usable(const elph_handle_s *h);
int elph_pg_mul(elph_handle_s *h, const ModelDev &m, int which, double *yS, const double *vS, int nvec);
int elph_pg_cg_ap(elph_handle_s *h, const CgBufs &B, const ModelDev &m, int nrhs, int parity, bool fused = false);      // fused: the p/x-fused iteration (reads the ready p)
int elph_wg_cooldown_step(elph_handle_s *h);                // one solve of the cool-down after a time-out (both resident kernels call it)
long long elph_shard_timeout_ms();                          // wait bound of the sharded solves (shard.hip)
// ---- workgroup-resident KPM-preconditioned CG (pcg_wg.hip): the whole preconditioned solve of 1..8 right-hand sides in one launch
bool elph_pcg_wg_usable(const elph_handle_s *h, int nrhs);
int elph_pcg_wg(elph_handle_s *h, const CgBufs &B, int nrhs, long long fixed_iters, bool *ran);
CgBufs elph_make_bufs(elph_handle_s *h, int nrhs);
int elph_fast_kpm_cheb(elph_handle_s *h, int nrhs, const CgState *st, double *rz_part = nullptr, int nrz = 0, bool *did_rz = nullptr,
                       const double *rr_part = nullptr, int fold_nct = 0);
int elph_choose_T(const elph_handle_s *h, int nrhs);
int elph_choose_T_px(const elph_handle_s *h, int nrhs);      // the p/x-fused kernel's own rule (three waves per SIMD)
// packs per-bond values (order of h_bi/h_bj) into the lane-program layout [NE][64] (idle slots = fill)
void elph_lp_pack(const elph_handle_s *h, const double *per_bond, double *out, double fill);

// ---- tau-axis transforms (dft.hip) -------------------------------------------------------------
int elph_dft_build_tables(elph_handle_s *h);
int elph_dft_fwd_twisted(elph_handle_s *h, double2 *nu, const double *vS, int N, int nrhs, const CgState *st);
int elph_dft_inv_twisted(elph_handle_s *h, double *outS, const double2 *nu, int N, int nrhs, const CgState *st,
                         const double *rvec, double *rz_part, int nrz);
int elph_dft_fwd_plain(elph_handle_s *h, double2 *nu, const double *vS, int N, int nrhs);
int elph_dft_inv_plain(elph_handle_s *h, double *outS, const double2 *nu, int N, int nrhs);
int elph_dft_accel(elph_handle_s *h, double *outS, const double *inS, const double *diagS, double power, int N, double2 *u,
                   int nvec = 1);

// ---- batched tau-axis transforms on the matrix cores (dft_mfma.hip); which: 0 twisted, 1 plain
int elph_dft_mfma_build_tables(elph_handle_s *h);
bool elph_dft_mfma_xr_usable(const elph_handle_s *h, int N, int nrhs);
bool elph_dft_mfma_px_usable(const elph_handle_s *h, int N, int nrhs);
bool elph_px_plan(elph_handle_s *h, int nrhs);      // kernels.hip: would a preconditioned batch of nrhs run p/x-fused?
int elph_dft_mfma_inv_px(elph_handle_s *h, const double2 *nu, int N, int nrhs, const CgState *st, double *pS, double *xS,
                         const double *alpha, const double *rz, int nrz);
bool elph_dft_mfma_fold_usable(const elph_handle_s *h);
bool elph_dft_big(const elph_handle_s *h);
int elph_dft_big_build_tables(elph_handle_s *h);
void elph_dft_big_free(elph_handle_s *h);
int elph_dft_big_fwd(elph_handle_s *h, bool twisted, double2 *nu, const double *vS, int N, int nvec);
int elph_dft_big_inv(elph_handle_s *h, bool twisted, double *outS, const double2 *nu, int N, int nvec, const double *rvec,
                     double *rz_part, int nrz);
bool elph_dft_mfma1_usable(const elph_handle_s *h, bool inverse, int N, int nrz_slots);
int elph_dft_mfma1_fwd(elph_handle_s *h, double2 *nu, const double *vS, int N, int nrhs, const CgState *st);
int elph_dft_mfma1_inv(elph_handle_s *h, double *outS, const double2 *nu, int N, int nrhs, const CgState *st, const double *rvec,
                       double *rz_part, int nrz);
int elph_dft_mfma_fwd_xr(elph_handle_s *h, double2 *nu, double *rS, const double *zS, const double *pap, int npap, double *rr,
                         double *alpha, int N, int nrhs, const CgState *st, const double *fold = nullptr, int fold_nch = 1,
                         double *frz = nullptr, int fnrz = 0, int fslot0 = 0);
void elph_dft_mfma_free(elph_handle_s *h);
bool elph_dft_mfma_usable(const elph_handle_s *h, int which, bool inverse, int N, int nrhs);
int elph_dft_mfma_fwd(elph_handle_s *h, int which, double2 *nu, const double *vS, int N, int nrhs, const CgState *st);
int elph_dft_mfma_inv(elph_handle_s *h, int which, double *outS, const double2 *nu, int N, int nrhs, const CgState *st,
                      const double *rvec, double *rz_part, int nrz);

// ---- host-side KPM setup (kpm_host.cpp) ---------------------------------------------------
void elph_kpm_coefficients(double *c_z, int order, double lam_lo, double lam_hi, double phi);
int elph_kpm_arnoldi(const elph_handle_s *h, int chain, const double *b_max, const double *b_min, double *e_min,
                     double *e_max);
int elph_hess_eigvals(std::vector<double> &a, int n, std::vector<double> &wr, std::vector<double> &wi);
int elph_kpm_bounds_dev(elph_handle_s *h, int nch, const double *d_bstart, double *d_eout);
