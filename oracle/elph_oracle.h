/*
 * elph_oracle.h — CPU restatement (plain C) of the ElPhDynamics hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle and the CPU baseline.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  Nothing under elphdynamics_amd/ links, imports or calls it.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference (Julia) ships no tests,
 * golden vectors or fixtures and cannot be run in this environment (no Julia).
 * The oracle is pinned instead by self-derived known answers (dense M from its
 * block definition, adjointness, CB*CB^-1 = I, single-site closed form,
 * scipy.fft, numpy dense solves) — see tests/golden/make_golden.py.
 *
 * Conventions follow the reference exactly:
 *   - vectors are flat double[N*L], tau fastest: idx(site,tau) = site*L + tau (0-based)
 *     (reference: src/Utilities.jl:12-15, 1-based)
 *   - integer tables carry the reference's 1-based site numbers, column-major
 *     2 x Nbonds (int64), so they can be compared bit-for-bit with Julia dumps.
 *
 * Each function cites the reference file:line it restates.
 */
#ifndef ELPH_ORACLE_H
#define ELPH_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- geometry */

/* Lattices.jl:149-168 (+ _pbc! :384-391). orbit is 1-based, returns 1-based site. */
int64_t elpho_loc_to_site(int64_t norbits, int64_t L1, int64_t L2, int64_t L3,
                          int64_t orbit, int64_t l1, int64_t l2, int64_t l3);

/* Lattices.jl:176-191. isite 1-based. */
int64_t elpho_site_to_site(int64_t norbits, int64_t L1, int64_t L2, int64_t L3,
                           int64_t isite, const int64_t d[3], int64_t orbit);

/* Lattices.jl:265-316. table must hold 2*ncells int64 (column-major 2 x n).
 * Returns the number of bonds kept after duplicate removal. */
int64_t elpho_calc_neighbor_table(int64_t norbits, int64_t L1, int64_t L2, int64_t L3,
                                  int64_t o1, int64_t o2, const int64_t d[3],
                                  int remove_duplicates, int64_t *table);

/* Stable ascending sortperm (Julia sortperm semantics), perm is 1-based. */
void elpho_sortperm(const int64_t *keys, int64_t n, int64_t *perm);

/* Lattices.jl:323-340: orient rows (row1<row2) in place, return sorting perm (1-based). */
void elpho_sorted_neighbor_table_perm(int64_t *table, int64_t nb, int64_t *perm);

/* Checkerboard.jl:471-515: greedy colouring; groups[nb] 1-based colours; returns #colours. */
int64_t elpho_checkerboard_groups(const int64_t *table, int64_t nb, int64_t *groups);

/* HolsteinModels.jl:484-517: cosh/sinh, sort, colour, permute.
 * table (in/out, 2 x nb), t (in, nb), cosht/sinht (out), cb_perm (out, 1-based),
 * groups_sorted (out, colour of each bond in final order; may be NULL).
 * Returns number of colours. */
int64_t elpho_holstein_initialize_model(int64_t *table, int64_t nb, const double *t, double dtau,
                                        double *cosht, double *sinht, int64_t *cb_perm,
                                        int64_t *groups_sorted);

/* SSHModels.jl:435-448: sort + colour + permute the table only.
 * inv_cb_perm = perm[new_perm], cb_perm = sortperm(inv_cb_perm) (both 1-based). */
int64_t elpho_ssh_initialize_table(int64_t *table, int64_t nb, int64_t *cb_perm,
                                   int64_t *inv_cb_perm, int64_t *groups_sorted);

/* HolsteinModels.jl:205: Ltau = round(Int, beta/dtau) (ties to even). */
int64_t elpho_ltau(double beta, double dtau);

/* ------------------------------------------------------------------- model */

typedef struct {
    int64_t kind;          /* 0 = Holstein, 1 = SSH */
    int64_t N;             /* sites */
    int64_t L;             /* Ltau */
    int64_t nb;            /* bonds */
    const int64_t *table;  /* 2 x nb, 1-based, checkerboard order */
    const double *c;       /* Holstein: cosht[nb]; SSH: cosht[L x nb] column-major (tau fastest) */
    const double *s;       /* same for sinh */
    const double *E;       /* Holstein: expnDtauV[N*L]; SSH: expDtauMu[N] */
    double *vp;            /* scratch v'  [N*L]  (Models.jl:218) */
    double *vppp;          /* scratch v''' [N*L] (Models.jl:94,151) */
} elpho_model;

/* HolsteinModels.jl:526-549 */
void elpho_update_model_holstein(int64_t N, int64_t L, double dtau, const double *x,
                                 const double *lambda, const double *lambda2, const double *mu,
                                 double *expV);

/* SSHModels.jl:510-535 (matrix-element part).
 * x[Nph*L] tau fastest; phonon_to_bond, cb_perm 1-based; cosht/sinht are L x nb. */
void elpho_update_model_ssh(int64_t N, int64_t L, int64_t nb, int64_t Nph, double dtau,
                            const double *x, const double *t, const double *alpha,
                            const double *alpha2, const double *mu,
                            const int64_t *phonon_to_bond, const int64_t *cb_perm,
                            double *cosht, double *sinht, double *expDtauMu);

/* Checkerboard.jl:57-83 / 149-175 / 238-264 / 323-349 (vector c,s + Ltau) */
void elpho_checkerboard_mul(double *y, const int64_t *table, const double *c, const double *s,
                            int64_t nb, int64_t L);
void elpho_checkerboard_transpose_mul(double *y, const int64_t *table, const double *c,
                                      const double *s, int64_t nb, int64_t L);
void elpho_checkerboard_inverse_mul(double *y, const int64_t *table, const double *c,
                                    const double *s, int64_t nb, int64_t L);
void elpho_checkerboard_inverse_transpose_mul(double *y, const int64_t *table, const double *c,
                                              const double *s, int64_t nb, int64_t L);
/* Checkerboard.jl:86-121 / 177-210 (matrix c[tau,n]) */
void elpho_checkerboard_mul_mat(double *y, const int64_t *table, const double *c,
                                const double *s, int64_t nb, int64_t L);
void elpho_checkerboard_transpose_mul_mat(double *y, const int64_t *table, const double *c,
                                          const double *s, int64_t nb, int64_t L);
/* Checkerboard.jl:123-141 / 212-230 / 298-316: N-vector forms, complex y (re,im interleaved) */
void elpho_checkerboard_mul_nvec_z(double *y, const int64_t *table, const double *c,
                                   const double *s, int64_t nb);
void elpho_checkerboard_transpose_mul_nvec_z(double *y, const int64_t *table, const double *c,
                                             const double *s, int64_t nb);
void elpho_checkerboard_mul_nvec(double *y, const int64_t *table, const double *c,
                                 const double *s, int64_t nb);
void elpho_checkerboard_inverse_mul_nvec(double *y, const int64_t *table, const double *c,
                                         const double *s, int64_t nb);

/* HolsteinModels.jl:569-626,631-684 / SSHModels.jl:581-640,646-701 */
void elpho_mulM(double *y, const elpho_model *m, const double *v);
void elpho_mulMT(double *y, const elpho_model *m, const double *v);
/* Models.jl:215-224 (uses m->vp) */
void elpho_mulMTM(double *y, const elpho_model *m, const double *v);

/* ------------------------------------------------------ twisted FFT (a17) */

/* TimeFreqFFTs.jl:55-73: out[N*L] complex (interleaved) = FFT_tau(Theta .* in), in real. */
void elpho_tau_to_omega(double *out_z, const double *in, int64_t N, int64_t L);
/* TimeFreqFFTs.jl:112-130: out real = Re(conj(Theta) .* iFFT_tau(in)) */
void elpho_omega_to_tau(double *out, const double *in_z, int64_t N, int64_t L);
/* plain batched DFT along tau on complex data; sign=-1 forward unnormalised,
 * sign=+1 inverse scaled 1/L (FFTW.jl fft / ifft conventions). */
void elpho_dft_tau(double *out_z, const double *in_z, int64_t N, int64_t L, int sign);

/* ------------------------------------------------ Fourier acceleration (a21) */

/* FourierAcceleration.jl:260-266 / 213-217 */
double elpho_element_Mi(int64_t k, double omega, double dtau, double m0, double c, int64_t L);
double elpho_element_Qi(int64_t k, double omega, double dtau, double m, int64_t L);
/* FourierAcceleration.jl:176-240: fill diag[N*L] for phonons with wmin<omega<wmax */
void elpho_update_M(double *Mdiag, int64_t Nph, int64_t L, double dtau, const double *omega,
                    double wmin, double wmax, double m0, double c);
void elpho_update_Q(double *Qdiag, int64_t Nph, int64_t L, double dtau, const double *omega,
                    double wmin, double wmax, double m);
/* FourierAcceleration.jl:91-143 (real in, real out): out = Re iFFT(diag^power .* FFT(in)) */
void elpho_fourier_accelerate(double *out, const double *in, const double *diag, double power,
                              int64_t N, int64_t L);

/* ------------------------------------------------------------- KPM (a18-a20) */

typedef struct {
    int64_t active;        /* KPMPreconditioners.jl:24 */
    int64_t N, L, nb;
    const int64_t *table;  /* model.neighbor_table */
    double *Ebar;          /* [N]  expnDtauVbar */
    double *cbar;          /* [nb] coshtbar */
    double *sbar;          /* [nb] sinhtbar */
    double lam_lo, lam_hi, lam_avg, lam_mag;
    double buf, c1, c2;
    int64_t Lo2;           /* cld(L,2) */
    int64_t *order;        /* [Lo2] */
    int64_t *coff;         /* [Lo2+1] offsets into coeff (in complex elements) */
    double *coeff;         /* complex interleaved, sum(order) entries; capacity coeff_cap */
    int64_t coeff_cap;
    double *v1, *v2;       /* complex [N*L] scratch */
    double *v3, *v4, *v5;  /* complex [N] scratch */
    int64_t checkerboard_count;
} elpho_kpm;

/* KPMPreconditioners.jl:332-349 (Holstein) / 355-381 (SSH) */
void elpho_kpm_update_A(elpho_kpm *P, const elpho_model *m);
/* KPMPreconditioners.jl:789-839,948-951: coefficients c[order] complex interleaved */
void elpho_kpm_coefficients(double *c_z, int64_t order, double lam_lo, double lam_hi, double phi);
/* KPMPreconditioners.jl:845-942 with the two random start vectors injected
 * (b_max[N], b_min[N]); n = Krylov dimension.  Returns e_min, e_max. */
void elpho_kpm_arnoldi_bounds(const elpho_kpm *P, int64_t n, const double *b_max,
                              const double *b_min, double *e_min, double *e_max);
/* KPMPreconditioners.jl:269-321 given (e_min,e_max): sets active, lam_*, order, coeff. */
void elpho_kpm_setup_from_bounds(elpho_kpm *P, double e_min, double e_max);
/* KPMPreconditioners.jl:426-481 (+ :606-693, :758-778) */
void elpho_kpm_apply(double *out, elpho_kpm *P, const double *in);
/* eigenvalues (real parts, imag parts) of a small dense real matrix a[n*n] col-major */
int elpho_eigvals(double *a, int64_t n, double *wr, double *wi);

/* --------------------------------------------------------- CG (a13-a16) */

/* IterativeSolvers.jl:239-314 (P==NULL) / :153-234 (P!=NULL).  A = MtM.
 * r,p,z are the solver's work vectors [N*L].  hist (optional, maxiter+1) receives
 * eps_0 .. eps_j.  Returns iterations. */
int64_t elpho_cg_solve(const elpho_model *m, double *x, const double *b, double tol,
                       int64_t maxiter, double kmax, elpho_kpm *P, double *r, double *p,
                       double *z, double *hist);

/* Models.jl:74-137 / 139-186.  solver_tol, solver_maxiter = model.solver.tol/.maxiter;
 * maxiter==0 => solver_maxiter. */
void elpho_ldiv(const elpho_model *m, double *x, const double *b, elpho_kpm *P, int64_t maxiter,
                double solver_tol, int64_t solver_maxiter, double kmax, double *r, double *p,
                double *z, int64_t *iters, double *resid, int64_t *flag);

/* ----------------------------------------------------- callers (a22, a23) */

/* HMC.jl:921-941 */
void elpho_update_Lambda(double *Lam, int64_t N, int64_t L, double dtau, const double *x,
                         const double *lambda, const double *lambda2);
/* HMC.jl:951-968 / 978-995 */
void elpho_mulLambda(double *out, const double *in, const double *Lam, int64_t N, int64_t L);
void elpho_mulLambdaInv(double *out, const double *in, const double *Lam, int64_t N, int64_t L);
/* HolsteinModels.jl:691-755 (uses m->vp as scratch) */
void elpho_muldMdx_holstein(double *dMdx, const double *u, const elpho_model *m, const double *v,
                            double dtau, const double *lambda, const double *lambda2,
                            const double *x);

/* HMC.jl:1005-1025 */
void elpho_muldLambdadx_holstein(double *dLdx, const double *vl, const double *vr, const double *Lam, int64_t N,
                                 int64_t L, double dtau, const double *lambda, const double *lambda2, const double *x);
/* HMC.jl:790-814 (accumulates into dSfdx; u, d: scratch [N*L]) */
void elpho_calc_dSfdx_holstein(double *dSfdx, const elpho_model *m, const double *Xp, const double *Xm,
                               const double *phip, const double *phim, const double *Lam, double dtau,
                               const double *lambda, const double *lambda2, const double *x, double *u, double *d);

/* ------------------------------------------------ HMC trajectory */

/* PhononAction.jl:11-66 / 114-187 (Holstein; shifted = false; no dispersive modes) */
double elpho_calc_Sb_holstein(int64_t N, int64_t L, double dtau, const double *x, const double *omega,
                              const double *omega4);
void elpho_calc_dSbdx_holstein(double *dSbdx, int64_t N, int64_t L, double dtau, const double *x, const double *omega,
                               const double *omega4);

typedef struct {
    int64_t N, L;
    double dtau;
    const double *omega, *omega4, *lambda, *lambda2, *mu;   /* per site */
    const double *fa_M;                                     /* FourierAccelerator.M, [N*L] (omega fastest) */
    double dt;                                              /* hmc.Δt */
    int64_t nt, nb;                                         /* hmc.Nt, hmc.Nb */
    double alpha;                                           /* partial momentum refresh */
    double solver_tol;
    int64_t solver_maxiter;
    double kmax;
    int64_t kpm_n;                                          /* Arnoldi dimension of setup!(P) */
} elpho_hmc_params;

/* HMC.jl:313-638 (update! -> standard_update! / multitimestep_update!) with the random numbers as inputs; see the
 * definition for out[8].  m->E must be writable (update_model! writes it).  Returns accepted (0/1). */
int64_t elpho_hmc_update_holstein(const elpho_hmc_params *hp, elpho_model *m, elpho_kpm *P, double *x, double *v,
                                  const double *R, const double *Rp, const double *Rm, const double *kpm_randn, double u,
                                  double *out);

/* LangevinDynamics.jl:334-384 (calc_dSdx!) and :81-328 (evolve! for Euler / Runge-Kutta / Heun dynamics), Holstein; the
 * random vectors are inputs.  See the definitions. */
int64_t elpho_langevin_dSdx(double *dSdx, const elpho_hmc_params *hp, elpho_model *m, elpho_kpm *P, const double *x, const double *g,
                            const double *b_max, const double *b_min, double *Minv_g, double *work);
int64_t elpho_langevin_evolve(int scheme, const elpho_hmc_params *hp, elpho_model *m, elpho_kpm *P, double *x, const double *fa_Q,
                              double dt, const double *eta, const double *g1, const double *g2, const double *kpm_randn);

/* SSH extras of the HMC update (bond phonons, SSHModels.jl:79-314): per phonon t is indexed by RAW bond as in ssh.t */
typedef struct {
    int64_t Nph;
    const double *t;                 /* [nbonds] bare hopping, raw bond order (ssh.t) */
    const double *alpha, *alpha2;    /* [Nph] */
    const int64_t *phonon_to_bond;   /* [Nph] 1-based raw bond of each phonon */
    const int64_t *cb_perm;          /* [nbonds] checkerboard_perm */
    const int64_t *bond_to_phonon_cb;/* [nbonds] 1-based phonon on checkerboard bond n, 0 = none */
    const int64_t *primary_field;    /* [Nph*L] 0-based ssh.primary_field (SSHModels.jl:480-502); NULL: every field its own */
} elpho_hmc_ssh;

int64_t elpho_hmc_update_ssh(const elpho_hmc_params *hp, const elpho_hmc_ssh *ssh, elpho_model *m, elpho_kpm *P, double *x,
                             double *v, const double *R, const double *Rp, const double *Rm, const double *kpm_randn, double u,
                             double *out);

/* evolve! of LangevinDynamics.jl for the SSH model (x, eta: Nph*L; g1, g2: N*L; fa_Q per phonon) */
int64_t elpho_langevin_evolve_ssh(int scheme, const elpho_hmc_params *hp, const elpho_hmc_ssh *ssh, elpho_model *m, elpho_kpm *P,
                                  double *x, const double *fa_Q, double dt, const double *eta, const double *g1, const double *g2,
                                  const double *kpm_randn);

/* SpecialUpdates.jl:103-136,205-275: one proposed reflection (kind 0) or swap (kind 1) move; see the definition */
int64_t elpho_special_move(const elpho_hmc_params *hp, const elpho_hmc_ssh *ssh, elpho_model *m, elpho_kpm *P, double *x, int kind,
                           int64_t ci, int64_t cj, const double *Rp, const double *Rm, const double *kpm_randn, double u,
                           double *out);


/* SSHModels.jl:707-829 (no equivalent fields); dMdx[Nph*L] is overwritten */
void elpho_muldMdx_ssh(double *dMdx, const double *u, const elpho_model *m, const double *v, double dtau,
                       const int64_t *bond_to_phonon_cb, const double *alpha, const double *alpha2, const double *x,
                       int64_t Nph);

#ifdef __cplusplus
}
#endif
#endif
