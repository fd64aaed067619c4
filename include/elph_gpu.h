/*
 * elph_gpu.h — C ABI of libelphgpu.so, the MI355X (gfx950) fermion-force solver for ElPhDynamics.
 *
 * This is the drop-in boundary for the reference's operator API.  The reference has
 * no FFI: its callers reach the path through Julia multiple dispatch on the model type
 * (Models.jl / IterativeSolvers.jl / KPMPreconditioners.jl / FourierAcceleration.jl).  Each entry
 * point below names the reference method it replaces (file:line under the reference's src/); the
 * Julia-side binding (`ccall`) a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no exceptions across the boundary.
 *   - Every function returns an int status: ELPH_OK (0) or a negative ELPH_E_* code.
 *     Solver outcomes (iterations, residual, flag 0/1/2) are out-parameters, exactly the tuple
 *     the reference's ldiv! returns (Models.jl:74,139) — non-convergence is NOT an error.
 *   - Host entry points take host pointers in the REFERENCE layout: flat double[Nsites*Ltau],
 *     tau fastest, idx = site*Ltau + tau (Utilities.jl:12-15); neighbour table int64[2*Nbonds],
 *     column-major 2 x Nbonds, 1-based, already in checkerboard order (HolsteinModels.jl:484-517).
 *     No host pointer is retained after return.
 *   - `_dev` twins take device pointers in the same reference layout and run on the handle's stream.
 *   - One handle = one model on one GPU; a handle is not reentrant (the reference's model is not
 *     either: shared scratch v', v'', v''', Models.jl:218,94).
 */
#ifndef ELPH_GPU_H
#define ELPH_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ELPH_OK 0
#define ELPH_E_ARG (-1)      /* bad argument / shape mismatch */
#define ELPH_E_HIP (-2)      /* HIP runtime error (elph_last_error() has the text) */
#define ELPH_E_STATE (-3)    /* call out of order (e.g. KPM apply before setup) */
#define ELPH_E_NOGPU (-4)    /* no usable gfx950 device */
#define ELPH_E_UNSUPPORTED (-5)

#define ELPH_MODEL_HOLSTEIN 0
#define ELPH_MODEL_SSH 1

typedef struct elph_handle_s *elph_handle;

/* Text of the last error on this thread (never NULL). */
const char *elph_last_error(void);

/* ABI version this header describes.  Bumped whenever an entry point is removed or changes meaning (2: round 4 removed the
 * elph_cgstep_*, elph_dev_buffer, elph_buffer_* exports and round 5 added elph_muldMdx_*); loaders check for EXACTLY the version they
 * were written against (julia/ElPhGPU.jl, elphdynamics_amd/_lib.py). */
#define ELPH_ABI_VERSION 2
/* ABI version of the loaded library. */
int elph_abi_version(void);
/* How this library was built: "libelphgpu abi=<ELPH_ABI_VERSION> src=<sha256 of csrc/ + include/, 16 hex digits> arch=gfx950 variant=... built=<UTC>
 * flags=... compiler=[hipcc --version]" — static storage.  elphdynamics_amd/build.py compares `src` with the sources at hand: a stale
 * shipped library is rebuilt, a current one reused; smoke() and bench.py print it so that a run's log names the code that ran. */
const char *elph_build_info(void);

/* Number of visible HIP devices (0 if none); does not create a context. */
int elph_device_count(void);

/* ---------------------------------------------------------------- model life cycle */

/* Replaces HolsteinModel / SSHModel construction + initialize_model!
 * (HolsteinModels.jl:192-314,484-517; SSHModels.jl:348-505) for the fields the path uses.
 *   kind            ELPH_MODEL_HOLSTEIN | ELPH_MODEL_SSH
 *   nsites, ltau    Nsites, Ltau (Ndim = nsites*ltau)
 *   nbonds          number of bonds (0 allowed: single-site deck, HolsteinModels.jl:486)
 *   neighbor_table  int64[2*nbonds], 1-based, checkerboard order (model.neighbor_table)
 *   cosht, sinht    Holstein: double[nbonds] (model.cosht/.sinht); may be NULL for SSH
 *                   (SSH matrix elements arrive through elph_update_model_ssh)
 *   device          HIP device ordinal
 * The colour boundaries are recomputed from the table (maximal runs of site-disjoint bonds),
 * which reproduces the reference's groups for any table produced by checkerboard_order!. */
int elph_create(elph_handle *out, int kind, int64_t nsites, int64_t ltau, int64_t nbonds,
                const int64_t *neighbor_table, const double *cosht, const double *sinht, int device);

/* Frees all device memory of the handle. */
int elph_destroy(elph_handle h);

/* Use an existing HIP stream (hipStream_t passed as void*; NULL = the handle's own stream). */
int elph_set_stream(elph_handle h, void *hip_stream);

/* Block until all work queued on the handle's stream has finished. */
int elph_synchronize(elph_handle h);

/* ---------------------------------------------------------------- update_model! */

/* update_model!(holstein) — HolsteinModels.jl:526-549:
 * expnDtauV[i,tau] = exp(-dtau*(lambda_i x + lambda2_i x^2 - mu_i)), computed on the device.
 * x: double[nsites*ltau] (reference layout); lambda, lambda2, mu: double[nsites]. */
int elph_update_model_holstein(elph_handle h, const double *x, const double *lambda,
                               const double *lambda2, const double *mu, double dtau);

/* Several independent phonon configurations (Markov chains; the reference runs them as separate processes with
 * run-IDs, ElPhDynamics.jl:90-95) resident in ONE handle: X is double[nchains*nsites*ltau], chain-major.  In a
 * subsequent batched solve / mat-vec right-hand side r uses the fermion matrix of chain r % nchains, so e.g.
 * nrhs = 2*nchains advances both pseudofermion solves of nchains HMC force evaluations together.  The KPM
 * preconditioner and the force call are per-configuration and require nchains == 1. */
int elph_update_model_holstein_chains(elph_handle h, int nchains, const double *X, const double *lambda,
                                      const double *lambda2, const double *mu, double dtau);

/* Same, but hands over an already exponentiated model.expnDtauV (double[nsites*ltau]). */
int elph_set_expV(elph_handle h, const double *expnDtauV);

/* update_model!(ssh) matrix elements — SSHModels.jl:510-535 (computed by the caller):
 * cosht, sinht: double[ltau*nbonds], Julia (Ltau x Nbonds) column-major (tau fastest),
 * already in checkerboard column order; expDtauMu: double[nsites]. */
int elph_update_model_ssh(elph_handle h, const double *cosht, const double *sinht,
                          const double *expDtauMu);

/* Several independent phonon configurations (chains) of one SSH deck in one handle (the reference runs chains as separate
 * processes, ElPhDynamics.jl:90-95): X[nchains][nph*ltau]; the other arguments as elph_update_model_ssh_fields.  In a batched
 * call right-hand side r then uses the hopping tables of chain r % nchains. */
int elph_update_model_ssh_fields_chains(elph_handle h, int nchains, const double *X, int64_t nph, const int64_t *cb_index,
                                        const double *t_ph, const double *alpha, const double *alpha2, const double *t_bare_cb,
                                        const double *mu, double dtau);

/* update_model!(ssh) computed on the device from the phonon fields — SSHModels.jl:510-562: expΔτμ = exp(Δτ μ); per field
 * t′ = t − (α x + sign(x) α₂ x²), cosht[τ, index] = cosh(Δτ t′), sinht likewise (no host cosh/sinh, no tables over PCIe).
 *   x         double[nph * ltau]  ssh.x (field = (phonon−1)·Lτ + τ)
 *   cb_index  int64[nph]          1-based checkerboard position of each phonon's bond: checkerboard_perm[phonon_to_bond[p]]
 *   t_ph, alpha, alpha2  double[nph]   ssh.t[bond of p], ssh.α, ssh.α₂
 *   t_bare_cb double[nbonds]      bare hopping of every bond in checkerboard order (bonds without a phonon keep cosh/sinh(Δτ t))
 *   mu        double[nsites]
 * elph_get_cosh_sinh: model.cosht / model.sinht as the reference stores them ((Lτ × Nbonds) column-major), for callers that
 * read those fields after a device-side update (Holstein handles: Nbonds values each). */
int elph_update_model_ssh_fields(elph_handle h, const double *x, int64_t nph, const int64_t *cb_index, const double *t_ph,
                                 const double *alpha, const double *alpha2, const double *t_bare_cb, const double *mu, double dtau);
int elph_get_cosh_sinh(elph_handle h, double *cosht, double *sinht);

/* ---------------------------------------------------------------- mul! family */

/* mulM!(y, model, v) — HolsteinModels.jl:569-626 / SSHModels.jl:581-640 */
int elph_mulM(elph_handle h, double *y, const double *v);
/* mulMᵀ!(y, model, v) — HolsteinModels.jl:631-684 / SSHModels.jl:646-701 */
int elph_mulMT(elph_handle h, double *y, const double *v);
/* mulMᵀM!(y, model, v) — Models.jl:215-224 (one fused kernel; no v' round trip) */
int elph_mulMTM(elph_handle h, double *y, const double *v);
/* mulMMᵀ!(y, model, v) — Models.jl:229-238 (the `transposed` branch of mul!, :192-209; only the non-CG solvers use it) */
int elph_mulMMT(elph_handle h, double *y, const double *v);
/* device-pointer twins (reference layout, handle's stream, asynchronous) */
int elph_mulM_dev(elph_handle h, double *y_dev, const double *v_dev);
int elph_mulMT_dev(elph_handle h, double *y_dev, const double *v_dev);
int elph_mulMTM_dev(elph_handle h, double *y_dev, const double *v_dev);

/* ---------------------------------------------------------------- solver */

/* ConjugateGradient fields — IterativeSolvers.jl:36-57 (model.solver.tol/.maxiter/.kmax). */
int elph_solver_set(elph_handle h, double tol, int64_t maxiter, double kappa_max);

/* solve!(x, model, b, cg[, P]; maxiter, tol, κmax) -> iters
 * IterativeSolvers.jl:239-314 (use_precond=0) / :153-234 (use_precond=1, KPM must be set up).
 * x is both the initial guess and the result (callers zero it: HMC.jl:854).
 * tol/maxiter/kappa_max of 0 select the handle's solver defaults (as iszero() does there).
 * eps_hist (optional, may be NULL): receives eps_0..eps_iters (needs maxiter+1 doubles). */
int elph_cg_solve(elph_handle h, double *x, const double *b, double tol, int64_t maxiter,
                  double kappa_max, int use_precond, int64_t *iters, double *eps_hist);

/* ldiv!(x, model, b, P; maxiter=0) -> (iters, residual_error, flag) — Models.jl:74-137 (use_precond=1)
 * and Models.jl:139-186 (use_precond=0): solve, true residual ‖Ax−b‖/‖b‖, flag 0/1/2 with x zeroed
 * when flag>0, and the un-preconditioned retry with 10*maxiter. */
int elph_ldiv(elph_handle h, double *x, const double *b, int use_precond, int64_t maxiter,
              int64_t *iters, double *residual_error, int *flag);

/* Batched ldiv!: nrhs independent right-hand sides of the same matrix advanced together
 * (φ₊/φ₋ of calc_O⁻¹Λϕ!, HMC.jl:851-886; the nᵥ vectors of GreensFunctions.update!, :201-234).
 * X, B: double[nrhs*ndim], RHS-major; iters/residual_error/flag: arrays of nrhs.
 * Each RHS follows exactly the single-RHS recurrences and stop rule. */
int elph_ldiv_batched(elph_handle h, int nrhs, double *X, const double *B, int use_precond,
                      int64_t maxiter, int64_t *iters, double *residual_error, int *flag);

/* device-pointer twins (reference layout) */
int elph_ldiv_dev(elph_handle h, double *x_dev, const double *b_dev, int use_precond,
                  int64_t maxiter, int64_t *iters, double *residual_error, int *flag);
int elph_ldiv_batched_dev(elph_handle h, int nrhs, double *X_dev, const double *B_dev,
                          int use_precond, int64_t maxiter, int64_t *iters,
                          double *residual_error, int *flag);

/* ---------------------------------------------------------------- fermion force */

/* One fermion-force evaluation of the Holstein model with the phonon field, both pseudofermion fields and both
 * solutions resident on the device between the steps:
 *   update_model!(model)                                   HolsteinModels.jl:526-549
 *   calc_O⁻¹Λϕ!(hmc, model, P, power)                      HMC.jl:820-915  (Λ: :921-968; two solves as one batch,
 *                                                          tolerance tol^power, iters = cld(total,2), flag)
 *   calc_dSfdx!(hmc, model)                                HMC.jl:790-814  (mulM!, muldMdx! HolsteinModels.jl:691-755,
 *                                                          muldΛdx! HMC.jl:1005-1025)
 * dSfdx is ACCUMULATED into (the reference adds the fermionic force to hmc.dSdx).  Xp_out / Xm_out (optional)
 * receive O⁻¹Λϕ₊ / O⁻¹Λϕ₋.  use_precond needs elph_kpm_setup on the *updated* field, so preconditioned callers
 * call elph_update_model_holstein + elph_kpm_setup first (setup! runs inside calc_O⁻¹Λϕ!, HMC.jl:834). */
int elph_fermion_force_holstein(elph_handle h, const double *x, const double *lambda, const double *lambda2,
                                const double *mu, double dtau, const double *phi_plus, const double *phi_minus,
                                int use_precond, double tol_power, double *dSfdx, double *Xp_out, double *Xm_out,
                                int64_t *iters, int *flag);

/* The same for the SSH model (bond phonons): the two solves of calc_O⁻¹Λϕ! on the given right-hand sides (Λ is the
 * identity for SSH, HMC.jl:943-946,970-973, so the caller passes hmc.Λϕ₊/Λϕ₋ = MᵀR±) and the bond-local part of
 * muldMdx! (SSHModels.jl:707-829) fused with mulM! (HMC.jl:797-806):
 *   q_out[n*Ltau + tau] = sum_± ( c_j b_i + c_i b_j ) for checkerboard bond n (tau fastest), so that
 *   dMdx[field(phonon(n), tau)] = sg(tau) * dtau * (alpha + 2 alpha2 x) * q_out[...],   sg(1) = -1, else +1,
 * which the caller scatters to the phonon fields (primary_field bookkeeping stays on the host) and SUBTRACTS from
 * dSdx (HMC.jl:803,808).  elph_update_model_ssh must have been called for the current field. */
int elph_fermion_force_ssh(elph_handle h, const double *rhs_plus, const double *rhs_minus, int use_precond,
                           double tol_power, double *q_out, double *Xp_out, double *Xm_out, int64_t *iters, int *flag);

/* As elph_fermion_force_ssh, with the scatter onto the phonon fields done on the device as well:
 *   dSdx[(p−1)·Lτ + τ] −= sg(τ)·Δτ·(α_p + 2 α₂_p x)·q[τ][bond(p)]      (SSHModels.jl:797-823, HMC.jl:803,808)
 * for the fields x, couplings and checkerboard positions of the last elph_update_model_ssh_fields call (one field per
 * bond and τ; the reference's primary_field aliasing of equivalent fields stays with the caller).  dSdx: double[nph·Lτ],
 * accumulated into. */
int elph_fermion_force_ssh_fields(elph_handle h, const double *rhs_plus, const double *rhs_minus, int use_precond,
                                  double tol_power, double *dSdx, double *Xp_out, double *Xm_out, int64_t *iters, int *flag);

/* muldMdx!(dMdx, u, model, v): dMdx[field] = uᵀ·(∂M/∂x_field)·v for caller-given u and v — the L3 operator that calc_dSfdx! applies
 * to (u, v) = (M·O⁻¹Λϕ±, O⁻¹Λϕ±) (HMC.jl:799,804) and LangevinDynamics.calc_dSfdx! to (g, M⁻¹g) (:378).  The matrix is the handle's
 * current one (the last update_model).
 *   Holstein (HolsteinModels.jl:691-755): dMdx[i,τ] = [CBᵀu](i,τ) · sg(τ) Δτ (λ_i + 2 λ₂_i x(i,τ)) e^{-ΔτV}(i,τ) · v(i,τ−1), sg(1) = −1 with
 *   v(i,0) = v(i,Lτ); x, u, v, dMdx: double[Ndim], reference layout; lambda, lambda2: double[nsites] (host arrays in both forms). */
int elph_muldMdx_holstein(elph_handle h, double *dMdx, const double *u, const double *v, const double *x,
                          const double *lambda, const double *lambda2, double dtau);
int elph_muldMdx_holstein_dev(elph_handle h, double *dMdx_dev, const double *u_dev, const double *v_dev, const double *x_dev,
                              const double *lambda, const double *lambda2, double dtau);
/*   SSH (SSHModels.jl:707-829), bond-local part: q_out[n·Lτ + τ] = c_j b_i + c_i b_j for checkerboard bond n (τ fastest) with
 *   b = the partial forward product of e^{Δτμ}v(τ−1), c = the partial inverse product of CBᵀu, both after bond n (:775-806), so that
 *   dMdx[primary_field[field(phonon(n), τ)]] += sg(τ) Δτ (α + 2 α₂ x) q_out[...]  (:797-823) is the caller's scatter — as for
 *   elph_fermion_force_ssh.  Works after either elph_update_model_ssh or elph_update_model_ssh_fields. */
int elph_muldMdx_ssh(elph_handle h, double *q_out, const double *u, const double *v);
/*   SSH with the scatter on the device, for the fields, couplings and bond positions of the last elph_update_model_ssh_fields:
 *   dMdx[(p−1)·Lτ + τ] = sg(τ) Δτ (α_p + 2 α₂_p x) q[τ][bond(p)], double[nph·Lτ]; the primary_field sum over equivalent fields
 *   (:820-826) stays with the caller. */
int elph_muldMdx_ssh_fields(elph_handle h, double *dMdx, const double *u, const double *v);
int elph_muldMdx_ssh_fields_dev(elph_handle h, double *dMdx_dev, const double *u_dev, const double *v_dev);

/* ---------------------------------------------------------------- HMC trajectory */

/* One HMC update of the Holstein model, update!(model, hmc, fa, P) — HMC.jl:313-337 — i.e. standard_update!
 * (:343-463, nb == 1) or multitimestep_update! (:469-638, nb > 1), with refresh_v!/refresh_ϕ! (:648-692), calc_H
 * (:697-784), calc_Sb/calc_dSbdx! (PhononAction.jl:11-66,114-187), calc_O⁻¹Λϕ! and calc_dSfdx! inside.  The phonon
 * field x, the velocity v, ϕ±, O⁻¹Λϕ± and dS/dx stay on the device for the whole trajectory and between updates.
 *
 * elph_hmc_create: per-site omega, omega4, lambda, lambda2, mu (double[nsites]), dtau, and FourierAccelerator.M
 *   (fa_mass: double[nsites*ltau], frequency index fastest — FourierAcceleration.jl:222-240).  Dispersive phonon modes
 *   are not supported (the reference's calc_Sb reads an undefined variable for them, PhononAction.jl:49).
 * elph_hmc_set_state / get_state: model.x and hmc.v (double[nsites*ltau], tau fastest; NULL = leave / skip).
 * elph_hmc_update: dt = hmc.Δt, nt = hmc.Nt, nb = hmc.Nb, alpha = partial momentum refresh.  The random numbers the
 *   reference draws from model.rng are inputs: R[Ndof] (refresh_v!), Rp, Rm [Ndim] (refresh_ϕ!), kpm_randn[(nt+2)*2*nsites]
 *   (the Arnoldi start vector pairs of the up to nt+2 setup!(P) calls, in call order; NULL without preconditioner) and
 *   u_accept (the uniform of the Metropolis test, :441).  Solver settings are elph_solver_set's (tol^2 for the two
 *   action evaluations, tol for the forces, :373,:398,:428).
 *   Outputs: accepted (0/1); iters_per_solve = cld(iters, nt+2) as the reference returns; energies[5] = H0, H1,
 *   S and K of the last calc_H, acceptance probability; flag = last linear-solve flag (> 0: trajectory killed, rejected).
 *   On rejection x is restored, v = -v0 and exp(-dtau V) is rebuilt (:447-456). */
int elph_hmc_create(elph_handle h, const double *omega, const double *omega4, const double *lambda, const double *lambda2,
                    const double *mu, double dtau, const double *fa_mass);
int elph_hmc_set_state(elph_handle h, const double *x, const double *v);
int elph_hmc_get_state(elph_handle h, double *x, double *v);
int elph_hmc_update(elph_handle h, double dt, int64_t nt, int nb, double alpha, int use_precond, const double *R,
                    const double *Rp, const double *Rm, const double *kpm_randn, double u_accept, int *accepted,
                    double *iters_per_solve, double *energies, int *flag);

/* The same trajectory for an SSH model (bond phonons; SURVEY config E): the fields are the Nph·Lτ bond-phonon displacements,
 * update_model! is the device-side cosh/sinh of t′ = t − (αx + sign(x)α₂x²) (SSHModels.jl:510-562), Λ ≡ 1
 * (HMC.jl:943-946), the force is −dMdx(M X±, X±) scattered from the bond brackets (SSHModels.jl:707-829), calc_Sb /
 * calc_dSbdx! are PhononAction.jl:68-112,189-234 with every field its own primary field.  elph_hmc_create_ssh replaces
 * elph_hmc_create (omega, omega4: double[nph]; fa_mass: double[nph·Lτ]; the other arguments as in
 * elph_update_model_ssh_fields); elph_hmc_set_state / _get_state / elph_hmc_update then work on double[nph·Lτ] fields
 * (R: [nph·Lτ]; Rp, Rm: [Ndim]). */
int elph_hmc_create_ssh(elph_handle h, int64_t nph, const double *omega, const double *omega4, const int64_t *cb_index,
                        const double *t_ph, const double *alpha, const double *alpha2, const double *t_bare_cb, const double *mu,
                        double dtau, const double *fa_mass);
/* The same for nchains independent chains of one SSH deck in lockstep (x, v, R chain-major as for elph_hmc_create_chains;
 * elph_hmc_update_chains / elph_langevin_evolve then act on all of them; special moves stay single-chain). */
int elph_hmc_create_ssh_chains(elph_handle h, int nchains, int64_t nph, const double *omega, const double *omega4,
                               const int64_t *cb_index, const double *t_ph, const double *alpha, const double *alpha2,
                               const double *t_bare_cb, const double *mu, double dtau, const double *fa_mass);

/* Several Markov chains (same deck, independent phonon fields — the reference runs them as separate processes,
 * ElPhDynamics.jl:90-95) advanced in LOCKSTEP by one handle: the leapfrog schedule is common, every chain has its own
 * x, v, ϕ±, energies, Metropolis test and failure flag, and all 2·nchains pseudofermion solves of a force / action
 * evaluation run as ONE batched CG (with use_precond: one KPM expansion per chain, elph_kpm_setup_chains).
 * elph_hmc_create_chains replaces elph_hmc_create; elph_hmc_set_state / _get_state then move double[nchains * ndim]
 * (chain-major).  elph_hmc_update_chains: R [nchains*Ndof], Rp, Rm [nchains*Ndim] chain-major;
 * kpm_randn [(nt+2)][2 (b_max, b_min)][nchains][nsites]; u_accept, accepted, iters_per_solve, flag [nchains];
 * energies [nchains][5].  A chain whose linear solve fails is rejected (flag > 0) without stopping the others. */
int elph_hmc_create_chains(elph_handle h, int nchains, const double *omega, const double *omega4, const double *lambda,
                           const double *lambda2, const double *mu, double dtau, const double *fa_mass);
int elph_hmc_update_chains(elph_handle h, double dt, int64_t nt, int nb, double alpha, int use_precond, const double *R,
                           const double *Rp, const double *Rm, const double *kpm_randn, const double *u_accept, int *accepted,
                           double *iters_per_solve, double *energies, int *flag);

/* model.μ changed while the field lives on the device (the chemical-potential tuner, MuFinder.jl:68-107 adds Δμ to every site):
 * new μ[nsites] for the HMC / Langevin state, followed by update_model!. */
int elph_hmc_set_mu(elph_handle h, const double *mu);
/* chains in lockstep, each with its own tuner: mu[nchains][nsites] (elph_hmc_set_mu / a new elph_hmc_create* return to one μ) */
int elph_hmc_set_mu_chains(elph_handle h, const double *mu);

/* SSH phonon types of the same name (the default "" included) share their fields: primary_field of initialize_model!
 * (SSHModels.jl:480-502).  primary_column[Nph]: 0-based column of the primary phonon of every phonon (itself for a primary; the
 * sharing is the same on every time slice).  Call after elph_hmc_create_ssh / elph_langevin_create_ssh.  Then, as in the
 * reference: the fermion force of a class is summed over its members and given to each (muldMdx!, :820-826), calc_Sb and
 * calc_K count primary fields only (PhononAction.jl:83, HMC.jl:720-738), elph_hmc_set_state refuses fields that differ
 * from their primary (update_model!, :549-559), generated random field vectors are copied from their primaries (randn!,
 * :568-575; host-supplied R / eta must already be v[primary_field]); elph_hmc_special_move returns ELPH_E_UNSUPPORTED. */
int elph_hmc_set_shared_fields(elph_handle h, const int64_t *primary_column);

/* Optional on-device generator for the random inputs of elph_hmc_update[_chains], elph_hmc_special_move and
 * elph_langevin_evolve (the reference draws them from model.rng on the CPU: randn! HMC.jl:655,675-676, LangevinDynamics.jl:97,360,
 * KPMPreconditioners.jl:860,903; with 64 chains the host generator and the PCIe transfer of 3 x nchains vectors per update cost
 * a quarter of the update).  After elph_hmc_set_rng(h, seed) any of R, Rp, Rm, eta, g1, g2, kpm_randn, u_accept may be NULL
 * (u_accept < 0 where it is passed by value): each missing input is one batch of the build's counter-based generator
 * (SplitMix64 -> uniform(0,1) -> Box-Muller, elphdynamics_amd/synth.py), batch b (1-based, counted per handle since
 * elph_hmc_set_rng) seeded with output b of SplitMix64(seed), numbered in the layout the host array would have had — so a run
 * is reproducible from (seed, batch count) and equals the run that is handed synth.randn(batch seed, n) explicitly.  Field and
 * site vectors are generated on the device; Arnoldi start vectors and Metropolis uniforms (small) on the host.  Order of the
 * batches within a call: HMC update R, Rp, Rm, kpm_randn, u_accept; special move Rp, Rm, kpm_randn, u_accept; Langevin step eta,
 * kpm_randn, g1, g2. */
int elph_hmc_set_rng(elph_handle h, uint64_t seed);
int elph_hmc_rng_batches(elph_handle h, uint64_t *batches);

/* One proposed move of special_update! (SpecialUpdates.jl:103-136 ReflectionUpdate, :205-275 SwapUpdate) on the device-resident
 * field of the (single-chain) HMC state:  S₀ = refresh_ϕ!(…, sample_R = true) with the fresh R± = Rp, Rm (HMC.jl:665-692);
 * kind 0: x_col(τ) → −x_col(τ) on phonon column col_i; kind 1: columns col_i, col_j exchange their world lines (0-based; sites
 * for Holstein, bond phonons for SSH); update_model!, calc_O⁻¹Λϕ!(…, 2.0), S₁ = calc_S; accepted iff u_accept <
 * min(1, e^{−(S₁−S₀)}) and flag == 0, else the move is undone.  kpm_randn: [2][nsites] or NULL.  Which sites / bonds are tried
 * (sample!(model.rng, …)) is the caller's bookkeeping. */
int elph_hmc_special_move(elph_handle h, int kind, int64_t col_i, int64_t col_j, const double *Rp, const double *Rm, int use_precond,
                          const double *kpm_randn, double u_accept, int *accepted, double *S0, double *S1, int64_t *iters, int *flag);
/* The same for chains in lockstep: every chain proposes its own move (col_i[c], col_j[c]) on its own field, one batched action
 * evaluation serves all of them, acceptance is per chain (all arrays [nchains]; Rp, Rm chain-major like elph_hmc_update_chains;
 * kpm_randn [2][nchains][nsites]). */
int elph_hmc_special_move_chains(elph_handle h, int kind, const int64_t *col_i, const int64_t *col_j, const double *Rp,
                                 const double *Rm, int use_precond, const double *kpm_randn, const double *u_accept,
                                 int *accepted, double *S0, double *S1, int64_t *iters, int *flag);

/* ---------------------------------------------------------------- Langevin dynamics (caller of the path) */

/* LangevinDynamics.jl on a Holstein handle.  elph_langevin_create: the arguments of elph_hmc_create with
 * fa_Q = FourierAccelerator.Q (evolve! calls fourier_accelerate! without use_mass); the field moves through
 * elph_hmc_set_state / elph_hmc_get_state (v = NULL).
 * elph_langevin_evolve = evolve!(model, dyn, fa, P): scheme 0 EulerDynamics (:81-130), 1 RungeKuttaDynamics (:162-232),
 * 2 HeunsDynamics (:272-328), with calc_dSdx! (:334-384: one solve MᵀM x = Mᵀg, dSf/dx = −2 gᵀ(∂M/∂x)M⁻¹g via muldMdx!
 * HolsteinModels.jl:691-755, plus the shifted calc_dSbdx!) on the device.  The random numbers are inputs: eta [Ndof], g1, g2
 * [Ndim] (g2 unused by Euler), kpm_randn [2][2][nsites] (b_max, b_min per setup!(P)).  iters: what evolve! returns; flag:
 * the last ldiv! flag (the reference ignores it). */
int elph_langevin_create(elph_handle h, const double *omega, const double *omega4, const double *lambda, const double *lambda2,
                         const double *mu, double dtau, const double *fa_Q);
/* SSH handles (examples/ssh_langevin_square.toml): the arguments of elph_hmc_create_ssh with fa_Q per phonon; eta has
 * nph·Lτ entries, calc_dSfdx! uses muldMdx! of SSHModels.jl:707-829 with u = g, and the shifted flag changes nothing. */
int elph_langevin_create_ssh(elph_handle h, int64_t nph, const double *omega, const double *omega4, const int64_t *cb_index,
                             const double *t_ph, const double *alpha, const double *alpha2, const double *t_bare_cb, const double *mu,
                             double dtau, const double *fa_Q);
/* nchains independent Langevin trajectories of one (Holstein) deck in lockstep: every step is one batched solve of nchains
 * right-hand sides with one KPM expansion per chain.  State and random vectors are chain-major (x, eta: [nchains·Ndof];
 * g1, g2: [nchains·Ndim]; kpm_randn: [2 set-ups][b_max | b_min][nchains][nsites]); elph_langevin_evolve then returns
 * iters and flag per chain ([nchains]). */
int elph_langevin_create_chains(elph_handle h, int nchains, const double *omega, const double *omega4, const double *lambda,
                                const double *lambda2, const double *mu, double dtau, const double *fa_Q);
int elph_langevin_evolve(elph_handle h, int scheme, double dt, int use_precond, const double *eta, const double *g1, const double *g2,
                         const double *kpm_randn, int64_t *iters, int *flag);

/* ---------------------------------------------------------------- Green's-function estimator */

/* EstimateGreensFunction(model, n_v) — GreensFunctions.jl:155-195.  norbits*L1*L2*L3 must equal nsites
 * (site = norbits*cell + orbit, cell = l1 + L1*(l2 + L2*l3), Lattices.jl:56-107).  n_v = max(2, nv) (:167). */
int elph_greens_create(elph_handle h, int norbits, int L1, int L2, int L3, int nv);
int elph_greens_nv(elph_handle h, int *nv);

/* update!(estimator, model, P) — GreensFunctions.jl:201-234: for every noise vector r (R: double[n_v * ndim], the
 * vectors the reference draws with randn!(model.rng, r₁), :212) solve MᵀM x = Mᵀ r from x = 0 with ldiv!'s semantics
 * (Models.jl:74-186; flags, zero-fill, un-preconditioned retry) — the n_v solves run as ONE batched CG.  R and M⁻¹R
 * stay on the device.  iters / residual_error / flag: per vector, each may be NULL.  With use_precond the caller has
 * run elph_kpm_setup on the current field (setup!(preconditioner), :206).  With several chains resident (elph_update_model_*_chains /
 * elph_hmc_create_*chains) row r of R belongs to chain r % nchains: create the estimator with n_v * nchains vectors and find
 * vector v of chain c at row v * nchains + c (elph_greens_setup then takes those 1-based indices). */
int elph_greens_update(elph_handle h, const double *R, int use_precond, int64_t *iters, double *residual_error, int *flag);

/* estimator.R / estimator.M⁻¹R (double[n_v * ndim], vector-major, tau fastest); NULL = skip.  set: replay of vectors
 * produced elsewhere (both must have been given before elph_greens_setup). */
int elph_greens_set_vectors(elph_handle h, const double *R, const double *MinvR);
int elph_greens_get_vectors(elph_handle h, double *R, double *MinvR);

/* setup!(estimator, n₁, n₂) — GreensFunctions.jl:239-288 with convolve! (:351-400), antiperiodic_copy! (:406-418),
 * periodic_product! (:424-440); n₁, n₂ 1-based.  Outputs (NULL = not copied to the host), each
 * Complex{Float64}[2L, n_s, n_s, L1, L2, L3] as interleaved (re, im) doubles, first index fastest:
 *   GD0      G[Δ,0]            measure_GΔ0      (:293-298)  = array[mod1(τ+1,2L), o₂, o₁, l₁+1, l₂+1, l₃+1]
 *   GD0_GD0  G[Δ,0]·G[Δ,0]     measure_GΔ0_GΔ0  (:303-308)
 *   GDD_G00  G[Δ,Δ]·G[0,0]     measure_GΔΔ_G00  (:313-318)
 *   GD0_G0D  G[Δ,0]·G[0,Δ]     measure_GΔ0_G0Δ  (:324-329)
 * Imaginary parts are exact zeros (the reference's are FFT round-off, ~1e-17). */
int elph_greens_setup(elph_handle h, int n1, int n2, double *GD0, double *GD0_GD0, double *GDD_G00, double *GD0_G0D);

/* Device-resident results of the last elph_greens_setup: arrays[0..3] in the order above, `count` complex numbers each
 * (valid until the next elph_greens_setup / elph_greens_create / elph_destroy). */
int elph_greens_dev_arrays(elph_handle h, void **arrays, int64_t *count);

/* ---------------------------------------------------------------- KPM preconditioner */

/* SymmetricKPMPreconditioner(model, n, buf, c1, c2) — KPMPreconditioners.jl:219-235, ctor :101-146 */
int elph_kpm_create(elph_handle h, int n, double buf, double c1, double c2);

/* KPMPreconditioners.setup!(P) — :259-321.  Averages expnDtauV over tau on the device
 * (update_A!, :332-381), estimates the spectrum of A by Arnoldi (:845-942) and rebuilds the
 * Chebyshev coefficients when the bounds moved by more than buf (:293-309).
 *   b_max, b_min  the two random Arnoldi start vectors (double[nsites]); the reference draws
 *                 them from model.rng (:859-861,902-904) — the caller supplies them.
 *   e_min, e_max  if both finite, skip Arnoldi and use these eigenvalue bounds (parity runs);
 *                 pass NaN to run Arnoldi.
 * Out: active flag (P.expansion.active), and the bounds used. */
int elph_kpm_setup(elph_handle h, const double *b_max, const double *b_min, double e_min,
                   double e_max, int *active, double *lam_lo, double *lam_hi);

/* The same for every phonon configuration resident after elph_update_model_holstein_chains (nchains of them): one
 * expansion per chain — its own Ē, eigenvalue bounds, orders and coefficients; in a batched solve right-hand side r is
 * preconditioned with the expansion of chain r % nchains.  b_max, b_min: double[nchains * nsites] (chain-major);
 * e_min, e_max: double[nchains] or NULL (NULL or non-finite entries: run Arnoldi for that chain); active, lam_lo,
 * lam_hi: per chain, each may be NULL.  A chain whose bounds fail the test of :280 is preconditioned with the identity. */
int elph_kpm_setup_chains(elph_handle h, const double *b_max, const double *b_min, const double *e_min,
                          const double *e_max, int *active, double *lam_lo, double *lam_hi);

/* Inspect the expansion (of chain 0): orders[cld(ltau,2)], total = sum(orders) (either pointer may be NULL). */
int elph_kpm_orders(elph_handle h, int64_t *orders, int64_t *total);

/* ldiv!(z, P, r) — KPMPreconditioners.jl:426-481 (identity copy when inactive, :475-478) */
int elph_kpm_apply(elph_handle h, double *z, const double *r);
int elph_kpm_apply_dev(elph_handle h, double *z_dev, const double *r_dev);

/* ---------------------------------------------------------------- Fourier acceleration */

/* fourier_accelerate!(v', fa, v, power; use_mass) — FourierAcceleration.jl:91-143, real in/out.
 * diag: the accelerator's M (use_mass=true) or Q (false) vector, double[nph*ltau],
 * frequency index fastest (update_M!/update_Q!, :149-167,176-266); nph = number of phonon
 * columns (nsites for Holstein, Nph for SSH). */
int elph_fourier_accelerate(elph_handle h, double *vout, const double *vin, const double *diag,
                            double power, int64_t nph);

/* τ_to_ω! / ω_to_τ! — TimeFreqFFTs.jl:55-73,112-130 (twisted FFT; complex interleaved re,im) */
int elph_tau_to_omega(elph_handle h, double *nu_complex, const double *v);
int elph_omega_to_tau(elph_handle h, double *v, const double *nu_complex);

/* ---------------------------------------------------------------- one solve over several GPUs
 *
 * Slabs of rows of cells along the slowest spatial index, one process per GPU, no reference equivalent (the reference is
 * single-process): solve!(x, A, b, cg) of IterativeSolvers.jl:239-314 with x0 = 0 for ONE fermion matrix whose lattice is
 * spread over `world` ranks.  The handle lives on the rank's SLAB lattice — [ghost rows from below | own rows | ghost rows
 * from above], site-contiguous, created with the bonds / exp(-dtau V) / (SSH) per-(tau, bond) tables of that slab; the ghost
 * rows are the dependency closure of the fused MtM on the own rows (elphdynamics_amd/sharded.py computes it from the bond
 * table).  Per CG iteration the ranks exchange the two partial inner products and ONE set of checkerboard boundary rows of
 * the residual, by device-initiated stores into each other's mailbox (hipIpc-mapped device memory: xGMI peer stores between
 * GPUs) from inside the resident CG kernel — no collective, no host.  Sequence on every rank:
 *   elph_shard_create   geometry (0-based local site ranges) -> a 64-byte IPC handle of this rank's mailbox
 *   (caller: all-gather the handles — torch.distributed / MPI)
 *   elph_shard_connect  map the other ranks' mailboxes
 *   per solve: elph_shard_prepare (zero the own mailbox) ; caller's BARRIER ; elph_shard_solve
 * own_lo / own_n: own sites [own_lo, own_lo + own_n) of the slab; n_to_prev / n_to_next: how many own sites (counted from
 * the bottom / from the top) the previous / next rank of the periodic ring holds as ghosts; cap_ghost: capacity in sites of
 * a ghost region of the mailbox, the SAME number on all ranks (>= every rank's ghost and send counts).
 * x_slab / b_slab: host vectors on the slab lattice, reference layout (site-major), b with its ghost entries filled from
 * the global right-hand side; x_slab's own rows are this rank's part of the solution.
 * iters / done / eps: identical on all ranks (done: 1 eps < tol, 2 kappa bound, 3 maxiter).
 * n_global / own_global_start / global_sites (0-based global site of every slab site; n_global = 0 and NULL when only
 * un-preconditioned solves are wanted): the geometry the preconditioned solve needs.
 * elph_shard_solve_kpm — solve!(x, A, b, cg, P) (IterativeSolvers.jl:153-234) with the KPM preconditioner under sharding:
 * hfull is a second handle on the WHOLE lattice (same bonds, exp(-dtau V) of all sites, elph_kpm_create + elph_kpm_setup done
 * with the same inputs on every rank).  Per iteration the ranks trade three partial inner products and all-gather the tau-spectrum
 * of the residual (peer stores into the mailboxes); every rank runs the per-frequency Chebyshev recursion on the whole lattice and
 * transforms back its own and ghost rows — no omega-sharded all-to-all: the recursion's critical path is its longest frequency
 * block either way (KPMPreconditioners.jl:426-481).
 * Ranks that live in ONE process (a host thread or task per GPU) use the same calls: a handle in the all-gathered list that this
 * process created itself is mapped by its device pointer (peer access enabled between devices) instead of through hipIpc.
 * elph_shard_shape — host arithmetic, no device: the team shape (waves per workgroup, workgroups per rank) the sharded kernel
 * takes for a time axis and the number of records world ranks put into a meeting against the capacity; ELPH_E_UNSUPPORTED
 * when they do not fit (every BASELINE config fits at 1, 2, 4 and 8 ranks: 8 x 20 = 160 of 256 at Ltau = 160). */
#define ELPH_SHARD_IPC_BYTES 64
int elph_shard_shape(int64_t ltau, int world, int *waves, int *groups, int *records, int *max_records);
int elph_shard_create(elph_handle h, int rank, int world, int64_t own_lo, int64_t own_n, int64_t n_to_prev,
                      int64_t n_to_next, int64_t cap_ghost, int64_t n_global, int64_t own_global_start,
                      const int64_t *global_sites, void *ipc_handle_out);
int elph_shard_connect(elph_handle h, const void *all_ipc_handles /* world * ELPH_SHARD_IPC_BYTES, rank order */);
int elph_shard_prepare(elph_handle h);
/* Preflight of the mailbox protocol, a collective like a solve (elph_shard_prepare on every rank, the caller's barrier, then this on
 * every rank): `rounds` lock-step exchanges of one tagged granule between ALL pairs of ranks through the mapped mailboxes — the stores
 * and polls a solve uses (xGMI peer stores between GPUs).  us_per_round[world] (may be NULL): mean time from this rank's store to the
 * sight of rank q's granule; *slowest_us: their maximum.  A rank that does not answer within ELPH_SHARD_SELFTEST_MS (10 s) gives
 * ELPH_E_HIP with the silent ranks named — at set-up instead of a time-out inside the first solve. */
int elph_shard_selftest(elph_handle h, int rounds, double *us_per_round, double *slowest_us);
/* *can = 1 when device dev_a can map memory of dev_b (hipDeviceCanAccessPeer; 1 on the diagonal): what a solve sharded over the
 * GPUs of a node needs between ring neighbours (boundary rows) and all pairs (rank records). */
int elph_peer_access(int dev_a, int dev_b, int *can);
int elph_shard_solve(elph_handle h, double *x_slab, const double *b_slab, double tol, int64_t maxiter, double kappa_max,
                     int64_t *iters, int *done, double *eps);
int elph_shard_solve_kpm(elph_handle h, elph_handle hfull, double *x_slab, const double *b_slab, double tol, int64_t maxiter,
                         double kappa_max, int64_t *iters, int *done, double *eps);
int elph_shard_destroy(elph_handle h);

/* The CALLERS of the solve on a sharded lattice (BASELINE configs "HMC ... spatial-sharded across 8 GPUs"): ldiv!'s wrapper, the fermion
 * force and — elph_hmc_update on a sharded handle — one HMC update.  Everything but the solve is pointwise in the site index or stays
 * inside the MᵀM closure the slab already holds; what crosses ranks besides the solve are a few scalars (true residual, energies) and,
 * once per force evaluation, the ghost rows of one vector (device to device through the mailboxes, see elph_shard_ghost_stats).  The
 * scalars and the barriers go through two host collectives the caller registers once per handle
 * (MPI.Barrier / MPI.Allreduce!(SUM), torch.distributed, a thread barrier for ranks that share a process); the CG iteration itself
 * keeps its device-initiated mailbox stores.  The library arms the mailbox itself (elph_shard_prepare + the barrier) before every solve
 * of these calls.  All vectors are SLAB vectors (own + ghost rows, reference layout, ghost entries filled from the global arrays);
 * results are valid on the OWN rows; scalars (iters, residual_error, flag, energies, accepted) come out identical on every rank.
 *   barrier(ctx) -> 0 on success;  allreduce_sum(ctx, buf, n): in-place sum over the ranks of n doubles, the SAME result on every rank. */
typedef int (*elph_shard_barrier_fn)(void *ctx);
typedef int (*elph_shard_allreduce_fn)(void *ctx, double *buf, int n);
int elph_shard_set_collectives(elph_handle h, elph_shard_barrier_fn barrier, elph_shard_allreduce_fn allreduce_sum, void *ctx);
/* ldiv!(x, model, b[, P]; maxiter) -> (iters, residual_error, flag) — Models.jl:74-137,139-186 — on a sharded lattice: the solve, the true
 * residual |MᵀM x − b| / |b| over all ranks' own rows, flag 1 / 2 with zero-fill, and with use_precond (hfull: the full-lattice handle
 * with elph_kpm_setup done, as for elph_shard_solve_kpm) the un-preconditioned retry with 10·maxiter.  Solver settings: elph_solver_set
 * on h. */
int elph_shard_ldiv(elph_handle h, elph_handle hfull, double *x_slab, const double *b_slab, int use_precond, int64_t maxiter,
                    int64_t *iters, double *residual_error, int *flag);
/* elph_fermion_force_holstein on a sharded lattice: update_model!, Λϕ±, the two solves (one after the other: the sharded kernel carries
 * one right-hand side; a failed first solve suppresses the second, iters = cld(total, 2): HMC.jl:851-909) and calc_dSfdx! on the slab;
 * dSfdx is accumulated into on the OWN rows only. */
int elph_shard_fermion_force_holstein(elph_handle h, elph_handle hfull, const double *x, const double *lambda, const double *lambda2,
                                      const double *mu, double dtau, const double *phi_plus, const double *phi_minus, int use_precond,
                                      double tol_power, double *dSfdx, double *Xp_out, double *Xm_out, int64_t *iters, int *flag);
/* elph_fermion_force_ssh on a sharded lattice: q_out[n·Ltau + tau] for the slab's bonds in their local checkerboard order; a bracket is
 * exact on the rank that owns the bond — the caller keeps those and scatters them onto the phonon fields. */
int elph_shard_fermion_force_ssh(elph_handle h, elph_handle hfull, const double *rhs_plus, const double *rhs_minus, int use_precond,
                                 double tol_power, double *q_out, double *Xp_out, double *Xm_out, int64_t *iters, int *flag);

/* elph_hmc_update on a SHARDED handle (elph_hmc_create / elph_hmc_create_ssh + elph_hmc_set_state on the slab handle, collectives set): one HMC
 * update of the whole lattice over the ranks — one chain; R, Rp, Rm and every per-site / per-phonon array are the slab's
 * part of the global ones (ghost entries included), u_accept the same number on every rank.  Bond phonons additionally need the slab's phonon
 * columns in the numbering of the whole lattice and their owners (the phonons of the slab's bonds; owner = the rank whose own rows hold the
 * bond's first site): global_column[nph_slab], own_weight[nph_slab] in {0, 1}; the state vectors of elph_hmc_set_state / _get_state are then
 * double[nph_slab * Ltau] in that column order. */
int elph_shard_hmc_set_columns(elph_handle h, const int64_t *global_column, int64_t n_global_columns, const double *own_weight);
/* A KPM-preconditioned update on a sharded Holstein handle (use_precond = 1): hfull = a handle on the WHOLE lattice with elph_kpm_create done
 * (it needs no field: every setup!(P) inside the update injects the τ-averaged exp(−ΔτV) of the whole lattice, summed over the ranks' own
 * rows); kpm_randn of elph_hmc_update then holds the (nt + 2) pairs of Arnoldi start vectors of the WHOLE lattice, the same on every rank. */
int elph_shard_set_full_lattice(elph_handle h, elph_handle hfull);
/* ... and on a sharded BOND-PHONON handle: hfull = a bond-phonon handle on the whole lattice with one elph_update_model_ssh (for exp(Δτμ), which does
 * not move) and elph_kpm_create done; every setup!(P) inside the update injects the τ-means of cosh / sinh of every bond of the lattice
 * (update_A!, KPMPreconditioners.jl:355-381), taken by each rank from its slab's tables for the bonds it owns and summed over the ranks.  For that
 * the slab's bonds in the numbering of the whole lattice: global_bond[nbonds_slab] = checkerboard position on the whole lattice (0-based),
 * own_weight[nbonds_slab] in {0, 1} — owner = the rank whose own rows hold the bond's first site, as for elph_shard_hmc_set_columns. */
int elph_shard_set_bonds(elph_handle h, const int64_t *global_bond, int64_t n_global_bonds, const double *own_weight);
/* The ghost rows of ϕ± (once per update) and of the fermion force (once per force evaluation) travel from their owners' devices into the
 * holders' mailboxes by peer stores — the spectrum area of the mailbox is the exchange area between solves; two of the caller's barriers per
 * exchange, no host copy of a vector.  ELPH_SHARD_GHOST_HOST=1 (A/B), or a vector that does not fit the area, stages through the caller's
 * all-reduce instead.  Counts of both since elph_shard_create: */
int elph_shard_ghost_stats(elph_handle h, int64_t *through_mailboxes, int64_t *through_host);

/* ---------------------------------------------------------------- health of the resident kernels */

/* The whole-solve-in-one-launch kernels (cg_wg.hip) wait for their team members with a wall-clock bound (ELPH_WG_TIMEOUT_MS, 2 s); a
 * launch that gives up — the GPU's CUs were held by other work — is solved again by the streaming iteration (same result to the
 * solver tolerance; ldiv!'s flags unchanged) and the handle stays on the streaming iteration for ELPH_WG_COOLDOWN solves (16) before
 * it tries again.  *cooling_down: solves left in that state (0 = the resident kernel is in use); *fallbacks: launches given up so far. */
int elph_wg_status(elph_handle h, int *cooling_down, int64_t *fallbacks);

#ifdef __cplusplus
}
#endif
#endif
