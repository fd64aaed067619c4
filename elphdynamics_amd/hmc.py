"""Immediate callers of the solve, mirrored on top of the GPU operator API.

    update_Lambda_(Lam, model)                     HMC.jl:921-941
    mulLambda_(out, v, Lam, model)                 HMC.jl:951-968
    mulLambdaInv_(out, v, Lam, model)              HMC.jl:978-995
    calc_OinvLambda_phi(model, phi_p, phi_m, ...)  HMC.jl:820-915   (the 2 CG solves of every force evaluation)
    GreensEstimator.update_(model, P, rng)         GreensFunctions.jl:201-234
    GreensEstimator.estimate(i, j, tau2, tau1)     GreensFunctions.jl:334-346
    HybridMonteCarlo, update_(model, hmc, fa, P)   HMC.jl:20-337 (struct + update!): the whole trajectory on the device

The two pseudofermion solves (and the n_v measurement solves) share one fermion matrix, so they go to
the GPU as ONE batched ldiv!; each right-hand side still follows the single-RHS recurrences and stop rule — the same
algorithm as the reference's sequential solves.  Bits: a right-hand side's result does not depend on its companions in the batch
as long as the batch keeps one kernel shape (un-preconditioned: fewer than 48 right-hand sides on the 16 x 16 square lattice, any
number elsewhere); larger batches sum the inner products in another tree (same values to ~1e-15 per iteration, the stop
iteration can move by one).
update_Lambda_/mulLambda_/mulLambdaInv_ are host utilities for building test inputs; inside calc_OinvLambda_phi and
calc_dSfdx_ the Λ operations run on the device (k_lambda_rhs, k_force_holstein).
"""
import numpy as np

from . import models
from . import preconditioners as pc


def update_Lambda_(Lam, model):
    """Lambda[i,tau] = exp(-dtau (lambda_i x + lambda2_i x^2)/2); identity for SSH (HMC.jl:921-946)."""
    if model.kind == models.SSH:
        Lam[:] = 1.0
        return
    X = model.x.reshape(model.Nph, model.Ltau)
    Lam[:] = np.exp(-model.dtau * (model.lam[:, None] * X + model.lam2[:, None] * X ** 2) / 2).reshape(-1)


def mulLambda_(out, v, Lam, model):
    """(Lambda v)[tau] = -Lambda[tau+1] v[tau+1], (Lambda v)[L] = Lambda[1] v[1]; no-op copy for SSH (HMC.jl:951-973)."""
    if model.kind == models.SSH:
        return
    N, L = model.Nsites, model.Ltau
    u, La, o = v.reshape(N, L), Lam.reshape(N, L), out.reshape(N, L)
    u1 = u[:, 0].copy()
    o[:, :L - 1] = -La[:, 1:] * u[:, 1:]
    o[:, L - 1] = La[:, 0] * u1


def mulLambdaInv_(out, v, Lam, model):
    """HMC.jl:978-995."""
    if model.kind == models.SSH:
        return
    N, L = model.Nsites, model.Ltau
    u, La, o = v.reshape(N, L), Lam.reshape(N, L), out.reshape(N, L)
    uL = u[:, L - 1].copy()
    o[:, 1:] = -(1.0 / La[:, 1:]) * u[:, :L - 1]
    o[:, 0] = (1.0 / La[:, 0]) * uL


def calc_OinvLambda_phi(model, phi_p, phi_m, P=None, power=1.0, rng=None, setup_kwargs=None):
    """calc_O⁻¹Λϕ!(hmc, model, P, power) -> (O⁻¹Λϕ₊, O⁻¹Λϕ₋, iters, flag)   (HMC.jl:820-915).

    tol is raised to `power` for the duration of the call (:827-828, restored :912), setup!(P) runs first (:834),
    the reported iteration count is cld(total, 2) when both solves converged (:907-909), and a failed first
    solve suppresses the second (:880): its output stays zero."""
    if model.kind == models.HOLSTEIN:
        # Λ (update_Λ!, mulΛ!, :921-968), both solves and the bookkeeping below run inside one device call; the force it
        # also assembles is discarded here
        pc.setup_(P, rng=rng, **(setup_kwargs or {}))
        it, fl, Xp, Xm = calc_dSfdx_(np.zeros(model.Ndof), model, phi_p, phi_m, P=P, power=power, return_solutions=True)
        return Xp, Xm, it, fl
    tol = model.solver.tol
    model.solver.tol = tol ** power
    try:
        pc.setup_(P, rng=rng, **(setup_kwargs or {}))
        B = np.empty((2, model.Ndim))
        B[0], B[1] = phi_p, phi_m            # mulΛ! is a no-op for SSH: Λϕ buffers hold what the caller put there
        X = np.zeros((2, model.Ndim))        # fill!(O⁻¹Λϕ, 0)  (:854, :883)
        it, res, fl = models.ldiv_batched_(X, model, B, P=P)
        flag = int(fl[0])
        iters = int(it[0])
        if flag == 0:
            iters += int(it[1])
            flag = int(fl[1])
        else:
            X[1] = 0.0
        if flag == 0:
            iters = -(-iters // 2)           # cld(iters, 2)
        return X[0], X[1], iters, flag
    finally:
        model.solver.tol = tol


class GreensEstimator:
    """EstimateGreensFunction reduced to the solve it drives (GreensFunctions.jl:23-196 state, :201-234 update!)."""

    def __init__(self, model, nv=10):
        self.model, self.nv, self.L = model, int(nv), model.Ltau
        self.R = np.zeros((self.nv, model.Ndim))
        self.MinvR = np.zeros((self.nv, model.Ndim))

    def update_(self, P=None, rng=None, R=None, setup_kwargs=None):
        """n_v random vectors r, solve MtM x = Mt r  =>  x = M^-1 r, as one batch (GreensFunctions.jl:208-231)."""
        m = self.model
        pc.setup_(P, rng=rng, **(setup_kwargs or {}))
        if R is None:
            rng = rng or np.random.default_rng()
            R = rng.standard_normal((self.nv, m.Ndim))
        self.R[:] = R
        B = np.empty_like(self.R)
        for i in range(self.nv):
            models.mulMt_(B[i], m, np.ascontiguousarray(self.R[i]))      # Mᵀr₁ (model.v″, :223-224)
        self.MinvR[:] = 0.0
        return models.ldiv_batched_(self.MinvR, m, B, P=P)

    def estimate(self, i, j, tau2, tau1, n=0):
        """G_ij(tau2, tau1) ~ (M^-1 r)[idx(tau2,i)] * r[idx(tau1,j)]  (1-based i, j, tau as in :334-346)."""
        mm = (j - 1) * self.L + (tau1 - 1)
        nn = (i - 1) * self.L + (tau2 - 1)
        return self.MinvR[n, nn] * self.R[n, mm]


def calc_dSfdx_(dSdx, model, phi_p, phi_m, P=None, power=1.0, return_solutions=False):
    """Fermion force with everything resident on the GPU between the steps:
    update_model! + calc_O⁻¹Λϕ!(…, P, power) + calc_dSfdx! (HMC.jl:790-915) in one C-ABI call.
    dSdx is accumulated into, as the reference does.  Returns (iters, flag[, O⁻¹Λϕ₊, O⁻¹Λϕ₋]).
    SSH models: phi_p / phi_m are the right-hand sides Λϕ± (Λ = identity) and dSdx has Ndof = Nph*Ltau entries.
    For a preconditioned force call update_model_ + setup_(P) on the current field first (HMC.jl:834)."""
    import ctypes as C
    from ._lib import check, dptr
    model._push_solver()
    it, fl = C.c_int64(), C.c_int()
    Xp = np.empty(model.Ndim) if return_solutions else None
    Xm = np.empty(model.Ndim) if return_solutions else None
    if model.kind == models.HOLSTEIN:
        check(model._lib.elph_fermion_force_holstein(
            model._h, dptr(np.ascontiguousarray(model.x)), dptr(model.lam), dptr(model.lam2), dptr(model.mu), model.dtau,
            dptr(np.ascontiguousarray(phi_p)), dptr(np.ascontiguousarray(phi_m)), 0 if P is None else 1, float(power),
            dptr(dSdx), dptr(Xp) if return_solutions else None, dptr(Xm) if return_solutions else None,
            C.byref(it), C.byref(fl)))
    else:
        # SSH: Λ is the identity, phi_p / phi_m are the right-hand sides hmc.Λϕ± = MᵀR± (HMC.jl:680-686,943-973);
        # update_model!, the two solves, the bond brackets and their scatter onto the phonon fields all run on the device
        # (SSHModels.jl:797-823; equivalent fields / primary_field are not modelled here).
        models.update_model_(model)
        check(model._lib.elph_fermion_force_ssh_fields(
            model._h, dptr(np.ascontiguousarray(phi_p)), dptr(np.ascontiguousarray(phi_m)), 0 if P is None else 1,
            float(power), dptr(dSdx), dptr(Xp) if return_solutions else None, dptr(Xm) if return_solutions else None,
            C.byref(it), C.byref(fl)))
    if return_solutions:
        return int(it.value), int(fl.value), Xp, Xm
    return int(it.value), int(fl.value)


def set_shared_fields_(model):
    """Tell the device which bond phonons share their fields (model.primary_field, SSHModels.jl:480-502)."""
    from ._lib import check, iptr
    if getattr(model, "has_shared_fields", False):
        L = model.Ltau
        prim_col = np.ascontiguousarray(model.primary_field[::L] // L, dtype=np.int64)
        check(model._lib.elph_hmc_set_shared_fields(model._h, iptr(prim_col)))


class HybridMonteCarlo:
    """HybridMonteCarlo(model, Δt, tr, α, Nb) (HMC.jl:20-245), reduced to what the device-resident trajectory needs.
    x, v, ϕ±, O⁻¹Λϕ± and dS/dx live on the GPU between updates; `model.x` / `hmc.v` on the host are refreshed by
    `pull_()` (and pushed by `push_()` after the host changed them)."""

    def __init__(self, model, fa, dt, tr, alpha=0.0, Nb=1, nchains=1):
        """nchains > 1: that many Markov chains of the same deck advance in lockstep on this handle (every force
        evaluation is one batched solve of 2*nchains right-hand sides); their fields / momenta are self.X / self.V
        (nchains, Ndof) and model.x / hmc.v are not used."""
        assert 0.0 <= alpha < 1.0                                   # HMC.jl:182
        self.model, self.fa = model, fa
        self.dt, self.tr, self.alpha, self.Nb = float(dt), float(tr), float(alpha), int(Nb)
        self.Nt = int(round(tr / dt))                               # :206
        self.dtp = self.dt / self.Nb                                # :209
        self.Ndof, self.Ndim = model.Ndof, model.Ndim
        self.nchains = int(nchains)
        self.v = np.zeros(model.Ndof)
        self.X = np.zeros((self.nchains, model.Ndof)) if self.nchains > 1 else None
        self.V = np.zeros((self.nchains, model.Ndof)) if self.nchains > 1 else None
        self.updates, self.accepted, self.iters = 1, False, 0
        self.H = self.S = self.K = 0.0
        self.H0 = self.H1 = self.P_accept = 0.0
        self.flag = 0
        self.device_rng = False
        from ._lib import check, dptr, iptr
        if model.kind == models.SSH:      # bond phonons: omega, omega4 and fa.M per phonon
            self.omega4 = getattr(model, "omega4", None)
            if self.omega4 is None:
                self.omega4 = model.omega4 = np.zeros(model.Nph)
            cb_index = np.ascontiguousarray(model.checkerboard_perm[model.phonon_to_bond - 1], dtype=np.int64)
            t_ph = np.ascontiguousarray(model.t[model.phonon_to_bond - 1], dtype=np.float64)
            check(model._lib.elph_hmc_create_ssh_chains(
                model._h, self.nchains, model.Nph, dptr(np.ascontiguousarray(model.omega)), dptr(np.ascontiguousarray(model.omega4)), iptr(cb_index),
                dptr(t_ph), dptr(np.ascontiguousarray(model.alpha)), dptr(np.ascontiguousarray(model.alpha2)), dptr(model.t_bare_cb),
                dptr(np.ascontiguousarray(model.mu)), model.dtau, dptr(np.ascontiguousarray(fa.M))))
            set_shared_fields_(model)
            model._cs_stale = True
        else:
            check(model._lib.elph_hmc_create_chains(model._h, self.nchains, dptr(model.omega), dptr(model.omega4), dptr(model.lam),
                                                    dptr(model.lam2), dptr(model.mu), model.dtau, dptr(np.ascontiguousarray(fa.M))))
        model._nchains = self.nchains
        if self.nchains == 1:
            self.push_()

    def sharing(self, dt, tr, alpha, Nb):
        """HybridMonteCarlo(hmc, Δt, tr, α, Nb) (HMC.jl:225-245): a second parameter set (burn-in) on the same arrays — here the
        same device state; dt, Nt, Nb and α travel with every update call."""
        import copy
        assert 0.0 <= alpha < 1.0
        other = copy.copy(self)
        other.dt, other.tr, other.alpha, other.Nb = float(dt), float(tr), float(alpha), int(Nb)
        other.Nt = int(round(other.tr / other.dt))
        other.dtp = other.dt / other.Nb
        return other

    def device_rng_(self, seed):
        """Draw the random inputs of every later update / special move on the GPU (elph_hmc_set_rng) instead of taking them from
        the host: batch b of the run is synth.randn(batch_seed(seed, b), n), see synth.batch_seed."""
        import ctypes as C
        from ._lib import check
        check(self.model._lib.elph_hmc_set_rng(self.model._h, C.c_uint64(int(seed) & (2 ** 64 - 1))))
        self.device_rng = True

    def push_(self):
        from ._lib import check, dptr
        if self.nchains > 1:
            check(self.model._lib.elph_hmc_set_state(self.model._h, dptr(np.ascontiguousarray(self.X)), dptr(np.ascontiguousarray(self.V))))
        else:
            check(self.model._lib.elph_hmc_set_state(self.model._h, dptr(np.ascontiguousarray(self.model.x)), dptr(self.v)))

    def pull_(self):
        from ._lib import check, dptr
        if self.nchains > 1:
            check(self.model._lib.elph_hmc_get_state(self.model._h, dptr(self.X), dptr(self.V)))
        else:
            check(self.model._lib.elph_hmc_get_state(self.model._h, dptr(self.model.x), dptr(self.v)))


_NO_RANDOMS = {}          # every input NULL: drawn by the handle's own generator (HybridMonteCarlo.device_rng_)


def set_mu_(model, hmc, mu):
    """The chemical-potential tuner moved μ (MuFinder.jl:68-107: model.μ .= μ′ between two updates): the device-side copy of an existing HMC state
    follows — elph_hmc_set_mu (one μ[N] for every chain) or elph_hmc_set_mu_chains (mu of shape (nchains, N): a tuner per chain) — and
    exp(−ΔτV) is rebuilt from the resident field."""
    from ._lib import check, dptr
    mu = np.ascontiguousarray(mu, dtype=np.float64)
    if mu.ndim == 2:
        assert mu.shape == (hmc.nchains, model.Nsites)
        check(model._lib.elph_hmc_set_mu_chains(model._h, dptr(mu.reshape(-1))))
    else:
        assert mu.shape == (model.Nsites,)
        model.mu[:] = mu
        check(model._lib.elph_hmc_set_mu(model._h, dptr(mu)))


def draw_randoms(hmc, rng, with_kpm):
    """The random numbers one update! consumes, in the order the reference draws them from model.rng
    (refresh_v! :652, refresh_ϕ! :674-675, one pair of Arnoldi start vectors per setup!(P), rand :441)."""
    m = hmc.model
    R = rng.standard_normal(m.Ndof)
    if getattr(m, "has_shared_fields", False):
        R = R[m.primary_field]                                      # randn!(R, ssh), SSHModels.jl:568-575
    Rp = rng.standard_normal(m.Ndim)
    Rm = rng.standard_normal(m.Ndim)
    kpm = rng.standard_normal((hmc.Nt + 2, 2, m.Nsites)) if with_kpm else None
    return dict(R=R, Rp=Rp, Rm=Rm, kpm_randn=kpm, u=float(rng.random()))


def update_(model, hmc, fa=None, P=None, rng=None, randoms=None, pull=True):
    """update!(model, hmc, fa, preconditioner) -> (accepted, iters)   (HMC.jl:313-337): one HMC update, the whole
    trajectory (standard_update! for Nb == 1, multitimestep_update! otherwise) in ONE C-ABI call."""
    import ctypes as C
    from ._lib import check, dptr
    if hmc.Ndof == 0:
        return True, 0                                              # :333-335
    if randoms is None:
        randoms = _NO_RANDOMS if hmc.device_rng else draw_randoms(hmc, rng or np.random.default_rng(), P is not None)
    model._push_solver()
    acc, fl = C.c_int(), C.c_int()
    its = C.c_double()
    en = np.zeros(5)
    c = lambda a: dptr(np.ascontiguousarray(a, dtype=np.float64).reshape(-1)) if a is not None else None
    u = randoms.get("u")
    check(model._lib.elph_hmc_update(
        model._h, hmc.dt, hmc.Nt, hmc.Nb, hmc.alpha, 0 if P is None else 1, c(randoms.get("R")), c(randoms.get("Rp")),
        c(randoms.get("Rm")), c(randoms.get("kpm_randn")), float(u) if u is not None else -1.0, C.byref(acc), C.byref(its), dptr(en),
        C.byref(fl)))
    hmc.accepted, hmc.iters, hmc.flag = bool(acc.value), its.value, int(fl.value)
    hmc.H0, hmc.H1, hmc.S, hmc.K, hmc.P_accept = (float(e) for e in en)
    hmc.H = hmc.H1
    hmc.updates += 1                                                # :329
    if pull:
        hmc.pull_()
    return hmc.accepted, hmc.iters


def draw_randoms_chains(hmc, rng, with_kpm):
    """Per chain what draw_randoms draws for one; kpm_randn is [(Nt+2)][b_max|b_min][chain][Nsites]."""
    m, nch = hmc.model, hmc.nchains
    return dict(R=rng.standard_normal((nch, m.Ndof)), Rp=rng.standard_normal((nch, m.Ndim)), Rm=rng.standard_normal((nch, m.Ndim)),
                kpm_randn=rng.standard_normal((hmc.Nt + 2, 2, nch, m.Nsites)) if with_kpm else None, u=rng.random(nch))


def update_chains_(model, hmc, fa=None, P=None, rng=None, randoms=None, pull=False):
    """One HMC update of every chain of `hmc` (nchains Markov chains in lockstep, one batched solve per force / action
    evaluation).  Returns (accepted[nchains] bool, iters[nchains]); per-chain energies in hmc.energies (nchains, 5) =
    H0, H1, S, K, acceptance probability; hmc.flags[nchains] > 0 marks chains rejected because a solve failed."""
    import ctypes as C
    from ._lib import P_int, check, dptr
    nch = hmc.nchains
    if randoms is None:
        randoms = _NO_RANDOMS if hmc.device_rng else draw_randoms_chains(hmc, rng or np.random.default_rng(), P is not None)
    model._push_solver()
    acc = np.zeros(nch, dtype=np.int32)
    fl = np.zeros(nch, dtype=np.int32)
    its, en = np.zeros(nch), np.zeros((nch, 5))
    c = lambda a: dptr(np.ascontiguousarray(a, dtype=np.float64).reshape(-1)) if a is not None else None
    check(model._lib.elph_hmc_update_chains(
        model._h, hmc.dt, hmc.Nt, hmc.Nb, hmc.alpha, 0 if P is None else 1, c(randoms.get("R")), c(randoms.get("Rp")),
        c(randoms.get("Rm")), c(randoms.get("kpm_randn")), c(randoms.get("u")), acc.ctypes.data_as(P_int), dptr(its), dptr(en),
        fl.ctypes.data_as(P_int)))
    hmc.accepted, hmc.iters, hmc.flags, hmc.energies = acc.astype(bool), its, fl, en
    hmc.updates += 1
    if pull:
        hmc.pull_()
    return hmc.accepted, hmc.iters


# ---------------------------------------------------------------------------------------------- special updates

REFLECT, SWAP = 0, 1


def special_move_(model, hmc, kind, col_i, col_j=0, P=None, rng=None, randoms=None):
    """One proposed move of special_update! (SpecialUpdates.jl:103-136 reflection, :205-275 swap) on the device-resident field:
    fresh pseudofermions, the move on phonon column(s) col_i[, col_j] (0-based), Metropolis test on S₁ − S₀.
    -> (accepted, S0, S1, iters, flag).  randoms: dict(Rp, Rm, kpm_randn (2, Nsites) or None, u)."""
    import ctypes as C
    from ._lib import check, dptr
    if randoms is None and hmc.device_rng:
        randoms = _NO_RANDOMS
    if randoms is None:
        rng = rng or np.random.default_rng()
        randoms = dict(Rp=rng.standard_normal(model.Ndim), Rm=rng.standard_normal(model.Ndim),
                       kpm_randn=rng.standard_normal((2, model.Nsites)) if P is not None else None, u=float(rng.random()))
    model._push_solver()
    acc, fl, it = C.c_int(), C.c_int(), C.c_int64()
    s0, s1 = C.c_double(), C.c_double()
    c = lambda a: dptr(np.ascontiguousarray(a, dtype=np.float64).reshape(-1)) if a is not None else None
    u = randoms.get("u")
    check(model._lib.elph_hmc_special_move(model._h, int(kind), int(col_i), int(col_j), c(randoms.get("Rp")), c(randoms.get("Rm")),
                                           0 if P is None else 1, c(randoms.get("kpm_randn")), float(u) if u is not None else -1.0,
                                           C.byref(acc), C.byref(s0), C.byref(s1), C.byref(it), C.byref(fl)))
    if model.kind == models.SSH:
        model._cs_stale = True
    return bool(acc.value), s0.value, s1.value, int(it.value), int(fl.value)


def special_move_chains_(model, hmc, kind, cols_i, cols_j=None, P=None, rng=None, randoms=None):
    """special_move_ for every chain of `hmc` at once: chain c proposes the move on ITS columns cols_i[c] (, cols_j[c]); one
    batched action evaluation, acceptance per chain.  -> (accepted[nch] bool, S0[nch], S1[nch], iters[nch], flag[nch]).
    randoms: dict(Rp (nch, Ndim), Rm, kpm_randn (2, nch, Nsites) or None, u (nch,)); None: drawn from rng / by the library."""
    import ctypes as C
    from ._lib import P_i64, P_int, check, dptr
    nch = hmc.nchains
    if randoms is None and hmc.device_rng:
        randoms = _NO_RANDOMS
    if randoms is None:
        rng = rng or np.random.default_rng()
        randoms = dict(Rp=rng.standard_normal((nch, model.Ndim)), Rm=rng.standard_normal((nch, model.Ndim)),
                       kpm_randn=rng.standard_normal((2, nch, model.Nsites)) if P is not None else None, u=rng.random(nch))
    model._push_solver()
    ci = np.ascontiguousarray(cols_i, dtype=np.int64)
    cj = np.ascontiguousarray(cols_j if cols_j is not None else np.zeros(nch), dtype=np.int64)
    acc, fl = np.zeros(nch, dtype=np.int32), np.zeros(nch, dtype=np.int32)
    it = np.zeros(nch, dtype=np.int64)
    s0, s1 = np.zeros(nch), np.zeros(nch)
    c = lambda a: dptr(np.ascontiguousarray(a, dtype=np.float64).reshape(-1)) if a is not None else None
    check(model._lib.elph_hmc_special_move_chains(
        model._h, int(kind), ci.ctypes.data_as(P_i64), cj.ctypes.data_as(P_i64), c(randoms.get("Rp")), c(randoms.get("Rm")),
        0 if P is None else 1, c(randoms.get("kpm_randn")), c(randoms.get("u")), acc.ctypes.data_as(P_int), dptr(s0), dptr(s1),
        it.ctypes.data_as(P_i64), fl.ctypes.data_as(P_int)))
    if model.kind == models.SSH:
        model._cs_stale = True
    return acc.astype(bool), s0, s1, it, fl


def _isapprox(a, b):
    """Julia's a ≈ b for vectors: ‖a − b‖ ≤ √eps · max(‖a‖, ‖b‖)."""
    return np.linalg.norm(a - b) <= 1.4901161193847656e-08 * max(np.linalg.norm(a), np.linalg.norm(b))


def reflection_update_(model, hmc, nsites, P=None, rng=None):
    """special_update!(model, hmc, ru::ReflectionUpdate, P) (SpecialUpdates.jl:103-136): nsites sites drawn with replacement,
    one move each; returns the accepted fraction.  Holstein only (null operation otherwise, :138-141)."""
    if model.kind != models.HOLSTEIN or nsites < 1:
        return 0.0
    rng = rng or np.random.default_rng()
    if hmc.nchains > 1:      # every chain draws its own sites; move k of all chains is one batched action evaluation
        nmv = min(model.Nph, nsites)
        sites = rng.integers(0, model.Nph, size=(nmv, hmc.nchains))
        return float(np.mean([special_move_chains_(model, hmc, REFLECT, sites[k], P=P, rng=rng)[0] for k in range(nmv)]))
    sites = rng.integers(0, model.Nph, size=min(model.Nph, nsites))
    return sum(special_move_(model, hmc, REFLECT, int(i), P=P, rng=rng)[0] for i in sites) / len(sites)


def swap_update_(model, hmc, nbonds, P=None, rng=None):
    """special_update!(model, hmc, su::SwapUpdate, P): Holstein (:205-236) — the two sites of nbonds randomly drawn bonds swap
    their phonon world lines; SSH (:302-362) — two randomly drawn bond phonons with different world lines swap them."""
    if nbonds < 1:
        return 0.0
    rng = rng or np.random.default_rng()
    if model.kind == models.SSH and hmc.nchains > 1:      # every chain draws its own pair of different world lines
        if model.Nph < 2:
            return 0.0
        acc, L, nch = 0.0, model.Ltau, hmc.nchains
        nbonds = min(model.Nbonds, nbonds)                          # SwapUpdate(model::SSHModel) clamps (SpecialUpdates.jl)
        for _ in range(nbonds):
            hmc.pull_()
            ci, cj = np.zeros(nch, dtype=np.int64), np.zeros(nch, dtype=np.int64)
            for c in range(nch):
                x = hmc.X[c].reshape(model.Nph, L)
                i = j = int(rng.integers(0, model.Nph))
                if not np.allclose(x, x[0], rtol=1e-8, atol=0):
                    j = int(rng.integers(0, model.Nph))
                    while _isapprox(x[i], x[j]):
                        j = int(rng.integers(0, model.Nph))
                ci[c], cj[c] = i, j                                 # i == j (all world lines equal): the swap is the identity
            acc += float(np.mean(special_move_chains_(model, hmc, SWAP, ci, cj, P=P, rng=rng)[0]))
        return acc / nbonds
    if model.kind == models.SSH:
        if model.Nph < 2:
            return 0.0
        acc, L = 0, model.Ltau
        nbonds = min(model.Nbonds, nbonds)                          # SwapUpdate(model::SSHModel) clamps (SpecialUpdates.jl)
        for _ in range(nbonds):
            hmc.pull_()                                             # the draw compares world lines (while xᵢ ≈ xⱼ, :325-327)
            x = model.x.reshape(model.Nph, L)
            if np.allclose(x, x[0], rtol=1e-8, atol=0):
                return acc / nbonds                                 # every phonon the same: the reference would loop forever
            i = int(rng.integers(0, model.Nph))
            j = int(rng.integers(0, model.Nph))
            while _isapprox(x[i], x[j]):
                j = int(rng.integers(0, model.Nph))
            acc += special_move_(model, hmc, SWAP, i, j, P=P, rng=rng)[0]
        return acc / nbonds
    if model.Nbonds == 0:
        return 0.0
    if hmc.nchains > 1:
        nmv = min(model.Nbonds, nbonds)
        bonds = rng.integers(0, model.Nbonds, size=(nmv, hmc.nchains))
        tab = model.neighbor_table
        return float(np.mean([special_move_chains_(model, hmc, SWAP, tab[bonds[k], 0] - 1, tab[bonds[k], 1] - 1, P=P, rng=rng)[0]
                              for k in range(nmv)]))
    bonds = rng.integers(0, model.Nbonds, size=min(model.Nbonds, nbonds))
    tab = model.neighbor_table
    return sum(special_move_(model, hmc, SWAP, int(tab[b, 0]) - 1, int(tab[b, 1]) - 1, P=P, rng=rng)[0] for b in bonds) / len(bonds)
