"""Host-side mirror of LangevinDynamics.jl: Euler / Runge-Kutta / Heun dynamics with the whole step on the GPU.

    dyn = EulerDynamics(model, fa, dt) | RungeKuttaDynamics(...) | HeunsDynamics(...)     LangevinDynamics.jl:25-78,135-160,245-270
    evolve_(model, dyn, fa, P=None, rng=None, randoms=None) -> iters                      :81-130, :162-232, :272-328

The field stays on the device between steps (`dyn.pull_()` refreshes model.x, `dyn.push_()` uploads it); the random numbers
the reference draws from model.rng (η, the noise vectors g of calc_dSfdx!, the Arnoldi start vectors of setup!(P)) are
inputs, drawn here in the reference's order when not given.
"""
import ctypes as C

import numpy as np

from . import models
from ._lib import P_int, check, dptr

EULER, RUNGE_KUTTA, HEUN = 0, 1, 2


class _Dynamics:
    scheme = None

    def __init__(self, model, fa, dt, nchains=1):
        """nchains > 1 (Holstein): that many independent trajectories of the same deck advance in lockstep on this handle
        (every step one batched solve of nchains right-hand sides, one KPM expansion per chain); their fields are
        self.X (nchains, Ndof) and model.x is not used."""
        self.model, self.fa, self.dt = model, fa, float(dt)
        self.Ndof, self.Ndim = model.Ndof, model.Ndim
        self.flag = 0
        self.nchains = int(nchains)
        if self.nchains < 1:
            raise ValueError("nchains < 1")
        self.X = np.tile(model.x, (self.nchains, 1)) if self.nchains > 1 else None
        self.flags = np.zeros(self.nchains, dtype=np.int32)
        self.device_rng = False
        if model.kind == models.SSH:
            from ._lib import iptr
            if getattr(model, "omega4", None) is None:
                model.omega4 = np.zeros(model.Nph)
            cb_index = np.ascontiguousarray(model.checkerboard_perm[model.phonon_to_bond - 1], dtype=np.int64)
            t_ph = np.ascontiguousarray(model.t[model.phonon_to_bond - 1], dtype=np.float64)
            check(model._lib.elph_hmc_create_ssh_chains(      # the Langevin state is the HMC state with fa.Q in place of fa.M
                model._h, self.nchains, model.Nph, dptr(np.ascontiguousarray(model.omega)), dptr(np.ascontiguousarray(model.omega4)), iptr(cb_index),
                dptr(t_ph), dptr(np.ascontiguousarray(model.alpha)), dptr(np.ascontiguousarray(model.alpha2)), dptr(model.t_bare_cb),
                dptr(np.ascontiguousarray(model.mu)), model.dtau, dptr(np.ascontiguousarray(fa.Q))))
            from .hmc import set_shared_fields_
            set_shared_fields_(model)
            model._cs_stale = True
        elif self.nchains > 1:
            check(model._lib.elph_langevin_create_chains(model._h, self.nchains, dptr(model.omega), dptr(model.omega4), dptr(model.lam),
                                                         dptr(model.lam2), dptr(model.mu), model.dtau, dptr(np.ascontiguousarray(fa.Q))))
        else:
            check(model._lib.elph_langevin_create(model._h, dptr(model.omega), dptr(model.omega4), dptr(model.lam), dptr(model.lam2),
                                                  dptr(model.mu), model.dtau, dptr(np.ascontiguousarray(fa.Q))))
        model._nchains = self.nchains
        self.push_()

    def device_rng_(self, seed):
        """Draw eta, g1, g2 (and the Arnoldi start vectors) of every later step inside the library (elph_hmc_set_rng)."""
        check(self.model._lib.elph_hmc_set_rng(self.model._h, C.c_uint64(int(seed) & (2 ** 64 - 1))))
        self.device_rng = True

    def _field(self):
        return self.X if self.nchains > 1 else self.model.x

    def push_(self):
        check(self.model._lib.elph_hmc_set_state(self.model._h, dptr(np.ascontiguousarray(self._field()).reshape(-1)), None))

    def pull_(self):
        check(self.model._lib.elph_hmc_get_state(self.model._h, dptr(self._field().reshape(-1)), None))


class EulerDynamics(_Dynamics):
    scheme = EULER


class RungeKuttaDynamics(_Dynamics):
    scheme = RUNGE_KUTTA


class HeunsDynamics(_Dynamics):
    scheme = HEUN


def draw_randoms(dyn, rng, with_kpm):
    m, nch = dyn.model, dyn.nchains
    out = dict(eta=rng.standard_normal((nch, m.Ndof)), g1=rng.standard_normal((nch, m.Ndim)))
    if getattr(m, "has_shared_fields", False):
        out["eta"] = out["eta"][:, m.primary_field]                  # randn!(η, model), LangevinDynamics.jl:97
    out["g2"] = rng.standard_normal((nch, m.Ndim)) if dyn.scheme != EULER else None
    out["kpm_randn"] = rng.standard_normal((2, 2, nch, m.Nsites)) if with_kpm else None     # [set-up][b_max|b_min][chain][site]
    return out


def evolve_(model, dyn, fa=None, P=None, rng=None, randoms=None, pull=True):
    """evolve!(model, dyn, fa, preconditioner) -> iters: one Langevin step, entirely on the device.  With dyn.nchains > 1 the
    random vectors are chain-major (see draw_randoms) and the return value is iters[nchains]; dyn.flags holds the solver flag
    of every chain."""
    if randoms is None:
        randoms = {} if dyn.device_rng else draw_randoms(dyn, rng or np.random.default_rng(), P is not None)
    model._push_solver()
    nch = dyn.nchains
    it, fl = (C.c_int64 * nch)(), (C.c_int * nch)()
    c = lambda a: dptr(np.ascontiguousarray(a, dtype=np.float64).reshape(-1)) if a is not None else None
    check(model._lib.elph_langevin_evolve(model._h, dyn.scheme, dyn.dt, 0 if P is None else 1, c(randoms.get("eta")), c(randoms.get("g1")),
                                          c(randoms.get("g2")), c(randoms.get("kpm_randn")), it, fl))
    dyn.flags = np.array(fl[:], dtype=np.int32)
    dyn.flag = int(dyn.flags.max())
    if model.kind == models.SSH:
        model._cs_stale = True
    if pull:
        dyn.pull_()
    return int(it[0]) if nch == 1 else np.array(it[:], dtype=np.int64)
