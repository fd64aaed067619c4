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

    def __init__(self, model, fa, dt):
        self.model, self.fa, self.dt = model, fa, float(dt)
        self.Ndof, self.Ndim = model.Ndof, model.Ndim
        self.flag = 0
        if model.kind == models.SSH:
            from ._lib import iptr
            if getattr(model, "omega4", None) is None:
                model.omega4 = np.zeros(model.Nph)
            cb_index = np.ascontiguousarray(model.checkerboard_perm[model.phonon_to_bond - 1], dtype=np.int64)
            t_ph = np.ascontiguousarray(model.t[model.phonon_to_bond - 1], dtype=np.float64)
            check(model._lib.elph_langevin_create_ssh(
                model._h, model.Nph, dptr(np.ascontiguousarray(model.omega)), dptr(np.ascontiguousarray(model.omega4)), iptr(cb_index),
                dptr(t_ph), dptr(np.ascontiguousarray(model.alpha)), dptr(np.ascontiguousarray(model.alpha2)), dptr(model.t_bare_cb),
                dptr(np.ascontiguousarray(model.mu)), model.dtau, dptr(np.ascontiguousarray(fa.Q))))
            model._cs_stale = True
        else:
            check(model._lib.elph_langevin_create(model._h, dptr(model.omega), dptr(model.omega4), dptr(model.lam), dptr(model.lam2),
                                                  dptr(model.mu), model.dtau, dptr(np.ascontiguousarray(fa.Q))))
        model._nchains = 1
        self.push_()

    def push_(self):
        check(self.model._lib.elph_hmc_set_state(self.model._h, dptr(np.ascontiguousarray(self.model.x)), None))

    def pull_(self):
        check(self.model._lib.elph_hmc_get_state(self.model._h, dptr(self.model.x), None))


class EulerDynamics(_Dynamics):
    scheme = EULER


class RungeKuttaDynamics(_Dynamics):
    scheme = RUNGE_KUTTA


class HeunsDynamics(_Dynamics):
    scheme = HEUN


def draw_randoms(dyn, rng, with_kpm):
    m = dyn.model
    out = dict(eta=rng.standard_normal(m.Ndof), g1=rng.standard_normal(m.Ndim))
    out["g2"] = rng.standard_normal(m.Ndim) if dyn.scheme != EULER else None
    out["kpm_randn"] = rng.standard_normal((2, 2, m.Nsites)) if with_kpm else None
    return out


def evolve_(model, dyn, fa=None, P=None, rng=None, randoms=None, pull=True):
    """evolve!(model, dyn, fa, preconditioner) -> iters: one Langevin step, entirely on the device."""
    if randoms is None:
        randoms = draw_randoms(dyn, rng or np.random.default_rng(), P is not None)
    model._push_solver()
    it, fl = C.c_int64(), C.c_int()
    c = lambda a: dptr(np.ascontiguousarray(a, dtype=np.float64).reshape(-1)) if a is not None else None
    check(model._lib.elph_langevin_evolve(model._h, dyn.scheme, dyn.dt, 0 if P is None else 1, c(randoms["eta"]), c(randoms["g1"]),
                                          c(randoms.get("g2")), c(randoms.get("kpm_randn")), C.byref(it), C.byref(fl)))
    dyn.flag = int(fl.value)
    if model.kind == models.SSH:
        model._cs_stale = True
    if pull:
        dyn.pull_()
    return int(it.value)
