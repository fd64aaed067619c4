"""Mirror of KPMPreconditioners.jl (symmetric / CG variant) and FourierAcceleration.jl on the GPU path.

    P = SymmetricKPMPreconditioner(model, n, buf, c1, c2)     KPMPreconditioners.jl:219-235
    setup_(P[, rng | e_min=, e_max=])                         :259-321
    ldiv_(z, P, r)                                            :426-481
    fa = FourierAccelerator(model); update_M_/update_Q_;      FourierAcceleration.jl:11-82,149-167
    fourier_accelerate_(v_out, fa, v, power, use_mass=False)  :91-143

The Left/Right/LeftRight KPM variants belong to GMRES/BiCGStab and are out of scope.
"""
import ctypes as C
import math

import numpy as np

from ._lib import check, dptr


class SymmetricKPMPreconditioner:
    def __init__(self, model, n=20, buf=0.05, c1=1.0, c2=1.0):
        self.model = model
        self.n, self.buf, self.c1, self.c2 = int(n), float(buf), float(c1), float(c2)
        check(model._lib.elph_kpm_create(model._h, self.n, self.buf, self.c1, self.c2))
        self.active = True
        self.lam_lo, self.lam_hi = 0.0, 2.0          # ctor defaults, KPMPreconditioners.jl:108-109

    @property
    def orders(self):
        Lo2 = (self.model.Ltau + 1) // 2
        o = np.zeros(Lo2, dtype=np.int64)
        tot = C.c_int64()
        check(self.model._lib.elph_kpm_orders(self.model._h, o.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(tot)))
        return o


def setup_(P, rng=None, e_min=None, e_max=None, b_max=None, b_min=None):
    """setup!(P).  `None` mirrors the reference's no-op for the identity (KPMPreconditioners.jl:323-326).

    The reference draws the two Arnoldi start vectors from model.rng (:859-861,902-904).  Here they come
    from `rng` (numpy Generator), or explicitly as b_max/b_min; or the eigenvalue bounds are injected
    (e_min/e_max) for parity runs, because RNG streams are not reproducible across languages."""
    if P is None:
        return
    m = P.model
    N = m.Nsites
    if e_min is None or e_max is None:
        e_min = e_max = math.nan
        if b_max is None:
            rng = rng or np.random.default_rng()
            b_max, b_min = rng.standard_normal(N), rng.standard_normal(N)
    act = C.c_int()
    lo, hi = C.c_double(), C.c_double()
    check(m._lib.elph_kpm_setup(m._h, dptr(np.ascontiguousarray(b_max)) if b_max is not None else None,
                                dptr(np.ascontiguousarray(b_min)) if b_min is not None else None,
                                e_min, e_max, C.byref(act), C.byref(lo), C.byref(hi)))
    P.active, P.lam_lo, P.lam_hi = bool(act.value), lo.value, hi.value


def setup_chains_(P, rng=None, e_min=None, e_max=None, b_max=None, b_min=None):
    """setup!(P) for every chain resident after models.update_model_chains_: one expansion per phonon configuration
    (the reference runs chains as separate processes, each with its own preconditioner).  Arrays are per chain:
    b_max/b_min (nchains, Nsites), e_min/e_max (nchains,).  Returns (active, lam_lo, lam_hi) arrays."""
    m = P.model
    nch = int(m._nchains)
    N = m.Nsites
    if e_min is None or e_max is None:
        e_min = e_max = None
        if b_max is None:
            rng = rng or np.random.default_rng()
            b_max, b_min = rng.standard_normal((nch, N)), rng.standard_normal((nch, N))
    else:
        e_min = np.ascontiguousarray(np.broadcast_to(np.asarray(e_min, dtype=np.float64), (nch,)))
        e_max = np.ascontiguousarray(np.broadcast_to(np.asarray(e_max, dtype=np.float64), (nch,)))
    act = np.zeros(nch, dtype=np.int32)
    lo, hi = np.zeros(nch), np.zeros(nch)
    arr = lambda a: dptr(np.ascontiguousarray(a, dtype=np.float64)) if a is not None else None
    check(m._lib.elph_kpm_setup_chains(m._h, arr(b_max), arr(b_min), arr(e_min), arr(e_max),
                                       act.ctypes.data_as(C.POINTER(C.c_int)), dptr(lo), dptr(hi)))
    P.active, P.lam_lo, P.lam_hi = bool(act.any()), float(lo[0]), float(hi[0])
    return act, lo, hi


def kpm_ldiv_(z, P, r):
    """ldiv!(z, P, r): z = P^-1 r; plain copy for the identity (IterativeSolvers.jl:14-17)."""
    if P is None:
        z[:] = r
        return
    m = P.model
    check(m._lib.elph_kpm_apply(m._h, dptr(z), dptr(r)))


class FourierAccelerator:
    """FourierAcceleration.jl:11-82: diagonal Q and M per (phonon, tau-mode), frequency index fastest."""

    def __init__(self, model):
        self.model = model
        self.N, self.L = model.Nph, model.Ltau
        self.Q = np.zeros(self.N * self.L)
        self.M = np.zeros(self.N * self.L)


def element_Qi(k, omega, dtau, m, L):
    """FourierAcceleration.jl:213-217."""
    return (m ** 2 + dtau * omega * omega + 4.0 / dtau) / (m ** 2 + dtau * omega * omega + (2 - 2 * math.cos(2 * math.pi * k / L)) / dtau)


def element_Mi(k, omega, dtau, m0, c, L):
    """FourierAcceleration.jl:260-266."""
    kp = min(k, L - k)
    m = m0 * math.exp(-(c * kp / L) ** 2)
    return dtau * (m ** 2 + omega ** 2 + (2 - 2 * math.cos(2 * math.pi * kp / L)) / dtau ** 2) / (m ** 2 + omega ** 2)


def update_Q_(fa, model, omega_min, omega_max, m):
    """FourierAcceleration.jl:149-155,176-193."""
    for ph in range(fa.N):
        if omega_min < model.omega[ph] < omega_max:
            fa.Q[ph * fa.L:(ph + 1) * fa.L] = [element_Qi(k, model.omega[ph], model.dtau, m, fa.L) for k in range(fa.L)]


def update_M_(fa, model, omega_min, omega_max, m0, c=0.0):
    """FourierAcceleration.jl:161-167,222-240."""
    for ph in range(fa.N):
        if omega_min < model.omega[ph] < omega_max:
            fa.M[ph * fa.L:(ph + 1) * fa.L] = [element_Mi(k, model.omega[ph], model.dtau, m0, c, fa.L) for k in range(fa.L)]


def fourier_accelerate_(v_out, fa, v, power, use_mass=False):
    """v_out = Re iFFT_tau( D^power .* FFT_tau(v) ), D = M (use_mass) or Q  (FourierAcceleration.jl:91-143)."""
    m = fa.model
    diag = fa.M if use_mass else fa.Q
    check(m._lib.elph_fourier_accelerate(m._h, dptr(v_out), dptr(np.ascontiguousarray(v)), dptr(diag), float(power), fa.N))
