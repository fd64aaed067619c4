"""Host-side mirror of GreensFunctions.jl: the stochastic Green's-function estimator.

    est = EstimateGreensFunction(model, nv)      GreensFunctions.jl:155-195
    update_(est, model, P=None, R=None)          :201-234   n_v noise vectors, n_v solves as one batched CG on the GPU
    setup_(est, n1, n2)                          :239-288   the four translation-averaged products (device τ/space DFTs)
    measure_GD0(est, l1, l2, l3, o1, o2, tau)    :293-298   (and _GD0_GD0, _GDD_G00, _GD0_G0D, :303-329)
    estimate(est, i, j, tau2, tau1, sigma)       :334-346

Arrays keep the reference's shapes: est.GD0 etc. are complex128 numpy arrays of shape (2L, n_s, n_s, L1, L2, L3) in
Fortran order (Julia's memory image), R / MinvR are (n_v, Ndim) (row = Julia column).  Indices are the reference's:
n1, n2, orbitals, sites and tau 1-based; cell offsets l1, l2, l3 0-based.
"""
import ctypes as C

import numpy as np

from . import preconditioners as pc
from ._lib import P_dbl, P_i64, P_int, check, dptr


class EstimateGreensFunction:
    def __init__(self, model, nv=2):
        lat = model.lattice
        self.model = model
        self.nv = max(2, int(nv))                                          # :167
        self.n1, self.n2 = 1, 2
        self.NL, self.L, self.N = model.Ndim, model.Ltau, model.Nsites
        self.L1, self.L2, self.L3, self.ns = lat.L1, lat.L2, lat.L3, lat.norbits
        self.R = np.zeros((self.nv, self.NL))
        self.MinvR = np.zeros((self.nv, self.NL))
        shp = (2 * self.L, self.ns, self.ns, self.L1, self.L2, self.L3)
        self.GD0 = np.zeros(shp, dtype=np.complex128, order="F")
        self.GDD_G00 = np.zeros(shp, dtype=np.complex128, order="F")
        self.GD0_GD0 = np.zeros(shp, dtype=np.complex128, order="F")
        self.GD0_G0D = np.zeros(shp, dtype=np.complex128, order="F")
        check(model._lib.elph_greens_create(model._h, self.ns, self.L1, self.L2, self.L3, self.nv))

    # views of the pair selected by setup_ (estimator.r₁ … M⁻¹r₂)
    @property
    def r1(self):
        return self.R[self.n1 - 1]

    @property
    def r2(self):
        return self.R[self.n2 - 1]

    @property
    def Minvr1(self):
        return self.MinvR[self.n1 - 1]

    @property
    def Minvr2(self):
        return self.MinvR[self.n2 - 1]


def update_(est, model, P=None, rng=None, R=None, setup_kwargs=None):
    """update!(estimator, model, preconditioner).  The noise vectors come from `R` ((n_v, Ndim)) or from `rng`
    (numpy Generator) — Julia's Xoshiro stream is not reproducible here, so parity runs pass R explicitly.
    Returns (iters, residual_error, flag) per vector (the reference discards them)."""
    m = est.model
    assert model is m
    if getattr(m, "_nchains", 1) > 1:
        # chains in lockstep: right-hand side r of the batch belongs to chain r % nchains, so an estimator made with
        # nv = n_v * nchains vectors holds vector v of chain c at index v * nchains + c (chain_vector below)
        if est.nv % m._nchains:
            raise ValueError("with chains resident the estimator needs a multiple of nchains vectors")
        if P is not None:
            pc.setup_chains_(P, rng=rng, **(setup_kwargs or {}))
    else:
        pc.setup_(P, rng=rng, **(setup_kwargs or {}))                      # :206
    if R is None:
        rng = rng or np.random.default_rng()
        R = rng.standard_normal((est.nv, m.Ndim))
    est.R[:] = R
    it = np.zeros(est.nv, dtype=np.int64)
    res = np.zeros(est.nv)
    fl = np.zeros(est.nv, dtype=np.int32)
    use_prec = 0 if P is None else 1
    check(m._lib.elph_greens_update(m._h, dptr(est.R), use_prec, it.ctypes.data_as(P_i64), dptr(res), fl.ctypes.data_as(P_int)))
    check(m._lib.elph_greens_get_vectors(m._h, None, dptr(est.MinvR)))
    return it, res, fl


def chain_vector(est, chain, v):
    """1-based index (for setup_ / estimate) of noise vector v (1-based, v <= est.nv // nchains) of chain `chain` (0-based) in an
    estimator that serves several resident chains: setup_(est, chain_vector(est, c, 1), chain_vector(est, c, 2)) selects the
    first pair of chain c."""
    nch = int(est.model._nchains)
    return (int(v) - 1) * nch + int(chain) + 1


def set_vectors_(est, R, MinvR):
    """Replay vectors produced elsewhere (e.g. dumped from a Julia run)."""
    est.R[:], est.MinvR[:] = R, MinvR
    check(est.model._lib.elph_greens_set_vectors(est.model._h, dptr(est.R), dptr(est.MinvR)))


def setup_(est, n1, n2):
    """setup!(estimator, n₁, n₂): fills est.GD0, est.GD0_GD0, est.GDD_G00, est.GD0_G0D."""
    est.n1, est.n2 = int(n1), int(n2)
    m = est.model
    ptr = lambda a: a.ctypes.data_as(P_dbl)                                # F-ordered complex128 = interleaved doubles
    check(m._lib.elph_greens_setup(m._h, est.n1, est.n2, ptr(est.GD0), ptr(est.GD0_GD0), ptr(est.GDD_G00), ptr(est.GD0_G0D)))


def _measure(est, G, l1, l2, l3, o1, o2, tau):
    return G[tau % (2 * est.L), o2 - 1, o1 - 1, l1, l2, l3]               # mod1(τ+1, 2L), o₂, o₁, l+1


def measure_GD0(est, l1, l2, l3, o1, o2, tau):
    return _measure(est, est.GD0, l1, l2, l3, o1, o2, tau)


def measure_GD0_GD0(est, l1, l2, l3, o1, o2, tau):
    return _measure(est, est.GD0_GD0, l1, l2, l3, o1, o2, tau)


def measure_GDD_G00(est, l1, l2, l3, o1, o2, tau):
    return _measure(est, est.GDD_G00, l1, l2, l3, o1, o2, tau)


def measure_GD0_G0D(est, l1, l2, l3, o1, o2, tau):
    return _measure(est, est.GD0_G0D, l1, l2, l3, o1, o2, tau)


def estimate(est, i, j, tau2, tau1, sigma):
    m = (j - 1) * est.L + tau1 - 1
    n = (i - 1) * est.L + tau2 - 1
    if sigma == 1:
        return est.Minvr1[n] * est.r1[m]
    if sigma == 2:
        return est.Minvr2[n] * est.r2[m]
    raise ValueError("sigma must be 1 or 2")                               # DomainError, :343
