"""Host-side mirror of the reference's model / operator API for the hot path.

The reference exposes the fermion matrix through Julia multiple dispatch on the model type
(Models.jl:65-248, HolsteinModels.jl, SSHModels.jl, IterativeSolvers.jl:36-57).  These classes keep the
same field names, argument order (output first, in place) and return conventions, and forward every
operation to libelphgpu.so through its C ABI (include/elph_gpu.h).  Julia's `f!` is spelled `f_` here.

    update_model_(model)                         HolsteinModels.jl:526 / SSHModels.jl:510
    mul_(y, model, v)                            Models.jl:192   (honours model.mul_by_M / .transposed)
    mulM_(y, model, v), mulMt_(y, model, v)      HolsteinModels.jl:569,631 / SSHModels.jl:581,646
    mulMtM_(y, model, v)                         Models.jl:215
    ldiv_(x, model, b, P=None, maxiter=0)        Models.jl:74,139  -> (iters, residual_error, flag)
    solve_(x, model, b, cg, P=None, ...)         IterativeSolvers.jl:153,239 -> iters
    transpose_(model)                            Models.jl:244

All vectors are numpy float64, flat, reference layout (tau fastest, Utilities.jl:12-15).
"""
import ctypes as C

import numpy as np

from . import _lib
from . import lattice as _lat
from ._lib import check, dptr, iptr

HOLSTEIN, SSH = 0, 1


def _rng(rng):
    if rng is None:
        raise ValueError("a disorder width (stddev) needs rng=numpy.random.Generator")
    return rng


class ConjugateGradient:
    """IterativeSolvers.jl:36-57 (tol, maxiter, kmax); the work vectors live on the GPU."""

    def __init__(self, ndim, tol=1e-4, maxiter=0, kmax=1e12):
        self.tol = float(tol)
        self.maxiter = int(maxiter) if maxiter >= 1 else int(ndim)
        self.kmax = float(kmax)
        self.N = int(ndim)


class AbstractModel:
    """Common part of HolsteinModel / SSHModel (Models.jl:65): owns the GPU handle."""

    kind = None

    def _create(self, cosht=None, sinht=None, device=0):
        lib = _lib.load()
        h = _lib.Handle()
        tab = np.ascontiguousarray(self.neighbor_table, dtype=np.int64)
        check(lib.elph_create(C.byref(h), self.kind, self.Nsites, self.Ltau, self.Nbonds,
                              iptr(tab) if self.Nbonds > 0 else None,
                              dptr(cosht) if cosht is not None and self.Nbonds > 0 else None,
                              dptr(sinht) if sinht is not None and self.Nbonds > 0 else None, device))
        self._h = h
        self._lib = lib
        self._push_solver()

    def _push_solver(self):
        check(self._lib.elph_solver_set(self._h, self.solver.tol, self.solver.maxiter, self.solver.kmax))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.elph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # Base.length / size (Models.jl:262-284)
    def __len__(self):
        return self.Ndim

    @property
    def shape(self):
        return (self.Ndim, self.Ndim)


class HolsteinModel(AbstractModel):
    """HolsteinModels.jl:22-314.  Build with the same incremental calls as ProcessInputFile.jl:216-326:
    assign_t_ per bond definition, assign_mu_/lambda_/..., then initialize_model_()."""

    kind = HOLSTEIN

    def __init__(self, lattice, beta, dtau, tol=1e-4, maxiter=10000, device=0):
        self.lattice = lattice
        self.beta, self.dtau = float(beta), float(dtau)
        self.Ltau = _lat.ltau_from_beta(beta, dtau)                 # HolsteinModels.jl:205
        self.Nsites = lattice.nsites
        self.Nph = self.Nsites
        self.Ndof = self.Nph * self.Ltau
        self.Ndim = self.Ndof
        self.Nbonds = 0
        self.nbonds = 0
        self.x = np.zeros(self.Ndof)
        self.expnDtauV = None                                       # lives on the GPU (layout S)
        self.t = np.zeros(0)
        self.neighbor_table = np.zeros((0, 2), dtype=np.int64)
        self.cosht = np.zeros(0)
        self.sinht = np.zeros(0)
        self.checkerboard_perm = np.zeros(0, dtype=np.int64)
        self.omega = np.zeros(self.Nph)
        self.omega4 = np.zeros(self.Nph)
        self.lam = np.zeros(self.Nph)
        self.lam2 = np.zeros(self.Nph)
        self.mu = np.zeros(self.Nsites)
        self.mul_by_M = False                                       # CG works on MtM (HolsteinModels.jl:268-276)
        self.transposed = False
        self.solver = ConjugateGradient(self.Ndim, tol=tol, maxiter=maxiter)
        self._device = device
        self._h = None

    # -- incremental specification (HolsteinModels.jl:323-444).  Disorder (stddev != 0) draws standard normals from `rng`
    #    (a numpy Generator) in the reference's order; Julia's Xoshiro stream itself is not reproducible here.
    def assign_t_(self, t, o1, o2, v, stddev=0.0, rng=None):
        new = self.lattice.calc_neighbor_table(o1, o2, v)
        self.neighbor_table = np.concatenate([self.neighbor_table, new], axis=0)
        n = new.shape[0]
        phase = t / abs(t)                                          # :437 (t = 0 is NaN in the reference too)
        t_new = np.full(n, abs(float(t)))
        if stddev != 0.0:
            t_new = t_new + stddev * _rng(rng).standard_normal(n)
        self.t = np.concatenate([self.t, phase * t_new])
        self.nbonds += 1

    def _assign(self, arr, val, orbit, stddev=0.0, rng=None):
        sel = slice(None) if orbit == 0 else slice(orbit - 1, None, self.lattice.norbits)
        n = len(arr[sel])
        arr[sel] = val + (stddev * _rng(rng).standard_normal(n) if stddev != 0.0 else 0.0)

    def assign_mu_(self, val, orbit=0, stddev=0.0, rng=None):
        self._assign(self.mu, val, orbit, stddev, rng)

    def assign_lambda_(self, val, orbit=0, stddev=0.0, rng=None):
        self._assign(self.lam, val, orbit, stddev, rng)

    def assign_lambda2_(self, val, orbit=0, stddev=0.0, rng=None):
        self._assign(self.lam2, val, orbit, stddev, rng)

    def assign_omega_(self, val, orbit=0, stddev=0.0, rng=None):
        self._assign(self.omega, val, orbit, stddev, rng)

    def assign_omega4_(self, val, orbit=0, stddev=0.0, rng=None):
        self._assign(self.omega4, val, orbit, stddev, rng)

    def initialize_model_(self):
        """HolsteinModels.jl:484-517, then create the GPU-side model."""
        if len(self.t) > 0:
            self.Nbonds = len(self.t)
            cb = _lat.initialize_checkerboard(self.neighbor_table, self.t, self.dtau)
            self.neighbor_table = cb["table"]
            self.cosht, self.sinht = cb["cosht"], cb["sinht"]
            self.checkerboard_perm = cb["cb_perm"]
            self.colours = cb["colours"]
        self._create(self.cosht, self.sinht, self._device)


class SSHModel(AbstractModel):
    """SSHModels.jl:79-314 reduced to what the path needs: bonds with optional bond phonons.

    update_model_ computes cosht / sinht on the GPU; the host-visible `model.cosht`, `model.sinht` ((Nbonds, Ltau), i.e.
    Julia's (Ltau x Nbonds) column-major) are fetched from the device on first access after an update."""

    kind = SSH

    def _cs(self, name):
        if getattr(self, "_cs_stale", False) and getattr(self, "_h", None):
            check(self._lib.elph_get_cosh_sinh(self._h, dptr(self._cosht.reshape(-1)), dptr(self._sinht.reshape(-1))))
            self._cs_stale = False
        return getattr(self, name)

    cosht = property(lambda self: self._cs("_cosht"), lambda self, v: setattr(self, "_cosht", v))
    sinht = property(lambda self: self._cs("_sinht"), lambda self, v: setattr(self, "_sinht", v))

    def __init__(self, lattice, beta, dtau, tol=1e-4, maxiter=10000, device=0):
        self.lattice = lattice
        self.beta, self.dtau = float(beta), float(dtau)
        self.Ltau = _lat.ltau_from_beta(beta, dtau)
        self.Nsites = lattice.nsites
        self.Ndim = self.Nsites * self.Ltau
        self.bond_definitions = []
        self.mu = np.zeros(self.Nsites)
        self.mul_by_M = False
        self.transposed = False
        self.solver = ConjugateGradient(self.Ndim, tol=tol, maxiter=maxiter)
        self._device = device
        self._h = None

    def assign_hopping_(self, t, alpha, alpha2, omega, o1, o2, v, has_phonon=None, omega4=0.0, name="", t_std=0.0,
                        omega_std=0.0, omega4_std=0.0, alpha_std=0.0, alpha2_std=0.0):
        """assign_hopping!(ssh, t, σt, ω, σω, ω₄, σω₄, α, σα, α₂, σα₂, o₁, o₂, dL, name) (SSHModels.jl:319-342); a bond
        definition carries a phonon iff ω or σω is non-zero (:74) unless has_phonon is given."""
        if has_phonon is None:
            has_phonon = (omega != 0.0) or (omega_std != 0.0)
        v = tuple(v) + (0,) * (3 - len(v))
        self.bond_definitions.append(dict(t=float(t), alpha=float(alpha), alpha2=float(alpha2), omega=float(omega), omega4=float(omega4),
                                          o1=o1, o2=o2, v=v, has_phonon=bool(has_phonon), name=str(name), t_std=float(t_std),
                                          omega_std=float(omega_std), omega4_std=float(omega4_std), alpha_std=float(alpha_std),
                                          alpha2_std=float(alpha2_std)))

    def initialize_model_(self, rng=None):
        """SSHModels.jl:348-505.  Disorder widths draw from `rng` (numpy Generator) in the reference's order (:381-411)."""
        tabs, t, alpha, alpha2, omega, omega4, ph2b, names = [], [], [], [], [], [], [], []
        nb_so_far = 0

        def spread(mean, std, n, signed):
            ph = 1.0 if (not signed or mean == 0.0) else mean / abs(mean)
            base = np.full(n, abs(mean) if signed else mean)
            if std != 0.0:
                base = base + std * _rng(rng).standard_normal(n)
            return list(ph * base)

        for d in self.bond_definitions:
            new = self.lattice.calc_neighbor_table(d["o1"], d["o2"], d["v"])
            n = new.shape[0]
            tabs.append(new)
            t += spread(d["t"], d.get("t_std", 0.0), n, True)
            if d["has_phonon"]:
                names.append(d.get("name", ""))
                omega += spread(d["omega"], d.get("omega_std", 0.0), n, False)
                omega4 += spread(d.get("omega4", 0.0), d.get("omega4_std", 0.0), n, False)
                alpha += spread(d["alpha"], d.get("alpha_std", 0.0), n, True)
                alpha2 += spread(d["alpha2"], d.get("alpha2_std", 0.0), n, True)
                ph2b += list(range(nb_so_far + 1, nb_so_far + n + 1))      # 1-based bond of each phonon (:413)
            nb_so_far += n
        self.nph = len(names)                                              # number of phonon types
        self.phonon_names = names
        raw = np.concatenate(tabs, axis=0) if tabs else np.zeros((0, 2), dtype=np.int64)
        cb = _lat.initialize_checkerboard(raw)
        self.neighbor_table = cb["table"]
        self.checkerboard_perm = cb["cb_perm"]
        self.inv_checkerboard_perm = cb["inv_cb_perm"]
        self.colours = cb["colours"]
        self.Nbonds = raw.shape[0]
        self.t = np.array(t)
        self.alpha, self.alpha2, self.omega, self.omega4 = np.array(alpha), np.array(alpha2), np.array(omega), np.array(omega4)
        self.phonon_to_bond = np.array(ph2b, dtype=np.int64)
        self.Nph = len(ph2b)
        self.Ndof = self.Nph * self.Ltau
        self.x = np.zeros(self.Ndof)
        # phonon types with the same name share their fields: primary_field (:480-502), 0-based here; the reference's
        # tabulation assumes equally many phonons per type (reshape to (Ndof/nph, nph))
        self.primary_field = np.arange(self.Ndof, dtype=np.int64)
        if self.nph > 1 and len(set(names)) < self.nph:
            if self.Ndof % self.nph:
                raise ValueError("phonon types of unequal size cannot share fields (SSHModels.jl:482)")
            per = self.Ndof // self.nph
            pf = self.primary_field.reshape(self.nph, per)
            for a in range(self.nph):
                for b in range(a + 1, self.nph):
                    if names[a] == names[b] and pf[b, 0] > a * per:
                        pf[b, :] = np.arange(a * per, (a + 1) * per)
        self.has_shared_fields = bool(np.any(self.primary_field != np.arange(self.Ndof)))
        L, nb = self.Ltau, self.Nbonds
        # cosht/sinht: Julia (Ltau x Nbonds) column-major == [bond][tau] here; bare values (:450-464)
        self._cs_stale = False
        self.cosht = np.zeros((nb, L))
        self.sinht = np.zeros((nb, L))
        self.t_bare_cb = np.zeros(nb)                                # bare hopping in checkerboard order
        self.t_bare_cb[self.checkerboard_perm - 1] = self.t
        self.expDtauMu = np.exp(self.dtau * self.mu)
        self._create(None, None, self._device)


# ----------------------------------------------------------------------------------------------
# update_model!
# ----------------------------------------------------------------------------------------------

def update_model_(model):
    """HolsteinModels.jl:526-549 (exp on the GPU) / SSHModels.jl:510-562 (cosh/sinh gather on the host, upload)."""
    model._nchains = 1
    if model.kind == HOLSTEIN:
        check(model._lib.elph_update_model_holstein(model._h, dptr(np.ascontiguousarray(model.x)), dptr(model.lam),
                                                    dptr(model.lam2), dptr(model.mu), model.dtau))
    else:
        # cosh/sinh of t' = t - (alpha x + sign(x) alpha2 x^2) and exp(dtau mu) are computed on the GPU (SSHModels.jl:510-562)
        model.expDtauMu = np.exp(model.dtau * model.mu)             # host-visible field only; the device computes its own
        cb_index = np.ascontiguousarray(model.checkerboard_perm[model.phonon_to_bond - 1], dtype=np.int64)
        t_ph = np.ascontiguousarray(model.t[model.phonon_to_bond - 1], dtype=np.float64)
        check(model._lib.elph_update_model_ssh_fields(
            model._h, dptr(np.ascontiguousarray(model.x)) if model.Nph else None, model.Nph, iptr(cb_index) if model.Nph else None,
            dptr(t_ph) if model.Nph else None, dptr(np.ascontiguousarray(model.alpha)) if model.Nph else None,
            dptr(np.ascontiguousarray(model.alpha2)) if model.Nph else None,
            dptr(model.t_bare_cb) if model.Nbonds else None, dptr(np.ascontiguousarray(model.mu)), model.dtau))
        model._cs_stale = True


def update_model_chains_(model, X):
    """Several independent phonon configurations (chains) in one handle: X is (nchains, Ndof); in a batched call
    right-hand side r then uses chain r % nchains (the reference runs chains as separate processes,
    ElPhDynamics.jl:90-95).  model.x is left untouched."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    assert X.ndim == 2 and X.shape[1] == model.Ndof
    if model.kind == SSH:
        cb_index = np.ascontiguousarray(model.checkerboard_perm[model.phonon_to_bond - 1], dtype=np.int64)
        t_ph = np.ascontiguousarray(model.t[model.phonon_to_bond - 1], dtype=np.float64)
        check(model._lib.elph_update_model_ssh_fields_chains(
            model._h, X.shape[0], dptr(X), model.Nph, iptr(cb_index), dptr(t_ph), dptr(np.ascontiguousarray(model.alpha)),
            dptr(np.ascontiguousarray(model.alpha2)), dptr(model.t_bare_cb), dptr(np.ascontiguousarray(model.mu)), model.dtau))
        model._cs_stale = True
        model._nchains = X.shape[0]
        return
    check(model._lib.elph_update_model_holstein_chains(model._h, X.shape[0], dptr(X), dptr(model.lam), dptr(model.lam2),
                                                       dptr(model.mu), model.dtau))
    model._nchains = X.shape[0]


# ----------------------------------------------------------------------------------------------
# mul! family
# ----------------------------------------------------------------------------------------------

def _vec(a, n):
    assert isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags["C_CONTIGUOUS"] and a.size == n, \
        f"expected a contiguous float64 vector of length {n}"
    return a


def mulM_(y, model, v):
    check(model._lib.elph_mulM(model._h, dptr(_vec(y, model.Ndim)), dptr(_vec(v, model.Ndim))))


def mulMt_(y, model, v):
    check(model._lib.elph_mulMT(model._h, dptr(_vec(y, model.Ndim)), dptr(_vec(v, model.Ndim))))


def mulMtM_(y, model, v):
    check(model._lib.elph_mulMTM(model._h, dptr(_vec(y, model.Ndim)), dptr(_vec(v, model.Ndim))))


def mulMMt_(y, model, v):
    """Models.jl:229-238 (two launches inside the library; not on the CG path, which always uses MtM)."""
    check(model._lib.elph_mulMMT(model._h, dptr(_vec(y, model.Ndim)), dptr(_vec(v, model.Ndim))))


def mul_(y, model, v):
    """Models.jl:192-209."""
    if model.mul_by_M:
        (mulMt_ if model.transposed else mulM_)(y, model, v)
    else:
        (mulMMt_ if model.transposed else mulMtM_)(y, model, v)


def transpose_(model):
    """Models.jl:244-248."""
    model.transposed = not model.transposed


def muldMdx_(dMdx, u, model, v):
    """muldMdx!(dMdx, u, model, v): dMdx[field] = uᵀ(∂M/∂x_field)v — HolsteinModels.jl:691-755 (dMdx: Ndim) / SSHModels.jl:707-829
    (dMdx: Ndof = Nph·Lτ, equivalent fields summed and copied as at :820-826).  The matrix is that of the last update_model_."""
    _vec(u, model.Ndim), _vec(v, model.Ndim)
    if model.kind == HOLSTEIN:
        check(model._lib.elph_muldMdx_holstein(model._h, dptr(_vec(dMdx, model.Ndim)), dptr(u), dptr(v), dptr(np.ascontiguousarray(model.x)),
                                               dptr(model.lam), dptr(model.lam2), model.dtau))
        return
    _vec(dMdx, model.Ndof)
    if model.Nph == 0:
        return
    raw = np.empty(model.Ndof)
    check(model._lib.elph_muldMdx_ssh_fields(model._h, dptr(raw), dptr(u), dptr(v)))
    pf = getattr(model, "primary_field", None)
    if pf is None:
        dMdx[:] = raw
    else:                                            # primary field of every field (SSHModels.jl:480-502), 0-based here
        acc = np.zeros(model.Ndof)
        np.add.at(acc, pf, raw)                      # dMdx[primary_field[field]] += dmdx  (:817)
        dMdx[:] = acc[pf]                            # @. dMdx = dMdx[primary_field]      (:826)


# ----------------------------------------------------------------------------------------------
# solvers
# ----------------------------------------------------------------------------------------------

def _use_prec(P):
    return 0 if P is None else 1


def solve_(x, model, b, cg=None, P=None, maxiter=0, tol=0.0, kmax=0.0, history=False):
    """solve!(x, A, b, cg[, P]; maxiter, tol, kmax) -> iters (IterativeSolvers.jl:153-234, 239-314).
    With history=True also returns eps_0..eps_iters."""
    cg = cg or model.solver
    model._push_solver()
    if P is not None:
        assert P.model is model
    mi = maxiter if maxiter else cg.maxiter
    it = C.c_int64()
    hist = np.full(mi + 1, np.nan) if history else None
    check(model._lib.elph_cg_solve(model._h, dptr(_vec(x, model.Ndim)), dptr(_vec(b, model.Ndim)), tol or cg.tol, mi,
                                   kmax or cg.kmax, _use_prec(P), C.byref(it), dptr(hist) if history else None))
    if history:
        return int(it.value), hist[:it.value + 1]
    return int(it.value)


def ldiv_(x, model, b, P=None, maxiter=0):
    """ldiv!(x, model, b[, P]; maxiter) -> (iters, residual_error, flag)  (Models.jl:74-186)."""
    assert not model.mul_by_M and not model.transposed, "the GPU path solves MtM x = b (CG), as the reference does"
    model._push_solver()
    it, fl, res = C.c_int64(), C.c_int(), C.c_double()
    check(model._lib.elph_ldiv(model._h, dptr(_vec(x, model.Ndim)), dptr(_vec(b, model.Ndim)), _use_prec(P), maxiter,
                               C.byref(it), C.byref(res), C.byref(fl)))
    return int(it.value), float(res.value), int(fl.value)


def ldiv_batched_(X, model, B, P=None, maxiter=0):
    """nrhs independent solves of the same matrix advanced together; X, B: (nrhs, Ndim) arrays."""
    model._push_solver()
    nrhs = B.shape[0]
    assert X.shape == B.shape == (nrhs, model.Ndim)
    it = np.zeros(nrhs, dtype=np.int64)
    res = np.zeros(nrhs)
    fl = np.zeros(nrhs, dtype=np.int32)
    check(model._lib.elph_ldiv_batched(model._h, nrhs, dptr(X), dptr(B), _use_prec(P), maxiter, iptr(it), dptr(res),
                                       fl.ctypes.data_as(_lib.P_int)))
    return it, res, fl
