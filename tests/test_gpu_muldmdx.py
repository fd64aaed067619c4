"""muldMdx!(dMdx, u, model, v) through the C ABI (elph_muldMdx_holstein / _ssh / _ssh_fields and the _dev twins) — the L3 operator the
reference's calc_dSfdx! calls at HMC.jl:799,804 and LangevinDynamics.jl:378 (HolsteinModels.jl:691-755, SSHModels.jl:707-829).

Checked against (i) the definition-level fixtures tests/golden/muldmdx_*.npz (complex-step derivative of uᵀMv on the dense M),
(ii) the CPU oracle at the BASELINE configurations b, B, C, D (Holstein) and e, E (bond phonons), (iii) the fused force
(elph_fermion_force_*) it must compose to, and (iv) a finite difference of the device's own uᵀ(M v).  Tolerance 1e-12 relative on the
operator (plain products, no solve inside)."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def lib():
    from elphdynamics_amd import _lib
    L = _lib.load()
    assert L.elph_device_count() >= 1, "no HIP device: the product has no CPU fallback"
    return L


def _create(lib, kind, N, L, table, c=None, s=None):
    from elphdynamics_amd import _lib
    h = _lib.Handle()
    tab = np.ascontiguousarray(table, dtype=np.int64)
    _lib.check(lib.elph_create(C.byref(h), kind, N, L, tab.shape[0], _lib.iptr(tab), _lib.dptr(np.ascontiguousarray(c)) if c is not None else None,
                               _lib.dptr(np.ascontiguousarray(s)) if s is not None else None, 0))
    return h


def test_muldMdx_holstein_golden(lib):
    from elphdynamics_amd import _lib
    g, d = golden("holstein_sq4_L8.npz"), golden("muldmdx_sq4_L8.npz")
    N, L, dtau = int(g["N"]), int(g["Ltau"]), float(g["dtau"])
    h = _create(lib, 0, N, L, g["table"], g["cosht"], g["sinht"])
    try:
        x, lam, lam2 = np.ascontiguousarray(g["x"]), np.ascontiguousarray(g["lam"]), np.ascontiguousarray(g["lam2"])
        u, v = np.ascontiguousarray(d["u"]), np.ascontiguousarray(d["v"])
        out = np.full(N * L, np.nan)
        # before update_model!: refused, not answered from an undefined matrix
        assert lib.elph_muldMdx_holstein(h, _lib.dptr(out), _lib.dptr(u), _lib.dptr(v), _lib.dptr(x), _lib.dptr(lam), _lib.dptr(lam2), dtau) == _lib.ELPH_E_STATE
        _lib.check(lib.elph_update_model_holstein(h, _lib.dptr(x), _lib.dptr(lam), _lib.dptr(lam2), _lib.dptr(np.ascontiguousarray(g["mu"])), dtau))
        _lib.check(lib.elph_muldMdx_holstein(h, _lib.dptr(out), _lib.dptr(u), _lib.dptr(v), _lib.dptr(x), _lib.dptr(lam), _lib.dptr(lam2), dtau))
        assert rel(out, d["dMdx"]) < 1e-13
        assert lib.elph_muldMdx_holstein(h, None, _lib.dptr(u), _lib.dptr(v), _lib.dptr(x), _lib.dptr(lam), _lib.dptr(lam2), dtau) == _lib.ELPH_E_ARG
        assert lib.elph_muldMdx_ssh(h, _lib.dptr(out), _lib.dptr(u), _lib.dptr(v)) == _lib.ELPH_E_ARG          # not an SSH handle
    finally:
        lib.elph_destroy(h)


def test_muldMdx_ssh_golden(lib):
    """Both SSH forms: the bond brackets (caller scatters, as julia/ElPhGPU.jl does after a host-side update_model!) and the
    device-side scatter onto the fields of elph_update_model_ssh_fields."""
    from elphdynamics_amd import _lib
    g, d = golden("ssh_sq4_L8_a.npz"), golden("muldmdx_ssh_sq4_L8_a.npz")
    N, L, dtau = int(g["N"]), int(g["Ltau"]), float(g["dtau"])
    nb, nph = g["table"].shape[0], g["phonon_to_bond"].shape[0]
    cbperm = g["cbperm"]
    u, v = np.ascontiguousarray(d["u"]), np.ascontiguousarray(d["v"])
    h = _create(lib, 1, N, L, g["table"])
    try:
        # host tables in (update_model! ran on the CPU): brackets out, scatter here exactly as SSHModels.jl:797-823
        _lib.check(lib.elph_update_model_ssh(h, _lib.dptr(np.ascontiguousarray(g["cosht"])), _lib.dptr(np.ascontiguousarray(g["sinht"])),
                                             _lib.dptr(np.ascontiguousarray(g["expDtauMu"]))))
        q = np.zeros(nb * L)
        _lib.check(lib.elph_muldMdx_ssh(h, _lib.dptr(q), _lib.dptr(u), _lib.dptr(v)))
        q = q.reshape(nb, L)
        X = g["x"].reshape(nph, L)
        sg = np.ones(L); sg[0] = -1.0
        out = np.zeros((nph, L))
        for p in range(nph):
            n = cbperm[g["phonon_to_bond"][p] - 1] - 1
            out[p] = sg * dtau * (g["alpha"][p] + 2 * g["alpha2"][p] * X[p]) * q[n]
        assert rel(out.reshape(-1), d["dMdx"]) < 1e-12
        out2 = np.zeros(nph * L)
        assert lib.elph_muldMdx_ssh_fields(h, _lib.dptr(out2), _lib.dptr(u), _lib.dptr(v)) == _lib.ELPH_E_STATE      # no device-side fields yet
        # device-side update + scatter
        cb_index = np.ascontiguousarray(cbperm[g["phonon_to_bond"] - 1], dtype=np.int64)
        t_cb = np.zeros(nb); t_cb[cbperm - 1] = g["t"]
        _lib.check(lib.elph_update_model_ssh_fields(h, _lib.dptr(np.ascontiguousarray(g["x"])), nph, _lib.iptr(cb_index),
                                                    _lib.dptr(np.ascontiguousarray(g["t"][g["phonon_to_bond"] - 1])), _lib.dptr(np.ascontiguousarray(g["alpha"])),
                                                    _lib.dptr(np.ascontiguousarray(g["alpha2"])), _lib.dptr(t_cb), _lib.dptr(np.ascontiguousarray(g["mu"])), dtau))
        _lib.check(lib.elph_muldMdx_ssh_fields(h, _lib.dptr(out2), _lib.dptr(u), _lib.dptr(v)))
        assert rel(out2, d["dMdx"]) < 1e-12
    finally:
        lib.elph_destroy(h)


@pytest.mark.parametrize("tag", ["b", "B", "C", "D", "d", "g"])
def test_muldMdx_holstein_vs_oracle(oracle, tag):
    from elphdynamics_amd import configs, models, synth
    from oracle.oracle import dp
    m = configs.make_model(tag, tol=1e-8)
    m.lam2[:] = 0.03 * synth.randn(5, m.Nsites)
    models.update_model_(m)
    E = oracle.update_model_holstein(m.Nsites, m.Ltau, m.dtau, m.x, m.lam, m.lam2, m.mu)
    om = oracle.make_model(0, m.Nsites, m.Ltau, m.neighbor_table, m.cosht, m.sinht, E)
    u, v = synth.randn(911, m.Ndim), synth.randn(912, m.Ndim)
    ref = np.zeros(m.Ndim)
    oracle.lib.elpho_muldMdx_holstein(dp(ref), dp(u), C.byref(om), dp(v), m.dtau, dp(m.lam), dp(m.lam2), dp(np.ascontiguousarray(m.x)))
    out = np.zeros(m.Ndim)
    models.muldMdx_(out, u, m, v)
    assert rel(out, ref) < 1e-12
    # composition: calc_dSfdx! = −Σ± muldMdx!(M X±, X±) + muldΛdx! terms (HMC.jl:797-811) — against the fused device force with Λ-terms
    # removed by taking λ-independent ϕ: checked in test_gpu_parity.py::test_fermion_force_vs_oracle; here the operator identity
    # uᵀ(∂M/∂x_f)v = d/dε uᵀ M(x + ε e_f) v on the device's own mat-vec
    x0 = m.x.copy()
    eps = 1e-6
    Mv = np.zeros(m.Ndim)
    for f in (0, m.Ltau - 1, m.Ltau, m.Ndim // 2 + 3, m.Ndim - 1):
        s = []
        for sgn in (+1, -1):
            m.x[:] = x0; m.x[f] += sgn * eps
            models.update_model_(m)
            models.mulM_(Mv, m, v)
            s.append(u @ Mv)
        fd = (s[0] - s[1]) / (2 * eps)
        assert abs(fd - out[f]) < 1e-7 * max(1.0, abs(out[f])), (tag, f, fd, out[f])
    m.x[:] = x0
    m.close()


@pytest.mark.parametrize("tag", ["e", "E"])
def test_muldMdx_ssh_vs_oracle(oracle, tag):
    from elphdynamics_amd import configs, models, synth
    from oracle.oracle import dp, ip
    m = configs.make_model(tag, tol=1e-8)
    m.alpha2[:] = 0.02
    models.update_model_(m)
    om = oracle.make_model(1, m.Nsites, m.Ltau, m.neighbor_table, np.ascontiguousarray(m.cosht).reshape(-1),
                           np.ascontiguousarray(m.sinht).reshape(-1), m.expDtauMu)
    b2p = np.zeros(m.Nbonds, dtype=np.int64)
    b2p[m.checkerboard_perm[m.phonon_to_bond - 1] - 1] = np.arange(1, m.Nph + 1)
    u, v = synth.randn(913, m.Ndim), synth.randn(914, m.Ndim)
    ref = np.zeros(m.Ndof)
    oracle.lib.elpho_muldMdx_ssh(dp(ref), dp(u), C.byref(om), dp(v), m.dtau, ip(b2p), dp(m.alpha), dp(m.alpha2), dp(np.ascontiguousarray(m.x)), m.Nph)
    out = np.zeros(m.Ndof)
    models.muldMdx_(out, u, m, v)
    assert rel(out, ref) < 1e-12
    m.close()


def test_muldMdx_ssh_sums_equivalent_fields():
    """Phonon types of one name share their fields: dMdx[primary_field[f]] += …, then dMdx = dMdx[primary_field] (SSHModels.jl:817-826)."""
    from elphdynamics_amd import configs, models, synth
    m = configs.make_model("e", tol=1e-8)
    models.update_model_(m)
    u, v = synth.randn(915, m.Ndim), synth.randn(916, m.Ndim)
    raw = np.zeros(m.Ndof)
    models.muldMdx_(raw, u, m, v)
    L = m.Ltau
    pf = np.arange(m.Ndof, dtype=np.int64)
    pf[L:2 * L] = np.arange(L)                                   # phonon 2 shares the fields of phonon 1
    m.primary_field = pf
    out = np.zeros(m.Ndof)
    models.muldMdx_(out, u, m, v)
    assert np.allclose(out[:L], raw[:L] + raw[L:2 * L], rtol=1e-15, atol=0) and np.array_equal(out[L:2 * L], out[:L])
    assert np.array_equal(out[2 * L:], raw[2 * L:])
    m.close()


def test_muldMdx_composes_to_the_fused_fermion_force(oracle):
    """calc_dSfdx! (HMC.jl:790-814) written with the operator — dSfdx −= muldMdx!(M X±, X±) + the muldΛdx! terms — equals the fused
    elph_fermion_force_holstein the device-resident HMC uses."""
    from elphdynamics_amd import configs, hmc, models, synth
    from oracle.oracle import dp
    m = configs.make_model("B", tol=1e-12)
    m.lam2[:] = 0.03 * synth.randn(5, m.Nsites)
    phi_p, phi_m = synth.randn(41, m.Ndim), synth.randn(42, m.Ndim)
    F = np.zeros(m.Ndim)
    it, fl, Xp, Xm = hmc.calc_dSfdx_(F, m, phi_p, phi_m, None, power=1.0, return_solutions=True)
    assert fl == 0
    N, L = m.Nsites, m.Ltau
    Lam = np.zeros(m.Ndim)
    oracle.lib.elpho_update_Lambda(dp(Lam), N, L, m.dtau, dp(np.ascontiguousarray(m.x)), dp(m.lam), dp(m.lam2))
    G = np.zeros(m.Ndim)
    MX, d = np.zeros(m.Ndim), np.zeros(m.Ndim)
    for X, phi in ((Xp, phi_p), (Xm, phi_m)):
        models.mulM_(MX, m, np.ascontiguousarray(X))
        models.muldMdx_(d, MX, m, np.ascontiguousarray(X))                 # HMC.jl:798-799
        G -= d                                                             # :803
        oracle.lib.elpho_muldLambdadx_holstein(dp(G), dp(np.ascontiguousarray(phi)), dp(np.ascontiguousarray(X)), dp(Lam), N, L, m.dtau,
                                               dp(m.lam), dp(m.lam2), dp(np.ascontiguousarray(m.x)))      # :806-809, accumulates
    assert rel(G, F) < 1e-10
    m.close()


class _DevBuf:
    """A device buffer through the HIP runtime the library itself links (ctypes on libamdhip64: no torch in the process — its bundled
    runtime is a second HIP in the same address space)."""
    hip = None

    def __init__(self, host=None, n=None):
        if _DevBuf.hip is None:
            _DevBuf.hip = C.CDLL("libamdhip64.so")
        self.n = int(n if host is None else host.size)
        self.p = C.c_void_p()
        assert _DevBuf.hip.hipMalloc(C.byref(self.p), C.c_size_t(8 * self.n)) == 0
        if host is not None:
            a = np.ascontiguousarray(host, dtype=np.float64)
            assert _DevBuf.hip.hipMemcpy(self.p, a.ctypes.data_as(C.c_void_p), C.c_size_t(8 * self.n), 1) == 0      # hipMemcpyHostToDevice

    def get(self):
        out = np.empty(self.n)
        assert _DevBuf.hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), self.p, C.c_size_t(8 * self.n), 2) == 0        # hipMemcpyDeviceToHost
        return out

    def free(self):
        _DevBuf.hip.hipFree(self.p)


def test_muldMdx_device_pointer_twins(lib):
    """_dev entry points on device buffers (reference layout) give the bits of the host entry points."""
    from elphdynamics_amd import _lib, configs, models, synth
    m = configs.make_model("B", tol=1e-8)
    models.update_model_(m)
    u, v = synth.randn(921, m.Ndim), synth.randn(922, m.Ndim)
    ref = np.zeros(m.Ndim)
    models.muldMdx_(ref, u, m, v)
    du, dv, dx, dout = _DevBuf(u), _DevBuf(v), _DevBuf(m.x), _DevBuf(n=m.Ndim)
    _lib.check(lib.elph_muldMdx_holstein_dev(m._h, dout.p, du.p, dv.p, dx.p, _lib.dptr(m.lam), _lib.dptr(m.lam2), m.dtau))
    _lib.check(lib.elph_synchronize(m._h))
    assert np.array_equal(dout.get(), ref)
    # the mat-vec twins on the same buffers (never driven with caller-owned device memory before)
    y = np.zeros(m.Ndim)
    models.mulM_(y, m, v)
    _lib.check(lib.elph_mulM_dev(m._h, dout.p, dv.p))
    _lib.check(lib.elph_synchronize(m._h))
    assert np.array_equal(dout.get(), y)
    for b in (du, dv, dx, dout):
        b.free()
    m.close()
    ms = configs.make_model("e", tol=1e-8)
    models.update_model_(ms)
    refs = np.zeros(ms.Ndof)
    us, vs = synth.randn(923, ms.Ndim), synth.randn(924, ms.Ndim)
    models.muldMdx_(refs, us, ms, vs)
    du, dv, dout = _DevBuf(us), _DevBuf(vs), _DevBuf(n=ms.Ndof)
    _lib.check(lib.elph_muldMdx_ssh_fields_dev(ms._h, dout.p, du.p, dv.p))
    _lib.check(lib.elph_synchronize(ms._h))
    assert np.array_equal(dout.get(), refs)
    for b in (du, dv, dout):
        b.free()
    ms.close()
