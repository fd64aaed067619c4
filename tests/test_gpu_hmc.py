"""GPU (-m gpu): the device-resident HMC update (include/elph_gpu.h: elph_hmc_*) against the dense golden
trajectory and against the CPU oracle on the same inputs (same random numbers handed to both).

Tolerances: the force solves stop at the solver tolerance (1e-6 here; the two action evaluations at tol^2 = 1e-12).
Two correct implementations whose stop test trips one iteration apart in one of the 2(nt+2) solves differ by that
tolerance in the force, so end points agree to a few 1e-7 relative at worst (observed 2e-9 .. 2e-8; bound written
below: 3e-7), H0 (tol^2 solves) to 1e-9, ΔH to 1e-7.  Against the exact-solve golden the bound is 1e-6 at tol 1e-7."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def _raw_update(lib, h, dt, nt, nb, alpha, use_prec, R, Rp, Rm, kpm, u):
    from elphdynamics_amd._lib import check, dptr
    acc, fl, its, en = C.c_int(), C.c_int(), C.c_double(), np.zeros(5)
    check(lib.elph_hmc_update(h, dt, nt, nb, alpha, use_prec, dptr(np.ascontiguousarray(R)), dptr(np.ascontiguousarray(Rp)),
                              dptr(np.ascontiguousarray(Rm)), dptr(np.ascontiguousarray(kpm)) if kpm is not None else None,
                              u, C.byref(acc), C.byref(its), dptr(en), C.byref(fl)))
    return bool(acc.value), its.value, en, fl.value


@pytest.mark.parametrize("nb", [1, 3])
def test_hmc_golden_through_the_raw_abi(nb):
    """What a Julia ccall wrapper would do: elph_create from the reference-layout tables, elph_hmc_create,
    elph_hmc_set_state, elph_hmc_update, elph_hmc_get_state — against the dense numpy trajectory."""
    from elphdynamics_amd import _lib
    from elphdynamics_amd._lib import check, dptr, iptr
    lib = _lib.load()
    g, hgold = golden(f"hmc_sq4_L8_nb{nb}.npz"), golden("holstein_sq4_L8.npz")
    N, L, dtau = int(g["N"]), int(g["Ltau"]), float(g["dtau"])
    h = _lib.Handle()
    tab = np.ascontiguousarray(hgold["table"], dtype=np.int64)
    check(lib.elph_create(C.byref(h), 0, N, L, tab.shape[0], iptr(tab), dptr(np.ascontiguousarray(hgold["cosht"])),
                          dptr(np.ascontiguousarray(hgold["sinht"])), 0))
    try:
        check(lib.elph_solver_set(h, 1e-7, 20000, 1e12))
        arrs = [np.ascontiguousarray(g[k]) for k in ("omega", "omega4", "lam", "lam2", "mu")]
        check(lib.elph_hmc_create(h, *(dptr(a) for a in arrs), dtau, dptr(np.ascontiguousarray(g["faM"]))))
        x, v = g["x0"].copy(), np.zeros(N * L)
        check(lib.elph_hmc_set_state(h, dptr(x), dptr(v)))
        acc, its, en, fl = _raw_update(lib, h, float(g["dt"]), int(g["nt"]), nb, 0.0, 0, g["R"], g["Rp"], g["Rm"], None, 0.0)
        check(lib.elph_hmc_get_state(h, dptr(x), dptr(v)))
        assert acc and fl == 0 and its > 0
        assert abs(en[0] - float(g["H0"])) < 1e-9 * abs(float(g["H0"]))
        assert abs(en[0] - float(g["H0_closed"])) < 1e-9 * abs(float(g["H0"]))
        assert abs(en[1] - float(g["H1"])) < 1e-6
        assert rel(x, g["x1"]) < 1e-6 and rel(v, g["v1"]) < 1e-6
    finally:
        lib.elph_destroy(h)


def _pair(oracle, tag, tol, lam2=0.0, seed=0):
    from elphdynamics_amd import configs, preconditioners as pc, synth
    m = configs.make_model(tag, tol=tol, maxiter=20000)
    m.omega4[:] = 0.02
    if lam2:
        m.lam2[:] = lam2
    fa = pc.FourierAccelerator(m)
    pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.3)
    E = oracle.update_model_holstein(m.Nsites, m.Ltau, m.dtau, m.x, m.lam, m.lam2, m.mu)
    om = oracle.make_model(0, m.Nsites, m.Ltau, m.neighbor_table, m.cosht, m.sinht, E)
    return m, fa, om


def _randoms(m, nt, seed, with_kpm, u):
    from elphdynamics_amd import synth
    return dict(R=synth.randn(seed, m.Ndof), Rp=synth.randn(seed + 1, m.Ndim), Rm=synth.randn(seed + 2, m.Ndim),
                kpm_randn=synth.randn(seed + 3, (nt + 2) * 2 * m.Nsites) if with_kpm else None, u=u)


def _oracle_update(oracle, om, m, fa, x, v, dt, nt, nb, alpha, rnd, P=None):
    return oracle.hmc_update_holstein(om, x, v, m.omega, m.omega4, m.lam, m.lam2, m.mu, m.dtau, fa.M, dt, nt, nb, alpha, rnd,
                                      P=P, tol=m.solver.tol, maxiter=m.solver.maxiter)


@pytest.mark.parametrize("tag,nb,nt", [("b", 1, 5), ("b", 3, 4), ("d", 1, 4), ("t", 1, 3), ("B", 1, 3), ("C", 1, 2), ("g", 1, 2),
                                       ("D", 1, 2),        # BASELINE config 4: "Holstein HMC honeycomb L=12 Ntau=120"
                                       ("q", 1, 2), ("z", 1, 2), ("r", 1, 2),      # the GRID / HGRID forms (L = 10 square, 10 x 10 honeycomb cells, 12 x 6 rectangle)
                                       ("t12", 1, 2), ("h", 1, 2), ("k", 1, 2)])   # the PGRID kernels (12 x 12 triangular, 18 x 18 honeycomb cells, 20 x 20 square)
def test_hmc_update_vs_oracle(oracle, tag, nb, nt):
    from elphdynamics_amd import hmc
    m, fa, om = _pair(oracle, tag, tol=1e-6, lam2=0.02)
    dt = 0.05
    H = hmc.HybridMonteCarlo(m, fa, dt, nt * dt, alpha=0.0, Nb=nb)
    assert H.Nt == nt
    rnd = _randoms(m, nt, 500, False, 0.0)
    x0 = m.x.copy()
    acc_o, x_o, v_o, info = _oracle_update(oracle, om, m, fa, x0, np.zeros(m.Ndof), dt, nt, nb, 0.0, rnd)
    acc, its = hmc.update_(m, H, fa, None, randoms=rnd)
    assert acc == acc_o and H.flag == info["flag"] == 0
    assert abs(H.H0 - info["H0"]) < 1e-9 * abs(info["H0"]) and abs(H.H1 - info["H1"]) < 1e-8 * abs(info["H1"])
    # ΔH inherits the force-solve tolerance through the end point: absolute, grows with the system size
    assert abs((H.H1 - H.H0) - (info["H1"] - info["H0"])) < max(5e-7, 1e-10 * abs(info["H0"]))
    assert rel(m.x, x_o) < 3e-7 and rel(H.v, v_o) < 3e-7
    assert abs(its - info["iters"]) <= 1                          # cld(iters, Nt+2): knife-edge stops can move it by one
    assert abs(H.K - info["K"]) < 1e-8 * abs(info["K"]) and abs(H.S - info["S"]) < 1e-8 * abs(info["S"])
    m.close()


def test_hmc_two_updates_stay_on_the_device(oracle):
    """State is resident between updates (no pull/push in between), partial momentum refresh uses the previous v."""
    from elphdynamics_amd import hmc
    m, fa, om = _pair(oracle, "b", tol=1e-7)
    dt, nt, alpha = 0.04, 4, 0.4
    H = hmc.HybridMonteCarlo(m, fa, dt, nt * dt, alpha=alpha, Nb=1)
    x, v = m.x.copy(), np.zeros(m.Ndof)
    accs = []
    for k, u in enumerate((0.0, 1.0, 0.0)):                       # accept, forced reject, accept
        rnd = _randoms(m, nt, 700 + 10 * k, False, u)
        acc_o, x, v, info = _oracle_update(oracle, om, m, fa, x, v, dt, nt, 1, alpha, rnd)
        acc, _ = hmc.update_(m, H, fa, None, randoms=rnd, pull=False)
        assert acc == acc_o
        accs.append(acc)
    assert accs == [True, False, True]
    H.pull_()
    assert rel(m.x, x) < 3e-7 and rel(H.v, v) < 3e-7
    m.close()


def test_hmc_reject_restores_bit_exactly_and_failed_solve_kills(oracle):
    from elphdynamics_amd import hmc, models
    m, fa, om = _pair(oracle, "b", tol=1e-6)
    dt, nt = 0.05, 3
    H = hmc.HybridMonteCarlo(m, fa, dt, nt * dt)
    x0 = m.x.copy()
    rnd = _randoms(m, nt, 900, False, 1.0)
    acc, _ = hmc.update_(m, H, fa, None, randoms=rnd)
    assert not acc and H.flag == 0 and 0.0 < H.P_accept <= 1.0
    assert np.array_equal(m.x, x0)                                 # copyto!(x, x0): bit-exact
    # exp(-dtau V) was rebuilt for the restored field: a mat-vec equals the one of a fresh model
    v = np.linspace(-1, 1, m.Ndim)
    y1, y2 = np.empty(m.Ndim), np.empty(m.Ndim)
    models.mulM_(y1, m, v)
    models.update_model_(m)
    models.mulM_(y2, m, v)
    assert rel(y1, y2) < 1e-15
    # a solve that cannot converge (maxiter = 3) kills the trajectory: flag 1, rejected, x restored
    m.solver.maxiter = 3
    acc, _ = hmc.update_(m, H, fa, None, randoms=_randoms(m, nt, 901, False, 0.0))
    assert not acc and H.flag == 1 and H.P_accept == 0.0 and np.array_equal(m.x, x0)
    m.close()


@pytest.mark.parametrize("tag", ["b", "C"])
def test_hmc_update_with_kpm_preconditioner(oracle, tag):
    """setup!(P) once per force evaluation from the caller's Arnoldi start vectors; same end point as the oracle's
    preconditioned trajectory (the two Arnoldi implementations agree to ~1e-7 on the bounds, which moves nothing by
    more than the solver tolerance)."""
    from elphdynamics_amd import hmc, preconditioners as pc
    m, fa, om = _pair(oracle, tag, tol=1e-7)
    dt, nt = 0.05, 2
    H = hmc.HybridMonteCarlo(m, fa, dt, nt * dt)
    P = pc.SymmetricKPMPreconditioner(m, n=min(20, m.Nsites), buf=0.05, c1=1.0, c2=1.0)
    Po = oracle.make_kpm(om, n=min(20, m.Nsites))
    rnd = _randoms(m, nt, 1100, True, 0.0)
    x0 = m.x.copy()
    acc_o, x_o, v_o, info = _oracle_update(oracle, om, m, fa, x0, np.zeros(m.Ndof), dt, nt, 1, 0.0, rnd, P=Po)
    acc, its = hmc.update_(m, H, fa, P, randoms=rnd)
    assert acc == acc_o and H.flag == 0 and info["kpm_calls"] == nt + 2
    assert abs(H.H0 - info["H0"]) < 1e-9 * abs(info["H0"])
    assert rel(m.x, x_o) < 1e-6 and rel(H.v, v_o) < 1e-6
    # and the unpreconditioned trajectory ends in the same place to solver tolerance
    m2, fa2, om2 = _pair(oracle, tag, tol=1e-7)
    H2 = hmc.HybridMonteCarlo(m2, fa2, dt, nt * dt)
    hmc.update_(m2, H2, fa2, None, randoms=rnd)
    assert rel(m.x, m2.x) < 1e-6 and its < H2.iters
    m.close()
    m2.close()


def test_hmc_energy_error_scales_as_dt_squared():
    """Size-independent property at the full config-C size: |ΔH| drops 4x when dt halves (same trajectory length)."""
    from elphdynamics_amd import configs, hmc, preconditioners as pc
    dH = []
    for dt, nt in ((0.02, 2), (0.01, 4)):
        m = configs.make_model("C", tol=1e-8, maxiter=40000)
        fa = pc.FourierAccelerator(m)
        pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.3)
        H = hmc.HybridMonteCarlo(m, fa, dt, nt * dt)
        hmc.update_(m, H, fa, None, randoms=_randoms(m, nt, 1300, False, 0.0))
        assert H.flag == 0
        dH.append(H.H1 - H.H0)
        m.close()
    assert 3.0 < dH[0] / dH[1] < 5.0, dH


def test_hmc_error_paths():
    from elphdynamics_amd import _lib, configs
    from elphdynamics_amd._lib import dptr
    lib = _lib.load()
    m = configs.make_model("b")
    z = np.zeros(m.Ndim)
    acc = C.c_int()
    rc = lib.elph_hmc_update(m._h, 0.1, 2, 1, 0.0, 0, dptr(z), dptr(z), dptr(z), None, 0.0, C.byref(acc), None, None, None)
    assert rc == _lib.ELPH_E_STATE                                  # elph_hmc_create missing
    one = np.ones(m.Nsites)
    _lib.check(lib.elph_hmc_create(m._h, dptr(one), dptr(one), dptr(one), dptr(one), dptr(one), m.dtau, dptr(np.ones(m.Ndim))))
    rc = lib.elph_hmc_update(m._h, 0.1, 2, 1, 0.0, 0, dptr(z), dptr(z), dptr(z), None, 0.0, C.byref(acc), None, None, None)
    assert rc == _lib.ELPH_E_STATE                                  # no state uploaded
    _lib.check(lib.elph_hmc_set_state(m._h, dptr(z), None))
    rc = lib.elph_hmc_update(m._h, 0.1, 2, 1, 1.0, 0, dptr(z), dptr(z), dptr(z), None, 0.0, C.byref(acc), None, None, None)
    assert rc == _lib.ELPH_E_ARG                                    # alpha must be in [0, 1)   (HMC.jl:182)
    rc = lib.elph_hmc_update(m._h, 0.1, 2, 1, 0.0, 1, dptr(z), dptr(z), dptr(z), None, 0.0, C.byref(acc), None, None, None)
    assert rc == _lib.ELPH_E_ARG                                    # preconditioner without start vectors
    m.close()
    e = configs.make_model("e")
    rc = lib.elph_hmc_create(e._h, dptr(one), dptr(one), dptr(one), dptr(one), dptr(one), e.dtau, dptr(np.ones(e.Ndim)))
    assert rc == _lib.ELPH_E_UNSUPPORTED                            # SSH: not built
    e.close()


@pytest.mark.parametrize("tag,nch,nb,with_kpm", [("b", 3, 1, False), ("B", 4, 3, True), ("d", 2, 1, True), ("q", 3, 1, True), ("y", 2, 1, True)])
def test_hmc_chains_in_lockstep_equal_single_chain_updates(tag, nch, nb, with_kpm):
    """elph_hmc_update_chains: nch Markov chains advanced in lockstep by one handle (all 2*nch pseudofermion solves of an
    evaluation as one batch, one KPM expansion per chain) — every chain ends where the single-chain update ends when it
    is given the same field, momenta and random numbers; accept/reject is taken per chain."""
    from elphdynamics_amd import configs, hmc, preconditioners as pc, synth
    nt, dt = 4, 0.05
    m = configs.make_model(tag, tol=1e-9, maxiter=20000)
    m.omega4[:] = 0.02
    fa = pc.FourierAccelerator(m)
    pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.3)
    X0 = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=900 + c) for c in range(nch)])
    V0 = np.stack([0.3 * synth.randn(950 + c, m.Ndof) for c in range(nch)])
    rnd = dict(R=np.stack([synth.randn(1000 + c, m.Ndof) for c in range(nch)]),
               Rp=np.stack([synth.randn(1100 + c, m.Ndim) for c in range(nch)]),
               Rm=np.stack([synth.randn(1200 + c, m.Ndim) for c in range(nch)]),
               kpm_randn=synth.randn(1300, (nt + 2) * 2 * nch * m.Nsites).reshape(nt + 2, 2, nch, m.Nsites) if with_kpm else None,
               u=np.array([0.0 if c % 2 == 0 else 1.5 for c in range(nch)]))     # even chains accept; u > 1 forces the odd ones to reject
    H = hmc.HybridMonteCarlo(m, fa, dt=dt, tr=nt * dt, alpha=0.2, Nb=nb, nchains=nch)
    H.X[:], H.V[:] = X0, V0
    H.push_()
    P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0) if with_kpm else None
    acc, its = hmc.update_chains_(m, H, fa, P, randoms=rnd, pull=True)
    assert not H.flags.any()
    Xb, Vb, Eb = H.X.copy(), H.V.copy(), H.energies.copy()
    m.close()
    for c in range(nch):
        m1 = configs.make_model(tag, tol=1e-9, maxiter=20000)
        m1.omega4[:] = 0.02
        fa1 = pc.FourierAccelerator(m1)
        pc.update_M_(fa1, m1, 0.0, np.inf, 1.0, 0.3)
        m1.x[:] = X0[c]
        H1 = hmc.HybridMonteCarlo(m1, fa1, dt=dt, tr=nt * dt, alpha=0.2, Nb=nb)
        H1.v[:] = V0[c]
        H1.push_()
        P1 = pc.SymmetricKPMPreconditioner(m1, 20, 0.05, 1.0, 1.0) if with_kpm else None
        r1 = dict(R=rnd["R"][c], Rp=rnd["Rp"][c], Rm=rnd["Rm"][c],
                  kpm_randn=np.ascontiguousarray(rnd["kpm_randn"][:, :, c, :]) if with_kpm else None, u=float(rnd["u"][c]))
        a1, i1 = hmc.update_(m1, H1, fa1, P1, randoms=r1)
        assert a1 == bool(acc[c]) and H1.flag == 0
        assert abs(i1 - its[c]) <= 1
        assert abs(H1.H0 - Eb[c, 0]) < 1e-8 * abs(H1.H0) and abs(H1.H1 - Eb[c, 1]) < 1e-8 * abs(H1.H1)
        assert np.abs(m1.x - Xb[c]).max() < 1e-7 * np.abs(m1.x).max() and np.abs(H1.v - Vb[c]).max() < 1e-7 * max(np.abs(H1.v).max(), 1.0)
        if not a1:      # rejected: field restored bit-exactly, momenta flipped (HMC.jl:447-456)
            assert np.array_equal(Xb[c], X0[c])
        m1.close()
    assert acc[0] and (nch < 2 or not acc[1])


# ---------------------------------------------------------------------------------------------- SSH (bond phonons)

def _ssh_golden_model(nb, shared=False):
    from elphdynamics_amd import lattice as lat
    from elphdynamics_amd import models
    g, hgold = golden(f"hmc_ssh_sq4_L8_a_nb{nb}{'_shared' if shared else ''}.npz"), golden("ssh_sq4_L8_a.npz")
    la = lat.Lattice(1, 4, 4, 1)
    L, dtau = int(g["Ltau"]), float(g["dtau"])
    m = models.SSHModel(la, L * dtau, dtau, tol=1e-7, maxiter=20000)
    for (o1, o2, d), name in zip(lat.SQUARE_BONDS, ("", "") if shared else ("x", "y")):
        m.assign_hopping_(1.0, 0.1, 0.0, 0.5, o1, o2, d, name=name)
    m.initialize_model_()
    assert np.array_equal(m.neighbor_table, hgold["table"]) and np.array_equal(m.checkerboard_perm, hgold["cbperm"])
    m.alpha[:], m.alpha2[:], m.mu[:] = hgold["alpha"], hgold["alpha2"], hgold["mu"]
    m.omega = np.array(g["omega"])
    m.omega4 = np.array(g["omega4"])
    m.x[:] = g["x0"]
    models.update_model_(m)
    return g, hgold, m


@pytest.mark.parametrize("nb", [1, 3])
def test_ssh_hmc_update_matches_dense_golden(nb):
    """elph_hmc_create_ssh + elph_hmc_update on the SSH model: the whole trajectory (device-side update_model! of the bond
    hoppings every step, Λ ≡ 1, bond-phonon force, multi-timestep variant) against the dense complex-step golden."""
    from elphdynamics_amd import hmc, preconditioners as pc
    g, hgold, m = _ssh_golden_model(nb)
    fa = pc.FourierAccelerator(m)
    fa.M[:] = g["faM"]
    H = hmc.HybridMonteCarlo(m, fa, float(g["dt"]), int(g["nt"]) * float(g["dt"]), alpha=0.0, Nb=nb)
    rnd = dict(R=g["R"], Rp=g["Rp"], Rm=g["Rm"], kpm_randn=None, u=0.0)
    acc, its = hmc.update_(m, H, fa, None, randoms=rnd)
    assert acc and H.flag == 0
    assert abs(H.H0 - float(g["H0"])) < 1e-9 * abs(float(g["H0"])) and abs(H.H0 - float(g["H0_closed"])) < 1e-9 * abs(float(g["H0"]))
    assert abs(H.H1 - float(g["H1"])) < 1e-6
    assert rel(m.x, g["x1"]) < 1e-6 and rel(H.v, g["v1"]) < 1e-6
    # the hopping tables on the device belong to the final field
    X = m.x.reshape(m.Nph, m.Ltau)
    idx = m.checkerboard_perm[m.phonon_to_bond - 1] - 1
    c = np.zeros((m.Nbonds, m.Ltau))
    c[idx] = np.cosh(m.dtau * (m.t[m.phonon_to_bond - 1][:, None] - m.alpha[:, None] * X))
    assert rel(m.cosht, c) < 1e-14
    m.close()


@pytest.mark.parametrize("nb", [1, 3])
def test_ssh_hmc_shared_fields_match_dense_golden(nb):
    """Two phonon types of the same name share their fields (primary_field, SSHModels.jl:480-502): class-summed fermion force,
    Sb and K over primary fields — against the dense golden trajectory of the independent variables alone."""
    from elphdynamics_amd import hmc, preconditioners as pc
    g, hgold, m = _ssh_golden_model(nb, shared=True)
    assert m.has_shared_fields and np.array_equal(m.primary_field[::m.Ltau] // m.Ltau, g["primary_column"])
    fa = pc.FourierAccelerator(m)
    fa.M[:] = g["faM"]
    H = hmc.HybridMonteCarlo(m, fa, float(g["dt"]), int(g["nt"]) * float(g["dt"]), alpha=0.0, Nb=nb)
    rnd = dict(R=g["R"], Rp=g["Rp"], Rm=g["Rm"], kpm_randn=None, u=0.0)
    acc, its = hmc.update_(m, H, fa, None, randoms=rnd)
    assert acc and H.flag == 0
    assert abs(H.H0 - float(g["H0"])) < 1e-9 * abs(float(g["H0"])) and abs(H.H0 - float(g["H0_closed"])) < 1e-9 * abs(float(g["H0"]))
    assert abs(H.H1 - float(g["H1"])) < 1e-6
    assert rel(m.x, g["x1"]) < 1e-6 and rel(H.v, g["v1"]) < 1e-6
    half = m.Ndof // 2
    assert np.array_equal(m.x[:half], m.x[half:]) and np.array_equal(H.v[:half], H.v[half:])      # the classes stay together exactly
    # fields that differ from their primary are refused, like update_model! (SSHModels.jl:549-559)
    m.x[half] += 1e-3
    with pytest.raises(Exception, match="primary"):
        H.push_()
    m.x[half] -= 1e-3
    H.push_()
    # generated momenta respect the sharing too
    H.device_rng_(5)
    acc, its = hmc.update_(m, H, fa, None)
    assert H.flag == 0 and np.array_equal(m.x[:half], m.x[half:]) and np.array_equal(H.v[:half], H.v[half:])
    with pytest.raises(Exception):
        hmc.special_move_(m, H, hmc.SWAP, 0, 1)
    m.close()


@pytest.mark.parametrize("tag,with_kpm,nb", [("e", False, 1), ("e", True, 2), ("E", True, 1), ("e12", True, 1), ("e24", True, 1)])
def test_ssh_hmc_update_vs_oracle(oracle, tag, with_kpm, nb):
    """Configs e / E (optical SSH square L = 4 / 16): device trajectory vs the oracle's, with alpha2 != 0 and the KPM
    preconditioner (tau-averaged cosh/sinh from the device tables), accept and reject."""
    from elphdynamics_amd import configs, hmc, preconditioners as pc, synth
    m = configs.make_model(tag, tol=1e-7, maxiter=20000)
    m.alpha2[:] = 0.01
    m.omega4 = np.full(m.Nph, 0.02)
    models_update = __import__("elphdynamics_amd.models", fromlist=["update_model_"]).update_model_
    models_update(m)
    fa = pc.FourierAccelerator(m)
    pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.3)
    nt, dt = 2, 0.05
    x0 = m.x.copy()
    om = oracle.make_model(1, m.Nsites, m.Ltau, m.neighbor_table, np.ascontiguousarray(m.cosht).reshape(-1).copy(),
                           np.ascontiguousarray(m.sinht).reshape(-1).copy(), np.exp(m.dtau * m.mu))
    n_arn = min(20, m.Nsites)
    Po = oracle.make_kpm(om, n=n_arn) if with_kpm else None
    P = pc.SymmetricKPMPreconditioner(m, n=n_arn, buf=0.05, c1=1.0, c2=1.0) if with_kpm else None
    H = hmc.HybridMonteCarlo(m, fa, dt, nt * dt, alpha=0.3, Nb=nb)
    v0 = 0.2 * synth.randn(77, m.Ndof)
    H.v[:] = v0
    H.push_()
    for u in (0.0, 1.5):      # accepted, then (u > 1) rejected
        rnd = dict(R=synth.randn(1500, m.Ndof), Rp=synth.randn(1501, m.Ndim), Rm=synth.randn(1502, m.Ndim),
                   kpm_randn=synth.randn(1503, (nt + 2) * 2 * m.Nsites) if with_kpm else None, u=u)
        x_in, v_in = m.x.copy(), H.v.copy()
        acc_o, x_o, v_o, info = oracle.hmc_update_ssh(om, x_in, v_in, m.omega, m.omega4, m.mu, m.dtau, fa.M, m.t, m.alpha, m.alpha2,
                                                      m.phonon_to_bond, m.checkerboard_perm, dt, nt, nb, 0.3, rnd, P=Po, tol=1e-7,
                                                      maxiter=20000)
        acc, its = hmc.update_(m, H, fa, P, randoms=rnd)
        assert acc == acc_o == (u == 0.0) and H.flag == 0 and info["flag"] == 0
        assert abs(H.H0 - info["H0"]) < 1e-8 * abs(info["H0"]) and abs(H.H1 - info["H1"]) < 1e-6 * abs(info["H1"])
        assert rel(m.x, x_o) < 1e-6 and rel(H.v, v_o) < 1e-6
        if not acc:
            assert np.array_equal(m.x, x_in)
    m.close()


def test_hmc_chains_failed_solve_kills_only_that_chain():
    """A chain whose linear solve fails (flag > 0, HMC.jl:405-408) is rejected and restored bit-exactly while the other
    chains of the lockstep update go on and end where they would have ended alone."""
    from elphdynamics_amd import configs, hmc, preconditioners as pc, synth
    tag, nch, nt, dt = "b", 3, 3, 0.05
    m = configs.make_model(tag, tol=1e-7, maxiter=400)
    m.omega4[:] = 0.02
    fa = pc.FourierAccelerator(m)
    pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.3)
    X0 = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=900 + c) for c in range(nch)])
    X0[1] *= 40.0                               # chain 1: exp(-dtau V) spans hundreds of orders of magnitude: CG cannot converge in 400 iterations
    V0 = np.zeros((nch, m.Ndof))
    rnd = dict(R=np.stack([synth.randn(1000 + c, m.Ndof) for c in range(nch)]),
               Rp=np.stack([synth.randn(1100 + c, m.Ndim) for c in range(nch)]),
               Rm=np.stack([synth.randn(1200 + c, m.Ndim) for c in range(nch)]), kpm_randn=None, u=np.zeros(nch))
    H = hmc.HybridMonteCarlo(m, fa, dt=dt, tr=nt * dt, alpha=0.0, Nb=1, nchains=nch)
    H.X[:], H.V[:] = X0, V0
    H.push_()
    acc, its = hmc.update_chains_(m, H, fa, None, randoms=rnd, pull=True)
    assert H.flags[1] > 0 and not acc[1] and H.flags[0] == 0 and H.flags[2] == 0 and acc[0] and acc[2]
    assert np.array_equal(H.X[1], X0[1])                                  # dead chain: field restored bit-exactly
    assert H.energies[1, 4] == 0.0                                        # acceptance probability of a killed trajectory
    Xb, Vb = H.X.copy(), H.V.copy()
    m.close()
    for c in (0, 2):
        m1 = configs.make_model(tag, tol=1e-7, maxiter=400)
        m1.omega4[:] = 0.02
        fa1 = pc.FourierAccelerator(m1)
        pc.update_M_(fa1, m1, 0.0, np.inf, 1.0, 0.3)
        m1.x[:] = X0[c]
        H1 = hmc.HybridMonteCarlo(m1, fa1, dt=dt, tr=nt * dt, alpha=0.0, Nb=1)
        r1 = dict(R=rnd["R"][c], Rp=rnd["Rp"][c], Rm=rnd["Rm"][c], kpm_randn=None, u=0.0)
        a1, i1 = hmc.update_(m1, H1, fa1, None, randoms=r1)
        assert a1 and np.abs(m1.x - Xb[c]).max() < 1e-6 * np.abs(m1.x).max() and np.abs(H1.v - Vb[c]).max() < 1e-6 * np.abs(H1.v).max()
        m1.close()
    # every chain failing: the update returns at once, all rejected, nothing moved
    m = configs.make_model(tag, tol=1e-12, maxiter=3)
    m.omega4[:] = 0.02
    H = hmc.HybridMonteCarlo(m, fa, dt=dt, tr=nt * dt, alpha=0.0, Nb=1, nchains=2)
    H.X[:] = X0[[0, 2]]
    H.push_()
    rnd2 = {k: (v[[0, 2]] if isinstance(v, np.ndarray) and v.ndim == 2 else v) for k, v in rnd.items()}
    rnd2["u"] = np.zeros(2)
    acc, its = hmc.update_chains_(m, H, fa, None, randoms=rnd2, pull=True)
    assert not acc.any() and (H.flags > 0).all() and np.array_equal(H.X, X0[[0, 2]])
    m.close()


# ---------------------------------------------------------------------------------------------- special updates

def test_special_moves_match_golden_and_oracle(oracle):
    """elph_hmc_special_move (SpecialUpdates.jl reflection / swap moves) vs the dense golden actions and the oracle: S₀, S₁,
    accept / reject, the field after the move, on the Holstein golden case; then SSH swaps and a KPM run vs the oracle."""
    from elphdynamics_amd import hmc, lattice as lat, models, preconditioners as pc, synth
    g, hg = golden("special_sq4_L8.npz"), golden("holstein_sq4_L8.npz")
    la = lat.Lattice(1, 4, 4, 1)
    m = models.HolsteinModel(la, int(g["Ltau"]) * float(hg["dtau"]), float(hg["dtau"]), tol=1e-6, maxiter=20000)
    m.neighbor_table, m.t = np.array(hg["raw"]), np.array(hg["t_raw"])
    m.initialize_model_()
    m.lam[:], m.lam2[:], m.mu[:] = hg["lam"], hg["lam2"], hg["mu"]
    m.omega[:], m.omega4[:] = g["omega"], g["omega4"]
    fa = pc.FourierAccelerator(m)
    pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.3)
    N, L = m.Nsites, m.Ltau
    X0 = hg["x"].reshape(N, L)
    for kind, ci, cj, key in ((0, 2, 0, "S1_reflect2"), (0, 7, 0, "S1_reflect7"), (1, 0, 1, "S1_swap0_1"), (1, 5, 9, "S1_swap5_9")):
        for u, want in ((0.0, True), (1.5, False)):
            m.x[:] = hg["x"]
            H = hmc.HybridMonteCarlo(m, fa, 0.05, 0.1)
            acc, S0, S1, it, fl = hmc.special_move_(m, H, kind, ci, cj, randoms=dict(Rp=g["Rp"], Rm=g["Rm"], kpm_randn=None, u=u))
            H.pull_()
            assert fl == 0 and acc == want
            assert abs(S0 - float(g["S0"])) < 1e-11 * abs(float(g["S0"])) and abs(S1 - float(g[key])) < 1e-8 * abs(float(g[key]))
            X = X0.copy()
            if want:
                if kind == 0:
                    X[ci] = -X[ci]
                else:
                    X[[ci, cj]] = X[[cj, ci]]
            assert np.array_equal(m.x, X.reshape(-1))
    # the drivers: accepted fractions are fractions, the field stays a permutation / sign flip of world lines
    m.x[:] = hg["x"]
    H = hmc.HybridMonteCarlo(m, fa, 0.05, 0.1)
    rng = np.random.default_rng(5)
    f1 = hmc.reflection_update_(m, H, 3, rng=rng)
    f2 = hmc.swap_update_(m, H, 3, rng=rng)
    H.pull_()
    assert 0.0 <= f1 <= 1.0 and 0.0 <= f2 <= 1.0
    assert np.allclose(np.sort(np.abs(m.x.reshape(N, L)).sum(axis=1)), np.sort(np.abs(X0).sum(axis=1)), rtol=1e-13)
    m.close()
    # SSH swap of two bond-phonon columns, with the KPM preconditioner, vs the oracle
    from elphdynamics_amd import configs
    e = configs.make_model("e", tol=1e-6, maxiter=20000)
    e.alpha2[:] = 0.01
    e.omega4 = np.full(e.Nph, 0.02)
    models.update_model_(e)
    fae = pc.FourierAccelerator(e)
    pc.update_M_(fae, e, 0.0, np.inf, 1.0, 0.3)
    om = oracle.make_model(1, e.Nsites, e.Ltau, e.neighbor_table, np.ascontiguousarray(e.cosht).reshape(-1).copy(),
                           np.ascontiguousarray(e.sinht).reshape(-1).copy(), np.exp(e.dtau * e.mu))
    Po = oracle.make_kpm(om, n=min(20, e.Nsites))
    Pe = pc.SymmetricKPMPreconditioner(e, n=min(20, e.Nsites), buf=0.05, c1=1.0, c2=1.0)
    He = hmc.HybridMonteCarlo(e, fae, 0.05, 0.1)
    rnd = dict(Rp=synth.randn(3100, e.Ndim), Rm=synth.randn(3101, e.Ndim), kpm_randn=synth.randn(3102, 2 * e.Nsites), u=0.3)
    x_in = e.x.copy()
    acc_o, x_o, info = oracle.special_move(om, x_in, 1, 3, 17, rnd["Rp"], rnd["Rm"], rnd["u"], e.omega, e.omega4, np.zeros(e.Nsites),
                                           np.zeros(e.Nsites), e.mu, e.dtau, P=Po, kpm_randn=rnd["kpm_randn"], tol=1e-6, maxiter=20000,
                                           ssh=dict(t=e.t, alpha=e.alpha, alpha2=e.alpha2, phonon_to_bond=e.phonon_to_bond,
                                                    cb_perm=e.checkerboard_perm))
    acc, S0, S1, it, fl = hmc.special_move_(e, He, 1, 3, 17, P=Pe, randoms=rnd)
    He.pull_()
    assert fl == 0 and info["flag"] == 0 and acc == acc_o
    assert abs(S0 - info["S0"]) < 1e-10 * abs(info["S0"]) and abs(S1 - info["S1"]) < 1e-7 * abs(info["S1"])
    assert np.array_equal(e.x, x_o)
    e.close()


# ---------------------------------------------------------------------------------------------- the library's own generator

@pytest.mark.parametrize("tag,nch,with_kpm", [("b", 1, False), ("B", 3, True)])
def test_device_rng_equals_explicit_batches(tag, nch, with_kpm):
    """elph_hmc_set_rng: the random inputs left NULL are drawn inside the library (field / site vectors on the GPU) — the run
    equals the one that is handed synth.randn(batch seed, n) explicitly, batch by batch, over two updates and a special move."""
    import ctypes as C
    from elphdynamics_amd import configs, hmc, preconditioners as pc, synth
    nt, dt, seed = 3, 0.05, 0xC0FFEE

    def make():
        m = configs.make_model(tag, tol=1e-9, maxiter=20000)
        fa = pc.FourierAccelerator(m)
        pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.3)
        H = hmc.HybridMonteCarlo(m, fa, dt=dt, tr=nt * dt, alpha=0.1, Nb=2, nchains=nch)
        if nch > 1:
            H.X[:] = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=900 + c) for c in range(nch)])
            H.push_()
        P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0) if with_kpm else None
        return m, fa, H, P

    upd = (lambda m, H, fa, P, r: hmc.update_chains_(m, H, fa, P, randoms=r, pull=True)) if nch > 1 else \
          (lambda m, H, fa, P, r: hmc.update_(m, H, fa, P, randoms=r))
    # a) the library draws
    m, fa, H, P = make()
    H.device_rng_(seed)
    res_a = [upd(m, H, fa, P, None) for _ in range(2)]
    Ea = np.array(H.energies if nch > 1 else [H.H0, H.H1, H.S, H.K, H.P_accept])
    if nch == 1:
        mv_a = hmc.special_move_(m, H, hmc.REFLECT, 3, P=P)
        H.pull_()
    nb_batches = C.c_uint64()
    assert m._lib.elph_hmc_rng_batches(m._h, C.byref(nb_batches)) == 0
    Xa = (H.X if nch > 1 else m.x).copy()
    m.close()
    # b) the same batches handed over
    m, fa, H, P = make()
    b = [0]

    def batch(n, uniform=False):
        b[0] += 1
        s = synth.batch_seed(seed, b[0])
        return synth.uniform01(s, n) if uniform else synth.randn(s, n)

    res_b = []
    for _ in range(2):
        r = dict(R=batch(nch * m.Ndof), Rp=batch(nch * m.Ndim), Rm=batch(nch * m.Ndim))
        r["kpm_randn"] = batch((nt + 2) * 2 * nch * m.Nsites) if with_kpm else None
        u = batch(nch, uniform=True)
        r["u"] = u if nch > 1 else float(u[0])
        res_b.append(upd(m, H, fa, P, r))
    Eb = np.array(H.energies if nch > 1 else [H.H0, H.H1, H.S, H.K, H.P_accept])
    if nch == 1:
        r = dict(Rp=batch(m.Ndim), Rm=batch(m.Ndim), kpm_randn=batch(2 * m.Nsites) if with_kpm else None)
        r["u"] = float(batch(1, uniform=True)[0])
        mv_b = hmc.special_move_(m, H, hmc.REFLECT, 3, P=P, randoms=r)
        H.pull_()
        assert mv_a[0] == mv_b[0] and abs(mv_a[1] - mv_b[1]) < 1e-8 * abs(mv_b[1]) and abs(mv_a[2] - mv_b[2]) < 1e-8 * abs(mv_b[2])
    assert nb_batches.value == b[0]
    Xb = (H.X if nch > 1 else m.x).copy()
    for (aa, ia), (ab, ib) in zip(res_a, res_b):
        assert np.array_equal(np.atleast_1d(aa), np.atleast_1d(ab)) and np.all(np.abs(np.atleast_1d(ia) - np.atleast_1d(ib)) <= 1)
    assert np.allclose(Ea, Eb, rtol=1e-8, atol=1e-8)
    assert np.abs(Xa - Xb).max() < 1e-8 * np.abs(Xb).max()
    m.close()


def test_device_rng_normals_have_the_right_moments():
    """The device generator itself: mean, variance, fourth moment and lag-1 correlation of 2.6M normals drawn as R of one
    update (read back through the momenta: alpha = 0, no accelerator mass => v = R after refresh_v!, and nt = 0 keeps it)."""
    from elphdynamics_amd import configs, hmc, preconditioners as pc, synth
    m = configs.make_model("C", tol=1e-5)
    fa = pc.FourierAccelerator(m)
    fa.M[:] = 1.0
    nch = 8
    H = hmc.HybridMonteCarlo(m, fa, dt=0.01, tr=0.0, alpha=0.0, Nb=1, nchains=nch)
    H.X[:] = synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau)
    H.push_()
    H.device_rng_(12345)
    hmc.update_chains_(m, H, fa, None, pull=True)
    v = H.V.reshape(-1) * np.where(H.accepted, 1.0, -1.0).repeat(m.Ndof)
    assert np.allclose(v, synth.randn(synth.batch_seed(12345, 1), nch * m.Ndof), rtol=0, atol=1e-12)
    n = v.size
    assert abs(v.mean()) < 5 / np.sqrt(n) and abs(v.var() - 1) < 5 * np.sqrt(2 / n) and abs((v ** 4).mean() - 3) < 5 * np.sqrt(96 / n)
    assert abs(np.mean(v[1:] * v[:-1])) < 5 / np.sqrt(n)
    m.close()


# ---------------------------------------------------------------------------------------------- SSH chains in lockstep

@pytest.mark.parametrize("tag,nch,nb,with_kpm", [("e", 3, 1, False), ("e", 2, 2, True), ("E", 4, 1, True)])
def test_ssh_hmc_chains_in_lockstep_equal_single_chain_updates(tag, nch, nb, with_kpm):
    """Bond-phonon chains: per-chain hopping tables in every kernel of the path, per-chain KPM expansions, per-chain bond-bracket
    force — every chain ends where the single-chain SSH update ends given the same field, momenta and random numbers."""
    from elphdynamics_amd import configs, hmc, preconditioners as pc, synth
    nt, dt = 3, 0.05

    def make():
        m = configs.make_model(tag, tol=1e-9, maxiter=20000)
        m.alpha2[:] = 0.01
        m.omega4 = np.full(m.Nph, 0.02)
        fa = pc.FourierAccelerator(m)
        pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.3)
        return m, fa

    m, fa = make()
    X0 = np.stack([m.x * (0.6 + 0.5 * c / nch) * (1.0 + 0.2 * synth.randn(900 + c, m.Ndof)) for c in range(nch)])
    V0 = np.stack([0.3 * synth.randn(950 + c, m.Ndof) for c in range(nch)])
    rnd = dict(R=np.stack([synth.randn(1000 + c, m.Ndof) for c in range(nch)]),
               Rp=np.stack([synth.randn(1100 + c, m.Ndim) for c in range(nch)]),
               Rm=np.stack([synth.randn(1200 + c, m.Ndim) for c in range(nch)]),
               kpm_randn=synth.randn(1300, (nt + 2) * 2 * nch * m.Nsites).reshape(nt + 2, 2, nch, m.Nsites) if with_kpm else None,
               u=np.array([0.0 if c % 2 == 0 else 1.5 for c in range(nch)]))
    H = hmc.HybridMonteCarlo(m, fa, dt=dt, tr=nt * dt, alpha=0.2, Nb=nb, nchains=nch)
    H.X[:], H.V[:] = X0, V0
    H.push_()
    P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0) if with_kpm else None
    acc, its = hmc.update_chains_(m, H, fa, P, randoms=rnd, pull=True)
    assert not H.flags.any()
    Xb, Vb, Eb = H.X.copy(), H.V.copy(), H.energies.copy()
    m.close()
    for c in range(nch):
        m1, fa1 = make()
        m1.x[:] = X0[c]
        H1 = hmc.HybridMonteCarlo(m1, fa1, dt=dt, tr=nt * dt, alpha=0.2, Nb=nb)
        H1.v[:] = V0[c]
        H1.push_()
        P1 = pc.SymmetricKPMPreconditioner(m1, 20, 0.05, 1.0, 1.0) if with_kpm else None
        r1 = dict(R=rnd["R"][c], Rp=rnd["Rp"][c], Rm=rnd["Rm"][c],
                  kpm_randn=np.ascontiguousarray(rnd["kpm_randn"][:, :, c, :]) if with_kpm else None, u=float(rnd["u"][c]))
        a1, i1 = hmc.update_(m1, H1, fa1, P1, randoms=r1)
        assert a1 == bool(acc[c]) and H1.flag == 0 and abs(i1 - its[c]) <= 1
        assert abs(H1.H0 - Eb[c, 0]) < 1e-8 * abs(H1.H0) and abs(H1.H1 - Eb[c, 1]) < 1e-8 * abs(H1.H1)
        assert np.abs(m1.x - Xb[c]).max() < 1e-7 * np.abs(m1.x).max() and np.abs(H1.v - Vb[c]).max() < 1e-7 * max(np.abs(H1.v).max(), 1.0)
        if not a1:
            assert np.array_equal(Xb[c], X0[c])
        m1.close()
    assert acc[0] and not acc[1]


def test_ssh_langevin_chains_match_single_trajectories():
    from elphdynamics_amd import configs, langevin, preconditioners as pc, synth
    nch = 3

    def make():
        m = configs.make_model("e", tol=1e-9, maxiter=20000)
        m.alpha2[:] = 0.01
        m.omega4 = np.full(m.Nph, 0.02)
        fa = pc.FourierAccelerator(m)
        pc.update_Q_(fa, m, 0.0, np.inf, 0.7)
        return m, fa, pc.SymmetricKPMPreconditioner(m, n=min(20, m.Nsites), buf=0.05, c1=1.0, c2=1.0)

    def rnd(step, c, m):
        return dict(eta=synth.randn(3000 + 10 * step + c, m.Ndof), g1=synth.randn(3100 + 10 * step + c, m.Ndim),
                    g2=synth.randn(3200 + 10 * step + c, m.Ndim), kpm_randn=synth.randn(3300 + 10 * step + c, 4 * m.Nsites))

    m, fa, P = make()
    x0 = m.x.copy()
    starts = [x0 * (0.7 + 0.2 * c) for c in range(nch)]
    singles = []
    for c in range(nch):
        m.x[:] = starts[c]
        dyn = langevin.HeunsDynamics(m, fa, 0.01)
        for step in range(2):
            langevin.evolve_(m, dyn, fa, P, randoms=rnd(step, c, m))
            assert dyn.flag == 0
        singles.append(m.x.copy())
    m.close()
    m, fa, P = make()
    dyn = langevin.HeunsDynamics(m, fa, 0.01, nchains=nch)
    for c in range(nch):
        dyn.X[c] = starts[c]
    dyn.push_()
    for step in range(2):
        rs = [rnd(step, c, m) for c in range(nch)]
        kr = np.stack([r["kpm_randn"].reshape(2, 2, m.Nsites) for r in rs], axis=2)
        langevin.evolve_(m, dyn, fa, P, randoms=dict(eta=np.stack([r["eta"] for r in rs]), g1=np.stack([r["g1"] for r in rs]),
                                                     g2=np.stack([r["g2"] for r in rs]), kpm_randn=kr))
        assert (dyn.flags == 0).all()
    for c in range(nch):
        assert rel(dyn.X[c] - starts[c], singles[c] - starts[c]) < 1e-7
    m.close()


@pytest.mark.parametrize("with_kpm", [False, True])
def test_special_moves_of_chains_equal_single_chain_moves(with_kpm):
    """elph_hmc_special_move_chains: every chain proposes its own reflection / swap; S0, S1 and the decision of each chain equal
    the single-chain move on that chain's field with the same random numbers; rejected chains get their field back."""
    from elphdynamics_amd import configs, hmc, preconditioners as pc, synth
    nch, tag = 3, "b"

    def make(nchains):
        m = configs.make_model(tag, tol=1e-9, maxiter=20000)
        fa = pc.FourierAccelerator(m)
        pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.3)
        H = hmc.HybridMonteCarlo(m, fa, dt=0.05, tr=0.1, alpha=0.0, Nb=1, nchains=nchains)
        P = pc.SymmetricKPMPreconditioner(m, 16, 0.05, 1.0, 1.0) if with_kpm else None
        return m, fa, H, P

    m, fa, H, P = make(nch)
    X0 = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=900 + c) for c in range(nch)])
    H.X[:] = X0
    H.push_()
    for kind, ci, cj in ((hmc.REFLECT, [1, 5, 9], None), (hmc.SWAP, [0, 3, 7], [1, 4, 11])):
        rnd = dict(Rp=np.stack([synth.randn(1100 + c, m.Ndim) for c in range(nch)]), Rm=np.stack([synth.randn(1200 + c, m.Ndim) for c in range(nch)]),
                   kpm_randn=synth.randn(1300, 2 * nch * m.Nsites).reshape(2, nch, m.Nsites) if with_kpm else None,
                   u=np.array([0.0, 1.5, 0.0]))                      # chain 1 is forced to reject
        H.X[:] = X0
        H.push_()
        acc, s0, s1, it, fl = hmc.special_move_chains_(m, H, kind, ci, cj, P=P, randoms=rnd)
        H.pull_()
        assert not fl.any() and acc[0] and not acc[1] and acc[2]
        assert np.array_equal(H.X[1], X0[1])                       # undone bit-exactly
        for c in range(nch):
            m1, fa1, H1, P1 = make(1)
            m1.x[:] = X0[c]
            H1.push_()
            r1 = dict(Rp=rnd["Rp"][c], Rm=rnd["Rm"][c], kpm_randn=np.ascontiguousarray(rnd["kpm_randn"][:, c, :]) if with_kpm else None,
                      u=float(rnd["u"][c]))
            a1, S0, S1, i1, f1 = hmc.special_move_(m1, H1, kind, ci[c], cj[c] if cj else 0, P=P1, randoms=r1)
            H1.pull_()
            assert a1 == bool(acc[c]) and abs(S0 - s0[c]) < 1e-9 * abs(S0) and abs(S1 - s1[c]) < 1e-8 * abs(S1)
            assert np.abs(m1.x - H.X[c]).max() < 1e-12
            m1.close()
    # the frequency-driven helpers run per chain too
    r = hmc.reflection_update_(m, H, 2, P, rng=np.random.default_rng(1))
    s = hmc.swap_update_(m, H, 2, P, rng=np.random.default_rng(2))
    assert 0.0 <= r <= 1.0 and 0.0 <= s <= 1.0
    m.close()


@pytest.mark.parametrize("mu", [0.0, -0.3])
def test_hmc_samples_the_exactly_solvable_single_site_model(mu):
    """End-to-end physics, independent of the restatement: the reference's single-site deck (holstein_hmc_single_site.toml:
    one site, no hopping) is exactly solvable.  The pseudofermion weight det(Λ⁻¹MᵀMΛ⁻¹) = det(M)² e^{Δτ λ Σx} is the
    particle-hole symmetric coupling H = p²/2 + ω²x²/2 + λ x (n − 1) − μ n, so E_n = −μ n − λ²(n−1)²/(2ω²) and
    <x> = −λ(<n> − 1)/ω²,  <x²> = Σ_n p_n x_n² + coth(βω/2)/(2ω).  64 chains in lockstep supply the statistics."""
    from elphdynamics_amd import hmc, lattice as lat, models, preconditioners as pc
    beta, dtau, w, lam, nch, nup = 2.0, 0.1, 1.0, 1.0, 64, 700
    E = [-mu * n - lam ** 2 * (n - 1) ** 2 / (2 * w ** 2) for n in (0, 1, 2)]
    wgt = np.array([1, 2, 1]) * np.exp(-beta * np.array(E))
    p = wgt / wgt.sum()
    n_exact = float(p @ np.array([0, 1, 2]))
    x_exact = -lam * (n_exact - 1) / w ** 2
    x2_exact = float(p @ (lam * (np.array([0, 1, 2]) - 1) / w ** 2) ** 2) + 1.0 / (2 * w * np.tanh(beta * w / 2))
    m = models.HolsteinModel(lat.Lattice(1, 1, 1, 1), beta, dtau, tol=1e-10, maxiter=1000)
    m.assign_omega_(w), m.assign_lambda_(lam), m.assign_mu_(mu)
    m.initialize_model_()
    fa = pc.FourierAccelerator(m)
    pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.0)
    H = hmc.HybridMonteCarlo(m, fa, dt=0.1, tr=1.0, alpha=0.0, Nb=1, nchains=nch)
    H.X[:] = 0.5 * np.random.default_rng(5).standard_normal((nch, 1))
    H.push_()
    H.device_rng_(20260131)
    xs, x2s, acc = [], [], 0.0
    for k in range(nup):
        a, it = hmc.update_chains_(m, H, fa, None, pull=True)
        acc += a.mean()
        if k >= 100:
            xs.append(H.X.mean())
            x2s.append(np.mean(H.X ** 2))
    xs, x2s = np.array(xs), np.array(x2s)
    nb = 20
    err = lambda v: v[:len(v) // nb * nb].reshape(nb, -1).mean(axis=1).std(ddof=1) / np.sqrt(nb)
    assert acc / nup > 0.9
    assert err(xs) < 0.03 and err(x2s) < 0.03                                  # the statistics are what they should be
    assert abs(xs.mean() - x_exact) < 4 * err(xs) + 0.01, (xs.mean(), x_exact, err(xs))
    assert abs(x2s.mean() - x2_exact) < 4 * err(x2s) + 0.015, (x2s.mean(), x2_exact, err(x2s))    # + O(Δτ²) of the discretised path
    m.close()


@pytest.mark.parametrize("mu", [0.0, -0.5])
def test_ssh_hmc_samples_the_exactly_solvable_two_site_model(mu):
    """The same for bond phonons (the reference's ssh_hmc_two_site deck): K = Σ_σ (c†₁σ c₂σ + h.c.) commutes with
    H = −(t − α x) K − μ N + p²/2 + ω² x²/2, every (N, k) sector is a displaced oscillator:
    E = −t k − μ N − α² k²/(2ω²),  <x> = −α <k>/ω²,  <x²> = α² <k²>/ω⁴ + coth(βω/2)/(2ω).  SSH chains in lockstep."""
    import itertools
    from elphdynamics_amd import hmc, lattice as lat, models, preconditioners as pc
    beta, dtau, w, t, alpha, nch, nup = 2.0, 0.1, 1.0, 1.0, 0.8, 64, 700
    one = [(0, 0), (1, 1), (1, -1), (2, 0)]                   # (N, k) of one spin species: empty, bonding, antibonding, full
    Z = k_av = k2_av = 0.0
    for (N1, k1), (N2, k2) in itertools.product(one, one):
        N, k = N1 + N2, k1 + k2
        wgt = np.exp(-beta * (-t * k - mu * N - alpha ** 2 * k ** 2 / (2 * w ** 2)))
        Z += wgt; k_av += k * wgt; k2_av += k * k * wgt
    x_exact = -alpha * (k_av / Z) / w ** 2
    x2_exact = alpha ** 2 * (k2_av / Z) / w ** 4 + 1.0 / (2 * w * np.tanh(beta * w / 2))
    m = models.SSHModel(lat.Lattice(1, 2, 1, 1), beta, dtau, tol=1e-10, maxiter=1000)
    m.assign_hopping_(t, alpha, 0.0, w, 1, 1, (1, 0, 0), name="b")
    m.initialize_model_()
    m.mu[:] = mu
    assert m.Nbonds == 1 and m.Nph == 1
    models.update_model_(m)
    fa = pc.FourierAccelerator(m)
    pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.0)
    H = hmc.HybridMonteCarlo(m, fa, dt=0.1, tr=1.0, alpha=0.0, Nb=1, nchains=nch)
    H.X[:] = 0.3 * np.random.default_rng(5).standard_normal((nch, 1))
    H.push_()
    H.device_rng_(20260132)
    xs, x2s, acc = [], [], 0.0
    for kk in range(nup):
        a, it = hmc.update_chains_(m, H, fa, None, pull=True)
        acc += a.mean()
        if kk >= 100:
            xs.append(H.X.mean())
            x2s.append(np.mean(H.X ** 2))
    xs, x2s = np.array(xs), np.array(x2s)
    nb = 20
    err = lambda v: v[:len(v) // nb * nb].reshape(nb, -1).mean(axis=1).std(ddof=1) / np.sqrt(nb)
    assert acc / nup > 0.9 and err(xs) < 0.02 and err(x2s) < 0.05
    assert abs(xs.mean() - x_exact) < 4 * err(xs) + 0.01, (xs.mean(), x_exact, err(xs))
    assert abs(x2s.mean() - x2_exact) < 4 * err(x2s) + 0.03, (x2s.mean(), x2_exact, err(x2s))
    m.close()


def test_ssh_swap_update_of_chains():
    """swap_update_ for bond-phonon chains: every chain swaps two of ITS phonon world lines; with a forced acceptance the two
    columns are exchanged exactly, with a forced rejection the field comes back bit for bit."""
    from elphdynamics_amd import configs, hmc, preconditioners as pc
    nch = 3
    m = configs.make_model("e", tol=1e-9, maxiter=20000)
    m.omega4 = np.full(m.Nph, 0.0)
    fa = pc.FourierAccelerator(m)
    pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.3)
    H = hmc.HybridMonteCarlo(m, fa, dt=0.05, tr=0.1, alpha=0.0, Nb=1, nchains=nch)
    rng = np.random.default_rng(3)
    X0 = np.stack([m.x * (0.8 + 0.1 * c) * (1.0 + 0.1 * rng.standard_normal(m.Ndof)) for c in range(nch)])
    H.X[:] = X0
    H.push_()
    L = m.Ltau
    ci, cj = np.array([0, 3, 5]), np.array([2, 4, 9])
    rnd = dict(Rp=rng.standard_normal((nch, m.Ndim)), Rm=rng.standard_normal((nch, m.Ndim)), kpm_randn=None, u=np.array([0.0, 1.5, 0.0]))
    acc, s0, s1, it, fl = hmc.special_move_chains_(m, H, hmc.SWAP, ci, cj, randoms=rnd)
    H.pull_()
    assert not fl.any() and acc.tolist() == [True, False, True]
    for c in range(nch):
        x0, x1 = X0[c].reshape(m.Nph, L), H.X[c].reshape(m.Nph, L)
        if acc[c]:
            assert np.array_equal(x1[ci[c]], x0[cj[c]]) and np.array_equal(x1[cj[c]], x0[ci[c]])
            rest = np.setdiff1d(np.arange(m.Nph), [ci[c], cj[c]])
            assert np.array_equal(x1[rest], x0[rest])
        else:
            assert np.array_equal(x1, x0)
    frac = hmc.swap_update_(m, H, 2, None, rng=rng)
    assert 0.0 <= frac <= 1.0
    m.close()


def test_chemical_potential_moved_by_the_tuner(oracle):
    """model.μ changes between two updates (MuFinder.jl:68-107; what julia/ElPhGPU.jl's resident update! does with elph_hmc_set_mu): the update that follows
    is the update of a state created with the new μ — same bits — and the oracle's with that μ; with chains in lockstep every chain can carry
    its own μ (elph_hmc_set_mu_chains) and is the single chain with that μ."""
    from elphdynamics_amd import hmc
    dt, nt = 0.05, 2
    # (a) one chain: create with mu = 0, move to mu1, update  ==  create with mu1, update
    m, fa, _ = _pair(oracle, "b", tol=1e-7)
    rnd = _randoms(m, nt, 2100, False, 0.0)
    mu1 = 0.3 + 0.05 * np.arange(m.Nsites) / m.Nsites
    H = hmc.HybridMonteCarlo(m, fa, dt, nt * dt)
    hmc.set_mu_(m, H, mu1)
    acc, its = hmc.update_(m, H, fa, None, randoms=rnd)
    m2, fa2, _ = _pair(oracle, "b", tol=1e-7)
    m2.mu[:] = mu1
    from elphdynamics_amd import models
    models.update_model_(m2)
    E = oracle.update_model_holstein(m2.Nsites, m2.Ltau, m2.dtau, m2.x, m2.lam, m2.lam2, m2.mu)
    om2 = oracle.make_model(0, m2.Nsites, m2.Ltau, m2.neighbor_table, m2.cosht, m2.sinht, E)
    x0 = m2.x.copy()
    H2 = hmc.HybridMonteCarlo(m2, fa2, dt, nt * dt)
    acc2, its2 = hmc.update_(m2, H2, fa2, None, randoms=rnd)
    assert acc == acc2 and its == its2 and H.H0 == H2.H0 and H.H1 == H2.H1 and np.array_equal(m.x, m2.x) and np.array_equal(H.v, H2.v)
    acc_o, x_o, v_o, info = _oracle_update(oracle, om2, m2, fa2, x0, np.zeros(m2.Ndof), dt, nt, 1, 0.0, rnd)
    assert acc2 == acc_o and abs(H2.H0 - info["H0"]) < 1e-9 * abs(info["H0"]) and rel(m2.x, x_o) < 3e-7
    m.close(); m2.close()
    # (b) two chains in lockstep, each with its own mu: chain c == the single chain with mu_c
    m3, fa3, _ = _pair(oracle, "b", tol=1e-7)
    Hc = hmc.HybridMonteCarlo(m3, fa3, dt, nt * dt, nchains=2)
    Hc.X[:] = m3.x
    Hc.push_()
    mus = np.stack([np.zeros(m3.Nsites), mu1])
    hmc.set_mu_(m3, Hc, mus)
    rc = dict(R=np.stack([rnd["R"], rnd["R"]]), Rp=np.stack([rnd["Rp"], rnd["Rp"]]), Rm=np.stack([rnd["Rm"], rnd["Rm"]]), kpm_randn=None, u=np.zeros(2))
    accs, itss = hmc.update_chains_(m3, Hc, fa3, None, randoms=rc, pull=True)
    assert accs[1] == acc2 and abs(Hc.energies[1, 0] - H2.H0) < 1e-12 * abs(H2.H0) and rel(Hc.X[1], m2.x) < 1e-9
    m4, fa4, _ = _pair(oracle, "b", tol=1e-7)
    H4 = hmc.HybridMonteCarlo(m4, fa4, dt, nt * dt)
    hmc.update_(m4, H4, fa4, None, randoms=rnd)
    assert abs(Hc.energies[0, 0] - H4.H0) < 1e-12 * abs(H4.H0) and rel(Hc.X[0], m4.x) < 1e-9 and not np.allclose(Hc.X[0], Hc.X[1])
    m3.close(); m4.close()
