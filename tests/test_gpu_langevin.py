"""GPU parity of the Langevin dynamics step (LangevinDynamics.jl: calc_dSdx!, evolve! for Euler / Runge-Kutta / Heun) —
elph_langevin_* through the C ABI vs the dense golden steps and vs the oracle at the BASELINE sizes."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def _golden_model(tol=1e-10):
    from elphdynamics_amd import lattice as lat
    from elphdynamics_amd import models, preconditioners as pc
    g, h = golden("langevin_sq4_L8.npz"), golden("holstein_sq4_L8.npz")
    la = lat.Lattice(1, 4, 4, 1)
    m = models.HolsteinModel(la, int(g["Ltau"]) * float(g["dtau"]), float(g["dtau"]), tol=tol, maxiter=20000)
    m.neighbor_table, m.t = np.array(h["raw"]), np.array(h["t_raw"])
    m.initialize_model_()
    m.lam[:], m.lam2[:], m.mu[:] = h["lam"], h["lam2"], h["mu"]
    m.omega[:], m.omega4[:] = g["omega"], g["omega4"]
    m.x[:] = h["x"]
    models.update_model_(m)
    fa = pc.FourierAccelerator(m)
    fa.Q[:] = g["faQ"]
    return g, h, m, fa


@pytest.mark.parametrize("cls,key", [("EulerDynamics", "x_euler"), ("RungeKuttaDynamics", "x_rk"), ("HeunsDynamics", "x_heun")])
def test_langevin_step_matches_dense_golden(cls, key):
    from elphdynamics_amd import langevin
    g, h, m, fa = _golden_model()
    dyn = getattr(langevin, cls)(m, fa, float(g["dt"]))
    it = langevin.evolve_(m, dyn, fa, None, randoms=dict(eta=g["eta"], g1=g["g1"], g2=g["g2"], kpm_randn=None))
    assert dyn.flag == 0 and it > 0
    assert rel(m.x - h["x"], g[key] - h["x"]) < 1e-7
    m.close()


@pytest.mark.parametrize("tag,scheme,with_kpm", [("b", 0, False), ("d", 1, True), ("B", 2, True), ("C", 2, True), ("D", 2, True), ("q", 2, True), ("z", 1, True)])
def test_langevin_step_vs_oracle(oracle, tag, scheme, with_kpm):
    """Two consecutive steps (the field stays on the device between them) vs the oracle, with the KPM preconditioner set up
    from the same Arnoldi start vectors; also the reference's iteration-count conventions."""
    from elphdynamics_amd import configs, langevin, preconditioners as pc, synth
    m = configs.make_model(tag, tol=1e-8, maxiter=20000)
    m.omega4[:] = 0.02
    fa = pc.FourierAccelerator(m)
    pc.update_Q_(fa, m, 0.0, np.inf, 0.7)
    E = oracle.update_model_holstein(m.Nsites, m.Ltau, m.dtau, m.x, m.lam, m.lam2, m.mu)
    om = oracle.make_model(0, m.Nsites, m.Ltau, m.neighbor_table, m.cosht, m.sinht, E)
    n_arn = min(20, m.Nsites)
    Po = oracle.make_kpm(om, n=n_arn) if with_kpm else None
    P = pc.SymmetricKPMPreconditioner(m, n=n_arn, buf=0.05, c1=1.0, c2=1.0) if with_kpm else None
    dyn = [langevin.EulerDynamics, langevin.RungeKuttaDynamics, langevin.HeunsDynamics][scheme](m, fa, 0.01)
    x_o = m.x.copy()
    for step in range(2):
        rnd = dict(eta=synth.randn(2000 + step, m.Ndof), g1=synth.randn(2100 + step, m.Ndim), g2=synth.randn(2200 + step, m.Ndim),
                   kpm_randn=synth.randn(2300 + step, 4 * m.Nsites) if with_kpm else None)
        x_prev = x_o
        x_o, it_o = oracle.langevin_evolve(scheme, om, x_o, fa.Q, 0.01, rnd["eta"], rnd["g1"], rnd["g2"], m.omega, m.omega4, m.lam,
                                           m.lam2, m.mu, m.dtau, P=Po, kpm_randn=rnd["kpm_randn"], tol=1e-8, maxiter=20000)
        it = langevin.evolve_(m, dyn, fa, P, randoms=rnd)
        assert dyn.flag == 0 and abs(it - it_o) <= 1
        assert rel(m.x - x_prev, x_o - x_prev) < 1e-6
    m.close()


@pytest.mark.parametrize("cls,key", [("EulerDynamics", "x_euler"), ("RungeKuttaDynamics", "x_rk"), ("HeunsDynamics", "x_heun")])
def test_ssh_langevin_step_matches_dense_golden(cls, key):
    """Bond-phonon Langevin step (examples/ssh_langevin_square.toml geometry at 4x4): device vs the dense golden."""
    from elphdynamics_amd import langevin, lattice as lat, models, preconditioners as pc
    g, hg = golden("langevin_ssh_sq4_L8_a.npz"), golden("ssh_sq4_L8_a.npz")
    la = lat.Lattice(1, 4, 4, 1)
    L, dtau = int(g["Ltau"]), float(g["dtau"])
    m = models.SSHModel(la, L * dtau, dtau, tol=1e-10, maxiter=20000)
    for (o1, o2, d) in lat.SQUARE_BONDS:
        m.assign_hopping_(1.0, 0.1, 0.0, 0.5, o1, o2, d, name="xy"[d.index(1)])
    m.initialize_model_()
    m.alpha[:], m.alpha2[:], m.mu[:] = hg["alpha"], hg["alpha2"], hg["mu"]
    m.omega, m.omega4 = np.array(g["omega"]), np.array(g["omega4"])
    m.x[:] = hg["x"]
    models.update_model_(m)
    fa = pc.FourierAccelerator(m)
    fa.Q[:] = g["faQ"]
    dyn = getattr(langevin, cls)(m, fa, float(g["dt"]))
    it = langevin.evolve_(m, dyn, fa, None, randoms=dict(eta=g["eta"], g1=g["g1"], g2=g["g2"], kpm_randn=None))
    assert dyn.flag == 0 and it > 0
    assert rel(m.x - hg["x"], g[key] - hg["x"]) < 1e-7
    m.close()


def test_ssh_langevin_step_vs_oracle_at_config_E(oracle):
    """Config E (optical SSH square L = 16, Ltau = 160) with alpha2 != 0 and the KPM preconditioner: Heun step vs the oracle."""
    from elphdynamics_amd import configs, langevin, models, preconditioners as pc, synth
    m = configs.make_model("E", tol=1e-8, maxiter=20000)
    m.alpha2[:] = 0.01
    m.omega4 = np.full(m.Nph, 0.02)
    models.update_model_(m)
    fa = pc.FourierAccelerator(m)
    pc.update_Q_(fa, m, 0.0, np.inf, 0.7)
    om = oracle.make_model(1, m.Nsites, m.Ltau, m.neighbor_table, np.ascontiguousarray(m.cosht).reshape(-1).copy(),
                           np.ascontiguousarray(m.sinht).reshape(-1).copy(), np.exp(m.dtau * m.mu))
    Po = oracle.make_kpm(om, n=20)
    P = pc.SymmetricKPMPreconditioner(m, n=20, buf=0.05, c1=1.0, c2=1.0)
    dyn = langevin.HeunsDynamics(m, fa, 0.01)
    rnd = dict(eta=synth.randn(2000, m.Ndof), g1=synth.randn(2100, m.Ndim), g2=synth.randn(2200, m.Ndim), kpm_randn=synth.randn(2300, 4 * m.Nsites))
    x_in = m.x.copy()
    ssh = dict(t=m.t, alpha=m.alpha, alpha2=m.alpha2, phonon_to_bond=m.phonon_to_bond, cb_perm=m.checkerboard_perm)
    x_o, it_o = oracle.langevin_evolve_ssh(2, om, x_in, fa.Q, 0.01, rnd["eta"], rnd["g1"], rnd["g2"], m.omega, m.omega4, m.mu, m.dtau, ssh,
                                           P=Po, kpm_randn=rnd["kpm_randn"], tol=1e-8, maxiter=20000)
    it = langevin.evolve_(m, dyn, fa, P, randoms=rnd)
    assert dyn.flag == 0 and abs(it - it_o) <= 1
    assert rel(m.x - x_in, x_o - x_in) < 1e-6
    m.close()


def test_langevin_error_paths():
    from elphdynamics_amd import _lib, configs
    m = configs.make_model("b")
    lib = m._lib
    z = np.zeros(m.Ndim)
    assert lib.elph_langevin_evolve(m._h, 0, 0.01, 0, _lib.dptr(z), _lib.dptr(z), None, None, None, None) == _lib.ELPH_E_STATE
    m.close()
    e = configs.make_model("e")     # the Holstein entry point refuses an SSH handle
    assert e._lib.elph_langevin_create(e._h, _lib.dptr(np.zeros(e.Nsites)), _lib.dptr(np.zeros(e.Nsites)), _lib.dptr(np.zeros(e.Nsites)),
                                       _lib.dptr(np.zeros(e.Nsites)), _lib.dptr(np.zeros(e.Nsites)), 0.05,
                                       _lib.dptr(np.zeros(e.Ndim))) == _lib.ELPH_E_UNSUPPORTED
    e.close()


@pytest.mark.parametrize("tag,scheme,with_kpm", [("b", 0, False), ("d", 1, True), ("B", 2, True)])
def test_langevin_chains_match_single_trajectories(tag, scheme, with_kpm):
    """nchains trajectories in lockstep (one batched solve per force, one KPM expansion per chain) == each trajectory run alone
    with its own random numbers; two steps so that the fields differ between chains at the second."""
    from elphdynamics_amd import configs, langevin, preconditioners as pc, synth
    nch = 3
    cls = [langevin.EulerDynamics, langevin.RungeKuttaDynamics, langevin.HeunsDynamics][scheme]

    def make():
        m = configs.make_model(tag, tol=1e-9, maxiter=20000)
        m.omega4[:] = 0.02
        fa = pc.FourierAccelerator(m)
        pc.update_Q_(fa, m, 0.0, np.inf, 0.7)
        P = pc.SymmetricKPMPreconditioner(m, n=min(20, m.Nsites), buf=0.05, c1=1.0, c2=1.0) if with_kpm else None
        return m, fa, P

    def rnd(step, c, m):
        return dict(eta=synth.randn(3000 + 10 * step + c, m.Ndof), g1=synth.randn(3100 + 10 * step + c, m.Ndim),
                    g2=synth.randn(3200 + 10 * step + c, m.Ndim), kpm_randn=synth.randn(3300 + 10 * step + c, 4 * m.Nsites))

    m, fa, P = make()
    x0 = m.x.copy()
    singles, its = [], []
    for c in range(nch):
        m.x[:] = x0 + 0.05 * synth.randn(3400 + c, m.Ndof)
        dyn = cls(m, fa, 0.01)
        it_c = []
        for step in range(2):
            r = rnd(step, c, m)
            if not with_kpm:
                r["kpm_randn"] = None
            it_c.append(langevin.evolve_(m, dyn, fa, P, randoms=r))
            assert dyn.flag == 0
        singles.append(m.x.copy())
        its.append(it_c)
    m.close()

    m, fa, P = make()
    dyn = cls(m, fa, 0.01, nchains=nch)
    for c in range(nch):
        dyn.X[c] = x0 + 0.05 * synth.randn(3400 + c, m.Ndof)
    dyn.push_()
    for step in range(2):
        rs = [rnd(step, c, m) for c in range(nch)]
        kr = np.stack([r["kpm_randn"].reshape(2, 2, m.Nsites) for r in rs], axis=2) if with_kpm else None   # [2][2][chain][site]
        it = langevin.evolve_(m, dyn, fa, P, randoms=dict(eta=np.stack([r["eta"] for r in rs]), g1=np.stack([r["g1"] for r in rs]),
                                                          g2=np.stack([r["g2"] for r in rs]), kpm_randn=kr))
        assert (dyn.flags == 0).all()
        assert np.all(np.abs(it - np.array([its[c][step] for c in range(nch)])) <= 1)
    for c in range(nch):
        assert rel(dyn.X[c] - x0, singles[c] - x0) < 1e-7
    m.close()


@pytest.mark.parametrize("scheme", [0, 2])
def test_ssh_langevin_shared_fields_vs_oracle(oracle, scheme):
    """The reference's ssh_langevin_square deck leaves both phonon types unnamed, so the x- and y-bond phonons share their
    fields (primary_field, SSHModels.jl:480-502): two device steps vs the oracle's restatement of the sharing rules."""
    from elphdynamics_amd import langevin, lattice as lat, models, preconditioners as pc, synth
    la = lat.Lattice(1, 8, 8, 1)
    m = models.SSHModel(la, 2.0, 0.05, tol=1e-9, maxiter=20000)
    for (o1, o2, d) in lat.SQUARE_BONDS:
        m.assign_hopping_(1.0, 0.1, 0.01, 0.5, o1, o2, d, omega4=0.02)            # no names, like the deck
    m.initialize_model_()
    assert m.has_shared_fields
    half = m.Ndof // 2
    y = 0.8 * synth.randn(41, half)
    m.x[:] = np.concatenate([y, y])
    models.update_model_(m)
    fa = pc.FourierAccelerator(m)
    pc.update_Q_(fa, m, 0.0, np.inf, 0.7)
    om = oracle.make_model(1, m.Nsites, m.Ltau, m.neighbor_table, np.ascontiguousarray(m.cosht).reshape(-1).copy(),
                           np.ascontiguousarray(m.sinht).reshape(-1).copy(), np.exp(m.dtau * m.mu))
    Po = oracle.make_kpm(om, n=20)
    P = pc.SymmetricKPMPreconditioner(m, n=20, buf=0.05, c1=1.0, c2=1.0)
    dyn = [langevin.EulerDynamics, langevin.RungeKuttaDynamics, langevin.HeunsDynamics][scheme](m, fa, 0.01)
    ssh = dict(t=m.t, alpha=m.alpha, alpha2=m.alpha2, phonon_to_bond=m.phonon_to_bond, cb_perm=m.checkerboard_perm,
               primary_field=m.primary_field)
    x_o = m.x.copy()
    for step in range(2):
        eta = synth.randn(2000 + step, m.Ndof)[m.primary_field]
        rnd = dict(eta=eta, g1=synth.randn(2100 + step, m.Ndim), g2=synth.randn(2200 + step, m.Ndim),
                   kpm_randn=synth.randn(2300 + step, 4 * m.Nsites))
        x_prev = x_o
        x_o, it_o = oracle.langevin_evolve_ssh(scheme, om, x_o, fa.Q, 0.01, rnd["eta"], rnd["g1"], rnd["g2"], m.omega, m.omega4, m.mu,
                                               m.dtau, ssh, P=Po, kpm_randn=rnd["kpm_randn"], tol=1e-9, maxiter=20000)
        it = langevin.evolve_(m, dyn, fa, P, randoms=rnd)
        assert dyn.flag == 0 and abs(it - it_o) <= 1
        assert rel(m.x - x_prev, x_o - x_prev) < 1e-6
        assert np.array_equal(m.x[:half], m.x[half:])
    m.close()


def test_langevin_samples_the_exactly_solvable_single_site_model():
    """Heun's dynamics (64 chains in lockstep, Δt = 0.02) on the single-site Holstein model: <x>, <x²> of
    H = p²/2 + ω²x²/2 + λx(n − 1) − μn within the statistical error plus the O(Δt²) and stochastic-force bias of the scheme."""
    from elphdynamics_amd import langevin, lattice as lat, models, preconditioners as pc
    beta, dtau, w, lam, mu, nch, nst = 2.0, 0.1, 1.0, 1.0, -0.3, 64, 5000
    E = [-mu * n - lam ** 2 * (n - 1) ** 2 / (2 * w ** 2) for n in (0, 1, 2)]
    wgt = np.array([1, 2, 1]) * np.exp(-beta * np.array(E))
    p = wgt / wgt.sum()
    x_exact = -lam * (float(p @ np.array([0, 1, 2])) - 1) / w ** 2
    x2_exact = float(p @ (lam * (np.array([0, 1, 2]) - 1) / w ** 2) ** 2) + 1.0 / (2 * w * np.tanh(beta * w / 2))
    m = models.HolsteinModel(lat.Lattice(1, 1, 1, 1), beta, dtau, tol=1e-10, maxiter=1000)
    m.assign_omega_(w), m.assign_lambda_(lam), m.assign_mu_(mu)
    m.initialize_model_()
    fa = pc.FourierAccelerator(m)
    pc.update_Q_(fa, m, 0.0, np.inf, 1.0)
    dyn = langevin.HeunsDynamics(m, fa, 0.02, nchains=nch)
    dyn.X[:] = 0.5 * np.random.default_rng(5).standard_normal((nch, 1))
    dyn.push_()
    dyn.device_rng_(20260133)
    xs, x2s = [], []
    for k in range(nst):
        langevin.evolve_(m, dyn, fa, None, pull=(k % 10 == 9))
        if k % 10 == 9 and k >= 500:
            xs.append(dyn.X.mean())
            x2s.append(np.mean(dyn.X ** 2))
    xs, x2s = np.array(xs), np.array(x2s)
    nb = 20
    err = lambda v: v[:len(v) // nb * nb].reshape(nb, -1).mean(axis=1).std(ddof=1) / np.sqrt(nb)
    assert err(xs) < 0.03 and err(x2s) < 0.03
    assert abs(xs.mean() - x_exact) < 4 * err(xs) + 0.02, (xs.mean(), x_exact, err(xs))
    assert abs(x2s.mean() - x2_exact) < 4 * err(x2s) + 0.03, (x2s.mean(), x2_exact, err(x2s))
    m.close()
