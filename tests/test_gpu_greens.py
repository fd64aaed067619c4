"""GPU parity of the stochastic Green's-function estimator: elph_greens_* through the C ABI vs the
golden direct-sum correlations and vs the oracle's restatement of GreensFunctions.jl on the same vectors.

Tolerance: the four products are bilinear in (M⁻¹R, R); with identical vectors the device and the FFTW-style oracle
differ only by summation order: 1e-12 of the largest element (observed ~1e-15).  End to end (own solve at
tol = 1e-11) the north_star's 1e-10 relative bound applies."""
import numpy as np
import pytest

from conftest import golden


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)

pytestmark = pytest.mark.gpu

NAMES = ["GD0", "GD0_GD0", "GDD_G00", "GD0_G0D"]


def _golden_model(name):
    from elphdynamics_amd import lattice as lat
    from elphdynamics_amd import models
    g = golden(name)
    norb, Lsp = (1, 4) if "sq4" in name else (2, 3)
    la = lat.Lattice(norb, Lsp, Lsp, 1)
    m = models.HolsteinModel(la, int(g["Ltau"]) * float(g["dtau"]), float(g["dtau"]), tol=1e-13)
    assert m.Ltau == int(g["Ltau"])
    m.neighbor_table, m.t = np.array(g["raw"]), np.array(g["t_raw"])
    m.initialize_model_()
    m.lam[:], m.lam2[:], m.mu[:] = g["lam"], g["lam2"], g["mu"]
    m.x[:] = g["x"]
    models.update_model_(m)
    return m


@pytest.mark.parametrize("hname,gname", [("holstein_sq4_L8.npz", "greens_sq4_L8.npz"), ("holstein_hc3_L6.npz", "greens_hc3_L6.npz")])
def test_setup_matches_golden_direct_correlations(hname, gname):
    from elphdynamics_amd import greens
    g = golden(gname)
    m = _golden_model(hname)
    est = greens.EstimateGreensFunction(m, nv=3)
    greens.set_vectors_(est, g["R"], g["MinvR"])
    for (n1, n2) in [(1, 2), (1, 3), (2, 3)]:
        greens.setup_(est, n1, n2)
        for nm in NAMES:
            got = getattr(est, nm).reshape(-1, order="F")
            ref = g["%s_%d%d" % (nm, n1, n2)]
            assert not got.imag.any()
            assert np.abs(got.real - ref).max() < 1e-12 * np.abs(ref).max(), (nm, n1, n2)
    # update!: the device's own batched solves reproduce the dense M⁻¹R, then the whole chain end to end
    it, res, fl = greens.update_(est, m, R=g["R"])
    assert not fl.any() and (res < 1e-9).all()
    # Green's-function elements within the north_star's 1e-10 (solve to 1e-13 against the dense M⁻¹R of the fixture)
    assert np.abs(est.MinvR - g["MinvR"]).max() < 1e-10 * np.abs(g["MinvR"]).max()
    greens.setup_(est, 2, 3)
    for nm in NAMES:
        ref = g["%s_23" % nm]
        assert np.abs(getattr(est, nm).reshape(-1, order="F").real - ref).max() < 1e-10 * np.abs(ref).max()
    # measure_* indexing (GreensFunctions.jl:293-329) and estimate (:334-346)
    L = m.Ltau
    G = g["GD0_23"].reshape(est.GD0.shape, order="F")
    o = est.ns
    assert abs(greens.measure_GD0(est, 2, 1, 0, 1, o, 3) - G[3, o - 1, 0, 2, 1, 0]) < 1e-10 * np.abs(G).max()
    assert abs(greens.measure_GD0(est, 0, 0, 0, o, o, 2 * L) - G[0, o - 1, o - 1, 0, 0, 0]) < 1e-10 * np.abs(G).max()
    assert greens.estimate(est, 2, 3, 4, 1, 2) == est.MinvR[2][(2 - 1) * L + 3] * est.R[2][(3 - 1) * L + 0]
    m.close()


@pytest.mark.parametrize("tag,nv", [("B", 4), ("D", 3), ("C", 2)])
def test_setup_matches_oracle_on_baseline_configs(tag, nv):
    """BASELINE sizes (square L=8/16, honeycomb L=12 with two orbitals): device vs the FFT restatement of the
    reference on identical (R, M⁻¹R); M⁻¹R from the device's own solve."""
    from elphdynamics_amd import configs, greens, synth
    from oracle.greens import EstimateGreensFunction as OracleEst
    m = configs.make_model(tag, tol=1e-8)
    est = greens.EstimateGreensFunction(m, nv=nv)
    R = np.stack([synth.randn(900 + i, m.Ndim) for i in range(nv)])
    it, res, fl = greens.update_(est, m, R=R)
    assert not fl.any()
    la = m.lattice
    orc = OracleEst(m.Ltau, la.norbits, la.L1, la.L2, la.L3, nv=nv)
    orc.R[:], orc.MinvR[:] = est.R, est.MinvR
    for (n1, n2) in [(1, 2), (nv - 1, nv)]:
        greens.setup_(est, n1, n2)
        orc.setup(n1, n2)
        for nm in NAMES:
            got, ref = getattr(est, nm), getattr(orc, nm)
            scale = np.abs(ref).max()
            assert np.abs(got - ref).max() < 1e-12 * scale, (tag, nm)
        # antiperiodic / periodic structure of the doubled time axis
        L = m.Ltau
        assert np.array_equal(est.GD0[L:], -est.GD0[:L]) and np.array_equal(est.GD0_GD0[L:], est.GD0_GD0[:L])
    # equal-time, zero-displacement element of G[Δ,0] is the noisy estimate of (1/NL) Σ M⁻¹[i,i] ≈ 1 - density
    assert 0.0 < est.GD0[0, 0, 0, 0, 0, 0].real < 1.0
    m.close()


def test_estimator_argument_checks():
    from elphdynamics_amd import _lib, configs, greens
    m = configs.make_model("b")
    lib = m._lib
    assert lib.elph_greens_setup(m._h, 1, 2, None, None, None, None) == _lib.ELPH_E_STATE      # not created
    assert lib.elph_greens_create(m._h, 1, 4, 3, 1, 2) == _lib.ELPH_E_ARG                      # 12 != 16 sites
    est = greens.EstimateGreensFunction(m, nv=1)
    assert est.nv == 2                                                                         # max(2, nv)
    assert lib.elph_greens_setup(m._h, 1, 2, None, None, None, None) == _lib.ELPH_E_STATE      # no vectors yet
    greens.update_(est, m, R=np.ones((2, m.Ndim)))
    assert lib.elph_greens_setup(m._h, 0, 2, None, None, None, None) == _lib.ELPH_E_ARG
    assert lib.elph_greens_setup(m._h, 1, 3, None, None, None, None) == _lib.ELPH_E_ARG
    assert lib.elph_greens_setup(m._h, 1, 2, None, None, None, None) == 0
    m.close()


def test_greens_estimator_serves_chains_in_lockstep():
    """One estimator for several resident chains: vector v of chain c lives at index v * nchains + c (right-hand side r of the
    batched solve uses chain r % nchains), its M⁻¹R and its translation-averaged tables equal those of the single-chain
    estimator on that chain's configuration."""
    from elphdynamics_amd import configs, greens, models, preconditioners as pc, synth
    nch, nv = 3, 2
    m = configs.make_model("b", tol=1e-9)
    X = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=400 + c) for c in range(nch)])
    R = np.stack([synth.randn(500 + r, m.Ndim) for r in range(nv * nch)])
    models.update_model_chains_(m, X)
    est = greens.EstimateGreensFunction(m, nv * nch)
    P = pc.SymmetricKPMPreconditioner(m, 16, 0.05, 1.0, 1.0)
    bmax, bmin = synth.randn(1, nch * m.Nsites).reshape(nch, -1), synth.randn(2, nch * m.Nsites).reshape(nch, -1)
    it, res, fl = greens.update_(est, m, P, R=R, setup_kwargs=dict(b_max=bmax, b_min=bmin))
    assert not fl.any()
    tabs = {}
    for c in range(nch):
        greens.setup_(est, greens.chain_vector(est, c, 1), greens.chain_vector(est, c, 2))
        tabs[c] = (est.GD0.copy(), est.GD0_G0D.copy(), est.MinvR[[c, nch + c]].copy())
    m.close()
    for c in range(nch):
        m1 = configs.make_model("b", tol=1e-9)
        m1.x[:] = X[c]
        models.update_model_(m1)
        e1 = greens.EstimateGreensFunction(m1, nv)
        P1 = pc.SymmetricKPMPreconditioner(m1, 16, 0.05, 1.0, 1.0)
        greens.update_(e1, m1, P1, R=R[[c, nch + c]], setup_kwargs=dict(b_max=bmax[c], b_min=bmin[c]))
        greens.setup_(e1, 1, 2)
        assert rel(tabs[c][2], e1.MinvR) < 1e-7
        assert rel(tabs[c][0], e1.GD0) < 1e-7 and rel(tabs[c][1], e1.GD0_G0D) < 1e-7
        m1.close()


def test_free_fermion_greens_function_in_expectation():
    """Physics check independent of the restatement: without electron-phonon coupling the estimator's translation-averaged
    G_r(τ) = <c_{i+r}(τ) c†_i(0)> must converge to the free result of the same Trotter decomposition,
    G(τ) = B^τ (1 + B^L)^{-1},  B = CB · diag(e^{Δτ μ})  (CB: the checkerboard product in the reference's bond order)."""
    from elphdynamics_amd import greens, lattice as lat, models, synth
    la = lat.Lattice(1, 4, 4, 1)
    L, dtau, mu = 8, 0.1, -0.3
    m = models.HolsteinModel(la, L * dtau, dtau, tol=1e-10, maxiter=20000)
    for (o1, o2, d) in lat.SQUARE_BONDS:
        m.assign_t_(1.0, o1, o2, d)
    m.assign_omega_(1.0), m.assign_lambda_(0.0), m.assign_mu_(mu)
    m.initialize_model_()
    m.x[:] = synth.randn(3, m.Ndof)                                          # irrelevant at λ = 0
    models.update_model_(m)
    N = m.Nsites
    CB = np.eye(N)
    for n in range(m.Nbonds):                                                 # checkerboard_mul!: bonds in order, acting on every column
        i, j = m.neighbor_table[n] - 1
        ri, rj = CB[i].copy(), CB[j].copy()
        CB[i] = m.cosht[n] * ri + m.sinht[n] * rj
        CB[j] = m.cosht[n] * rj + m.sinht[n] * ri
    B = CB * np.exp(dtau * mu)[None]
    G0 = np.linalg.inv(np.eye(N) + np.linalg.matrix_power(B, L))
    exact = np.zeros((L + 1, 4, 4))
    for t in range(L + 1):
        Gt = np.linalg.matrix_power(B, t) @ G0 if t < L else np.eye(N) - G0       # G(β) = 1 − G(0)
        for l1 in range(4):
            for l2 in range(4):
                exact[t, l1, l2] = np.mean([Gt[la.loc_to_site(1, a + l1, b + l2) - 1, la.loc_to_site(1, a, b) - 1]
                                            for a in range(4) for b in range(4)])
    nv = 12
    est = greens.EstimateGreensFunction(m, nv)
    acc = np.zeros((L + 1, 4, 4), dtype=complex)
    npairs, rms_early = 0, None
    for rep in range(16):
        R = np.stack([synth.randn(7000 + 100 * rep + v, m.Ndim) for v in range(nv)])
        it, res, fl = greens.update_(est, m, None, R=R)
        assert not fl.any()
        for i in range(1, nv, 2):                                            # independent pairs of noise vectors
            greens.setup_(est, i, i + 1)
            acc[:L] += est.GD0[:L, 0, 0, :, :, 0]
            acc[L] += (np.eye(4)[0][:, None] * np.eye(4)[0][None, :]) - est.GD0[0, 0, 0, :, :, 0]
            npairs += 1
        if rep == 1:
            rms_early = np.sqrt(np.mean(np.abs(acc / npairs - exact) ** 2))
    got = acc / npairs
    err = np.abs(got - exact).max()
    rms = np.sqrt(np.mean(np.abs(got - exact) ** 2))
    # pure statistical error (≈ 0.046 / sqrt(repetitions) per element, measured), shrinking as it must — no offset, no wrong sign
    assert rms < 0.02 and err < 0.06 and rms < 0.6 * rms_early, (err, rms, rms_early)
    assert np.abs(got.imag).max() < 0.05
    assert abs(exact[0, 0, 0] - 0.5) > 0.02                                   # μ ≠ 0: away from half filling, the sign of μ matters
    m.close()
