"""The chemical-potential tuner (MuFinder.jl mirror): scalar bookkeeping on the CPU; on the GPU the new μ reaches the
device-resident state and the density estimates that drive it come from the batched solves."""
import os

import numpy as np
import pytest

DECKS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decks")


@pytest.mark.parametrize("c", [1.0, 0.75, 0.3])
def test_forgetful_statistics_equal_the_windowed_ones(c):
    """forgetful_mean / forgetful_welfords (MuFinder.jl:210-261) are incremental forms of the mean / standard deviation over
    the most recent fraction c of the history, x[i:] with i = 1 + floor((1 - c) N) (1-based)."""
    from mu_tuner import forgetful_mean, forgetful_welfords
    rng = np.random.default_rng(4)
    x, xb, wb, ws = [], 0.0, 0.0, 0.0
    for n in range(1, 60):
        x.append(float(rng.standard_normal()) + 0.1 * n)
        xb = forgetful_mean(x, xb, c)
        wb, ws = forgetful_welfords(x, wb, ws, c)
        i = 1 + int(np.floor((1.0 - c) * n))
        w = np.array(x[i - 1:])
        assert abs(xb - w.mean()) < 1e-12 * max(1.0, abs(w.mean()))
        assert abs(wb - w.mean()) < 1e-12 * max(1.0, abs(w.mean()))
        if len(w) > 1:
            assert abs(ws - w.std(ddof=1)) < 1e-10 * max(1.0, w.std(ddof=1))


def test_tuner_moves_mu_against_the_density_error():
    """update_μ!(tuner, N, N²) (:112-166): μ = μ̄ + (target − N̄)/κ̄ with κ̄ clamped to [κ_min/√n, √var N/σ_μ]."""
    from mu_tuner import MuTuner, estimate_mu
    t = MuTuner(True, 0.0, 16.0, 16, 2.0, 0.1, 0.75, 1.6)
    mu = t.update(20.0, 20.0 ** 2 + 3.0)             # too many electrons -> μ goes down
    assert mu < 0.0 and t.kappa_bar == pytest.approx(1.6) and t.L == 20
    # a linear response N = 16 + 8 μ + noise: the tuner converges to μ = 0 … here to the μ where N = 16
    rng = np.random.default_rng(0)
    t = MuTuner(True, 0.5, 16.0, 16, 2.0, 0.1, 0.75, 1.6)
    mu = 0.5
    for _ in range(200):
        N = 16.0 + 8.0 * mu + 0.3 * rng.standard_normal()
        mu = t.update(N, N * N + 4.0)
    estimate_mu(t)
    assert abs(t.mu_avg) < 0.05 and t.mu_err < 0.2
    off = MuTuner(False, 0.3, 16.0, 16, 2.0, 0.1, 0.75, 0.1)
    estimate_mu(off)
    assert off.mu_avg == 0.3 and off.mu_err == 0.0


@pytest.mark.gpu
def test_tune_density_deck_runs_and_mu_reaches_the_device():
    from elphdynamics_amd import greens, models, mu_tuner, process_input as pi, run_simulation as rs
    inp = pi.read_deck(os.path.join(DECKS, "holstein_hmc_honeycomb_L3.toml"))
    inp["tune_density"] = dict(density=0.8, memory=0.75, kappa_min=0.1)
    inp["hmc"]["burnin_updates"], inp["hmc"]["simulation_updates"] = 6, 4
    sim = pi.process_input_file(inp)
    m, tuner = sim.model, sim.mu_tuner
    assert tuner.active and tuner.target_N == pytest.approx(0.8 * m.Nsites) and tuner.mu == pytest.approx(np.mean(m.mu))
    mu_start = m.mu.copy()
    stats = rs.run_simulation_(sim)
    assert len(tuner.N_traj) == 6 + 4 and len(tuner.mu_traj) == 11
    shift = m.mu - mu_start
    assert np.allclose(shift, shift[0]) and shift[0] != 0.0 and tuner.mu == pytest.approx(np.mean(m.mu))
    assert all(0.0 < n / m.Nsites < 2.0 for n in tuner.N_traj)
    # the device-resident model carries the tuned μ: M v from the HMC state equals M v of a model updated on the host
    v = np.random.default_rng(1).standard_normal(m.Ndim)
    y_dev = np.zeros(m.Ndim)
    models.mulM_(y_dev, m, v)
    models.update_model_(m)                                                            # host x (pulled at the end) and host μ
    y_host = np.zeros(m.Ndim)
    models.mulM_(y_host, m, v)
    assert np.abs(y_dev - y_host).max() < 1e-13 * np.abs(y_host).max()
    # the density the tuner sees (diagonal estimator r·M⁻¹r) agrees with the Green's-function table (cross estimator of the two
    # noise vectors) within the stochastic error of a 3-vector estimate on 18 sites
    greens.update_(sim.Gr, m, sim.preconditioner, rng=m.rng)
    greens.setup_(sim.Gr, 1, 2)
    n = mu_tuner.measure_density(sim.Gr)
    g00 = np.mean([np.real(greens.measure_GD0(sim.Gr, 0, 0, 0, o, o, 0)) for o in (1, 2)])
    assert abs(n - 2.0 * (1.0 - g00)) < 0.3
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("deck", ["holstein_hmc_honeycomb_L3.toml", "ssh_langevin_square_L4.toml"])
def test_tune_density_with_chains_in_lockstep(deck):
    """Chains in lockstep with [tune_density]: one tuner and one chemical potential per chain (elph_hmc_set_mu_chains); the
    device model of every chain carries ITS μ."""
    from elphdynamics_amd import models, process_input as pi, run_simulation as rs
    nch = 3
    inp = pi.read_deck(os.path.join(DECKS, deck))
    inp["tune_density"] = dict(density=0.9, memory=0.75, kappa_min=0.1)
    if "hmc" in inp:
        inp["hmc"]["burnin_updates"], inp["hmc"]["simulation_updates"] = 3, 2
    else:
        inp["langevin"]["burnin_timesteps"], inp["langevin"]["simulation_timesteps"], inp["langevin"]["meas_freq"] = 2, 2, 1
        inp["measurements"]["num_random_vectors"] = 3
    sim = pi.process_input_file(inp, nchains=nch)
    m, dyn = sim.model, sim.simulation_dynamics
    rng = np.random.default_rng(2)
    for c in range(nch):
        dyn.X[c] = m.x * (1.0 + 0.05 * c) + 0.02 * rng.standard_normal(m.Ndof)
    dyn.push_()
    mu_deck = m.mu.copy()
    stats = rs.run_simulation_(sim)
    tuners = sim.mu_tuners
    assert len(tuners) == nch and all(len(t.N_traj) >= 4 for t in tuners)
    mus = np.array([t.mu for t in tuners])
    assert len(set(np.round(mus, 10))) == nch and np.array_equal(m.mu, mu_deck)            # each chain went its own way
    assert np.allclose(dyn.mu_chains.mean(axis=1), mus)
    # chain c of the device model = a single model with chain c's field and chain c's μ: same solution of MᵀM x = b
    dyn.pull_()
    B = rng.standard_normal((nch, m.Ndim))
    Xs = np.zeros_like(B)
    it, res, fl = models.ldiv_batched_(Xs, m, B)
    assert not fl.any()
    for c in range(nch):
        s1 = pi.process_input_file(pi.read_deck(os.path.join(DECKS, deck)))
        m1 = s1.model
        m1.x[:] = dyn.X[c]
        m1.mu[:] = dyn.mu_chains[c]
        models.update_model_(m1)
        x1 = np.zeros(m.Ndim)
        it1, res1, fl1 = models.ldiv_(x1, m1, np.ascontiguousarray(B[c]))
        assert fl1 == 0 and np.linalg.norm(x1 - Xs[c]) < 1e-6 * np.linalg.norm(x1)
        m1.close()
    m.close()
