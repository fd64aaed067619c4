"""The deck reader (ProcessInputFile.jl mirror): host logic on the CPU; the full build of a run from a deck and a
few updates of it on the GPU."""
import os

import numpy as np
import pytest

DECKS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decks")


def test_deck_tables_and_simulation_params():
    from elphdynamics_amd import process_input as pi
    inp = pi.read_deck(os.path.join(DECKS, "holstein_hmc_honeycomb_L3.toml"))
    sp = pi.initialize_simulation_params(inp)
    assert (sp.burnin, sp.nsteps, sp.meas_freq, sp.num_bins, sp.checkpoint_freq) == (4, 8, 1, 4, 10)
    lat = pi._lattice(inp)
    assert (lat.norbits, lat.L1, lat.L2, lat.L3, lat.nsites) == (2, 3, 3, 1, 18)
    b, s = pi.initialize_reflect_update(inp, None)
    assert s.nsites == 2 and s.freq == 1 and b is s
    b, s = pi.initialize_swap_update(inp, None)
    assert s.nbonds == 3 and s.freq == 2
    rng = pi.initialize_rng(inp)
    assert rng.standard_normal() == np.random.default_rng(904375938239483).standard_normal()
    inp = pi.read_deck(os.path.join(DECKS, "ssh_langevin_square_L4.toml"))
    sp = pi.initialize_simulation_params(inp)
    assert (sp.burnin, sp.nsteps, sp.meas_freq) == (4, 8, 2)
    assert pi.initialize_reflect_update(inp, None) == (None, None) and pi.initialize_swap_update(inp, None) == (None, None)
    inp["langevin"]["burnin_timesteps"] = 5
    with pytest.raises(ValueError):
        pi.initialize_simulation_params(inp)
    inp["solver"]["type"] = "GMRES"
    with pytest.raises(NotImplementedError):
        pi._check_solver(inp)
    with pytest.raises(ValueError):
        pi.initialize_model({"holstein": {}, "ssh": {}})


def test_shared_fields_of_equally_named_phonon_types():
    """primary_field (SSHModels.jl:480-502): phonon types with the same name (the default "" included) share their fields."""
    from elphdynamics_amd import lattice as lat, models
    m = models.SSHModel.__new__(models.SSHModel)
    models.SSHModel.__init__(m, lat.Lattice(1, 4, 4, 1), 0.4, 0.1)
    m._create = lambda *a, **k: None                                   # host logic only
    for (o1, o2, d), name in zip(lat.SQUARE_BONDS, ("", "")):
        m.assign_hopping_(1.0, 0.1, 0.0, 0.5, o1, o2, d, name=name)
    m.assign_hopping_(0.3, 0.0, 0.0, 0.0, 1, 1, (1, 1, 0))             # a bond type without phonon (omega = 0)
    m.initialize_model_()
    L, per = m.Ltau, 16 * m.Ltau
    assert m.nph == 2 and m.Nph == 32 and m.Nbonds == 48 and m.has_shared_fields
    assert np.array_equal(m.primary_field[:per], np.arange(per)) and np.array_equal(m.primary_field[per:], np.arange(per))
    m2 = models.SSHModel.__new__(models.SSHModel)
    models.SSHModel.__init__(m2, lat.Lattice(1, 4, 4, 1), 0.4, 0.1)
    m2._create = lambda *a, **k: None
    for (o1, o2, d), name in zip(lat.SQUARE_BONDS, ("x", "y")):
        m2.assign_hopping_(1.0, 0.1, 0.0, 0.5, o1, o2, d, name=name)
    m2.initialize_model_()
    assert not m2.has_shared_fields and np.array_equal(m2.primary_field, np.arange(m2.Ndof))
    with pytest.raises(ValueError):
        m2.assign_hopping_(1.0, 0.1, 0.0, 0.5, 1, 1, (1, 1, 0), t_std=0.1)
        m2.initialize_model_()                                          # a disorder width without rng


@pytest.mark.gpu
def test_holstein_hmc_deck_builds_and_runs():
    from elphdynamics_amd import hmc, lattice as lat, models, preconditioners as pc, process_input as pi
    deck = os.path.join(DECKS, "holstein_hmc_honeycomb_L3.toml")
    sim = pi.process_input_file(deck)
    m = sim.model
    # the same model through the incremental calls, with the deck's generator
    rng = np.random.default_rng(904375938239483)
    ref = models.HolsteinModel(lat.Lattice(2, 3, 3, 1), 1.2, 0.1, tol=1e-8, maxiter=10000)
    ref.assign_omega_(1.0, 1), ref.assign_omega_(1.0, 2)
    ref.assign_mu_(-0.1, 1), ref.assign_mu_(-0.1, 2)
    ref.assign_omega4_(0.01, 1), ref.assign_omega4_(0.01, 2)
    for d in ([0, 0, 0], [-1, 0, 0], [0, -1, 0]):
        ref.assign_t_(1.0, 1, 2, d)
    ref.assign_lambda_(1.0, 1)
    ref.assign_lambda_(0.8, 2, 0.05, rng)
    ref.initialize_model_()
    for k in ("omega", "omega4", "mu", "lam", "lam2", "t", "cosht", "sinht"):
        assert np.array_equal(getattr(m, k), getattr(ref, k)), k
    assert np.array_equal(m.neighbor_table, ref.neighbor_table) and m.Ltau == 12 and m.Nbonds == 27
    assert np.std(m.lam[1::2]) > 0 and np.all(m.lam[0::2] == 1.0)
    ref.close()
    # phonon start: tau-constant world lines, model updated
    x = m.x.reshape(m.Nsites, m.Ltau)
    assert np.all(x == x[:, :1]) and np.std(x[:, 0]) > 0.1
    # the pieces
    P, fa, H, B = sim.preconditioner, sim.fa, sim.simulation_dynamics, sim.burnin_dynamics
    assert (P.n, P.buf) == (12, 0.05) and sim.Gr.nv == 3
    fa_ref = pc.FourierAccelerator(m)
    pc.update_Q_(fa_ref, m, 0.0, 10.0, 1.0), pc.update_M_(fa_ref, m, 0.0, 10.0, 1.0, 0.2)
    assert np.array_equal(fa.M, fa_ref.M) and np.array_equal(fa.Q, fa_ref.Q)
    assert (H.dt, H.Nt, H.Nb, H.alpha) == (0.05, 4, 3, 0.1) and (B.dt, B.Nt, B.Nb, B.alpha) == (0.1, 2, 2, 0.1)
    # burn-in and simulation updates move the same device-resident field
    x0 = m.x.copy()
    acc_b, it_b = hmc.update_(m, B, fa, P, rng=m.rng)
    x1 = m.x.copy()
    acc_s, it_s = hmc.update_(m, H, fa, P, rng=m.rng)
    assert B.flag == 0 and H.flag == 0 and it_b > 0 and it_s > 0
    assert (not acc_b) or np.abs(x1 - x0).max() > 0
    assert abs(H.H1 - H.H0) < 1.0
    r = hmc.reflection_update_(m, H, sim.sim_reflect_update.nsites, P, rng=m.rng)
    s = hmc.swap_update_(m, H, sim.sim_swap_update.nbonds, P, rng=m.rng)
    assert 0.0 <= r <= 1.0 and 0.0 <= s <= 1.0
    m.close()


@pytest.mark.gpu
def test_ssh_langevin_deck_builds_and_runs():
    from elphdynamics_amd import langevin, process_input as pi
    sim = pi.process_input_file(os.path.join(DECKS, "ssh_langevin_square_L4.toml"))
    m, dyn = sim.model, sim.simulation_dynamics
    assert m.kind == 1 and m.Ltau == 8 and m.Nph == 32 and not m.has_shared_fields and sim.preconditioner is None
    assert isinstance(dyn, langevin.HeunsDynamics) and sim.burnin_dynamics is dyn and dyn.dt == 1e-3
    assert np.all(m.omega == 0.5) and np.all(m.omega4[:16] == 0.01) and np.all(m.omega4[16:] == 0.0) and np.all(m.alpha2[:16] == 0.02)
    assert np.all(m.t[:16] == 1.0) and np.std(m.t[16:]) > 0 and np.all(m.mu == 0.05)
    x = m.x.reshape(m.Nph, m.Ltau)
    assert np.all(x == x[:, :1]) and abs(np.mean(x[:, 0]) + 2 * 0.1 / 0.25) < 1.5      # the -2 alpha / omega^2 offset of named types
    x0 = m.x.copy()
    it = langevin.evolve_(m, dyn, sim.fa, sim.preconditioner, rng=m.rng)
    assert dyn.flag == 0 and it > 0 and 0 < np.abs(m.x - x0).max() < 1.0
    m.close()


@pytest.mark.gpu
def test_ssh_deck_in_lockstep_chains():
    """process_input_file(..., nchains) for a bond-phonon deck: Langevin trajectories of several chains on one GPU."""
    from elphdynamics_amd import langevin, process_input as pi
    sim = pi.process_input_file(os.path.join(DECKS, "ssh_langevin_square_L4.toml"), nchains=3)
    dyn, m = sim.simulation_dynamics, sim.model
    assert dyn.nchains == 3 and dyn.X.shape == (3, m.Ndof)
    for c in range(3):
        dyn.X[c] = m.x * (0.8 + 0.1 * c)
    dyn.push_()
    X0 = dyn.X.copy()
    dyn.device_rng_(11)
    it = langevin.evolve_(m, dyn, sim.fa, sim.preconditioner)
    assert (dyn.flags == 0).all() and it.min() > 0
    d = np.abs(dyn.X - X0).max(axis=1)
    assert np.all(d > 0) and np.all(d < 1.0) and len(set(np.round(d, 12))) == 3       # every chain moved, each its own way
    m.close()


@pytest.mark.gpu
def test_holstein_deck_in_lockstep_chains():
    """process_input_file(..., nchains): independent runs of one deck advance in lockstep on one GPU; one estimator serves all chains."""
    from elphdynamics_amd import hmc, process_input as pi
    sim = pi.process_input_file(os.path.join(DECKS, "holstein_hmc_honeycomb_L3.toml"), nchains=4)
    H, m = sim.simulation_dynamics, sim.model
    rng = np.random.default_rng(5)
    for c in range(4):
        H.X[c] = m.x + 0.1 * rng.standard_normal(m.Ndof)
    H.push_()
    H.device_rng_(77)
    acc, its = hmc.update_chains_(m, sim.burnin_dynamics, sim.fa, sim.preconditioner, pull=True)
    acc2, its2 = hmc.update_chains_(m, H, sim.fa, sim.preconditioner, pull=True)
    assert not H.flags.any() and its.min() > 0 and its2.min() > 0 and np.all(np.isfinite(H.X))
    assert sim.Gr.nv == 3 * 4                      # 3 vectors per chain in this deck
    m.close()
