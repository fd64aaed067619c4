"""Driver-loop, checkpoint and chain tests of the example control plane (examples/control_plane/run_simulation.py)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, ".."))
DECKS = os.path.join(ROOT, "tests", "decks")
@pytest.mark.gpu
@pytest.mark.parametrize("deck", ["holstein_hmc_honeycomb_L3.toml", "ssh_langevin_square_L4.toml"])
def test_deck_runs_through_the_driver_loop(deck):
    """run_simulation_ (RunSimulation.jl's loops around the GPU path): burn-in, simulation updates, special updates, and a Green's
    function estimate per measurement — the equal-time density from it is a sane number."""
    from elphdynamics_amd import greens, process_input as pi
    import run_simulation as rs
    sim = pi.process_input_file(os.path.join(DECKS, deck))
    dens = []

    def measure(sim, n):
        est = sim.Gr
        greens.setup_(est, 1, 2)
        dens.append(1.0 - float(np.real(greens.measure_GD0(est, 0, 0, 0, 1, 1, 0))))      # <n> = 1 - G(r = 0, tau = 0)

    stats = rs.run_simulation_(sim, measure=measure)
    sp = sim.sim_params
    assert len(dens) == sp.nsteps // sp.meas_freq and all(-0.5 < d < 1.5 for d in dens)
    assert stats["iters"] > 0 and 0.0 <= stats["acceptance_rate"] <= 1.0
    assert 0.0 <= stats["reflect_acceptance_rate"] <= 1.0 and 0.0 <= stats["swap_acceptance_rate"] <= 1.0
    assert np.all(np.isfinite(sim.model.x))
    sim.model.close()


@pytest.mark.gpu
def test_holstein_deck_in_lockstep_chains():
    """process_input_file(..., nchains): independent runs of one deck advance in lockstep on one GPU."""
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
    # the driver loop with chains: one estimator serves all of them (3 vectors per chain in this deck)
    from elphdynamics_amd import greens
    import run_simulation as rs
    assert sim.Gr.nv == 3 * 4
    dens = []

    def measure(sim, n):
        for c in range(4):
            greens.setup_(sim.Gr, greens.chain_vector(sim.Gr, c, 1), greens.chain_vector(sim.Gr, c, 2))
            dens.append(1.0 - float(np.real(greens.measure_GD0(sim.Gr, 0, 0, 0, 1, 1, 0))))

    stats = rs.run_simulation_(sim, measure=measure)
    assert len(dens) == 4 * (sim.sim_params.nsteps // sim.sim_params.meas_freq) and all(-0.5 < d < 1.5 for d in dens)
    assert 0.0 <= stats["acceptance_rate"] <= 1.0 and stats["iters"] > 0
    m.close()


@pytest.mark.gpu
def test_checkpoint_and_resume_continue_the_same_run(tmp_path):
    """A run interrupted after the burn-in phase and resumed from its checkpoint ends where the uninterrupted run ends: field,
    momenta, μ, tuner and generator state travel in the checkpoint; dynamics, accelerator, preconditioner and estimator are
    rebuilt, as in the reference — the rebuilt preconditioner starts its bound hysteresis afresh, so solves agree to the solver
    tolerance rather than bit for bit."""
    from elphdynamics_amd import process_input as pi
    import run_simulation as rs
    deck = os.path.join(DECKS, "holstein_hmc_honeycomb_L3.toml")

    def build():
        inp = pi.read_deck(deck)
        inp["tune_density"] = dict(density=0.9, memory=0.75, kappa_min=0.1)
        inp["hmc"]["burnin_updates"], inp["hmc"]["simulation_updates"] = 3, 3
        return pi.process_input_file(inp)

    a = build()
    stats_a = rs.run_simulation_(a)
    xa, mua = a.model.x.copy(), a.model.mu.copy()
    a.model.close()
    ck = str(tmp_path / "checkpoint.pkl")
    b = build()
    b.sim_params.nsteps = 0                                     # "crash" at the end of the burn-in (checkpoint written there)
    rs.run_simulation_(b, checkpoint=ck)
    b.model.close()
    c = build()
    stats_c = rs.run_simulation_(c, checkpoint=ck, resume=True)
    assert np.abs(c.model.x - xa).max() < 1e-5 * np.abs(xa).max() and np.abs(c.model.mu - mua).max() < 1e-5
    assert stats_c["acceptance_rate"] == stats_a["acceptance_rate"] and abs(stats_c["iters"] - stats_a["iters"]) < 1.0
    assert len(c.mu_tuner.N_traj) == len(a.mu_tuner.N_traj)
    c.model.close()


def test_checkpoint_carries_caller_state_and_refuses_another_deck(tmp_path):
    """The caller's accumulated state (the measurement sums of __main__) is written into every checkpoint and restored in place on
    resume (the reference serialises its measurement container, RunSimulation.jl:54-59); a checkpoint of another deck is refused."""
    import pytest
    import run_simulation as rs

    class _M:
        Nsites, Ltau, kind, Ndof = 4, 8, 0, 32
        x = np.zeros(32); mu = np.zeros(4)

    class _D:
        nchains = 1
        def pull_(self): pass
        def push_(self): pass

    class _S:
        model, simulation_dynamics, mu_tuner = _M(), _D(), None

    ck = str(tmp_path / "c.pkl")
    state = dict(n=7, glob=dict(density=1.5), corr=dict(G=np.arange(6.0)))
    rs.save_checkpoint(ck, _S(), 1, 3, dict(iters=2.0), None, extra=state)
    back = dict(n=0, glob=dict(density=0.0), corr=dict(G=np.zeros(6)))
    g_ref = back["corr"]["G"]
    S2 = _S()
    S2.model._lib = type("L", (), {"elph_hmc_set_mu": staticmethod(lambda *a: 0)})()
    S2.model._h = None
    phase, n, stats = rs.load_checkpoint(ck, S2, None, extra=back)
    assert (phase, n, stats["iters"]) == (1, 3, 2.0)
    assert back["n"] == 7 and back["glob"]["density"] == 1.5 and back["corr"]["G"] is g_ref and np.array_equal(g_ref, np.arange(6.0))
    S3 = _S()
    S3.model = _M()
    S3.model.Ltau = 16
    with pytest.raises(ValueError):
        rs.load_checkpoint(ck, S3, None)
