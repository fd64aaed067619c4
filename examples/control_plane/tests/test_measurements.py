"""Correlation-function measurements (Measurements.jl mirror, consumers of the Green's-function tables): the whole-table forms
against the literal per-displacement restatements on the CPU; identities and a deck run on the GPU."""
import os
from types import SimpleNamespace

import numpy as np
import pytest

DECKS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decks")


@pytest.mark.parametrize("kind", ["Greens", "DenDen", "SpinSpin", "PairGreens"])
def test_whole_table_forms_equal_the_literal_ones(kind):
    from elphdynamics_amd import lattice as lat
    import measurements as M
    rng = np.random.default_rng(3)
    L, ns, L1, L2, L3 = 5, 2, 3, 4, 2
    model = SimpleNamespace(Ltau=L, lattice=lat.Lattice(ns, L1, L2, L3))
    shp = (2 * L, ns, ns, L1, L2, L3)
    est = SimpleNamespace(L=L, **{k: rng.standard_normal(shp) + 1j * rng.standard_normal(shp) for k in ("GD0", "GD0_GD0", "GDD_G00", "GD0_G0D")})
    pairs = [(1, 1), (1, 2), (2, 1), (2, 2)]
    c = np.zeros((L + 1, L1, L2, L3, len(pairs)), dtype=complex)
    M.correlation_(c, pairs, model, est, kind)
    for p, (o1, o2) in enumerate(pairs):
        for t in range(L + 1):
            for idx in np.ndindex(L1, L2, L3):
                assert abs(c[(t,) + idx + (p,)] - M._SCALAR[kind](model, est, *idx, o1, o2, t)) < 1e-13


def test_translational_average_is_the_periodic_cross_correlation():
    import measurements as M
    rng = np.random.default_rng(1)
    f, g = rng.standard_normal((4, 3, 2)), rng.standard_normal((4, 3, 2))
    ta = M.translational_average(f.astype(complex), g)
    shape = np.array(f.shape)
    for r in np.ndindex(f.shape):
        ref = sum(f[tuple((np.array(i) + np.array(r)) % shape)] * g[i] for i in np.ndindex(f.shape))
        assert abs(ta[r] - ref) < 1e-12


@pytest.mark.gpu
def test_measurements_of_a_deck_run():
    from elphdynamics_amd import process_input as pi
    import measurements as M
    import run_simulation as rs
    sim = pi.process_input_file(os.path.join(DECKS, "holstein_hmc_honeycomb_L3.toml"))
    m = sim.model
    acc = M.new_accumulator(m)

    def measure(sim, n):
        M.make_measurements_(acc, sim.model, sim.Gr, sim.preconditioner, rng=sim.model.rng)

    rs.run_simulation_(sim, measure=lambda s, n: None)                       # thermalise a little, then measure on the final field
    M.make_measurements_(acc, m, sim.Gr, sim.preconditioner, rng=m.rng)
    npairs = sim.Gr.nv * (sim.Gr.nv - 1) // 2
    assert acc["n"] == npairs
    n = acc["n"]
    lat, L = m.lattice, m.Ltau
    # phonon Green's function: equal-time, zero displacement = <x^2> of that orbital, exactly; and the definition at one displacement
    x = m.x.reshape(lat.L3, lat.L2, lat.L1, lat.norbits, L).transpose(4, 3, 2, 1, 0)
    PG = acc["corr"]["PhononGreens"] / n
    for p, (o1, o2) in enumerate(acc["pairs"]):
        if o1 == o2:
            assert abs(PG[0, 0, 0, 0, p] - np.mean(x[:, o1 - 1] ** 2)) < 1e-12
        ref = np.mean(np.roll(np.roll(x[:, o1 - 1], -2, axis=0), -1, axis=1) * x[:, o2 - 1])     # τ = 2, l1 = 1
        assert abs(PG[2, 1, 0, 0, p] - ref) < 1e-12
        assert abs(PG[L, 1, 0, 0, p] - PG[0, 1, 0, 0, p]) < 1e-14
    # on-site observables: exact identities with the field and with the global density of the same noise vectors
    on = {k: v / n for k, v in acc["onsite"].items()}
    xs = m.x.reshape(m.Nsites, L)
    for o in range(lat.norbits):
        assert abs(on["x2"][o] - np.mean(xs[o::lat.norbits] ** 2)) < 1e-12 and abs(on["mu"][o] - m.mu[o]) < 1e-14
    assert abs(on["density"].mean() - acc["glob"]["density"] / n) < 1e-12
    # Green's function: G_r(β) = δ_r − G_r(0); equal-time diagonal consistent with the density within the stochastic error
    G = acc["corr"]["Greens"] / n
    for p, (o1, o2) in enumerate(acc["pairs"]):
        d = 1.0 if o1 == o2 else 0.0
        assert abs(G[L, 0, 0, 0, p] - (d - G[0, 0, 0, 0, p])) < 1e-13
    g00 = np.mean([np.real(G[0, 0, 0, 0, p]) for p, (o1, o2) in enumerate(acc["pairs"]) if o1 == o2])
    assert abs(2 * (1 - g00) - acc["glob"]["density"] / n) < 0.3
    assert all(np.all(np.isfinite(v)) for v in acc["corr"].values())
    m.close()


@pytest.mark.gpu
def test_measurements_of_a_bond_phonon_deck():
    from elphdynamics_amd import process_input as pi
    import measurements as M
    sim = pi.process_input_file(os.path.join(DECKS, "ssh_langevin_square_L4.toml"))
    m = sim.model
    acc = M.new_accumulator(m)
    M.make_measurements_(acc, m, sim.Gr, sim.preconditioner, rng=m.rng)
    n = acc["n"]
    assert n == sim.Gr.nv * (sim.Gr.nv - 1) // 2 and "PhononGreens" not in acc["corr"]
    on = {k: v / n for k, v in acc["onsite"].items()}
    assert abs(on["density"].mean() - acc["glob"]["density"] / n) < 1e-12 and abs(on["mu"][0] - 0.05) < 1e-14
    G = acc["corr"]["Greens"] / n
    assert abs(G[m.Ltau, 0, 0, 0, 0] - (1.0 - G[0, 0, 0, 0, 0])) < 1e-13
    assert all(np.all(np.isfinite(v)) for v in acc["corr"].values())
    m.close()
