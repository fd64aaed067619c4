"""Host-side mirror of ProcessInputFile.jl: build the model, preconditioner, Fourier accelerator, dynamics and
special updates of a run from one of the reference's TOML decks (examples/*.toml), so that a deck written for the Julia code
drives the GPU path unchanged.

    sim = process_input_file("holstein_hmc_square.toml")      ProcessInputFile.jl:34-121
    sim.model, sim.Gr, sim.mu_tuner, sim.sim_params, sim.simulation_dynamics, sim.burnin_dynamics, sim.burnin_reflect_update,
    sim.sim_reflect_update, sim.burnin_swap_update, sim.sim_swap_update, sim.fa, sim.preconditioner

What is read: [lattice], [holstein] / [ssh] (all tables, disorder widths included), [solver] (+ [solver.preconditioner]),
[[fourier_acceleration]], [hmc] (+ [hmc.burnin], [hmc.reflection_update], [hmc.swap_update]) / [langevin],
[measurements].num_random_vectors, [simulation] (seed, counts, names — kept in sim.sim_params, nothing is created on disk).
[tune_density] (its parameters only; the tuner is control plane and stays with the caller).  What is not: the measurement container and its folders, logging, checkpoints — the reference's
control plane (SURVEY §8 "out of scope"); sim.input keeps the whole parsed deck for a driver that wants them.

Only solver.type = "CG" exists on the GPU (the path of BASELINE.json); GMRES / BiCGStab decks raise.
"""
import os
from types import SimpleNamespace

import numpy as np

from . import greens, hmc, initialize_phonons, io, langevin, lattice as _lat, models
from . import preconditioners as pc


def read_deck(filename):
    import tomli
    with open(filename, "rb") as f:
        return tomli.load(f)


def initialize_rng(inp):
    """ProcessInputFile.jl:589-606: numpy Generator seeded with simulation.random_seed (drawn and recorded when absent)."""
    sim = inp.setdefault("simulation", {})
    if "random_seed" not in sim:
        sim["random_seed"] = int(np.random.SeedSequence().entropy % (2 ** 63))
    return np.random.default_rng(sim["random_seed"])


def _lattice(inp):
    la = inp["lattice"]                                                      # :227-234; Lattice(unit_cell, L) Lattices.jl:120-133
    ndim, L = int(la["ndim"]), la["L"]
    Ls = [int(v) for v in L] if isinstance(L, (list, tuple)) else [int(L)] * ndim
    Ls += [1] * (3 - len(Ls))
    lat = _lat.Lattice(int(la["norbits"]), *Ls[:3])
    lat.ndim = ndim
    lat.lattice_vectors = np.array(la.get("lattice_vectors", np.eye(ndim)), dtype=float)
    lat.basis_vectors = np.array(la.get("basis_vectors", np.zeros((lat.norbits, ndim))), dtype=float)
    return lat


def _check_solver(inp):
    s = inp["solver"]
    if str(s["type"]).lower() != "cg":
        raise NotImplementedError(f"solver.type = {s['type']!r}: the GPU path is the CG solve of MᵀM (solver.type = \"CG\")")
    return float(s["tol"]), int(s["maxiter"])


def initialize_holstein_model(inp, rng, device=0):
    """ProcessInputFile.jl:216-326."""
    tol, maxiter = _check_solver(inp)
    d = inp["holstein"]
    m = models.HolsteinModel(_lattice(inp), d["beta"], d["dtau"], tol=tol, maxiter=maxiter, device=device)
    for key, assign in (("omega", m.assign_omega_), ("mu", m.assign_mu_), ("omega4", m.assign_omega4_)):
        for e in d.get(key, []):
            for orbit in e["orbit"]:
                assign(e["val"], orbit, e.get("stddev", 0.0), rng)
    for e in d.get("t", []):
        dL = list(e["dL"]) + [0] * (3 - len(e["dL"]))
        m.assign_t_(e["val"], e["orbit"][0], e["orbit"][1], dL, e.get("stddev", 0.0), rng)
    for key, assign in (("lambda", m.assign_lambda_), ("lambda2", m.assign_lambda2_)):
        for e in d.get(key, []):
            for orbit in e["orbit"]:
                assign(e["val"], orbit, e.get("stddev", 0.0), rng)
    m.initialize_model_()
    return m


def initialize_ssh_model(inp, rng, device=0):
    """ProcessInputFile.jl:331-441."""
    tol, maxiter = _check_solver(inp)
    d = inp["ssh"]
    m = models.SSHModel(_lattice(inp), d["beta"], d["dtau"], tol=tol, maxiter=maxiter, device=device)
    for e in d["mu"]:
        for orbit in e["orbit"]:
            sel = slice(orbit - 1, None, m.lattice.norbits) if orbit else slice(None)
            n = len(m.mu[sel])
            m.mu[sel] = e["val"] + (e.get("stddev", 0.0) * rng.standard_normal(n) if e.get("stddev", 0.0) else 0.0)
    for e in d.get("hopping", []):
        g = lambda k: float(e.get(k, 0.0))
        m.assign_hopping_(g("t_avg"), g("alpha_avg"), g("alpha2_avg"), g("omega_avg"), e["orbits"][0], e["orbits"][1], list(e["dL"]),
                          omega4=g("omega4_avg"), name=e.get("name", ""), t_std=g("t_std"), omega_std=g("omega_std"),
                          omega4_std=g("omega4_std"), alpha_std=g("alpha_std"), alpha2_std=g("alpha2_std"))
    m.initialize_model_(rng)
    return m


def initialize_model(inp, rng=None, device=0):
    """ProcessInputFile.jl:186-211."""
    if ("holstein" in inp) == ("ssh" in inp):
        raise ValueError("the deck must hold exactly one of the [holstein] and [ssh] tables")
    rng = rng or initialize_rng(inp)
    m = initialize_holstein_model(inp, rng, device) if "holstein" in inp else initialize_ssh_model(inp, rng, device)
    m.rng = rng
    m.datafolder = inp.get("simulation", {}).get("datafolder", "")
    return m


def initialize_phonon_fields_(inp, model):
    """ProcessInputFile.jl:446-468: read the configuration file named in the deck, or the half-filling start."""
    d = inp["holstein" if "holstein" in inp else "ssh"]
    if d.setdefault("read_phonon_config", False):
        io.read_phonons_(model, d["phonon_config_file"])
    else:
        initialize_phonons.init_phonons_half_filled_(model, model.rng)


def initialize_preconditioner(inp, model):
    """ProcessInputFile.jl:473-514 (None plays the role of `I`)."""
    p = inp["solver"].get("preconditioner")
    if p is None:
        return None
    return pc.SymmetricKPMPreconditioner(model, n=int(p.get("n", 20)), buf=float(p.get("buf", 0.05)), c1=float(p.get("c1", 1.0)),
                                         c2=float(p.get("c2", 1.0)))


def initialize_fourieraccelerator(inp, model):
    """ProcessInputFile.jl:519-537."""
    fa = pc.FourierAccelerator(model)
    for d in inp["fourier_acceleration"]:
        pc.update_Q_(fa, model, d["omega_min"], d["omega_max"], d["mass"])
        pc.update_M_(fa, model, d["omega_min"], d["omega_max"], d["mass"], d.get("c", 0.0))
    return fa


def initialize_simulation_params(inp):
    """ProcessInputFile.jl:542-584 without the side effects (no folder, no log file)."""
    if "hmc" in inp:
        meas_freq, nsteps, burnin = inp["hmc"]["meas_freq"], inp["hmc"]["simulation_updates"], inp["hmc"]["burnin_updates"]
    else:
        lv = inp["langevin"]
        if lv["burnin_timesteps"] % lv["meas_freq"]:
            raise ValueError("langevin.burnin_timesteps must be a multiple of langevin.meas_freq")
        meas_freq, nsteps, burnin = lv["meas_freq"], lv["simulation_timesteps"], lv["burnin_timesteps"]
    sim = inp["simulation"]
    sim.setdefault("checkpoint_freq", 10)
    sim.setdefault("datafolder", os.path.join(sim.get("filepath", "."), sim.get("foldername", "")))
    return SimpleNamespace(burnin=int(burnin), nsteps=int(nsteps), meas_freq=int(meas_freq), num_bins=int(sim["num_bins"]),
                           checkpoint_freq=sim["checkpoint_freq"], filepath=sim.get("filepath", "."),
                           foldername=sim.get("foldername", ""), datafolder=sim["datafolder"])


def initialize_mutuner(inp, model):
    """ProcessInputFile.jl:611-624 — the PARAMETERS of the [tune_density] table only (active, μ₀, target ⟨N⟩, memory, κ_min·N): the
    tuner itself (MuFinder.jl) is control plane and not part of this package; the device side of a μ update is elph_hmc_set_mu / hmc.set_mu_."""
    mu0 = float(np.mean(model.mu))
    td = inp.get("tune_density")
    if td is None:
        return SimpleNamespace(active=False, mu0=mu0, N_target=1.0 * model.Nsites, nsites=model.Nsites, beta=model.beta, dtau=model.dtau,
                               memory=0.75, kappa_min=0.1)
    return SimpleNamespace(active=True, mu0=mu0, N_target=td["density"] * model.Nsites, nsites=model.Nsites, beta=model.beta,
                           dtau=model.dtau, memory=td["memory"], kappa_min=td["kappa_min"] * model.Nsites)


def initialize_dynamics(inp, model, fa, nchains=1):
    """ProcessInputFile.jl:626-700 -> (burnin_dynamics, simulation_dynamics).  The burn-in HybridMonteCarlo shares the
    device state of the simulation one (HybridMonteCarlo(simulation_dynamics, Δt, tr, α, Nb), HMC.jl:225-245)."""
    if ("hmc" in inp) == ("langevin" in inp):
        raise ValueError("the deck must hold exactly one of the [hmc] and [langevin] tables")
    if "hmc" in inp:
        h = inp["hmc"]
        dt, tr, alpha, nb = h["dt"], h["trajectory_time"], h["momentum_conservation_fraction"], h["num_multitimesteps"]
        sim = hmc.HybridMonteCarlo(model, fa, dt, tr, alpha, nb, nchains=nchains)
        b = h.get("burnin", {})
        burn = sim.sharing(b.get("dt", dt), b.get("trajectory_time", tr), b.get("momentum_conservation_fraction", alpha),
                           b.get("num_multitimesteps", nb))
        return burn, sim
    lv = inp["langevin"]
    cls = {1: langevin.EulerDynamics, 2: langevin.RungeKuttaDynamics, 3: langevin.HeunsDynamics}[int(lv["update_method"])]
    dyn = cls(model, fa, lv["dt"], nchains=nchains)
    return dyn, dyn


def initialize_reflect_update(inp, model):
    """ProcessInputFile.jl:705-733 -> (burnin, simulation); None is the NullUpdate."""
    ru = inp.get("hmc", {}).get("reflection_update") if "holstein" in inp else None
    sim = SimpleNamespace(freq=int(ru["freq"]), nsites=int(ru["nsites"])) if ru else None
    return sim, sim


def initialize_swap_update(inp, model):
    """ProcessInputFile.jl:738-766."""
    su = inp.get("hmc", {}).get("swap_update")
    sim = SimpleNamespace(freq=int(su["freq"]), nbonds=int(su["nbonds"])) if su else None
    return sim, sim


def process_input_file(deck, device=0, nchains=1, rng=None):
    """deck: path of a TOML file or an already parsed dict.  nchains > 1 (Holstein): the dynamics advance that many independent
    runs of the deck in lockstep on this GPU (what the reference does with one process per run ID, ElPhDynamics.jl:90-95)."""
    inp = read_deck(deck) if isinstance(deck, (str, os.PathLike)) else deck
    sim_params = initialize_simulation_params(inp)
    model = initialize_model(inp, rng, device)
    initialize_phonon_fields_(inp, model)
    mu_tuner = initialize_mutuner(inp, model)
    P = initialize_preconditioner(inp, model)
    fa = initialize_fourieraccelerator(inp, model)
    burn, sim = initialize_dynamics(inp, model, fa, nchains)
    if nchains > 1:      # every chain its own start configuration (a file gives all chains the same one, as separate runs would get)
        d = inp["holstein" if "holstein" in inp else "ssh"]
        for c in range(nchains):
            if c > 0 and not d.get("read_phonon_config", False):
                initialize_phonons.init_phonons_half_filled_(model, model.rng)
            sim.X[c] = model.x
        sim.push_()
        model._nchains = int(nchains)                # (init_phonons' update_model! was a single-configuration one)
    b_ref, s_ref = initialize_reflect_update(inp, model)
    b_swap, s_swap = initialize_swap_update(inp, model)
    # chains in lockstep: n_v vectors per chain in one estimator (vector v of chain c at greens.chain_vector(est, c, v))
    Gr = greens.EstimateGreensFunction(model, max(2, int(inp.get("measurements", {}).get("num_random_vectors", 2))) * max(1, int(nchains)))
    return SimpleNamespace(model=model, Gr=Gr, mu_tuner=mu_tuner, sim_params=sim_params, simulation_dynamics=sim, burnin_dynamics=burn,
                           burnin_reflect_update=b_ref, sim_reflect_update=s_ref, burnin_swap_update=b_swap, sim_swap_update=s_swap,
                           fa=fa, preconditioner=P, input=inp)
