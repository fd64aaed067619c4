"""Host-side mirror of RunSimulation.jl's driver loops (run_simulation!, :25-144 Langevin, :149-300 HMC) around the GPU path:
burn-in and simulation updates, reflection / swap updates at their frequencies, and at every meas_freq-th simulation update a
fresh Green's-function estimate handed to the caller's `measure(sim, n_meas)` — the measurement containers, binning, files,
checkpoints of the reference are control plane and stay with the caller; the chemical-potential tuner (mu_tuner.py)
runs where the reference runs it.

    sim = process_input.process_input_file("deck.toml")
    stats = run_simulation_(sim, measure=lambda sim, n: ...)
    -> dict(iters, acceptance_rate, reflect_acceptance_rate, swap_acceptance_rate, simulation_time, measurement_time)   (:130-141, :283-297)
"""
import time

import numpy as np

from elphdynamics_amd import greens, hmc, langevin
from mu_tuner import MuTuner, make_chain_tuners, update_mu_, update_mu_chains_


def _special(sim, dyn, n, reflect, swap, stats, P, rng):
    m = sim.model
    if reflect is not None and n % reflect.freq == 0:                                   # :188-191
        stats["reflect_acceptance_rate"] += hmc.reflection_update_(m, dyn, reflect.nsites, P, rng=rng)
    if swap is not None and n % swap.freq == 0:                                         # :194-197
        stats["swap_acceptance_rate"] += hmc.swap_update_(m, dyn, swap.nbonds, P, rng=rng)


def _deck_identity(sim):
    m, dyn = sim.model, sim.simulation_dynamics
    return dict(nsites=int(m.Nsites), ltau=int(m.Ltau), kind=int(m.kind), ndof=int(m.Ndof), nchains=int(getattr(dyn, "nchains", 1)))


def _restore_into(dst, src):
    """Deep in-place update: nested dicts are descended into, arrays are overwritten element-wise (the caller keeps its references)."""
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _restore_into(dst[k], v)
        elif isinstance(v, np.ndarray) and isinstance(dst.get(k), np.ndarray) and dst[k].shape == v.shape:
            dst[k][...] = v
        else:
            dst[k] = v


def save_checkpoint(path, sim, phase, n, stats, rng, extra=None):
    """Checkpoint of a run (the reference serialises (model, μ_tuner, container, burnin_start, sim_start, sim_stats) to
    checkpoint.jls, RunSimulation.jl:54-59,120-126,175-180; dynamics, accelerator, preconditioner and estimator are rebuilt from
    the deck on resume, ProcessInputFile.jl:122-177 — the same split here): field(s), momenta, chemical potential(s), tuner(s),
    generator state, position in the run, statistics."""
    import pickle
    dyn = sim.simulation_dynamics
    dyn.pull_()
    # `extra`: state the CALLER accumulates over the run (the measurement sums of __main__: the reference serialises its measurement
    # container in every checkpoint, RunSimulation.jl:54-59); `identity`: a checkpoint only resumes the deck it was written for
    state = dict(format="elphdynamics_amd checkpoint 2", identity=_deck_identity(sim), extra=extra,
                 phase=phase, n=n, stats=dict(stats), x=sim.model.x.copy(), mu=sim.model.mu.copy(),
                 X=None if getattr(dyn, "X", None) is None else dyn.X.copy(), V=None if getattr(dyn, "V", None) is None else dyn.V.copy(),
                 v=None if getattr(dyn, "v", None) is None else np.array(dyn.v).copy(), mu_chains=getattr(dyn, "mu_chains", None),
                 mu_tuner=getattr(sim, "mu_tuner", None), mu_tuners=getattr(sim, "mu_tuners", None),
                 rng=None if rng is None else rng.bit_generator.state)
    tmp = str(path) + ".tmp"
    with open(tmp, "wb") as f:
        pickle.dump(state, f)
    import os
    os.replace(tmp, path)


def load_checkpoint(path, sim, rng, extra=None):
    """Put a run built from the same deck (process_input_file) back where save_checkpoint left it -> (phase, n, stats).
    `extra` (a dict the caller owns) is updated in place with what the caller handed to save_checkpoint.  The file is a pickle
    written by this module: only load checkpoints of your own runs; one that belongs to another deck (lattice, time axis, model
    family, number of chains) is refused."""
    import pickle
    from elphdynamics_amd._lib import check, dptr
    with open(path, "rb") as f:
        st = pickle.load(f)
    if not isinstance(st, dict) or not {"phase", "n", "stats", "x", "mu"} <= set(st):
        raise ValueError(f"{path}: not a checkpoint of elphdynamics_amd")
    if "format" not in st:          # version 1 (before the identity / caller-state fields): resumed without the deck check
        import warnings
        warnings.warn(f"{path}: checkpoint without a format tag (version 1): resuming without the deck-identity check")
        st = dict(st, identity=_deck_identity(sim), extra=None)
    elif st["format"] != "elphdynamics_amd checkpoint 2":
        raise ValueError(f"{path}: checkpoint format {st['format']!r} is not known to this version")
    if st["identity"] != _deck_identity(sim):
        raise ValueError(f"{path} belongs to another deck: {st['identity']} != {_deck_identity(sim)}")
    if extra is not None and st.get("extra") is not None:
        _restore_into(extra, st["extra"])
    m, dyn = sim.model, sim.simulation_dynamics
    m.x[:], m.mu[:] = st["x"], st["mu"]
    if st["X"] is not None:
        dyn.X[:] = st["X"]
        if st["V"] is not None and getattr(dyn, "V", None) is not None:
            dyn.V[:] = st["V"]
    elif st["v"] is not None and getattr(dyn, "v", None) is not None:
        dyn.v[:] = st["v"]
    sim.mu_tuner = st["mu_tuner"] if st["mu_tuner"] is not None else sim.mu_tuner
    if st["mu_tuners"] is not None:
        sim.mu_tuners = st["mu_tuners"]
    dyn.push_()
    if st["mu_chains"] is not None:
        dyn.mu_chains = st["mu_chains"]
        check(m._lib.elph_hmc_set_mu_chains(m._h, dptr(np.ascontiguousarray(dyn.mu_chains).reshape(-1))))
    else:
        check(m._lib.elph_hmc_set_mu(m._h, dptr(np.ascontiguousarray(m.mu))))
    if rng is not None and st["rng"] is not None:
        rng.bit_generator.state = st["rng"]
    return st["phase"], st["n"], st["stats"]


def run_simulation_(sim, measure=None, rng=None, checkpoint=None, checkpoint_every=600.0, resume=False, checkpoint_state=None):
    """checkpoint: file written every `checkpoint_every` seconds (sim_params.checkpoint_freq is in minutes in the decks) and at
    the end of every phase; resume=True continues from it when it exists (the reference resumes when the data folder exists,
    ElPhDynamics.jl:102-107).  checkpoint_state: a dict of the caller's own run state (e.g. the sums its `measure` callback
    accumulates) — written into every checkpoint and restored IN PLACE on resume, so averages cover the whole run."""
    import os
    m, fa, P, sp = sim.model, sim.fa, sim.preconditioner, sim.sim_params
    rng = rng or getattr(m, "rng", None)
    tp = getattr(sim, "mu_tuner", None)
    if tp is not None and not isinstance(tp, MuTuner):                   # the deck reader hands over the [tune_density] parameters only
        sim.mu_tuner = MuTuner(tp.active, tp.mu0, tp.N_target, tp.nsites, tp.beta, tp.dtau, tp.memory, tp.kappa_min)
    stats = dict(simulation_time=0.0, measurement_time=0.0, write_time=0.0, iters=0.0, acceptance_rate=0.0,
                 reflect_acceptance_rate=0.0, swap_acceptance_rate=0.0)
    is_hmc = isinstance(sim.simulation_dynamics, hmc.HybridMonteCarlo)
    nch = int(getattr(sim.simulation_dynamics, "nchains", 1))
    start_phase, start_n = 0, 1
    if checkpoint and resume and os.path.exists(checkpoint):
        start_phase, last_n, stats = load_checkpoint(checkpoint, sim, rng, extra=checkpoint_state)
        start_n = last_n + 1
    t_ckpt = time.perf_counter()
    if nch > 1 and getattr(sim, "mu_tuner", None) is not None and sim.mu_tuner.active and getattr(sim, "mu_tuners", None) is None:
        sim.mu_tuners = make_chain_tuners(sim.mu_tuner, nch)            # every chain tunes its own chemical potential
    phases = ((sim.burnin_dynamics, sp.burnin, sim.burnin_reflect_update, sim.burnin_swap_update, False),
              (sim.simulation_dynamics, sp.nsteps, sim.sim_reflect_update, sim.sim_swap_update, True))
    tuner = getattr(sim, "mu_tuner", None)
    tuning = tuner is not None and tuner.active
    mu_freq = max(sp.meas_freq, 1)                                                      # :47
    for iphase, (dyn, nsteps, reflect, swap, measuring) in enumerate(phases):
        if iphase < start_phase:
            continue
        for n in range(start_n if iphase == start_phase else 1, nsteps + 1):
            t0 = time.perf_counter()
            if is_hmc and nch > 1:                                                      # chains in lockstep: means over the chains
                acc, it = hmc.update_chains_(m, dyn, fa, P, rng=rng, pull=False)
                stats["iters"] += float(it.mean())
                stats["acceptance_rate"] += float(acc.mean())
                _special(sim, dyn, n, reflect, swap, stats, P, rng)
            elif is_hmc:
                acc, it = hmc.update_(m, dyn, fa, P, rng=rng, pull=False)
                stats["iters"] += it
                stats["acceptance_rate"] += float(acc)
                _special(sim, dyn, n, reflect, swap, stats, P, rng)
            else:
                stats["iters"] += float(np.mean(langevin.evolve_(m, dyn, fa, P, rng=rng, pull=False)))
            if tuning and not measuring and (is_hmc or n % mu_freq == 0):               # burn-in: :65-68 (Langevin), :198-201 (HMC)
                greens.update_(sim.Gr, m, P, rng=rng)
                if nch > 1:
                    update_mu_chains_(m, sim.mu_tuners, sim.Gr, dyn)
                else:
                    update_mu_(m, tuner, sim.Gr, dyn)
            stats["simulation_time"] += time.perf_counter() - t0
            if measuring and n % sp.meas_freq == 0:                                     # :91-95 / :250-254
                t0 = time.perf_counter()
                dyn.pull_()                                                             # model.x for the caller's observables
                greens.update_(sim.Gr, m, P, rng=rng)                                   # make_measurements! starts with update!(Gr, …)
                if measure is not None:
                    measure(sim, n // sp.meas_freq)
                if tuning and nch > 1:
                    update_mu_chains_(m, sim.mu_tuners, sim.Gr, dyn)
                elif tuning:                                                            # :98-100 / :255-257
                    update_mu_(m, tuner, sim.Gr, dyn)
                stats["measurement_time"] += time.perf_counter() - t0
            if checkpoint and (time.perf_counter() - t_ckpt > checkpoint_every or n == nsteps):
                t0 = time.perf_counter()
                save_checkpoint(checkpoint, sim, iphase, n, stats, rng, extra=checkpoint_state)
                t_ckpt = time.perf_counter()
                stats["write_time"] += t_ckpt - t0
    total = sp.nsteps + sp.burnin
    stats["iters"] /= max(total, 1)                                                     # :131 / :284
    if is_hmc:
        stats["acceptance_rate"] /= max(total, 1)                                       # :287-289
        nref = sum(nst // r.freq for r, nst in ((sim.burnin_reflect_update, sp.burnin), (sim.sim_reflect_update, sp.nsteps)) if r)
        nswp = sum(nst // s.freq for s, nst in ((sim.burnin_swap_update, sp.burnin), (sim.sim_swap_update, sp.nsteps)) if s)
        stats["reflect_acceptance_rate"] /= max(nref, 1)
        stats["swap_acceptance_rate"] /= max(nswp, 1)
    else:
        stats["acceptance_rate"] = 1.0                                                  # :137
    sim.simulation_dynamics.pull_()
    return stats
