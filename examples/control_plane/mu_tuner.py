"""Host-side mirror of MuFinder.jl: tune the chemical potential towards a target density from the stochastic Green's-function
estimates the GPU path produces (the solves are the hot path; the tuner itself is scalar bookkeeping).

    tuner = MuTuner(active, init_mu, target_N, N, beta, dtau, forgetful_c, kappa_min)        MuFinder.jl:16-63
    update_mu_(model, tuner, estimator, dyn=None) -> mu                                        :68-107  (all pairs of noise vectors)
    tuner.update(N, N2) -> mu                                                                   :112-166
    estimate_mu(tuner)                                                                          :172-201
    measure_density(est), measure_N2(model, est)                                                Measurements.jl:1283-1312

`dyn` (HybridMonteCarlo or Langevin dynamics) keeps the field on the device: the new μ is sent there (elph_hmc_set_mu) and the
model refreshed, as update_model! would on the host.
"""
import math

import numpy as np

from elphdynamics_amd import greens, models
from elphdynamics_amd._lib import check, dptr


def forgetful_mean(x, xbar_prev, c):
    """MuFinder.jl:210-226: mean over the most recent fraction c of the history, updated incrementally."""
    N = len(x)
    if N == 1:
        return x[0]
    i_p = 1 + math.floor((1.0 - c) * (N - 1))
    n = N - i_p + 1
    xbar = xbar_prev + (x[-1] - xbar_prev) / n
    i = 1 + math.floor((1.0 - c) * N)
    if i != i_p:
        n = N - i + 1
        xbar = xbar - (x[i_p - 1] - xbar) / n
    return xbar


def forgetful_welfords(x, xbar_prev, s_prev, c):
    """MuFinder.jl:232-261 -> (mean, standard deviation) over the most recent fraction c of the history."""
    N = len(x)
    if N == 1:
        return x[0], 0.0
    i_p = 1 + math.floor((1.0 - c) * (N - 1))
    n = N - i_p + 1
    xn = x[-1]
    M = (n - 2) * s_prev ** 2
    xbar = xbar_prev + (xn - xbar_prev) / n
    M = M + (xn - xbar) * (xn - xbar_prev)
    i = 1 + math.floor((1.0 - c) * N)
    if i != i_p:
        n = N - i + 1
        x0 = x[i_p - 1]
        xb = xbar
        xbar = xb - (x0 - xb) / n
        M = M - (x0 - xbar) * (x0 - xb)
    s = math.sqrt(M / (n - 1)) if n > 1 else 0.0
    return xbar, s


class MuTuner:
    def __init__(self, active, init_mu, target_N, N, beta, dtau, forgetful_c, kappa_min, logfile=""):
        self.active = bool(active)
        self.mu_traj, self.N_traj, self.N2_traj = [float(init_mu)], [], []
        self.forgetful_c = float(forgetful_c)
        self.mu = float(init_mu)
        self.N, self.beta, self.dtau, self.L = int(N), float(beta), float(dtau), int(round(beta / dtau))
        self.target_N = float(target_N)
        self.mu_bar, self.mu_std = float(init_mu), 0.0
        self.kappa_bar = float(kappa_min)
        self.N_bar, self.N_std, self.N2_bar = -1.0, 0.0, -1.0
        self.mu_bar_traj, self.kappa_bar_traj, self.N_bar_traj, self.N2_bar_traj = [], [], [], []
        self.kappa_min = float(kappa_min)
        self.mu_avg, self.mu_err = float(init_mu), 0.0
        self.logfile = logfile
        if self.active and logfile:
            import os
            if not os.path.isfile(logfile):
                with open(logfile, "w") as f:
                    f.write("mu_bar kappa_bar n_bar Nsqr_bar mu n Nsqr\n")

    def update(self, N, N2):
        """update_μ!(tuner, N, N²) (:112-166)."""
        self.N_traj.append(float(N))
        self.N2_traj.append(float(N2))
        self.mu_bar, self.mu_std = forgetful_welfords(self.mu_traj, self.mu_bar, self.mu_std, self.forgetful_c)
        self.N_bar = forgetful_mean(self.N_traj, self.N_bar, self.forgetful_c)
        self.N2_bar = forgetful_mean(self.N2_traj, self.N2_bar, self.forgetful_c)
        self.mu_bar_traj.append(self.mu_bar), self.N_bar_traj.append(self.N_bar), self.N2_bar_traj.append(self.N2_bar)
        n = len(self.N_traj)
        varN = self.N2_bar - self.N_bar ** 2
        k_lo = self.kappa_min / math.sqrt(n)
        k_hi = k_lo if (n == 1 or varN < 0.0 or self.mu_std <= 0.0) else math.sqrt(varN) / self.mu_std
        self.kappa_bar = max(min(self.beta * varN, k_hi), k_lo)
        self.kappa_bar_traj.append(self.kappa_bar)
        if self.active and self.logfile:
            with open(self.logfile, "a") as f:
                f.write("%.8f %.8f %.8f %.8f %.8f %.8f %.8f\n" % (self.mu_bar, self.kappa_bar / self.N, self.N_bar / self.N, self.N2_bar,
                                                                 self.mu, N / self.N, N2))
        self.mu = self.mu_bar + (self.target_N - self.N_bar) / self.kappa_bar
        self.mu_traj.append(self.mu)
        return self.mu


def estimate_mu(tuner):
    """estimate_μ(tuner) (:172-201): best estimate and spread of μ from the recent part of the trajectory."""
    if not tuner.active:
        tuner.mu_avg, tuner.mu_err = tuner.mu, 0.0
        return
    c = 0.5 if tuner.forgetful_c == 1.0 else tuner.forgetful_c
    idx = math.ceil(c * len(tuner.mu_traj))
    tr = np.array(tuner.mu_traj[idx - 1:])
    med = np.median(tr)
    tuner.mu_err = float(np.sqrt(np.sum((tr - med) ** 2) / (len(tr) - 1))) if len(tr) > 1 else float("nan")     # stdm(x, median)
    tuner.mu_avg = tuner.mu_bar


def measure_density(est):
    """Measurements.jl:1283-1292."""
    N, L = est.N, est.L
    N1 = 2 * (N - np.dot(est.Minvr1, est.r1) / L)
    N2 = 2 * (N - np.dot(est.Minvr2, est.r2) / L)
    return (N1 + N2) / (2 * N)


def measure_N2(model, est):
    """Measurements.jl:1297-1312: <N^2> from the pair of noise vectors selected by setup!."""
    N, L = est.N, est.L
    trG1 = np.dot(est.Minvr1, est.r1) / L
    trG2 = np.dot(est.Minvr2, est.r2) / L
    N1, N2 = 2 * (N - trG1), 2 * (N - trG2)
    return float(np.real(N1 * N2 + trG1 + trG2 - 2 * (N / est.ns) * np.sum(est.GD0_G0D[0])))


def update_mu_(model, tuner, est, dyn=None):
    """update_μ!(model, tuner, estimator) (:68-107)."""
    mu0 = float(np.mean(model.mu))
    if not tuner.active:
        tuner.mu = mu0
        return mu0
    Nsum = N2sum = 0.0
    npairs = 0
    for i in range(1, est.nv):
        for j in range(i + 1, est.nv + 1):
            greens.setup_(est, i, j)
            Nsum += model.Nsites * measure_density(est)
            N2sum += measure_N2(model, est)
            npairs += 1
    mu1 = tuner.update(Nsum / npairs, N2sum / npairs)
    model.mu += mu1 - mu0
    tuner.mu = mu1
    if dyn is not None:
        check(model._lib.elph_hmc_set_mu(model._h, dptr(np.ascontiguousarray(model.mu))))
        if model.kind == models.SSH:
            model._cs_stale = True
    else:
        models.update_model_(model)
    return mu1


def make_chain_tuners(tuner, nchains):
    """One tuner per chain for a deck run as chains in lockstep (copies of the deck's tuner, advanced independently)."""
    import copy
    return [copy.deepcopy(tuner) for _ in range(int(nchains))]


def update_mu_chains_(model, tuners, est, dyn):
    """update_μ! for every chain of a lockstep run: chain c's ⟨N⟩, ⟨N²⟩ come from ITS noise vectors in the shared estimator
    (greens.chain_vector), its tuner moves ITS chemical potential; the per-chain μ (dyn.mu_chains, (nchains, Nsites)) goes to the
    device state with elph_hmc_set_mu_chains.  model.mu keeps the deck's value.  -> array of the new chain means of μ."""
    nch = len(tuners)
    nvc = est.nv // nch
    if getattr(dyn, "mu_chains", None) is None:
        dyn.mu_chains = np.tile(np.asarray(model.mu, dtype=np.float64), (nch, 1))
    out = np.zeros(nch)
    for c, tuner in enumerate(tuners):
        mu0 = float(np.mean(dyn.mu_chains[c]))
        if not tuner.active:
            tuner.mu = out[c] = mu0
            continue
        Nsum = N2sum = 0.0
        npairs = 0
        for i in range(1, nvc):
            for j in range(i + 1, nvc + 1):
                greens.setup_(est, greens.chain_vector(est, c, i), greens.chain_vector(est, c, j))
                Nsum += model.Nsites * measure_density(est)
                N2sum += measure_N2(model, est)
                npairs += 1
        mu1 = tuner.update(Nsum / npairs, N2sum / npairs)
        dyn.mu_chains[c] += mu1 - mu0
        tuner.mu = out[c] = mu1
    check(model._lib.elph_hmc_set_mu_chains(model._h, dptr(np.ascontiguousarray(dyn.mu_chains).reshape(-1))))
    if model.kind == models.SSH:
        model._cs_stale = True
    return out
