"""End-to-end physics check: HMC on the single-site Holstein model (the reference's exactly solvable deck, holstein_hmc_single_site)
against the exact thermal averages.  The pseudofermion weight det(Λ⁻¹ MᵀM Λ⁻¹) = det(M)² e^{+Δτ λ Σx} (HMC.jl:820-915, update_Λ!) is
the particle-hole symmetric coupling:  H = p²/2 + w² x²/2 + lam x (n − 1) − mu n  (n = n_up + n_dn), so
E_n = −mu n − lam² (n − 1)² / (2 w²),  <x> = −lam (<n> − 1) / w²,  <x²> = Σ_n p_n x_n² + coth(β w / 2) / (2 w);  half filling at mu = 0."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import hmc, lattice as lat, models, preconditioners as pc

beta, dtau, w, lam = 2.0, 0.1, 1.0, 1.0
mu = float(os.environ.get("MU", "-0.3"))
E = [-mu * n - lam ** 2 * (n - 1) ** 2 / (2 * w ** 2) for n in (0, 1, 2)]
g = [1, 2, 1]
Z = sum(gi * np.exp(-beta * e) for gi, e in zip(g, E))
n_exact = sum(n * gi * np.exp(-beta * e) for n, gi, e in zip((0, 1, 2), g, E)) / Z
p_n = [gi * np.exp(-beta * e) / Z for gi, e in zip(g, E)]
x_exact = -lam * (n_exact - 1) / w ** 2
x2_exact = sum(p * (lam * (n - 1) / w ** 2) ** 2 for p, n in zip(p_n, (0, 1, 2))) + 1.0 / (2 * w * np.tanh(beta * w / 2))
m = models.HolsteinModel(lat.Lattice(1, 1, 1, 1), beta, dtau, tol=1e-10, maxiter=1000)
m.assign_omega_(w); m.assign_lambda_(lam); m.assign_mu_(mu)
m.initialize_model_()
m.x[:] = 0.5
models.update_model_(m)
fa = pc.FourierAccelerator(m)
pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.0)
nch = int(os.environ.get("NCH", "1"))
H = hmc.HybridMonteCarlo(m, fa, dt=0.1, tr=1.0, alpha=0.0, Nb=1, nchains=nch)
if nch > 1:
    H.X[:] = 0.5 * np.random.default_rng(5).standard_normal((nch, 1))
    H.push_()
H.device_rng_(int(os.environ.get("SEED", "1")))
nup = int(os.environ.get("NUP", "3000"))
xs, x2s, acc = [], [], 0
t0 = time.perf_counter()
for k in range(nup):
    if nch > 1:
        a, it = hmc.update_chains_(m, H, fa, None, pull=True)
        acc += a.mean()
        X = H.X
    else:
        a, it = hmc.update_(m, H, fa, None, pull=True)
        acc += a
        X = m.x[None]
    if k >= nup // 10:
        xs.append(X.mean())
        x2s.append(np.mean(X ** 2))
xs, x2s = np.array(xs), np.array(x2s)
nb = 20
bins = xs[:len(xs) // nb * nb].reshape(nb, -1).mean(axis=1)
b2 = x2s[:len(x2s) // nb * nb].reshape(nb, -1).mean(axis=1)
print(f"mu {mu:.3f}  exact <n> {n_exact:.4f} <x> {x_exact:.4f} <x2> {x2_exact:.4f} | HMC <x> {xs.mean():.4f} +- {bins.std(ddof=1) / np.sqrt(nb):.4f}  "
      f"<x2> {x2s.mean():.4f} +- {b2.std(ddof=1) / np.sqrt(nb):.4f}  acceptance {acc / nup:.3f}  "
      f"{time.perf_counter() - t0:.1f} s for {nup} updates")
m.close()
