"""End-to-end physics check of the bond-phonon path: the two-site SSH model (the reference's ssh_hmc_two_site deck) is exactly
solvable — the hopping operator K = Σ_σ (c†₁σ c₂σ + h.c.) commutes with H = −(t − α x) K − μ N + p²/2 + ω² x²/2, so every
(N, k) sector is a displaced oscillator:  E = −t k − μ N − α² k² / (2 ω²),  <x> = −α <k> / ω²,
<x²> = Σ p x_k² + coth(β ω / 2) / (2 ω)."""
import itertools, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import hmc, lattice as lat, models, preconditioners as pc

beta, dtau, w, t, alpha = 2.0, 0.1, 1.0, 1.0, float(os.environ.get("ALPHA", "0.8"))
mu = float(os.environ.get("MU", "0.0"))
one = [(0, 0), (1, 1), (1, -1), (2, 0)]                       # (N, k) of one spin species: empty, bonding, antibonding, full
Z = n_av = k_av = k2_av = 0.0
for (N1, k1), (N2, k2) in itertools.product(one, one):
    N, k = N1 + N2, k1 + k2
    wgt = np.exp(-beta * (-t * k - mu * N - alpha ** 2 * k ** 2 / (2 * w ** 2)))
    Z += wgt; n_av += N * wgt; k_av += k * wgt; k2_av += k * k * wgt
n_av, k_av, k2_av = n_av / Z, k_av / Z, k2_av / Z
x_exact = -alpha * k_av / w ** 2
x2_exact = alpha ** 2 * k2_av / w ** 4 + 1.0 / (2 * w * np.tanh(beta * w / 2))
nch = int(os.environ.get("NCH", "64"))
m = models.SSHModel(lat.Lattice(1, 2, 1, 1), beta, dtau, tol=1e-10, maxiter=1000)
m.assign_hopping_(t, alpha, 0.0, w, 1, 1, (1, 0, 0), name="b")
m.initialize_model_()
m.mu[:] = mu
assert m.Nbonds == 1 and m.Nph == 1
models.update_model_(m)
fa = pc.FourierAccelerator(m)
pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.0)
H = hmc.HybridMonteCarlo(m, fa, dt=0.1, tr=1.0, alpha=0.0, Nb=1, nchains=nch)
H.X[:] = 0.3 * np.random.default_rng(5).standard_normal((nch, 1))
H.push_()
H.device_rng_(int(os.environ.get("SEED", "1")))
nup = int(os.environ.get("NUP", "1500"))
xs, x2s, acc = [], [], 0.0
t0 = time.perf_counter()
for kk in range(nup):
    a, it = hmc.update_chains_(m, H, fa, None, pull=True)
    acc += a.mean()
    if kk >= nup // 10:
        xs.append(H.X.mean()); x2s.append(np.mean(H.X ** 2))
xs, x2s = np.array(xs), np.array(x2s)
nb = 20
err = lambda v: v[:len(v) // nb * nb].reshape(nb, -1).mean(axis=1).std(ddof=1) / np.sqrt(nb)
print(f"mu {mu:.2f} alpha {alpha}: exact <N> {n_av:.4f} <k> {k_av:.4f} <x> {x_exact:.4f} <x2> {x2_exact:.4f} | HMC <x> {xs.mean():.4f} +- {err(xs):.4f} "
      f"<x2> {x2s.mean():.4f} +- {err(x2s):.4f}  acceptance {acc / nup:.3f}  {time.perf_counter() - t0:.1f} s")
m.close()
