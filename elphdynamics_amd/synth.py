"""Deterministic synthetic inputs (phonon fields, right-hand sides) for tests and bench.

The reference draws from Julia's Xoshiro stream (ProcessInputFile.jl:598), which cannot be
reproduced here, so every parity run takes explicit arrays.  This module is the build's own
counter-based generator: SplitMix64 -> uniform(0,1) -> Box-Muller, pure numpy, identical on
every box.  Distributions follow the reference's initialisation:

  cold start  (InitializePhonons.jl:71-115): per site, tau-constant
              x = (lambda/omega^2) * u + sigma * g,  u in {-1,0,1},  g ~ N(0,1),
              sigma = 1/sqrt(2 omega tanh(beta omega/2))
  rough       cold start + i.i.d. N(0, dtau) per (site, tau)
"""
import numpy as np

SEED_FIELDS = 20260131
SEED_RHS = 20260132

_MASK = (1 << 64) - 1


def splitmix64(seed, n):
    """n 64-bit outputs of SplitMix64 started at `seed` (vectorised, counter-based)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed & _MASK) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed, n):
    """n doubles in (0,1): top 53 bits + half-ulp offset so that log() is finite."""
    u = (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64)
    return (u + 0.5) * (1.0 / 9007199254740992.0)


def randn(seed, n):
    """n standard normals by Box-Muller on 2*ceil(n/2) uniforms."""
    m = (n + 1) // 2
    u = uniform01(seed, 2 * m)
    r = np.sqrt(-2.0 * np.log(u[:m]))
    th = 2.0 * np.pi * u[m:]
    out = np.empty(2 * m)
    out[0::2] = r * np.cos(th)
    out[1::2] = r * np.sin(th)
    return np.ascontiguousarray(out[:n])


def batch_seed(seed, b):
    """Seed of batch b (1-based) of the library's generator (elph_hmc_set_rng): output b of SplitMix64(seed)."""
    return int(splitmix64(seed, b)[b - 1])


def phonon_field(nph, ltau, beta, dtau, omega=1.0, lam=1.0, rough=True, seed=SEED_FIELDS):
    """Flat x[nph*ltau], tau fastest (Utilities.jl:12-15)."""
    u = (splitmix64(seed ^ 0x5151, nph) % np.uint64(3)).astype(np.int64) - 1
    g = randn(seed ^ 0xA0A0, nph)
    sigma = 1.0 / np.sqrt(2.0 * omega * np.tanh(beta * omega / 2.0)) if omega > 0 else 1.0
    x0 = (lam / omega ** 2) * u + sigma * g
    x = np.repeat(x0, ltau)
    if rough:
        x = x + np.sqrt(dtau) * randn(seed ^ 0x0F0F, nph * ltau)
    return np.ascontiguousarray(x)


def rhs(ndim, seed=SEED_RHS):
    """i.i.d. N(0,1) right-hand side R (GreensFunctions.jl:212)."""
    return randn(seed, ndim)
