"""Host-side mirror of the correlation-function measurements of Measurements.jl that consume the Green's-function tables the GPU
path produces: Greens, DenDen, SpinSpin, PairGreens, PhononGreens (Holstein), the global and on-site
Holstein observables, and the pair loop of make_measurements!.  Binning, momentum-space transforms, susceptibilities and the
file formats of the reference's measurement container are not restated (control plane).

    scalar, literal restatements (one displacement):       measure_Greens / _DenDen / _SpinSpin / _PairGreens       Measurements.jl:1469-1558
    whole tables at once (what measure_X! accumulates):    correlation_(container, pairs, est, kind)                :1561-1593
    measure_PhononGreens_(container, pairs, model)                                                                  :1598-1653
    global_measurements(model, est), onsite_measurements_holstein / _ssh(model, est)                                :845-862, :916-1023
    make_measurements_(acc, model, est, P, rng=None, R=None, kinds=…)                                               :545-569

Containers are complex arrays (L+1, L1, L2, L3, n_pairs): time displacement 0..β, cell displacement, orbital pair — the
reference's `position` arrays.  Orbitals are 1-based, displacements 0-based, as in greens.py.
"""
import numpy as np

from elphdynamics_amd import greens, models
from mu_tuner import measure_N2, measure_density

KINDS = ("Greens", "DenDen", "SpinSpin", "PairGreens")


def _d(*a):
    return 1.0 if all(v == 0 for v in a) else 0.0


# ---------------------------------------------------------------------------------------------- literal scalar forms

def measure_Greens(model, est, l1, l2, l3, o1, o2, tau):
    L = model.Ltau
    g = greens.measure_GD0(est, l1, l2, l3, o1, o2, tau % L)
    if tau == L:                                                          # G_r(β) = δ_r − G_r(0)
        g = _d(l1, l2, l3) * (1.0 if o1 == o2 else 0.0) - g
    return g


def measure_DenDen(model, est, l1, l2, l3, o1, o2, tau):
    L = model.Ltau
    t = tau % L
    Gr0 = greens.measure_GD0(est, l1, l2, l3, o1, o2, t)
    G00 = greens.measure_GD0(est, 0, 0, 0, o1, o1, 0)
    Grr = greens.measure_GD0(est, 0, 0, 0, o2, o2, 0)
    GrrG00 = greens.measure_GDD_G00(est, l1, l2, l3, o1, o2, t)
    Gr0G0r = greens.measure_GD0_G0D(est, l1, l2, l3, o1, o2, t)
    dr = _d(l1, l2, l3) * (1.0 if o1 == o2 else 0.0)
    return 4.0 * (1.0 - Grr - G00 + GrrG00 + 0.5 * (dr * _d(t) * Gr0 - Gr0G0r))


def measure_SpinSpin(model, est, l1, l2, l3, o1, o2, tau):
    L, lat = model.Ltau, model.lattice
    if tau == L:                                                          # ⟨s(i+r, β) s(i, 0)⟩ = ⟨s(i−r, 0) s(i, 0)⟩
        tau, o1, o2 = 0, o2, o1
        l1, l2, l3 = (-l1) % lat.L1, (-l2) % lat.L2, (-l3) % lat.L3
    Gr0G0r = greens.measure_GD0_G0D(est, l1, l2, l3, o1, o2, tau)
    Gr0 = greens.measure_GD0(est, l1, l2, l3, o1, o2, tau)
    dr = _d(l1, l2, l3) * (1.0 if o1 == o2 else 0.0)
    return -2 * Gr0G0r + 2 * dr * _d(tau) * Gr0


def measure_PairGreens(model, est, l1, l2, l3, o1, o2, tau):
    L = model.Ltau
    if tau == L:                                                          # P_r(β) = P_r(0) + δ_r (1 − G↑₀(0) − G↓₀(0))
        p = greens.measure_GD0_GD0(est, l1, l2, l3, o1, o2, 0)
        if l1 == 0 and l2 == 0 and l3 == 0 and o1 == o2:
            p = p + 1.0 - 2 * greens.measure_GD0(est, 0, 0, 0, o1, o1, 0)
        return p
    return greens.measure_GD0_GD0(est, l1, l2, l3, o1, o2, tau)


_SCALAR = dict(Greens=measure_Greens, DenDen=measure_DenDen, SpinSpin=measure_SpinSpin, PairGreens=measure_PairGreens)


# ---------------------------------------------------------------------------------------------- whole tables

def correlation_(container, pairs, model, est, kind):
    """container[τ, l1, l2, l3, p] += measure_<kind>(model, est, l1, l2, l3, o1, o2, τ) for every τ = 0..Lτ, displacement and
    orbital pair (o1, o2) = pairs[p] — the generated measure_<kind>! of the reference, on whole tables."""
    L = model.Ltau
    lat = model.lattice
    z = (0, 0, 0)
    for p, (o1, o2) in enumerate(pairs):
        a, b = o2 - 1, o1 - 1                                             # table axes are [τ, o₂, o₁, l1, l2, l3]
        same = 1.0 if o1 == o2 else 0.0
        dr = np.zeros((lat.L1, lat.L2, lat.L3))
        dr[0, 0, 0] = same
        GD0 = est.GD0[:L, a, b]                                           # (L, L1, L2, L3), time displacement 0..L−1
        out = np.zeros((L + 1, lat.L1, lat.L2, lat.L3), dtype=np.complex128)
        if kind == "Greens":
            out[:L] = GD0
            out[L] = dr - GD0[0]
        elif kind == "DenDen":
            G00 = est.GD0[(0, o1 - 1, o1 - 1) + z]
            Grr = est.GD0[(0, o2 - 1, o2 - 1) + z]
            body = 1.0 - Grr - G00 + est.GDD_G00[:L, a, b] - 0.5 * est.GD0_G0D[:L, a, b]
            body[0] += 0.5 * dr * GD0[0]
            out[:L] = 4.0 * body
            out[L] = out[0]                                               # τ % L: the β slice repeats the equal-time value
        elif kind == "SpinSpin":
            out[:L] = -2 * est.GD0_G0D[:L, a, b]
            out[0] += 2 * dr * GD0[0]
            # τ = β: τ → 0, orbitals exchanged, displacement negated
            g = -2 * est.GD0_G0D[0, b, a] + 2 * dr * est.GD0[0, b, a]
            idx = np.ix_((-np.arange(lat.L1)) % lat.L1, (-np.arange(lat.L2)) % lat.L2, (-np.arange(lat.L3)) % lat.L3)
            out[L] = g[idx]
        elif kind == "PairGreens":
            out[:L] = est.GD0_GD0[:L, a, b]
            out[L] = est.GD0_GD0[0, a, b] + dr * (1.0 - 2 * est.GD0[(0, o1 - 1, o1 - 1) + z])
        else:
            raise ValueError(kind)
        container[..., p] += out


def translational_average(f, g):
    """Σ_i f[i + r] g[i], periodic in every axis; the caller divides by the number of elements (Utilities.jl:49-60:
    ifft(fft(f) · reverse(circshift(fft(g), size − 1)) / N) — the reversed transform is fft(g) at the negated frequency)."""
    gh = np.fft.fftn(g)
    idx = np.ix_(*[(-np.arange(n)) % n for n in gh.shape])
    return np.fft.ifftn(np.fft.fftn(f) * gh[idx])


def measure_PhononGreens_(container, pairs, model):
    """⟨x_{i+r}(τ' + τ) x_i(τ')⟩ averaged over i and τ' for Holstein phonons (Measurements.jl:1598-1653)."""
    assert model.kind == models.HOLSTEIN
    lat, L = model.lattice, model.Ltau
    x = model.x.reshape(lat.L3, lat.L2, lat.L1, lat.norbits, L).transpose(4, 3, 2, 1, 0)      # (τ, o, l1, l2, l3)
    for p, (o1, o2) in enumerate(pairs):
        xx = translational_average(x[:, o1 - 1].astype(np.complex128), x[:, o2 - 1]) / x[:, 0].size
        if container.shape[0] == 1:
            container[0, ..., p] += xx[0]
        else:
            container[:L, ..., p] += xx
            container[L, ..., p] += xx[0]


# ---------------------------------------------------------------------------------------------- global / on-site

def global_measurements(model, est):
    """make_global_measurements! (:845-862) for the pair selected by greens.setup_."""
    return dict(density=measure_density(est), Nsqr=measure_N2(model, est), mu=float(np.mean(model.mu)))


def onsite_measurements_holstein(model, est):
    """make_onsite_measurements! for the Holstein model (:916-973): per orbital type, averaged over sites and time slices."""
    lat, L, N = model.lattice, model.Ltau, model.Nsites
    R1, R2, X1, X2 = (a.reshape(N, L) for a in (est.r1, est.r2, est.Minvr1, est.Minvr2))
    G1, G2 = X1 * R1, X2 * R2                                              # estimate(Gr, site, site, τ, τ, σ)  (:334-346)
    x = model.x.reshape(N, L)
    dx = np.roll(x, -1, axis=1) - x
    dtau = model.dtau
    out = {}
    per_site = dict(density=(1 - G1) + (1 - G2), double_occ=(1 - G1) * (1 - G2), phonon_ke=0.5 / dtau - dx ** 2 / dtau ** 2 / 2,
                    phonon_pe=model.omega[:, None] ** 2 * x ** 2 / 2 + model.omega4[:, None] * x ** 4,
                    elph_energy=model.lam[:, None] * x * (2.0 - G1 - G2), x=x, x2=x ** 2, x4=x ** 4,
                    mu=np.repeat(model.mu[:, None], L, axis=1))
    for k, v in per_site.items():
        out[k] = np.array([v[o::lat.norbits].mean() for o in range(lat.norbits)])
    return out


def onsite_measurements_ssh(model, est):
    """make_onsite_measurements! for the SSH model (:978-1023): density, double occupancy and μ per orbital type."""
    lat, L, N = model.lattice, model.Ltau, model.Nsites
    R1, R2, X1, X2 = (a.reshape(N, L) for a in (est.r1, est.r2, est.Minvr1, est.Minvr2))
    G1, G2 = X1 * R1, X2 * R2
    per_site = dict(density=(1 - G1) + (1 - G2), double_occ=(1 - G1) * (1 - G2), mu=np.repeat(model.mu[:, None], L, axis=1))
    return {k: np.array([v[o::lat.norbits].mean() for o in range(lat.norbits)]) for k, v in per_site.items()}


def new_accumulator(model, kinds=KINDS, pairs=None, phonon_greens=True):
    lat = model.lattice
    pairs = pairs or [(o1, o2) for o1 in range(1, lat.norbits + 1) for o2 in range(o1, lat.norbits + 1)]
    shape = (model.Ltau + 1, lat.L1, lat.L2, lat.L3, len(pairs))
    acc = dict(pairs=pairs, n=0, corr={k: np.zeros(shape, dtype=np.complex128) for k in kinds}, glob=dict(density=0.0, Nsqr=0.0, mu=0.0),
               onsite=None)
    if phonon_greens and model.kind == models.HOLSTEIN:
        acc["corr"]["PhononGreens"] = np.zeros(shape, dtype=np.complex128)
    return acc


def make_measurements_(acc, model, est, P=None, rng=None, R=None):
    """make_measurements! (:545-569): fresh noise vectors and solves (one batched CG on the GPU), then every pair of noise
    vectors contributes to every requested correlation function.  acc["n"] counts the contributions; divide by it."""
    greens.update_(est, model, P, rng=rng, R=R)
    for i in range(1, est.nv):
        for j in range(i + 1, est.nv + 1):
            greens.setup_(est, i, j)
            g = global_measurements(model, est)
            for k in acc["glob"]:
                acc["glob"][k] += g[k]
            for kind in KINDS:
                if kind in acc["corr"]:
                    correlation_(acc["corr"][kind], acc["pairs"], model, est, kind)
            if "PhononGreens" in acc["corr"]:
                measure_PhononGreens_(acc["corr"]["PhononGreens"], acc["pairs"], model)
            o = onsite_measurements_holstein(model, est) if model.kind == models.HOLSTEIN else onsite_measurements_ssh(model, est)
            acc["onsite"] = o if acc["onsite"] is None else {k: acc["onsite"][k] + o[k] for k in o}
            acc["n"] += 1
    return acc
