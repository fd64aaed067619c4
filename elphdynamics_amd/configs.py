"""The BASELINE.json configurations (SURVEY.md §8 sizes table) as ready-made GPU models with the
deterministic synthetic phonon fields of synth.py.  Parameters follow the example decks:
t=1, omega=1, lambda=1, mu=0, dtau=0.1 (examples/holstein_hmc_square.toml:39-76);
SSH: t=1, alpha=0.1, omega=0.1, dtau=0.05 (examples/ssh_hmc_square.toml:39-58)."""
import numpy as np

from . import lattice as lat
from . import models, synth

CONFIGS = {
    # tag: (kind, norbits, Lspatial, bonds, beta, dtau)
    "A": ("holstein", 1, 1, [], 2.0, 0.1),                          # holstein_hmc_single_site.toml
    "B": ("holstein", 1, 8, lat.SQUARE_BONDS, 4.0, 0.1),
    "C": ("holstein", 1, 16, lat.SQUARE_BONDS, 16.0, 0.1),
    "D": ("holstein", 2, 12, lat.HONEYCOMB_BONDS, 12.0, 0.1),
    "E": ("ssh", 1, 16, lat.SQUARE_BONDS, 8.0, 0.05),
    # small variants for fast parity tests
    "b": ("holstein", 1, 4, lat.SQUARE_BONDS, 2.0, 0.1),
    "d": ("holstein", 2, 3, lat.HONEYCOMB_BONDS, 1.2, 0.1),
    "e": ("ssh", 1, 4, lat.SQUARE_BONDS, 1.0, 0.05),
    "e8": ("ssh", 1, 8, lat.SQUARE_BONDS, 1.0, 0.05),               # bond phonons beyond the two deck sizes (round 6): N = 64 — one site per lane
    "e12": ("ssh", 1, 12, lat.SQUARE_BONDS, 1.0, 0.05),             # N = 144 — three sites per lane
    "e20": ("ssh", 1, 20, lat.SQUARE_BONDS, 1.0, 0.05),             # N = 400 — seven sites per lane
    "e24": ("ssh", 1, 24, lat.SQUARE_BONDS, 1.0, 0.05),             # N = 576 — beyond the lane program: the generic LDS kernels
    "t": ("holstein", 1, 3, lat.TRIANGULAR_BONDS, 1.0, 0.125),      # odd L: 9 ragged colours (generic kernels)
    "u": ("holstein", 1, 4, lat.TRIANGULAR_BONDS, 1.0, 0.125),      # even-L triangular: 6 colours (lane program lp6)
    "T": ("holstein", 1, 16, lat.TRIANGULAR_BONDS, 16.0, 0.1),      # holstein_hmc_triangular.toml geometry at config-C size
    # even-L square lattices other than 8 and 16: the GRID register-exchange forms (2 x 2 patches on an L/2 x L/2 grid of lanes)
    "s": ("holstein", 1, 6, lat.SQUARE_BONDS, 2.0, 0.1),            # N = 36, Ltau = 20
    "q": ("holstein", 1, 10, lat.SQUARE_BONDS, 4.0, 0.1),           # N = 100, Ltau = 40
    "Q": ("holstein", 1, 14, lat.SQUARE_BONDS, 4.0, 0.1),           # N = 196, Ltau = 40
    "S": ("holstein", 1, 12, lat.SQUARE_BONDS, 16.0, 0.1),          # N = 144, Ltau = 160: config C's time axis on a 12 x 12 lattice
    # honeycomb lattices other than 12 x 12 cells: the HGRID register-exchange forms (1, 2 or 4 cells per lane on a grid of lanes)
    "y": ("holstein", 2, 6, lat.HONEYCOMB_BONDS, 2.0, 0.1),         # N = 72,  Ltau = 20   (one cell per lane)
    "z": ("holstein", 2, 10, lat.HONEYCOMB_BONDS, 4.0, 0.1),        # N = 200, Ltau = 40   (two cells per lane)
    "Y": ("holstein", 2, 16, lat.HONEYCOMB_BONDS, 4.0, 0.1),        # N = 512, Ltau = 40   (four cells per lane: 512 sites in one wave)
    # rectangular periodic lattices (Lspatial = (L1, L2)): the same forms on a rectangular grid of lanes — the shapes a sharded solve's
    # ring-closed slabs take, here as whole lattices
    "r": ("holstein", 1, (12, 6), lat.SQUARE_BONDS, 4.0, 0.1),      # N = 72,  Ltau = 40   (6 x 3 lanes)
    "R": ("holstein", 1, (8, 16), lat.SQUARE_BONDS, 4.0, 0.1),      # N = 128, Ltau = 40   (4 x 8 lanes)
    "w": ("holstein", 2, (6, 4), lat.HONEYCOMB_BONDS, 2.4, 0.1),    # N = 48,  Ltau = 24   (one cell per lane)
    "W": ("holstein", 2, (12, 8), lat.HONEYCOMB_BONDS, 2.4, 0.1),   # N = 192, Ltau = 24   (two cells per lane)
    # lattices beyond 512 sites: multi-wavefront workgroups of the generic kernels
    "g": ("holstein", 1, 24, lat.SQUARE_BONDS, 0.8, 0.1),           # N = 576  (2 wavefronts per slice)
    "G": ("holstein", 1, 32, lat.SQUARE_BONDS, 0.8, 0.1),           # N = 1024
    "h": ("holstein", 2, 18, lat.HONEYCOMB_BONDS, 0.6, 0.1),        # N = 648 honeycomb
    # even-L square lattices beyond 16 x 16 with a PGRID patch (pgrid_dev.h: the KPM recursion in registers); g (24) and G (32) above are two more
    "k": ("holstein", 1, 20, lat.SQUARE_BONDS, 1.0, 0.1),           # N = 400: 2 x 4 patches on 10 x 5 lanes
    "j": ("holstein", 1, 28, lat.SQUARE_BONDS, 0.6, 0.1),           # N = 784: 4 x 4 patches on 7 x 7 lanes
    "i": ("holstein", 1, 18, lat.SQUARE_BONDS, 0.6, 0.1),           # N = 324: 2 x 6 patches on 9 x 3 lanes
    "K": ("holstein", 1, 24, lat.SQUARE_BONDS, 4.0, 0.1),           # N = 576, Ltau = 40: long recursions (order ~ 50 at the lowest frequency)
    "l36": ("holstein", 1, 36, lat.SQUARE_BONDS, 0.6, 0.1),         # N = 1296: 4 x 6 patches on 9 x 6 lanes (round 6)
    "L36": ("holstein", 1, 36, lat.SQUARE_BONDS, 8.8, 0.1),         # N = 1296, Ltau = 88: long recursions, the fused preconditioned iteration (81 column tiles <= 88 slices)
    "G40": ("holstein", 1, 32, lat.SQUARE_BONDS, 4.0, 0.1),         # N = 1024, Ltau = 40: more column tiles (64) than time slices — the residual update's r.r slots per workgroup (round 6)
    # square lattices whose patches need several wavefronts per slice (round 6: pgrid_dev.h, pick_patch_mw) — l22 and l26 below are two more
    "l34": ("holstein", 1, 34, lat.SQUARE_BONDS, 0.6, 0.1),         # N = 1156: 2 x 2 patches on 5 wavefronts
    "l40": ("holstein", 1, 40, lat.SQUARE_BONDS, 0.6, 0.1),         # N = 1600: 4 x 4 patches on 2 wavefronts
    "l48": ("holstein", 1, 48, lat.SQUARE_BONDS, 0.6, 0.1),         # N = 2304: 4 x 4 patches on 3
    "l64": ("holstein", 1, 64, lat.SQUARE_BONDS, 0.4, 0.1),         # N = 4096: 4 x 4 patches on 4
    "L26": ("holstein", 1, 26, lat.SQUARE_BONDS, 4.8, 0.1),         # N = 676, Ltau = 48: long recursions on 3 wavefronts per slice, the fused iteration
    "L40": ("holstein", 1, 40, lat.SQUARE_BONDS, 10.0, 0.1),        # N = 1600, Ltau = 100
    "h30": ("holstein", 2, 30, lat.HONEYCOMB_BONDS, 0.6, 0.1),      # honeycomb 30 x 30 cells (N = 1800): 3 x 3 cells per thread on 2 wavefronts (round 6)
    "h22": ("holstein", 2, 22, lat.HONEYCOMB_BONDS, 0.6, 0.1),      # honeycomb 22 x 22 cells (N = 968): 2 x 2 cells on 2 wavefronts
    "H27": ("holstein", 2, 27, lat.HONEYCOMB_BONDS, 9.6, 0.1),      # honeycomb 27 x 27 cells (N = 1458), Ltau = 96: long recursions, the fused iteration (92 column tiles <= 96)
    "l22": ("holstein", 1, 22, lat.SQUARE_BONDS, 4.0, 0.1),         # N = 484, Ltau = 40: 22 = 2 x 11 has no single-wave patch — the generic LDS kernels (round 6: p/x-fused)
    "l26": ("holstein", 1, 26, lat.SQUARE_BONDS, 4.8, 0.1),         # N = 676, Ltau = 48
    "k40": ("holstein", 1, 20, lat.SQUARE_BONDS, 4.0, 0.1),         # N = 400, Ltau = 40: the lane-program family WITH the patch-form Chebyshev kernel — its p/x-fused iteration (round 6)
    "l30": ("holstein", 1, 30, lat.SQUARE_BONDS, 0.6, 0.1),         # N = 900: 2 x 10 patches on 15 x 3 lanes (round 5)
    # honeycomb lattices beyond 16 x 16 cells with a PGRID cell patch (h above, 18 x 18 cells: 3 x 2 cells per lane, is one more)
    "h20": ("holstein", 2, 20, lat.HONEYCOMB_BONDS, 0.6, 0.1),      # N = 800:  4 x 2 cells per lane on 5 x 10 lanes
    "h21": ("holstein", 2, 21, lat.HONEYCOMB_BONDS, 0.6, 0.1),      # N = 882:  3 x 3 cells on 7 x 7 lanes (odd L)
    "h24": ("holstein", 2, 24, lat.HONEYCOMB_BONDS, 0.6, 0.1),      # N = 1152: 3 x 3 cells on 8 x 8 lanes
    "H18": ("holstein", 2, 18, lat.HONEYCOMB_BONDS, 3.0, 0.1),      # N = 648, Ltau = 30: long recursions
    # triangular lattices in the patch layout (pgrid::Tri; t, u, T above are three more: t — odd L — keeps the generic kernels)
    "t6": ("holstein", 1, 6, lat.TRIANGULAR_BONDS, 1.0, 0.1),       # 2 x 2 patches on 3 x 3 lanes
    "t12": ("holstein", 1, 12, lat.TRIANGULAR_BONDS, 2.0, 0.1),     # N = 144, Ltau = 20
    "t20": ("holstein", 1, 20, lat.TRIANGULAR_BONDS, 1.0, 0.1),     # 2 x 4 patches (still the lane-program family for the mat-vec)
    "t24": ("holstein", 1, 24, lat.TRIANGULAR_BONDS, 0.8, 0.1),     # N = 576: 2 x 6 patches, generic family: mat-vec kernels in patches too
    "t32": ("holstein", 1, 32, lat.TRIANGULAR_BONDS, 0.6, 0.1),     # N = 1024: 4 x 4 patches
    # production-size lattices beyond the BASELINE ones, for bench.py's `large_lattices` record (PGRID kernels, csrc/pgrid.hip)
    "X32": ("holstein", 1, 32, lat.SQUARE_BONDS, 16.0, 0.1),        # square 32 x 32, Ltau = 160: 163 840 unknowns
    "X24": ("holstein", 2, 24, lat.HONEYCOMB_BONDS, 12.0, 0.1),     # honeycomb 24 x 24 cells, Ltau = 120: 138 240 unknowns
    "XT24": ("holstein", 1, 24, lat.TRIANGULAR_BONDS, 16.0, 0.1),   # triangular 24 x 24, Ltau = 160: 92 160 unknowns
    # a long time axis: 1280 slices (beyond the direct-DFT tables: dft_big.hip)
    "l": ("holstein", 1, 4, lat.SQUARE_BONDS, 128.0, 0.1),
    "l800": ("holstein", 1, 4, lat.SQUARE_BONDS, 80.0, 0.1),        # 800 time slices: between the matrix-core transforms (<= 400) and 1024 — the Cooley-Tukey split since round 6
}


def make_model(tag, tol=1e-5, maxiter=10000, rough=True, device=0, seed=synth.SEED_FIELDS, t_stddev=0.0):
    kind, norb, Ls, bonds, beta, dtau = CONFIGS[tag]
    L1, L2 = Ls if isinstance(Ls, tuple) else (Ls, Ls if Ls > 1 else 1)
    lattice = lat.Lattice(norb, L1, L2, 1)
    if kind == "holstein":
        m = models.HolsteinModel(lattice, beta, dtau, tol=tol, maxiter=maxiter, device=device)
        for (o1, o2, d) in bonds:
            m.assign_t_(1.0, o1, o2, d, stddev=t_stddev, rng=np.random.default_rng(seed + 991))   # (t_stddev: hopping disorder, :427-447)
        m.assign_omega_(1.0)
        m.assign_lambda_(1.0)
        m.assign_mu_(0.0)
        m.initialize_model_()
        m.x[:] = synth.phonon_field(m.Nph, m.Ltau, beta, dtau, omega=1.0, lam=1.0, rough=rough, seed=seed)
    else:
        m = models.SSHModel(lattice, beta, dtau, tol=tol, maxiter=maxiter, device=device)
        for (o1, o2, d) in bonds:
            m.assign_hopping_(1.0, 0.1, 0.0, 0.1, o1, o2, d, name="xyz"[d.index(1)])     # names "x", "y" as in the deck (:46,:62)
        m.initialize_model_()
        m.x[:] = synth.phonon_field(m.Nph, m.Ltau, beta, dtau, omega=0.1, lam=0.0, rough=rough, seed=seed)
        # keep |alpha x| < t (SSHModels.jl:537-539 warns beyond that): the omega=0.1 QHO is wide
        m.x *= 0.25
    models.update_model_(m)
    return m


def rhs(model, nrhs=1, seed=synth.SEED_RHS):
    """b = Mt R for i.i.d. N(0,1) R (GreensFunctions.jl:212-225); returns (R, b) as (nrhs, Ndim) arrays."""
    R = np.stack([synth.rhs(model.Ndim, seed=seed + 7919 * i) for i in range(nrhs)])
    B = np.empty_like(R)
    for i in range(nrhs):
        models.mulMt_(B[i], model, R[i])
    return R, B
