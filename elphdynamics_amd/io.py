"""On-disk interop with runs of the reference: phonon configurations and the M-matrix dump.

Text formats, byte-for-byte those of the reference so that a configuration written by a Julia run elsewhere can be
replayed through the GPU path (and the other way round):

    write_phonons_(model, filename) / read_phonons_(model, filename)
        Holstein  "L3 L2 L1 orbit tau x"   one line per (cell, orbit, tau), "%d %d %d %d %d %.6f"
                  HolsteinModels.jl:764-805 (write), :810-852 (read, ends with update_model!)
        SSH       "type loc tau x"         one line per (phonon type, bond of that type, tau), "%d %d %d %.6f"
                  SSHModels.jl:838-868 (write, nothing written when Nph == 0), :873-913 (read)
    construct_M(model, threshold=1e-14)  -> rows, cols, vals (1-based)       Models.jl:300-341
    write_M_matrix_(model, filename, threshold=1e-10)   "col row real imag", "%d %d %.10f %.10f"   Models.jl:347-367
    read_M_matrix(filename) -> rows, cols, vals          (reader for dumps of a Julia run; the reference has none)

All indices in the files are the reference's: cells 0-based, orbit / tau / bond / matrix indices 1-based.
"""
import numpy as np

from . import models


def _is_holstein(model):
    return model.kind == models.HOLSTEIN


def write_phonons_(model, filename):
    L = model.Ltau
    if _is_holstein(model):
        lat = model.lattice
        x = np.asarray(model.x)
        with open(filename, "w") as f:
            f.write("L3 L2 L1 orbit tau x\n")
            for l3 in range(lat.L3):
                for l2 in range(lat.L2):
                    for l1 in range(lat.L1):
                        for orbit in range(1, lat.norbits + 1):
                            site = lat.loc_to_site(orbit, l1, l2, l3)
                            base = (site - 1) * L                      # get_index(τ, site, Lτ), Utilities.jl:12-15
                            f.write("".join("%d %d %d %d %d %.6f\n" % (l3, l2, l1, orbit, tau, x[base + tau - 1])
                                            for tau in range(1, L + 1)))
        return
    if model.Nph > 0:                                                  # SSHModels.jl:840
        n = _ssh_ntypes(model)
        N = model.Nph // n
        x = np.asarray(model.x).reshape(n, N, L)                       # Julia (L, N, n) column-major
        with open(filename, "w") as f:
            f.write("type loc tau x\n")
            for phonon in range(1, n + 1):
                for i in range(1, N + 1):
                    f.write("".join("%d %d %d %.6f\n" % (phonon, i, tau, x[phonon - 1, i - 1, tau - 1])
                                    for tau in range(1, L + 1)))


def read_phonons_(model, filename):
    """Assigns model.x from the file (entries not named in the file keep their value, as in the reference) and calls
    update_model! — HolsteinModels.jl:849, SSHModels.jl:910."""
    L = model.Ltau
    x = np.array(model.x, dtype=np.float64).reshape(-1)
    with open(filename, "r") as f:
        f.readline()                                                   # header
        if _is_holstein(model):
            lat = model.lattice
            for line in f:
                a = line.rstrip("\n").split(" ")
                l3, l2, l1, orbit, tau = int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4])
                site = lat.loc_to_site(orbit, l1, l2, l3)
                if not (1 <= tau <= L and 1 <= orbit <= lat.norbits):
                    raise IndexError("phonon file entry outside the model: %r" % line)   # Julia: BoundsError
                x[(site - 1) * L + tau - 1] = float(a[5])
        else:
            n = _ssh_ntypes(model)
            N = model.Nph // n if n else 0
            for line in f:
                a = line.rstrip("\n").split(" ")
                phonon, cell, tau = int(a[0]), int(a[1]), int(a[2])
                if not (1 <= phonon <= n and 1 <= cell <= N and 1 <= tau <= L):
                    raise IndexError("phonon file entry outside the model: %r" % line)
                x[((phonon - 1) * N + cell - 1) * L + tau - 1] = float(a[3])
    model.x = x
    models.update_model_(model)


def _ssh_ntypes(model):
    """ssh.nph: number of bond definitions that carry a phonon (SSHModels.jl:845,877)."""
    return sum(1 for d in model.bond_definitions if d["has_phonon"])


def construct_M(model, threshold=1e-14):
    """Models.jl:300-341: column `col` of M is M·e_col (one device mat-vec per column); entries with |.| > threshold."""
    n = model.Ndim
    rows, cols, vals = [], [], []
    unit = np.zeros(n)
    colv = np.zeros(n)
    for col in range(n):
        unit[col - 1] = 0.0
        unit[col] = 1.0
        models.mulM_(colv, model, unit)
        nz = np.nonzero(np.abs(colv) > threshold)[0]
        rows.append(nz + 1)
        cols.append(np.full(nz.size, col + 1, dtype=np.int64))
        vals.append(colv[nz].copy())
    return (np.concatenate(rows).astype(np.int64), np.concatenate(cols), np.concatenate(vals))


def write_M_matrix_(model, filename, threshold=1e-10):
    rows, cols, vals = construct_M(model, threshold)
    with open(filename, "w") as f:
        f.write("col row real imag\n")
        f.write("".join("%d %d %.10f %.10f\n" % (c, r, v, 0.0) for r, c, v in zip(rows, cols, vals)))


def read_M_matrix(filename):
    """Reads a `write_M_matrix!` dump: returns 1-based rows, cols and (real) values."""
    rows, cols, vals = [], [], []
    with open(filename, "r") as f:
        f.readline()
        for line in f:
            a = line.split()
            cols.append(int(a[0])); rows.append(int(a[1])); vals.append(float(a[2]))
    return np.array(rows, dtype=np.int64), np.array(cols, dtype=np.int64), np.array(vals)
