"""Lattice geometry, neighbour tables and the checkerboard ordering (host-side set-up).

Mirrors the reference's L0 layer for the functions the hot path depends on:
Lattices.jl:56-107 (site numbering), :149-191 (loc_to_site/site_to_site), :265-316
(calc_neighbor_table), :323-340 (sorted_neighbor_table_perm!), Checkerboard.jl:442-446,471-515
(checkerboard_groups!/checkerboard_order!), HolsteinModels.jl:484-517 (initialize_model!).

All site numbers are 1-based int64 exactly as in the reference, so that the tables can be compared
bit-for-bit with dumps of a Julia run; the C ABI (include/elph_gpu.h) takes them in that form.
This is integer set-up work done once per model (O(Nbonds^2) like the reference) and stays on the
host; the vectors it indexes live on the GPU.
"""
import numpy as np


class Lattice:
    """Lattices.jl:16-109: norbits orbitals per cell, L1 x L2 x L3 cells, site = norbits*cell + orbit."""

    def __init__(self, norbits, L1, L2=1, L3=1):
        assert norbits >= 1 and L1 >= 1 and L2 >= 1 and L3 >= 1
        self.norbits, self.L1, self.L2, self.L3 = int(norbits), int(L1), int(L2), int(L3)
        self.ncells = self.L1 * self.L2 * self.L3
        self.nsites = self.ncells * self.norbits

    def loc_to_site(self, orbit, l1, l2=0, l3=0):
        """1-based site of `orbit` (1-based) in the cell at (l1,l2,l3), periodic (Lattices.jl:149-168,384-391)."""
        cell = (l1 % self.L1) + (l2 % self.L2) * self.L1 + (l3 % self.L3) * self.L1 * self.L2
        return self.norbits * cell + orbit

    def site_to_site(self, isite, displacement, orbit):
        """Lattices.jl:176-191."""
        cell = (isite - 1) // self.norbits
        l1, l2, l3 = cell % self.L1, (cell // self.L1) % self.L2, cell // (self.L1 * self.L2)
        return self.loc_to_site(orbit, l1 + displacement[0], l2 + displacement[1], l3 + displacement[2])

    def calc_neighbor_table(self, orbit1, orbit2, displacement, remove_duplicates=True):
        """Lattices.jl:265-316 -> int64 array (nbonds, 2); row n is Julia column n."""
        pairs = []
        for isite in range(orbit1, self.nsites + 1, self.norbits):
            pairs.append((isite, self.site_to_site(isite, displacement, orbit2)))
        if remove_duplicates:
            keep = [True] * len(pairs)
            for i in range(len(pairs) - 1):
                if not keep[i]:
                    continue
                a, b = pairs[i]
                for j in range(i + 1, len(pairs)):
                    a2, b2 = pairs[j]
                    if (a == a2 and b == b2) or (a == b2 and b == a2):
                        keep[j] = False
            pairs = [p for p, k in zip(pairs, keep) if k]
        return np.array(pairs, dtype=np.int64).reshape(-1, 2)


def sorted_neighbor_table_perm(table):
    """Lattices.jl:323-340: orient rows in place (smaller site first), return the 0-based stable sort perm."""
    swap = table[:, 0] > table[:, 1]
    table[swap] = table[swap][:, ::-1]
    vals = table.max() * table[:, 0] + table[:, 1]
    return np.argsort(vals, kind="stable")


def checkerboard_groups(table):
    """Checkerboard.jl:471-515: greedy colouring of a sorted table; returns 1-based colours."""
    nb = table.shape[0]
    groups = np.zeros(nb, dtype=np.int64)
    group = 0
    nassigned = 0
    while nassigned < nb:
        group += 1
        members = []            # bonds already in this colour (all earlier-indexed than the candidate)
        for n in range(nb):
            if groups[n] != 0:
                continue
            i, j = table[n]
            if any(i == table[p, 0] or j == table[p, 1] or i == table[p, 1] or j == table[p, 0] for p in members):
                continue
            groups[n] = group
            members.append(n)
            nassigned += 1
    return groups


def checkerboard_order(groups):
    """Checkerboard.jl:442-446: stable sortperm of the colours (0-based)."""
    return np.argsort(groups, kind="stable")


def initialize_checkerboard(table, t=None, dtau=None):
    """HolsteinModels.jl:484-517 / SSHModels.jl:435-448.

    table: raw (nbonds,2) 1-based.  Returns dict with the final table, colours, checkerboard_perm and
    inv_checkerboard_perm (both 1-based) and, if `t` is given, cosht/sinht in final order."""
    table = np.array(table, dtype=np.int64, copy=True).reshape(-1, 2)
    nb = table.shape[0]
    if nb == 0:
        z = np.zeros(0, dtype=np.int64)
        return dict(table=table, colours=z, cb_perm=z, inv_cb_perm=z, cosht=np.zeros(0), sinht=np.zeros(0), ncolours=0)
    perm = sorted_neighbor_table_perm(table)
    table = table[perm]
    groups = checkerboard_groups(table)
    new_perm = checkerboard_order(groups)
    table = np.ascontiguousarray(table[new_perm])
    inv_cb_perm = perm[new_perm]
    cb_perm = np.argsort(inv_cb_perm, kind="stable")
    out = dict(table=table, colours=groups[new_perm], cb_perm=cb_perm + 1, inv_cb_perm=inv_cb_perm + 1,
               ncolours=int(groups.max()))
    if t is not None:
        t = np.asarray(t, dtype=np.float64)
        out["cosht"] = np.ascontiguousarray(np.cosh(dtau * t)[inv_cb_perm])
        out["sinht"] = np.ascontiguousarray(np.sinh(dtau * t)[inv_cb_perm])
    return out


def ltau_from_beta(beta, dtau):
    """HolsteinModels.jl:205: Ltau = round(Int, beta/dtau), ties to even (Python round() is ties-to-even too)."""
    return int(round(beta / dtau))


# bond definitions of the example decks (orbit1, orbit2, displacement), in deck order
SQUARE_BONDS = [(1, 1, (1, 0, 0)), (1, 1, (0, 1, 0))]                           # examples/holstein_hmc_square.toml
HONEYCOMB_BONDS = [(1, 2, (0, 0, 0)), (1, 2, (-1, 0, 0)), (1, 2, (0, -1, 0))]   # examples/holstein_hmc_honeycomb.toml
TRIANGULAR_BONDS = [(1, 1, (1, 0, 0)), (1, 1, (0, 1, 0)), (1, 1, (1, -1, 0))]   # examples/holstein_hmc_triangular.toml
