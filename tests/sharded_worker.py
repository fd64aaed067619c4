"""Worker of the 2-rank sharded-CG tests (launched once per rank by tests/test_sharded_*.py).

usage: sharded_worker.py numpy[-spatial] | collectives <out.npz>
  numpy    : local compute by a numpy/oracle stand-in backend (CPU, gloo) — TEST-ONLY code path (tests/protocol_reference.py)
  -spatial : slabs of rows of cells (SpatialShardedCG) instead of tau-slabs (ShardedCG)
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from elphdynamics_amd import dist, synth  # noqa: E402
import protocol_reference as sharded  # noqa: E402
from elphdynamics_amd import lattice as lat  # noqa: E402


class NumpyBackend:
    """Local back end with the phase semantics of k_cg_init / k_cg_state0 / k_cg_ap / k_cg_xr, local
    operator applied by the CPU oracle.  Test infrastructure only."""

    def __init__(self, N, L2, table, c, s):
        from oracle.oracle import Oracle
        self.orc = Oracle()
        self.N, self.L, self.table, self.c, self.s = N, L2, table, c, s
        self.buf = {k: np.zeros(L2) for k in (sharded.PAP, sharded.RR, sharded.BB)}
        self.mask = np.ones(N)          # sites that enter the inner products (elph_set_dot_range)

    def set_dot_range(self, lo, hi):
        self.mask = np.zeros(self.N)
        self.mask[lo:hi] = 1.0

    def _dot(self, a, b):            # per-slice partial sums over the masked sites
        return (self._slices(a) * self._slices(b) * self.mask[:, None]).sum(axis=0)

    def read_rows(self, which, site_lo, nsites):
        return self._slices(self._vec(which))[site_lo:site_lo + nsites, :].T.copy()

    def write_rows(self, which, site_lo, values):
        self._slices(self._vec(which))[site_lo:site_lo + values.shape[1], :] = np.asarray(values).T

    def set_expV(self, E_loc):
        self.m = self.orc.make_model(0, self.N, self.L, self.table, self.c, self.s, np.ascontiguousarray(E_loc).reshape(-1))

    def _slices(self, v):            # reference layout (N, L) -> per-slice view
        return v.reshape(self.N, self.L)

    def begin(self, b_loc, tol, maxiter, kmax):
        self.tol, self.maxiter, self.kmax = tol, maxiter, kmax
        self.b = np.ascontiguousarray(b_loc).reshape(-1).copy()
        self.x = np.zeros_like(self.b)
        self.r = self.b.copy()
        self.p = self.b.copy()
        self.buf[sharded.RR] = self._dot(self.r, self.r)
        self.buf[sharded.BB] = self._dot(self.b, self.b)

    def state0(self):
        rr, bb = self.buf[sharded.RR].sum(), self.buf[sharded.BB].sum()
        self.normb = np.sqrt(bb)
        self.eps0 = self.eps = np.sqrt(rr) / self.normb
        self.rho, self.kmin, self.seq, self.done = rr, 0.0, 0, 0

    def ap(self):
        if self.done:
            return
        if self.seq > 0:
            rr = self.buf[sharded.RR].sum()
            self.eps = np.sqrt(rr) / self.normb
            q = 2.0 * self.seq / np.log(2.0 * self.eps0 / self.eps)
            self.kmin = max(self.kmin, q * q)
            if self.eps < self.tol:
                self.done = 1
            elif self.kmin > self.kmax:
                self.done = 2
            elif self.seq >= self.maxiter:
                self.done = 3
            if self.done:
                return
            beta = rr / self.rho
            self.rho = rr
            self.p = self.r + beta * self.p
        self.z = self.orc.mulMTM(self.m, np.ascontiguousarray(self.p))
        self.buf[sharded.PAP] = self._dot(self.p, self.z)
        self.seq += 1

    def xr(self):
        if self.done:
            return
        alpha = self.rho / self.buf[sharded.PAP].sum()
        self.x += alpha * self.p
        self.r -= alpha * self.z
        self.buf[sharded.RR] = self._dot(self.r, self.r)

    def status(self):
        return (self.seq if self.done else self.seq), self.done, float(self.eps)

    def _vec(self, which):          # device layout (tau, site) flat view of r / x
        return self.r if which == sharded.RVEC else self.x

    def read(self, which, offset, count):
        if which in self.buf:
            return self.buf[which][offset:offset + count].copy()
        S = self._slices(self._vec(which)).T.reshape(-1)            # (tau, site) order
        return S[offset:offset + count].copy()

    def write(self, which, offset, values):
        values = np.asarray(values, dtype=np.float64)
        if which in self.buf:
            self.buf[which][offset:offset + values.size] = values
            return
        v = self._vec(which)
        S = self._slices(v).T.copy().reshape(-1)
        S[offset:offset + values.size] = values
        v[:] = S.reshape(self.L, self.N).T.reshape(-1)

    def close(self):
        pass


def main_spatial(mode, out, comm):
    """Two lattices through SpatialShardedCG: square 8x8 (ghost rows 2+2) and honeycomb 4x4x2 (ghost rows 1+1)."""
    res = {}
    for tag, norb, Ls, bonds, Ltau in (("sq", 1, 8, lat.SQUARE_BONDS, 8), ("hc", 2, 4, lat.HONEYCOMB_BONDS, 6)):
        dtau = 0.1
        la = lat.Lattice(norb, Ls, Ls, 1)
        raw = np.concatenate([la.calc_neighbor_table(o1, o2, d) for (o1, o2, d) in bonds], axis=0)
        tvals = 1.0 + 0.1 * synth.randn(5, raw.shape[0])             # disordered hoppings: every bond is distinguishable
        cb = lat.initialize_checkerboard(raw, tvals, dtau)
        N = la.nsites
        x = synth.phonon_field(N, Ltau, Ltau * dtau, dtau, seed=123)
        E = np.exp(-dtau * (1.0 * x - 0.0))
        b = synth.randn(321, N * Ltau)
        solver = sharded.SpatialShardedCG(comm, norb, Ls, Ls, Ltau, cb["table"], cb["cosht"], cb["sinht"],
                                          backend_factory=lambda n, l, t, c, s_: NumpyBackend(n, l, t, c, s_))
        solver.update_model(E)
        xs, it, done = solver.solve(b, tol=1e-9, maxiter=2000, check_every=4)
        res.update({f"{tag}_x": xs, f"{tag}_it": it, f"{tag}_done": done, f"{tag}_E": E, f"{tag}_b": b, f"{tag}_table": cb["table"],
                    f"{tag}_c": cb["cosht"], f"{tag}_s": cb["sinht"], f"{tag}_N": N, f"{tag}_Ltau": Ltau,
                    f"{tag}_halo": np.array([solver.sl["lo"], solver.sl["hi"]])})
        solver.close()
    np.savez(out + f".rank{comm.rank}", **res)
    comm.close()


class NumpyLocal:
    """The slab's operator for elphdynamics_amd.sharded_rccl.CollectiveShardedSolver by the CPU oracle (CPU tensors): test infrastructure — the
    product's local operator is LibraryLocal (the HIP library)."""

    def __init__(self, torch, Nloc, ltau, ltab, cosht, sinht, device_index):
        from oracle.oracle import Oracle
        self.orc, self.torch = Oracle(), torch
        self.N, self.L, self.table, self.c, self.s = int(Nloc), int(ltau), ltab, np.ascontiguousarray(cosht), np.ascontiguousarray(sinht)
        self.device = torch.device("cpu")

    def set_expV(self, E_loc):
        self.m = self.orc.make_model(0, self.N, self.L, self.table, self.c, self.s, np.ascontiguousarray(E_loc).reshape(-1))

    def mtm(self, z, p):
        z.copy_(self.torch.from_numpy(self.orc.mulMTM(self.m, np.ascontiguousarray(p.numpy()))))

    def close(self):
        pass


def main_collectives(out, comm):
    """The COLLECTIVE transport of the sharded solve (sharded_rccl.CollectiveShardedSolver: all-reduces of p.z and r.r, one grouped ghost-row
    exchange of p per iteration) on the two lattices of main_spatial, next to the host-spelled mailbox protocol on the same system."""
    from elphdynamics_amd import sharded_rccl
    res = {}
    for tag, norb, Ls, bonds, Ltau in (("sq", 1, 8, lat.SQUARE_BONDS, 8), ("hc", 2, 4, lat.HONEYCOMB_BONDS, 6)):
        dtau = 0.1
        la = lat.Lattice(norb, Ls, Ls, 1)
        raw = np.concatenate([la.calc_neighbor_table(o1, o2, d) for (o1, o2, d) in bonds], axis=0)
        tvals = 1.0 + 0.1 * synth.randn(5, raw.shape[0])
        cb = lat.initialize_checkerboard(raw, tvals, dtau)
        N = la.nsites
        x = synth.phonon_field(N, Ltau, Ltau * dtau, dtau, seed=123)
        E = np.exp(-dtau * (1.0 * x - 0.0))
        b = synth.randn(321, N * Ltau)
        solver = sharded_rccl.CollectiveShardedSolver(comm, norb, Ls, Ls, Ltau, cb["table"], cb["cosht"], cb["sinht"], local_factory=NumpyLocal)
        solver.update_model(E)
        xs, it, done = solver.solve(b, tol=1e-9, maxiter=2000, check_every=4)
        ncoll = solver.collectives
        x5, it5, done5 = solver.solve(b, fixed_iters=5)                   # (the measurement form: exactly five iterations)
        ref = sharded.SpatialShardedCG(comm, norb, Ls, Ls, Ltau, cb["table"], cb["cosht"], cb["sinht"],
                                       backend_factory=lambda n, l, t, c, s_: NumpyBackend(n, l, t, c, s_))
        ref.update_model(E)
        xr, itr, doner = ref.solve(b, tol=1e-9, maxiter=2000, check_every=4)
        res.update({f"{tag}_x": xs, f"{tag}_it": it, f"{tag}_done": done, f"{tag}_E": E, f"{tag}_b": b, f"{tag}_table": cb["table"],
                    f"{tag}_c": cb["cosht"], f"{tag}_s": cb["sinht"], f"{tag}_N": N, f"{tag}_Ltau": Ltau,
                    f"{tag}_halo": np.array([solver.sl["lo"], solver.sl["hi"]]), f"{tag}_ncoll": ncoll, f"{tag}_it5": it5, f"{tag}_x5": x5,
                    f"{tag}_xref": xr, f"{tag}_itref": itr})
        solver.close()
        ref.close()
    np.savez(out + f".rank{comm.rank}", **res)
    comm.close()


def main():
    mode, out = sys.argv[1], sys.argv[2]
    comm = dist.Comm(backend="gloo")
    if mode == "collectives":
        return main_collectives(out, comm)
    if mode.endswith("-spatial"):
        return main_spatial(mode.split("-")[0], out, comm)
    # Holstein square lattice, small enough for the CPU backend: L = 4, Ltau = 16
    Ls, Ltau, dtau = 4, 16, 0.1
    la = lat.Lattice(1, Ls, Ls, 1)
    raw = np.concatenate([la.calc_neighbor_table(o1, o2, d) for (o1, o2, d) in lat.SQUARE_BONDS], axis=0)
    cb = lat.initialize_checkerboard(raw, np.ones(raw.shape[0]), dtau)
    N = la.nsites
    x = synth.phonon_field(N, Ltau, Ltau * dtau, dtau, seed=123)
    E = np.exp(-dtau * (1.0 * x - 0.0))
    b = synth.randn(321, N * Ltau)
    assert mode == "numpy", mode
    factory = lambda: NumpyBackend(N, Ltau // comm.world + 2, cb["table"], cb["cosht"], cb["sinht"])  # noqa: E731
    solver = sharded.ShardedCG(comm, N, Ltau, cb["table"], cb["cosht"], cb["sinht"], backend_factory=factory)
    solver.update_model(E)
    xs, it, done = solver.solve(b, tol=1e-9, maxiter=2000, check_every=4)
    np.savez(out + f".rank{comm.rank}", x=xs, it=it, done=done, E=E, b=b, table=cb["table"], c=cb["cosht"], s=cb["sinht"],
             N=N, Ltau=Ltau)
    solver.close()
    comm.close()


if __name__ == "__main__":
    main()
