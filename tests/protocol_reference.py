"""TEST HELPER, not product code: the host-driven protocol of one CG solve over several ranks, with a pluggable local back end.

The product's sharded solve lives inside the library (elphdynamics_amd/csrc/shard.hip, driven by elphdynamics_amd.sharded.ShardedSolver):
the ranks' resident kernels exchange partial sums and boundary rows themselves.  What is kept here is the round-1 harness that spells the
same exchange out on the host — per iteration one ghost exchange of r and two scalar combinations — over `dist.Comm` (gloo in the CPU
tests) and a numpy stand-in for the local kernels (tests/sharded_worker.py: NumpyBackend).  It is the CPU-testable statement of what a
rank must send, receive and sum, and it exercises the product's slab arithmetic (`sharded.SpatialSlabs`, `mtm_dependency_closure`) at
world size 2 without a GPU.  Nothing under elphdynamics_amd/ imports it.

  ShardedCG         slabs along the imaginary-time axis (one halo slice on each side)
  SpatialShardedCG  slabs of rows of cells + ghost rows (the decomposition of SURVEY §8e / the north_star)
"""
import numpy as np

from elphdynamics_amd.sharded import SpatialSlabs

PAP, RR, BB, RVEC, XVEC = 0, 1, 2, 3, 4


class ShardedCG:
    """Un-preconditioned CG on MtM x = b for ONE Holstein fermion matrix, tau-slabs over comm.world ranks.  The local handle has
    Lloc + 2 slices; exp(-dtau V) of local slice 0 is zero (no wrap inside the local operator) and the slice that is global tau = 0
    carries -exp(-dtau V): the anti-periodic corner M[1, Ltau] = +B(1) (HolsteinModels.jl:575-581)."""

    def __init__(self, comm, nsites, ltau, table, cosht, sinht, backend_factory):
        self.comm = comm
        self.P, self.rank = comm.world, comm.rank
        assert ltau % self.P == 0, "Ltau must be divisible by the number of ranks"
        self.N, self.Ltau = int(nsites), int(ltau)
        self.Lloc = self.Ltau // self.P
        self.t0 = self.rank * self.Lloc
        self.be = backend_factory()

    def _taus(self):
        return np.arange(self.t0 - 1, self.t0 + self.Lloc + 1) % self.Ltau

    def update_model(self, expV_global):
        Eg = np.asarray(expV_global).reshape(self.N, self.Ltau)
        taus = self._taus()
        E = Eg[:, taus].copy()
        E[:, taus == 0] *= -1.0
        E[:, 0] = 0.0
        self.be.set_expV(E)

    def _combine(self, which):
        own = self.be.read(which, 1, self.Lloc)
        total = float(np.sum(self.comm.allgather(own)))      # same array, same order on every rank => same bits
        buf = np.zeros(self.Lloc + 2)
        buf[0] = total
        self.be.write(which, 0, buf)

    def _exchange_r_halo(self):
        N, Ll = self.N, self.Lloc
        first = self.be.read(RVEC, 1 * N, N)                  # own first slice -> previous rank's upper halo
        last = self.be.read(RVEC, Ll * N, N)                  # own last slice  -> next rank's lower halo
        from_prev, from_next = self.comm.ring_exchange(send_to_prev=first, send_to_next=last)
        self.be.write(RVEC, 0, from_prev)
        self.be.write(RVEC, (Ll + 1) * N, from_next)

    def solve(self, b_global, tol=1e-5, maxiter=10000, kmax=1e12, check_every=8):
        """Returns (x_global (N*Ltau,), iterations, done_flag) — identical on every rank."""
        bg = np.asarray(b_global).reshape(self.N, self.Ltau)
        self.be.begin(bg[:, self._taus()], tol, maxiter, kmax)
        self._combine(RR)
        self._combine(BB)
        self.be.state0()
        it, done, launched = 0, 0, 0
        while not done and launched <= maxiter + 1:
            for _ in range(check_every):
                self.be.ap()
                self._combine(PAP)
                self.be.xr()
                self._combine(RR)
                self._exchange_r_halo()
                launched += 1
            it, done, _ = self.be.status()
        x_own = self.be.read(XVEC, 1 * self.N, self.Lloc * self.N).reshape(self.Lloc, self.N)   # device layout (tau, site)
        x_all = self.comm.allgather(x_own.reshape(-1)).reshape(self.P * self.Lloc, self.N)
        return np.ascontiguousarray(x_all.T).reshape(-1), it, done

    def close(self):
        self.be.close()


class SpatialShardedCG:
    """The same solve over slabs of rows of cells.  Rank q's local lattice = own rows + the ghost rows found by
    `mtm_dependency_closure`; the fused mat-vec on it yields the exact z = MᵀM p on the own rows with no exchange inside; per
    iteration the ranks exchange the ghost rows of r once and combine the inner products from partial sums over own sites."""

    def __init__(self, comm, norbits, L1, L2, ltau, table, cosht, sinht, backend_factory):
        self.comm, self.P, self.rank = comm, comm.world, comm.rank
        self.Ltau = int(ltau)
        self.slabs = SpatialSlabs(norbits, L1, L2, table, self.P)
        self.N = self.slabs.N
        sl = self.sl = self.slabs.slabs[self.rank]
        self.row = self.slabs.row
        self.Nloc = sl["rows"].size * self.row
        self.own_lo, self.own_n = sl["lo"] * self.row, sl["R"] * self.row
        ltab = self.slabs.local_table(self.rank, table)
        c, s = np.asarray(cosht)[sl["bonds"]], np.asarray(sinht)[sl["bonds"]]
        self.be = backend_factory(self.Nloc, self.Ltau, ltab, c, s)
        self.gsites = self.slabs.global_sites(self.rank)
        if self.P > 1:
            self.be.set_dot_range(self.own_lo, self.own_lo + self.own_n)
            prev, nxt = (self.rank - 1) % self.P, (self.rank + 1) % self.P
            sp, sn = self.slabs.slabs[prev], self.slabs.slabs[nxt]
            if sl["lo"] > sp["R"] or sl["hi"] > sn["R"]:
                raise ValueError("ghost rows reach beyond the neighbouring rank: use fewer ranks")
            self.n_to_next, self.n_to_prev = sn["lo"] * self.row, sp["hi"] * self.row
            self.n_from_prev, self.n_from_next = sl["lo"] * self.row, sl["hi"] * self.row

    def update_model(self, expV_global):
        Eg = np.asarray(expV_global).reshape(self.N, self.Ltau)
        self.be.set_expV(Eg[self.gsites, :])

    def _combine(self, which):
        if self.P == 1:
            return
        own = self.be.read(which, 0, self.Ltau)
        total = float(np.sum(self.comm.allgather(own)))
        buf = np.zeros(self.Ltau)
        buf[0] = total
        self.be.write(which, 0, buf)

    def _exchange_r_halo(self):
        if self.P == 1:
            return
        lo, n = self.own_lo, self.own_n
        to_prev = self.be.read_rows(RVEC, lo, self.n_to_prev) if self.n_to_prev else np.zeros((self.Ltau, 0))
        to_next = self.be.read_rows(RVEC, lo + n - self.n_to_next, self.n_to_next) if self.n_to_next else np.zeros((self.Ltau, 0))
        from_prev, from_next = self.comm.ring_exchange(send_to_prev=to_prev.reshape(-1), send_to_next=to_next.reshape(-1),
                                                       recv_prev_n=self.Ltau * self.n_from_prev,
                                                       recv_next_n=self.Ltau * self.n_from_next)
        if self.n_from_prev:
            self.be.write_rows(RVEC, 0, np.asarray(from_prev).reshape(self.Ltau, self.n_from_prev))
        if self.n_from_next:
            self.be.write_rows(RVEC, lo + n, np.asarray(from_next).reshape(self.Ltau, self.n_from_next))

    def solve(self, b_global, tol=1e-5, maxiter=10000, kmax=1e12, check_every=8):
        """Returns (x_global (N*Ltau,), iterations, done_flag) — identical on every rank."""
        self.be.begin(np.asarray(b_global).reshape(self.N, self.Ltau)[self.gsites, :], tol, maxiter, kmax)
        self._combine(RR)
        self._combine(BB)
        self.be.state0()
        it, done, launched = 0, 0, 0
        while not done and launched <= maxiter + 1:
            for _ in range(check_every):
                self.be.ap()
                self._combine(PAP)
                self.be.xr()
                self._combine(RR)
                self._exchange_r_halo()
                launched += 1
            it, done, _ = self.be.status()
        x_own = self.be.read_rows(XVEC, self.own_lo, self.own_n)                       # (Ltau, own sites)
        parts = self.comm.allgather_object(x_own) if self.P > 1 else [x_own]
        x = np.concatenate(parts, axis=1)                                              # (Ltau, N): ranks own ascending rows
        return np.ascontiguousarray(x.T).reshape(-1), it, done

    def close(self):
        self.be.close()
