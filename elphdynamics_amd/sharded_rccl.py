"""The sharded solve with COLLECTIVES as its transport (north_star: "RCCL halo exchange over xGMI of the checkerboard
boundary slice once per mat-vec" + all-reduces of the two inner products) — `CollectiveShardedSolver`.

Why it exists next to `sharded.ShardedSolver` (device-initiated stores into peer-mapped mailboxes from inside the resident kernel):
  * it is the north_star's literal design, and the only way the mailbox form's claim ("faster than collectives at these sizes") can be A/B-ed;
  * it is the fall-back when peer-mapped mailboxes are not to be had between two devices (`ShardedSolver`'s preflight fails): `make_solver`
    below picks it then, so the first contact with a multi-GPU node yields a number either way.

Same decomposition as the mailbox form (`sharded.SpatialSlabs`: slabs of rows of cells + the ghost rows the fused z = Mᵀ(M p) reads, so the
UNMODIFIED mat-vec of the library is exact on the own rows), the reference's recurrences (IterativeSolvers.jl:239-314) on the own rows:

    per iteration:   z = MᵀM p  (library, on the slab)                     ── no exchange inside
                     (p·z)      all-reduce #1  -> α
                     x += α p,  r -= α z  (own rows)
                     (r·r)      all-reduce #2  -> ε, stop test, β
                     p = r + β p (own rows);  ghost rows of p  <- neighbours   ── ONE grouped send/recv pair per direction

Everything stays on the device between the collectives: vectors are torch tensors in the library's reference layout (site-major: the rows of a
slab are contiguous, a halo is one contiguous slice), the library works on their storage through its device-pointer entry points
(`elph_mulMTM_dev`, on torch's stream), the scalars α, β, ε, κ are 0-dim device tensors, and the host looks at the `done` flag every
`check_every` iterations only.  Every rank evaluates the stop rule on the same all-reduced numbers, hence takes the same decision at the same
iteration.  torch.distributed backend "nccl" is RCCL on ROCm (one process per GPU); "gloo" serves the CPU tests (world 2, numpy local operator)
and rehearsals with several ranks on one GPU (vectors on the device, messages staged through the host).

Nothing here is on the single-GPU product path; `bench.py`'s `spatial` record reports `transport` and `rccl_ranks`.
"""
import ctypes as C
import math
import os

import numpy as np

from .sharded import SpatialSlabs


class LibraryLocal:
    """The slab's operator through libelphgpu.so on device pointers (Holstein; reference layout)."""

    def __init__(self, torch, Nloc, ltau, ltab, cosht, sinht, device_index):
        from . import _lib
        self._lm, self.lib, self.torch = _lib, _lib.load(), torch
        self.h = _lib.Handle()
        nb = ltab.shape[0]
        c, s = np.ascontiguousarray(cosht), np.ascontiguousarray(sinht)
        _lib.check(self.lib.elph_create(C.byref(self.h), 0, int(Nloc), int(ltau), nb, _lib.iptr(np.ascontiguousarray(ltab, dtype=np.int64)) if nb else None,
                                        _lib.dptr(c) if nb else None, _lib.dptr(s) if nb else None, int(device_index)))
        self.device = torch.device("cuda", int(device_index))
        # ONE stream for the torch ops of the iteration and the library's launches: no synchronisation between a torch op and the mat-vec that
        # follows it.  A stream of our own, not torch's default one: the default stream's handle is NULL, which elph_set_stream reads as "keep
        # your own (non-blocking) stream" — the mat-vec would then race with the torch ops around it.
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.Stream(device=self.device)
        assert self.stream.cuda_stream != 0
        _lib.check(self.lib.elph_set_stream(self.h, C.c_void_p(self.stream.cuda_stream)))

    def stream_ctx(self):
        """Context in which the solver issues its torch ops (and its collectives): the stream the library launches on."""
        return self.torch.cuda.stream(self.stream)

    def set_expV(self, E_loc):
        self._lm.check(self.lib.elph_set_expV(self.h, self._lm.dptr(np.ascontiguousarray(E_loc, dtype=np.float64).reshape(-1))))

    def mtm(self, z, p):
        """z = Mᵀ(M p) on the slab (exact on the own rows)."""
        self._lm.check(self.lib.elph_mulMTM_dev(self.h, C.c_void_p(z.data_ptr()), C.c_void_p(p.data_ptr())))

    def close(self):
        if self.h:
            self.lib.elph_destroy(self.h)
            self.h = None


class CollectiveShardedSolver:
    """ONE un-preconditioned solve of MᵀM x = b (x0 = 0) over comm.world ranks with torch.distributed collectives as the transport."""

    transport = "collectives"

    def __init__(self, comm, norbits, L1, L2, ltau, table, cosht, sinht, device=None, local_factory=None):
        """local_factory(torch, Nloc, ltau, local_table, cosht_local, sinht_local, device_index) -> object with set_expV(E_loc), mtm(z, p),
        close() and a `.device` (torch.device of the vectors).  Default: `LibraryLocal` (the HIP library on this rank's GPU)."""
        import torch
        self.torch, self.comm = torch, comm
        self.P, self.rank, self.Ltau = comm.world, comm.rank, int(ltau)
        self.slabs = SpatialSlabs(norbits, L1, L2, table, self.P)
        self.N, self.row = self.slabs.N, self.slabs.row
        sl = self.sl = self.slabs.slabs[self.rank]
        self.Nloc = sl["rows"].size * self.row
        self.own_lo, self.own_n = sl["lo"] * self.row, sl["R"] * self.row
        self.gsites = self.slabs.global_sites(self.rank)
        ltab = np.ascontiguousarray(self.slabs.local_table(self.rank, table), dtype=np.int64)
        bonds = sl["bonds"]
        self.n_to_prev = self.n_to_next = self.n_from_prev = self.n_from_next = 0
        if self.P > 1:
            prev, nxt = (self.rank - 1) % self.P, (self.rank + 1) % self.P
            sp, sn = self.slabs.slabs[prev], self.slabs.slabs[nxt]
            if sl["lo"] > sp["R"] or sl["hi"] > sn["R"]:
                raise ValueError("ghost rows reach beyond the neighbouring rank: use fewer ranks")
            self.n_to_next, self.n_to_prev = sn["lo"] * self.row, sp["hi"] * self.row
            self.n_from_prev, self.n_from_next = sl["lo"] * self.row, sl["hi"] * self.row
        dev = comm.device_index() if device is None else int(device)
        factory = local_factory or LibraryLocal
        self.local = factory(torch, self.Nloc, self.Ltau, ltab, np.asarray(cosht)[bonds], np.asarray(sinht)[bonds], dev)
        self.device = self.local.device
        # are the messages device tensors (RCCL) or staged through the host (gloo)?
        self.direct = (getattr(comm, "backend", None) == "nccl")
        if self.P > 1 and getattr(comm, "dist", None) is None:
            raise ValueError("CollectiveShardedSolver needs a torch.distributed communicator (dist.Comm)")
        self.collectives = 0                  # collective calls of the last solve (all-reduces + grouped exchanges)
        comm.barrier()

    # ---- slab <-> lattice ------------------------------------------------------------------------------------------------------
    def _local(self, v_global):
        return np.ascontiguousarray(np.asarray(v_global, dtype=np.float64).reshape(self.N, self.Ltau)[self.gsites, :]).reshape(-1)

    def update_model(self, expV_global):
        """Holstein: exp(-Δτ V) of the whole lattice, reference layout (update_model!, HolsteinModels.jl:526-549)."""
        self.local.set_expV(self._local(expV_global))

    # ---- the two collectives ---------------------------------------------------------------------------------------------------
    def _allreduce(self, t):
        """In-place SUM over the ranks of a small device tensor; the same bits on every rank."""
        if self.P == 1 and not (self.direct and getattr(self.comm, "dist", None) is not None):
            return t                          # (a one-rank group that WAS initialised — ELPH_DIST_FORCE_INIT — still goes through RCCL: the world-1 test)
        self.collectives += 1
        if self.direct:
            self.comm.dist.all_reduce(t, op=self.comm.dist.ReduceOp.SUM)
            return t
        c = t.detach().to("cpu")
        self.comm.dist.all_reduce(c, op=self.comm.dist.ReduceOp.SUM)
        t.copy_(c)
        return t

    def _exchange_ghosts(self, v):
        """Ghost rows of the slab vector v <- the neighbours' own rows: one grouped send/recv pair per direction (the `halo exchange once per
        mat-vec pair` of SURVEY §8e; reference layout: rows of sites are contiguous, so every message is one contiguous slice)."""
        if self.P == 1:
            return
        L, lo, n = self.Ltau, self.own_lo, self.own_n
        d = self.comm.dist
        prev, nxt = (self.rank - 1) % self.P, (self.rank + 1) % self.P
        to_prev = v[lo * L:(lo + self.n_to_prev) * L]
        to_next = v[(lo + n - self.n_to_next) * L:(lo + n) * L]
        from_prev = v[0:self.n_from_prev * L]
        from_next = v[(lo + n) * L:(lo + n + self.n_from_next) * L]
        if not self.direct:
            sp, sn = to_prev.to("cpu"), to_next.to("cpu")
            rp = self.torch.empty(self.n_from_prev * L, dtype=v.dtype)
            rn = self.torch.empty(self.n_from_next * L, dtype=v.dtype)
        else:
            sp, sn, rp, rn = to_prev, to_next, from_prev, from_next
        ops = []
        # (two ranks: previous and next are the same peer — the pairing of its two messages is by order: what I send "to prev" is what it
        #  receives "from next", so each side posts send-to-prev with recv-from-next first, then send-to-next with recv-from-prev)
        if sp.numel():
            ops.append(d.P2POp(d.isend, sp, prev))
        if rn.numel():
            ops.append(d.P2POp(d.irecv, rn, nxt))
        if sn.numel():
            ops.append(d.P2POp(d.isend, sn, nxt))
        if rp.numel():
            ops.append(d.P2POp(d.irecv, rp, prev))
        if ops:
            self.collectives += 1
            for req in d.batch_isend_irecv(ops):
                req.wait()
        if not self.direct:
            if rp.numel():
                from_prev.copy_(rp)
            if rn.numel():
                from_next.copy_(rn)

    # ---- the solve --------------------------------------------------------------------------------------------------------------
    def _own(self, v):
        return v[self.own_lo * self.Ltau:(self.own_lo + self.own_n) * self.Ltau]

    def solve(self, b_global, tol=1e-5, maxiter=10000, kmax=1e12, check_every=8, fixed_iters=0):
        """Returns (x_global (N·Ltau,), iterations, done) — identical on every rank.  done: 1 ε < tol, 2 κ > κmax, 3 maxiter (the library's
        codes).  fixed_iters > 0: exactly that many iterations, no stop test (measurement)."""
        import contextlib
        ctx = self.local.stream_ctx() if hasattr(self.local, "stream_ctx") else contextlib.nullcontext()
        with ctx:
            return self._solve(b_global, tol, maxiter, kmax, check_every, fixed_iters)

    def _solve(self, b_global, tol, maxiter, kmax, check_every, fixed_iters):
        torch = self.torch
        f64 = torch.float64
        dev = self.device
        b = torch.from_numpy(self._local(b_global)).to(dev)
        x = torch.zeros_like(b)
        r = b.clone()
        p = b.clone()
        z = torch.empty_like(b)
        self.collectives = 0
        xo, ro, po, zo = self._own(x), self._own(r), self._own(p), self._own(z)
        two = torch.zeros(2, dtype=f64, device=dev)
        two[0] = torch.dot(ro, ro)
        two[1] = two[0]                                   # x0 = 0: r0 = b
        self._allreduce(two)
        normb = torch.sqrt(two[1])
        rho = two[0].clone()
        eps0 = torch.sqrt(rho) / normb
        kmin = torch.zeros((), dtype=f64, device=dev)
        done = torch.zeros((), dtype=torch.int32, device=dev)
        iters = torch.zeros((), dtype=torch.int64, device=dev)
        one = torch.ones((), dtype=f64, device=dev)
        zero = torch.zeros((), dtype=f64, device=dev)
        tol_t, kmax_t = torch.tensor(float(tol), dtype=f64, device=dev), torch.tensor(float(kmax), dtype=f64, device=dev)
        s1 = torch.zeros(1, dtype=f64, device=dev)
        eps_t = eps0.clone()
        j, limit = 0, (int(fixed_iters) if fixed_iters > 0 else int(maxiter))
        finished = False
        while j < limit and not finished:
            for _ in range(min(check_every, limit - j)):
                j += 1
                live = (done == 0)
                self.local.mtm(z, p)                                       # z = A p (exact on the own rows)
                s1[0] = torch.dot(po, zo)
                self._allreduce(s1)                                        # p·z
                alpha = torch.where(live, rho / s1[0], zero)               # a finished solve takes no step
                xo.add_(po * alpha)
                ro.sub_(zo * alpha)
                s1[0] = torch.dot(ro, ro)
                self._allreduce(s1)                                        # r·r
                rr = s1[0]
                eps = torch.sqrt(rr) / normb
                eps_t = torch.where(live, eps, eps_t)                      # ε of the last iteration that was taken
                if fixed_iters <= 0:
                    q = (2.0 * j) / torch.log(2.0 * eps0 / eps)
                    kmin = torch.where(live, torch.maximum(kmin, q * q), kmin)
                    hit = torch.where(eps < tol_t, 1, torch.where(kmin > kmax_t, 2, 0)).to(torch.int32)
                    newly = live & (hit != 0)
                    iters = torch.where(live, torch.full_like(iters, j), iters)
                    done = torch.where(newly, hit, done)
                    live = (done == 0)
                else:
                    iters = torch.full_like(iters, j)
                beta = torch.where(live, rr / rho, zero)
                rho = torch.where(live, rr, rho)
                po.copy_(torch.where(live, ro + beta * po, po))
                self._exchange_ghosts(p)                                   # the ghost rows of the new search direction
            if fixed_iters <= 0:
                finished = int(done.item()) != 0                           # the only host read of the loop (every check_every iterations)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        it, dn = int(iters.item()), int(done.item())
        if fixed_iters <= 0 and dn == 0:
            dn = 3
        self.eps = float(eps_t.item())
        x_own = xo.detach().to("cpu").numpy().reshape(self.own_n, self.Ltau)
        parts = self.comm.allgather_object(x_own) if self.P > 1 else [x_own]
        return np.ascontiguousarray(np.concatenate(parts, axis=0)).reshape(-1), it, dn

    def iterate(self, b_global, k):
        """Exactly k iterations (no stop test); returns the wall time in ms on this rank (bench.py's `spatial` record)."""
        import time
        if self.device.type == "cuda":
            self.torch.cuda.synchronize(self.device)
        self.comm.barrier()
        t0 = time.perf_counter()
        self.solve(b_global, fixed_iters=int(k))
        return 1e3 * (time.perf_counter() - t0)

    def close(self):
        if self.local is not None:
            self.local.close()
            self.local = None


def make_solver(comm, norbits, L1, L2, ltau, table, cosht, sinht, kind=0, device=None, transport=None):
    """The sharded solver of this run: ELPH_SHARD_TRANSPORT = mailbox | rccl | auto (default).  auto: the in-library mailbox form
    (`sharded.ShardedSolver`) when its preflight passes on EVERY rank, else — peer mapping refused, a silent rank — the collective transport
    (Holstein only).  Returns (solver, transport name, reason)."""
    from .sharded import ShardedSolver
    want = (transport or os.environ.get("ELPH_SHARD_TRANSPORT") or "auto").lower()
    if want not in ("mailbox", "rccl", "collectives", "auto"):
        raise ValueError(f"ELPH_SHARD_TRANSPORT={want}: mailbox, rccl or auto")
    if want in ("rccl", "collectives"):
        return CollectiveShardedSolver(comm, norbits, L1, L2, ltau, table, cosht, sinht, device=device), "collectives", "requested"
    err = None
    solver = None
    try:
        solver = ShardedSolver(comm, norbits, L1, L2, ltau, table, kind=kind, cosht=cosht, sinht=sinht, device=device, selftest=True)
    except Exception as e:      # noqa: BLE001 — the preflight names the failing ranks; every rank must take the same branch below
        err = f"{type(e).__name__}: {e}"
    bad = [e for e in (comm.allgather_object(err) if comm.world > 1 else [err]) if e]
    if not bad:
        return solver, "mailbox", "preflight passed"
    if solver is not None:
        solver.close()
    if want == "mailbox" or kind != 0:
        raise RuntimeError("sharded solve: the mailbox transport failed its preflight (" + "; ".join(bad) + ")" + (" and the collective transport serves site phonons only" if kind != 0 else ""))
    return CollectiveShardedSolver(comm, norbits, L1, L2, ltau, table, cosht, sinht, device=device), "collectives", "mailbox preflight failed: " + "; ".join(bad)[:300]
