"""ONE conjugate-gradient solve sharded over several GPUs (north_star "lattice shards over the GPUs of
a node with RCCL halo exchange once per mat-vec").

Decomposition: slabs along the IMAGINARY-TIME axis.  In the device layout a tau-slice is N contiguous doubles, the
two bidiagonal factors of MtM couple only tau and tau±1, and the checkerboard sweep is entirely inside a slice — so a
rank that owns tau in [t0, t0+Lloc) plus ONE halo slice on each side applies the fused MtM kernel unchanged:

  * the local handle has Lloc+2 slices; expnDtauV of local slice 0 is set to ZERO, which removes the periodic wrap
    of the local operator, and the slice that is global tau = 0 carries -expnDtauV, which reproduces the
    anti-periodic "+B(1)" corner (HolsteinModels.jl:575-581) with the kernel's regular "-" sign;
  * per CG iteration there is ONE halo exchange (the two boundary slices of r, 2 x N doubles, before the mat-vec)
    and two scalar combinations (p.Ap and r.r): every rank all-gathers the per-slice partial sums of the OWNED slices
    and overwrites its partial-sum buffer with the global total, so every rank's kernels take bit-identical
    alpha, beta and stop decisions (IterativeSolvers.jl:277-310) — no host-side CG arithmetic.

The spatial-axis variant of SURVEY §8e needs an exchange in the MIDDLE of the colour sweep (the last colour crosses
the slab boundary) and therefore a split mat-vec; the tau-axis variant needs neither, works for every lattice
geometry and any checkerboard colouring, and moves 10x fewer bytes per exchange (2 KB instead of 20 KB at config C).
Either way a sharded solve at these sizes is latency-bound by the collectives: use it when one
fermion matrix no longer fits the time budget of one GPU, not for throughput — independent chains scale perfectly.

Communication goes through `dist.Comm` (torch.distributed: "nccl" = RCCL over xGMI on GPU boxes, "gloo" in the CPU
tests); the local compute goes through the step-wise C ABI (`elph_cgstep_*`, include/elph_gpu.h) or, in the CPU
tests only, through a numpy stand-in backend with the same interface (tests/test_sharded_gloo.py).
"""
import ctypes as C

import numpy as np

PAP, RR, BB, RVEC, XVEC = 0, 1, 2, 3, 4


class GpuBackend:
    """Local compute on one GPU through libelphgpu's step-wise CG entry points."""

    def __init__(self, nsites, lloc2, table, cosht, sinht, device=0):
        from . import _lib
        self._lib_mod = _lib
        self.lib = _lib.load()
        self.h = _lib.Handle()
        tab = np.ascontiguousarray(table, dtype=np.int64)
        nb = tab.shape[0]
        _lib.check(self.lib.elph_create(C.byref(self.h), 0, nsites, lloc2, nb, _lib.iptr(tab) if nb else None,
                                        _lib.dptr(np.ascontiguousarray(cosht)) if nb else None,
                                        _lib.dptr(np.ascontiguousarray(sinht)) if nb else None, device))
        self.N, self.L = nsites, lloc2

    def set_expV(self, E_loc):                      # E_loc: (N, Lloc+2) reference layout
        self._lib_mod.check(self.lib.elph_set_expV(self.h, self._lib_mod.dptr(np.ascontiguousarray(E_loc).reshape(-1))))

    def begin(self, b_loc, tol, maxiter, kmax):
        self._lib_mod.check(self.lib.elph_cgstep_begin(self.h, self._lib_mod.dptr(np.ascontiguousarray(b_loc).reshape(-1)),
                                                       tol, maxiter, kmax))

    def state0(self):
        self._lib_mod.check(self.lib.elph_cgstep_state0(self.h))

    def ap(self):
        self._lib_mod.check(self.lib.elph_cgstep_ap(self.h))

    def xr(self):
        self._lib_mod.check(self.lib.elph_cgstep_xr(self.h))

    def status(self):
        it, done, eps = C.c_int64(), C.c_int(), C.c_double()
        self._lib_mod.check(self.lib.elph_cgstep_status(self.h, C.byref(it), C.byref(done), C.byref(eps)))
        return int(it.value), int(done.value), float(eps.value)

    def read(self, which, offset, count):
        out = np.empty(count)
        self._lib_mod.check(self.lib.elph_buffer_read(self.h, which, offset, count, self._lib_mod.dptr(out)))
        return out

    def write(self, which, offset, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._lib_mod.check(self.lib.elph_buffer_write(self.h, which, offset, v.size, self._lib_mod.dptr(v)))

    # ---- spatial shards: inner products over the own sites only; rows (site ranges over all tau) of r / x
    def set_dot_range(self, lo, hi):
        self._lib_mod.check(self.lib.elph_set_dot_range(self.h, lo, hi))

    def read_rows(self, which, site_lo, nsites):
        out = np.empty((self.L, nsites))
        self._lib_mod.check(self.lib.elph_buffer_read_rows(self.h, which, site_lo, nsites, self._lib_mod.dptr(out)))
        return out

    def write_rows(self, which, site_lo, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        assert v.shape[0] == self.L
        self._lib_mod.check(self.lib.elph_buffer_write_rows(self.h, which, site_lo, v.shape[1], self._lib_mod.dptr(v)))

    # ---- zero-copy torch views of the device buffers (nccl path: collectives act on them directly, no host hop)
    def tensor(self, which, torch):
        ptr, cnt = C.c_void_p(), C.c_int64()
        self._lib_mod.check(self.lib.elph_dev_buffer(self.h, which, C.byref(ptr), C.byref(cnt)))

        class _Buf:       # minimal __cuda_array_interface__ carrier
            pass
        b = _Buf()
        b.__cuda_array_interface__ = {"shape": (int(cnt.value),), "typestr": "<f8", "data": (int(ptr.value), False), "version": 2}
        return torch.as_tensor(b, device="cuda")

    def use_stream(self, cuda_stream_ptr):
        self._lib_mod.check(self.lib.elph_set_stream(self.h, C.c_void_p(cuda_stream_ptr)))

    def close(self):
        if self.h:
            self.lib.elph_destroy(self.h)
            self.h = None


class ShardedCG:
    """Un-preconditioned CG on MtM x = b for ONE Holstein fermion matrix, tau-slabs over comm.world ranks."""

    def __init__(self, comm, nsites, ltau, table, cosht, sinht, backend_factory=None, device=None):
        self.comm = comm
        self.P, self.rank = comm.world, comm.rank
        assert ltau % self.P == 0, "Ltau must be divisible by the number of ranks"
        self.N, self.Ltau = int(nsites), int(ltau)
        self.Lloc = self.Ltau // self.P
        self.t0 = self.rank * self.Lloc
        if self.Lloc < 1:
            raise ValueError("more ranks than time slices")
        dev = comm.local_rank if device is None else device
        factory = backend_factory or (lambda: GpuBackend(self.N, self.Lloc + 2, table, cosht, sinht, dev))
        self.be = factory()
        # device-resident communication (RCCL acts directly on the solver's buffers, kernels and collectives ordered on
        # one stream, no host round trip per iteration) whenever the communicator runs on GPUs
        self.dev = None
        if getattr(comm, "backend", None) == "nccl" and isinstance(self.be, GpuBackend):
            torch = comm.torch
            stream = torch.cuda.Stream()             # a real (non-null) stream shared by the kernels and the collectives
            self.be.use_stream(stream.cuda_stream)
            t = {k: self.be.tensor(k, torch) for k in (PAP, RR, BB, RVEC, XVEC)}
            with torch.cuda.stream(stream):
                gbuf = torch.empty(self.P * self.Lloc, dtype=torch.float64, device="cuda")
            self.dev = dict(torch=torch, t=t, gbuf=gbuf, r=t[RVEC].view(self.Lloc + 2, self.N), stream=stream)

    # ---- local views of global (N, Ltau) reference-layout arrays: own slices + one halo slice on each side
    def _taus(self):
        return np.arange(self.t0 - 1, self.t0 + self.Lloc + 1) % self.Ltau

    def update_model(self, expV_global):
        """expV_global: (N*Ltau,) reference layout of exp(-dtau V) (update_model!, HolsteinModels.jl:526-549)."""
        Eg = np.asarray(expV_global).reshape(self.N, self.Ltau)
        taus = self._taus()
        E = Eg[:, taus].copy()
        E[:, taus == 0] *= -1.0          # the anti-periodic corner: M[1, Ltau] = +B(1)
        E[:, 0] = 0.0                    # no wrap inside the local operator
        self.be.set_expV(E)

    # ---- cross-rank combination of the per-slice partial sums of the OWNED slices
    def _combine(self, which):
        if self.dev is not None:
            d = self.dev
            t = d["t"][which]
            own = t[1:1 + self.Lloc].contiguous()
            self.comm.dist.all_gather_into_tensor(d["gbuf"], own)
            total = d["gbuf"].sum()                              # same data, same kernel on every rank => same bits
            t.zero_()
            t[0] = total
            return None
        own = self.be.read(which, 1, self.Lloc)
        total = float(np.sum(self.comm.allgather(own)))      # same array, same order on every rank => same bits
        buf = np.zeros(self.Lloc + 2)
        buf[0] = total
        self.be.write(which, 0, buf)
        return total

    def _exchange_r_halo(self):
        N, Ll = self.N, self.Lloc
        if self.dev is not None:
            dist, r = self.comm.dist, self.dev["r"]
            prev, nxt = (self.rank - 1) % self.P, (self.rank + 1) % self.P
            if self.P == 1:
                r[0].copy_(r[Ll]); r[Ll + 1].copy_(r[1])
                return
            send_first, send_last = r[1].contiguous(), r[Ll].contiguous()
            if self.P == 2:     # both messages go to the same peer: pair them by order
                ops = [dist.P2POp(dist.isend, send_first, prev), dist.P2POp(dist.irecv, r[Ll + 1], nxt),
                       dist.P2POp(dist.isend, send_last, nxt), dist.P2POp(dist.irecv, r[0], prev)]
            else:
                ops = [dist.P2POp(dist.isend, send_first, prev), dist.P2POp(dist.isend, send_last, nxt),
                       dist.P2POp(dist.irecv, r[0], prev), dist.P2POp(dist.irecv, r[Ll + 1], nxt)]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            return
        first = self.be.read(RVEC, 1 * N, N)                  # own first slice -> previous rank's upper halo
        last = self.be.read(RVEC, Ll * N, N)                  # own last slice  -> next rank's lower halo
        from_prev, from_next = self.comm.ring_exchange(send_to_prev=first, send_to_next=last)
        self.be.write(RVEC, 0, from_prev)
        self.be.write(RVEC, (Ll + 1) * N, from_next)

    def solve(self, b_global, tol=1e-5, maxiter=10000, kmax=1e12, check_every=8):
        """Returns (x_global (N*Ltau,), iterations, done_flag) — identical on every rank."""
        if self.dev is not None:
            with self.dev["torch"].cuda.stream(self.dev["stream"]):
                return self._solve(b_global, tol, maxiter, kmax, check_every)
        return self._solve(b_global, tol, maxiter, kmax, check_every)

    def _solve(self, b_global, tol, maxiter, kmax, check_every):
        bg = np.asarray(b_global).reshape(self.N, self.Ltau)
        self.be.begin(bg[:, self._taus()], tol, maxiter, kmax)
        self._combine(RR)
        self._combine(BB)
        self.be.state0()
        it, done, eps = 0, 0, np.nan
        launched = 0
        while not done and launched <= maxiter + 1:
            for _ in range(check_every):
                self.be.ap()
                self._combine(PAP)
                self.be.xr()
                self._combine(RR)
                self._exchange_r_halo()
                launched += 1
            it, done, eps = self.be.status()
        x_own = self.be.read(XVEC, 1 * self.N, self.Lloc * self.N).reshape(self.Lloc, self.N)   # device layout (tau, site)
        x_all = self.comm.allgather(x_own.reshape(-1)).reshape(self.P * self.Lloc, self.N)
        return np.ascontiguousarray(x_all.T).reshape(-1), it, done

    def run_iterations(self, k):
        """Exactly k CG iterations (no status check; with tol = 0 the stop test never fires) — used by bench.py."""
        def body():
            for _ in range(k):
                self.be.ap()
                self._combine(PAP)
                self.be.xr()
                self._combine(RR)
                self._exchange_r_halo()
        if self.dev is not None:
            with self.dev["torch"].cuda.stream(self.dev["stream"]):
                body()
                self.dev["stream"].synchronize()
        else:
            body()
            self.be.status()

    def prepare(self, b_global, tol=0.0, maxiter=1 << 40):
        bg = np.asarray(b_global).reshape(self.N, self.Ltau)
        ctx = self.dev["torch"].cuda.stream(self.dev["stream"]) if self.dev is not None else None
        if ctx is not None:
            ctx.__enter__()
        try:
            self.be.begin(bg[:, self._taus()], tol, maxiter, 1e300)
            self._combine(RR)
            self._combine(BB)
            self.be.state0()
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)

    def close(self):
        self.be.close()


# =====================================================================================================================
# Spatial slabs (the decomposition SURVEY §8e / the north_star describe)
# =====================================================================================================================

def mtm_dependency_closure(own_sites, table0):
    """Sites of p that  z = Mᵀ(M p)  on `own_sites` depends on, and the bonds that carry the dependency.

    table0: (nb, 2) 0-based bonds in the order the checkerboard applies them (bond 0 first in M; Mᵀ applies them last
    to first — Checkerboard.jl:57-141).  Walking the factors backwards from the output: a bond that touches the current
    set pulls in its other end.  The τ-shift, exp(-ΔτV) and the ± are pointwise in the site index and add nothing."""
    S = np.zeros(int(table0.max()) + 1 if table0.size else 0, dtype=bool)
    S[np.asarray(own_sites)] = True
    need = np.zeros(table0.shape[0], dtype=bool)
    nb = table0.shape[0]
    for n in list(range(nb)) + list(range(nb - 1, -1, -1)):      # Mᵀ backwards (bond 0 was applied last), then M backwards
        i, j = table0[n]
        if S[i] or S[j]:
            S[i] = S[j] = True
            need[n] = True
    return S, need


class SpatialSlabs:
    """Row decomposition of a lattice (site = norbits*(l1 + L1*l2) + orbit, L3 = 1) over P ranks along l2, with the ghost
    rows each rank's fused MᵀM needs.  Pure integer set-up, identical on every rank."""

    def __init__(self, norbits, L1, L2, table, P):
        self.ns, self.L1, self.L2, self.P = int(norbits), int(L1), int(L2), int(P)
        self.row = self.ns * self.L1                              # sites per row of cells
        self.N = self.row * self.L2
        t0 = np.asarray(table, dtype=np.int64) - 1
        self.starts = [(q * self.L2) // self.P for q in range(self.P + 1)]
        if min(np.diff(self.starts)) < 1:
            raise ValueError("more ranks than rows of cells")
        self.slabs = [self._slab(q, t0) for q in range(self.P)]

    def _slab(self, q, t0):
        r0, r1 = self.starts[q], self.starts[q + 1]
        R = r1 - r0
        own_sites = np.arange(r0 * self.row, r1 * self.row)
        if self.P == 1:
            return dict(R=R, lo=0, hi=0, rows=np.arange(self.L2), bonds=np.arange(t0.shape[0]), r0=r0)
        S, need = mtm_dependency_closure(own_sites, t0)
        rows_needed = np.unique(np.nonzero(S)[0] // self.row)
        lo = hi = 0
        for g in rows_needed:
            d = (g - r0) % self.L2
            if d < R:
                continue
            up, down = d - (R - 1), self.L2 - d                  # distance above the last / below the first own row
            if up <= down:
                hi = max(hi, up)
            else:
                lo = max(lo, down)
        if lo + R + hi > self.L2:
            raise ValueError(f"rank {q}: own rows {R} + ghost rows {lo}+{hi} exceed the {self.L2} rows of the lattice")
        rows = (np.arange(r0 - lo, r1 + hi)) % self.L2            # local row j -> global row
        loc_of = -np.ones(self.L2, dtype=np.int64)
        loc_of[rows] = np.arange(rows.size)
        gi, gj = t0[:, 0] // self.row, t0[:, 1] // self.row
        li, lj = loc_of[gi], loc_of[gj]
        inc = (li >= 0) & (lj >= 0) & (np.abs(li - lj) <= 1)      # both ends in the slab, no wrap through its open ends
        if (need & ~inc).any():
            raise ValueError(f"rank {q}: a bond the own rows depend on leaves the slab")
        return dict(R=R, lo=lo, hi=hi, rows=rows, bonds=np.nonzero(inc)[0], r0=r0)

    def local_table(self, q, table):
        """1-based local neighbour table of rank q's slab, bonds in the global checkerboard order."""
        sl = self.slabs[q]
        t0 = np.asarray(table, dtype=np.int64) - 1
        loc_of = -np.ones(self.L2, dtype=np.int64)
        loc_of[sl["rows"]] = np.arange(sl["rows"].size)
        b = t0[sl["bonds"]]
        loc = loc_of[b // self.row] * self.row + (b % self.row)
        return loc + 1

    def global_sites(self, q):
        sl = self.slabs[q]
        return (sl["rows"][:, None] * self.row + np.arange(self.row)[None, :]).reshape(-1)


class SpatialShardedCG:
    """Un-preconditioned CG on MᵀM x = b for ONE Holstein fermion matrix, slabs of rows of cells over comm.world ranks.

    Rank q's handle lives on its slab: own rows + the ghost rows found by `mtm_dependency_closure` (2 below and 2 above
    for the even-aligned square lattice, where only the last colour crosses the slab boundary — SURVEY §8e).  The fused
    kernel then yields the exact z = MᵀM p on the own rows with NO exchange inside the mat-vec; per iteration the ranks
    exchange the ghost rows of r once (the checkerboard boundary rows: 2 x L1 x Ltau doubles each way at config C =
    41 KB) and rebuild p on the ghosts locally (p = r + βp is pointwise), and combine the two inner products from
    per-slice partial sums over own sites (elph_set_dot_range) so every rank takes bit-identical α, β, stop decisions."""

    def __init__(self, comm, norbits, L1, L2, ltau, table, cosht, sinht, backend_factory=None, device=None):
        self.comm, self.P, self.rank = comm, comm.world, comm.rank
        self.Ltau = int(ltau)
        self.slabs = SpatialSlabs(norbits, L1, L2, table, self.P)
        self.N = self.slabs.N
        sl = self.slabs.slabs[self.rank]
        self.sl = sl
        self.row = self.slabs.row
        self.Nloc = sl["rows"].size * self.row
        self.own_lo, self.own_n = sl["lo"] * self.row, sl["R"] * self.row
        ltab = self.slabs.local_table(self.rank, table)
        c, s = np.asarray(cosht)[sl["bonds"]], np.asarray(sinht)[sl["bonds"]]
        dev = comm.local_rank if device is None else device
        factory = backend_factory or (lambda N, L, t, cc, ss: GpuBackend(N, L, t, cc, ss, dev))
        self.be = factory(self.Nloc, self.Ltau, ltab, c, s)
        self.gsites = self.slabs.global_sites(self.rank)
        if self.P > 1:
            self.be.set_dot_range(self.own_lo, self.own_lo + self.own_n)
            prev, nxt = (self.rank - 1) % self.P, (self.rank + 1) % self.P
            sp, sn = self.slabs.slabs[prev], self.slabs.slabs[nxt]
            if sl["lo"] > sp["R"] or sl["hi"] > sn["R"]:
                raise ValueError("ghost rows reach beyond the neighbouring rank: use fewer ranks")
            # what the neighbours need from me: next wants my top sn.lo rows, prev wants my bottom sp.hi rows
            self.n_to_next, self.n_to_prev = sn["lo"] * self.row, sp["hi"] * self.row
            self.n_from_prev, self.n_from_next = sl["lo"] * self.row, sl["hi"] * self.row
        self.dev = None
        if getattr(comm, "backend", None) == "nccl" and isinstance(self.be, GpuBackend):
            torch = comm.torch
            stream = torch.cuda.Stream()
            self.be.use_stream(stream.cuda_stream)
            t = {k: self.be.tensor(k, torch) for k in (PAP, RR, BB, RVEC, XVEC)}
            with torch.cuda.stream(stream):
                gbuf = torch.empty(self.P * self.Ltau, dtype=torch.float64, device="cuda")
            self.dev = dict(torch=torch, t=t, gbuf=gbuf, r=t[RVEC].view(self.Ltau, self.Nloc), stream=stream)

    def update_model(self, expV_global):
        Eg = np.asarray(expV_global).reshape(self.N, self.Ltau)
        self.be.set_expV(Eg[self.gsites, :])

    def _local(self, v_global):
        return np.asarray(v_global).reshape(self.N, self.Ltau)[self.gsites, :]

    def _combine(self, which):
        if self.P == 1:
            return
        if self.dev is not None:
            d = self.dev
            t = d["t"][which]
            self.comm.dist.all_gather_into_tensor(d["gbuf"], t.contiguous())
            total = d["gbuf"].sum()
            t.zero_()
            t[0] = total
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
        if self.dev is not None:
            dist, r, torch = self.comm.dist, self.dev["r"], self.dev["torch"]
            prev, nxt = (self.rank - 1) % self.P, (self.rank + 1) % self.P
            to_prev = r[:, lo:lo + self.n_to_prev].contiguous()
            to_next = r[:, lo + n - self.n_to_next:lo + n].contiguous()
            from_prev = torch.empty((self.Ltau, self.n_from_prev), dtype=torch.float64, device="cuda")
            from_next = torch.empty((self.Ltau, self.n_from_next), dtype=torch.float64, device="cuda")
            if self.P == 2:
                ops = [dist.P2POp(dist.isend, to_prev, prev), dist.P2POp(dist.irecv, from_next, nxt),
                       dist.P2POp(dist.isend, to_next, nxt), dist.P2POp(dist.irecv, from_prev, prev)]
            else:
                ops = [dist.P2POp(dist.isend, to_prev, prev), dist.P2POp(dist.isend, to_next, nxt),
                       dist.P2POp(dist.irecv, from_prev, prev), dist.P2POp(dist.irecv, from_next, nxt)]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            r[:, :lo].copy_(from_prev)
            r[:, lo + n:].copy_(from_next)
            return
        to_prev = self.be.read_rows(RVEC, lo, self.n_to_prev) if self.n_to_prev else np.zeros((self.Ltau, 0))
        to_next = self.be.read_rows(RVEC, lo + n - self.n_to_next, self.n_to_next) if self.n_to_next else np.zeros((self.Ltau, 0))
        from_prev, from_next = self.comm.ring_exchange(send_to_prev=to_prev.reshape(-1), send_to_next=to_next.reshape(-1),
                                                       recv_prev_n=self.Ltau * self.n_from_prev,
                                                       recv_next_n=self.Ltau * self.n_from_next)
        if self.n_from_prev:
            self.be.write_rows(RVEC, 0, np.asarray(from_prev).reshape(self.Ltau, self.n_from_prev))
        if self.n_from_next:
            self.be.write_rows(RVEC, lo + n, np.asarray(from_next).reshape(self.Ltau, self.n_from_next))

    def _iteration(self):
        self.be.ap()
        self._combine(PAP)
        self.be.xr()
        self._combine(RR)
        self._exchange_r_halo()

    def _begin(self, b_global, tol, maxiter, kmax):
        self.be.begin(self._local(b_global), tol, maxiter, kmax)
        self._combine(RR)
        self._combine(BB)
        self.be.state0()

    def _ctx(self):
        import contextlib
        return self.dev["torch"].cuda.stream(self.dev["stream"]) if self.dev is not None else contextlib.nullcontext()

    def solve(self, b_global, tol=1e-5, maxiter=10000, kmax=1e12, check_every=8):
        """Returns (x_global (N*Ltau,), iterations, done_flag) — identical on every rank."""
        with self._ctx():
            self._begin(b_global, tol, maxiter, kmax)
            it, done, launched = 0, 0, 0
            while not done and launched <= maxiter + 1:
                for _ in range(check_every):
                    self._iteration()
                    launched += 1
                it, done, _ = self.be.status()
            x_own = self.be.read_rows(XVEC, self.own_lo, self.own_n)                   # (Ltau, own sites)
        parts = self.comm.allgather_object(x_own) if self.P > 1 else [x_own]
        x = np.concatenate(parts, axis=1)                                              # (Ltau, N): ranks own ascending rows
        return np.ascontiguousarray(x.T).reshape(-1), it, done

    def prepare(self, b_global, tol=0.0, maxiter=1 << 40):
        with self._ctx():
            self._begin(b_global, tol, maxiter, 1e300)

    def run_iterations(self, k):
        with self._ctx():
            for _ in range(k):
                self._iteration()
            if self.dev is not None:
                self.dev["stream"].synchronize()
            else:
                self.be.status()

    def close(self):
        self.be.close()


# =====================================================================================================================
# The in-library sharded solve (csrc/shard.hip + the SHARD form of the resident CG kernel, csrc/cg_wg.hip)
# =====================================================================================================================

class ShardedSolver:
    """ONE un-preconditioned solve of MᵀM x = b (x0 = 0) over comm.world GPUs, slabs of rows of cells (`SpatialSlabs`), Holstein
    or bond-phonon (SSH) models.  Everything inside an iteration happens on the devices: the rank's resident CG kernel stores
    its partial sums and the checkerboard boundary rows of the residual into the neighbours' mailboxes (hipIpc-mapped device
    memory; xGMI peer stores between GPUs) and polls its own — the host only all-gathers the 64-byte mailbox handles once
    and provides the barrier between `elph_shard_prepare` and `elph_shard_solve`.  Also runs with several ranks on ONE GPU
    (the test box), which RCCL refuses.

    kind 0 (Holstein): cosht/sinht per bond, `update_model(expV_global)`;
    kind 1 (SSH): `update_model_ssh(cosht_global[Nbonds, Ltau], sinht_global, expDtauMu_global)` — the per-(τ, bond) tables are
    sharded by bond owner: a rank holds the columns of the bonds inside its slab (SSHModels.jl:581-701)."""

    def __init__(self, comm, norbits, L1, L2, ltau, table, kind=0, cosht=None, sinht=None, device=None):
        from . import _lib
        self._lib_mod, self.lib = _lib, _lib.load()
        self.comm, self.P, self.rank = comm, comm.world, comm.rank
        self.kind, self.Ltau = int(kind), int(ltau)
        self.slabs = SpatialSlabs(norbits, L1, L2, table, self.P)
        self.N, self.row = self.slabs.N, self.slabs.row
        sl = self.sl = self.slabs.slabs[self.rank]
        self.Nloc = sl["rows"].size * self.row
        self.own_lo, self.own_n = sl["lo"] * self.row, sl["R"] * self.row
        self.gsites = self.slabs.global_sites(self.rank)
        ltab = np.ascontiguousarray(self.slabs.local_table(self.rank, table), dtype=np.int64)
        self.bonds = sl["bonds"]
        prev, nxt = (self.rank - 1) % self.P, (self.rank + 1) % self.P
        sp, sn = self.slabs.slabs[prev], self.slabs.slabs[nxt]
        if self.P > 1 and (sl["lo"] > sp["R"] or sl["hi"] > sn["R"]):
            raise ValueError("ghost rows reach beyond the neighbouring rank: use fewer ranks")
        n_to_next = sn["lo"] * self.row if self.P > 1 else 0         # next rank's ghosts below its own rows = my top rows
        n_to_prev = sp["hi"] * self.row if self.P > 1 else 0         # previous rank's ghosts above its own rows = my bottom rows
        cap = max(max(s["lo"], s["hi"]) for s in self.slabs.slabs) * self.row
        dev = comm.device_index() if device is None else device
        self.h = _lib.Handle()
        nb = ltab.shape[0]
        c = np.ascontiguousarray(np.asarray(cosht)[self.bonds]) if (kind == 0 and nb) else None
        s = np.ascontiguousarray(np.asarray(sinht)[self.bonds]) if (kind == 0 and nb) else None
        _lib.check(self.lib.elph_create(C.byref(self.h), self.kind, self.Nloc, self.Ltau, nb, _lib.iptr(ltab) if nb else None,
                                        _lib.dptr(c) if c is not None else None, _lib.dptr(s) if s is not None else None, dev))
        hbuf = (C.c_ubyte * 64)()
        gs = np.ascontiguousarray(self.gsites, dtype=np.int64)
        _lib.check(self.lib.elph_shard_create(self.h, self.rank, self.P, self.own_lo, self.own_n, n_to_prev, n_to_next, cap,
                                              self.N, int(sl["r0"]) * self.row, _lib.iptr(gs), C.cast(hbuf, C.c_void_p)))
        self.hf = None
        self._full = (np.ascontiguousarray(table, dtype=np.int64), cosht, sinht, dev)
        if self.P > 1:
            allh = b"".join(comm.allgather_object(bytes(hbuf)))
            self._allh = C.create_string_buffer(allh, len(allh))
            _lib.check(self.lib.elph_shard_connect(self.h, C.cast(self._allh, C.c_void_p)))
        comm.barrier()

    def _local(self, v_global):
        return np.ascontiguousarray(np.asarray(v_global).reshape(self.N, self.Ltau)[self.gsites, :]).reshape(-1)

    def update_model(self, expV_global):
        """Holstein: exp(-Δτ V) of the whole lattice, reference layout (update_model!, HolsteinModels.jl:526-549)."""
        self._lib_mod.check(self.lib.elph_set_expV(self.h, self._lib_mod.dptr(self._local(expV_global))))

    def update_model_ssh(self, cosht_global, sinht_global, expDtauMu_global):
        """SSH: cosht/sinht[Nbonds, Ltau] in checkerboard order (model.cosht as the reference stores it, transposed to
        bond-major rows) and exp(Δτ μ)[N]; this rank takes the bonds of its slab."""
        c = np.ascontiguousarray(np.asarray(cosht_global).reshape(-1, self.Ltau)[self.bonds]).reshape(-1)
        s = np.ascontiguousarray(np.asarray(sinht_global).reshape(-1, self.Ltau)[self.bonds]).reshape(-1)
        e = np.ascontiguousarray(np.asarray(expDtauMu_global)[self.gsites])
        d = self._lib_mod.dptr
        self._lib_mod.check(self.lib.elph_update_model_ssh(self.h, d(c), d(s), d(e)))

    def setup_kpm(self, expV_global, n=20, buf=0.05, c1=1.0, c2=1.0, seed=7, e_min=None, e_max=None):
        """KPM preconditioner under sharding (Holstein; SSH with expV_global = (cosht, sinht, exp(Δτμ)) of the whole lattice):
        a second handle on the WHOLE lattice carries Ē, the Arnoldi bounds, the
        orders and coefficients — set up identically on every rank (same inputs, same start vectors) — and runs the per-frequency
        Chebyshev recursion; see elph_shard_solve_kpm.  Returns (active, lam_lo, lam_hi)."""
        lm, lib = self._lib_mod, self.lib
        tab, c, s, dev = self._full
        if self.hf is None:
            self.hf = lm.Handle()
            if self.kind == 0:
                lm.check(lib.elph_create(C.byref(self.hf), 0, self.N, self.Ltau, tab.shape[0], lm.iptr(tab),
                                         lm.dptr(np.ascontiguousarray(c)), lm.dptr(np.ascontiguousarray(s)), dev))
            else:
                lm.check(lib.elph_create(C.byref(self.hf), 1, self.N, self.Ltau, tab.shape[0], lm.iptr(tab), None, None, dev))
            lm.check(lib.elph_kpm_create(self.hf, n, buf, c1, c2))
        if self.kind == 0:
            lm.check(lib.elph_set_expV(self.hf, lm.dptr(np.ascontiguousarray(np.asarray(expV_global).reshape(-1)))))
        else:
            # bond phonons: expV_global = (cosht[Nbonds, Ltau], sinht[Nbonds, Ltau], exp(dtau mu)[N]) of the WHOLE lattice — the expansion
            # takes the tau-means of the hopping tables (update_A!, KPMPreconditioners.jl:355-381)
            cg, sg, eg = expV_global
            d = lm.dptr
            lm.check(lib.elph_update_model_ssh(self.hf, d(np.ascontiguousarray(np.asarray(cg).reshape(-1))),
                                               d(np.ascontiguousarray(np.asarray(sg).reshape(-1))), d(np.ascontiguousarray(eg))))
        rng = np.random.default_rng(seed)
        bmax, bmin = rng.standard_normal(self.N), rng.standard_normal(self.N)
        act, lo, hi = C.c_int(), C.c_double(), C.c_double()
        lm.check(lib.elph_kpm_setup(self.hf, lm.dptr(bmax), lm.dptr(bmin), float("nan") if e_min is None else e_min,
                                    float("nan") if e_max is None else e_max, C.byref(act), C.byref(lo), C.byref(hi)))
        return act.value, lo.value, hi.value

    def solve(self, b_global, tol=1e-5, maxiter=10000, kmax=1e12, precond=False):
        """Returns (x_global (N·Ltau,), iterations, done) — identical on every rank."""
        b = self._local(b_global)
        x = np.zeros(self.Nloc * self.Ltau)
        it, done, eps = C.c_int64(), C.c_int(), C.c_double()
        self._lib_mod.check(self.lib.elph_shard_prepare(self.h))
        self.comm.barrier()                                  # every mailbox is zero before any rank stores into it
        import time
        t_lib = time.perf_counter()
        if precond:
            assert self.hf is not None, "setup_kpm first"
            self._lib_mod.check(self.lib.elph_shard_solve_kpm(self.h, self.hf, self._lib_mod.dptr(x), self._lib_mod.dptr(b), tol, maxiter,
                                                              kmax, C.byref(it), C.byref(done), C.byref(eps)))
        else:
            self._lib_mod.check(self.lib.elph_shard_solve(self.h, self._lib_mod.dptr(x), self._lib_mod.dptr(b), tol, maxiter, kmax,
                                                          C.byref(it), C.byref(done), C.byref(eps)))
        self.last_solve_s = time.perf_counter() - t_lib      # inside the library (host pointers in and out), without the gather below
        x_own = x.reshape(self.Nloc, self.Ltau)[self.own_lo:self.own_lo + self.own_n, :]
        parts = self.comm.allgather_object(x_own) if self.P > 1 else [x_own]
        self.eps = float(eps.value)
        return np.ascontiguousarray(np.concatenate(parts, axis=0)).reshape(-1), int(it.value), int(done.value)

    def iterate(self, b_global, k):
        """Exactly k iterations (no stop test); returns this rank's HIP-event time of the launch in ms (bench.py)."""
        b = self._local(b_global)
        ms = C.c_double()
        self._lib_mod.check(self.lib.elph_shard_prepare(self.h))
        self.comm.barrier()
        self._lib_mod.check(self.lib.elph_shard_iterate(self.h, self._lib_mod.dptr(b), int(k), C.byref(ms)))
        return float(ms.value)

    def close(self):
        if self.h:
            self.lib.elph_shard_destroy(self.h)
            self.lib.elph_destroy(self.h)
            self.h = None
        if getattr(self, "hf", None):
            self.lib.elph_destroy(self.hf)
            self.hf = None
