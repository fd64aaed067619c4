"""ONE conjugate-gradient solve sharded over several GPUs (north_star "the lattice shards over the GPUs of a node
along the spatial axis with halo exchange of the checkerboard boundary slice once per mat-vec").

Decomposition: slabs of rows of cells along the slowest spatial index.  A rank's handle lives on its slab = own rows + the ghost
rows the fused z = Mᵀ(M p) reads (`mtm_dependency_closure`: 2 + 2 rows on the even-aligned square lattice, 1 + 1 on honeycomb), so
the unmodified mat-vec yields the exact z on the own rows with no exchange inside it.  Everything inside an iteration happens on the
devices (csrc/shard.hip + the SHARD form of the resident CG kernel, csrc/cg_wg.hip): partial sums and boundary rows travel as
device-initiated stores into the neighbours' mailboxes.  This module is the host side: the slab arithmetic (`SpatialSlabs`) and
`ShardedSolver`, which creates the handles, all-gathers the 64-byte mailbox handles once and provides the barrier before a solve
(through `dist.Comm` / `dist.HybridComm`: torch.distributed "nccl" = RCCL on GPU boxes, "gloo" in the CPU tests).

(The host-driven protocol of round 1 — one collective per phase between the kernels of a step-wise API — is a test helper now:
tests/protocol_reference.py.)
"""
import ctypes as C
import os

import numpy as np


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

    def __init__(self, norbits, L1, L2, table, P, ring=False):
        """ring: close every slab into a RING — the bonds that leave through its last row re-enter at its first row.  The own rows do
        not see the difference (the ring bond lies beyond the dependency closure that fixed the ghost rows: it only stirs the outermost
        ghost rows, whose values are discarded anyway), but the slab becomes a periodic rectangle in the reference's colouring, which the
        library's register-exchange forms recognise (GRID / HGRID, csrc/cg_fast_common.h)."""
        self.ring = bool(ring)
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
        ringb = np.zeros_like(inc)
        if self.ring:
            # bonds between the slab's last row and the global row above it: their outer end is re-attached to the slab's first row
            nxt = (rows[-1] + 1) % self.L2
            ringb = ~inc & (((li == rows.size - 1) & (gj == nxt)) | ((lj == rows.size - 1) & (gi == nxt)))
            # the ring must not change the colouring (the library recomputes colours as maximal runs of site-disjoint bonds): a ring bond
            # that meets another bond of its own colour at the first row — a slab whose first and last rows do not sit at a colour
            # boundary — would be applied as a colour of its own, in the middle of the sweep: then the slab stays open
            def ncolours(sel):
                loc_of2 = loc_of.copy()
                loc_of2[nxt] = 0 if loc_of2[nxt] < 0 else loc_of2[nxt]
                b = t0[np.nonzero(sel)[0]]
                loc = loc_of2[b // self.row] * self.row + (b % self.row)
                seen, n = set(), 1
                for i, j in loc:
                    if i in seen or j in seen:
                        seen, n = set(), n + 1
                    seen.add(int(i)); seen.add(int(j))
                return n
            if ringb.any() and ncolours(inc | ringb) != ncolours(inc):
                ringb[:] = False
        if not ringb.any():
            return dict(R=R, lo=lo, hi=hi, rows=rows, bonds=np.nonzero(inc)[0], r0=r0, ring_next=-1)
        return dict(R=R, lo=lo, hi=hi, rows=rows, bonds=np.nonzero(inc | ringb)[0], r0=r0, ring_next=int((rows[-1] + 1) % self.L2) if self.ring else -1)

    def local_table(self, q, table):
        """1-based local neighbour table of rank q's slab, bonds in the global checkerboard order."""
        sl = self.slabs[q]
        t0 = np.asarray(table, dtype=np.int64) - 1
        loc_of = -np.ones(self.L2, dtype=np.int64)
        if sl.get("ring_next", -1) >= 0:
            loc_of[sl["ring_next"]] = 0                          # the ring: the row above the slab's last row IS its first row
        loc_of[sl["rows"]] = np.arange(sl["rows"].size)           # (a slab that covers the whole lattice: the true row wins — the same one)
        b = t0[sl["bonds"]]
        loc = loc_of[b // self.row] * self.row + (b % self.row)
        assert (loc >= 0).all()
        return loc + 1

    def global_sites(self, q):
        sl = self.slabs[q]
        return (sl["rows"][:, None] * self.row + np.arange(self.row)[None, :]).reshape(-1)


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

    def __init__(self, comm, norbits, L1, L2, ltau, table, kind=0, cosht=None, sinht=None, device=None, selftest=True):
        from . import _lib
        self._lib_mod, self.lib = _lib, _lib.load()
        self.comm, self.P, self.rank = comm, comm.world, comm.rank
        self.kind, self.Ltau = int(kind), int(ltau)
        # Holstein slabs are closed into rings (exact on the own rows; lets the library run its register-exchange forms on the slab)
        ring = (int(kind) == 0 and self.P > 1 and os.environ.get("ELPH_SHARD_RING", "1") != "0")
        self.slabs = SpatialSlabs(norbits, L1, L2, table, self.P, ring=ring)
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
            # every rank learns of every rank's failure to map a peer's mailbox (hipIpcOpenMemHandle refused between two devices, say) and raises
            # with it: nobody is left waiting in the barrier below, and sharded_rccl.make_solver can fall back to collectives on all ranks alike
            rc = self.lib.elph_shard_connect(self.h, C.cast(self._allh, C.c_void_p))
            err = self.lib.elph_last_error().decode() if rc else ""
            failed = [(r, e) for r, c_, e in comm.allgather_object((self.rank, rc, err)) if c_]
            if failed:
                raise _lib.ElphError(-2, "mailbox mapping failed: " + "; ".join(f"rank {r}: {e}" for r, e in failed))
        comm.barrier()
        # preflight of the mailbox protocol between every pair of ranks: a broken peer mapping shows HERE, with the silent ranks named,
        # instead of as a time-out inside the first solve.  selftest_us[q]: mean time from this rank's store to the sight of rank q's.
        self.selftest_us = None
        if selftest and os.environ.get("ELPH_SHARD_NO_SELFTEST") != "1":
            self.selftest_us = self.selftest()

    def selftest(self, rounds=64):
        """elph_shard_selftest: `rounds` lock-step granule exchanges between all pairs of ranks; returns us per round per peer."""
        us = np.zeros(self.P)
        worst = C.c_double()
        self._lib_mod.check(self.lib.elph_shard_prepare(self.h))
        self.comm.barrier()
        rc = self.lib.elph_shard_selftest(self.h, int(rounds), self._lib_mod.dptr(us), C.byref(worst))
        err = self.lib.elph_last_error().decode() if rc else ""
        bad = self.comm.allgather_object((self.rank, rc, err)) if self.P > 1 else [(self.rank, rc, err)]
        self.comm.barrier()
        failed = [(r, e) for r, c, e in bad if c]
        if failed:
            raise self._lib_mod.ElphError(-2, "; ".join(f"rank {r}: {e}" for r, e in failed))
        self.selftest_slowest_us = float(worst.value)
        return us

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

    # ---- the callers of the solve on a sharded lattice (include/elph_gpu.h: elph_shard_set_collectives, elph_shard_ldiv,
    # ---- elph_shard_fermion_force_*, elph_hmc_update on the slab handle) ---------------------------------------------------
    BARRIER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
    ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)

    def install_collectives(self):
        """Register the two host collectives the sharded callers need (a barrier; an in-place sum over the ranks, added in rank order
        so that every rank gets the same bits) — here through the communicator that already carries the mailbox handles."""
        if getattr(self, "_coll", None):
            return

        def bar(ctx):
            try:
                self.comm.barrier()
                return 0
            except Exception:                                # never let an exception cross the C boundary
                return 1

        def ars(ctx, buf, n):
            try:
                a = np.ctypeslib.as_array(buf, shape=(n,))
                parts = self.comm.allgather_object(a.copy()) if self.P > 1 else [a.copy()]
                tot = np.array(parts[0], dtype=np.float64, copy=True)
                for q in parts[1:]:
                    tot += q
                a[:] = tot
                return 0
            except Exception:
                return 1

        self._coll = (self.BARRIER_FN(bar), self.ALLREDUCE_FN(ars))
        self._lib_mod.check(self.lib.elph_shard_set_collectives(self.h, C.cast(self._coll[0], C.c_void_p), C.cast(self._coll[1], C.c_void_p), None))

    def set_solver(self, tol, maxiter, kmax=1e12):
        self._lib_mod.check(self.lib.elph_solver_set(self.h, float(tol), int(maxiter), float(kmax)))

    def _gather_own(self, v_slab):
        """Own rows of a slab vector from every rank -> the vector on the whole lattice (identical on every rank)."""
        own = np.asarray(v_slab).reshape(self.Nloc, self.Ltau)[self.own_lo:self.own_lo + self.own_n, :]
        parts = self.comm.allgather_object(own) if self.P > 1 else [own]
        return np.ascontiguousarray(np.concatenate(parts, axis=0)).reshape(-1)

    def ldiv(self, b_global, precond=False, maxiter=0):
        """ldiv!(x, model, b[, P]) over the ranks — Models.jl:74-137,139-186: (x_global, iters, residual_error, flag), identical everywhere."""
        self.install_collectives()
        lm = self._lib_mod
        b = self._local(b_global)
        x = np.zeros(self.Nloc * self.Ltau)
        it, res, fl = C.c_int64(), C.c_double(), C.c_int()
        lm.check(self.lib.elph_shard_ldiv(self.h, self.hf if precond else None, lm.dptr(x), lm.dptr(b), 1 if precond else 0, int(maxiter),
                                          C.byref(it), C.byref(res), C.byref(fl)))
        return self._gather_own(x), int(it.value), float(res.value), int(fl.value)

    def fermion_force_holstein(self, x_global, lam, lam2, mu, dtau, phi_p, phi_m, precond=False, power=1.0):
        """update_model! + calc_O⁻¹Λϕ! + calc_dSfdx! of the Holstein model over the ranks (HMC.jl:790-915): returns
        (dSf/dx on the whole lattice, O⁻¹Λϕ₊, O⁻¹Λϕ₋, iters, flag)."""
        self.install_collectives()
        lm, d = self._lib_mod, self._lib_mod.dptr
        site = lambda a: np.ascontiguousarray(np.asarray(a)[self.gsites])      # noqa: E731
        F = np.zeros(self.Nloc * self.Ltau)
        Xp, Xm = np.zeros_like(F), np.zeros_like(F)
        it, fl = C.c_int64(), C.c_int()
        lm.check(self.lib.elph_shard_fermion_force_holstein(
            self.h, self.hf if precond else None, d(self._local(x_global)), d(site(lam)), d(site(lam2)), d(site(mu)), float(dtau),
            d(self._local(phi_p)), d(self._local(phi_m)), 1 if precond else 0, float(power), d(F), d(Xp), d(Xm), C.byref(it), C.byref(fl)))
        return self._gather_own(F), self._gather_own(Xp), self._gather_own(Xm), int(it.value), int(fl.value)

    def owned_bonds(self):
        """Local bonds (positions in this slab's table) whose force bracket this rank owns: the bond's first site (i < j in the
        reference's tables, Lattices.jl:323-340) lies in the own rows.  Every bond of the lattice has exactly one owner."""
        ltab = self.slabs.local_table(self.rank, self._full[0]) - 1
        return np.nonzero((ltab[:, 0] >= self.own_lo) & (ltab[:, 0] < self.own_lo + self.own_n))[0]

    def fermion_force_ssh(self, rhs_p, rhs_m, nbonds_global, precond=False, power=1.0):
        """The two solves of calc_O⁻¹Λϕ! and the bond brackets of muldMdx! (SSHModels.jl:707-829) over the ranks: returns
        (q[nbonds_global, Ltau] in the global checkerboard order, iters, flag) — what elph_fermion_force_ssh returns for one handle."""
        self.install_collectives()
        lm, d = self._lib_mod, self._lib_mod.dptr
        nbl = len(self.bonds)
        q = np.zeros(nbl * self.Ltau)
        it, fl = C.c_int64(), C.c_int()
        lm.check(self.lib.elph_shard_fermion_force_ssh(self.h, self.hf if precond else None, d(self._local(rhs_p)), d(self._local(rhs_m)),
                                                       1 if precond else 0, float(power), d(q), None, None, C.byref(it), C.byref(fl)))
        q = q.reshape(nbl, self.Ltau)
        own = self.owned_bonds()
        mine = (np.asarray(self.bonds)[own], q[own])
        parts = self.comm.allgather_object(mine) if self.P > 1 else [mine]
        out = np.full((int(nbonds_global), self.Ltau), np.nan)
        for gb, qq in parts:
            assert np.all(np.isnan(out[gb])), "a bond with two owners"
            out[gb] = qq
        assert not np.isnan(out).any(), "a bond without an owner"
        return out, int(it.value), int(fl.value)

    def ssh_phonon_columns(self, checkerboard_perm, phonon_to_bond):
        """The slab's phonon columns for a bond-phonon model: (global phonon index of every local column, its local checkerboard position
        1-based, owned 1.0 / 0.0).  A slab holds the phonons of its bonds, ghost bonds included; the owner of a phonon is the owner of its
        bond (`owned_bonds`)."""
        cbp = np.asarray(checkerboard_perm, dtype=np.int64)              # raw bond (1-based index - 1) -> checkerboard position (1-based)
        p2b = np.asarray(phonon_to_bond, dtype=np.int64)
        pos_of_phonon = cbp[p2b - 1] - 1                                  # global checkerboard position (0-based) of every phonon's bond
        phonon_at = -np.ones(len(self._full[0]), dtype=np.int64)
        phonon_at[pos_of_phonon] = np.arange(len(p2b))
        own = np.zeros(len(self.bonds), dtype=bool)
        own[self.owned_bonds()] = True
        gcol, cbl, w = [], [], []
        for k, gb in enumerate(np.asarray(self.bonds)):
            ph = phonon_at[gb]
            if ph >= 0:
                gcol.append(int(ph)); cbl.append(k + 1); w.append(1.0 if own[k] else 0.0)
        return np.asarray(gcol, dtype=np.int64), np.asarray(cbl, dtype=np.int64), np.asarray(w, dtype=np.float64)

    def set_bonds(self, nbonds_global):
        """elph_shard_set_bonds: the slab's bonds in the checkerboard numbering of the whole lattice and which of them this rank owns — what a
        KPM-preconditioned HMC update of a bond-phonon model needs to put the τ-means of every bond's hopping onto the full-lattice handle."""
        lm = self._lib_mod
        own = np.zeros(len(self.bonds))
        own[self.owned_bonds()] = 1.0
        gb = np.ascontiguousarray(np.asarray(self.bonds), dtype=np.int64)
        lm.check(self.lib.elph_shard_set_bonds(self.h, lm.iptr(gb), int(nbonds_global), lm.dptr(own)))

    def ghost_stats(self):
        """(ghost exchanges through the mailboxes, through the host collectives) since the shard was created."""
        a, b = C.c_int64(), C.c_int64()
        self._lib_mod.check(self.lib.elph_shard_ghost_stats(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

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
