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
