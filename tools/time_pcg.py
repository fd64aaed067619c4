#!/usr/bin/env python3
"""Resident preconditioned CG (pcg_wg.hip) vs the streaming five-kernel iteration: agreement and time per iteration (config C)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import _lib, configs, models, preconditioners as pc          # noqa: E402
from elphdynamics_amd._lib import check, dptr                                      # noqa: E402

lib = _lib.load()
tag = sys.argv[1] if len(sys.argv) > 1 else "C"
m = configs.make_model(tag, tol=1e-5)
P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
pc.setup_(P, rng=np.random.default_rng(7))
ms = C.c_double()
for nrhs in (1, 2, 4, 8):
    R, B = configs.rhs(m, nrhs)
    res = {}
    for mode in ("resident", "streaming"):
        os.environ["ELPH_PCG_WG"] = "1" if mode == "resident" else "0"
        X = np.zeros_like(B)
        it, rs, fl = models.ldiv_batched_(X, m, B, P=P)
        X[:] = 0
        t0 = time.perf_counter()
        it, rs, fl = models.ldiv_batched_(X, m, B, P=P)
        dt = time.perf_counter() - t0
        res[mode] = (X.copy(), it.copy())
        print(f"{tag} nrhs={nrhs} {mode:9s}: iters {it.tolist()} flags {fl.tolist()} res {rs.max():.2e} solve {1e3*dt:.3f} ms", flush=True)
    a, b = res["resident"], res["streaming"]
    print(f"   resident vs streaming: |dx|/|x| = {np.linalg.norm(a[0]-b[0])/np.linalg.norm(b[0]):.2e}, iters equal: {np.array_equal(a[1], b[1])}")
    os.environ["ELPH_PCG_WG"] = "1"
    for what, name in ((10, "resident"), (3, "streaming")):
        for reps in (32, 320):
            check(lib.elph_bench_prepare(m._h, what, nrhs, dptr(np.ascontiguousarray(B))))
            check(lib.elph_bench_run(m._h, what, nrhs, reps, 0, C.byref(ms)))
        print(f"   {name:9s} preconditioned iteration: {1e3*ms.value/320:.2f} us", flush=True)
m.close()
