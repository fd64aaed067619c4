#!/usr/bin/env python3
"""Workgroup-resident CG (cg_wg.hip) vs the two-kernel streaming iteration: agreement and time per iteration.
usage: python3 tools/time_wg.py [config tags; a trailing ~ adds hopping disorder (Holstein: stddev 0.1)]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import _lib, configs, models          # noqa: E402
from elphdynamics_amd._lib import check                    # noqa: E402

lib = _lib.load()
for tag in (sys.argv[1:] or ["b", "B", "C", "D", "E"]):
    m = configs.make_model(tag.rstrip("~"), tol=1e-5, t_stddev=0.1 if tag.endswith("~") else 0.0)
    us, T, W, G = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    check(lib.elph_bench_wg_info(m._h, 1, C.byref(us), C.byref(T), C.byref(W), C.byref(G)))
    print(f"== {tag}: N={m.Nsites} Ltau={m.Ltau} wg usable={us.value} T={T.value} W={W.value} G={G.value}", flush=True)
    R, B = configs.rhs(m, 4)
    res = {}
    for mode in ("wg", "stream"):
        os.environ["ELPH_NO_WG"] = "0" if mode == "wg" else "1"
        X = np.zeros_like(B)
        it, rs, fl = models.ldiv_batched_(X, m, B)
        t0 = time.perf_counter()
        X[:] = 0
        it, rs, fl = models.ldiv_batched_(X, m, B)
        dt = time.perf_counter() - t0
        x1 = np.zeros(m.Ndim)
        it1, rs1, fl1 = models.ldiv_(x1, m, np.ascontiguousarray(B[0]))
        res[mode] = (X.copy(), it.copy(), x1.copy(), it1)
        print(f"   {mode:6s}: iters {it.tolist()} flags {fl.tolist()} res {rs.max():.2e} batched solve {1e3*dt:.2f} ms; single iters {it1} "
              f"single==batched[0]: {np.array_equal(x1, X[0])}", flush=True)
    a, b = res["wg"], res["stream"]
    print(f"   wg vs stream: |dx|/|x| = {np.linalg.norm(a[0]-b[0])/np.linalg.norm(b[0]):.2e}, iters equal: {np.array_equal(a[1], b[1])}")
    os.environ["ELPH_NO_WG"] = "0"
    if us.value:
        for nr in (1, 2, 8, 24, 48, 64, 256):
            _, Bs = configs.rhs(m, nr)
            ms = C.c_double()
            for reps in (200, 1000):
                check(lib.elph_bench_prepare(m._h, 1, nr, _lib.dptr(np.ascontiguousarray(Bs))))
                check(lib.elph_bench_run(m._h, 9, nr, reps, 0, C.byref(ms)))
            per = 1e3 * ms.value / 1000
            print(f"   wg nrhs={nr:3d}: {per:7.2f} us per iteration of the batch = {per/nr:6.3f} us per rhs-iteration, "
                  f"{2*nr/per:.3f} M mat-vecs/s", flush=True)
    m.close()
