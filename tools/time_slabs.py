#!/usr/bin/env python3
"""The slab form of the resident un-preconditioned solve on lattices beyond one wave's slice (slabs.hip) against the streaming iteration:
per-iteration time for 1 and 2 right-hand sides, and a whole solve through elph_ldiv in both forms (same iterations, solutions compared).
usage: python3 tools/time_slabs.py [Lspace ...]      (Holstein, L x L square lattices, Ltau = ELPH_TIME_LTAU or 160)"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import models, synth                          # noqa: E402
from elphdynamics_amd._lib import check, dptr                      # noqa: E402

from elphdynamics_amd import configs                                # noqa: E402

from elphdynamics_amd import lattice as lat                         # noqa: E402

for Ls in [int(a) for a in sys.argv[1:]] or [18, 20, 24, 28, 30, 32]:
    configs.CONFIGS["_slab"] = ("holstein", 1, Ls, lat.SQUARE_BONDS, 0.1 * int(os.environ.get("ELPH_TIME_LTAU", "160")), 0.1)
    m = configs.make_model("_slab", tol=1e-5, maxiter=20000)
    Lt = m.Ltau
    lib = m._lib
    ms = C.c_double()
    line = f"L = {Ls:2d} (N = {m.Nsites:4d}, Ltau = {Lt}):"
    for nrhs in (1, 2):
        _, B = configs.rhs(m, nrhs)
        out = {}
        for what in (1, 12):
            try:
                for reps in (64, 400):
                    check(lib.elph_bench_prepare(m._h, 1, nrhs, dptr(np.ascontiguousarray(B))))
                    check(lib.elph_bench_run(m._h, what, nrhs, reps, 0, C.byref(ms)))
                out[what] = 1e3 * ms.value / 400
            except Exception as e:
                out[what] = float("nan")
                print("   ", str(e)[:150])
        line += f"   {nrhs} rhs: streaming {out[1]:6.2f} us/iter, slabs {out[12]:6.2f}"
    # whole solves through ldiv!: slab form (default where it applies) against the streaming form
    b = np.ascontiguousarray(configs.rhs(m, 1)[1][0])
    res = {}
    for mode in ("1", "0"):
        os.environ["ELPH_SLABS"] = mode
        x = np.zeros(m.Ndim)
        models.ldiv_(x, m, b)                      # warm
        x[:] = 0.0
        t0 = time.perf_counter()
        it, resid, flag = models.ldiv_(x, m, b)
        res[mode] = (1e3 * (time.perf_counter() - t0), it, resid, flag, x.copy())
    os.environ.pop("ELPH_SLABS")
    d = np.abs(res["1"][4] - res["0"][4]).max() / np.abs(res["0"][4]).max()
    line += (f"   ldiv!: slabs {res['1'][0]:6.2f} ms ({res['1'][1]} it, flag {res['1'][3]}), streaming {res['0'][0]:6.2f} ms ({res['0'][1]} it); "
             f"max |dx|/|x| {d:.1e}")
    print(line, flush=True)
    m.close()
