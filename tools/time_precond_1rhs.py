#!/usr/bin/env python3
"""KPM apply (2) and preconditioned CG iteration (3) for ONE right-hand side, config C: us per unit (graph replay).  ELPH_FUSE_XR=0: the
residual update as a kernel of its own."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import _lib, configs, preconditioners as pc
from elphdynamics_amd._lib import check
lib = _lib.load()
m = configs.make_model("C", tol=1e-5)
P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
pc.setup_(P, rng=np.random.default_rng(7))
ms = C.c_double()
for what in (2, 3):
    check(lib.elph_bench_prepare(m._h, what, 1, None))
    check(lib.elph_bench_run(m._h, what, 1, 160, 0, C.byref(ms)))
    check(lib.elph_bench_run(m._h, what, 1, 1600, 1, C.byref(ms)))
    print(what, 1e3 * ms.value / 1600, "us")
