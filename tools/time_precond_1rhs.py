#!/usr/bin/env python3
"""KPM apply (2) and preconditioned CG iteration (3) for one, two, ... right-hand sides (the lone chain's shape): us per unit, plain launches and
graph replay (profiles/r06/lone_chain_xr_folded_into_one_tile_transform_rejected.log was made with it).
usage: python3 tools/time_precond_1rhs.py [config] [nrhs ...]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import _lib, configs, preconditioners as pc
from elphdynamics_amd._lib import check
lib = _lib.load()
tag = sys.argv[1] if len(sys.argv) > 1 else "C"
m = configs.make_model(tag, tol=1e-5)
P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
pc.setup_(P, rng=np.random.default_rng(7))
ms = C.c_double()
for nrhs in [int(a) for a in sys.argv[2:]] or [1, 2, 4]:
    R, B = configs.rhs(m, nrhs)
    out = []
    for what in (2, 3):
        check(lib.elph_bench_prepare(m._h, what, nrhs, _lib.dptr(np.ascontiguousarray(B))))
        check(lib.elph_bench_run(m._h, what, nrhs, 160, 0, C.byref(ms)))
        check(lib.elph_bench_prepare(m._h, what, nrhs, None))
        check(lib.elph_bench_run(m._h, what, nrhs, 320, 0, C.byref(ms)))
        plain = 1e3 * ms.value / 320
        check(lib.elph_bench_prepare(m._h, what, nrhs, None))
        check(lib.elph_bench_run(m._h, what, nrhs, 1600, 1, C.byref(ms)))
        out.append((plain, 1e3 * ms.value / 1600))
    print(f"{tag} nrhs={nrhs}: kpm_apply {out[0][0]:.2f} us (graph {out[0][1]:.2f})   pcg_iter {out[1][0]:.2f} us (graph {out[1][1]:.2f})")
m.close()
