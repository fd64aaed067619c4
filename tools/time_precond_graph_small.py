import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from elphdynamics_amd import _lib, configs, models, preconditioners as pc
lib = _lib.load()
for tag in ("C", "D", "E"):
    m = configs.make_model(tag, tol=1e-5)
    P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
    pc.setup_(P, rng=np.random.default_rng(7))
    for nrhs in (1, 2):
        _, B = configs.rhs(m, nrhs)
        for g in (0, 1):
            _lib.check(lib.elph_bench_prepare(m._h, 3, nrhs, _lib.dptr(np.ascontiguousarray(B))))
            ms = C.c_double()
            _lib.check(lib.elph_bench_run(m._h, 3, nrhs, 32, g, C.byref(ms)))
            _lib.check(lib.elph_bench_run(m._h, 3, nrhs, 320, g, C.byref(ms)))
            print(tag, "nrhs", nrhs, "graph" if g else "eager", round(1e3 * ms.value / 320, 2), "us per preconditioned iteration", flush=True)
    m.close()
