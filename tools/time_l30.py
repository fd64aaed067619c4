"""Square 30 x 30 (the Sq<2,10> patch, round 5) against the generic kernels (ELPH_NO_PG=1): CG iteration, KPM apply, preconditioned iteration.
    python tools/time_l30.py ; ELPH_NO_PG=1 python tools/time_l30.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from elphdynamics_amd import _lib, lattice as lat, models, preconditioners as pc, synth
lib = _lib.load()
la = lat.Lattice(1, 30, 30, 1)
m = models.HolsteinModel(la, 16.0, 0.1, tol=1e-5, maxiter=20000)
for (o1, o2, d) in lat.SQUARE_BONDS:
    m.assign_t_(1.0, o1, o2, d)
m.assign_omega_(1.0), m.assign_lambda_(1.0), m.assign_mu_(0.0)
m.initialize_model_()
m.x[:] = synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau)
models.update_model_(m)
P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
pc.setup_(P, rng=np.random.default_rng(3))
for nrhs in (1, 72):
    B = np.stack([synth.randn(300 + r, m.Ndim) for r in range(nrhs)])
    for name, prep, wh in (("cg_iter", 1, 1), ("kpm_apply", 3, 2), ("pcg_iter", 3, 3)):
        _lib.check(lib.elph_bench_prepare(m._h, prep, nrhs, _lib.dptr(np.ascontiguousarray(B))))
        ms = C.c_double()
        _lib.check(lib.elph_bench_run(m._h, wh, nrhs, 10, 0, C.byref(ms)))
        _lib.check(lib.elph_bench_run(m._h, wh, nrhs, 100, 0, C.byref(ms)))
        print(f"L=30 Ltau=160 nrhs {nrhs:3d} {name:10s} {1e3 * ms.value / 100:8.1f} us   NO_PG={os.environ.get('ELPH_NO_PG')}", flush=True)
m.close()
