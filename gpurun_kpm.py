import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, '.')
from elphdynamics_amd import configs, models, preconditioners as pc
from elphdynamics_amd._lib import check
tag = sys.argv[1] if len(sys.argv) > 1 else "C"
m = configs.make_model(tag, tol=1e-5)
lib = m._lib
P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
t0=time.time(); pc.setup_(P, rng=np.random.default_rng(1)); t1=time.time()
print("setup", t1-t0, "active", P.active, P.lam_lo, P.lam_hi, "orders sum", P.orders.sum(), "max", P.orders.max())
t0=time.time(); pc.setup_(P, rng=np.random.default_rng(2)); t1=time.time(); print("setup again", t1-t0)
R, B = configs.rhs(m, 1)
b = np.ascontiguousarray(B[0])
for tol in (1e-5, 1e-10):
    m.solver.tol = tol
    x = np.zeros(m.Ndim); t0=time.time(); it0, r0, f0 = models.ldiv_(x, m, b); t1=time.time()
    x = np.zeros(m.Ndim); t2=time.time(); it1, r1, f1 = models.ldiv_(x, m, b, P=P); t3=time.time()
    x = np.zeros(m.Ndim); t2=time.time(); it1, r1, f1 = models.ldiv_(x, m, b, P=P); t3=time.time()
    print(f"tol={tol:g}: plain CG {it0} iters {1e3*(t1-t0):.2f} ms | KPM-CG {it1} iters {1e3*(t3-t2):.2f} ms  flags {f0} {f1}")
ms = C.c_double()
for nrhs in (1, 16, 64):
    R, B = configs.rhs(m, nrhs)
    for what, name in ((2, "kpm_apply"), (3, "prec_cg_iter"), (1, "cg_iter")):
        check(lib.elph_bench_prepare(m._h, what, nrhs, np.ascontiguousarray(B).ctypes.data_as(C.POINTER(C.c_double))))
        check(lib.elph_bench_run(m._h, what, nrhs, 32, 0, C.byref(ms)))
        check(lib.elph_bench_run(m._h, what, nrhs, 160, 0, C.byref(ms)))
        print(f"nrhs={nrhs} {name}: {1e3*ms.value/160:.2f} us")
