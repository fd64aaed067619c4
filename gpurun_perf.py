import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, '.')
from elphdynamics_amd import configs, models
from elphdynamics_amd._lib import check
tag = sys.argv[1] if len(sys.argv) > 1 else "C"
m = configs.make_model(tag, tol=1e-5)
lib = m._lib
for nrhs in (1, 2, 4, 16, 64):
    R, B = configs.rhs(m, nrhs)
    X = np.zeros_like(B)
    t0 = time.time(); it, res, fl = models.ldiv_batched_(X, m, B); t1 = time.time()
    X[:] = 0
    t0 = time.time(); it, res, fl = models.ldiv_batched_(X, m, B); t1 = time.time()
    ms = C.c_double()
    out = {}
    for what, name, g in ((0, "MtM", 0), (1, "cg_iter", 0), (1, "cg_iter_graph", 1), (4, "ap", 0), (5, "xr", 0)):
        check(lib.elph_bench_prepare(m._h, what, nrhs, None))
        check(lib.elph_bench_run(m._h, what, nrhs, 160, g, C.byref(ms)))
        check(lib.elph_bench_run(m._h, what, nrhs, 800, g, C.byref(ms)))
        out[name] = ms.value * 1e3 / 800
    ndim = m.Ndim
    print(f"{tag} nrhs={nrhs:3d} ldiv wall {1e3*(t1-t0):8.2f} ms iters={it.max()} ({1e6*(t1-t0)/it.max():.1f} us/iter)  "
          f"MtM {out['MtM']:.2f} us  ap {out['ap']:.2f} xr {out['xr']:.2f}  cg_iter eager {out['cg_iter']:.2f} graph {out['cg_iter_graph']:.2f} us  -> {2*nrhs/out['cg_iter_graph']:.3f} M matvec/s, "
          f"alg BW {120*ndim*nrhs/out['cg_iter_graph']/1e6:.3f} TB/s")
