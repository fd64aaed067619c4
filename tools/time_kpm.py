import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
from elphdynamics_amd import configs, models, preconditioners as pc
from elphdynamics_amd._lib import check, dptr
tag = sys.argv[1] if len(sys.argv) > 1 else "C"
m = configs.make_model(tag, tol=1e-5)
lib = m._lib
P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
pc.setup_(P, rng=np.random.default_rng(7))
print("orders sum", int(P.orders.sum()), "max", int(P.orders.max()), "Lo2", len(P.orders))
ms = C.c_double()
for nrhs in ([int(a) for a in sys.argv[2:]] or (1, 2, 10, 64, 128)):
    R, B = configs.rhs(m, nrhs)
    out = {}
    for what, name in ((1, "cg_iter"), (2, "kpm_apply"), (3, "pcg_iter")):
        check(lib.elph_bench_prepare(m._h, what, nrhs, dptr(np.ascontiguousarray(B))))
        check(lib.elph_bench_run(m._h, what, nrhs, 32, 0, C.byref(ms)))
        check(lib.elph_bench_prepare(m._h, what, nrhs, None))
        check(lib.elph_bench_run(m._h, what, nrhs, 320, 0, C.byref(ms)))
        out[name] = ms.value * 1e3 / 320
    X = np.zeros_like(B)
    models.ldiv_batched_(X, m, B, P=P); X[:] = 0
    t0 = time.perf_counter(); it, res, fl = models.ldiv_batched_(X, m, B, P=P); t1 = time.perf_counter()
    X[:] = 0
    models.ldiv_batched_(X, m, B); X[:] = 0
    t2 = time.perf_counter(); it0, res0, fl0 = models.ldiv_batched_(X, m, B); t3 = time.perf_counter()
    print(f"{tag} nrhs={nrhs:3d} cg_iter {out['cg_iter']:.1f} us  kpm_apply {out['kpm_apply']:.1f} us  pcg_iter {out['pcg_iter']:.1f} us | "
          f"solve kpm {1e3*(t1-t0):.2f} ms ({it.max()} it)  plain {1e3*(t3-t2):.2f} ms ({it0.max()} it)")
