"""Fixed cost of ONE launch of the workgroup-resident CG kernel: event time of K iterations of the whole batch for several K, and the
intercept / slope of the line through them.    time_wg_intercept.py [config] [nrhs ...]
(the driver times `bench.py --steps 20`: one launch of 20 iterations, where the intercept is a fifth of the time)"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from elphdynamics_amd import configs
from elphdynamics_amd._lib import check, dptr
tag = sys.argv[1] if len(sys.argv) > 1 else "C"
m = configs.make_model(tag, tol=1e-5)
lib = m._lib
ms = C.c_double()
for nrhs in [int(a) for a in sys.argv[2:]] or [48, 288]:
    R, B = configs.rhs(m, nrhs)
    Bc = np.ascontiguousarray(B)
    check(lib.elph_bench_prepare(m._h, 1, nrhs, dptr(Bc)))
    check(lib.elph_bench_run(m._h, 9, nrhs, 50, 0, C.byref(ms)))
    pts = []
    for K in (1, 2, 5, 10, 20, 40, 100, 400):
        best = 1e30
        for rep in range(5):
            check(lib.elph_bench_prepare(m._h, 1, nrhs, None))
            check(lib.elph_bench_run(m._h, 9, nrhs, K, 0, C.byref(ms)))
            best = min(best, ms.value * 1e3)
        pts.append((K, best))
    (k1, t1), (k2, t2) = pts[-2], pts[-1]
    slope = (t2 - t1) / (k2 - k1)
    print(f"{tag} nrhs={nrhs}: us per iteration {slope:.2f};  " + "  ".join(f"K={k}: {t:.1f} us (fixed {t - slope * k:.1f})" for k, t in pts))
m.close()
