"""CG iteration time vs batch size around the wave-count sweet spots (config C)."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from elphdynamics_amd import configs, models, synth
from elphdynamics_amd._lib import check, dptr
m = configs.make_model("C", tol=1e-5)
lib = m._lib
ms = C.c_double()
T = C.c_int()
for nrhs in [int(a) for a in sys.argv[1:]] or [96, 100, 102, 104, 112, 128, 192, 204, 208]:
    nch = nrhs // 2
    Xc = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=100 + 17 * c) for c in range(nch)])
    models.update_model_chains_(m, Xc)
    R, B = configs.rhs(m, nrhs)
    check(lib.elph_bench_prepare(m._h, 1, nrhs, dptr(np.ascontiguousarray(B))))
    check(lib.elph_bench_run(m._h, 1, nrhs, 160, 0, C.byref(ms)))
    check(lib.elph_bench_run(m._h, 1, nrhs, 1600, 0, C.byref(ms)))
    us = ms.value * 1e3 / 1600
    check(lib.elph_bench_info(m._h, nrhs, C.byref(T)))
    print(f"nrhs={nrhs:4d} T={T.value:2d} waves={nrhs*m.Ltau//max(T.value,1):5d}: {us:7.2f} us/iter  {us/nrhs*1e3:6.1f} ns per rhs-iter  {2*nrhs/us:.3f} M matvec/s")
