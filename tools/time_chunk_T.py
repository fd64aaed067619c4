import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
from elphdynamics_amd import configs, models, synth
from elphdynamics_amd._lib import check, dptr
T = os.environ.get("ELPH_CHUNK_T", "auto")
m = configs.make_model("C", tol=1e-5)
lib = m._lib
ms = C.c_double()
for nrhs, nch in ((64, 32), (128, 64), (256, 128)):
    Xc = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=100 + 17 * c) for c in range(nch)])
    models.update_model_chains_(m, Xc)
    R, B = configs.rhs(m, nrhs)
    out = {}
    for what, name in ((1, "cg_iter"), (4, "ap"), (5, "xr")):
        check(lib.elph_bench_prepare(m._h, 1, nrhs, dptr(np.ascontiguousarray(B))))
        check(lib.elph_bench_run(m._h, what, nrhs, 160, 0, C.byref(ms)))
        check(lib.elph_bench_run(m._h, what, nrhs, 1600, 0, C.byref(ms)))
        out[name] = ms.value * 1e3 / 1600
    print(f"T={T} nrhs={nrhs:3d} chains={nch}: iter {out['cg_iter']:.2f} us  ap {out['ap']:.2f}  xr {out['xr']:.2f}  -> {2*nrhs/out['cg_iter']:.3f} M matvec/s")
