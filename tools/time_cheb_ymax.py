import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from elphdynamics_amd import _lib, configs, models, preconditioners as pc, synth
from elphdynamics_amd._lib import check
lib = _lib.load()
m = configs.make_model("C", tol=1e-5)
nrhs, nch = int(sys.argv[1]), int(sys.argv[2])
R, B = configs.rhs(m, nrhs)
if nch > 1:
    Xc = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=100 + 17 * c) for c in range(nch)])
    models.update_model_chains_(m, Xc)
P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
(pc.setup_chains_ if nch > 1 else pc.setup_)(P, rng=np.random.default_rng(7))
ms = C.c_double()
check(lib.elph_bench_prepare(m._h, 3, nrhs, _lib.dptr(np.ascontiguousarray(B))))
check(lib.elph_bench_run(m._h, 3, nrhs, 2, 0, C.byref(ms)))
for wh in (7,):
    check(lib.elph_bench_run(m._h, wh, nrhs, 32, 0, C.byref(ms)))
    check(lib.elph_bench_run(m._h, wh, nrhs, 320, 0, C.byref(ms)))
    print(f"ymax={os.environ.get('ELPH_CHEB_DBG_YMAX')} nrhs={nrhs} chains={nch}: chebyshev {1e3*ms.value/320:.2f} us; orders {sorted(P.orders.tolist(), reverse=True)[:12] if nch==1 else ''}")
