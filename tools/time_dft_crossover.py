"""KPM apply time vs batch size with the scalar-twiddle transforms (ELPH_DFT_MFMA=0) and the MFMA ones (=1): where to switch."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from elphdynamics_amd import configs, preconditioners as pc
from elphdynamics_amd._lib import check, dptr
tag = sys.argv[1] if len(sys.argv) > 1 else "C"
m = configs.make_model(tag, tol=1e-5)
lib = m._lib
P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
pc.setup_(P, rng=np.random.default_rng(7))
ms = C.c_double()
for nrhs in (2, 4, 6, 8, 12, 16, 24, 32, 64):
    R, B = configs.rhs(m, nrhs)
    row = []
    for mode in ("0", "1"):
        os.environ["ELPH_DFT_MFMA"] = mode
        for what in (2, 3):
            check(lib.elph_bench_prepare(m._h, what, nrhs, dptr(np.ascontiguousarray(B))))
            check(lib.elph_bench_run(m._h, what, nrhs, 32, 0, C.byref(ms)))
            check(lib.elph_bench_prepare(m._h, what, nrhs, None))
            check(lib.elph_bench_run(m._h, what, nrhs, 320, 0, C.byref(ms)))
            row.append(ms.value * 1e3 / 320)
    print(f"{tag} nrhs={nrhs:3d}  scalar: apply {row[0]:6.1f} pcg {row[1]:6.1f} us   mfma: apply {row[2]:6.1f} pcg {row[3]:6.1f} us")
