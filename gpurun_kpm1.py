import sys, ctypes as C
import numpy as np
sys.path.insert(0, '.')
from elphdynamics_amd import configs, models, preconditioners as pc
from elphdynamics_amd._lib import check
m = configs.make_model("C", tol=1e-5)
lib = m._lib
P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
pc.setup_(P, rng=np.random.default_rng(1))
nrhs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
R, B = configs.rhs(m, nrhs)
ms = C.c_double()
check(lib.elph_bench_prepare(m._h, 2, nrhs, np.ascontiguousarray(B).ctypes.data_as(C.POINTER(C.c_double))))
check(lib.elph_bench_run(m._h, 2, nrhs, 200, 0, C.byref(ms)))
print("kpm_apply us", 1e3*ms.value/200)
