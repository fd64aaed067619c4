"""Un-preconditioned CG iteration, streaming (two kernels) against workgroup-resident form, per config and batch size:
the data the form-selection rule of elph_wg_cg (cg_wg.hip) is fitted to.  usage: python3 tools/time_forms.py B D E C"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from elphdynamics_amd import _lib, configs
from elphdynamics_amd._lib import check
lib = _lib.load()
for tag in sys.argv[1:]:
    m = configs.make_model(tag, tol=1e-5)
    for nr in (8, 16, 24, 64, 256):
        _, Bs = configs.rhs(m, nr)
        ms = C.c_double()
        out = {}
        for what in (1, 9):
            for reps in (160, 640):
                check(lib.elph_bench_prepare(m._h, 1, nr, _lib.dptr(np.ascontiguousarray(Bs))))
                check(lib.elph_bench_run(m._h, what, nr, reps, 1 if what == 1 else 0, C.byref(ms)))
            out[what] = 1e3 * ms.value / 640
        print(f"{tag} nrhs={nr:3d}: streaming {out[1]:7.2f} us/iter ({2*nr/out[1]:.2f} M)   resident {out[9]:7.2f} us/iter ({2*nr/out[9]:.2f} M)", flush=True)
    m.close()
