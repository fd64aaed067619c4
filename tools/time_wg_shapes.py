#!/usr/bin/env python3
"""Per-iteration time of the resident kernel for given batch sizes under the shape knobs in the environment (ELPH_WG_T, ELPH_WG_W).
usage: [ELPH_WG_T=2 ELPH_WG_W=4] python3 tools/time_wg_shapes.py C 24 48 288"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import _lib, configs
from elphdynamics_amd._lib import check
lib = _lib.load()
tag = sys.argv[1]
m = configs.make_model(tag, tol=1e-5)
for nr in [int(v) for v in sys.argv[2:]]:
    _, Bs = configs.rhs(m, nr)
    us, T, W, G = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    check(lib.elph_bench_wg_info(m._h, nr, C.byref(us), C.byref(T), C.byref(W), C.byref(G)))
    ms = C.c_double()
    for reps in (200, 1000):
        check(lib.elph_bench_prepare(m._h, 1, nr, _lib.dptr(np.ascontiguousarray(Bs))))
        check(lib.elph_bench_run(m._h, 9, nr, reps, 0, C.byref(ms)))
    print(f"{tag} T={os.environ.get('ELPH_WG_T','-')} Wcap={os.environ.get('ELPH_WG_W','-')} nrhs={nr:3d}: shape T={T.value} W={W.value} G={G.value}  {1e3*ms.value/1000:7.2f} us/iter ({2*nr/(1e3*ms.value/1000):.2f} M)", flush=True)
m.close()
