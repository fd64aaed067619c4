#!/usr/bin/env python3
"""Where an iteration of the workgroup-resident CG spends its time (diagnostic build: tools/build_wg_stamps.sh first).
usage: ELPH_LIB=elphdynamics_amd/libelphgpu_stamps.so python3 tools/time_wg_phases.py [tags]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import _lib, configs          # noqa: E402
from elphdynamics_amd._lib import check            # noqa: E402

lib = _lib.load()
NAMES = ["mat-vec + four sums + boundary stores of z", "the meeting (records, boundary slices of z)", "barrier", "x, r updates, halo of r", "barrier (r in LDS)", "-", "-",
         "stop test", "p update", "(loop top)"]
for tag in (sys.argv[1:] or ["b", "C"]):
    m = configs.make_model(tag, tol=1e-5)
    for nr in [int(v) for v in os.environ.get("ELPH_NRS", "1,24,48").split(",")]:
        _, Bs = configs.rhs(m, nr)
        ms = C.c_double()
        reps = 1000
        check(lib.elph_bench_prepare(m._h, 1, nr, _lib.dptr(np.ascontiguousarray(Bs))))
        check(lib.elph_bench_run(m._h, 9, nr, reps, 0, C.byref(ms)))
        out = (C.c_ulonglong * 16)()
        assert lib.elph_debug_wg_stamps(out) == 0
        tot = sum(out[k] for k in range(10))
        print(f"== {tag} nrhs={nr}: {1e3*ms.value/reps:.2f} us per iteration (events); stamped {tot/100/reps:.2f} us; iterations {out[10]}")
        for k in range(10):
            print(f"     {NAMES[k]:28s} {out[k]/100/reps:7.3f} us")
    m.close()
