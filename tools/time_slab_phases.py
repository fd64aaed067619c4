#!/usr/bin/env python3
"""Where an iteration of the slab form (slabs.hip: k_cg_wg<..., SHARD, RANKS>) spends its time — wave 0 of workgroup 0 of slab 0
(diagnostic build: tools/build_wg_stamps.sh first).  usage: ELPH_LIB=elphdynamics_amd/libelphgpu_stamps.so python3 tools/time_slab_phases.py [L ...]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import _lib, configs, lattice as lat
from elphdynamics_amd._lib import check
lib = _lib.load()
NAMES = ["mat-vec + four sums + boundary stores of z", "the meeting (workgroups of the slab, then the slabs; ghost rows of z)", "barrier", "x, r updates, halo of r", "barrier (r in LDS)", "-", "-",
         "stop test", "p update", "(loop top)"]
for Ls in [int(a) for a in sys.argv[1:]] or [24, 32]:
    configs.CONFIGS["_slab"] = ("holstein", 1, Ls, lat.SQUARE_BONDS, 16.0, 0.1)
    m = configs.make_model("_slab", tol=1e-5)
    _, B = configs.rhs(m, 1)
    ms = C.c_double()
    reps = 400
    for r in (64, reps):
        check(lib.elph_bench_prepare(m._h, 1, 1, _lib.dptr(np.ascontiguousarray(B))))
        check(lib.elph_bench_run(m._h, 12, 1, r, 0, C.byref(ms)))
    out = (C.c_ulonglong * 16)()
    assert lib.elph_debug_wg_stamps(out) == 0
    tot = sum(out[k] for k in range(10))
    print(f"== square {Ls} x {Ls}: {1e3*ms.value/reps:.2f} us per iteration (events); stamped {tot/100/reps:.2f} us; iterations {out[10]}")
    for k in range(10):
        print(f"     {NAMES[k]:72s} {out[k]/100/reps:7.3f} us")
    m.close()
