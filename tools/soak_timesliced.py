#!/usr/bin/env python3
"""Resident-kernel solves while other processes share the GPU (time-slicing: a team's members can be held up for milliseconds): every
solve must complete in the resident kernel (no time-out, no fallback) and agree with the first one bit for bit.
usage: python3 tools/soak_timesliced.py C 24 20     (config, right-hand sides, solves) — start several at once"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import _lib, configs, models
lib = _lib.load()
tag, nr, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
m = configs.make_model(tag, tol=1e-9)
_, B = configs.rhs(m, nr)
ref = None
t0 = time.time()
for k in range(n):
    X = np.zeros_like(B)
    it, res, fl = models.ldiv_batched_(X, m, B)
    assert not fl.any()
    if ref is None:
        ref = (X.copy(), it.copy())
    same = np.array_equal(X, ref[0]) and np.array_equal(it, ref[1])
    cd, fb = C.c_int(), C.c_int64()
    _lib.check(lib.elph_wg_status(m._h, C.byref(cd), C.byref(fb)))
    if not same or fb.value:
        print(f"{tag} pid {os.getpid()} solve {k}: same bits {same}, fallbacks {fb.value}", flush=True)
print(f"{tag} nrhs={nr} pid {os.getpid()}: {n} solves in {time.time()-t0:.1f} s, fallbacks {fb.value}, iterations {int(it.max())}", flush=True)
m.close()
