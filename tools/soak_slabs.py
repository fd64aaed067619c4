#!/usr/bin/env python3
"""The slab form (slabs.hip) while other processes share the GPU: every solve must be, bit for bit, EITHER the slab form's solution or the
streaming iteration's (a launch that timed out is re-solved by it; 16 solves of cool-down follow) — never a third thing.  Start several at once.
usage: python3 tools/soak_slabs.py X32 400"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import _lib, configs, models
lib = _lib.load()
tag, n = sys.argv[1], int(sys.argv[2])
m = configs.make_model(tag, tol=1e-9)
_, B = configs.rhs(m, 1)
b = np.ascontiguousarray(B[0])
def solve():
    x = np.zeros(m.Ndim)
    it, res, fl = models.ldiv_(x, m, b)
    assert fl == 0
    return x, it
os.environ["ELPH_SLABS"] = "0"
x_str, it_str = solve()
os.environ.pop("ELPH_SLABS")
kinds = {"slab": 0, "streaming": 0, "OTHER": 0}
x_slab = None
t0 = time.time()
for k in range(n):
    x, it = solve()
    if np.array_equal(x, x_str):
        kinds["streaming"] += 1
    elif x_slab is None or np.array_equal(x, x_slab):
        if x_slab is None:
            x_slab = x.copy()
        kinds["slab"] += 1
    else:
        kinds["OTHER"] += 1
        print(f"{tag} pid {os.getpid()} solve {k}: NEITHER form's bits; iterations {it} (streaming {it_str}); max rel diff to slab {np.abs(x - x_slab).max() / np.abs(x_slab).max():.2e}", flush=True)
cd, fb = C.c_int(), C.c_int64()
_lib.check(lib.elph_wg_status(m._h, C.byref(cd), C.byref(fb)))
print(f"{tag} pid {os.getpid()}: {n} solves in {time.time()-t0:.1f} s: {kinds}, time-outs {fb.value}", flush=True)
m.close()
