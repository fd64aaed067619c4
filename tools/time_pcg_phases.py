#!/usr/bin/env python3
"""Where an iteration of the resident preconditioned CG spends its time (diagnostic build: tools/build_pcg_stamps.sh first).
usage: ELPH_LIB=elphdynamics_amd/libelphgpu_pcgstamps.so python3 tools/time_pcg_phases.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import _lib, configs, preconditioners as pc          # noqa: E402
from elphdynamics_amd._lib import check, dptr                              # noqa: E402

os.environ["ELPH_PCG_WG"] = "1"
lib = _lib.load()
CG = ["mat-vec + four sums", "barrier", "meeting (records)", "updates + stop test", "r to memory + drain + barrier", "wait for P^-1 r (all helper stages)",
      "p = P^-1 r + beta p (loads)", "-", "-", "-", "-", "(loop top)"]
HP = ["wait for the residual (flag B)", "forward transform tile + drain + barrier", "flag C: wait for all tiles", "Chebyshev recursions (longest first)",
      "drain + barrier", "record D: wait for all frequencies", "inverse transform tile + drain + barrier", "-", "-", "-", "-", "(loop top)"]
m = configs.make_model("C", tol=1e-5)
P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
pc.setup_(P, rng=np.random.default_rng(7))
for nr in (1, 2):
    _, Bs = configs.rhs(m, nr)
    ms = C.c_double()
    reps = 320
    check(lib.elph_bench_prepare(m._h, 10, nr, dptr(np.ascontiguousarray(Bs))))
    check(lib.elph_bench_run(m._h, 10, nr, reps, 0, C.byref(ms)))
    out = (C.c_ulonglong * 32)()
    assert lib.elph_debug_pcg_stamps(out) == 0
    print(f"== C nrhs={nr}: {1e3*ms.value/reps:.2f} us per preconditioned iteration (events)")
    print("   CG workgroup 0, wave 0:")
    for k in range(12):
        if CG[k] != "-":
            print(f"     {CG[k]:48s} {out[k]/100/reps:7.3f} us")
    print("   helper workgroup 0, wave 0 (longest recursion):")
    for k in range(12):
        if HP[k] != "-":
            print(f"     {HP[k]:48s} {out[12+k]/100/reps:7.3f} us")
m.close()
