#!/usr/bin/env python3
"""The row form of the resident kernel (k_cg_row, ELPH_WG_ROW=1) against the 2 x 2 patch form (k_cg_wg) on config C: iteration counts, solutions,
time per iteration.  usage: python3 tools/check_row_form.py [nrhs ...]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import _lib, configs, models
from elphdynamics_amd._lib import check
lib = _lib.load()
for nr in [int(a) for a in sys.argv[1:]] or [25, 48, 288]:
    m = configs.make_model("C", tol=1e-9)
    _, B = configs.rhs(m, nr)
    B = np.ascontiguousarray(B)
    res = {}
    for mode in ("0", "1"):
        os.environ["ELPH_WG_ROW"] = mode
        X = np.zeros_like(B)
        it, r, fl = models.ldiv_batched_(X, m, B)
        ms = C.c_double()
        us = float("nan")
        try:
            for reps in (100, 1000):
                check(lib.elph_bench_prepare(m._h, 1, nr, _lib.dptr(B)))
                check(lib.elph_bench_run(m._h, 9, nr, reps, 0, C.byref(ms)))
            us = 1e3 * ms.value / 1000
        except Exception as e:
            print("   bench:", str(e)[:120])
        res[mode] = (it.copy(), X.copy(), fl.copy(), us)
    a, b = res["0"], res["1"]
    d = np.abs(a[1] - b[1]).max() / np.abs(a[1]).max()
    print(f"nrhs {nr:3d}: iterations {int(a[0].min())}..{int(a[0].max())} / row form {int(b[0].min())}..{int(b[0].max())} (max |diff| {int(np.abs(a[0]-b[0]).max())}); flags {int(a[2].max())}/{int(b[2].max())}; "
          f"max |dx|/|x| {d:.2e}; us per iteration {a[3]:.2f} / row form {b[3]:.2f}", flush=True)
    m.close()
