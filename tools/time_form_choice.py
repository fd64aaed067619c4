"""Whole batched solves with the form chosen by the library (auto), with the resident kernel forced (ELPH_WG_ALWAYS=1) and with the
streaming kernels forced (ELPH_NO_WG=1).  usage: python3 tools/time_form_choice.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from elphdynamics_amd import configs, models
for tag, nr in (("E", 64), ("D", 256), ("D", 16), ("C", 64), ("B", 256)):
    m = configs.make_model(tag, tol=1e-5)
    R, B = configs.rhs(m, nr)
    for mode in ("auto", "always", "never"):
        os.environ["ELPH_WG_ALWAYS"] = "1" if mode == "always" else "0"
        os.environ["ELPH_NO_WG"] = "1" if mode == "never" else "0"
        X = np.zeros_like(B); models.ldiv_batched_(X, m, B); X[:] = 0
        t0 = time.perf_counter(); it, rs, fl = models.ldiv_batched_(X, m, B); dt = time.perf_counter() - t0
        print(f"{tag} nrhs={nr:3d} {mode:6s}: {1e3*dt:7.2f} ms  iters max {it.max()} flags {int(fl.any())}", flush=True)
    m.close()
