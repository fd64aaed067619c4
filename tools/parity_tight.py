#!/usr/bin/env python3
"""How close do the GPU solution and the oracle's get when both solve M^T M x = M^T R to a tight tolerance?
(north_star: Green's-function elements / M^-1 R within 1e-10 relative.)  usage: python3 tools/parity_tight.py [tags]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import configs, models          # noqa: E402
from oracle.oracle import Oracle                      # noqa: E402


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


orc = Oracle()
for tag in (sys.argv[1:] or ["b", "B", "C", "D", "E"]):
    m = configs.make_model(tag, tol=1e-5, maxiter=100000)
    if m.kind == 0:
        E = orc.update_model_holstein(m.Nsites, m.Ltau, m.dtau, m.x, m.lam, m.lam2, m.mu)
        om = orc.make_model(0, m.Nsites, m.Ltau, m.neighbor_table, m.cosht, m.sinht, E)
    else:
        om = orc.make_model(1, m.Nsites, m.Ltau, m.neighbor_table, np.ascontiguousarray(m.cosht).reshape(-1),
                            np.ascontiguousarray(m.sinht).reshape(-1), m.expDtauMu)
    R, B = configs.rhs(m, 1)
    b = np.ascontiguousarray(B[0])
    for tol in (1e-11, 1e-12, 1e-13, 1e-14):
        m.solver.tol = tol
        x = np.zeros(m.Ndim)
        it, res, fl = models.ldiv_(x, m, b)
        xo, ito, reso, flo = orc.ldiv(om, b, solver_tol=tol, solver_maxiter=100000)
        Mx = np.empty(m.Ndim)
        models.mulM_(Mx, m, x)
        print(f"{tag} tol={tol:.0e}: iters gpu={it} oracle={ito} flags {fl}/{flo} true-res gpu={res:.2e} oracle={reso:.2e} "
              f"|x-xo|/|xo|={rel(x, xo):.2e} |Mx-R|/|R|={rel(Mx, R[0]):.2e}", flush=True)
    m.close()
