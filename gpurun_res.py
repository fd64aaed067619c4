"""Resident CG kernel: correctness vs the two-kernel path and per-iteration timing (scratch driver for gpurun)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from elphdynamics_amd import configs, models

def solve(m, B, tol, resident, T=0):
    os.environ["ELPH_NO_RESIDENT"] = "0" if resident else "1"
    if T: os.environ["ELPH_RESIDENT_T"] = str(T)
    else: os.environ.pop("ELPH_RESIDENT_T", None)
    m.solver.tol = tol
    X = np.zeros_like(B)
    t0 = time.perf_counter()
    it, res, fl = models.ldiv_batched_(X, m, B)
    dt = time.perf_counter() - t0
    return X, np.asarray(it), np.asarray(fl), dt

for tag in ("b", "C", "D"):
    m = configs.make_model(tag, tol=1e-5, maxiter=20000)
    for nrhs in (1, 2, 8, 16):
        R, B = configs.rhs(m, nrhs)
        X0, it0, fl0, _ = solve(m, B, 1e-8, False)
        for T in (1, 2, 4):
            if m.Ltau % T or ((m.Nsites + 63) // 64) * T > 8: continue
            X1, it1, fl1, _ = solve(m, B, 1e-8, True, T)
            X1, it1, fl1, dt1 = solve(m, B, 1e-8, True, T)
            _, _, _, dt0 = solve(m, B, 1e-8, False)
            d = np.abs(X1 - X0).max() / np.abs(X0).max()
            print(f"{tag} nrhs={nrhs} T={T}: iters {it1.tolist()[:4]} vs {it0.tolist()[:4]} maxdiff {d:.2e} flags {fl1.tolist()[:4]} "
                  f"resident {1e6*dt1/max(it1.max(),1):.2f} us/iter two-kernel {1e6*dt0/max(it0.max(),1):.2f} us/iter", flush=True)
    m.close()
