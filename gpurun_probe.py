import sys, time
import numpy as np
sys.path.insert(0, '.')
from elphdynamics_amd import configs, models
from oracle.oracle import Oracle
orc = Oracle()
for tag in ("b", "B", "C", "D"):
    m = configs.make_model(tag, tol=1e-5)
    E = orc.update_model_holstein(m.Nsites, m.Ltau, m.dtau, m.x, m.lam, m.lam2, m.mu)
    om = orc.make_model(0, m.Nsites, m.Ltau, m.neighbor_table, m.cosht, m.sinht, E)
    R, B = configs.rhs(m, 1)
    b = np.ascontiguousarray(B[0])
    bo = orc.mulMT(om, np.ascontiguousarray(R[0]))
    print(tag, "ncol", m.colours.max() if m.Nbonds else 0, "MT err", np.linalg.norm(b-bo)/np.linalg.norm(bo))
    for tol in (1e-5, 1e-10):
        x = np.zeros(m.Ndim)
        t0 = time.time(); it, hist = models.solve_(x, m, b, tol=tol, history=True); t1 = time.time()
        xo, ito, histo = orc.cg_solve(om, b, tol=tol, maxiter=10000, history=True); t2 = time.time()
        n = min(len(hist), len(histo))
        rel = np.abs(hist[:n]-histo[:n])/histo[:n]
        first_bad = int(np.argmax(rel > 1e-10)) if (rel > 1e-10).any() else -1
        print(f"  tol={tol:g} iters gpu={it} oracle={ito} gpu_time={t1-t0:.4f}s oracle_time={t2-t1:.3f}s  |x-xo|/|xo|={np.linalg.norm(x-xo)/np.linalg.norm(xo):.2e}"
              f" hist rel diff: j<=10 {rel[:11].max():.1e}, j<=50 {rel[:51].max():.1e}, all {rel.max():.1e}, first j with >1e-10: {first_bad}")
