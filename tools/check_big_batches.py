"""(round 5) Batches far beyond the bench sizes: 560-660 right-hand sides of 280-330 chains on configs C, D, E — KPM-preconditioned and plain batched solves agree with each other and with single solves.  usage: python3 tools/check_big_batches.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from elphdynamics_amd import configs, models, preconditioners as pc, synth
for tag, nch in (("C", 330), ("D", 300), ("E", 280)):
    m = configs.make_model(tag, tol=1e-6)
    nrhs = 2 * nch
    if m.kind == 0:
        X = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=100 + c) for c in range(nch)])
    else:
        X = np.stack([m.x * (0.6 + 0.8 * c / nch) * (1.0 + 0.2 * synth.randn(100 + c, m.Ndof)) for c in range(nch)])
    models.update_model_chains_(m, X)
    B = np.stack([synth.randn(7000 + r, m.Ndim) for r in range(nrhs)])
    P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
    rng = np.random.default_rng(3)
    bmax, bmin = rng.standard_normal((nch, m.Nsites)), rng.standard_normal((nch, m.Nsites))
    pc.setup_chains_(P, b_max=bmax, b_min=bmin)
    t0 = time.perf_counter()
    Xp = np.zeros_like(B); itp, resp, flp = models.ldiv_batched_(Xp, m, B, P=P)
    t1 = time.perf_counter()
    Xs = np.zeros_like(B); its, ress, fls = models.ldiv_batched_(Xs, m, B)
    t2 = time.perf_counter()
    assert not flp.any() and not fls.any()
    d = np.abs(Xp - Xs).max() / np.abs(Xs).max()
    worst = 0.0
    for r in (0, 1, nch - 1, nch, nrhs - 1, nrhs // 2 + 7):
        c = r % nch
        m1 = configs.make_model(tag, tol=1e-6)
        m1.x[:] = X[c]; models.update_model_(m1)
        x1 = np.zeros(m.Ndim); it1 = models.ldiv_(x1, m1, np.ascontiguousarray(B[r]))[0]
        worst = max(worst, np.abs(x1 - Xs[r]).max() / np.abs(x1).max())
        assert abs(it1 - its[r]) <= 2, (tag, r, it1, its[r])
        m1.close()
    print(f"{tag}: {nrhs} right-hand sides of {nch} chains: preconditioned {1e3*(t1-t0):.0f} ms ({int(itp.max())} it), plain {1e3*(t2-t1):.0f} ms ({int(its.max())} it); "
          f"max |x_prec - x_plain|/|x| {d:.1e}; against single solves {worst:.1e}", flush=True)
    m.close()
