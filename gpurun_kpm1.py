import sys, time
import numpy as np
sys.path.insert(0, '.')
from elphdynamics_amd import configs, models, hmc, preconditioners as pc, synth
m = configs.make_model("C", tol=1e-5, maxiter=20000)
fa = pc.FourierAccelerator(m)
pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.1)
for with_kpm in (False, True):
    for nb in (1, 10):
        H = hmc.HybridMonteCarlo(m, fa, dt=0.01, tr=0.2, alpha=0.0, Nb=nb)
        P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0) if with_kpm else None
        rng = np.random.default_rng(3)
        hmc.update_(m, H, fa, P, rng=rng)
        t0 = time.perf_counter()
        acc, its = hmc.update_(m, H, fa, P, rng=rng)
        t1 = time.perf_counter()
        print(f"kpm={with_kpm} Nb={nb} Nt={H.Nt}: update {1e3*(t1-t0):.1f} ms = {1e3*(t1-t0)/(H.Nt+2):.2f} ms per force/action evaluation; iters/solve {its:.1f} accepted {acc} dH {H.H1-H.H0:.3e}")
