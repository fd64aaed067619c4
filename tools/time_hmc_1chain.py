import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from elphdynamics_amd import configs, hmc, preconditioners as pc, synth
m = configs.make_model("C", tol=1e-5, maxiter=20000)
fa = pc.FourierAccelerator(m); pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.3)
P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
H = hmc.HybridMonteCarlo(m, fa, 0.05, 0.5, alpha=0.0, Nb=1)
H.device_rng_(1234)
for _ in range(2): hmc.update_(m, H, fa, P, pull=False)
t0 = time.perf_counter(); n = 5
for _ in range(n): hmc.update_(m, H, fa, P, pull=False)
print("ms per update", 1e3 * (time.perf_counter() - t0) / n, "Nt", H.Nt, "iters", H.iters, flush=True)
