"""One HMC update (KPM-preconditioned force solves, Fourier acceleration) on the 4 x 4 lattice with 1280 time slices: the long-axis
transforms of dft_big.hip under the whole update.  usage: python3 tools/check_long_axis_hmc.py"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import configs, hmc, preconditioners as pc
m = configs.make_model("l", tol=1e-5, maxiter=20000)
fa = pc.FourierAccelerator(m)
pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.1)
H = hmc.HybridMonteCarlo(m, fa, dt=0.01, tr=0.03, alpha=0.0, Nb=1, nchains=1)
P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
H.device_rng_(3)
t0 = time.perf_counter(); acc, its = hmc.update_(m, H, fa, P, rng=np.random.default_rng(3)); t1 = time.perf_counter()
print("long-axis HMC update ok: Ltau", m.Ltau, "accepted", acc, "iters", its, f"{1e3*(t1-t0):.1f} ms")
m.close()
