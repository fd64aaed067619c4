import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
from elphdynamics_amd import configs, models, preconditioners as pc, synth
m = configs.make_model("C", tol=1e-5)
nch = 32
Xc = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=100 + 17 * c) for c in range(nch)])
t0 = time.perf_counter(); models.update_model_chains_(m, Xc); t1 = time.perf_counter()
print("update_model_chains %.2f ms" % (1e3 * (t1 - t0)))
P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
rng = np.random.default_rng(1)
bmax, bmin = rng.standard_normal((nch, m.Nsites)), rng.standard_normal((nch, m.Nsites))
for i in range(4):
    t0 = time.perf_counter(); act, lo, hi = pc.setup_chains_(P, b_max=bmax, b_min=bmin); t1 = time.perf_counter()
    print("setup_chains call %d: %.3f ms  (active %d)" % (i, 1e3 * (t1 - t0), act.sum()))
# perturb fields slightly (as in HMC): bounds move < 5% -> no coefficient recompute
models.update_model_chains_(m, Xc * 1.001)
t0 = time.perf_counter(); act, lo, hi = pc.setup_chains_(P, b_max=bmax, b_min=bmin); t1 = time.perf_counter()
print("setup_chains after small move: %.3f ms" % (1e3 * (t1 - t0)))
models.update_model_(m)
P1 = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
for i in range(3):
    t0 = time.perf_counter(); pc.setup_(P1, b_max=bmax[0], b_min=bmin[0]); t1 = time.perf_counter()
    print("single setup call %d: %.3f ms" % (i, 1e3 * (t1 - t0)))
