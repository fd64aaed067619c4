"""Time of one measurement's Green's-function work at a BASELINE config: update! (n_v solves) and setup! over all pairs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from elphdynamics_amd import configs, greens, preconditioners as pc, synth
tag = sys.argv[1] if len(sys.argv) > 1 else "C"
nv = 10
m = configs.make_model(tag, tol=1e-5)
P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
est = greens.EstimateGreensFunction(m, nv=nv)
R = np.stack([synth.randn(900 + i, m.Ndim) for i in range(nv)])
for rep in range(2):
    t0 = time.perf_counter(); it, res, fl = greens.update_(est, m, P=P, R=R, rng=np.random.default_rng(1)); t1 = time.perf_counter()
    npairs = 0
    for i in range(1, nv):
        for j in range(i + 1, nv + 1):
            greens.setup_(est, i, j); npairs += 1
    t2 = time.perf_counter()
    lib, h = m._lib, m._h
    for i in range(1, nv):
        for j in range(i + 1, nv + 1):
            lib.elph_greens_setup(h, i, j, None, None, None, None)
    t3 = time.perf_counter()
print(f"{tag}: update! ({nv} solves, KPM, {int(it.max())} its) {1e3*(t1-t0):.2f} ms; setup! x {npairs} pairs: {1e3*(t2-t1):.1f} ms "
      f"({1e3*(t2-t1)/npairs:.3f} ms each, 4 arrays to the host); device only {1e3*(t3-t2):.1f} ms ({1e3*(t3-t2)/npairs:.3f} ms each)")
