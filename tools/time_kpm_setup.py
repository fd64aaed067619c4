"""setup!(P) (KPMPreconditioners.jl:259-321) per call: one chain (host Arnoldi by default) and many chains (device Arnoldi), wall clock.
    python tools/time_kpm_setup.py          (set ELPH_KPM_HOST=1 / ELPH_KPM_DEVICE=1 to pin the path)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elphdynamics_amd import configs, models, preconditioners as pc, synth  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "C"
for nch in (1, 2, 4, 16, 64, 144):
    m = configs.make_model(tag, tol=1e-5)
    if nch > 1:
        X = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=100 + c) for c in range(nch)])
        models.update_model_chains_(m, X)
    P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
    rng = np.random.default_rng(7)
    f = pc.setup_chains_ if nch > 1 else pc.setup_
    f(P, rng=rng)
    t0 = time.perf_counter()
    reps = 30
    for _ in range(reps):
        f(P, rng=rng)
    dt = (time.perf_counter() - t0) / reps
    print(f"{tag} chains {nch:4d}: setup! {1e6 * dt:8.1f} us per call   HOST={os.environ.get('ELPH_KPM_HOST')} DEVICE={os.environ.get('ELPH_KPM_DEVICE')}", flush=True)
    m.close()
