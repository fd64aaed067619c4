"""Time setup!(P) for the chains resident in one handle (config C) against the number of host threads.
usage: python tools/time_kpm_setup.py [nchains]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import configs, models, synth          # noqa: E402
from elphdynamics_amd import preconditioners as pc           # noqa: E402

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = configs.make_model("C")
Xc = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=synth.SEED_FIELDS + 17 * c) for c in range(nch)])
models.update_model_chains_(m, Xc)
P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
for thr in ("1", "2", "4", "8", "16", "32"):
    os.environ["ELPH_KPM_THREADS"] = thr
    pc.setup_chains_(P, rng=np.random.default_rng(7))
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        pc.setup_chains_(P, rng=np.random.default_rng(7))
        ts.append(time.perf_counter() - t0)
    print(f"threads {thr:>2}: setup of {nch} chains  min {1e3 * min(ts):.3f} ms  median {1e3 * sorted(ts)[10]:.3f} ms", flush=True)
