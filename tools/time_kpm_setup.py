"""Time setup!(P) for the chains resident in one handle (config C): the C-ABI call alone, start vectors prepared beforehand
(steady state of a run: the bounds move by less than buf, coefficients are not recomputed).
usage: python tools/time_kpm_setup.py [nchains]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import _lib, configs, models, synth    # noqa: E402
from elphdynamics_amd import preconditioners as pc           # noqa: E402

lib = _lib.load()
for nch in ([int(sys.argv[1])] if len(sys.argv) > 1 else [1, 64, 144]):
    m = configs.make_model("C")
    if nch > 1:
        Xc = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=synth.SEED_FIELDS + 17 * c) for c in range(nch)])
        models.update_model_chains_(m, Xc)
    P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
    rng = np.random.default_rng(7)
    bmax, bmin = rng.standard_normal((nch, m.Nsites)), rng.standard_normal((nch, m.Nsites))
    nan = np.full(nch, np.nan)
    act = (C.c_int * nch)()
    for mode in ("device", "host"):
        os.environ["ELPH_KPM_HOST"] = "1" if mode == "host" else "0"     # (read once per process: the second mode needs its own run)
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            _lib.check(lib.elph_kpm_setup_chains(m._h, _lib.dptr(bmax), _lib.dptr(bmin), _lib.dptr(nan), _lib.dptr(nan), act, None, None))
            ts.append(time.perf_counter() - t0)
        print(f"{nch:4d} chains: elph_kpm_setup_chains ({mode} flag) min {1e3 * min(ts[5:]):.3f} ms  median {1e3 * sorted(ts[5:])[12]:.3f} ms  "
              f"active {sum(act)}", flush=True)
        break
    m.close()
