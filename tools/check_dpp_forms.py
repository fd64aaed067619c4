#!/usr/bin/env python3
"""Bond phonons on the 16 x 16 lattice: the DPP form of the resident kernel against its lane-program form and the streaming iteration
(quick A/B next to the parity tests).  usage: python3 tools/check_ssh_dpp.py [E|D]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import configs, models
tag = sys.argv[1] if len(sys.argv) > 1 else "E"
m = configs.make_model(tag, tol=1e-12)
for nr in (1, 8, 24, 30):
    R, B = configs.rhs(m, nr)
    res = {}
    for name, env in (("dpp", {}), ("lane", {"ELPH_WG_NO_DPP": "1"}), ("stream", {"ELPH_NO_WG": "1"})):
        for k in ("ELPH_WG_NO_DPP", "ELPH_NO_WG"):
            os.environ.pop(k, None)
        os.environ.update(env)
        X = np.zeros_like(B)
        it, rs, fl = models.ldiv_batched_(X, m, B)
        res[name] = (X.copy(), np.asarray(it).max(), bool(np.asarray(fl).any()))
    sc = np.abs(res["stream"][0]).max()
    print(f"{tag} nrhs={nr}: dpp-lane {np.abs(res['dpp'][0]-res['lane'][0]).max()/sc:.2e}  dpp-stream {np.abs(res['dpp'][0]-res['stream'][0]).max()/sc:.2e}"
          f"  iters {res['dpp'][1]} {res['lane'][1]} {res['stream'][1]}  flags {res['dpp'][2]} {res['lane'][2]} {res['stream'][2]}", flush=True)
m.close()
