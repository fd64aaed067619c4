"""Time-to-solution of the batched un-preconditioned solve: two-kernel iteration vs the resident kernel (ELPH_RESIDENT_WAVES=1)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from elphdynamics_amd import configs, models, synth
m = configs.make_model(sys.argv[1] if len(sys.argv) > 1 else "C", tol=1e-5)
for nrhs, nch in ((1, 1), (2, 1), (16, 8), (64, 32), (128, 64)):
    if nch > 1:
        Xc = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=100 + 17 * c) for c in range(nch)])
        models.update_model_chains_(m, Xc)
    else:
        models.update_model_(m)
    R, B = configs.rhs(m, nrhs)
    res = {}
    for mode, env in (("two-kernel", {}), ("resident T=1", {"ELPH_RESIDENT_WAVES": "1", "ELPH_RESIDENT_T": "1"}),
                      ("resident T=2", {"ELPH_RESIDENT_WAVES": "1", "ELPH_RESIDENT_T": "2"})):
        for k in ("ELPH_RESIDENT_WAVES", "ELPH_RESIDENT_T"):
            os.environ.pop(k, None)
        os.environ.update(env)
        X = np.zeros_like(B)
        try:
            models.ldiv_batched_(X, m, B); X[:] = 0
            t0 = time.perf_counter(); it, r, fl = models.ldiv_batched_(X, m, B); t1 = time.perf_counter()
            res[mode] = (1e3 * (t1 - t0), int(it.max()), X.copy())
        except Exception as e:
            res[mode] = (float("nan"), -1, None); print(mode, "failed:", e)
    base = res["two-kernel"]
    line = f"nrhs={nrhs:3d} chains={nch:2d}: "
    for k, (ms, it, X) in res.items():
        same = "" if X is None or k == "two-kernel" else (" same" if np.array_equal(X, base[2]) else f" rel {np.linalg.norm(X-base[2])/np.linalg.norm(base[2]):.1e}")
        line += f"{k} {ms:.2f} ms ({it} it, {1e3*ms/max(it,1):.1f} us/it){same} | "
    print(line)
