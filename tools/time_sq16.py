"""(round 6) The p/x-fused k_cg_ap of config C in its two forms — lane program through LDS slabs (ELPH_SQ16_AP=0) and checkerboard in registers
(cg_sq16.hip; ELPH_SQ16_SHAPE=<ring depth><waves per SIMD> picks a measured alternative) — alone, and the whole preconditioned iteration on one
stream / two streams.    python tools/time_sq16.py [nrhs ...]   (the environment is read when the process starts: run once per setting)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elphdynamics_amd import _lib, configs, models, preconditioners as pc, synth  # noqa: E402

lib = _lib.load()
for nrhs in [int(a) for a in sys.argv[1:]] or [288]:
    m = configs.make_model(os.environ.get("ELPH_TIME_TAG", "C"), tol=1e-5)
    nch = nrhs // 2
    X = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=100 + c) for c in range(nch)])
    models.update_model_chains_(m, X)
    P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
    pc.setup_chains_(P, rng=np.random.default_rng(7))
    _, B = configs.rhs(m, nrhs)

    def run(what, reps):
        ms = C.c_double()
        _lib.check(lib.elph_bench_run(m._h, what, nrhs, reps, 0, C.byref(ms)))
        return 1e3 * ms.value / reps

    _lib.check(lib.elph_bench_prepare(m._h, 3, nrhs, _lib.dptr(np.ascontiguousarray(B))))
    T = C.c_int()
    _lib.check(lib.elph_bench_info(m._h, nrhs, C.byref(T)))
    out = {"nrhs": nrhs, "T": T.value}
    for name, wh in (("ap", 4), ("fwd", 6), ("cheb", 7), ("inv", 8), ("iter", 3), ("iter2", 11)):
        _lib.check(lib.elph_bench_prepare(m._h, 3, nrhs, None))
        run(3, 2)
        try:
            run(wh, 32)
            out[name] = round(min(run(wh, 320) for _ in range(3)), 2)
        except Exception as e:
            out[name] = repr(e)[:60]
    f = C.c_int()
    _lib.check(lib.elph_bench_px_info(m._h, C.byref(f)))
    out["form"] = {0: "unfused", 1: "lane program", 2: "registers"}[f.value]
    print("SQ16_AP=%s SHAPE=%s CHUNK_T=%s" % (os.environ.get("ELPH_SQ16_AP"), os.environ.get("ELPH_SQ16_SHAPE"), os.environ.get("ELPH_CHUNK_T")), out, flush=True)
    m.close()
