"""(round 6) The p/x-fused k_cg_ap of config C in its two forms — lane program through LDS slabs (ELPH_SQ16_AP=0) and checkerboard in registers
(cg_sq16.hip; ELPH_SQ16_SHAPE=<ring depth><waves per SIMD> picks a measured alternative) — alone, and the whole preconditioned iteration on one
stream / two streams.  The settings are switched INSIDE one process, round robin, several rounds (both switches are read per launch): box-to-box
and run-to-run drift (3 % on this pool) does not enter the comparison.
    python tools/time_sq16.py [nrhs ...]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elphdynamics_amd import _lib, configs, models, preconditioners as pc, synth  # noqa: E402

lib = _lib.load()
SETTINGS = [("0", "0"), ("1", "0"), ("1", "24"), ("1", "33"), ("1", "42")] if os.environ.get("ELPH_TIME_TAG", "C") == "C" else [("0", "0"), ("1", "0"), ("1", "22"), ("1", "32")]      # (D: ring depth / waves of k_cg_ap_hc12_px; 0 = the default 2 / 3)
ROUNDS = int(os.environ.get("ELPH_TIME_ROUNDS", "4"))
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
    res = {s: {"ap": [], "iter": [], "iter2": []} for s in SETTINGS}
    for rnd in range(ROUNDS):
        for s in SETTINGS:
            os.environ["ELPH_SQ16_AP"], os.environ["ELPH_SQ16_SHAPE"] = s
            for name, wh in (("ap", 4), ("iter", 3), ("iter2", 11)):
                _lib.check(lib.elph_bench_prepare(m._h, 3, nrhs, None))
                run(3, 2)
                run(wh, 32)
                res[s][name].append(run(wh, 320))
    for s in SETTINGS:
        print(f"nrhs {nrhs} T {T.value} CHUNK_T={os.environ.get('ELPH_CHUNK_T')} SQ16_AP={s[0]} SHAPE={s[1]}: " +
              "  ".join(f"{k} min {min(v):7.2f} med {sorted(v)[len(v) // 2]:7.2f}" for k, v in res[s].items()), flush=True)
    m.close()
