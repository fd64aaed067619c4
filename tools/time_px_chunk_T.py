"""The p/x-fused k_cg_ap_chunk<PX> alone and the whole preconditioned iteration (one stream / two streams) for forced chunk lengths.
    python tools/time_px_chunk_T.py [nrhs]        (run once per ELPH_CHUNK_T value: the chunk length is read when the handle is made)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elphdynamics_amd import _lib, configs, models, preconditioners as pc, synth  # noqa: E402

nrhs = int(sys.argv[1]) if len(sys.argv) > 1 else 288
lib = _lib.load()
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
        out[name] = round(run(wh, 320), 2)
    except Exception as e:
        out[name] = repr(e)[:60]
print(os.environ.get("ELPH_CHUNK_T"), out, flush=True)
