"""(round 6) The preconditioned batch iteration of config C as 1 stream and as 2 … 8 parts on streams of their own (ELPH_SPLIT_WAYS, read per call:
switched inside one process, round robin).  A part must hold whole groups of chains, so the chain count is chosen to divide every part:
    python tools/time_split_ways.py [nrhs] [chains]        (default 288 right-hand sides of 36 chains: parts of 144, 72, 48, 36)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elphdynamics_amd import _lib, configs, models, preconditioners as pc, synth  # noqa: E402

lib = _lib.load()
nrhs = int(sys.argv[1]) if len(sys.argv) > 1 else 288
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 36
m = configs.make_model(os.environ.get("ELPH_TIME_TAG", "C"), tol=1e-5)
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
ways = [w for w in (2, 3, 4, 6, 8) if nrhs % w == 0 and (nrhs // w) % nch == 0]
res = {w: [] for w in [1] + ways}
for rnd in range(4):
    for w in [1] + ways:
        os.environ["ELPH_SPLIT_WAYS"] = str(max(w, 2))
        _lib.check(lib.elph_bench_prepare(m._h, 3, nrhs, None))
        run(3, 2)
        wh = 3 if w == 1 else 11
        run(wh, 32)
        res[w].append(run(wh, 320))
for w, v in res.items():
    print(f"nrhs {nrhs} chains {nch} parts {w}: iteration min {min(v):7.2f} med {sorted(v)[len(v) // 2]:7.2f} us", flush=True)
m.close()
