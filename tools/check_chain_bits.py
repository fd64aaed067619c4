#!/usr/bin/env python3
"""Bond-phonon chains (config E): a right-hand side solved in a batch over several resident chains against the same solve alone on a
fresh single-configuration handle — bits and iteration counts, repeated; on a mismatch the state of both handles is printed.
usage: python3 tools/check_chain_bits.py [rounds]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import _lib, configs, models, synth
lib = _lib.load()

def status(m):
    cd, fb = C.c_int(), C.c_int64()
    _lib.check(lib.elph_wg_status(m._h, C.byref(cd), C.byref(fb)))
    us, T, W, G = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    _lib.check(lib.elph_bench_wg_info(m._h, 1, C.byref(us), C.byref(T), C.byref(W), C.byref(G)))
    return dict(cooldown=cd.value, fallbacks=fb.value, usable=us.value, T=T.value, W=W.value, G=G.value)

tag, nchains, per = "E", 4, 2
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
# some history in the process first (other lattices, other handles), as a test run has
for t in ("b", "C", "e"):
    mm = configs.make_model(t, tol=1e-5)
    _, Bq = configs.rhs(mm, 3)
    Xq = np.zeros_like(Bq); models.ldiv_batched_(Xq, mm, Bq); mm.close()
m = configs.make_model(tag, tol=1e-5)
X = np.stack([m.x * (0.55 + 0.9 * c / nchains) * (1.0 + 0.2 * synth.randn(5000 + c, m.Ndof)) for c in range(nchains)])
nrhs = nchains * per
B = np.stack([synth.randn(7000 + r, m.Ndim) for r in range(nrhs)])
models.update_model_chains_(m, X)
Xs = np.zeros_like(B)
it, res, fl = models.ldiv_batched_(Xs, m, B)
print("batch", it.tolist(), status(m), flush=True)
bad = 0
for k in range(rounds):
    for r in range(nrhs):
        m1 = configs.make_model(tag, tol=1e-5)
        m1.x[:] = X[r % nchains]
        models.update_model_(m1)
        x1 = np.zeros(m.Ndim)
        it1, res1, fl1 = models.ldiv_(x1, m1, np.ascontiguousarray(B[r]))
        if it1 != it[r] or not np.array_equal(x1, Xs[r]):
            bad += 1
            x2 = np.zeros(m.Ndim); it2, _, _ = models.ldiv_(x2, m1, np.ascontiguousarray(B[r]))
            os.environ["ELPH_NO_WG"] = "1"
            x3 = np.zeros(m.Ndim); it3, _, _ = models.ldiv_(x3, m1, np.ascontiguousarray(B[r]))
            del os.environ["ELPH_NO_WG"]
            print(f"round {k} rhs {r}: single {it1} batch {int(it[r])} | same handle again {it2} (bits as first {np.array_equal(x1, x2)}, as batch {np.array_equal(x2, Xs[r])})"
                  f" | streaming {it3} (bits as first {np.array_equal(x1, x3)}) | rel diff {np.abs(x1 - Xs[r]).max() / np.abs(x1).max():.2e} | {status(m1)}", flush=True)
        m1.close()
print("mismatches", bad, "of", rounds * nrhs)
m.close()
