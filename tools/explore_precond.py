#!/usr/bin/env python3
"""Round-4 exploration of the KPM-preconditioned batch iteration (config C): per-kernel times in flight and alone, and whether two
half batches on two streams (two handles, two host threads) overlap the latency-bound Chebyshev kernel of one half with the
HBM-bound kernels of the other.  usage: explore_precond.py [nrhs [chains]]"""
import ctypes as C, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import _lib, configs, models, preconditioners as pc, synth
from elphdynamics_amd._lib import check
lib = _lib.load()


def make(nrhs, nch, seed0=100):
    m = configs.make_model("C", tol=1e-5)
    R, B = configs.rhs(m, nrhs)
    if nch > 1:
        Xc = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=seed0 + 17 * c) for c in range(nch)])
        models.update_model_chains_(m, Xc)
    P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
    (pc.setup_chains_ if nch > 1 else pc.setup_)(P, rng=np.random.default_rng(7))
    check(lib.elph_bench_prepare(m._h, 3, nrhs, _lib.dptr(np.ascontiguousarray(B))))
    return m, P


GRAPH = 1 if os.environ.get("ELPH_USE_GRAPH") == "1" else 0


def run(m, what, nrhs, reps):
    ms = C.c_double()
    check(lib.elph_bench_run(m._h, what, nrhs, reps, GRAPH, C.byref(ms)))
    return 1e3 * ms.value / reps


nrhs = int(sys.argv[1]) if len(sys.argv) > 1 else 288
nch = int(sys.argv[2]) if len(sys.argv) > 2 else nrhs // 2
m, P = make(nrhs, nch)
run(m, 3, nrhs, 8)
names = {3: "iteration", 4: "k_cg_ap", 6: "forward+xr", 7: "chebyshev", 8: "inverse"}
out = {w: run(m, w, nrhs, 160) for w in (3, 4, 6, 7, 8)}
print(f"nrhs={nrhs} chains={nch}: " + "  ".join(f"{names[w]} {out[w]:.1f}" for w in out) + f"  | sum of four {out[4] + out[6] + out[7] + out[8]:.1f} us", flush=True)

if len(sys.argv) > 3 and sys.argv[3] == "split":
    # two half batches, two handles, two streams
    h = nrhs // 2
    ma, Pa = make(h, max(1, nch // 2), 100)
    mb, Pb = make(h, max(1, nch // 2), 5000)
    for mm in (ma, mb):
        run(mm, 3, h, 8)
    alone = run(ma, 3, h, 160)
    res = [0.0, 0.0]

    def worker(i, mm):
        res[i] = run(mm, 3, h, 320)
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(i, mm)) for i, mm in enumerate((ma, mb))]
    for t in th: t.start()
    for t in th: t.join()
    wall = (time.perf_counter() - t0) * 1e6 / 320
    print(f"split {h}+{h}: one half alone {alone:.1f} us/iteration; both at once: stream times {res[0]:.1f} / {res[1]:.1f}, wall {wall:.1f} us per iteration of the whole batch "
          f"(single handle: {out[3]:.1f})", flush=True)
