"""(round 6) Hopping disorder on square lattices in the patch layout: the preconditioned batch iteration with the generic LDS kernels (ELPH_PG_DIS=0)
and with the table variants of the patch kernels (=1), next to the same lattice with uniform hopping.    python tools/time_disorder_patches.py [nrhs] [L ...]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elphdynamics_amd import lattice as lat, models, preconditioners as pc, synth  # noqa: E402
from elphdynamics_amd._lib import check, dptr  # noqa: E402

nrhs = int(sys.argv[1]) if len(sys.argv) > 1 else 96
sizes = [int(a) for a in sys.argv[2:]] or [24, 32, 26, 20]


def iteration_us(m, nrhs, B):
    ms = C.c_double()
    lib = m._lib
    out = []
    for what in (3, 11):
        try:
            check(lib.elph_bench_prepare(m._h, 3, nrhs, dptr(B)))
            check(lib.elph_bench_run(m._h, what, nrhs, 16, 0, C.byref(ms)))
            if what == 3:
                check(lib.elph_bench_prepare(m._h, 3, nrhs, None))
            check(lib.elph_bench_run(m._h, what, nrhs, 96, 0, C.byref(ms)))
            out.append(ms.value * 1e3 / 96)
        except Exception:
            out.append(float("nan"))
    return out


for Ls in sizes:
    line = f"L = {Ls}, Ltau = 160, {nrhs} right-hand sides, us per preconditioned iteration (one stream / two streams):"
    for dis, mode in ((0.0, "1"), (0.1, "0"), (0.1, "1")):
        os.environ["ELPH_PG_DIS"] = mode
        la = lat.Lattice(1, Ls, Ls, 1)
        m = models.HolsteinModel(la, 16.0, 0.1, tol=1e-5, maxiter=20000)
        for (o1, o2, d) in lat.SQUARE_BONDS:
            m.assign_t_(1.0, o1, o2, d, stddev=dis, rng=np.random.default_rng(3))
        m.assign_omega_(1.0); m.assign_lambda_(1.0); m.assign_mu_(0.0)
        m.initialize_model_()
        m.x[:] = synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau)
        models.update_model_(m)
        P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
        pc.setup_(P, rng=np.random.default_rng(1))
        B = np.ascontiguousarray(np.stack([synth.randn(100 + r, m.Ndim) for r in range(nrhs)]))
        one, two = iteration_us(m, nrhs, B)
        name = "uniform" if dis == 0.0 else ("disordered, generic kernels" if mode == "0" else "disordered, patch kernels with tables")
        line += f"  {name}: {one:.1f} / {two:.1f}"
        m.close()
    print(line, flush=True)
