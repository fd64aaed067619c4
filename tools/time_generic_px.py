"""(round 6) The preconditioned batch iteration on lattices of the GENERIC family (no lane program, no patch form) unfused (ELPH_GEN_PX=0) and p/x-fused,
one stream and two: square L = 22, 26 and a disordered 24 x 24.    python tools/time_generic_px.py [nrhs]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elphdynamics_amd import lattice as lat, models, preconditioners as pc, synth  # noqa: E402
from elphdynamics_amd._lib import check, dptr  # noqa: E402

nrhs = int(sys.argv[1]) if len(sys.argv) > 1 else 96
for Ls, Lt, dis in ((22, 160, 0.0), (26, 160, 0.0), (24, 160, 0.1)):
    la = lat.Lattice(1, Ls, Ls, 1)
    m = models.HolsteinModel(la, Lt * 0.1, 0.1, tol=1e-5, maxiter=20000)
    for (o1, o2, d) in lat.SQUARE_BONDS:
        m.assign_t_(1.0, o1, o2, d, stddev=dis, rng=np.random.default_rng(3))
    m.assign_omega_(1.0); m.assign_lambda_(1.0); m.assign_mu_(0.0)
    m.initialize_model_()
    m.x[:] = synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau)
    models.update_model_(m)
    P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
    pc.setup_(P, rng=np.random.default_rng(1))
    B = np.ascontiguousarray(np.stack([synth.randn(100 + r, m.Ndim) for r in range(nrhs)]))
    ms = C.c_double()
    lib = m._lib
    line = f"L = {Ls} (N = {m.Nsites}, Ltau = {Lt}, hopping disorder {dis}), {nrhs} right-hand sides:"
    for mode in ("0", "1"):
        os.environ["ELPH_GEN_PX"] = os.environ["ELPH_LDS_CHEB_PX"] = mode
        check(lib.elph_bench_prepare(m._h, 3, nrhs, dptr(B)))
        check(lib.elph_bench_run(m._h, 3, nrhs, 16, 0, C.byref(ms)))
        check(lib.elph_bench_prepare(m._h, 3, nrhs, None))
        check(lib.elph_bench_run(m._h, 3, nrhs, 96, 0, C.byref(ms)))
        line += f"  GEN_PX={mode}: {ms.value * 1e3 / 96:.1f} us"
        try:
            check(lib.elph_bench_prepare(m._h, 3, nrhs, None))
            check(lib.elph_bench_run(m._h, 11, nrhs, 16, 0, C.byref(ms)))
            check(lib.elph_bench_run(m._h, 11, nrhs, 96, 0, C.byref(ms)))
            line += f" (two streams {ms.value * 1e3 / 96:.1f})"
        except Exception:
            pass
    print(line, flush=True)
    m.close()
