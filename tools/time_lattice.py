"""cg_iter / kpm_apply / pcg_iter times for a Holstein square lattice of any size:  time_lattice.py Lspace Ltau [nrhs ...]
(ELPH_TIME_HONEYCOMB=1: a honeycomb lattice of Lspace x Lspace two-site cells; ELPH_TIME_TRIANGULAR=1: a triangular lattice)"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from elphdynamics_amd import lattice as lat, models, preconditioners as pc, synth
from elphdynamics_amd._lib import check, dptr
Ls, Lt = int(sys.argv[1]), int(sys.argv[2])
HC = os.environ.get("ELPH_TIME_HONEYCOMB") == "1"
TRI = os.environ.get("ELPH_TIME_TRIANGULAR") == "1"
la = lat.Lattice(2 if HC else 1, Ls, Ls, 1)
m = models.HolsteinModel(la, Lt * 0.1, 0.1, tol=1e-5, maxiter=20000)
for (o1, o2, d) in (lat.HONEYCOMB_BONDS if HC else (lat.TRIANGULAR_BONDS if TRI else lat.SQUARE_BONDS)):
    m.assign_t_(1.0, o1, o2, d)
m.assign_omega_(1.0); m.assign_lambda_(1.0); m.assign_mu_(0.0)
m.initialize_model_()
m.x[:] = synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau)
models.update_model_(m)
P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
pc.setup_(P, rng=np.random.default_rng(1))
print("N", m.Nsites, "Ltau", Lt, "orders sum", int(P.orders.sum()), "max", int(P.orders.max()))
ms = C.c_double()
lib = m._lib
for nrhs in [int(a) for a in sys.argv[3:]] or [1, 16]:
    B = np.stack([synth.randn(100 + r, m.Ndim) for r in range(nrhs)])
    out = {}
    for what, name in ((1, "cg_iter"), (2, "kpm_apply"), (3, "pcg_iter")):
        check(lib.elph_bench_prepare(m._h, what, nrhs, dptr(np.ascontiguousarray(B))))
        check(lib.elph_bench_run(m._h, what, nrhs, 32, 0, C.byref(ms)))
        check(lib.elph_bench_prepare(m._h, what, nrhs, None))
        check(lib.elph_bench_run(m._h, what, nrhs, 160, 0, C.byref(ms)))
        out[name] = ms.value * 1e3 / 160
    two = ""
    try:      # the preconditioned iteration as two half-batches on two streams (what a solve runs from 64 / 192 right-hand sides, where the fused form exists)
        check(lib.elph_bench_prepare(m._h, 3, nrhs, None))
        check(lib.elph_bench_run(m._h, 11, nrhs, 32, 0, C.byref(ms)))
        check(lib.elph_bench_run(m._h, 11, nrhs, 160, 0, C.byref(ms)))
        two = f"  pcg_iter on two streams {ms.value * 1e3 / 160:.1f} us"
    except Exception:
        pass
    print(f"nrhs={nrhs:3d} cg_iter {out['cg_iter']:.1f} us  kpm_apply {out['kpm_apply']:.1f} us  pcg_iter {out['pcg_iter']:.1f} us{two}")
m.close()
