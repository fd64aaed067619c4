"""The three kernels of the KPM apply (forward transform, Chebyshev recursion, inverse transform) one at a time, the apply and the preconditioned iteration,
for a Holstein square lattice of any size and time axis.    python tools/time_kpm_parts.py Lspace Ltau nrhs [nrhs ...]"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from elphdynamics_amd import lattice as lat, models, preconditioners as pc, synth
from elphdynamics_amd._lib import check, dptr
Ls, Lt = int(sys.argv[1]), int(sys.argv[2])
la = lat.Lattice(1, Ls, Ls, 1)
m = models.HolsteinModel(la, Lt * 0.1, 0.1, tol=1e-5, maxiter=20000)
for (o1, o2, d) in lat.SQUARE_BONDS: m.assign_t_(1.0, o1, o2, d)
m.assign_omega_(1.0); m.assign_lambda_(1.0); m.assign_mu_(0.0)
m.initialize_model_()
m.x[:] = synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau)
models.update_model_(m)
P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
pc.setup_(P, rng=np.random.default_rng(1))
ms = C.c_double(); lib = m._lib
for nrhs in [int(a) for a in sys.argv[3:]]:
    B = np.stack([synth.randn(100 + r, m.Ndim) for r in range(nrhs)])
    check(lib.elph_bench_prepare(m._h, 3, nrhs, dptr(np.ascontiguousarray(B))))
    out = []
    for what in (6, 7, 8, 2, 3):
        check(lib.elph_bench_run(m._h, what, nrhs, 8, 0, C.byref(ms)))
        check(lib.elph_bench_run(m._h, what, nrhs, 32, 0, C.byref(ms)))
        out.append(ms.value * 1e3 / 32)
    print(f"L={Ls} Ltau={Lt} nrhs={nrhs}: fwd {out[0]:.1f}  cheb {out[1]:.1f}  inv {out[2]:.1f}  apply {out[3]:.1f}  pcg_iter {out[4]:.1f} us")
m.close()
