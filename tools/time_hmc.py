import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
from elphdynamics_amd import configs, models, hmc, preconditioners as pc, synth
nt, dt = 20, 0.01
TAG = __import__('os').environ.get("TAG", "C")       # TAG=E: the optical SSH square lattice (bond phonons)
for nch in ([int(a) for a in sys.argv[1:]] or [1, 8, 32]):
    for with_kpm in (True,):
        for nb in (1,):
            m = configs.make_model(TAG, tol=1e-5, maxiter=20000)
            fa = pc.FourierAccelerator(m)
            pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.1)
            H = hmc.HybridMonteCarlo(m, fa, dt=dt, tr=nt * dt, alpha=0.0, Nb=nb, nchains=nch)
            if nch > 1:
                if m.kind == models.SSH:
                    H.X[:] = np.stack([m.x * (0.7 + 0.5 * c / nch) for c in range(nch)])
                else:
                    H.X[:] = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=100 + 17 * c) for c in range(nch)])
                H.push_()
            P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0) if with_kpm else None
            rng = np.random.default_rng(3)
            if not __import__('os').environ.get("HOST_RNG"):
                H.device_rng_(3)
            upd = (lambda: hmc.update_chains_(m, H, fa, P, rng=rng)) if nch > 1 else (lambda: hmc.update_(m, H, fa, P, rng=rng))
            upd()
            t0 = time.perf_counter(); acc, its = upd(); t1 = time.perf_counter()
            ms = 1e3 * (t1 - t0)
            print(f"chains={nch:2d} kpm={with_kpm} Nb={nb:2d} Nt={nt}: update {ms:8.1f} ms -> {ms/nch:7.2f} ms per chain-update, "
                  f"{ms/(nt+2):.2f} ms per evaluation, iters/solve {np.mean(its):.1f}, accepted {np.mean(acc):.2f}")
            m.close()
