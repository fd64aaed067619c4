"""Device memory comes back: models that used every stateful part of the library (solver workspaces, KPM expansion, HMC state with chains, the
Green's-function estimator, a sharded handle with its mailbox, bond-phonon tables made on the device) are created and destroyed in a loop and
the device's free memory stays where it was after the first cycles (the allocator's pools).  Found with this kind of loop in round 5: slab handles
of large lattices that were never freed; d_ssh_bar."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_bytes(hip):
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value


def _cycle(tag):
    from elphdynamics_amd import configs, greens, hmc, models, preconditioners as pc, synth
    m = configs.make_model(tag, tol=1e-6, maxiter=20000)
    fa = pc.FourierAccelerator(m)
    pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.3)
    P = pc.SymmetricKPMPreconditioner(m, n=min(20, m.Nsites), buf=0.05, c1=1.0, c2=1.0)
    H = hmc.HybridMonteCarlo(m, fa, 0.05, 0.1)
    H.device_rng_(11)
    hmc.update_(m, H, fa, P, pull=False)
    est = greens.EstimateGreensFunction(m, nv=2)
    greens.update_(est, m, R=np.stack([synth.randn(1 + i, m.Ndim) for i in range(est.nv)]))
    greens.setup_(est, 1, 2)
    _, B = configs.rhs(m, 3)
    X = np.zeros_like(B)
    models.ldiv_batched_(X, m, np.ascontiguousarray(B), P)
    m.close()
    # chains in lockstep
    m2 = configs.make_model(tag, tol=1e-6, maxiter=20000)
    fa2 = pc.FourierAccelerator(m2)
    pc.update_M_(fa2, m2, 0.0, np.inf, 1.0, 0.3)
    Hc = hmc.HybridMonteCarlo(m2, fa2, 0.05, 0.1, nchains=3)
    Hc.X[:] = m2.x
    Hc.push_()
    Hc.device_rng_(12)
    hmc.update_chains_(m2, Hc, fa2, None)
    m2.close()


@pytest.mark.parametrize("tag", ["b", "e"])      # Holstein and bond phonons (4 x 4, short time axis)
def test_create_use_destroy_returns_device_memory(tag):
    hip = C.CDLL("libamdhip64.so")
    for _ in range(3):
        _cycle(tag)
    f0 = _free_bytes(hip)
    for _ in range(12):
        _cycle(tag)
    # (the allocator returns memory in 2 MiB granules and may keep a few: a real leak — one forgotten workspace per model — is tens of MB over 12 cycles)
    assert abs(_free_bytes(hip) - f0) < (12 << 20), (f0, _free_bytes(hip))


def test_sharded_handles_return_device_memory():
    from elphdynamics_amd import configs, dist
    from elphdynamics_amd.sharded import ShardedSolver
    hip = C.CDLL("libamdhip64.so")
    m = configs.make_model("B", tol=1e-6)
    la = m.lattice

    def cycle():
        comm = dist.Comm()
        s = ShardedSolver(comm, la.norbits, la.L1, la.L2, m.Ltau, m.neighbor_table, kind=0, cosht=m.cosht, sinht=m.sinht, device=0, selftest=True)
        s.update_model(np.exp(-0.01 * np.arange(m.Ndim) / m.Ndim))
        x, it, done = s.solve(np.ones(m.Ndim), tol=1e-6)
        assert done == 1
        s.close()

    cycle()
    f0 = _free_bytes(hip)
    for _ in range(12):
        cycle()
    assert abs(_free_bytes(hip) - f0) < (12 << 20)
    m.close()
