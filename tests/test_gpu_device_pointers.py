"""The device-pointer side of the C ABI (include/elph_gpu.h: the *_dev twins, elph_set_stream, elph_greens_dev_arrays / _nv) driven with caller-owned
device memory and a caller-owned stream — what a Julia host with GPU arrays (or a torch caller) binds.  Buffers through the HIP runtime the library
links (ctypes on libamdhip64), reference layout."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_muldmdx import _DevBuf
from test_gpu_parity import rel

pytestmark = pytest.mark.gpu


def test_solve_and_preconditioner_twins_and_a_callers_stream():
    from elphdynamics_amd import _lib, configs, models, preconditioners as pc
    from elphdynamics_amd._lib import check
    lib = _lib.load()
    m = configs.make_model("B", tol=1e-8)
    _, B = configs.rhs(m, 3)
    B = np.ascontiguousarray(B)
    # host entry points: the reference
    x = np.zeros(m.Ndim)
    it, res, fl = models.ldiv_(x, m, B[0])
    X = np.zeros_like(B)
    itB, resB, flB = models.ldiv_batched_(X, m, B)
    P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
    pc.setup_(P, rng=np.random.default_rng(5))
    z = np.zeros(m.Ndim)
    pc.kpm_ldiv_(z, P, B[1])
    xp = np.zeros(m.Ndim)
    itp, resp, flp = models.ldiv_(xp, m, B[0], P)
    # device twins on caller-owned buffers
    db, dx = _DevBuf(B[0]), _DevBuf(np.zeros(m.Ndim))
    i1, r1, f1 = C.c_int64(), C.c_double(), C.c_int()
    check(lib.elph_ldiv_dev(m._h, dx.p, db.p, 0, 0, C.byref(i1), C.byref(r1), C.byref(f1)))
    assert (i1.value, f1.value) == (it, fl) and r1.value == res and np.array_equal(dx.get(), x)
    dB, dX = _DevBuf(B), _DevBuf(np.zeros(B.size))
    its = np.zeros(3, dtype=np.int64); rs = np.zeros(3); fs = np.zeros(3, dtype=np.int32)
    check(lib.elph_ldiv_batched_dev(m._h, 3, dX.p, dB.p, 0, 0, its.ctypes.data_as(_lib.P_i64), _lib.dptr(rs), fs.ctypes.data_as(_lib.P_int)))
    assert np.array_equal(its, itB) and np.array_equal(rs, resB) and np.array_equal(dX.get().reshape(3, -1), X)
    dr, dz = _DevBuf(B[1]), _DevBuf(n=m.Ndim)
    check(lib.elph_kpm_apply_dev(m._h, dz.p, dr.p))
    check(lib.elph_synchronize(m._h))
    assert np.array_equal(dz.get(), z)
    dxp = _DevBuf(np.zeros(m.Ndim))
    check(lib.elph_ldiv_dev(m._h, dxp.p, db.p, 1, 0, C.byref(i1), C.byref(r1), C.byref(f1)))
    assert (i1.value, f1.value) == (itp, flp) and np.array_equal(dxp.get(), xp)
    # a caller's stream: the same results; NULL gives the handle a stream of its own again
    hip = _DevBuf.hip
    st = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(st)) == 0
    check(lib.elph_set_stream(m._h, st))
    dx2 = _DevBuf(np.zeros(m.Ndim))
    check(lib.elph_ldiv_dev(m._h, dx2.p, db.p, 0, 0, C.byref(i1), C.byref(r1), C.byref(f1)))
    assert i1.value == it and np.array_equal(dx2.get(), x)
    x3 = np.zeros(m.Ndim)
    assert models.ldiv_(x3, m, B[0])[0] == it and np.array_equal(x3, x)
    check(lib.elph_set_stream(m._h, None))
    x4 = np.zeros(m.Ndim)
    assert models.ldiv_(x4, m, B[0])[0] == it and np.array_equal(x4, x)
    assert hip.hipStreamDestroy(st) == 0
    for d in (db, dx, dB, dX, dr, dz, dxp, dx2):
        d.free()
    m.close()


def test_greens_function_results_stay_on_the_device():
    """elph_greens_dev_arrays: the four measured arrays of the last setup! as device pointers (a host that keeps accumulating on the GPU), equal to
    what elph_greens_setup copied out; elph_greens_nv: max(2, n_v) (GreensFunctions.jl:167)."""
    from elphdynamics_amd import _lib, configs, greens, synth
    from elphdynamics_amd._lib import check
    lib = _lib.load()
    m = configs.make_model("b", tol=1e-8)
    est = greens.EstimateGreensFunction(m, nv=1)
    nv = C.c_int()
    check(lib.elph_greens_nv(m._h, C.byref(nv)))
    assert nv.value == 2 == est.nv
    R = np.stack([synth.randn(950 + i, m.Ndim) for i in range(est.nv)])
    greens.update_(est, m, R=R)
    greens.setup_(est, 1, 2)
    arrs = (C.c_void_p * 4)()
    cnt = C.c_int64()
    check(lib.elph_greens_dev_arrays(m._h, arrs, C.byref(cnt)))
    assert cnt.value == est.GD0.size
    hip = C.CDLL("libamdhip64.so")
    for k, nm in enumerate(("GD0", "GD0_GD0", "GDD_G00", "GD0_G0D")):
        out = np.empty(2 * cnt.value)
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(arrs[k]), C.c_size_t(16 * cnt.value), 2) == 0
        host = np.asarray(getattr(est, nm)).reshape(-1, order="F").view(np.float64)
        assert np.array_equal(out, host), nm
    m.close()


def test_two_host_threads_each_with_its_own_handle():
    """Handles are independent: two host threads (one handle each, as a Julia process with two tasks or the rank threads of the sharded path hold them) solve
    at the same time — plain, preconditioned, batched — and every result is the bits of the same call made alone."""
    import threading
    from elphdynamics_amd import configs, models, preconditioners as pc

    def work(tag, seed, out, n):
        m = configs.make_model(tag, tol=1e-8)
        _, B = configs.rhs(m, 3, seed=seed)
        B = np.ascontiguousarray(B)
        P = pc.SymmetricKPMPreconditioner(m, min(20, m.Nsites), 0.05, 1.0, 1.0)
        pc.setup_(P, rng=np.random.default_rng(seed))
        res = []
        for _ in range(n):
            x = np.zeros(m.Ndim)
            it = models.ldiv_(x, m, B[0])[0]
            xp = np.zeros(m.Ndim)
            itp = models.ldiv_(xp, m, B[1], P)[0]
            X = np.zeros_like(B)
            itb = models.ldiv_batched_(X, m, B)[0]
            res.append((it, itp, tuple(itb), x.copy(), xp.copy(), X.copy()))
        m.close()
        out.append(res)

    alone_a, alone_b = [], []
    work("B", 41, alone_a, 1)
    work("d", 43, alone_b, 1)
    ta, tb = [], []
    th = [threading.Thread(target=work, args=("B", 41, ta, 12)), threading.Thread(target=work, args=("d", 43, tb, 12))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert len(ta) == 1 and len(tb) == 1
    for got, ref in ((ta[0], alone_a[0][0]), (tb[0], alone_b[0][0])):
        for r in got:
            assert r[:3] == ref[:3] and all(np.array_equal(a, b) for a, b in zip(r[3:], ref[3:]))
