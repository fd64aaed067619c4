"""The slab form of the resident un-preconditioned solve (csrc/slabs.hip): a lattice beyond one wave's slice (N > 320 sites) cut into slabs
of rows on ONE device, the sharded resident kernel per slab, all slabs in one launch.  Against the streaming iteration (ELPH_SLABS=0) and
the oracle (IterativeSolvers.jl:239-314, Models.jl:74-186)."""
import ctypes as C
import os

import numpy as np
import pytest

from test_gpu_parity import _oracle_model, rel

pytestmark = pytest.mark.gpu


def _info(m, nrhs=1):
    use, P, nloc, own = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    from elphdynamics_amd._lib import check
    check(m._lib.elph_bench_slabs_info(m._h, nrhs, C.byref(use), C.byref(P), C.byref(nloc), C.byref(own)))
    return use.value, P.value, nloc.value, own.value


@pytest.fixture
def slabs_env():
    old = {k: os.environ.get(k) for k in ("ELPH_SLABS", "ELPH_SLABS_P", "ELPH_SLABS_RING")}
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


# square 18, 20, 24, 28, 30, 32; honeycomb 18 x 18 and 20 x 20 cells (two sites per cell); triangular 24 x 24 has six colours: no slabs
@pytest.mark.parametrize("tag", ["i", "k", "g", "j", "l30", "G", "H18", "h20"])
def test_slab_solve_vs_streaming_and_oracle(oracle, slabs_env, tag):
    from elphdynamics_amd import configs, models
    os.environ["ELPH_SLABS"] = "1"                       # wherever the decomposition exists (the default rule takes a subset: next test)
    m = configs.make_model(tag, tol=1e-5)
    use, P, nloc, own = _info(m)
    assert use == 1 and 2 <= P <= 8 and own * P == m.Nsites and own < nloc <= 320, (use, P, nloc, own)
    om = _oracle_model(oracle, m)
    R, B = configs.rhs(m, 2)
    b = np.ascontiguousarray(B[0])
    x = np.zeros(m.Ndim)
    it, res, flag = models.ldiv_(x, m, b)                # x = 0 on entry: the slab form
    os.environ["ELPH_SLABS"] = "0"
    xs = np.zeros(m.Ndim)
    its, ress, flags = models.ldiv_(xs, m, b)
    xo, ito, reso, flago = oracle.ldiv(om, b, solver_tol=1e-5, solver_maxiter=10000)
    assert flag == flags == flago == 0 and abs(it - its) <= 1 and abs(it - ito) <= 1
    assert 0.5 * reso < res < 2.0 * reso and res <= np.sqrt(1e-5)
    assert rel(x, xs) < 1e-4 and not np.array_equal(x, xs)      # (another summation tree: the same solve to the tolerance, not the same bits; the tight solve below is the parity check)
    # deterministic: the same bits again
    os.environ["ELPH_SLABS"] = "1"
    x2 = np.zeros(m.Ndim)
    assert models.ldiv_(x2, m, b)[0] == it and np.array_equal(x, x2)
    # a caller's initial guess (x != 0) is the streaming iteration's: the bits of ELPH_SLABS=0
    g0 = 0.5 * xs
    xa, xb = g0.copy(), g0.copy()
    ita = models.ldiv_(xa, m, b)[0]
    os.environ["ELPH_SLABS"] = "0"
    itb = models.ldiv_(xb, m, b)[0]
    assert ita == itb and np.array_equal(xa, xb)
    # two right-hand sides — two SETS of slabs in one launch where they fit the chip together, else one launch after the other — are the
    # single solves, bit for bit; three: a pair and a single
    os.environ["ELPH_SLABS"] = "1"
    x1 = np.zeros(m.Ndim)
    it1 = models.ldiv_(x1, m, np.ascontiguousarray(B[1]))[0]
    X = np.zeros((2, m.Ndim))
    itB, resB, flB = models.ldiv_batched_(X, m, np.ascontiguousarray(B))
    assert itB[0] == it and itB[1] == it1 and np.array_equal(X[0], x) and np.array_equal(X[1], x1) and flB[1] == 0
    B3 = np.ascontiguousarray(np.stack([B[1], B[0], B[1]]))
    X3 = np.zeros((3, m.Ndim))
    it3b = models.ldiv_batched_(X3, m, B3)[0]
    assert list(it3b) == [it1, it, it1] and np.array_equal(X3[0], x1) and np.array_equal(X3[1], x) and np.array_equal(X3[2], x1)
    # tight solve against the oracle: the north_star's bound on M^-1 R
    m.solver.tol = 1e-13
    x3 = np.zeros(m.Ndim)
    it3, res3, flag3 = models.ldiv_(x3, m, b)
    xo3, ito3, *_ = oracle.ldiv(om, b, solver_tol=1e-13, solver_maxiter=10000)
    assert flag3 == 0 and abs(it3 - ito3) <= max(3, ito3 // 100)
    assert rel(x3, xo3) < 1e-10
    m.close()


def test_slab_rule_and_hopping_disorder(oracle, slabs_env):
    """The default rule (no ELPH_SLABS): from 576 sites, an even slab count, slabs of at most 256 sites, one right-hand side; below that and
    for two right-hand sides the streaming iteration.  Hopping disorder takes the per-bond tables of the slabs."""
    from elphdynamics_amd import configs, models
    os.environ.pop("ELPH_SLABS", None)
    os.environ.pop("ELPH_SLABS_P", None)
    for tag, want in (("g", (1, 6, 192, 96)), ("G", (1, 8, 256, 128)), ("k", (0, 0, 0, 0)), ("j", (0, 0, 0, 0))):
        m = configs.make_model(tag, tol=1e-5)
        assert _info(m, 1) == want, (tag, _info(m, 1))
        assert _info(m, 2)[0] == want[0] and _info(m, 3)[0] == 0      # (8 time slices: one workgroup per slab, two sets fit the chip)
        m.close()
    # 160 time slices = 20 workgroups per slab: two sets of 6 slabs are 240 workgroups (fit), two sets of 8 are 320 (do not)
    from elphdynamics_amd import lattice as lat
    for Ls, pair in ((24, 1), (32, 0)):
        configs.CONFIGS["_slab_rule"] = ("holstein", 1, Ls, lat.SQUARE_BONDS, 16.0, 0.1)
        m = configs.make_model("_slab_rule", tol=1e-5)
        assert _info(m, 1)[0] == 1 and _info(m, 2)[0] == pair
        m.close()
    configs.CONFIGS.pop("_slab_rule")
    m = configs.make_model("g", tol=1e-5, t_stddev=0.1)
    assert _info(m, 1)[0] == 1
    om = _oracle_model(oracle, m)
    _, B = configs.rhs(m, 1)
    b = np.ascontiguousarray(B[0])
    m.solver.tol = 1e-13
    x = np.zeros(m.Ndim)
    it, res, flag = models.ldiv_(x, m, b)
    xo, ito, *_ = oracle.ldiv(om, b, solver_tol=1e-13, solver_maxiter=10000)
    assert flag == 0 and abs(it - ito) <= max(3, ito // 100) and rel(x, xo) < 1e-10
    # the model moves (update_model!): the slabs follow
    m.x[:] = 0.7 * m.x[::-1]
    models.update_model_(m)
    om2 = _oracle_model(oracle, m)
    x2 = np.zeros(m.Ndim)
    it2, res2, flag2 = models.ldiv_(x2, m, b)
    xo2, ito2, *_ = oracle.ldiv(om2, b, solver_tol=1e-13, solver_maxiter=10000)
    assert flag2 == 0 and abs(it2 - ito2) <= max(3, ito2 // 100) and rel(x2, xo2) < 1e-10
    m.close()


def test_slabs_closed_into_rings_take_the_grid_form(oracle, slabs_env):
    """ELPH_SLABS_RING=1 (opt-in: measured slower): on a recognised square lattice the slabs are closed into rings — periodic rectangles in the
    reference's colouring, the GRID form of the sharded kernel; the own sites do not see the ring bond.  Same solve."""
    from elphdynamics_amd import configs, models
    os.environ["ELPH_SLABS"] = "1"
    res = {}
    for ring in ("0", "1"):
        os.environ["ELPH_SLABS_RING"] = ring
        m = configs.make_model("G", tol=1e-13)
        _, B = configs.rhs(m, 1)
        b = np.ascontiguousarray(B[0])
        x = np.zeros(m.Ndim)
        it, r, flag = models.ldiv_(x, m, b)
        assert flag == 0
        res[ring] = (it, x)
        if ring == "1":
            xo, ito, *_ = oracle.ldiv(_oracle_model(oracle, m), b, solver_tol=1e-13, solver_maxiter=10000)
            assert abs(it - ito) <= max(3, ito // 100) and rel(x, xo) < 1e-10
        m.close()
    assert abs(res["0"][0] - res["1"][0]) <= 2 and rel(res["0"][1], res["1"][1]) < 1e-10 and not np.array_equal(res["0"][1], res["1"][1])


@pytest.mark.parametrize("how", ["1", "2"])
def test_slab_time_out_falls_back_and_comes_back(slabs_env, how):
    """A slab launch that gives up (ELPH_SLABS_TEST_TIMEOUT: 1 = forced on the host side, 2 = a real one — the last slab's workgroups are
    not launched and the others wait for them until their bound, 200 ms here) is re-solved by the streaming iteration — the solution is the
    streaming form's, the handle counts the event, cools down for 16 eligible solves and then takes the slab form again."""
    from elphdynamics_amd import configs, models
    from elphdynamics_amd._lib import check
    os.environ.pop("ELPH_SLABS", None)
    m = configs.make_model("G", tol=1e-9)
    _, B = configs.rhs(m, 1)
    b = np.ascontiguousarray(B[0])

    def solve():
        x = np.zeros(m.Ndim)
        it, res, flag = models.ldiv_(x, m, b)
        assert flag == 0
        cd, fb = C.c_int(), C.c_int64()
        check(m._lib.elph_wg_status(m._h, C.byref(cd), C.byref(fb)))
        return x, it, cd.value, fb.value

    x_slab, it_slab, cd, fb = solve()
    assert cd == 0 and fb == 0
    os.environ["ELPH_SLABS"] = "0"
    x_str, it_str, _, _ = solve()
    os.environ.pop("ELPH_SLABS")
    assert not np.array_equal(x_slab, x_str) and rel(x_slab, x_str) < 1e-7
    os.environ["ELPH_SLABS_TEST_TIMEOUT"] = how
    old_to = os.environ.get("ELPH_WG_TIMEOUT_MS")
    os.environ["ELPH_WG_TIMEOUT_MS"] = "200"
    try:
        x1, it1, cd1, fb1 = solve()
    finally:
        os.environ.pop("ELPH_SLABS_TEST_TIMEOUT")
        if old_to is None:
            os.environ.pop("ELPH_WG_TIMEOUT_MS")
        else:
            os.environ["ELPH_WG_TIMEOUT_MS"] = old_to
    assert fb1 == 1 and cd1 > 0 and np.array_equal(x1, x_str) and it1 == it_str        # given up, re-solved by the streaming iteration
    seen_slab_again = False
    for k in range(20):
        x, it, cd, fb = solve()
        assert fb == 1
        if np.array_equal(x, x_slab):
            seen_slab_again = True
            assert cd == 0 and k >= 15
            break
        assert np.array_equal(x, x_str) and cd > 0
    assert seen_slab_again
    m.close()


def test_slab_form_edge_cases_match_the_streaming_iteration(slabs_env):
    """ldiv!'s outcomes other than convergence, through the slab form and through the streaming pair: an iteration limit that is reached
    (flag and zero-fill, Models.jl:150-186), the κ stop, a zero right-hand side, and solve!'s residual history (IterativeSolvers.jl:286-295)."""
    from elphdynamics_amd import configs, models
    m = configs.make_model("G", tol=1e-9, maxiter=10000)
    _, B = configs.rhs(m, 1)
    b = np.ascontiguousarray(B[0])

    def both(fn):
        out = {}
        for mode in ("1", "0"):
            os.environ["ELPH_SLABS"] = mode
            out[mode] = fn()
        return out["1"], out["0"]

    # the iteration limit: same count, same flag, same zero-fill
    def limited():
        x = np.full(m.Ndim, 0.0)
        it, res, flag = models.ldiv_(x, m, b, maxiter=7)
        return it, flag, float(np.abs(x).max()), res
    a, s = both(limited)
    assert a[0] == s[0] == 7 and a[1] == s[1] and a[1] != 0 and (a[2] == 0.0) == (s[2] == 0.0) and abs(a[3] - s[3]) <= 1e-9 * abs(s[3])

    # a zero right-hand side is 0/0 in the reference's CG as well (eps = |r| / |b|, alpha = r.r / p.Ap: IterativeSolvers.jl:259-285 — it runs
    # to the iteration limit on NaNs): both forms do exactly that, with the same count and flag
    def zero_rhs():
        x = np.zeros(m.Ndim)
        it, res, flag = models.ldiv_(x, m, np.zeros(m.Ndim), maxiter=50)
        return it, flag, bool(np.isnan(x).all())
    a, s = both(zero_rhs)
    assert a == s

    # solve! with its history: the same eps sequence to rounding, the same count
    def hist():
        x = np.zeros(m.Ndim)
        it, h = models.solve_(x, m, b, tol=1e-9, history=True)
        return it, h, x
    a, s = both(hist)
    assert abs(a[0] - s[0]) <= 1
    n = min(a[0], s[0], 41)
    assert np.max(np.abs(a[1][:n] - s[1][:n]) / s[1][:n]) < 1e-10 and a[1][0] == s[1][0] == 1.0
    assert a[1][a[0]] < 1e-9 <= a[1][a[0] - 1]
    assert rel(a[2], s[2]) < 1e-7

    # the kappa stop (IterativeSolvers.jl:289-295): a tiny kappa_max ends both forms at the same early iteration
    def kappa():
        x = np.zeros(m.Ndim)
        return models.solve_(x, m, b, tol=1e-9, kmax=4.0)
    a, s = both(kappa)
    assert a == s and 1 <= a < 41
    m.close()


def test_hmc_update_on_a_large_lattice_through_the_slab_form(slabs_env):
    """One un-preconditioned HMC update (HMC.jl:343-463) on the 24 x 24 lattice: the two pseudofermion systems of every force / action
    evaluation run as two sets of slabs in one launch (x0 = 0 is the library's own: fill!(O⁻¹Λϕ, 0), HMC.jl:854) — the same trajectory as
    with the streaming pair, to the solver tolerance."""
    from elphdynamics_amd import configs, hmc, preconditioners as pc
    from test_gpu_hmc import _randoms
    out = {}
    for mode in ("1", "0"):
        os.environ["ELPH_SLABS"] = mode
        m = configs.make_model("g", tol=1e-8, maxiter=40000)
        fa = pc.FourierAccelerator(m)
        pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.3)
        dt, nt = 0.02, 2
        H = hmc.HybridMonteCarlo(m, fa, dt, nt * dt)
        acc, its = hmc.update_(m, H, fa, None, randoms=_randoms(m, nt, 1500, False, 0.0))
        assert H.flag == 0
        if mode == "1":
            use, P, nloc, own = _info(m, 2)
            assert use == 1 and P == 6
        out[mode] = (acc, its, H.H0, H.H1, m.x.copy(), H.v.copy())
        m.close()
    a, s = out["1"], out["0"]
    assert a[0] == s[0] and abs(a[1] - s[1]) <= 1
    assert abs(a[2] - s[2]) < 1e-9 * abs(s[2]) and abs(a[3] - s[3]) < 1e-8 * abs(s[3])
    assert rel(a[4], s[4]) < 1e-7 and rel(a[5], s[5]) < 1e-7 and not np.array_equal(a[4], s[4])


def test_slabs_follow_the_callers_stream_and_are_freed_with_the_handle(slabs_env):
    """elph_set_stream on the lattice's handle after its slabs exist: the slab handles are re-bound (one launch needs ONE stream) and the solve is the
    same bits; creating and destroying models that grew slabs (two sets: a pair of right-hand sides) leaves the device's free memory where it was."""
    from elphdynamics_amd import _lib, configs, models
    from elphdynamics_amd._lib import check
    from test_gpu_muldmdx import _DevBuf
    lib = _lib.load()
    os.environ.pop("ELPH_SLABS", None)
    m = configs.make_model("g", tol=1e-9)
    _, B = configs.rhs(m, 2)
    B = np.ascontiguousarray(B)
    X = np.zeros_like(B)
    it, res, fl = models.ldiv_batched_(X, m, B)
    assert _info(m, 2)[0] == 1 and not fl.any()
    if _DevBuf.hip is None:
        _DevBuf.hip = C.CDLL("libamdhip64.so")
    hip = _DevBuf.hip
    st = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(st)) == 0
    check(lib.elph_set_stream(m._h, st))
    X2 = np.zeros_like(B)
    it2, _, fl2 = models.ldiv_batched_(X2, m, B)
    assert np.array_equal(it, it2) and np.array_equal(X, X2) and not fl2.any()
    check(lib.elph_set_stream(m._h, None))
    X3 = np.zeros_like(B)
    assert np.array_equal(models.ldiv_batched_(X3, m, B)[0], it) and np.array_equal(X, X3)
    assert hip.hipStreamDestroy(st) == 0
    m.close()

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    def cycle():
        mm = configs.make_model("g", tol=1e-6)
        Xc = np.zeros_like(B)
        models.ldiv_batched_(Xc, mm, B)
        mm.close()

    cycle()
    f0 = free_bytes()
    for _ in range(6):
        cycle()
    assert abs(free_bytes() - f0) < (8 << 20), (f0, free_bytes())      # (the allocator may keep a few MB of pools; a leaked slab set is ~40 MB per cycle)
