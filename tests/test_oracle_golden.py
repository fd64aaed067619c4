"""Pin the CPU oracle (oracle/elph_oracle.c) against the self-derived golden fixtures.

The reference has no tests or golden vectors, so the fixtures come from an
independent dense numpy/scipy restatement of the definitions (tests/golden/make_golden.py).
Integer tables must match bit-for-bit; floating point within 1e-12 relative (dense products vs
loop order), CG solutions within 1e-9 of the dense solve.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import golden
from elphdynamics_amd import synth
from oracle.oracle import dp, ip

SQUARE = [(1, 1, (1, 0, 0)), (1, 1, (0, 1, 0))]
HONEY = [(1, 2, (0, 0, 0)), (1, 2, (-1, 0, 0)), (1, 2, (0, -1, 0))]
TRI = [(1, 1, (1, 0, 0)), (1, 1, (0, 1, 0)), (1, 1, (1, -1, 0))]
CASES = [("sq2", 1, 2, 2, SQUARE), ("sq4", 1, 4, 4, SQUARE), ("sq8", 1, 8, 8, SQUARE), ("sq16", 1, 16, 16, SQUARE),
         ("hc3", 2, 3, 3, HONEY), ("hc12", 2, 12, 12, HONEY), ("tri3", 1, 3, 3, TRI),
         ("chain6", 1, 6, 1, [(1, 1, (1, 0, 0))])]


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("tag,norb,L1,L2,bonds", CASES)
def test_tables_bit_exact(oracle, tag, norb, L1, L2, bonds):
    g = golden("tables.npz")
    raw = oracle.neighbor_table(norb, L1, L2, 1, bonds)
    assert np.array_equal(raw, g[tag + "_raw"])
    tab, c, s, perm, grp, ng = oracle.holstein_initialize(raw, np.ones(raw.shape[0]), 0.1)
    assert np.array_equal(tab, g[tag + "_table"])
    assert np.array_equal(grp, g[tag + "_colour"])
    assert np.array_equal(perm, g[tag + "_cbperm"])
    assert ng == g[tag + "_colour"].max()
    tab2, perm2, iperm2, grp2, ng2 = oracle.ssh_initialize_table(raw)
    assert np.array_equal(tab2, tab) and np.array_equal(perm2, perm) and ng2 == ng
    # cb_perm is the inverse of inv_cb_perm (SSHModels.jl:446-447)
    assert np.array_equal(iperm2[perm2 - 1], np.arange(1, raw.shape[0] + 1))


def test_colour_structure_square_and_honeycomb():
    """even-L square -> 4 perfect matchings; honeycomb -> 3."""
    g = golden("tables.npz")
    for tag, ncol, n in (("sq16", 4, 256), ("sq8", 4, 64), ("hc12", 3, 288)):
        col = g[tag + "_colour"]
        tab = g[tag + "_table"]
        assert col.max() == ncol
        for cc in range(1, ncol + 1):
            sites = tab[col == cc].reshape(-1)
            assert len(sites) == n and len(set(sites.tolist())) == n
    assert golden("tables.npz")["tri3_colour"].max() > 3   # odd-L triangular: ragged colours


def test_ltau_round_half_even(oracle):
    assert oracle.lib.elpho_ltau(16.0, 0.1) == 160
    assert oracle.lib.elpho_ltau(2.0, 0.1) == 20
    assert oracle.lib.elpho_ltau(2.5, 1.0) == 2 and oracle.lib.elpho_ltau(3.5, 1.0) == 4


def _holstein_model(oracle, g):
    N, L = int(g["N"]), int(g["Ltau"])
    tab, c, s, perm, grp, ng = oracle.holstein_initialize(g["raw"], g["t_raw"], float(g["dtau"]))
    assert np.array_equal(tab, g["table"])
    assert rel(c, g["cosht"]) < 1e-15 and rel(s, g["sinht"]) < 1e-15
    E = oracle.update_model_holstein(N, L, float(g["dtau"]), g["x"], g["lam"], g["lam2"], g["mu"])
    assert rel(E, g["E"]) < 1e-15
    return oracle.make_model(0, N, L, tab, c, s, E), N, L


@pytest.mark.parametrize("name", ["holstein_sq4_L8.npz", "holstein_hc3_L6.npz", "holstein_tri3_L5.npz",
                                  "holstein_sq4_L40.npz"])
def test_holstein_matvec_and_solve(oracle, name):
    g = golden(name)
    m, N, L = _holstein_model(oracle, g)
    v = np.ascontiguousarray(g["v"])
    assert rel(oracle.mulM(m, v), g["Mv"]) < 1e-13
    assert rel(oracle.mulMT(m, v), g["MTv"]) < 1e-13
    assert rel(oracle.mulMTM(m, v), g["MTMv"]) < 1e-13
    # checkerboard kernels vs dense product / inverse
    tab, c, s = m._keep["table"], m._keep["c"], m._keep["s"]
    nb = tab.shape[0]
    for fn, key in (("elpho_checkerboard_mul", "CBv"), ("elpho_checkerboard_transpose_mul", "CBTv"),
                    ("elpho_checkerboard_inverse_mul", "CBinv_v")):
        y = v.copy()
        getattr(oracle.lib, fn)(dp(y), ip(tab), dp(c), dp(s), nb, L)
        assert rel(y, g[key]) < 1e-13
    y = v.copy()
    oracle.lib.elpho_checkerboard_mul(dp(y), ip(tab), dp(c), dp(s), nb, L)
    oracle.lib.elpho_checkerboard_inverse_mul(dp(y), ip(tab), dp(c), dp(s), nb, L)
    assert rel(y, v) < 1e-13
    y = v.copy()
    oracle.lib.elpho_checkerboard_transpose_mul(dp(y), ip(tab), dp(c), dp(s), nb, L)
    oracle.lib.elpho_checkerboard_inverse_transpose_mul(dp(y), ip(tab), dp(c), dp(s), nb, L)
    assert rel(y, v) < 1e-13
    # adjointness <u, M v> = <M^T u, v>
    u = np.ascontiguousarray(g["R"])
    assert abs(u @ oracle.mulM(m, v) - oracle.mulMT(m, u) @ v) < 1e-12 * np.linalg.norm(u) * np.linalg.norm(v)
    # CG on MtM x = Mt R  vs dense solve; ldiv! flags
    b = np.ascontiguousarray(g["b"])
    x, it, hist = oracle.cg_solve(m, b, tol=1e-12, maxiter=5000, history=True)
    assert rel(x, g["xsol"]) < 1e-9
    assert rel(x, g["Minv_R"]) < 1e-9          # Green's-function observable M^-1 R (GreensFunctions.jl:223-225)
    assert hist[-1] < 1e-12 <= hist[-2]
    x2, it2, res, flag = oracle.ldiv(m, b, solver_tol=1e-10, solver_maxiter=5000)
    assert flag == 0 and res <= 1e-5 and rel(x2, g["xsol"]) < 1e-7
    # hit-maxiter path: flag 1, x zeroed (Models.jl:157-166)
    x3, it3, res3, flag3 = oracle.ldiv(m, b, solver_tol=1e-14, solver_maxiter=3)
    assert it3 == 3 and flag3 == 1 and not x3.any()
    # explicit maxiter != solver.maxiter -> reference quirk yields flag 2 (Models.jl:160)
    x4, it4, res4, flag4 = oracle.ldiv(m, b, maxiter=3, solver_tol=1e-14, solver_maxiter=5000)
    assert it4 == 3 and flag4 == 2 and not x4.any()


def test_single_site_closed_form(oracle):
    g = golden("holstein_single_site.npz")
    L = int(g["Ltau"])
    E = oracle.update_model_holstein(1, L, float(g["dtau"]), g["x"], np.ones(1), np.zeros(1), np.zeros(1))
    assert rel(E, g["E"]) < 1e-15
    m = oracle.make_model(0, 1, L, np.zeros((0, 2), dtype=np.int64), np.zeros(1), np.zeros(1), E)
    b = np.ascontiguousarray(g["b"])
    assert rel(oracle.mulM(m, b), g["Mb"]) < 1e-14
    assert rel(oracle.mulMT(m, b), g["MTb"]) < 1e-14
    x, it = oracle.cg_solve(m, oracle.mulMT(m, b), tol=1e-13, maxiter=200)
    assert rel(x, g["Minv_b"]) < 1e-10
    assert abs(g["detM"] - g["det_closed"]) < 1e-12
    assert np.allclose(g["G_tt"], g["G_closed"], rtol=1e-12)


def test_ssh_matvec_and_solve(oracle):
    g = golden("ssh_sq4_L8.npz")
    N, L, dtau = int(g["N"]), int(g["Ltau"]), float(g["dtau"])
    tab, perm, iperm, grp, ng = oracle.ssh_initialize_table(g["raw"])
    assert np.array_equal(tab, g["table"]) and np.array_equal(perm, g["cbperm"])
    assert np.array_equal(iperm, g["inv_cbperm"])
    nb = tab.shape[0]
    Nph = nb
    c = np.zeros(nb * L)
    s = np.zeros(nb * L)
    Emu = np.zeros(N)
    oracle.lib.elpho_update_model_ssh(N, L, nb, Nph, dtau, dp(np.ascontiguousarray(g["x"])), dp(g["t"]),
                                      dp(g["alpha"]), dp(g["alpha2"]), dp(g["mu"]),
                                      ip(g["phonon_to_bond"]), ip(perm), dp(c), dp(s), dp(Emu))
    assert rel(c, g["cosht"]) < 1e-15 and rel(s, g["sinht"]) < 1e-15 and rel(Emu, g["expDtauMu"]) < 1e-15
    m = oracle.make_model(1, N, L, tab, c, s, Emu)
    v = np.ascontiguousarray(g["v"])
    assert rel(oracle.mulM(m, v), g["Mv"]) < 1e-13
    assert rel(oracle.mulMT(m, v), g["MTv"]) < 1e-13
    assert rel(oracle.mulMTM(m, v), g["MTMv"]) < 1e-13
    x, it = oracle.cg_solve(m, np.ascontiguousarray(g["b"]), tol=1e-12, maxiter=5000)
    assert rel(x, g["xsol"]) < 1e-9


def test_muldMdx_holstein_is_the_derivative_of_uMv(oracle):
    """elpho_muldMdx_holstein (HolsteinModels.jl:691-755) against ∂(uᵀMv)/∂x_f from complex-step differentiation of the dense M
    (tests/golden/make_golden.py::gen_dmdx — the definition, not a loop restatement)."""
    g, d = golden("holstein_sq4_L8.npz"), golden("muldmdx_sq4_L8.npz")
    m, N, L = _holstein_model(oracle, g)
    out = np.zeros(N * L)
    oracle.lib.elpho_muldMdx_holstein(dp(out), dp(np.ascontiguousarray(d["u"])), C.byref(m), dp(np.ascontiguousarray(d["v"])),
                                      float(g["dtau"]), dp(g["lam"]), dp(g["lam2"]), dp(np.ascontiguousarray(g["x"])))
    assert rel(out, d["dMdx"]) < 1e-13


def test_muldMdx_ssh_is_the_derivative_of_uMv(oracle):
    """elpho_muldMdx_ssh (SSHModels.jl:707-829; α₂ = 0 so that the reference's ∂K/∂x is the true derivative) against the same
    definition-level fixture."""
    g, d = golden("ssh_sq4_L8_a.npz"), golden("muldmdx_ssh_sq4_L8_a.npz")
    N, L, dtau = int(g["N"]), int(g["Ltau"]), float(g["dtau"])
    tab = np.ascontiguousarray(g["table"])
    nb = tab.shape[0]
    m = oracle.make_model(1, N, L, tab, np.ascontiguousarray(g["cosht"]), np.ascontiguousarray(g["sinht"]), np.ascontiguousarray(g["expDtauMu"]))
    nph = g["phonon_to_bond"].shape[0]
    b2p = np.zeros(nb, dtype=np.int64)                                  # phonon (1-based) on checkerboard bond n, 0 = none
    b2p[g["cbperm"][g["phonon_to_bond"] - 1] - 1] = np.arange(1, nph + 1)
    out = np.zeros(nph * L)
    oracle.lib.elpho_muldMdx_ssh(dp(out), dp(np.ascontiguousarray(d["u"])), C.byref(m), dp(np.ascontiguousarray(d["v"])), dtau, ip(b2p),
                                 dp(np.ascontiguousarray(g["alpha"])), dp(np.ascontiguousarray(g["alpha2"])), dp(np.ascontiguousarray(g["x"])), nph)
    assert rel(out, d["dMdx"]) < 1e-12


@pytest.mark.parametrize("L", [8, 20, 40, 120, 160, 7])
def test_fft_and_fourier_acceleration(oracle, L):
    g = golden("fft.npz")
    N = 3
    v = np.ascontiguousarray(g[f"L{L}_v"])
    nu = np.zeros(2 * N * L)
    oracle.lib.elpho_tau_to_omega(dp(nu), dp(v), N, L)
    assert rel(nu[0::2], g[f"L{L}_nu_re"]) < 1e-13 and rel(nu[1::2], g[f"L{L}_nu_im"]) < 1e-13
    back = np.zeros(N * L)
    oracle.lib.elpho_omega_to_tau(dp(back), dp(nu), N, L)
    assert rel(back, g[f"L{L}_back"]) < 1e-13 and rel(back, v) < 1e-13
    Mi = np.array([oracle.lib.elpho_element_Mi(k, 1.0, 0.1, 0.1, 2.0, L) for k in range(L)])
    Qi = np.array([oracle.lib.elpho_element_Qi(k, 1.0, 0.1, 0.5, L) for k in range(L)])
    assert rel(Mi, g[f"L{L}_Mi"]) < 1e-14 and rel(Qi, g[f"L{L}_Qi"]) < 1e-14
    diag = np.zeros(N * L)
    oracle.lib.elpho_update_M(dp(diag), N, L, 0.1, dp(np.ones(N)), 0.0, 10.0, 0.1, 2.0)
    for power in (-1.0, -0.5, 1.0):
        out = np.zeros(N * L)
        oracle.lib.elpho_fourier_accelerate(dp(out), dp(v), dp(diag), power, N, L)
        assert rel(out, g[f"L{L}_fa_M_p{power}"]) < 1e-13


def test_eigvals_small_dense(oracle):
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 7, 20):
        a = rng.standard_normal((n, n))
        ref = np.sort_complex(np.linalg.eigvals(a))
        af = np.asfortranarray(a).reshape(-1, order="F").copy()
        wr, wi = np.zeros(n), np.zeros(n)
        assert oracle.lib.elpho_eigvals(dp(af), n, dp(wr), dp(wi)) == 0
        got = np.sort_complex(wr + 1j * wi)
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("tag", ["sq4_L8", "sq4_L40"])
def test_kpm_against_dense(oracle, tag):
    g = golden(f"holstein_{tag}.npz")
    k = golden(f"kpm_{tag}.npz")
    m, N, L = _holstein_model(oracle, g)
    P = oracle.make_kpm(m, n=20, buf=float(k["buf"]), c1=float(k["c1"]), c2=float(k["c2"]))
    assert rel(P._keep["Ebar"], k["Ebar"]) < 1e-14
    # Arnoldi bounds bracket the exact spectrum of A = CBbar diag(Ebar) (n=min(20,N)=N steps => exact)
    rng = np.random.default_rng(1)
    e_min, e_max = oracle.kpm_setup(P, b_max=rng.standard_normal(N), b_min=rng.standard_normal(N))
    assert abs(e_min - float(k["e_min"])) < 1e-8 and abs(e_max - float(k["e_max"])) < 1e-8
    # inject exact bounds -> identical lam_lo/hi, orders, coefficients, apply
    P = oracle.make_kpm(m, n=20, buf=float(k["buf"]), c1=float(k["c1"]), c2=float(k["c2"]))
    oracle.kpm_setup(P, e_min=float(k["e_min"]), e_max=float(k["e_max"]))
    assert P.active == 1
    assert abs(P.lam_lo - float(k["lam_lo"])) < 1e-15 and abs(P.lam_hi - float(k["lam_hi"])) < 1e-15
    Lo2 = (L + 1) // 2
    assert np.array_equal(P._keep["order"][:Lo2], k["orders"])
    ntot = int(k["orders"].sum())
    cz = P._keep["coeff"][:2 * ntot]
    assert rel(cz[0::2], k["coeff_re"]) < 1e-12 and rel(cz[1::2], k["coeff_im"]) < 1e-12
    out = oracle.kpm_apply(P, np.ascontiguousarray(k["vin"]))
    assert rel(out, k["vout"]) < 1e-12
    assert P.checkerboard_count == 2 * sum(o - 1 for o in k["orders"])
    # preconditioned CG reaches the same solution, in fewer iterations than plain CG
    b = np.ascontiguousarray(g["b"])
    x0, it0 = oracle.cg_solve(m, b, tol=1e-10, maxiter=5000)
    x1, it1 = oracle.cg_solve(m, b, tol=1e-10, maxiter=5000, P=P)
    assert rel(x1, g["xsol"]) < 1e-8 and rel(x0, g["xsol"]) < 1e-8
    if L >= 40:
        assert it1 < it0
    # implausible bounds deactivate the preconditioner -> identity (KPMPreconditioners.jl:312-318,475-478)
    oracle.lib.elpho_kpm_setup_from_bounds(C.byref(P), 1.5, 1.2)
    assert P.active == 0
    assert np.array_equal(oracle.kpm_apply(P, b), b)


def test_lambda_ops_roundtrip(oracle):
    g = golden("holstein_sq4_L8.npz")
    N, L = int(g["N"]), int(g["Ltau"])
    Lam = np.zeros(N * L)
    oracle.lib.elpho_update_Lambda(dp(Lam), N, L, float(g["dtau"]), dp(np.ascontiguousarray(g["x"])),
                                   dp(g["lam"]), dp(g["lam2"]))
    X = g["x"].reshape(N, L)
    assert rel(Lam, np.exp(-float(g["dtau"]) * (g["lam"][:, None] * X + g["lam2"][:, None] * X ** 2) / 2).reshape(-1)) < 1e-15
    v = np.ascontiguousarray(g["v"])
    a, b = np.zeros(N * L), np.zeros(N * L)
    oracle.lib.elpho_mulLambda(dp(a), dp(v), dp(Lam), N, L)
    oracle.lib.elpho_mulLambdaInv(dp(b), dp(a), dp(Lam), N, L)
    assert rel(b, v) < 1e-14


def test_muldMdx_finite_difference(oracle):
    """<u, dM/dx_k v> from muldMdx! equals a central finite difference of <u, M(x) v>."""
    g = golden("holstein_sq4_L8.npz")
    m, N, L = _holstein_model(oracle, g)
    dtau = float(g["dtau"])
    u, v, x = (np.ascontiguousarray(g[k]) for k in ("R", "v", "x"))
    out = np.zeros(N * L)
    oracle.lib.elpho_muldMdx_holstein(dp(out), dp(u), C.byref(m), dp(v), dtau, dp(g["lam"]), dp(g["lam2"]), dp(x))
    h = 1e-6
    for k in (0, 5, L, 3 * L + 2, N * L - 1):
        f = []
        for sgn in (+1, -1):
            xx = x.copy()
            xx[k] += sgn * h
            E = oracle.update_model_holstein(N, L, dtau, xx, g["lam"], g["lam2"], g["mu"])
            mm = oracle.make_model(0, N, L, m._keep["table"], m._keep["c"], m._keep["s"], E)
            f.append(u @ oracle.mulM(mm, v))
        assert abs((f[0] - f[1]) / (2 * h) - out[k]) < 1e-6 * max(1.0, abs(out[k]))


# ---------------------------------------------------------------------------------------------- kinetic-matrix aid

@pytest.mark.parametrize("norb,Lsp,bonds", [(1, 4, [(1, 1, (1, 0, 0)), (1, 1, (0, 1, 0))]),
                                            (2, 3, [(1, 2, (0, 0, 0)), (1, 2, (-1, 0, 0)), (1, 2, (0, -1, 0))])])
def test_checkerboard_product_approximates_expm_of_kinetic_matrix(oracle, norb, Lsp, bonds):
    """The reference's SSH kinetic-matrix dump (SSHModels.jl:915-944) exists to compare exp(-dtau K) with the checkerboard product:
    the product of the 2x2 bond blocks equals exp(-dtau K(tau)) up to the O(dtau^2) Trotter error of non-commuting colours —
    here with hoppings that differ on every bond and time slice (the SSH form), B(tau) read off mulM on unit vectors."""
    import scipy.linalg
    raw = oracle.neighbor_table(norb, Lsp, Lsp, 1, bonds)
    tab, perm, iperm, grp, ng = oracle.ssh_initialize_table(raw)
    N, L, nb = norb * Lsp * Lsp, 3, tab.shape[0]
    tp = 1.0 + 0.3 * synth.randn(99, nb * L).reshape(nb, L)                  # t'[bond (checkerboard order)][tau]
    errs = []
    for dtau in (0.1, 0.05, 0.025):
        c, s = np.cosh(dtau * tp), np.sinh(dtau * tp)
        om = oracle.make_model(1, N, L, tab, c.reshape(-1), s.reshape(-1), np.ones(N))
        err = 0.0
        for t in range(L):
            B = np.zeros((N, N))
            for j in range(N):                                               # column j of B(t): y(t) = v(t) -/+ B(t) v(t-1)
                v = np.zeros(N * L)
                v[j * L + (t - 1) % L] = 1.0
                y = oracle.mulM(om, v).reshape(N, L)[:, t]
                B[:, j] = y if t == 0 else -y
            K = np.zeros((N, N))
            for n in range(nb):
                i, jn = tab[n, 0] - 1, tab[n, 1] - 1
                K[i, jn] -= tp[n, t]
                K[jn, i] -= tp[n, t]
            err = max(err, np.abs(B - scipy.linalg.expm(-dtau * K)).max())
            assert abs(np.linalg.det(B) - 1.0) < 1e-12                        # every block has cosh^2 - sinh^2 = 1
        errs.append(err)
    assert errs[0] < 0.05 and 3.0 < errs[0] / errs[1] < 5.0 and 3.0 < errs[1] / errs[2] < 5.0      # O(dtau^2)
