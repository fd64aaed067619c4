"""ctypes binding of the CPU oracle (oracle/elph_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (elphdynamics_amd/) never
imports this module.  PARITY UNPINNED BY THE REFERENCE (see elph_oracle.h).

All arrays are numpy float64 / int64, flat, in the reference's layout
(tau fastest; neighbor tables 2 x Nbonds column-major, 1-based).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

c_i64 = C.c_int64
c_dbl = C.c_double
P_i64 = C.POINTER(C.c_int64)
P_dbl = C.POINTER(C.c_double)


def build(fast=False, omp=False):
    """(Re)build the oracle shared library with gcc via oracle/Makefile."""
    target = "omp" if omp else ("fast" if fast else "all")
    subprocess.run(["make", "-s", "-C", _HERE, target], check=True)
    return os.path.join(_BUILD, {"omp": "libelph_oracle_omp.so", "fast": "libelph_oracle_fast.so", "all": "libelph_oracle.so"}[target])


def cg_iterations_omp(model, b, niter, nthreads):
    """All-host-cores variant of the CG iteration (CPU baseline only; NOT the reference's configuration)."""
    lib = C.CDLL(build(omp=True))
    lib.elpho_cg_iterations_omp.restype = c_dbl
    lib.elpho_cg_iterations_omp.argtypes = [C.POINTER(Model), P_dbl, P_dbl, c_i64, C.c_int]
    x = np.zeros_like(b)
    secs = lib.elpho_cg_iterations_omp(C.byref(model), dp(x), dp(b), niter, nthreads)
    return secs, x


def _ptr(a, ty):
    if a is None:
        return None
    return a.ctypes.data_as(ty)


def dp(a):
    assert a is None or (a.dtype == np.float64 and a.flags["C_CONTIGUOUS"])
    return _ptr(a, P_dbl)


def ip(a):
    assert a is None or (a.dtype == np.int64 and a.flags["C_CONTIGUOUS"])
    return _ptr(a, P_i64)


class Model(C.Structure):
    _fields_ = [("kind", c_i64), ("N", c_i64), ("L", c_i64), ("nb", c_i64),
                ("table", P_i64), ("c", P_dbl), ("s", P_dbl), ("E", P_dbl),
                ("vp", P_dbl), ("vppp", P_dbl)]


class KPM(C.Structure):
    _fields_ = [("active", c_i64), ("N", c_i64), ("L", c_i64), ("nb", c_i64),
                ("table", P_i64), ("Ebar", P_dbl), ("cbar", P_dbl), ("sbar", P_dbl),
                ("lam_lo", c_dbl), ("lam_hi", c_dbl), ("lam_avg", c_dbl), ("lam_mag", c_dbl),
                ("buf", c_dbl), ("c1", c_dbl), ("c2", c_dbl),
                ("Lo2", c_i64), ("order", P_i64), ("coff", P_i64), ("coeff", P_dbl),
                ("coeff_cap", c_i64), ("v1", P_dbl), ("v2", P_dbl),
                ("v3", P_dbl), ("v4", P_dbl), ("v5", P_dbl), ("checkerboard_count", c_i64)]


class HmcParams(C.Structure):
    _fields_ = [("N", c_i64), ("L", c_i64), ("dtau", c_dbl), ("omega", P_dbl), ("omega4", P_dbl), ("lam", P_dbl),
                ("lam2", P_dbl), ("mu", P_dbl), ("fa_M", P_dbl), ("dt", c_dbl), ("nt", c_i64), ("nb", c_i64),
                ("alpha", c_dbl), ("solver_tol", c_dbl), ("solver_maxiter", c_i64), ("kmax", c_dbl), ("kpm_n", c_i64)]


class HmcSsh(C.Structure):
    _fields_ = [("Nph", c_i64), ("t", P_dbl), ("alpha", P_dbl), ("alpha2", P_dbl), ("phonon_to_bond", P_i64), ("cb_perm", P_i64),
                ("bond_to_phonon_cb", P_i64), ("primary_field", P_i64)]


class Oracle:
    """Loaded oracle library + thin numpy-level helpers."""

    def __init__(self, fast=False):
        path = build(fast=fast)
        self.lib = C.CDLL(path)
        L = self.lib
        L.elpho_loc_to_site.restype = c_i64
        L.elpho_loc_to_site.argtypes = [c_i64] * 8
        L.elpho_site_to_site.restype = c_i64
        L.elpho_site_to_site.argtypes = [c_i64, c_i64, c_i64, c_i64, c_i64, P_i64, c_i64]
        L.elpho_calc_neighbor_table.restype = c_i64
        L.elpho_calc_neighbor_table.argtypes = [c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, P_i64, C.c_int, P_i64]
        L.elpho_sortperm.argtypes = [P_i64, c_i64, P_i64]
        L.elpho_sorted_neighbor_table_perm.argtypes = [P_i64, c_i64, P_i64]
        L.elpho_checkerboard_groups.restype = c_i64
        L.elpho_checkerboard_groups.argtypes = [P_i64, c_i64, P_i64]
        L.elpho_holstein_initialize_model.restype = c_i64
        L.elpho_holstein_initialize_model.argtypes = [P_i64, c_i64, P_dbl, c_dbl, P_dbl, P_dbl, P_i64, P_i64]
        L.elpho_ssh_initialize_table.restype = c_i64
        L.elpho_ssh_initialize_table.argtypes = [P_i64, c_i64, P_i64, P_i64, P_i64]
        L.elpho_ltau.restype = c_i64
        L.elpho_ltau.argtypes = [c_dbl, c_dbl]
        L.elpho_update_model_holstein.argtypes = [c_i64, c_i64, c_dbl, P_dbl, P_dbl, P_dbl, P_dbl, P_dbl]
        L.elpho_update_model_ssh.argtypes = [c_i64, c_i64, c_i64, c_i64, c_dbl, P_dbl, P_dbl, P_dbl, P_dbl,
                                             P_dbl, P_i64, P_i64, P_dbl, P_dbl, P_dbl]
        for nm in ("elpho_checkerboard_mul", "elpho_checkerboard_transpose_mul",
                   "elpho_checkerboard_inverse_mul", "elpho_checkerboard_inverse_transpose_mul",
                   "elpho_checkerboard_mul_mat", "elpho_checkerboard_transpose_mul_mat"):
            getattr(L, nm).argtypes = [P_dbl, P_i64, P_dbl, P_dbl, c_i64, c_i64]
        for nm in ("elpho_checkerboard_mul_nvec_z", "elpho_checkerboard_transpose_mul_nvec_z",
                   "elpho_checkerboard_mul_nvec", "elpho_checkerboard_inverse_mul_nvec"):
            getattr(L, nm).argtypes = [P_dbl, P_i64, P_dbl, P_dbl, c_i64]
        for nm in ("elpho_mulM", "elpho_mulMT", "elpho_mulMTM"):
            getattr(L, nm).argtypes = [P_dbl, C.POINTER(Model), P_dbl]
        L.elpho_tau_to_omega.argtypes = [P_dbl, P_dbl, c_i64, c_i64]
        L.elpho_omega_to_tau.argtypes = [P_dbl, P_dbl, c_i64, c_i64]
        L.elpho_dft_tau.argtypes = [P_dbl, P_dbl, c_i64, c_i64, C.c_int]
        L.elpho_element_Mi.restype = c_dbl
        L.elpho_element_Mi.argtypes = [c_i64, c_dbl, c_dbl, c_dbl, c_dbl, c_i64]
        L.elpho_element_Qi.restype = c_dbl
        L.elpho_element_Qi.argtypes = [c_i64, c_dbl, c_dbl, c_dbl, c_i64]
        L.elpho_update_M.argtypes = [P_dbl, c_i64, c_i64, c_dbl, P_dbl, c_dbl, c_dbl, c_dbl, c_dbl]
        L.elpho_update_Q.argtypes = [P_dbl, c_i64, c_i64, c_dbl, P_dbl, c_dbl, c_dbl, c_dbl]
        L.elpho_fourier_accelerate.argtypes = [P_dbl, P_dbl, P_dbl, c_dbl, c_i64, c_i64]
        L.elpho_kpm_update_A.argtypes = [C.POINTER(KPM), C.POINTER(Model)]
        L.elpho_kpm_coefficients.argtypes = [P_dbl, c_i64, c_dbl, c_dbl, c_dbl]
        L.elpho_kpm_arnoldi_bounds.argtypes = [C.POINTER(KPM), c_i64, P_dbl, P_dbl, P_dbl, P_dbl]
        L.elpho_kpm_setup_from_bounds.argtypes = [C.POINTER(KPM), c_dbl, c_dbl]
        L.elpho_kpm_apply.argtypes = [P_dbl, C.POINTER(KPM), P_dbl]
        L.elpho_eigvals.restype = C.c_int
        L.elpho_eigvals.argtypes = [P_dbl, c_i64, P_dbl, P_dbl]
        L.elpho_cg_solve.restype = c_i64
        L.elpho_cg_solve.argtypes = [C.POINTER(Model), P_dbl, P_dbl, c_dbl, c_i64, c_dbl, C.POINTER(KPM),
                                     P_dbl, P_dbl, P_dbl, P_dbl]
        L.elpho_ldiv.argtypes = [C.POINTER(Model), P_dbl, P_dbl, C.POINTER(KPM), c_i64, c_dbl, c_i64, c_dbl,
                                 P_dbl, P_dbl, P_dbl, P_i64, P_dbl, P_i64]
        L.elpho_update_Lambda.argtypes = [P_dbl, c_i64, c_i64, c_dbl, P_dbl, P_dbl, P_dbl]
        L.elpho_mulLambda.argtypes = [P_dbl, P_dbl, P_dbl, c_i64, c_i64]
        L.elpho_mulLambdaInv.argtypes = [P_dbl, P_dbl, P_dbl, c_i64, c_i64]
        L.elpho_muldMdx_holstein.argtypes = [P_dbl, P_dbl, C.POINTER(Model), P_dbl, c_dbl, P_dbl, P_dbl, P_dbl]
        L.elpho_muldMdx_ssh.argtypes = [P_dbl, P_dbl, C.POINTER(Model), P_dbl, c_dbl, P_i64, P_dbl, P_dbl, P_dbl, c_i64]
        L.elpho_muldLambdadx_holstein.argtypes = [P_dbl, P_dbl, P_dbl, P_dbl, c_i64, c_i64, c_dbl, P_dbl, P_dbl, P_dbl]
        L.elpho_calc_dSfdx_holstein.argtypes = [P_dbl, C.POINTER(Model), P_dbl, P_dbl, P_dbl, P_dbl, P_dbl, c_dbl, P_dbl, P_dbl, P_dbl, P_dbl, P_dbl]

        L.elpho_calc_Sb_holstein.restype = c_dbl
        L.elpho_calc_Sb_holstein.argtypes = [c_i64, c_i64, c_dbl, P_dbl, P_dbl, P_dbl]
        L.elpho_calc_dSbdx_holstein.argtypes = [P_dbl, c_i64, c_i64, c_dbl, P_dbl, P_dbl, P_dbl]
        L.elpho_langevin_evolve_ssh.restype = c_i64
        L.elpho_langevin_evolve_ssh.argtypes = [C.c_int, C.POINTER(HmcParams), C.POINTER(HmcSsh), C.POINTER(Model), C.POINTER(KPM), P_dbl,
                                                P_dbl, c_dbl, P_dbl, P_dbl, P_dbl, P_dbl]
        L.elpho_special_move.restype = c_i64
        L.elpho_special_move.argtypes = [C.POINTER(HmcParams), C.POINTER(HmcSsh), C.POINTER(Model), C.POINTER(KPM), P_dbl, C.c_int, c_i64,
                                         c_i64, P_dbl, P_dbl, P_dbl, c_dbl, P_dbl]
        L.elpho_langevin_dSdx.restype = c_i64
        L.elpho_langevin_dSdx.argtypes = [P_dbl, C.POINTER(HmcParams), C.POINTER(Model), C.POINTER(KPM), P_dbl, P_dbl, P_dbl, P_dbl,
                                          P_dbl, P_dbl]
        L.elpho_langevin_evolve.restype = c_i64
        L.elpho_langevin_evolve.argtypes = [C.c_int, C.POINTER(HmcParams), C.POINTER(Model), C.POINTER(KPM), P_dbl, P_dbl, c_dbl,
                                            P_dbl, P_dbl, P_dbl, P_dbl]
        L.elpho_hmc_update_ssh.restype = c_i64
        L.elpho_hmc_update_ssh.argtypes = [C.POINTER(HmcParams), C.POINTER(HmcSsh), C.POINTER(Model), C.POINTER(KPM), P_dbl, P_dbl,
                                           P_dbl, P_dbl, P_dbl, P_dbl, c_dbl, P_dbl]
        L.elpho_hmc_update_holstein.restype = c_i64
        L.elpho_hmc_update_holstein.argtypes = [C.POINTER(HmcParams), C.POINTER(Model), C.POINTER(KPM), P_dbl, P_dbl, P_dbl,
                                                P_dbl, P_dbl, P_dbl, c_dbl, P_dbl]

    # ----------------------------------------------------------------- HMC
    def calc_Sb_holstein(self, N, L, dtau, x, omega, omega4):
        return float(self.lib.elpho_calc_Sb_holstein(N, L, dtau, dp(x), dp(omega), dp(omega4)))

    def calc_dSbdx_holstein(self, N, L, dtau, x, omega, omega4):
        d = np.zeros(N * L)
        self.lib.elpho_calc_dSbdx_holstein(dp(d), N, L, dtau, dp(x), dp(omega), dp(omega4))
        return d

    def hmc_update_holstein(self, m, x, v, omega, omega4, lam, lam2, mu, dtau, fa_M, dt, nt, nb, alpha, randoms, P=None,
                            tol=1e-5, maxiter=10000, kmax=1e12):
        """update!(model, hmc, fa, P): returns (accepted, x', v', dict(H0, H1, S, K, iters, flag, P_accept, kpm_calls)).
        m.E is overwritten (update_model!)."""
        hp = HmcParams()
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (omega, omega4, lam, lam2, mu, fa_M)]
        hp.N, hp.L, hp.dtau = m.N, m.L, dtau
        hp.omega, hp.omega4, hp.lam, hp.lam2, hp.mu, hp.fa_M = (dp(a) for a in arrs)
        hp.dt, hp.nt, hp.nb, hp.alpha = dt, nt, nb, alpha
        hp.solver_tol, hp.solver_maxiter, hp.kmax = tol, maxiter, kmax
        hp.kpm_n = P._n if P is not None else 0
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        v = np.ascontiguousarray(v, dtype=np.float64).copy()
        out = np.zeros(8)
        kr = randoms.get("kpm_randn")
        kr = np.ascontiguousarray(kr, dtype=np.float64) if kr is not None else None
        R, Rp, Rm = (np.ascontiguousarray(randoms[k], dtype=np.float64) for k in ("R", "Rp", "Rm"))
        acc = self.lib.elpho_hmc_update_holstein(C.byref(hp), C.byref(m), C.byref(P) if P is not None else None, dp(x), dp(v),
                                                 dp(R), dp(Rp), dp(Rm), dp(kr) if kr is not None else None, float(randoms["u"]),
                                                 dp(out))
        info = dict(H0=out[0], H1=out[1], S=out[2], K=out[3], iters=out[4], flag=int(out[5]), P_accept=out[6],
                    kpm_calls=int(out[7]))
        return bool(acc), x, v, info

    def special_move(self, m, x, kind, ci, cj, Rp, Rm, u, omega, omega4, lam, lam2, mu, dtau, P=None, kpm_randn=None, tol=1e-5,
                     maxiter=10000, kmax=1e12, ssh=None):
        """One proposed reflection (kind 0) / swap (kind 1) move of SpecialUpdates.jl on phonon columns ci, cj (0-based).
        ssh: None (Holstein) or dict(t, alpha, alpha2, phonon_to_bond, cb_perm).  -> (accepted, x', dict(S0, S1, iters, flag, P))"""
        hp, keep = self._langevin_params(m, omega, omega4, lam, lam2, mu, dtau, P, tol, maxiter, kmax)
        sp = None
        if ssh is not None:
            sp = HmcSsh()
            Nph = len(ssh["alpha"])
            fa_ = [np.ascontiguousarray(ssh[k], dtype=np.float64) for k in ("t", "alpha", "alpha2")]
            ia_ = [np.ascontiguousarray(ssh[k], dtype=np.int64) for k in ("phonon_to_bond", "cb_perm")]
            b2p = np.zeros(m.nb, dtype=np.int64)
            b2p[ia_[1][ia_[0] - 1] - 1] = np.arange(1, Nph + 1)
            sp.Nph = Nph
            sp.t, sp.alpha, sp.alpha2 = (dp(a) for a in fa_)
            sp.phonon_to_bond, sp.cb_perm, sp.bond_to_phonon_cb = ip(ia_[0]), ip(ia_[1]), ip(b2p)
            sp.primary_field = None
            keep += fa_ + ia_ + [b2p]
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        out = np.zeros(5)
        kr = np.ascontiguousarray(kpm_randn, dtype=np.float64) if kpm_randn is not None else None
        acc = self.lib.elpho_special_move(C.byref(hp), C.byref(sp) if sp is not None else None, C.byref(m),
                                          C.byref(P) if P is not None else None, dp(x), int(kind), int(ci), int(cj),
                                          dp(np.ascontiguousarray(Rp, dtype=np.float64)), dp(np.ascontiguousarray(Rm, dtype=np.float64)),
                                          dp(kr) if kr is not None else None, float(u), dp(out))
        return bool(acc), x, dict(S0=out[0], S1=out[1], iters=int(out[2]), flag=int(out[3]), P=out[4])

    def _langevin_params(self, m, omega, omega4, lam, lam2, mu, dtau, P, tol, maxiter, kmax):
        hp = HmcParams()
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (omega, omega4, lam, lam2, mu)]
        hp.N, hp.L, hp.dtau = m.N, m.L, dtau
        hp.omega, hp.omega4, hp.lam, hp.lam2, hp.mu = (dp(a) for a in arrs)
        hp.solver_tol, hp.solver_maxiter, hp.kmax = tol, maxiter, kmax
        hp.kpm_n = P._n if P is not None else 0
        return hp, arrs

    def langevin_dSdx(self, m, x, g, omega, omega4, lam, lam2, mu, dtau, P=None, b_max=None, b_min=None, tol=1e-5,
                      maxiter=10000, kmax=1e12):
        """calc_dSdx!(dSdx, g, M⁻¹g, model, P) of LangevinDynamics.jl -> (dSdx, M⁻¹g, iters); m.E must hold update_model!(x)."""
        hp, keep = self._langevin_params(m, omega, omega4, lam, lam2, mu, dtau, P, tol, maxiter, kmax)
        n = m.N * m.L
        dS, Mg, work = np.zeros(n), np.zeros(n), np.zeros(5 * n)
        it = self.lib.elpho_langevin_dSdx(dp(dS), C.byref(hp), C.byref(m), C.byref(P) if P is not None else None,
                                          dp(np.ascontiguousarray(x, dtype=np.float64)), dp(np.ascontiguousarray(g, dtype=np.float64)),
                                          dp(np.ascontiguousarray(b_max)) if b_max is not None else None,
                                          dp(np.ascontiguousarray(b_min)) if b_min is not None else None, dp(Mg), dp(work))
        return dS, Mg, int(it)

    def langevin_evolve(self, scheme, m, x, fa_Q, dt, eta, g1, g2, omega, omega4, lam, lam2, mu, dtau, P=None, kpm_randn=None,
                        tol=1e-5, maxiter=10000, kmax=1e12):
        """evolve!(model, dyn, fa, P): scheme 0 Euler, 1 Runge-Kutta, 2 Heun -> (x', iters).  m.E is overwritten."""
        hp, keep = self._langevin_params(m, omega, omega4, lam, lam2, mu, dtau, P, tol, maxiter, kmax)
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        c = lambda a: dp(np.ascontiguousarray(a, dtype=np.float64)) if a is not None else None
        fq, e, a1, a2, kr = (np.ascontiguousarray(a, dtype=np.float64) if a is not None else None for a in (fa_Q, eta, g1, g2, kpm_randn))
        it = self.lib.elpho_langevin_evolve(int(scheme), C.byref(hp), C.byref(m), C.byref(P) if P is not None else None, dp(x), dp(fq),
                                            float(dt), dp(e), dp(a1), dp(a2) if a2 is not None else dp(a1), dp(kr) if kr is not None else None)
        return x, int(it)

    def langevin_evolve_ssh(self, scheme, m, x, fa_Q, dt, eta, g1, g2, omega, omega4, mu, dtau, ssh, P=None, kpm_randn=None, tol=1e-5,
                            maxiter=10000, kmax=1e12):
        """evolve! for an SSH model: ssh = dict(t, alpha, alpha2, phonon_to_bond, cb_perm) -> (x', iters)."""
        Nph = len(ssh["alpha"])
        zeros = np.zeros(max(m.N, Nph))
        hp, keep = self._langevin_params(m, omega, omega4, zeros, zeros, mu, dtau, P, tol, maxiter, kmax)
        sp = HmcSsh()
        fa_ = [np.ascontiguousarray(ssh[k], dtype=np.float64) for k in ("t", "alpha", "alpha2")]
        ia_ = [np.ascontiguousarray(ssh[k], dtype=np.int64) for k in ("phonon_to_bond", "cb_perm")]
        b2p = np.zeros(m.nb, dtype=np.int64)
        b2p[ia_[1][ia_[0] - 1] - 1] = np.arange(1, Nph + 1)
        sp.Nph = Nph
        sp.t, sp.alpha, sp.alpha2 = (dp(a) for a in fa_)
        sp.phonon_to_bond, sp.cb_perm, sp.bond_to_phonon_cb = ip(ia_[0]), ip(ia_[1]), ip(b2p)
        pf = np.ascontiguousarray(ssh["primary_field"], dtype=np.int64) if ssh.get("primary_field") is not None else None
        sp.primary_field = ip(pf) if pf is not None else None
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        fq, e, a1, a2, kr = (np.ascontiguousarray(a, dtype=np.float64) if a is not None else None for a in (fa_Q, eta, g1, g2, kpm_randn))
        it = self.lib.elpho_langevin_evolve_ssh(int(scheme), C.byref(hp), C.byref(sp), C.byref(m), C.byref(P) if P is not None else None,
                                                dp(x), dp(fq), float(dt), dp(e), dp(a1), dp(a2) if a2 is not None else dp(a1),
                                                dp(kr) if kr is not None else None)
        return x, int(it)

    def hmc_update_ssh(self, m, x, v, omega, omega4, mu, dtau, fa_M, t, alpha, alpha2, phonon_to_bond, cb_perm, dt, nt, nb,
                       alpha_mom, randoms, P=None, tol=1e-5, maxiter=10000, kmax=1e12, primary_field=None):
        """update!(model, hmc, fa, P) for an SSH model (bond phonons): x, v, R are (Nph*L,), Rp/Rm (N*L,).  m.c, m.s, m.E are
        overwritten (update_model!).  Same return convention as hmc_update_holstein."""
        hp, sp = HmcParams(), HmcSsh()
        Nph = len(alpha)
        zeros = np.zeros(max(m.N, Nph))
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (omega, omega4, zeros, zeros, mu, fa_M, t, alpha, alpha2)]
        iarrs = [np.ascontiguousarray(a, dtype=np.int64) for a in (phonon_to_bond, cb_perm)]
        b2p = np.zeros(m.nb, dtype=np.int64)
        b2p[iarrs[1][iarrs[0] - 1] - 1] = np.arange(1, Nph + 1)
        hp.N, hp.L, hp.dtau = m.N, m.L, dtau
        hp.omega, hp.omega4, hp.lam, hp.lam2, hp.mu, hp.fa_M = (dp(a) for a in arrs[:6])
        hp.dt, hp.nt, hp.nb, hp.alpha = dt, nt, nb, alpha_mom
        hp.solver_tol, hp.solver_maxiter, hp.kmax = tol, maxiter, kmax
        hp.kpm_n = P._n if P is not None else 0
        sp.Nph = Nph
        sp.t, sp.alpha, sp.alpha2 = dp(arrs[6]), dp(arrs[7]), dp(arrs[8])
        sp.phonon_to_bond, sp.cb_perm, sp.bond_to_phonon_cb = ip(iarrs[0]), ip(iarrs[1]), ip(b2p)
        pf = np.ascontiguousarray(primary_field, dtype=np.int64) if primary_field is not None else None     # 0-based, [Nph*L]
        sp.primary_field = ip(pf) if pf is not None else None
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        v = np.ascontiguousarray(v, dtype=np.float64).copy()
        out = np.zeros(8)
        kr = randoms.get("kpm_randn")
        kr = np.ascontiguousarray(kr, dtype=np.float64) if kr is not None else None
        R, Rp, Rm = (np.ascontiguousarray(randoms[k], dtype=np.float64) for k in ("R", "Rp", "Rm"))
        acc = self.lib.elpho_hmc_update_ssh(C.byref(hp), C.byref(sp), C.byref(m), C.byref(P) if P is not None else None, dp(x),
                                            dp(v), dp(R), dp(Rp), dp(Rm), dp(kr) if kr is not None else None,
                                            float(randoms["u"]), dp(out))
        info = dict(H0=out[0], H1=out[1], S=out[2], K=out[3], iters=out[4], flag=int(out[5]), P_accept=out[6],
                    kpm_calls=int(out[7]))
        return bool(acc), x, v, info

    # ------------------------------------------------------------ geometry
    def neighbor_table(self, norbits, L1, L2, L3, bonds):
        """bonds: list of (o1, o2, (d1,d2,d3)) in deck order -> concatenated raw table (2 x nb, Fortran order
        returned as int64 array shape (nb, 2), i.e. row n = bond n = Julia column n)."""
        out = []
        ncells = L1 * L2 * L3
        for (o1, o2, d) in bonds:
            tab = np.zeros(2 * ncells, dtype=np.int64)
            dd = np.asarray(d, dtype=np.int64)
            n = self.lib.elpho_calc_neighbor_table(norbits, L1, L2, L3, o1, o2, ip(dd), 1, ip(tab))
            out.append(tab[:2 * n].reshape(n, 2).copy())
        if not out:
            return np.zeros((0, 2), dtype=np.int64)
        return np.ascontiguousarray(np.concatenate(out, axis=0))

    def holstein_initialize(self, table, t, dtau):
        """Returns (table_cb (nb,2), cosht, sinht, cb_perm, colours, ncolours)."""
        nb = table.shape[0]
        tab = np.ascontiguousarray(table.copy())
        t = np.ascontiguousarray(t, dtype=np.float64)
        c = np.zeros(nb)
        s = np.zeros(nb)
        perm = np.zeros(nb, dtype=np.int64)
        grp = np.zeros(nb, dtype=np.int64)
        ng = self.lib.elpho_holstein_initialize_model(ip(tab), nb, dp(t), dtau, dp(c), dp(s), ip(perm), ip(grp))
        return tab, c, s, perm, grp, int(ng)

    def ssh_initialize_table(self, table):
        nb = table.shape[0]
        tab = np.ascontiguousarray(table.copy())
        perm = np.zeros(nb, dtype=np.int64)
        iperm = np.zeros(nb, dtype=np.int64)
        grp = np.zeros(nb, dtype=np.int64)
        ng = self.lib.elpho_ssh_initialize_table(ip(tab), nb, ip(perm), ip(iperm), ip(grp))
        return tab, perm, iperm, grp, int(ng)

    # --------------------------------------------------------------- model
    def make_model(self, kind, N, L, table, c, s, E):
        """Bundle arrays into a Model struct; keeps references alive on the returned object."""
        m = Model()
        m.kind, m.N, m.L, m.nb = kind, N, L, table.shape[0]
        keep = dict(table=np.ascontiguousarray(table, dtype=np.int64),
                    c=np.ascontiguousarray(c, dtype=np.float64),
                    s=np.ascontiguousarray(s, dtype=np.float64),
                    E=np.ascontiguousarray(E, dtype=np.float64),
                    vp=np.zeros(N * L), vppp=np.zeros(N * L))
        m.table, m.c, m.s, m.E = ip(keep["table"]), dp(keep["c"]), dp(keep["s"]), dp(keep["E"])
        m.vp, m.vppp = dp(keep["vp"]), dp(keep["vppp"])
        m._keep = keep
        return m

    def update_model_holstein(self, N, L, dtau, x, lam, lam2, mu):
        E = np.zeros(N * L)
        self.lib.elpho_update_model_holstein(N, L, dtau, dp(x), dp(lam), dp(lam2), dp(mu), dp(E))
        return E

    def mulM(self, m, v):
        y = np.zeros_like(v)
        self.lib.elpho_mulM(dp(y), C.byref(m), dp(v))
        return y

    def mulMT(self, m, v):
        y = np.zeros_like(v)
        self.lib.elpho_mulMT(dp(y), C.byref(m), dp(v))
        return y

    def mulMTM(self, m, v):
        y = np.zeros_like(v)
        self.lib.elpho_mulMTM(dp(y), C.byref(m), dp(v))
        return y

    # ----------------------------------------------------------------- KPM
    def make_kpm(self, m, n=20, buf=0.05, c1=1.0, c2=1.0, coeff_cap=None):
        N, L, nb = m.N, m.L, m.nb
        Lo2 = (L + 1) // 2
        if coeff_cap is None:
            coeff_cap = 64 * Lo2 + 4096
        P = KPM()
        keep = dict(Ebar=np.zeros(N), cbar=np.zeros(max(nb, 1)), sbar=np.zeros(max(nb, 1)),
                    order=np.ones(Lo2, dtype=np.int64), coff=np.arange(Lo2 + 1, dtype=np.int64),
                    coeff=np.zeros(2 * coeff_cap), v1=np.zeros(2 * N * L), v2=np.zeros(2 * N * L),
                    v3=np.zeros(2 * N), v4=np.zeros(2 * N), v5=np.zeros(2 * N))
        P.active, P.N, P.L, P.nb = 1, N, L, nb
        P.table = m.table
        P.Ebar, P.cbar, P.sbar = dp(keep["Ebar"]), dp(keep["cbar"]), dp(keep["sbar"])
        P.lam_lo, P.lam_hi, P.lam_avg, P.lam_mag = 0.0, 2.0, 1.0, 1.0
        P.buf, P.c1, P.c2 = buf, c1, c2
        P.Lo2 = Lo2
        P.order, P.coff, P.coeff, P.coeff_cap = ip(keep["order"]), ip(keep["coff"]), dp(keep["coeff"]), coeff_cap
        P.v1, P.v2, P.v3, P.v4, P.v5 = (dp(keep[k]) for k in ("v1", "v2", "v3", "v4", "v5"))
        P.checkerboard_count = 0
        P._keep = keep
        P._model = m
        P._n = n
        self.lib.elpho_kpm_update_A(C.byref(P), C.byref(m))
        return P

    def kpm_setup(self, P, e_min=None, e_max=None, b_max=None, b_min=None):
        """setup!(P): either inject (e_min,e_max) or run the Arnoldi estimate from injected start vectors."""
        self.lib.elpho_kpm_update_A(C.byref(P), C.byref(P._model))
        if e_min is None:
            em = C.c_double()
            eM = C.c_double()
            self.lib.elpho_kpm_arnoldi_bounds(C.byref(P), P._n, dp(b_max), dp(b_min), C.byref(em), C.byref(eM))
            e_min, e_max = em.value, eM.value
        self.lib.elpho_kpm_setup_from_bounds(C.byref(P), e_min, e_max)
        return e_min, e_max

    def kpm_apply(self, P, v):
        out = np.zeros_like(v)
        self.lib.elpho_kpm_apply(dp(out), C.byref(P), dp(v))
        return out

    # ------------------------------------------------------------------ CG
    def cg_solve(self, m, b, x0=None, tol=1e-5, maxiter=10000, kmax=1e12, P=None, history=False):
        n = m.N * m.L
        x = np.zeros(n) if x0 is None else np.ascontiguousarray(x0, dtype=np.float64).copy()
        r, p, z = np.zeros(n), np.zeros(n), np.zeros(n)
        hist = np.full(maxiter + 1, np.nan) if history else None
        it = self.lib.elpho_cg_solve(C.byref(m), dp(x), dp(b), tol, maxiter, kmax,
                                     C.byref(P) if P is not None else None, dp(r), dp(p), dp(z), dp(hist))
        if history:
            return x, int(it), hist[:it + 1]
        return x, int(it)

    def ldiv(self, m, b, P=None, maxiter=0, solver_tol=1e-5, solver_maxiter=10000, kmax=1e12, x0=None):
        n = m.N * m.L
        x = np.zeros(n) if x0 is None else np.ascontiguousarray(x0, dtype=np.float64).copy()
        r, p, z = np.zeros(n), np.zeros(n), np.zeros(n)
        it, fl, res = C.c_int64(), C.c_int64(), C.c_double()
        self.lib.elpho_ldiv(C.byref(m), dp(x), dp(b), C.byref(P) if P is not None else None, maxiter,
                            solver_tol, solver_maxiter, kmax, dp(r), dp(p), dp(z),
                            C.byref(it), C.byref(res), C.byref(fl))
        return x, int(it.value), float(res.value), int(fl.value)
