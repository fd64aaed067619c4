"""Worker of the tests of the sharded CALLERS (elph_shard_ldiv, elph_shard_fermion_force_holstein / _ssh, elph_hmc_update on a sharded
handle): one process per rank (or ELPH_RANKS_PER_PROC rank threads per process), all on device 0, gloo for the host collectives.
Rank 0 also computes the same quantity with ONE handle on the whole lattice; every rank writes <out>.rank<r>.npz.

usage: shard_callers_worker.py <what> <config> <out prefix>     what in {ldiv, force, hmc}"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

from elphdynamics_amd import _lib, configs, dist, hmc, models, preconditioners as pc, sharded, synth  # noqa: E402
from elphdynamics_amd import lattice as lat  # noqa: E402


def geometry(tag):
    kind, norb, Ls, bonds, beta, dtau = configs.CONFIGS[tag]
    return kind, norb, Ls, lat.ltau_from_beta(beta, dtau), dtau


def run_rank(comm, what, tag, out):
    lib = _lib.load()
    kind, norb, Ls, Ltau, dtau = geometry(tag)
    # ---- ONE handle on the whole lattice first: the inputs (host arrays) on every rank, the reference on rank 0 — and closed before the
    # ---- sharded handles exist (rank threads of one process share the process's hardware queues: a kernel that waits for a peer must not
    # ---- sit behind an idle handle's stream on the same queue)
    m = configs.make_model(tag, tol=1e-12, maxiter=20000)
    res = dict(world=comm.world)
    ref0 = comm.rank == 0
    if kind == "holstein":
        m.lam2[:] = 0.03 * synth.randn(5, m.Nsites)
    else:
        m.alpha2[:] = 0.02
    models.update_model_(m)
    table, Nb, Ndim, Ndof = np.array(m.neighbor_table), m.Nbonds, m.Ndim, m.Ndof
    if kind == "holstein":
        x0, lam, lam2, mu, cosht, sinht = m.x.copy(), m.lam.copy(), m.lam2.copy(), m.mu.copy(), np.array(m.cosht), np.array(m.sinht)
        E = np.exp(-dtau * (np.repeat(lam, Ltau) * x0 + np.repeat(lam2, Ltau) * x0 ** 2 - np.repeat(mu, Ltau)))
    else:
        c = np.ascontiguousarray(m.cosht).reshape(Nb, Ltau).copy()
        s = np.ascontiguousarray(m.sinht).reshape(Nb, Ltau).copy()
        emu = np.array(m.expDtauMu)
    b = synth.randn(331, Ndim)
    pp, pm = synth.randn(41, Ndim), synth.randn(42, Ndim)
    bp, bm = synth.randn(61, Ndim), synth.randn(62, Ndim)
    dt, nt, nb = 0.05, 2, int(os.environ.get("ELPH_TEST_NB", "1"))
    rnd = dict(R=synth.randn(500, Ndof), Rp=synth.randn(501, Ndim), Rm=synth.randn(502, Ndim), u=0.0)
    with_kpm = os.environ.get("ELPH_TEST_KPM") == "1"
    if with_kpm:
        rnd["kpm_randn"] = synth.randn(503, (nt + 2) * 2 * m.Nsites)
    if what == "hmc":
        m.solver.tol = 1e-10
        if kind != "holstein" and getattr(m, "omega4", None) is None:
            m.omega4 = np.zeros(m.Nph)
        m.omega4[:] = 0.02
        fa = pc.FourierAccelerator(m)
        pc.update_M_(fa, m, 0.0, np.inf, 1.0, 0.3)
        omega, omega4, faM = m.omega.copy(), m.omega4.copy(), np.array(fa.M)
        if kind != "holstein":
            x0 = m.x.copy()
            ssh = dict(cbperm=np.array(m.checkerboard_perm), p2b=np.array(m.phonon_to_bond), t=np.array(m.t), alpha=np.array(m.alpha),
                       alpha2=np.array(m.alpha2), t_bare_cb=np.array(m.t_bare_cb), mu=np.array(m.mu), Nph=m.Nph)
    if ref0 and what == "ldiv":
        xr = np.zeros(Ndim)
        itr, rsr, flr = models.ldiv_(xr, m, np.ascontiguousarray(b))
        res.update(x_ref=xr, it_ref=itr, resid_ref=rsr, flag_ref=flr)
    elif ref0 and what == "force" and kind == "holstein":
        Fr = np.zeros(Ndim)
        itr, flr, Xpr, Xmr = hmc.calc_dSfdx_(Fr, m, pp, pm, None, power=1.0, return_solutions=True)
        res.update(F_ref=Fr, Xp_ref=Xpr, Xm_ref=Xmr, it_ref=itr, flag_ref=flr)
    elif ref0 and what == "force":
        qr = np.zeros(Nb * Ltau)
        itr, flr = C.c_int64(), C.c_int()
        m._push_solver()
        _lib.check(lib.elph_fermion_force_ssh(m._h, _lib.dptr(bp), _lib.dptr(bm), 0, 1.0, _lib.dptr(qr), None, None, C.byref(itr), C.byref(flr)))
        res.update(q_ref=qr.reshape(Nb, Ltau), it_ref=itr.value, flag_ref=flr.value)
    elif ref0 and what == "hmc":
        H = hmc.HybridMonteCarlo(m, fa, dt, nt * dt, alpha=0.0, Nb=nb)
        Pk = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0) if with_kpm else None
        a, i = hmc.update_(m, H, fa, Pk, randoms=rnd)
        res.update(accepted_ref=int(a), iters_ref=i, flag_ref=H.flag, energies_ref=np.array([H.H0, H.H1, H.S, H.K, H.P_accept]), x_ref=m.x.copy(), v_ref=H.v.copy())
    m.close()
    comm.barrier()
    # ---- the same over the ranks -------------------------------------------------------------------------------------------------
    if kind == "holstein":
        S = sharded.ShardedSolver(comm, norb, Ls, Ls, Ltau, table, kind=0, cosht=cosht, sinht=sinht, device=0)
        S.update_model(E)
    else:
        S = sharded.ShardedSolver(comm, norb, Ls, Ls, Ltau, table, kind=1, device=0)
        S.update_model_ssh(c, s, emu)
    S.set_solver(1e-12, 20000)
    if what == "ldiv":
        x, it, rs, fl = S.ldiv(b)
        res.update(x=x, it=it, resid=rs, flag=fl)
        # a solve cut short: flag 1 (hit maxiter), x zero-filled, the same on every rank — Models.jl:157-180
        S.set_solver(1e-12, 5)
        x5, it5, rs5, fl5 = S.ldiv(b)
        res.update(x5=x5, it5=it5, flag5=fl5, resid5=rs5)
        # ... and with a call-level maxiter below the solver's: flag 2 (Models.jl:160 compares solver.maxiter)
        S.set_solver(1e-12, 20000)
        x6, it6, rs6, fl6 = S.ldiv(b, maxiter=5)
        res.update(it6=it6, flag6=fl6, nz6=float(np.abs(x6).sum()))
    elif what == "force" and kind == "holstein":
        F, Xp, Xm, it, fl = S.fermion_force_holstein(x0, lam, lam2, mu, dtau, pp, pm)
        res.update(F=F, Xp=Xp, Xm=Xm, it=it, flag=fl)
    elif what == "force":
        q, it, fl = S.fermion_force_ssh(bp, bm, Nb)
        res.update(q=q, it=it, flag=fl)
    elif what == "hmc" and kind == "holstein":
        S.set_solver(1e-10, 20000)
        d = _lib.dptr
        site = lambda a: np.ascontiguousarray(np.asarray(a)[S.gsites])       # noqa: E731
        S.install_collectives()
        # the sharded update: the slab's part of every global array
        _lib.check(lib.elph_hmc_create(S.h, d(site(omega)), d(site(omega4)), d(site(lam)), d(site(lam2)), d(site(mu)), dtau, d(S._local(faM))))
        _lib.check(lib.elph_hmc_set_state(S.h, d(S._local(x0)), d(np.zeros(S.Nloc * Ltau))))
        if with_kpm:      # the expansion lives on a handle of the whole lattice (created by setup_kpm; its Ē is injected at every setup!(P))
            S.setup_kpm(E, n=20, buf=0.05, c1=1.0, c2=1.0, seed=1)
            _lib.check(lib.elph_shard_set_full_lattice(S.h, S.hf))
        acc, fl, its, en = C.c_int(), C.c_int(), C.c_double(), np.zeros(5)
        _lib.check(lib.elph_hmc_update(S.h, dt, nt, nb, 0.0, 1 if with_kpm else 0, d(S._local(rnd["R"])), d(S._local(rnd["Rp"])), d(S._local(rnd["Rm"])),
                                       d(rnd["kpm_randn"]) if with_kpm else None, rnd["u"], C.byref(acc), C.byref(its), d(en), C.byref(fl)))
        xs, vs = np.zeros(S.Nloc * Ltau), np.zeros(S.Nloc * Ltau)
        _lib.check(lib.elph_hmc_get_state(S.h, d(xs), d(vs)))
        res.update(accepted=acc.value, flag=fl.value, iters=its.value, energies=en, x=S._gather_own(xs), v=S._gather_own(vs))
    elif what == "hmc":
        # bond phonons (BASELINE config 5): the slab's phonon columns = the phonons of its bonds; per-phonon arrays in that order
        S.set_solver(1e-10, 20000)
        d, ip = _lib.dptr, _lib.iptr
        S.install_collectives()
        gcol, cbl, wown = S.ssh_phonon_columns(ssh["cbperm"], ssh["p2b"])
        nphl = len(gcol)
        col = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64)[gcol])                                   # noqa: E731
        fld = lambda a: np.ascontiguousarray(np.asarray(a).reshape(ssh["Nph"], Ltau)[gcol]).reshape(-1)               # noqa: E731
        t_ph = np.ascontiguousarray(ssh["t"][ssh["p2b"] - 1][gcol])
        t_bare_loc = np.ascontiguousarray(ssh["t_bare_cb"][np.asarray(S.bonds)])
        _lib.check(lib.elph_hmc_create_ssh(S.h, nphl, d(col(omega)), d(col(omega4)), ip(np.ascontiguousarray(cbl)), d(t_ph), d(col(ssh["alpha"])),
                                           d(col(ssh["alpha2"])), d(t_bare_loc), d(np.ascontiguousarray(ssh["mu"][S.gsites])), dtau, d(fld(faM))))
        _lib.check(lib.elph_shard_hmc_set_columns(S.h, ip(np.ascontiguousarray(gcol)), ssh["Nph"], d(wown)))
        _lib.check(lib.elph_hmc_set_state(S.h, d(fld(x0)), d(np.zeros(nphl * Ltau))))
        if with_kpm:      # the expansion on a bond-phonon handle of the whole lattice: exp(Δτμ) once, the τ-means of the hoppings injected at every setup!(P)
            S.setup_kpm((c, s, emu), n=20, buf=0.05, c1=1.0, c2=1.0, seed=1)
            _lib.check(lib.elph_shard_set_full_lattice(S.h, S.hf))
            S.set_bonds(Nb)
        acc, fl, its, en = C.c_int(), C.c_int(), C.c_double(), np.zeros(5)
        _lib.check(lib.elph_hmc_update(S.h, dt, nt, nb, 0.0, 1 if with_kpm else 0, d(fld(rnd["R"])), d(S._local(rnd["Rp"])), d(S._local(rnd["Rm"])),
                                       d(rnd["kpm_randn"]) if with_kpm else None, rnd["u"], C.byref(acc), C.byref(its), d(en), C.byref(fl)))
        xs, vs = np.zeros(nphl * Ltau), np.zeros(nphl * Ltau)
        _lib.check(lib.elph_hmc_get_state(S.h, d(xs), d(vs)))
        # the owned columns of every rank -> the field on the whole lattice
        mine = (gcol[wown == 1.0], xs.reshape(nphl, Ltau)[wown == 1.0], vs.reshape(nphl, Ltau)[wown == 1.0])
        parts = comm.allgather_object(mine) if comm.world > 1 else [mine]
        xg, vg = np.full((ssh["Nph"], Ltau), np.nan), np.full((ssh["Nph"], Ltau), np.nan)
        for g_, x_, v_ in parts:
            assert np.isnan(xg[g_]).all(), "a phonon with two owners"
            xg[g_], vg[g_] = x_, v_
        assert not np.isnan(xg).any(), "a phonon without an owner"
        res.update(accepted=acc.value, flag=fl.value, iters=its.value, energies=en, x=xg.reshape(-1), v=vg.reshape(-1))
    if what == "hmc":
        gd, gh = S.ghost_stats()
        res.update(ghost_dev=gd, ghost_host=gh)
    S.close()
    np.savez(out + f".rank{comm.rank}", **res)
    comm.close()


def main():
    what, tag, out = sys.argv[1], sys.argv[2], sys.argv[3]
    per_proc = int(os.environ.get("ELPH_RANKS_PER_PROC", "1"))
    comm = dist.Comm(backend="gloo")
    if per_proc == 1:
        run_rank(comm, what, tag, out)
    else:
        dist.HybridComm.spawn(comm, per_proc, lambda c: run_rank(c, what, tag, out))
        comm.close()


if __name__ == "__main__":
    main()
