"""Worker of the in-library sharded-solve tests (one process per rank — or, ELPH_RANKS_PER_PROC = k, k rank threads per process:
the test box admits six processes on its card, BASELINE.json names eight ranks — all on device 0; gloo carries the mailbox
handles, the barriers and the final gather — the iterations run entirely on the GPU, csrc/shard.hip).

usage: shard_worker.py <case> <out prefix>      case in {sq8, hc4, e8, C, D, E}
Every rank writes <out>.rank<r>.npz with the assembled solution and the inputs the test needs to check it."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

from elphdynamics_amd import dist, sharded, synth  # noqa: E402
from elphdynamics_amd import lattice as lat  # noqa: E402

CASES = {
    # tag: (kind, norbits, Lspatial, bonds, Ltau, dtau)
    "sq8": (0, 1, 8, lat.SQUARE_BONDS, 8, 0.1),
    "hc4": (0, 2, 4, lat.HONEYCOMB_BONDS, 6, 0.1),
    "C": (0, 1, 16, lat.SQUARE_BONDS, 160, 0.1),            # BASELINE config C
    "D": (0, 2, 12, lat.HONEYCOMB_BONDS, 120, 0.1),         # BASELINE config D: uneven slabs for 4 ranks? 12 rows / 4 = 3
    "e8": (1, 1, 8, lat.SQUARE_BONDS, 20, 0.05),
    "E": (1, 1, 16, lat.SQUARE_BONDS, 160, 0.05),           # BASELINE config E (optical SSH)
}


def run_rank(comm, case, out, tol):
    kind, norb, Ls, bonds, Ltau, dtau = CASES[case]
    la = lat.Lattice(norb, Ls, Ls, 1)
    raw = np.concatenate([la.calc_neighbor_table(o1, o2, d) for (o1, o2, d) in bonds], axis=0)
    N, nb = la.nsites, raw.shape[0]
    tvals = 1.0 + (0.1 * synth.randn(5, nb) if case in ("sq8", "hc4") else 0.0)      # small cases: every bond distinguishable
    cb = lat.initialize_checkerboard(raw, tvals * np.ones(nb), dtau)
    b = synth.randn(321, N * Ltau)
    res = dict(N=N, Ltau=Ltau, kind=kind, table=cb["table"], b=b)
    if os.environ.get("ELPH_TEST_TRANSPORT") == "collectives":
        # the same solve with torch.distributed collectives as the transport (sharded_rccl.py: RCCL between GPUs; gloo here when several ranks share the card)
        from elphdynamics_amd import sharded_rccl
        solver = sharded_rccl.CollectiveShardedSolver(comm, norb, Ls, Ls, Ltau, cb["table"], cb["cosht"], cb["sinht"], device=0)
        res.update(selftest_us=np.zeros(comm.world), selftest_slowest_us=1.0, direct=int(solver.direct))
    else:
        solver = sharded.ShardedSolver(comm, norb, Ls, Ls, Ltau, cb["table"], kind=kind, cosht=cb["cosht"], sinht=cb["sinht"], device=0)
        # the preflight of the mailbox protocol ran in the constructor (elph_shard_selftest): us per lock-step round, per peer
        res.update(selftest_us=np.asarray(solver.selftest_us), selftest_slowest_us=solver.selftest_slowest_us)
    if kind == 0:
        x = synth.phonon_field(N, Ltau, Ltau * dtau, dtau, seed=123)
        E = np.exp(-dtau * x)
        solver.update_model(E)
        res.update(E=E, c=cb["cosht"], s=cb["sinht"])
    else:
        # bond phonons on every bond: t' = t - alpha x, cosh / sinh(dtau t') per (bond, tau) (SSHModels.jl:510-562), mu = 0.1
        xb = 0.25 * synth.phonon_field(nb, Ltau, Ltau * dtau, dtau, omega=0.1, lam=0.0, seed=77).reshape(nb, Ltau)
        tp = 1.0 - 0.1 * xb
        c, s = np.cosh(dtau * tp), np.sinh(dtau * tp)
        emu = np.exp(dtau * (0.1 + 0.01 * synth.randn(9, N)))
        solver.update_model_ssh(c, s, emu)
        res.update(E=emu, c=c, s=s)
    xs, it, done = solver.solve(b, tol=tol, maxiter=20000)
    res.update(collectives=getattr(solver, "collectives", 0))
    res.update(x=xs, it=it, done=done, eps=solver.eps, halo=np.array([solver.sl["lo"], solver.sl["hi"]]), rows=np.array([s_["R"] for s_ in solver.slabs.slabs]))
    # a second solve on the same handle (mailbox re-zeroed, new barrier): same bits
    xs2, it2, done2 = solver.solve(b, tol=tol, maxiter=20000)
    res.update(x2=xs2, it2=it2)
    if os.environ.get("ELPH_TEST_KPM") == "1":
        # KPM-preconditioned solve under sharding: same Arnoldi start vectors on every rank (and in the un-sharded check of the test)
        act, lo, hi = solver.setup_kpm(E if kind == 0 else (c, s, emu), n=20, buf=0.05, c1=1.0, c2=1.0, seed=7)
        xk, itk, donek = solver.solve(b, tol=tol, maxiter=20000, precond=True)
        res.update(xk=xk, itk=itk, donek=donek, kpm_active=act, lam_lo=lo, lam_hi=hi)
        if os.environ.get("ELPH_TEST_TIMING") == "1":      # (tools/time_shard_kpm.sh: wall time of a second, warm solve)
            import time
            comm.barrier()
            xk2, itk2, _ = solver.solve(b, tol=tol, maxiter=20000, precond=True)
            dt = comm.max(solver.last_solve_s)
            xs3, it3, _ = solver.solve(b, tol=tol, maxiter=20000)
            dt3 = comm.max(solver.last_solve_s)
            if comm.rank == 0:
                print(f"TIMING {case} world={comm.world}: KPM solve {1e3*dt:.2f} ms / {itk2} it = {1e6*dt/max(itk2,1):.1f} us per iteration (library call, host pointers in and out); "
                      f"plain {1e3*dt3:.2f} ms / {it3} it = {1e6*dt3/max(it3,1):.1f} us per iteration", flush=True)
    solver.close()
    np.savez(out + f".rank{comm.rank}", **res)
    comm.close()


def main():
    case, out = sys.argv[1], sys.argv[2]
    tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-9
    per_proc = int(os.environ.get("ELPH_RANKS_PER_PROC", "1"))
    comm = dist.Comm(backend=os.environ.get("ELPH_TEST_BACKEND", "gloo"))
    if per_proc == 1:
        run_rank(comm, case, out, tol)
    else:
        dist.HybridComm.spawn(comm, per_proc, lambda c: run_rank(c, case, out, tol))
        comm.close()


if __name__ == "__main__":
    main()
