"""The in-library sharded solve (include/elph_gpu.h: elph_shard_*; csrc/shard.hip, SHARD form of csrc/cg_wg.hip): ONE CG solve
over 2 and 4 ranks (processes) sharing the test box's one GPU — the same device code as between GPUs: device-initiated
stores into hipIpc-mapped mailboxes, no collective, no host in the iteration — against the oracle's un-sharded solve and
against the un-sharded GPU handle, on small lattices and on BASELINE configs C, D (honeycomb L = 12, Nτ = 120) and E (optical
SSH L = 16, Nτ = 160: the per-(τ, bond) hopping tables sharded by bond owner)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(case, tmp_path, world, tol=1e-9, kpm=False, per_proc=1, pin_timeout=True):
    """world ranks as world / per_proc processes of per_proc rank threads each (the box admits six processes on its card: eight
    ranks run as four processes of two)."""
    port = _free_port()
    out = str(tmp_path / f"shard_{case}_{world}")
    procs = []
    assert world % per_proc == 0
    nproc = world // per_proc
    for r in range(nproc):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), ELPH_FORCE_DEVICE="0", ELPH_TEST_KPM="1" if kpm else "0", ELPH_RANKS_PER_PROC=str(per_proc))
        if pin_timeout:
            env["ELPH_WG_TIMEOUT_MS"] = "60000"
        else:          # the library's own wait bound of a sharded solve (ELPH_SHARD_TIMEOUT_MS / 20 s; there is no fallback behind it)
            env.pop("ELPH_WG_TIMEOUT_MS", None)
            env.pop("ELPH_SHARD_TIMEOUT_MS", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), case, out, repr(tol)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    errs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        errs.append((p.returncode, e[-3000:]))
    assert all(rc == 0 for rc, _ in errs), errs
    return [np.load(out + f".rank{r}.npz") for r in range(world)]


def _oracle_model(oracle, a):
    kind = int(a["kind"])
    if kind == 0:
        return oracle.make_model(0, int(a["N"]), int(a["Ltau"]), a["table"], a["c"], a["s"], np.ascontiguousarray(a["E"]))
    return oracle.make_model(1, int(a["N"]), int(a["Ltau"]), a["table"], np.ascontiguousarray(a["c"]).reshape(-1),
                             np.ascontiguousarray(a["s"]).reshape(-1), np.ascontiguousarray(a["E"]))


def _check(res, oracle, tol):
    a = res[0]
    for b in res[1:]:
        assert int(a["it"]) == int(b["it"]) and int(b["done"]) == 1
        assert np.array_equal(a["x"], b["x"])                                     # every rank assembled the same solution
    assert int(a["done"]) == 1 and float(a["eps"]) < tol
    assert np.array_equal(a["x"], a["x2"]) and int(a["it"]) == int(a["it2"])      # a repeated solve gives the same bits
    om = _oracle_model(oracle, a)
    bb = np.ascontiguousarray(a["b"])
    xo, ito = oracle.cg_solve(om, bb, tol=tol, maxiter=20000)
    assert abs(int(a["it"]) - ito) <= max(2, ito // 100), (int(a["it"]), ito)
    err = np.linalg.norm(a["x"] - xo) / np.linalg.norm(xo)
    r = oracle.mulMTM(om, np.ascontiguousarray(a["x"])) - bb
    return err, np.linalg.norm(r) / np.linalg.norm(bb)


@pytest.mark.parametrize("case,world,halo", [("sq8", 2, (2, 2)), ("hc4", 2, (1, 1)), ("sq8", 4, (2, 2)), ("hc4", 4, (1, 1)), ("e8", 2, (2, 2)), ("e8", 4, (2, 2))])
def test_sharded_solve_small_lattices(tmp_path, oracle, case, world, halo):
    res = _run(case, tmp_path, world, tol=1e-9)
    assert tuple(res[0]["halo"].tolist()) == halo
    err, rres = _check(res, oracle, 1e-9)
    assert err < 1e-7 and rres < 1e-8


@pytest.mark.parametrize("case,world,backend,tol", [("sq8", 2, "gloo", 1e-9), ("hc4", 2, "gloo", 1e-9), ("hc4", 4, "gloo", 1e-9), ("C", 2, "gloo", 1e-13), ("D", 4, "gloo", 1e-13),
                                                    ("C", 1, "nccl", 1e-13)])
def test_sharded_solve_with_collectives_as_transport(tmp_path, oracle, monkeypatch, case, world, backend, tol):
    """sharded_rccl.CollectiveShardedSolver on the GPU: the library's un-modified mat-vec on the slab (device pointers, torch's stream), two
    all-reduces and one grouped ghost-row exchange per iteration through torch.distributed — RCCL ("nccl") where every rank has a GPU of its
    own, here gloo with the ranks sharing the box's one card (messages staged through the host; RCCL refuses two ranks on one device) and RCCL
    itself at world size 1 (all-reduces of a one-rank group: ELPH_DIST_FORCE_INIT).  Against the oracle's un-sharded solve: iteration count,
    solution (BASELINE configs C and D at tol 1e-13: the north_star's 1e-10), true residual; every rank the same bits; a repeated solve the
    same bits.  UNMEASURED ON HARDWARE between two devices, like the mailbox transport."""
    monkeypatch.setenv("ELPH_TEST_TRANSPORT", "collectives")
    monkeypatch.setenv("ELPH_TEST_BACKEND", backend)
    if world == 1:
        monkeypatch.setenv("ELPH_DIST_FORCE_INIT", "1")
    res = _run(case, tmp_path, world, tol=tol)
    err, rres = _check(res, oracle, tol)
    a = res[0]
    assert int(a["direct"]) == (1 if backend == "nccl" else 0)
    launched = (int(a["it"]) + 7) // 8 * 8                       # the host looks at the done flag every 8 iterations
    per_it = 3 if world > 1 else 2                               # two all-reduces + one grouped exchange (no neighbours at world 1)
    assert int(a["collectives"]) == 1 + per_it * launched
    if tol < 1e-12:
        assert err < 1e-10 and rres < 1e-11, (err, rres)
    else:
        assert err < 1e-7 and rres < 1e-8, (err, rres)


@pytest.mark.parametrize("case,world", [("C", 2), ("C", 4), ("D", 2), ("D", 4), ("E", 2), ("E", 4), ("C", 8), ("D", 8), ("E", 8)])
def test_sharded_solve_baseline_configs(tmp_path, oracle, case, world):
    """Configs D and E (and C) at full size, sharded over 2, 4 and 8 ranks (BASELINE.json: "1, 2, 4 and 8 GPUs"), solved to 1e-13 on
    both sides: the sharded solution is within the north_star's 1e-10 of the oracle's un-sharded one.  Eight ranks run as four
    processes of two rank threads (the box admits six processes on its card): 8 x 20 = 160 records per meeting at Ltau = 160,
    config D in slabs of 1,2,1,2,... rows."""
    # (config C runs at the library's DEFAULT wait bound — the others pin a long one: a sharded time-out has no fallback)
    res = _run(case, tmp_path, world, tol=1e-13, per_proc=2 if world == 8 else 1, pin_timeout=(case != "C"))
    for a in res:      # the preflight (elph_shard_selftest, in the solver's constructor) saw every peer: finite times, own entry included
        assert a["selftest_us"].shape == (world,) and np.all(np.isfinite(a["selftest_us"])) and np.all(a["selftest_us"] >= 0.0)
        assert 0.0 < float(a["selftest_slowest_us"]) < 5e6
    if world == 8:
        assert int(res[0]["rows"].sum()) == (12 if case == "D" else 16) and len(res[0]["rows"]) == 8
    err, rres = _check(res, oracle, 1e-13)
    assert err < 1e-10, err
    assert rres < 1e-11


def test_sharded_solve_one_rank_equals_the_unsharded_handle(tmp_path):
    """world = 1: the SHARD form of the kernel (x0 = 0 seeded in-kernel, records through the mailbox) against elph_ldiv of an ordinary
    handle on the same lattice: same iteration count, solutions equal to the solver tolerance."""
    import ctypes as C
    from elphdynamics_amd import _lib
    lib = _lib.load()
    a = _run("sq8", tmp_path, 1, tol=1e-9)[0]
    assert int(a["done"]) == 1
    N, L = int(a["N"]), int(a["Ltau"])
    h = _lib.Handle()
    tab = np.ascontiguousarray(a["table"], dtype=np.int64)
    _lib.check(lib.elph_create(C.byref(h), 0, N, L, tab.shape[0], _lib.iptr(tab), _lib.dptr(np.ascontiguousarray(a["c"])),
                               _lib.dptr(np.ascontiguousarray(a["s"])), 0))
    try:
        _lib.check(lib.elph_set_expV(h, _lib.dptr(np.ascontiguousarray(a["E"]))))
        _lib.check(lib.elph_solver_set(h, 1e-9, 20000, 1e12))
        x = np.zeros(N * L)
        it, fl, res = C.c_int64(), C.c_int(), C.c_double()
        _lib.check(lib.elph_ldiv(h, _lib.dptr(x), _lib.dptr(np.ascontiguousarray(a["b"])), 0, 0, C.byref(it), C.byref(res), C.byref(fl)))
        assert fl.value == 0 and abs(it.value - int(a["it"])) <= 1
        assert np.linalg.norm(x - a["x"]) / np.linalg.norm(x) < 1e-7
    finally:
        lib.elph_destroy(h)


@pytest.mark.parametrize("case,world", [("sq8", 2), ("C", 2), ("D", 2), ("D", 4), ("e8", 2), ("E", 2), ("C", 8), ("E", 8)])
def test_sharded_kpm_preconditioned_solve(tmp_path, case, world):
    """SURVEY 8e 'KPM under sharding': the preconditioned solve over 2 / 4 ranks against the preconditioned solve of ONE handle on the
    whole lattice with the same Arnoldi start vectors: same expansion (bounds), same iteration count (+-1 at the knife edge), and —
    both solved to 1e-13 — solutions within 1e-10."""
    import ctypes as C
    from elphdynamics_amd import _lib
    lib = _lib.load()
    tol = 1e-13
    res = _run(case, tmp_path, world, tol=tol, kpm=True, per_proc=2 if world == 8 else 1)
    a = res[0]
    for b in res[1:]:
        assert int(a["itk"]) == int(b["itk"]) and np.array_equal(a["xk"], b["xk"])
    assert int(a["donek"]) == 1 and int(a["kpm_active"]) == 1
    N, L, kind = int(a["N"]), int(a["Ltau"]), int(a["kind"])
    h = _lib.Handle()
    tab = np.ascontiguousarray(a["table"], dtype=np.int64)
    if kind == 0:
        _lib.check(lib.elph_create(C.byref(h), 0, N, L, tab.shape[0], _lib.iptr(tab), _lib.dptr(np.ascontiguousarray(a["c"])),
                                   _lib.dptr(np.ascontiguousarray(a["s"])), 0))
    else:
        _lib.check(lib.elph_create(C.byref(h), 1, N, L, tab.shape[0], _lib.iptr(tab), None, None, 0))
    try:
        if kind == 0:
            _lib.check(lib.elph_set_expV(h, _lib.dptr(np.ascontiguousarray(a["E"]))))
        else:       # bond phonons: per-(bond, tau) tables of the whole lattice, tau-means inside the expansion (KPMPreconditioners.jl:355-381)
            _lib.check(lib.elph_update_model_ssh(h, _lib.dptr(np.ascontiguousarray(a["c"]).reshape(-1)),
                                                 _lib.dptr(np.ascontiguousarray(a["s"]).reshape(-1)), _lib.dptr(np.ascontiguousarray(a["E"]))))
        _lib.check(lib.elph_kpm_create(h, 20, 0.05, 1.0, 1.0))
        rng = np.random.default_rng(7)
        bmax, bmin = rng.standard_normal(N), rng.standard_normal(N)
        act, lo, hi = C.c_int(), C.c_double(), C.c_double()
        _lib.check(lib.elph_kpm_setup(h, _lib.dptr(bmax), _lib.dptr(bmin), float("nan"), float("nan"), C.byref(act), C.byref(lo), C.byref(hi)))
        assert act.value == 1 and lo.value == float(a["lam_lo"]) and hi.value == float(a["lam_hi"])
        _lib.check(lib.elph_solver_set(h, tol, 20000, 1e12))
        x = np.zeros(N * L)
        it, fl, rs = C.c_int64(), C.c_int(), C.c_double()
        _lib.check(lib.elph_ldiv(h, _lib.dptr(x), _lib.dptr(np.ascontiguousarray(a["b"])), 1, 0, C.byref(it), C.byref(rs), C.byref(fl)))
        assert fl.value == 0
        assert abs(it.value - int(a["itk"])) <= 1, (it.value, int(a["itk"]))
        assert int(a["itk"]) < int(a["it"])                                      # the preconditioner does its job under sharding too
        err = np.linalg.norm(a["xk"] - x) / np.linalg.norm(x)
        assert err < 1e-10, err
        assert np.linalg.norm(a["xk"] - a["x"]) / np.linalg.norm(a["x"]) < 1e-10  # and agrees with the sharded un-preconditioned solve
    finally:
        lib.elph_destroy(h)


def test_shard_selftest_names_a_silent_rank(tmp_path):
    """elph_shard_selftest with a peer that never answers (rank 1 of 2 skips the call): rank 0 gets ELPH_E_HIP naming rank 1 within the
    bound (ELPH_SHARD_SELFTEST_MS), not a hang and not a time-out inside a solve."""
    code = (
        "import os, sys, ctypes as C\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import numpy as np\n"
        "from elphdynamics_amd import dist, sharded, _lib, lattice as lat\n"
        "comm = dist.Comm(backend='gloo')\n"
        "la = lat.Lattice(1, 8, 8, 1)\n"
        "raw = np.concatenate([la.calc_neighbor_table(o1, o2, d) for (o1, o2, d) in lat.SQUARE_BONDS], axis=0)\n"
        "cb = lat.initialize_checkerboard(raw, np.ones(raw.shape[0]), 0.1)\n"
        "s = sharded.ShardedSolver(comm, 1, 8, 8, 8, cb['table'], kind=0, cosht=cb['cosht'], sinht=cb['sinht'], device=0, selftest=False)\n"
        "lib = _lib.load()\n"
        "_lib.check(lib.elph_shard_prepare(s.h)); comm.barrier()\n"
        "if comm.rank == 0:\n"
        "    us = np.zeros(2); w = C.c_double()\n"
        "    rc = lib.elph_shard_selftest(s.h, 8, _lib.dptr(us), C.byref(w))\n"
        "    print('RC', rc, lib.elph_last_error().decode(), flush=True)\n"
        "comm.barrier(); s.close(); comm.close()\n")
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ELPH_FORCE_DEVICE="0", ELPH_SHARD_SELFTEST_MS="300")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [e[-2000:] for _, e in outs]
    line = [l for l in outs[0][0].splitlines() if l.startswith("RC")][0]
    assert line.startswith("RC -2") and "rank(s) 1" in line and "300 ms" in line, line


# ---------------------------------------------------------------------------------------------------------------------------------
# The CALLERS of the solve on a sharded lattice (BASELINE configs 4 and 5: "HMC ... spatial-sharded"): ldiv!'s wrapper, the fermion force,
# one HMC update — several ranks on the box's one GPU against ONE handle on the whole lattice.
# ---------------------------------------------------------------------------------------------------------------------------------

def _run_callers(what, tag, tmp_path, world, per_proc=1, extra_env=None):
    port = _free_port()
    out = str(tmp_path / f"callers_{what}_{tag}_{world}")
    nproc = world // per_proc
    procs = []
    for r in range(nproc):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ELPH_FORCE_DEVICE="0", ELPH_RANKS_PER_PROC=str(per_proc), ELPH_WG_TIMEOUT_MS="60000")
        if per_proc > 1:      # rank threads of one process: every rank's stream on a hardware queue of its own (kernels that wait for each other)
            env["GPU_MAX_HW_QUEUES"] = "8"
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_callers_worker.py"), what, tag, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    errs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        errs.append((p.returncode, e[-3000:]))
    assert all(rc == 0 for rc, _ in errs), errs
    return [np.load(out + f".rank{r}.npz") for r in range(world)]


def _rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


@pytest.mark.parametrize("tag,world", [("D", 2), ("D", 4), ("E", 2)])
def test_sharded_ldiv_has_the_reference_semantics(tmp_path, tag, world):
    """elph_shard_ldiv = ldiv!(x, model, b) over the ranks (Models.jl:139-186): iteration count, true residual and flag identical on every
    rank and equal to ONE handle's; flag 1 (hit maxiter) / 2 (false convergence) with x zero-filled everywhere."""
    res = _run_callers("ldiv", tag, tmp_path, world)
    a = res[0]
    for b in res[1:]:
        assert int(b["it"]) == int(a["it"]) and float(b["resid"]) == float(a["resid"]) and int(b["flag"]) == int(a["flag"]) and np.array_equal(a["x"], b["x"])
        assert int(b["flag5"]) == int(a["flag5"]) and int(b["it5"]) == int(a["it5"])
    assert int(a["flag"]) == 0 and int(a["flag_ref"]) == 0 and abs(int(a["it"]) - int(a["it_ref"])) <= 2
    assert _rel(a["x"], a["x_ref"]) < 1e-10
    assert float(a["resid"]) < 1e-10 and abs(float(a["resid"]) - float(a["resid_ref"])) < 1e-11
    assert int(a["it5"]) == 5 and int(a["flag5"]) == 1 and not a["x5"].any() and float(a["resid5"]) > 1e-3
    assert int(a["it6"]) == 5 and int(a["flag6"]) == 2 and float(a["nz6"]) == 0.0


@pytest.mark.parametrize("tag,world,per_proc", [("D", 2, 1), ("D", 4, 1), ("D", 8, 2), ("E", 2, 1), ("E", 4, 1), ("E", 8, 2)])
def test_sharded_fermion_force_vs_one_handle(tmp_path, tag, world, per_proc):
    """BASELINE configs 4 and 5: the fermion force of the Holstein honeycomb lattice (D) and the bond brackets of the optical SSH model (E)
    sharded over 2 / 4 / 8 ranks reproduce elph_fermion_force_* of ONE handle to 1e-10 (solves to 1e-12 on both sides); iters and flag
    identical on every rank."""
    res = _run_callers("force", tag, tmp_path, world, per_proc=per_proc)
    a = res[0]
    key = "F" if tag == "D" else "q"
    for b in res[1:]:
        assert int(b["it"]) == int(a["it"]) and int(b["flag"]) == 0 and np.array_equal(a[key], b[key])
    assert int(a["flag"]) == 0 and int(a["flag_ref"]) == 0 and abs(int(a["it"]) - int(a["it_ref"])) <= 2
    assert _rel(a[key], a[key + "_ref"]) < 1e-10, _rel(a[key], a[key + "_ref"])
    if tag == "D":
        assert _rel(a["Xp"], a["Xp_ref"]) < 1e-10 and _rel(a["Xm"], a["Xm_ref"]) < 1e-10


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_hmc_update_bond_phonons_vs_one_handle(tmp_path, world):
    """BASELINE config 5 (optical SSH, L = 16, Ntau = 160): one HMC update on a lattice sharded over 2 / 4 ranks — the slab's phonon columns are
    the phonons of its bonds (ghost bonds included), update_model! of the hoppings on the device per slab, the bond-bracket force exact on
    the owner of a bond and handed to the other holders, S_b and K over owned columns — against the update of ONE handle."""
    res = _run_callers("hmc", "E", tmp_path, world, per_proc=2 if world == 8 else 1)      # (eight ranks: four processes of two rank threads)
    a = res[0]
    for b in res[1:]:
        assert int(b["accepted"]) == int(a["accepted"]) and np.array_equal(a["energies"], b["energies"]) and np.array_equal(a["x"], b["x"])
    assert int(a["accepted"]) == int(a["accepted_ref"]) == 1 and int(a["flag"]) == int(a["flag_ref"]) == 0
    e, er = a["energies"], a["energies_ref"]
    assert abs(e[0] - er[0]) < 1e-9 * abs(er[0]) and abs(e[1] - er[1]) < 1e-8 * abs(er[1])
    assert abs(e[2] - er[2]) < 1e-8 * abs(er[2]) and abs(e[3] - er[3]) < 1e-8 * abs(er[3])
    assert _rel(a["x"], a["x_ref"]) < 1e-9 and _rel(a["v"], a["v_ref"]) < 1e-8
    assert abs(float(a["iters"]) - float(a["iters_ref"])) <= 1


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_hmc_update_with_the_kpm_preconditioner(tmp_path, world):
    """The production shape of BASELINE config 4: every force / action evaluation of the sharded update solves with the KPM preconditioner —
    the expansion on a handle of the whole lattice, set up at every setup!(P) from the SAME Arnoldi start vectors on every rank and the
    τ-averaged exp(−ΔτV) summed over the ranks' own rows (elph_shard_set_full_lattice) — against the preconditioned update of ONE handle."""
    res = _run_callers("hmc", "D", tmp_path, world, extra_env={"ELPH_TEST_KPM": "1"})
    a = res[0]
    for b in res[1:]:
        assert int(b["accepted"]) == int(a["accepted"]) and np.array_equal(a["energies"], b["energies"]) and np.array_equal(a["x"], b["x"])
    assert int(a["accepted"]) == int(a["accepted_ref"]) == 1 and int(a["flag"]) == int(a["flag_ref"]) == 0
    e, er = a["energies"], a["energies_ref"]
    assert abs(e[0] - er[0]) < 1e-9 * abs(er[0]) and abs(e[1] - er[1]) < 1e-8 * abs(er[1])
    assert _rel(a["x"], a["x_ref"]) < 1e-9 and _rel(a["v"], a["v_ref"]) < 1e-8
    assert abs(float(a["iters"]) - float(a["iters_ref"])) <= 1 and float(a["iters"]) < 100     # (preconditioned at tol 1e-10: 61 iterations per solve; plain: several hundred)


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_hmc_update_bond_phonons_with_the_kpm_preconditioner(tmp_path, world):
    """The production shape of BASELINE config 5: the sharded update of the optical SSH model with the KPM preconditioner — the expansion on a
    bond-phonon handle of the whole lattice, its τ-averaged cosh / sinh of EVERY bond taken at each setup!(P) from the owners' slabs
    (elph_shard_set_bonds; update_A!, KPMPreconditioners.jl:355-381) — against the preconditioned update of ONE handle."""
    res = _run_callers("hmc", "E", tmp_path, world, extra_env={"ELPH_TEST_KPM": "1"})
    a = res[0]
    for b in res[1:]:
        assert int(b["accepted"]) == int(a["accepted"]) and np.array_equal(a["energies"], b["energies"]) and np.array_equal(a["x"], b["x"])
    assert int(a["accepted"]) == int(a["accepted_ref"]) == 1 and int(a["flag"]) == int(a["flag_ref"]) == 0
    e, er = a["energies"], a["energies_ref"]
    assert abs(e[0] - er[0]) < 1e-9 * abs(er[0]) and abs(e[1] - er[1]) < 1e-8 * abs(er[1])
    assert _rel(a["x"], a["x_ref"]) < 1e-9 and _rel(a["v"], a["v_ref"]) < 1e-8
    assert abs(float(a["iters"]) - float(a["iters_ref"])) <= 1 and float(a["iters"]) < 150     # (plain: several hundred per solve)
    assert all(int(r["ghost_dev"]) > 0 and int(r["ghost_host"]) == 0 for r in res)


@pytest.mark.parametrize("tag", ["D", "E"])
def test_ghost_rows_through_the_mailboxes_equal_the_host_staged_exchange(tmp_path, tag):
    """The ghost rows of ϕ± and of the fermion force travel device to device through the mailboxes (elph_shard_ghost_stats counts them); staged
    through the caller's all-reduce instead (ELPH_SHARD_GHOST_HOST=1) the update ends on the same bits — the exchange moves values, it adds nothing."""
    (tmp_path / "dev").mkdir()
    (tmp_path / "host").mkdir()
    dev = _run_callers("hmc", tag, tmp_path / "dev", 2)
    host = _run_callers("hmc", tag, tmp_path / "host", 2, extra_env={"ELPH_SHARD_GHOST_HOST": "1"})
    assert all(int(r["ghost_dev"]) >= 3 and int(r["ghost_host"]) == 0 for r in dev)           # ϕ± once, the force at every evaluation
    assert all(int(r["ghost_dev"]) == 0 and int(r["ghost_host"]) >= 3 for r in host)
    for a, b in zip(dev, host):
        assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["v"], b["v"]) and np.array_equal(a["energies"], b["energies"])


@pytest.mark.parametrize("world,nb", [(2, 1), (4, 1), (2, 3), (8, 1)])
def test_sharded_hmc_update_vs_one_handle(tmp_path, world, nb):
    """One HMC update of BASELINE config 4 (Holstein honeycomb L = 12, Ntau = 120) on a lattice sharded over 2 / 4 ranks —
    elph_hmc_update on the slab handle: sharded solves, own-row energies summed over the ranks, ghost rows of ϕ± and of the fermion force
    from their owners — against the update of ONE handle with the same random numbers: same decision, H, S, K and the field to 1e-9."""
    res = _run_callers("hmc", "D", tmp_path, world, per_proc=2 if world == 8 else 1, extra_env={"ELPH_TEST_NB": str(nb)})
    a = res[0]
    for b in res[1:]:
        assert int(b["accepted"]) == int(a["accepted"]) and np.array_equal(a["energies"], b["energies"]) and np.array_equal(a["x"], b["x"])
    assert int(a["accepted"]) == int(a["accepted_ref"]) == 1 and int(a["flag"]) == int(a["flag_ref"]) == 0
    e, er = a["energies"], a["energies_ref"]
    assert abs(e[0] - er[0]) < 1e-9 * abs(er[0]) and abs(e[1] - er[1]) < 1e-8 * abs(er[1])
    assert abs(e[2] - er[2]) < 1e-8 * abs(er[2]) and abs(e[3] - er[3]) < 1e-8 * abs(er[3])
    assert _rel(a["x"], a["x_ref"]) < 1e-9 and _rel(a["v"], a["v_ref"]) < 1e-8
    assert abs(float(a["iters"]) - float(a["iters_ref"])) <= 1
