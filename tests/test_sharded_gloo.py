"""Sharded single solve, world_size 2, CPU (gloo): the host-spelled protocol of tests/protocol_reference.py — slab decomposition
(the product's `sharded.SpatialSlabs`), ghost exchange of r, cross-rank combination of the partial sums — with a numpy/oracle stand-in
for the local kernels, checked against an un-sharded oracle solve of the same system; the slab and team-shape arithmetic of the
in-library sharded solve at 1, 2, 4 and 8 ranks; `dist.HybridComm`.  (The in-library solve itself on the GPU: tests/test_gpu_shard.py.)"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(mode, tmp_path, world=2):
    port = _free_port()
    out = str(tmp_path / "shard")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_worker.py"), mode, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e[-3000:]
    return [np.load(out + f".rank{r}.npz") for r in range(world)]


def _check(res, oracle):
    a, b = res[0], res[1]
    assert int(a["it"]) == int(b["it"]) and int(a["done"]) == int(b["done"]) == 1
    assert np.array_equal(a["x"], b["x"])                            # every rank assembled the same solution
    N, L = int(a["N"]), int(a["Ltau"])
    om = oracle.make_model(0, N, L, a["table"], a["c"], a["s"], np.ascontiguousarray(a["E"]))
    xo, ito = oracle.cg_solve(om, np.ascontiguousarray(a["b"]), tol=1e-9, maxiter=2000)
    assert abs(int(a["it"]) - ito) <= 2
    assert np.linalg.norm(a["x"] - xo) / np.linalg.norm(xo) < 1e-7
    r = oracle.mulMTM(om, np.ascontiguousarray(a["x"])) - a["b"]
    assert np.linalg.norm(r) / np.linalg.norm(a["b"]) < 1e-8         # true residual of the sharded solution


def test_sharded_protocol_two_ranks_cpu(tmp_path, oracle):
    _check(_run("numpy", tmp_path), oracle)


# ---------------------------------------------------------------------------------------------- spatial slabs

def _check_spatial(res, oracle):
    a, b = res[0], res[1]
    for tag, halo in (("sq", [2, 2]), ("hc", [1, 1])):
        assert a[f"{tag}_halo"].tolist() == halo                               # ghost rows found by the dependency closure
        assert int(a[f"{tag}_it"]) == int(b[f"{tag}_it"]) and int(a[f"{tag}_done"]) == int(b[f"{tag}_done"]) == 1
        assert np.array_equal(a[f"{tag}_x"], b[f"{tag}_x"])
        N, L = int(a[f"{tag}_N"]), int(a[f"{tag}_Ltau"])
        om = oracle.make_model(0, N, L, a[f"{tag}_table"], a[f"{tag}_c"], a[f"{tag}_s"], np.ascontiguousarray(a[f"{tag}_E"]))
        bb = np.ascontiguousarray(a[f"{tag}_b"])
        xo, ito = oracle.cg_solve(om, bb, tol=1e-9, maxiter=2000)
        assert abs(int(a[f"{tag}_it"]) - ito) <= 2
        assert np.linalg.norm(a[f"{tag}_x"] - xo) / np.linalg.norm(xo) < 1e-7
        r = oracle.mulMTM(om, np.ascontiguousarray(a[f"{tag}_x"])) - bb
        assert np.linalg.norm(r) / np.linalg.norm(bb) < 1e-8


def test_spatial_sharded_protocol_two_ranks_cpu(tmp_path, oracle):
    """Slabs of rows of cells + ghost rows (SpatialShardedCG): square 8x8 and honeycomb 4x4, 2 gloo ranks, numpy/oracle
    stand-in for the local kernels — the slab tables, the masked inner products, the ghost-row exchange of r."""
    _check_spatial(_run("numpy-spatial", tmp_path), oracle)


def test_collective_transport_two_ranks_cpu(tmp_path, oracle):
    """sharded_rccl.CollectiveShardedSolver — the sharded solve with torch.distributed collectives as its transport (RCCL on GPUs; gloo here):
    two all-reduces (p.z, r.r) and one grouped ghost-row exchange of p per iteration, scalars and stop rule on every rank from the same
    all-reduced numbers (IterativeSolvers.jl:239-314).  World 2, the CPU oracle as the slab operator: same solution on both ranks, the
    un-sharded oracle solve's iteration count and solution, the true residual, and agreement with the host-spelled MAILBOX protocol
    (tests/protocol_reference.py) on the same system — same iteration count, solutions to 1e-9 after ~100 iterations at tol 1e-9 (other summation order of the inner
    products: partial sums per time slice there, one dot product per rank here)."""
    res = _run("collectives", tmp_path)
    _check_spatial(res, oracle)
    a = res[0]
    for tag in ("sq", "hc"):
        assert int(a[f"{tag}_it"]) == int(a[f"{tag}_itref"])
        assert np.linalg.norm(a[f"{tag}_x"] - a[f"{tag}_xref"]) / np.linalg.norm(a[f"{tag}_xref"]) < 1e-9
        assert int(a[f"{tag}_ncoll"]) == 1 + 3 * ((int(a[f"{tag}_it"]) + 3) // 4 * 4)      # start-up all-reduce + 3 per launched iteration (checked every 4)
        assert int(a[f"{tag}_it5"]) == 5 and np.array_equal(a[f"{tag}_x5"], res[1][f"{tag}_x5"])


def test_spatial_slab_tables():
    """Integer set-up of the spatial decomposition on the BASELINE lattices: ghost rows from the dependency closure of
    the fused MᵀM (2+2 for the even-aligned square lattice — only the last colour crosses the slab boundary, SURVEY §8e;
    1+1 for honeycomb), uneven splits, and the too-many-ranks error."""
    from elphdynamics_amd import lattice as lat
    from elphdynamics_amd import sharded

    def slabs(ns, L, bonds, P):
        la = lat.Lattice(ns, L, L, 1)
        raw = np.concatenate([la.calc_neighbor_table(o1, o2, d) for (o1, o2, d) in bonds], axis=0)
        cb = lat.initialize_checkerboard(raw, np.ones(raw.shape[0]), 0.1)
        return sharded.SpatialSlabs(ns, L, L, cb["table"], P), cb["table"]

    S, tab = slabs(1, 16, lat.SQUARE_BONDS, 8)
    assert [(s["R"], int(s["lo"]), int(s["hi"])) for s in S.slabs] == [(2, 2, 2)] * 8
    lt = S.local_table(3, tab)
    assert lt.min() == 1 and lt.max() == 6 * 16 and lt.shape[0] == 6 * 16 + 5 * 16      # 6 rows of x-bonds, 5 row-pairs of y-bonds
    assert np.array_equal(S.global_sites(0)[:32], np.arange(14 * 16, 16 * 16))         # rank 0's ghosts wrap to rows 14, 15
    S, _ = slabs(2, 12, lat.HONEYCOMB_BONDS, 8)
    assert [s["R"] for s in S.slabs] == [1, 2, 1, 2, 1, 2, 1, 2] and all(s["lo"] == 1 and s["hi"] == 1 for s in S.slabs)
    S, _ = slabs(1, 16, lat.SQUARE_BONDS, 3)
    assert [(s["R"], int(s["lo"]), int(s["hi"])) for s in S.slabs] == [(5, 2, 3), (5, 3, 2), (6, 2, 2)]
    with pytest.raises(ValueError):
        slabs(1, 8, lat.SQUARE_BONDS, 16)           # more ranks than rows of cells
    S1, tab1 = slabs(1, 8, lat.SQUARE_BONDS, 1)
    assert S1.slabs[0]["lo"] == 0 and len(S1.slabs[0]["bonds"]) == tab1.shape[0]


# ---- eight ranks (BASELINE.json: "1, 2, 4 and 8 GPUs"): the host-side arithmetic of the in-library sharded solve -----------------

@pytest.mark.parametrize("tag,norb,Ls,Ltau,rows8,halo", [("C", 1, 16, 160, [2] * 8, (2, 2)), ("D", 2, 12, 120, [1, 2] * 4, (1, 1)),
                                                         ("E", 1, 16, 160, [2] * 8, (2, 2))])
def test_slabs_and_team_shape_accept_eight_ranks(tag, norb, Ls, Ltau, rows8, halo):
    """SpatialSlabs (own rows + dependency-closure ghost rows) and the library's team-shape arithmetic (elph_shard_shape: host
    code, no device) take every BASELINE config at 1, 2, 4 and 8 ranks: no ELPH_E_UNSUPPORTED, ghost rows inside the neighbour."""
    import ctypes as C
    from elphdynamics_amd import _lib, sharded
    from elphdynamics_amd import lattice as lat
    lib = _lib.load()
    la = lat.Lattice(norb, Ls, Ls, 1)
    bonds = lat.SQUARE_BONDS if norb == 1 else lat.HONEYCOMB_BONDS
    raw = np.concatenate([la.calc_neighbor_table(o1, o2, d) for (o1, o2, d) in bonds], axis=0)
    cb = lat.initialize_checkerboard(raw, np.ones(raw.shape[0]), 0.1)
    for world in (1, 2, 4, 8):
        sl = sharded.SpatialSlabs(norb, Ls, Ls, cb["table"], world)
        rows = [s["R"] for s in sl.slabs]
        assert sum(rows) == Ls
        if world == 8:
            assert rows == rows8
        covered = np.zeros(la.nsites, dtype=int)
        for q in range(world):
            s, sp, sn = sl.slabs[q], sl.slabs[(q - 1) % world], sl.slabs[(q + 1) % world]
            if world > 1:
                assert (s["lo"], s["hi"]) == halo
                assert s["lo"] <= sp["R"] and s["hi"] <= sn["R"]          # the mailbox protocol: ghosts come from the ring neighbours only
            gs = sl.global_sites(q)
            covered[gs[s["lo"] * sl.row:(s["lo"] + s["R"]) * sl.row]] += 1
            assert (s["lo"] + s["R"] + s["hi"]) * sl.row <= 320           # the resident kernel's slab limit (5 sites per lane)
        assert np.all(covered == 1)                                        # every site owned exactly once
        W, G, rec, cap = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        rc = lib.elph_shard_shape(Ltau, world, C.byref(W), C.byref(G), C.byref(rec), C.byref(cap))
        assert rc == 0, lib.elph_last_error()
        assert W.value * G.value == Ltau and rec.value == world * G.value <= cap.value == 256
    # what does not fit is refused cleanly: a prime time axis has one wave per workgroup
    assert lib.elph_shard_shape(1021, 2, None, None, None, None) == -5 and b"records" in lib.elph_last_error()
    assert lib.elph_shard_shape(160, 9, None, None, None, None) == -1


def _hybrid_worker_main():
    """(run as a subprocess by the test below) two processes x three rank threads = six ranks over gloo + thread barriers"""
    from elphdynamics_amd import dist
    comm = dist.Comm(backend="gloo")

    def body(c):
        got = c.allgather_object(("rank", c.rank))
        assert got == [("rank", r) for r in range(c.world)], got
        for k in range(3):
            c.barrier()
            got = c.allgather_object(c.rank * 10 + k)
            assert got == [r * 10 + k for r in range(c.world)]
        c.close()
        return c.rank

    ranks = dist.HybridComm.spawn(comm, 3, body)
    assert ranks == [comm.rank * 3 + t for t in range(3)]
    comm.close()


def test_hybrid_comm_threads_times_processes(tmp_path):
    """dist.HybridComm: several ranks per process (threads) times several processes (gloo) — what drives eight ranks of the sharded
    solve from four processes on the one-GPU box, and a single-process multi-GPU host in production."""
    port = _free_port()
    procs = []
    code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_sharded_gloo as t; t._hybrid_worker_main()" % (ROOT, os.path.join(ROOT, "tests"))
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-3000:]


# ---- ring-closed slabs: the periodic rectangle the register-exchange forms run on is exact on the own rows ---------------------------

@pytest.mark.parametrize("norb,Ls,bonds,worlds", [(1, 16, "sq", (1, 2, 3, 4, 5, 8)), (2, 12, "hc", (1, 2, 4, 5, 8)), (1, 12, "sq", (1, 2, 3))])
def test_ring_closed_slab_matvec_is_exact_on_the_own_rows(oracle, norb, Ls, bonds, worlds):
    """SpatialSlabs(ring=True): the bonds that leave a slab's last ghost row re-enter at its first row.  z = Mᵀ(M p) of the ring slab,
    computed by the oracle's checkerboard mat-vec on the slab's own table (disordered hoppings, so a bond attached to the wrong values
    shows), equals the whole lattice's z on the slab's OWN rows — the statement the sharded solve on the GRID / HGRID forms rests on —
    and the ring did close (every site of the slab has its full coordination: a periodic rectangle)."""
    from elphdynamics_amd import lattice as lat
    from elphdynamics_amd import sharded
    bl = lat.SQUARE_BONDS if bonds == "sq" else lat.HONEYCOMB_BONDS
    la = lat.Lattice(norb, Ls, Ls, 1)
    raw = np.concatenate([la.calc_neighbor_table(o1, o2, d) for (o1, o2, d) in bl], axis=0)
    rng = np.random.default_rng(5)
    cb = lat.initialize_checkerboard(raw, 1.0 + 0.3 * rng.standard_normal(raw.shape[0]), 0.1)
    N, L = la.nsites, 3
    E = np.exp(-0.1 * rng.standard_normal((N, L)))                 # (the reference's layout: index = site * Ltau + tau)
    p = rng.standard_normal((N, L))
    glob = oracle.make_model(0, N, L, cb["table"], cb["cosht"], cb["sinht"], E.reshape(-1))
    z = oracle.mulMTM(glob, p.reshape(-1)).reshape(N, L)
    coord = 4 if bonds == "sq" else 3
    for world in worlds:
        S = sharded.SpatialSlabs(norb, Ls, Ls, cb["table"], world, ring=True)
        for q in range(world):
            s = S.slabs[q]
            lt = S.local_table(q, cb["table"])
            gs = S.global_sites(q)
            n = gs.size
            if True:
                assert s.get("ring_next", -1) >= 0 or s["rows"].size == Ls, (world, q)
                deg = np.bincount(lt.reshape(-1) - 1, minlength=n)
                if s["rows"].size > 2:                                                  # (two-row rings double their y-bonds)
                    assert np.all(deg == coord), (world, q, deg.min(), deg.max())
            loc = oracle.make_model(0, n, L, lt, cb["cosht"][s["bonds"]], cb["sinht"][s["bonds"]], np.ascontiguousarray(E[gs]).reshape(-1))
            zl = oracle.mulMTM(loc, np.ascontiguousarray(p[gs]).reshape(-1)).reshape(n, L)
            own = slice(s["lo"] * S.row, (s["lo"] + s["R"]) * S.row)
            err = np.max(np.abs(zl[own] - z[gs[own]])) / np.max(np.abs(z))
            assert err < 1e-14, (world, q, err)
