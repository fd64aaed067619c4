"""Sharded single solve (elphdynamics_amd/sharded.py), world_size 2:
  * CPU (gloo): the protocol — tau-slab decomposition, zeroed-wrap / sign-flipped local expV, r-halo exchange, cross-rank
    combination of the per-slice partial sums — with a numpy/oracle stand-in for the local kernels;
  * GPU (marked gpu): the same driver on libelphgpu's step-wise entry points, two ranks sharing device 0,
    collectives staged through gloo (the box has one GPU; on a multi-GPU node the backend is nccl = RCCL).
Both are checked against an un-sharded oracle solve of the same system."""
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


@pytest.mark.gpu
def test_sharded_solve_two_ranks_one_gpu(tmp_path, oracle):
    _check(_run("gpu", tmp_path), oracle)


@pytest.mark.gpu
def test_sharded_solver_device_resident_nccl_path(tmp_path):
    """The nccl (RCCL) flavour of the driver: collectives act on zero-copy torch views of the solver's device buffers,
    kernels and collectives ordered on one stream.  One rank here (single-GPU box)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sharded_nccl_worker.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0 and "OK" in p.stdout, (p.stdout[-1500:], p.stderr[-1500:])
