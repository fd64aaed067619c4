"""world_size-2 CPU test (gloo) of the N>1 path of bench.py: rank/seed assignment, barrier-bracketed timing,
MAX over ranks of the time and SUM of the work (SURVEY.md §8e replica mode: no data-path collective)."""
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    from elphdynamics_amd import dist, synth
    comm = dist.Comm(backend="gloo")
    assert comm.world == 2 and comm.rank in (0, 1)
    seed = comm.chain_seed(synth.SEED_FIELDS)
    x = synth.phonon_field(4, 5, 1.0, 0.1, seed=seed)            # each rank = its own chain (different field)
    def run_steps(k):
        time.sleep(0.05 * (1 + comm.rank))                         # rank 1 is slower: MAX must pick it up
        return 2.0 * 3 * k                                          # mat-vecs this rank performed (2 * nrhs * steps)
    elapsed, work = dist.timed_steps(comm, run_steps, 10)
    # every rank sees the same reduced numbers
    assert abs(work - 2 * 2.0 * 3 * 10) < 1e-12
    assert elapsed >= 0.1 - 1e-3
    assert comm.max(comm.rank) == 1.0 and comm.sum(1.0) == 2.0
    print(json.dumps({"rank": comm.rank, "seed": seed, "x0": float(x[0]), "elapsed": elapsed, "work": work}), flush=True)
    comm.close()
""") % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_replica_path(tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            raise
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    a, b = sorted(outs, key=lambda d: d["rank"])
    assert a["seed"] != b["seed"] and a["x0"] != b["x0"]          # independent chains
    assert a["elapsed"] == b["elapsed"] and a["work"] == b["work"] == 120.0
    assert a["elapsed"] >= 0.1 - 1e-3                               # MAX over ranks (rank 1 slept 0.1 s)
