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


SPATIAL_WORKER = textwrap.dedent("""
    import json, sys, time
    sys.path.insert(0, %r)
    import bench
    from elphdynamics_amd import dist
    comm = dist.Comm(backend="gloo")

    class FakeSolver:                       # stands in for sharded.ShardedSolver (no GPU here): what the record reads from it
        sl = {"lo": 2, "hi": 2}
        Nloc = 96
        selftest_us = [0.4, 1.3]
        def iterate(self, b, k):
            time.sleep(0.001 * (1 + comm.rank))
            return 0.006 * k * (1 + comm.rank)      # ms of the launch: rank 1 is slower, MAX must pick it up
        def close(self):
            pass

    def factory(tag, comm_):
        if tag == "D":
            raise ValueError("ghost rows reach beyond the neighbouring rank: use fewer ranks")
        return FakeSolver(), None, 256, 160, 512, "stand-in slabs"

    rec = bench.spatial_records(comm, 200, factory=factory)
    print(json.dumps({"rank": comm.rank, "rec": rec}), flush=True)
    comm.close()
""") % ROOT


def test_spatial_sub_record_schema_two_ranks_gloo(tmp_path):
    """bench.py's `spatial` sub-record (ONE solve of C, D, E sharded over the ranks) at world 2 over gloo with a stand-in solver: the
    keys the north_star's curve needs, MAX over ranks of the device time, the rank count the process group sees, identical records
    on every rank, and a config that does not fit recorded with its reason instead of killing the line."""
    import json
    script = tmp_path / "worker.py"
    script.write_text(SPATIAL_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=240)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    a, b = sorted(outs, key=lambda d: d["rank"])
    assert a["rec"] == b["rec"]
    rec = a["rec"]
    assert set(rec) >= {"C", "D", "E"}
    c = rec["C"]
    for key in ("config", "ranks", "rccl_ranks", "dist_backend", "iterations", "us_per_iteration_device", "us_per_iteration_host",
                "matvecs_per_sec", "cg_iters_per_sec", "bound", "achieved", "peak", "unit", "frac", "peer_access",
                "selftest_us_per_round", "scaling", "devices"):
        assert key in c, key
    assert c["ranks"] == 2 and c["rccl_ranks"] == 2 and c["dist_backend"] == "gloo" and c["scaling"] == "strong"
    assert abs(c["us_per_iteration_device"] - 12.0) < 1e-9                 # MAX over ranks: 0.006 ms x 2 per iteration
    assert abs(c["matvecs_per_sec"] - 2.0 / 12e-6) < 1e-3
    assert c["peak"] == 2 * 78.6 and 0.0 < c["frac"] < 1.0 and c["unit"] == "TFLOP/s"
    assert "error" in rec["D"] and "ghost rows" in rec["D"]["error"] and rec["D"]["ranks"] == 2


def test_bench_self_launches_its_ranks_or_refuses():
    """`python bench.py --gpus 2` WITHOUT a launcher: bench.py starts the two rank processes itself (torch.distributed.run on 127.0.0.1,
    before touching a GPU) and the line says n_gpus = 2 — never a one-GPU line for a two-GPU request.  Here on the CPU over gloo with
    ELPH_BENCH_DRY=1 (launch / rendezvous / reduction path only: value is null and the line is marked a dry run)."""
    import json
    env = dict(os.environ, ELPH_BENCH_DRY="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                     # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["value"] is None and d["self_launched"] is True
    assert d["work_all_ranks"] == 2 * 2.0 * 288 * 3 and d["elapsed_max"] >= 0.02 - 1e-3      # SUM of the work, MAX of the time (rank 1 sleeps 0.02 s, rank 0 0.01 s)
    assert "starting 2 rank process(es)" in p.stderr
    # refusal instead of a silent one-GPU measurement when self-launch is switched off
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, ELPH_BENCH_NO_SELF_LAUNCH="1"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith("{")] and "needs 2 ranks" in p.stderr
    # a launcher that started the wrong number of ranks is refused too (was: accepted when WORLD_SIZE == 1)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0",
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), ELPH_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=30)
    assert p.returncode != 0
