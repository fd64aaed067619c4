#!/usr/bin/env python3
"""Wall time per iteration of the KPM-preconditioned sharded solve (elph_shard_solve_kpm) with 1, 2, 4, 8 ranks sharing one GPU."""
import os, socket, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p
for case in (sys.argv[1:] or ["C"]):
    for world, pp in ((1, 1), (2, 1), (4, 1), (8, 2)):
        port = free_port(); nproc = world // pp
        procs = []
        for r in range(nproc):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       ELPH_FORCE_DEVICE="0", ELPH_TEST_KPM="1", ELPH_TEST_TIMING="1", ELPH_RANKS_PER_PROC=str(pp))
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), case, f"/tmp/tsk_{case}_{world}", "1e-5"],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        for p in procs:
            o, e = p.communicate(timeout=600)
            for l in o.splitlines():
                if l.startswith("TIMING"): print(l, flush=True)
            if p.returncode != 0: print("FAILED", e[-400:])
