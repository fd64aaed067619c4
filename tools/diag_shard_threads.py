#!/usr/bin/env python3
"""Diagnostic: the in-library sharded solve with several rank threads per process (tests/shard_worker.py), short timeouts."""
import os, socket, subprocess, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p
def run(case, world, per_proc, extra_env=None):
    port = free_port(); nproc = world // per_proc; out = f"/tmp/diag_{case}_{world}_{per_proc}"
    procs = []
    t0 = time.time()
    for r in range(nproc):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ELPH_FORCE_DEVICE="0", ELPH_WG_TIMEOUT_MS="8000", ELPH_RANKS_PER_PROC=str(per_proc), ELPH_SHARD_DEBUG="1", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), case, out, "1e-9"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    ok = True
    for p in procs:
        try:
            o, e = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            for q in procs: q.kill()
            print(f"{case} world={world} per_proc={per_proc}: HUNG", flush=True); return
        if p.returncode != 0:
            ok = False
            print("   stderr:", e[-600:].replace("\n", " | "))
        else:
            print("   dbg:", " | ".join(l for l in e.splitlines() if "[shard]" in l)[:400])
    print(f"{case} world={world} per_proc={per_proc} env={extra_env}: {'ok' if ok else 'FAILED'} in {time.time()-t0:.1f}s", flush=True)
for case, world, pp, env in [("sq8", 2, 2, None), ("C", 2, 2, None), ("C", 4, 2, None), ("C", 8, 2, None), ("C", 8, 2, {"GPU_MAX_HW_QUEUES": "2"}), ("C", 4, 4, {"GPU_MAX_HW_QUEUES": "8"}), ("D", 8, 2, None), ("E", 8, 2, None)]:
    run(case, world, pp, env)
