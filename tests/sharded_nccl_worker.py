"""nccl-backend run of the sharded solver (device-resident collectives).  On the single-GPU test box this runs with
WORLD_SIZE=1 (RCCL communicator of one rank): it exercises the zero-copy torch views of the solver's buffers, the shared
stream and the collective calls; the multi-rank data movement itself is covered by the gloo tests."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (first: libelphgpu then shares torch's HIP runtime)

from elphdynamics_amd import configs, dist, models, sharded, synth  # noqa: E402

comm = dist.Comm(backend="nccl") if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None
if comm is None:
    import torch.distributed as td
    td.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    comm = dist.Comm.__new__(dist.Comm)
    comm.rank, comm.local_rank, comm.world = 0, 0, 1
    comm.torch, comm.dist, comm.backend, comm.device = torch, td, "nccl", torch.device("cuda", 0)
m = configs.make_model("B", tol=1e-9)
E = np.exp(-m.dtau * (m.lam[:, None] * m.x.reshape(m.Nsites, m.Ltau) - m.mu[:, None])).reshape(-1)
b = synth.randn(99, m.Ndim)
s = sharded.ShardedCG(comm, m.Nsites, m.Ltau, m.neighbor_table, m.cosht, m.sinht, device=0)
assert s.dev is not None
s.update_model(E)
xs, it, done = s.solve(b, tol=1e-9, maxiter=5000, check_every=8)
x = np.zeros(m.Ndim)
it1 = models.solve_(x, m, b, tol=1e-9)
err = np.linalg.norm(xs - x) / np.linalg.norm(x)
print("sharded(nccl, world=1) iters", it, "single-handle iters", it1, "rel diff", err, "done", done)
assert done == 1 and abs(it - it1) <= 2 and err < 1e-7
s.close(); m.close()
td = comm.dist
td.destroy_process_group()
print("OK")
