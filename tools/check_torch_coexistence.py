"""Does the HIP library coexist with torch in ONE process the way `bench.py --gpus N` (N > 1) needs it to?

torch (ROCm 7.0 wheels) bundles its own libamdhip64.so; libelphgpu.so links /opt/rocm's (7.2).  Loaded in bench.py's order — torch first
(dist.Comm), the library second — the loader resolves the library's NEEDED libamdhip64.so.7 to the copy torch already mapped (same
SONAME): one runtime.  In the other order torch maps a second runtime and reports "No HIP GPUs" (seen in a pytest process).  This
script takes bench.py's order with the nccl (= RCCL) backend at world size 1 on one GPU: init_process_group, one all-reduce on the
device, then a solve through the C ABI checked against numpy — what every rank of an N-GPU run does before and around its timed region.

    python tools/check_torch_coexistence.py            # on the GPU box
"""
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

print("torch", torch.__version__, "cuda available", torch.cuda.is_available(), "devices", torch.cuda.device_count(), flush=True)
assert torch.cuda.is_available()
torch.cuda.set_device(0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(8, dtype=torch.float64, device="cuda")
dist.all_reduce(t)
torch.cuda.synchronize()
print("RCCL all_reduce at world 1:", t.sum().item(), flush=True)

from elphdynamics_amd import _lib, configs, models  # noqa: E402

lib = _lib.load()
print("library sees", lib.elph_device_count(), "device(s);", lib.elph_build_info().decode()[:90], flush=True)
maps = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l]
print("HIP runtimes mapped:", sorted(set(maps)), flush=True)
m = configs.make_model("B", tol=1e-10)
R, B = configs.rhs(m, 1)
x = np.zeros(m.Ndim)
it, res, fl = models.ldiv_(x, m, np.ascontiguousarray(B[0]))
y = np.zeros(m.Ndim)
models.mulMtM_(y, m, x)
err = np.linalg.norm(y - B[0]) / np.linalg.norm(B[0])
print(f"solve after torch+RCCL init: {it} iterations, flag {fl}, |MtM x - b|/|b| = {err:.2e}", flush=True)
assert fl == 0 and err < 1e-8
dist.barrier()
dist.destroy_process_group()
m.close()
print("COEXISTENCE OK")
