import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from elphdynamics_amd import configs, preconditioners as pc
m = configs.make_model("C", tol=1e-5)
P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
pc.setup_(P, rng=np.random.default_rng(7))
o = np.sort(np.asarray(P.orders))[::-1]
print("orders", o.tolist(), "sum", o.sum(), "n>=2", (o >= 2).sum())
