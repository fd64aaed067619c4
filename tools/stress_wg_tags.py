"""One handle, batches of changing size back to back through the workgroup-resident kernel: the meeting records are numbered per launch
and never zeroed in between (cg_wg.hip: WgCtl::epoch0), their memory is reinterpreted with every batch size — every solution must
still equal the streaming form's.  usage: python3 tools/stress_wg_tags.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import configs, models

m = configs.make_model("C", tol=1e-5)
R, B = configs.rhs(m, 50)
os.environ["ELPH_NO_WG"] = "1"
Xref = np.zeros_like(B)
itref, _, fl = models.ldiv_batched_(Xref, m, B)
assert not fl.any()
os.environ["ELPH_NO_WG"] = "0"
worst = 0.0
for rnd in range(4):
    for nr in (1, 48, 3, 24, 50, 2, 40, 8, 25):
        X = np.zeros((nr, m.Ndim))
        it, res, fl = models.ldiv_batched_(X, m, np.ascontiguousarray(B[:nr]))
        assert not fl.any(), (rnd, nr)
        err = max(np.linalg.norm(X[i] - Xref[i]) / np.linalg.norm(Xref[i]) for i in range(nr))
        dit = int(np.max(np.abs(it - itref[:nr])))
        assert err < 1e-4 and dit <= 3, (rnd, nr, err, dit)          # (another summation tree: the count moves at the knife edge)
        worst = max(worst, err)
print("stress ok: 36 batches of changing size on one handle, worst |dx|/|x| vs the streaming form", worst)
m.close()
