"""Worker of test_lds_sync_build_is_bit_identical: prints one JSON line of result digests for the library named by ELPH_LIB
(default: the product build).  Same inputs in every run; the digests of the two builds must be equal."""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import _lib, configs, models, preconditioners as pc, synth      # noqa: E402


def dig(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


out = {"lib": os.path.basename(_lib.library_path())}
for tag in sys.argv[1:]:
    m = configs.make_model(tag, tol=1e-5)
    v = synth.randn(77, m.Ndim)
    y = np.empty(m.Ndim)
    models.mulMtM_(y, m, v)
    R, B = configs.rhs(m, 3)
    X = np.zeros_like(B)
    it, res, fl = models.ldiv_batched_(X, m, B)
    x1 = np.zeros(m.Ndim)
    it1, res1, fl1 = models.ldiv_(x1, m, np.ascontiguousarray(B[0]))
    e = {"MtMv": dig(y), "X": dig(X), "iters": [int(i) for i in it], "x1": dig(x1), "it1": int(it1)}
    if m.kind == 0:        # the KPM Chebyshev kernels (k_kpm_cheb_ri / _fast) order their slabs the same way
        P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
        pc.setup_(P, rng=np.random.default_rng(5))
        z = np.empty(m.Ndim)
        pc.kpm_ldiv_(z, P, v)
        xp = np.zeros(m.Ndim)
        itp, *_ = models.ldiv_(xp, m, np.ascontiguousarray(B[0]), P=P)
        e.update({"kpm_z": dig(z), "xp": dig(xp), "itp": int(itp)})
    out[tag] = e
    m.close()
print(json.dumps(out))
