#!/usr/bin/env python3
"""The first and the second solve of the same right-hand side on a FRESH handle must be the same bits (and the eps histories equal).
Found with it (round 3): thread 0's late initialisation of a word that aliased wave 2's z.z partial of the first iteration — one
first solve in sixteen of the bond-phonon DPP form lost its first beta and took one iteration more.
usage: [HIST=0] python3 tools/check_first_solve_bits.py [E|C|D] [rounds]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import configs, models, synth
tag, nchains, per = (sys.argv[1] if len(sys.argv) > 1 else "E"), 4, 2
hist = os.environ.get("HIST", "1") == "1"
bad = 0
for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    m = configs.make_model(tag, tol=1e-5)
    if m.kind == models.SSH:
        X = np.stack([m.x * (0.55 + 0.9 * c / nchains) * (1.0 + 0.2 * synth.randn(5000 + c, m.Ndof)) for c in range(nchains)])
    else:
        X = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=5000 + c) for c in range(nchains)])
    nrhs = nchains * per
    B = np.stack([synth.randn(7000 + r, m.Ndim) for r in range(nrhs)])
    models.update_model_chains_(m, X)
    Xs = np.zeros_like(B)
    it, res, fl = models.ldiv_batched_(Xs, m, B)
    for r in range(nrhs):
        m1 = configs.make_model(tag, tol=1e-5)
        m1.x[:] = X[r % nchains]
        models.update_model_(m1)
        outs = []
        for rep in range(2):
            x1 = np.zeros(m.Ndim)
            if hist:
                o = models.solve_(x1, m1, np.ascontiguousarray(B[r]), history=True)
                outs.append((x1, o[0], o[1].copy()))
            else:
                o = models.ldiv_(x1, m1, np.ascontiguousarray(B[r]))
                outs.append((x1, o[0], None))
        if outs[0][1] != outs[1][1] or not np.array_equal(outs[0][0], outs[1][0]):
            bad += 1
            msg = f"round {k} rhs {r}: first {outs[0][1]} second {outs[1][1]} batch {int(it[r])}"
            if hist:
                h0, h1 = outs[0][2], outs[1][2]
                n = min(len(h0), len(h1))
                d = np.nonzero(h0[:n] != h1[:n])[0]
                msg += f" | histories differ first at iteration {d[0] if len(d) else None} of {n}: {h0[d[0]] if len(d) else ''} vs {h1[d[0]] if len(d) else ''}; count of differing entries {len(d)}; tail first {h0[-3:]} second {h1[-3:]}"
            print(msg, flush=True)
        m1.close()
    m.close()
    print("round", k, "done; mismatches so far", bad, flush=True)
