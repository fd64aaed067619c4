#!/usr/bin/env python3
"""Wall time of HMC updates of `nch` chains in lockstep with the KPM preconditioner (bench.py's hmc_chain_update_ms_64chains), for a
kernel-time breakdown under rocprofv3:
    rocprofv3 --kernel-trace --stats -d gpurun_out/hmc64 -- python3 tools/time_hmc_chains.py C 64 4"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import configs, hmc as ehmc, preconditioners as pc, synth
tag = sys.argv[1] if len(sys.argv) > 1 else "C"
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
mh = configs.make_model(tag, tol=1e-5, maxiter=20000)
fah = pc.FourierAccelerator(mh)
pc.update_M_(fah, mh, 0.0, np.inf, 1.0, 0.1)
Hh = ehmc.HybridMonteCarlo(mh, fah, dt=0.01, tr=0.1, alpha=0.0, Nb=1, nchains=nch)
Hh.X[:] = np.stack([synth.phonon_field(mh.Nph, mh.Ltau, mh.beta, mh.dtau, seed=100 + 17 * c) for c in range(nch)])
Hh.push_()
Ph = pc.SymmetricKPMPreconditioner(mh, 20, 0.05, 1.0, 1.0)
Hh.device_rng_(3)
ehmc.update_chains_(mh, Hh, fah, Ph)
ts = []
for i in range(n):
    t0 = time.perf_counter()
    acc, its = ehmc.update_chains_(mh, Hh, fah, Ph)
    ts.append(time.perf_counter() - t0)
    print(f"update {i}: {1e3*ts[-1]:.2f} ms = {1e3*ts[-1]/nch:.3f} ms per chain; iterations per solve {np.mean(its):.1f}; accepted {np.mean(acc):.2f}", flush=True)
mh.close()
