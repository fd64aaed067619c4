#!/usr/bin/env python3
"""Wall time of single-chain HMC updates with the KPM preconditioner (bench.py's hmc_update_ms_1chain), for a kernel-time breakdown
under rocprofv3:  rocprofv3 --kernel-trace --stats -d gpurun_out/hmc_prof -- python3 tools/time_hmc_update.py C 10"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import configs, hmc as ehmc, preconditioners as pc
tag = sys.argv[1] if len(sys.argv) > 1 else "C"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mh = configs.make_model(tag, tol=1e-5, maxiter=20000)
fah = pc.FourierAccelerator(mh)
pc.update_M_(fah, mh, 0.0, np.inf, 1.0, 0.1)
Hh = ehmc.HybridMonteCarlo(mh, fah, dt=0.01, tr=0.1, alpha=0.0, Nb=1, nchains=1)
Ph = pc.SymmetricKPMPreconditioner(mh, 20, 0.05, 1.0, 1.0)
Hh.device_rng_(3)
ehmc.update_(mh, Hh, fah, Ph, pull=False)
ts = []
for i in range(n):
    t0 = time.perf_counter()
    acc, its = ehmc.update_(mh, Hh, fah, Ph, pull=False)
    ts.append(time.perf_counter() - t0)
print(f"{tag}: {n} updates, ms per update min {1e3*min(ts):.2f} median {1e3*sorted(ts)[n//2]:.2f}; iterations per solve {np.mean(its):.1f}; accepted {acc}")
mh.close()
