"""(round 5) The slab form (slabs.hip) for every admissible slab count of L x L square lattices, Ltau = 160, one right-hand side: us per iteration of the streaming pair (1) and of the slab form (12).  The data of the rule in elph_i_slabs_usable.  usage: python3 tools/sweep_slabs.py"""
import ctypes as C, os, sys, subprocess, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
def child(Ls, P):
    import numpy as np
    sys.path.insert(0, ROOT)
    from elphdynamics_amd import configs, lattice as lat
    from elphdynamics_amd._lib import check, dptr
    configs.CONFIGS["_slab"] = ("holstein", 1, Ls, lat.SQUARE_BONDS, 16.0, 0.1)
    m = configs.make_model("_slab", tol=1e-5, maxiter=20000)
    lib = m._lib; ms = C.c_double()
    _, B = configs.rhs(m, 1)
    out = {}
    for what in (1, 12):
        try:
            for reps in (64, 400):
                check(lib.elph_bench_prepare(m._h, 1, 1, dptr(np.ascontiguousarray(B))))
                check(lib.elph_bench_run(m._h, what, 1, reps, 0, C.byref(ms)))
            out[what] = round(1e3 * ms.value / 400, 2)
        except Exception as e:
            out[what] = str(e)[:80]
    print("RESULT", Ls, P, out, flush=True)
if len(sys.argv) > 2:      # (child: lattice size and the slab count it was started for — 0: whatever the environment says)
    child(int(sys.argv[1]), int(sys.argv[2]))
else:
    for Ls in (18, 20, 24, 28, 30, 32):
        for P in range(2, 9):
            if (Ls * Ls) % P: continue
            env = dict(os.environ, ELPH_SLABS_P=str(P), ELPH_SLABS_DEBUG="1", ELPH_WG_TIMEOUT_MS="500")
            p = subprocess.run([sys.executable, os.path.abspath(__file__), str(Ls), str(P)], env=env, capture_output=True, text=True, timeout=120)
            for l in (p.stdout + p.stderr).splitlines():
                if l.startswith("RESULT") or l.startswith("[slabs]"): print(l, flush=True)
