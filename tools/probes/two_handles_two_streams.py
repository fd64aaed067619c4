"""(round 6 probe) the two-stream preconditioned iteration on a handle created AFTER another handle has used its own two streams: HIP deals streams onto a
few hardware queues per process — does the second handle's pair still overlap?   python tools/probes/two_handles_two_streams.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from elphdynamics_amd import _lib, configs, models, preconditioners as pc, synth
lib = _lib.load()

def prep(tag, nrhs, nch):
    m = configs.make_model(tag, tol=1e-5)
    if nch > 1:
        X = np.stack([synth.phonon_field(m.Nph, m.Ltau, m.beta, m.dtau, seed=100 + c) for c in range(nch)])
        models.update_model_chains_(m, X)
    P = pc.SymmetricKPMPreconditioner(m, 20, 0.05, 1.0, 1.0)
    (pc.setup_chains_ if nch > 1 else pc.setup_)(P, rng=np.random.default_rng(7))
    B = np.ascontiguousarray(np.stack([synth.randn(300 + r, m.Ndim) for r in range(nrhs)]))
    _lib.check(lib.elph_bench_prepare(m._h, 3, nrhs, _lib.dptr(B)))
    return m, P

def t(m, nrhs, what):
    ms = C.c_double()
    _lib.check(lib.elph_bench_prepare(m._h, 3, nrhs, None))
    _lib.check(lib.elph_bench_run(m._h, what, nrhs, 32, 0, C.byref(ms)))
    _lib.check(lib.elph_bench_run(m._h, what, nrhs, 160, 0, C.byref(ms)))
    return 1e3 * ms.value / 160

order = sys.argv[1] if len(sys.argv) > 1 else "CX"
keep = []
for tag in order:
    tg, nrhs, nch = ("C", 288, 144) if tag == "C" else ("X32", 72, 1)
    m, P = prep(tg, nrhs, nch)
    print(f"{tg}: one stream {t(m, nrhs, 3):.1f} us, two streams {t(m, nrhs, 11):.1f} us  (handles alive before this one: {len(keep)})", flush=True)
    keep.append((m, P))
